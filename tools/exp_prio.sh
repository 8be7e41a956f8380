set -e
cd /root/repo
export TMPDIR=/tmp
L=gpurun_out/exp_march_first_trip.log
: > $L
LIBS="build_ab/lib_7318.so raytracing.jl_amd/csrc/librt_segmentize.so"
timeout -k 10 500 python tools/ab.py --libs $LIBS --reps 3 >> $L 2>&1
timeout -k 10 500 python tools/ab.py --mesh bwr_like.msh --nazim 64 --delta 2e-3 --libs $LIBS --reps 2 >> $L 2>&1
timeout -k 10 500 python tools/ab.py --nohash --mesh bwr_like.msh --nazim 128 --delta 5e-4 --libs $LIBS --reps 2 >> $L 2>&1
timeout -k 10 500 python tools/ab.py --nazim 32 split=0 --libs $LIBS --reps 2 >> $L 2>&1
grep -v amdgpu.ids $L | sed 's/records sha \([0-9a-f]*\).*/sha \1/; s/ volumes 0.0000//' | cut -c1-330
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_multi.py tests/test_c_abi_harness.py -x -q -m gpu 2>&1 | tail -3

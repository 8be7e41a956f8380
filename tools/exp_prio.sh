# round 5: k_materialise_lin with the one-group fast path and the table loads spread over the waves (A/B against the committed library)
set -e
cd /root/repo
export TMPDIR=/tmp
L=gpurun_out/exp_lin_header.log
: > $L
git_lib=build_ab/lib_before.so
LIBS="$git_lib raytracing.jl_amd/csrc/librt_segmentize.so"
timeout -k 10 500 python tools/ab.py --libs $LIBS --reps 3 >> $L 2>&1
timeout -k 10 500 python tools/ab.py --mesh bwr_like.msh --nazim 64 --delta 2e-3 --libs $LIBS --reps 2 >> $L 2>&1
timeout -k 10 500 python tools/ab.py --nohash --mesh bwr_like.msh --nazim 128 --delta 5e-4 --libs $LIBS --reps 2 >> $L 2>&1
echo "== stamps C3 / C5" >> $L
RT_SEGMENTIZE_LIB=build_ab/lib_lintiming.so timeout -k 10 300 python tools/ab.py --what calls --calls 16 >> $L 2>&1
RT_SEGMENTIZE_LIB=build_ab/lib_lintiming.so timeout -k 10 300 python tools/ab.py --what calls --calls 16 --mesh bwr_like.msh --nazim 128 --delta 5e-4 >> $L 2>&1
grep -v amdgpu.ids $L | sed 's/records sha \([0-9a-f]*\).*/sha \1/; s/ volumes 0.0000//' | cut -c1-330
timeout -k 10 600 python -m pytest tests/test_gpu_materialise_lin.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3

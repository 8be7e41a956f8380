#!/bin/bash
# round 4, run 6: version D with the certified fill_volumes (shallow crossings tallied exactly by k_materialise)
set -o pipefail
mkdir -p gpurun_out/r04
python tools/r04_debug.py 32 5e-3 2>&1 | grep "volumes max\|offsets equal" | cut -c1-120
python tools/r04_debug.py 32 5e-3 test_tally_tau=-1 2>&1 | grep "volumes max" | cut -c1-120
python tools/r04_debug.py 32 5e-3 test_tally_tau=20000000 2>&1 | grep "volumes max" | cut -c1-120
python tools/r04_voldev.py 2>&1 | grep -v libdrm | awk '{print $1,$2,$4,$6,$8,"maxdev",$12,"n>1e-10",$16}' | column -t > gpurun_out/r04/voldev.log; awk '$8+0 > 2e-11' gpurun_out/r04/voldev.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_sweep.py tests/test_gpu_walk_regime.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04/gpu_tests_tp_d.log
L=$PWD/build_ab
for rep in 1 2; do
 for lib in librt_base.so librt_tp_d.so; do
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py pincell.msh 128 1e-3 2>&1 | tail -1
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 64 2e-3 2>&1 | tail -1
  AB_NOHASH=1 RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 128 5e-4 2>&1 | tail -1
 done
done | tee gpurun_out/r04/exp_tp_d.log

#!/bin/bash
# round 4, run 5: two-phase march, version D (materialise one unit per workgroup; fill_volumes back in the march): parity + timing
set -o pipefail
mkdir -p gpurun_out/r04
python tools/r04_debug.py 32 5e-3 2>&1 | grep -v "^unit\|libdrm" | cut -c1-200 | head -12
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_sweep.py tests/test_gpu_walk_regime.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04/gpu_tests_tp_d.log
L=$PWD/build_ab
for rep in 1 2; do
 for lib in librt_base.so librt_tp_d.so; do
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py pincell.msh 128 1e-3 2>&1 | tail -1
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 64 2e-3 2>&1 | tail -1
  AB_NOHASH=1 RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 128 5e-4 2>&1 | tail -1
 done
done | tee gpurun_out/r04/exp_tp_d.log

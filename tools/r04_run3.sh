#!/bin/bash
# round 4, run 3: k_materialise version C (16-row passes): parity on the core tests, then shapes (teams per workgroup) x register budgets
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_sweep.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04/gpu_tests_tp_c.log
L=$PWD/build_ab
{
RT_SEGMENTIZE_LIB=$L/librt_base.so timeout -k 10 300 python tools/exp_march_ab.py pincell.msh 128 1e-3 2>&1 | tail -1
RT_SEGMENTIZE_LIB=$L/librt_base.so timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 64 2e-3 2>&1 | tail -1
AB_NOHASH=1 RT_SEGMENTIZE_LIB=$L/librt_base.so timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 128 5e-4 2>&1 | tail -1
for lib in librt_tp_c_occ4.so librt_tp_c_occ3.so; do
 for tm in 0 1 2 3 4; do
  echo "== $lib mat_teams=$tm"
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py pincell.msh 128 1e-3 mat_teams=$tm 2>&1 | tail -1
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 64 2e-3 mat_teams=$tm 2>&1 | tail -1
  AB_NOHASH=1 RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 128 5e-4 mat_teams=$tm 2>&1 | tail -1
 done
done
} | tee gpurun_out/r04/exp_tp_c_shapes.log

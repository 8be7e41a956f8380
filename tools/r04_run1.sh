#!/bin/bash
# round 4, run 1: decision-only march probe (knock-out build) against the current library, same box; C5 oracle-sample test
set -o pipefail
mkdir -p gpurun_out/r04
L=$PWD/build_ab
for rep in 1 2; do
 for lib in librt_base.so librt_ko.so; do
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py pincell.msh 128 1e-3 2>&1 | tail -1
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 64 2e-3 2>&1 | tail -1
  AB_NOHASH=1 RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 128 5e-4 2>&1 | tail -1
 done
done | tee gpurun_out/r04/exp_knockout_arith.log
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py -x -q -s -k "uid_sample" 2>&1 | tail -15 | tee gpurun_out/r04/c5_uid_sample_test.log

#!/bin/bash
# round 4, run 2: two-phase march (codes + k_materialise), version A: parity tests, then same-box timing against the round-3 library
set -o pipefail
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r04/gpu_tests_two_phase_a.log
L=$PWD/build_ab
for rep in 1 2; do
 for lib in librt_base.so librt_two_phase_a.so; do
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py pincell.msh 128 1e-3 2>&1 | tail -1
  RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 64 2e-3 2>&1 | tail -1
  AB_NOHASH=1 RT_SEGMENTIZE_LIB=$L/$lib timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 128 5e-4 2>&1 | tail -1
 done
done | tee gpurun_out/r04/exp_two_phase_a.log

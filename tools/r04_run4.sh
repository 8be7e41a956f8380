#!/bin/bash
# round 4, run 4: k_materialise persistent against one unit per workgroup (experiment "mat_np")
L=$PWD/build_ab
{
for opt in "mat_np=0" "mat_np=1 mat_teams=1" "mat_np=1 mat_teams=2" "mat_np=0 mat_teams=1" ; do
  echo "== $opt"
  RT_SEGMENTIZE_LIB=$L/librt_tp_c2.so timeout -k 10 300 python tools/exp_march_ab.py pincell.msh 128 1e-3 $opt 2>&1 | tail -1
  RT_SEGMENTIZE_LIB=$L/librt_tp_c2.so timeout -k 10 300 python tools/exp_march_ab.py bwr_like.msh 64 2e-3 $opt 2>&1 | tail -1
done
} | tee gpurun_out/r04_np.log

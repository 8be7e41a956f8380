#!/bin/bash
# A/B of library builds on ONE box: tools/ab.sh "<gpu_modes args>" libA.so libB.so ...   (run under gpurun)
ARGS="$1"; shift
for rep in 1 2 3; do
  for L in "$@"; do
    echo -n "$(basename $L) : "
    RT_SEGMENTIZE_LIB=$PWD/$L python tools/gpu_modes.py $ARGS 2>&1 | tail -1 | sed -e "s/np.float64(//g; s/)//g" | cut -c1-200
  done
done

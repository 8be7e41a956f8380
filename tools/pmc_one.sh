#!/bin/bash
# one PMC pass: tools/pmc_one.sh <out_subdir> "<counters>" [bench args]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; PMC="$2"; shift; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pass1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass1.log 2>&1

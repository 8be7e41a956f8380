#!/bin/bash
# one PMC pass over an arbitrary python script: tools/pmc_cmd.sh <out_subdir> "<counters>" script.py [args...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; PMC="$2"; shift; shift
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pass1 -- python3 $SCRIPT "$@" > $OUT/pass1.log 2>&1

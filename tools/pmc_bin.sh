#!/bin/bash
# one PMC pass over a binary: tools/pmc_bin.sh <out_subdir> "<counters>" <binary relative to repo> [args...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; PMC="$2"; shift; shift
BIN=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pass1 -- $BIN "$@" > $OUT/pass1.log 2>&1

#!/bin/bash
# The GPU fuzz of a round's FINAL library in one gpurun call: the library's sha256 first (the log's first line is what bench.py reports as
# config.library_sha256), then tools/fuzz_many.py over [first, first + count) — six modes per mesh —, optionally with FUZZ_TINY / FUZZ_SHUFFLE.
#   usage (GPU box): bash tools/fuzz_final.sh <first_seed> <count> <logfile under gpurun_out/>
set -u
R=$GRAFT_REPO_ROOT
LOG=$R/gpurun_out/$3
mkdir -p "$(dirname "$LOG")"
sha256sum $R/raytracing.jl_amd/csrc/librt_segmentize.so | sed "s#$R/##" > "$LOG"
cd $R && timeout -k 10 ${FUZZ_LIMIT:-1100} python tools/fuzz_many.py $1 $2 >> "$LOG" 2>&1
rc=$?
tail -2 "$LOG"
exit $rc

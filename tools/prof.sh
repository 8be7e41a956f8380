#!/bin/bash
# Everything profiles/rNN/ holds, in one gpurun call:  bash tools/prof.sh <tag>      -> gpurun_out/<tag>/{c3,c5,sweep}/...
#   per workload: rocprofv3 --kernel-trace --stats summary of `bench.py` (kernel_stats.csv), the PMC passes of the same command
#   (pmc_summary.json / .txt, sha256 of the library recorded), the un-profiled bench line; rt_sweep: stats + PMC of tools/ab.py
#   --what sweep; one call's kernel timeline (trace_one_call.txt).
set -u
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
stats() {  # stats <subdir> <program> args...
  local D=$R/gpurun_out/$TAG/$1; shift
  mkdir -p $D
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- "$@" > $D/stats.log 2>&1 )
  find $D/stats -name "*kernel_stats.csv" -exec cp {} $D/kernel_stats.csv \;
  rm -rf $D/stats
}
B="--steps 10 --warmup 2 --no-cpu-baseline --no-concurrent --no-extras"
stats c3 python3 $R/bench.py $B
bash $R/tools/pmc.sh $TAG/c3 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-extras
( cd $R && python3 bench.py > gpurun_out/$TAG/c3/bench.json 2> gpurun_out/$TAG/c3/bench.log )
stats c5 python3 $R/bench.py --workload c5 $B
bash $R/tools/pmc.sh $TAG/c5 python3 $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-extras
stats sweep python3 $R/tools/ab.py --what sweep
bash $R/tools/pmc.sh $TAG/sweep python3 $R/tools/ab.py --what sweep
bash $R/tools/trace_one_call.sh $TAG/trace > $R/gpurun_out/$TAG/trace_one_call.txt 2>&1
find $R/gpurun_out/$TAG -name "*.db" -delete
head -8 $R/gpurun_out/$TAG/c3/kernel_stats.csv | cut -c1-150; head -6 $R/gpurun_out/$TAG/c5/kernel_stats.csv | cut -c1-150; tail -12 $R/gpurun_out/$TAG/trace_one_call.txt

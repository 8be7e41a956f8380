#!/bin/bash
# rocprofv3 kernel stats of tools/ab.py --what calls with rt_set_option pairs:  bash tools/kstats.sh <tag> [ab.py args...]  -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
D=$R/gpurun_out/$TAG.d
mkdir -p $D
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools/ab.py --what calls --calls 40 "$@" > $D/stats.log 2>&1 )
find $D/stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats.csv \;
tail -2 $D/stats.log
rm -rf $D
cut -d, -f1-4,6-8 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160 | head -12

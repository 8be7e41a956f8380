#!/bin/bash
# rocprofv3 kernel stats of any python command:  bash tools/kstats_cmd.sh <tag> script.py [args...]  -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
D=$R/gpurun_out/$TAG.d
mkdir -p $D
SCRIPT=$1; shift
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/$SCRIPT "$@" > $D/stats.log 2>&1 )
find $D/stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats.csv \;
grep -v "rocprofv3\|amdgpu.ids" $D/stats.log | tail -3
rm -rf $D
python3 - $R/gpurun_out/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('rocclr', 'slot_arr', 'prologue')): continue
    print(f"  {r['Name'].split('(')[0][:58]:58s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
PY

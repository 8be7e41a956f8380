#!/bin/bash
# rocprofv3 kernel trace of a few rt_segmentize calls (tools/ab.py --what calls: default options, no HIP events between the kernels): per-kernel
# start / end of the last two calls, to see overlap and the gap between calls.
# usage (GPU box): bash tools/trace_one_call.sh <out_subdir> [ab.py args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/ab.py --what calls --calls 6 "$@" > $OUT/trace.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/trace/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last two calls = from the last but one k_march on
marches = [i for i, r in enumerate(rows) if "k_march" in r["Kernel_Name"]]
idx = marches[-2] if len(marches) > 1 else marches[-1]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print("%-60s start %8.1f us  end %8.1f us  (%.1f us)  queue %s" % (r["Kernel_Name"][:60], (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?")))
PY

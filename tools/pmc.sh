#!/bin/bash
# rocprofv3 PMC passes (counters only + kernel-trace; one small group per pass) of a command, summarised per kernel:
#   tools/pmc.sh <out_subdir under gpurun_out> <program> [args]       e.g.  tools/pmc.sh r04/c3 python3 $GRAFT_REPO_ROOT/bench.py --steps 3 ...
# (run under gpurun; the program itself follows `--`: never env / bash -c, see the brief's rocprofv3 note)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
# (the command runs from /tmp: give scripts by absolute path, e.g. $GRAFT_REPO_ROOT/bench.py)
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $OUT $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*.db" -delete
rm -rf $OUT/pass*/

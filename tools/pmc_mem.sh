#!/bin/bash
# Memory-path PMC passes (round 5: where does a record kernel's time go between the CU and HBM?) of a command, summarised per kernel:
#   tools/pmc_mem.sh <out_subdir under gpurun_out> <program> [args]
# Counters only + kernel-trace, two or three counters per pass (larger TCP / TA groups did not finish on this pool), 100 s per pass.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $OUT > $OUT/pmc_mem_summary.txt 2>&1
find $OUT -name "*.db" -delete
rm -rf $OUT/pass*/

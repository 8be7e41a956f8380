#!/bin/bash
# PMC passes of plain rt_segmentize calls (tools/exp_calls.py) for the library in RT_SEGMENTIZE_LIB:
#   tools/pmc_lib.sh <out_subdir> [exp_calls args]      (run under gpurun; counters only, one small group per pass)
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES_EQ_64 SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/tools/exp_calls.py "$@" > $OUT/pass$i.log 2>&1
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py gpurun_out/$(basename $OUT) > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete

#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 10 100 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*.db" -delete
rm -rf $OUT/pass*/

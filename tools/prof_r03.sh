# Round-3 profiling on the GPU box (run under gpurun from the repo root): everything profiles/r03/ holds.
#   C3 (bench.py default): rocprofv3 kernel stats, PMC passes, the un-profiled bench line        -> gpurun_out/<tag>/        (tools/prof_round.sh)
#   C5 on one GPU (bench.py --workload c5): kernel stats + PMC passes                             -> gpurun_out/<tag>/c5/
#   rt_sweep at C3 (tools/exp_sweep_one.py): kernel stats + two PMC passes                        -> gpurun_out/<tag>/sweep/
#   one call's kernel timeline without HIP events (tools/trace_one_call.sh)                       -> gpurun_out/<tag>/trace/
# usage: bash tools/prof_r03.sh <tag>
set -u
TAG=${1:-r03prof}
R=$GRAFT_REPO_ROOT
bash $R/tools/prof_round.sh $TAG || exit 1
export TMPDIR=/tmp
C5="--workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-concurrent --no-extras"
mkdir -p $R/gpurun_out/$TAG/c5 $R/gpurun_out/$TAG/sweep
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/c5/stats -- python3 $R/bench.py $C5 > $R/gpurun_out/$TAG/c5/stats.log 2>&1 || exit 1
find $R/gpurun_out/$TAG/c5/stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$TAG/c5/kernel_stats.csv \;
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/c5/pmc/pass$i -- python3 $R/bench.py $C5 > $R/gpurun_out/$TAG/c5/pmc_pass$i.log 2>&1 || exit 1
done
cd $R && python3 tools/pmc_summary.py gpurun_out/$TAG/c5/pmc gpurun_out/$TAG/c5/pmc_summary.json > gpurun_out/$TAG/c5/pmc_summary.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/sweep/stats -- python3 $R/tools/exp_sweep_one.py > $R/gpurun_out/$TAG/sweep/stats.log 2>&1 || exit 1
find $R/gpurun_out/$TAG/sweep/stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$TAG/sweep/kernel_stats.csv \;
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/sweep/pmc/pass$i -- python3 $R/tools/exp_sweep_one.py > $R/gpurun_out/$TAG/sweep/pmc_pass$i.log 2>&1 || exit 1
done
cd $R && python3 tools/pmc_summary.py gpurun_out/$TAG/sweep/pmc gpurun_out/$TAG/sweep/pmc_summary.json > gpurun_out/$TAG/sweep/pmc_summary.txt 2>&1
bash tools/trace_one_call.sh $TAG/trace > gpurun_out/$TAG/trace_one_call.txt 2>&1
find gpurun_out/$TAG -name "*.db" -delete
tail -8 gpurun_out/$TAG/trace_one_call.txt; head -6 gpurun_out/$TAG/c5/kernel_stats.csv | cut -c1-160; head -5 gpurun_out/$TAG/sweep/kernel_stats.csv | cut -c1-160

# One profiling round on the GPU box (run under gpurun from the repo root): rocprofv3 kernel stats, PMC passes and the
# un-profiled bench line of the same build -> gpurun_out/<tag>/ ; copy what is to be kept into profiles/rNN/.
# usage: bash tools/prof_round.sh <tag>
set -u
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-concurrent --no-extras > $R/gpurun_out/$TAG/stats.log 2>&1
find $R/gpurun_out/$TAG/stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$TAG/kernel_stats.csv \;
cd $R && bash tools/pmc_passes.sh $TAG/pmc --no-concurrent --no-extras > gpurun_out/$TAG/pmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/$TAG/pmc gpurun_out/$TAG/pmc_summary.json > gpurun_out/$TAG/pmc_summary.txt 2>&1
rm -rf gpurun_out/$TAG/pmc/pass*/*/*.db gpurun_out/$TAG/stats/*/*.db 2>/dev/null
python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
head -c 300 gpurun_out/$TAG/bench.json; echo; head -8 gpurun_out/$TAG/kernel_stats.csv

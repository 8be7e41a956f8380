set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/v13
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/v13/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-concurrent > $R/gpurun_out/v13/stats.log 2>&1
find $R/gpurun_out/v13/stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/v13/v13_kernel_stats.csv \;
cd $R && bash tools/pmc_passes.sh v13/pmc --no-concurrent > gpurun_out/v13/pmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/v13/pmc gpurun_out/v13/pmc_summary.json > gpurun_out/v13/pmc_summary.txt 2>&1
python3 bench.py > gpurun_out/v13/v13_bench.json 2> gpurun_out/v13/bench.err
head -c 400 gpurun_out/v13/v13_bench.json; echo; head -8 gpurun_out/v13/v13_kernel_stats.csv

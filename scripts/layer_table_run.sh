set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 1 > $O/stats_bench.log 2>&1
cd $R
python3 scripts/trace_stats.py $O/stats/run_kernel_trace.csv > $O/trace_summary.txt 2>&1
python3 scripts/layer_table.py $O/stats/run_kernel_trace.csv > $O/layer_table.txt 2>&1 || true
rm -f $O/stats/run_kernel_trace.csv
head -50 $O/trace_summary.txt

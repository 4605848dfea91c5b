# kernel trace of another workload -> gpurun_out/$1/trace_summary.txt   usage: r4_prof_wl.sh outdir workload
O=gpurun_out/$1; W=$2
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o run -- python3 $R/bench.py --workload $W --no-cpu-baseline --no-roofline --steps 4 --warmup 1 > $R/$O/stats_bench.log 2>&1
cd $R
python3 scripts/trace_stats.py $O/stats/run_kernel_trace.csv > $O/trace_summary.txt 2>&1
rm -f $O/stats/run_kernel_trace.csv
head -34 $O/trace_summary.txt

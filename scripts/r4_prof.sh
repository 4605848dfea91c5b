# kernel trace of the production step -> gpurun_out/$1/{trace_summary,layer_table}.txt   usage: r4_prof.sh outdir [ENV=val ...]
O=gpurun_out/$1; shift
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
env "$@" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 1 > $R/$O/stats_bench.log 2>&1
cd $R
python3 scripts/trace_stats.py $O/stats/run_kernel_trace.csv > $O/trace_summary.txt 2>&1
python3 scripts/layer_table.py $O/stats/run_kernel_trace.csv > $O/layer_table.txt 2>&1 || true
rm -f $O/stats/run_kernel_trace.csv
head -45 $O/trace_summary.txt

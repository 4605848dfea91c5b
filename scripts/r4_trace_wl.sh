# GPU box: kernel trace summary of another workload   usage: r4_trace_wl.sh <workload> <outdir>
O=$GRAFT_REPO_ROOT/gpurun_out/$2
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o run -- python3 $R/bench.py --workload $1 --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $O/tr.log 2>&1
cd $R
python3 scripts/trace_stats.py $O/tr/run_kernel_trace.csv 0.5 > $O/summary.txt 2>&1
rm -rf $O/tr
head -60 $O/summary.txt

# round 4 experiments: 3x3 weight gradients on the tapped ring kernel, fewer ring workgroups, production layer table
mkdir -p gpurun_out/r4e2
O=gpurun_out/r4e2
MMSKIN_MIX_OP=wgrad MMSKIN_WGRAD_3X3=0 timeout -k 10 300 python scripts/conv_mix.py no3x3 > $O/mix_no3x3.txt 2>&1 || exit 1; tail -1 $O/mix_no3x3.txt
run() { env "$@" timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'])"; }
for rep in 1 2; do
run A=0 || exit 1
run MMSKIN_WGRAD_3X3=0 || exit 1
run MMSKIN_WGRAD_RING_BLOCKS=128 || exit 1
run MMSKIN_WGRAD_RING_BLOCKS=192 || exit 1
run MMSKIN_WGRAD_RING_DEEP=0 || exit 1
run MMSKIN_WGRAD_RING_BLOCKS=128 MMSKIN_WGRAD_3X3=0 || exit 1
done > $O/ab.txt 2>&1
cat $O/ab.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o run -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 1 > $R/$O/stats_bench.log 2>&1
cd $R
python3 scripts/trace_stats.py $O/stats/run_kernel_trace.csv > $O/trace_summary.txt 2>&1
python3 scripts/layer_table.py $O/stats/run_kernel_trace.csv > $O/layer_table.txt 2>&1 || true
rm -f $O/stats/run_kernel_trace.csv
tail -60 $O/layer_table.txt

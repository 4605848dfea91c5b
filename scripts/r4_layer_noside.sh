# GPU box: per-layer table of the step with every kernel serialised on one stream (MMSKIN_NO_SIDE_STREAM=1): each launch at its isolated time, real epilogues
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export MMSKIN_NO_SIDE_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o run -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $O/tr.log 2>&1
cd $R
python3 scripts/layer_table.py $O/tr/run_kernel_trace.csv > $O/layer_table_noside.txt 2>&1
rm -rf $O/tr
tail -n 3 $O/layer_table_noside.txt

# GPU box: kernel trace of the production step -> per-segment timeline (gpurun_out/$1/timeline.txt)
O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o run -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 > $O/tr.log 2>&1
cd $R
python3 scripts/step_timeline.py $O/tr/run_kernel_trace.csv --full > $O/timeline.txt 2>&1
rm -rf $O/tr
head -60 $O/timeline.txt

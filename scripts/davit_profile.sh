# GPU box: kernel trace of BASELINE config 4 (DaViT-tiny + tab-transformer + gfcam, batch 64, everything trainable)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3_davit}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --workload davit-tiny-gfcam --no-cpu-baseline --no-roofline --steps 4 --warmup 1 > $O/bench.log 2>&1
cd $R
python3 scripts/trace_stats.py $O/prof/run_kernel_trace.csv > $O/trace_summary.txt 2>&1
rm -f $O/prof/run_kernel_trace.csv
head -4 $O/trace_summary.txt

# GPU box (gpurun -- bash scripts/secondary_bench.sh): BASELINE configs 3 - 5 on the sources in the tree (builder-run, not driver-timed),
# plus the kernel stats of config 5 (BEiT-v2 large + BERT) for profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_secondary
mkdir -p $O
cd $R
for w in densenet169-metablock davit-tiny-gfcam beitv2-large-bert-rgatt; do
  timeout -k 10 400 python3 bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err
  tail -c 700 $O/bench_$w.json
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/beit -o run -- python3 $R/bench.py --workload beitv2-large-bert-rgatt --no-cpu-baseline --no-roofline --steps 4 --warmup 1 > $O/beit_bench.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dn -o run -- python3 $R/bench.py --workload densenet169-metablock --no-cpu-baseline --no-roofline --steps 4 --warmup 1 > $O/dn_bench.log 2>&1
cd $R
python3 scripts/trace_stats.py $O/beit/run_kernel_trace.csv > $O/beit_trace_summary.txt 2>&1
python3 scripts/trace_stats.py $O/dn/run_kernel_trace.csv > $O/dn_trace_summary.txt 2>&1
rm -f $O/beit/run_kernel_trace.csv
head -5 $O/beit_trace_summary.txt

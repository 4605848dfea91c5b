# GPU box (gpurun -- bash scripts/profile_round4.sh): the judged evidence for the sources in the tree -> gpurun_out/r4_final/
#   1 kernel trace + stats of the production step           -> trace summary, layer table, conv_gemm / wgrad production averages
#   2/3 HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes): dominant kernel per launch AND the whole step by kernel class
#   4 matrix-core busy cycles of the dominant kernel (SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + GRBM_GUI_ACTIVE)
# Counters only with --kernel-trace (never with the hip / hsa / memory-copy trace domains); python3 directly after `--`.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-roofline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $B --steps 10 --warmup 1 > $O/stats_bench.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o run -- $B --steps 1 --warmup 1 > $O/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o run -- $B --steps 1 --warmup 1 > $O/pmc_w.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m -o run -- $B --steps 1 --warmup 1 > $O/pmc_m.log 2>&1
cd $R
python3 scripts/pmc_traffic.py $O/pmc_f $O/pmc_w $O/conv_gemm_traffic.json $O/pmc_m $O/stats/run_kernel_trace.csv > $O/traffic.log 2>&1
python3 scripts/step_traffic.py $O/pmc_f $O/pmc_w $O/step_traffic.txt "" $O/step_traffic.json > /dev/null 2>&1
python3 scripts/step_traffic.py $O/pmc_f $O/pmc_w $O/step_traffic_by_kernel.txt resnet50-crossattention > /dev/null 2>&1
python3 scripts/trace_stats.py $O/stats/run_kernel_trace.csv > $O/trace_summary.txt 2>&1
python3 scripts/layer_table.py $O/stats/run_kernel_trace.csv > $O/layer_table.txt 2>&1 || true
cp $O/stats/run_kernel_stats.csv $O/kernel_stats.csv || true
rm -rf $O/pmc_f $O/pmc_w $O/pmc_m $O/stats
tail -12 $O/traffic.log
head -14 $O/step_traffic.txt
head -6 $O/trace_summary.txt

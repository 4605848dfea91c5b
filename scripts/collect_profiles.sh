# after `gpurun -- bash scripts/profile_round.sh`: copy the judged summaries from gpurun_out/r2_final into profiles/ (run in the repo root)
set -e
H=$(python -c 'import bench; print(bench.source_hash())')
grep -q "$H" gpurun_out/r2_final/conv_gemm_traffic.json || { echo "traffic JSON is not for sources $H"; exit 1; }
cp gpurun_out/r2_final/conv_gemm_traffic.json profiles/conv_gemm_traffic.json
{ echo "r02: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-roofline   (1x MI355X, ResNet-50 + crossattention, batch 256, bf16;"
  echo "last 60 % of the dispatches; scripts/trace_stats.py; sources $H; the profiler's per-launch host cost shows as idle time: un-profiled the step takes the union-busy time)"
  echo; cat gpurun_out/r2_final/trace_summary.txt; } > profiles/r02_kernel_stats_resnet50-crossattention.txt
cp gpurun_out/r2_final/stats/run_kernel_stats.csv profiles/r02_kernel_stats_resnet50-crossattention.csv
echo "profiles/ updated for sources $H"

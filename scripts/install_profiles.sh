# Copy the judged evidence of scripts/profile_round4.sh (gpurun_out/r4_final/) into profiles/ under this round's names.
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/r4_final
cp $S/conv_gemm_traffic.json profiles/conv_gemm_traffic.json
cp $S/step_traffic.json profiles/step_traffic.json
cp $S/step_traffic.txt profiles/r04_step_traffic.txt
cp $S/step_traffic_by_kernel.txt profiles/r04_step_traffic_resnet50-crossattention_by_kernel.txt
cp $S/trace_summary.txt profiles/r04_kernel_stats_resnet50-crossattention.txt
cp $S/kernel_stats.csv profiles/r04_kernel_stats_resnet50-crossattention.csv
cp $S/layer_table.txt profiles/r04_layer_table.txt
cp $S/traffic.log profiles/r04_conv_gemm_traffic.txt
cp $S/bench_line.json profiles/r04_bench_line.json
[ -f gpurun_out/parity_report.jsonl ] && cp gpurun_out/parity_report.jsonl profiles/r04_parity_report.jsonl
ls -la profiles/ | grep -E "r04_|traffic.json"

set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2_final
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_final/stats -o run -- python3 $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/r2_final/stats_bench.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_final/pmc_f -o run -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/r2_final/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_final/pmc_w -o run -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/r2_final/pmc_w.log 2>&1
cd $R
python3 scripts/pmc_traffic.py gpurun_out/r2_final/pmc_f gpurun_out/r2_final/pmc_w gpurun_out/r2_final/conv_gemm_traffic.json > gpurun_out/r2_final/traffic.log 2>&1
python3 scripts/trace_stats.py gpurun_out/r2_final/stats/run_kernel_trace.csv > gpurun_out/r2_final/trace_summary.txt 2>&1
# keep the merge small: counters of the conv kernels only are needed locally
rm -rf gpurun_out/r2_final/pmc_f gpurun_out/r2_final/pmc_w
rm -f gpurun_out/r2_final/stats/run_kernel_trace.csv
tail -3 gpurun_out/r2_final/traffic.log
head -5 gpurun_out/r2_final/trace_summary.txt

# GPU box: HBM bytes of one whole training step by kernel class (scripts/step_traffic.py) -> gpurun_out/r3_traffic/step_traffic[_<workload>].txt
# usage: step_traffic.sh [workload]
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_traffic
W=${1:-}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-roofline"
if [ -n "$W" ]; then B="$B --workload $W"; fi
rm -rf $O/pmc_f $O/pmc_w
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o run -- $B --steps 1 --warmup 1 > $O/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o run -- $B --steps 1 --warmup 1 > $O/pmc_w.log 2>&1
cd $R
if [ -n "$W" ]; then python3 scripts/step_traffic.py $O/pmc_f $O/pmc_w $O/step_traffic_$W.txt $W; else python3 scripts/step_traffic.py $O/pmc_f $O/pmc_w $O/step_traffic.txt; fi
rm -rf $O/pmc_f $O/pmc_w

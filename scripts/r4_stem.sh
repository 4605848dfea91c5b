O=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export MMSKIN_STEM7X7=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st$v -o run -- python3 $R/scripts/stem_bench.py > $O/st$v.log 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/st$v/run_kernel_stats.csv")))
for r in rows[:8]: print("$v", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
  rm -rf $O/st$v
done

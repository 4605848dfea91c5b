# usage: ab.sh VAR v1 v2 ...   -- bench.py ms/step for each value of an env knob, twice, on the same box
var=$1; shift
for rep in 1 2; do for v in "$@"; do
  env $var=$v timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$var=$v', d['ms_per_step'])" || exit 1
done; done

# usage: ab_env.sh VAR v0 v1 : bench.py ms/step with VAR=v0 / VAR=v1, twice, same box
for rep in 1 2; do for v in $2 $3; do
  env $1=$v timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1=$v', d['ms_per_step'])" || exit 1
done; done

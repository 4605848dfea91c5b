# usage: ab_env_long.sh VAR v0 v1 : bench.py ms/step with VAR=v0 / VAR=v1, three alternations of 60 steps, same box
for rep in 1 2 3; do for v in $2 $3; do
  env $1=$v timeout -k 10 180 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1=$v', d['ms_per_step'])" || exit 1
done; done

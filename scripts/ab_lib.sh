# usage: ab_lib.sh [ENV=val ...]  -- bench.py ms/step with build_ab/libA.so vs libB.so swapped in, twice, on the same box
L=multimodal-model-skin-lesion-classifier_amd/libmmskin_hip.so
cp $L build_ab/orig.so
trap 'cp build_ab/orig.so $L' EXIT   # a failed run must not leave libA / libB installed as the production library
for rep in 1 2; do for v in ${AB_LIBS:-A B}; do
  cp build_ab/lib$v.so $L
  env "$@" timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lib$v $*', d['ms_per_step'])" || exit 1
done; done

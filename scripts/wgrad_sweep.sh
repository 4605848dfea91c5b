run() { env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'])"; }
run MMSKIN_WGRAD_FLOOR=0 || exit 1
run MMSKIN_WGRAD_FLOOR=1 || exit 1
run MMSKIN_WGRAD_FLOOR=1 MMSKIN_WGRAD_B1=512 || exit 1
run MMSKIN_WGRAD_FLOOR=1 MMSKIN_WGRAD_B1=512 MMSKIN_WGRAD_B3BIG=512 || exit 1
run MMSKIN_WGRAD_FLOOR=1 MMSKIN_WGRAD_B1=512 MMSKIN_WGRAD_B3=512 || exit 1
run MMSKIN_WGRAD_FLOOR=1 MMSKIN_WGRAD_B1=768 || exit 1
run MMSKIN_WGRAD_FLOOR=1 MMSKIN_WGRAD_B1=768 MMSKIN_WGRAD_B3=768 || exit 1

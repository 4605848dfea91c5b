run() { env "$@" timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$*', d['ms_per_step'])"; }
for rep in 1 2; do
run A=0 || exit 1
run MMSKIN_CONV_PIPE_MINTILES=128 || exit 1
run MMSKIN_CONV_PIPE_MINTILES=96 || exit 1
run MMSKIN_CONV_PIPE_MINTILES=128 MMSKIN_CONV_PIPE_MINK=512 || exit 1
run MMSKIN_WGRAD_RING_BLOCKS=128 MMSKIN_WGRAD3_RING_BLOCKS=128 || exit 1
run MMSKIN_NO_SIDE_STREAM=1 || exit 1
done

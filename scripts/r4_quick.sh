# quick regression + step time: BatchNorm kernel tests, ResNet e2e, bench   usage: r4_quick.sh outdir
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "batchnorm or conv_forward_backward" 2>&1 | tail -5 > $O/tests_k.log; cat $O/tests_k.log
grep -q passed $O/tests_k.log && ! grep -q failed $O/tests_k.log || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -k "resnet_end_to_end" 2>&1 | tail -5 > $O/tests_m.log; cat $O/tests_m.log
grep -q passed $O/tests_m.log && ! grep -q failed $O/tests_m.log || exit 1
for i in 1 2; do timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'])"; done | tee $O/bench.txt

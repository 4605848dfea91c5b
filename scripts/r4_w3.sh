mkdir -p gpurun_out/r4w3
O=gpurun_out/r4w3
python -m pytest tests/test_gpu_wgrad_ring.py tests/test_gpu_kernels.py -x -q -k "ring or wgrad3x3 or conv_forward_backward" 2>&1 | tail -15 > $O/tests.log; cat $O/tests.log
grep -q passed $O/tests.log && ! grep -q failed $O/tests.log || exit 1
for v in 0 1; do MMSKIN_MIX_OP=wgrad MMSKIN_WGRAD3_RING=$v timeout -k 10 300 python scripts/conv_mix.py w3ring$v > $O/mix_w3ring$v.txt 2>&1 || exit 1; grep "3x3\|TOTAL" $O/mix_w3ring$v.txt; done
bash scripts/ab.sh MMSKIN_WGRAD3_RING 0 1 > $O/ab_step.txt 2>&1 || { cat $O/ab_step.txt; exit 1; }
cat $O/ab_step.txt

# round 4: algebraic BatchNorm backward -- parity, e2e tests, step A/B
mkdir -p gpurun_out/r4abn
O=gpurun_out/r4abn
python -m pytest tests/test_gpu_abn.py tests/test_gpu_wgrad_ring.py -x -q 2>&1 | tail -15 > $O/tests_abn.log; cat $O/tests_abn.log
grep -q passed $O/tests_abn.log && ! grep -q failed $O/tests_abn.log || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -k "resnet_end_to_end or ragged or freeze or baseline_shape_properties" 2>&1 | tail -15 > $O/tests_model.log; cat $O/tests_model.log
grep -q passed $O/tests_model.log && ! grep -q failed $O/tests_model.log || exit 1
bash scripts/ab.sh MMSKIN_ABN 0 1 > $O/ab_step.txt 2>&1 || { cat $O/ab_step.txt; exit 1; }
cat $O/ab_step.txt

mkdir -p gpurun_out/r4f2p
O=gpurun_out/r4f2p
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -k "resnet_end_to_end or baseline_shape or gamma or production_size or ragged or freeze or checkpoint" 2>&1 | tail -15 > $O/tests_model.log; cat $O/tests_model.log
grep -q passed $O/tests_model.log && ! grep -q failed $O/tests_model.log || exit 1
bash scripts/ab.sh MMSKIN_FWD2P 0 1 > $O/ab_step.txt 2>&1 || { cat $O/ab_step.txt; exit 1; }
cat $O/ab_step.txt

# round 4: ring weight-gradient kernel -- parity, isolated per-layer timing (A/B with the register-staged kernel), step A/B
set -o pipefail
mkdir -p gpurun_out/r4wg
python -m pytest tests/test_gpu_wgrad_ring.py -x -q 2>&1 | tail -15 > gpurun_out/r4wg/tests.log || { cat gpurun_out/r4wg/tests.log; exit 1; }
cat gpurun_out/r4wg/tests.log
for v in 0 1; do MMSKIN_MIX_OP=wgrad MMSKIN_WGRAD_RING=$v timeout -k 10 300 python scripts/conv_mix.py ring$v > gpurun_out/r4wg/mix_ring$v.txt 2>&1 || exit 1; tail -1 gpurun_out/r4wg/mix_ring$v.txt; done
MMSKIN_MIX_OP=wgrad MMSKIN_WGRAD_RING_DEEP=0 timeout -k 10 300 python scripts/conv_mix.py shallow > gpurun_out/r4wg/mix_shallow.txt 2>&1 || exit 1; tail -1 gpurun_out/r4wg/mix_shallow.txt
MMSKIN_MIX_OP=wgrad MMSKIN_WGRAD_RING_BLOCKS=512 timeout -k 10 300 python scripts/conv_mix.py b512 > gpurun_out/r4wg/mix_b512.txt 2>&1 || exit 1; tail -1 gpurun_out/r4wg/mix_b512.txt
bash scripts/ab.sh MMSKIN_WGRAD_RING 0 1 > gpurun_out/r4wg/ab_step.txt 2>&1 || exit 1
cat gpurun_out/r4wg/ab_step.txt

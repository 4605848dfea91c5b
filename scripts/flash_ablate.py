"""GPU box: the fused attention kernel with parts of its work removed (`make -C multimodal-model-skin-lesion-classifier_amd/csrc ablate`
-> build_ab/libmmskin_hip_ablate.so; the production library has no such switch).  MMSKIN_FLASH_ABLATE bits: 1 no K / V global loads
after the first tile, 2 no softmax, 4 no MFMA, 8 no K / V LDS stores.  One process per setting (the value is read per call).
usage: flash_ablate.py [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch
from mmskin import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_ab", "libmmskin_hip_ablate.so")
from mmskin import ops
ops.set_linear_dtype("bf16")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, H, L, D = 128, 12, 512, 64
qkv = torch.randn(B, L, 3, H, D, device="cuda").bfloat16()
mask = torch.zeros(B, L, device="cuda")
for abl in (0, 1, 2, 4, 8, 9, 6, 15):
    os.environ["MMSKIN_FLASH_ABLATE"] = str(abl)
    with torch.no_grad():
        for _ in range(3):
            ops.attention_blhd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0.0, False, mask_add=mask)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ops.attention_blhd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 0.0, False, mask_add=mask)
        torch.cuda.synchronize()
    print(f"ablate {abl:2d}: {(time.perf_counter() - t0) / iters * 1e6:8.1f} us", flush=True)

"""GPU box: time single conv layers under ablations (MMSKIN_CONV_ABLATE).  Needs the timing-experiment library
(`make -C multimodal-model-skin-lesion-classifier_amd/csrc ablate` -> build_ab/libmmskin_hip_ablate.so): the production
library has the work-skipping switches compiled out."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch
from mmskin import _lib
from mmskin._lib import ptr, stream
_lib.LIB_PATH = os.path.join(ROOT, "build_ab", "libmmskin_hip_ablate.so")
lib = _lib.load()
LAYERS = {  # name: (N, Cin, H, W, Cout, k, stride, pad)
    "l1.c2 3x3 64->64 @56": (256, 64, 56, 56, 64, 3, 1, 1),
    "l1.c3 1x1 64->256 @56": (256, 64, 56, 56, 256, 1, 1, 0),
    "l1.c1 1x1 256->64 @56": (256, 256, 56, 56, 64, 1, 1, 0),
    "l2.c3 1x1 128->512 @28": (256, 128, 28, 28, 512, 1, 1, 0),
    "l2.c2 3x3 128->128 @28": (256, 128, 28, 28, 128, 3, 1, 1),
    "l3.c1 1x1 1024->256 @14": (256, 1024, 14, 14, 256, 1, 1, 0),
    "l3.c2 3x3 256->256 @14": (256, 256, 14, 14, 256, 3, 1, 1),
    "l3.c3 1x1 256->1024 @14": (256, 256, 14, 14, 1024, 1, 1, 0),
    "l4.c2 3x3 512->512 @7": (256, 512, 7, 7, 512, 3, 1, 1),
}
ws = torch.zeros(2 << 30, dtype=torch.uint8, device="cuda")
torch.manual_seed(0)
ws[: 1 << 30].copy_(torch.randint(0, 255, (1 << 30,), dtype=torch.uint8, device="cuda") & 0x3F)  # small finite bf16 values
configs = [("simple", 0), ("simple", 7), ("simple", 15), ("simple", 31), ("simple", 16), ("simple", 24)]
print(f"{'layer':26s}" + "".join(f"{v+':'+str(a):>10s}" for v, a in configs) + "   (us; TF/s for unablated)")
for name, (N, Cin, H, W, Cout, k, s, p) in LAYERS.items():
    OH = (H + 2 * p - k) // s + 1
    flops = 2.0 * N * OH * OH * Cout * Cin * k * k
    row = f"{name:26s}"
    for var, abl in configs:
        os.environ["MMSKIN_CONV_ABLATE"] = str(abl)
        us = lib.mmskin_conv2d_time(N, Cin, H, W, Cout, k, k, s, p, _lib.BF16, 20, ptr(ws), stream())
        row += f"{us:10.1f}"
        if abl == 0:
            row += f"({flops / us / 1e6:4.0f})"
    print(row)

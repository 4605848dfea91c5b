"""GPU box: the layer-1 3x3 kernel (csrc/conv3x3_c64.hip) under ablations on the `make ablate` library (MMSKIN_C3_ABLATE bits: 1 no row
prefetch, 4 no MFMA, 8 no output stores, 32 no fragment reads).  usage: c3_ablate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch
from mmskin import _lib
from mmskin._lib import ptr, stream
_lib.LIB_PATH = os.path.join(ROOT, "build_ab", "libmmskin_hip_ablate.so")
lib = _lib.load()
ws = torch.zeros(3 << 30, dtype=torch.uint8, device="cuda")
torch.manual_seed(0)
hi = torch.randint(0x3c, 0x40, (1 << 29,), dtype=torch.int16, device="cuda") << 8
lo = torch.randint(0, 256, (1 << 29,), dtype=torch.int16, device="cuda")
ws[: 1 << 30].view(torch.int16).copy_(hi | lo)
for fn, nm in ((lib.mmskin_conv2d_time, "fwd"), (lib.mmskin_conv2d_dgrad_time, "dgrad")):
    row = f"{nm:6s}"
    for abl in (0, 4, 32, 36, 8, 1, 45):
        os.environ["MMSKIN_C3_ABLATE"] = str(abl)
        us = min(fn(256, 64, 56, 56, 64, 3, 3, 1, 1, _lib.BF16, 20, ptr(ws), stream()) for _ in range(3))
        row += f"  abl {abl}: {us:6.1f}"
    print(row, flush=True)

"""GPU box: time ONE conv shape's dgrad (or fwd / wgrad via MMSKIN_MIX_OP) -- usage: dgrad_one.py Cin H Cout k stride [label]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch
from mmskin import _lib
from mmskin._lib import ptr, stream
if os.environ.get("MMSKIN_CONV_ABLATE") or os.environ.get("MMSKIN_WGRAD_ABLATE"):   # `make ablate` library only
    _lib.LIB_PATH = os.path.join(ROOT, "build_ab", "libmmskin_hip_ablate.so")
lib = _lib.load()
Cin, H, Cout, k, s = (int(v) for v in sys.argv[1:6])
N = 256
ws = torch.zeros(3 << 30, dtype=torch.uint8, device="cuda")
ws[: 1 << 30].view(torch.int16).copy_(torch.randint(0x3c00, 0x4000, (1 << 29,), dtype=torch.int16, device="cuda"))
fn = {"wgrad": lib.mmskin_conv2d_wgrad_time, "fwd": lib.mmskin_conv2d_time}.get(os.environ.get("MMSKIN_MIX_OP"), lib.mmskin_conv2d_dgrad_time)
us = fn(N, Cin, H, H, Cout, k, k, s, k // 2, _lib.BF16, 20, ptr(ws), stream())
print(f"{sys.argv[6] if len(sys.argv) > 6 else ''} Cin={Cin} H={H} Cout={Cout} k={k} s={s}: {us:.1f} us")

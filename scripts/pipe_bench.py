"""GPU box: time the conv / dgrad launch of a few GEMM-heavy shapes on the production library (or the one given as argv[2]).
Usage: [MMSKIN_CONV_PIPE_FORCE=1] [MMSKIN_CONV_PIPE_TILE=..] python scripts/pipe_bench.py [fwd|dgrad] [lib.so]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch
from mmskin import _lib
from mmskin._lib import ptr, stream
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
lib = _lib.load()
LAYERS = {  # name: (N, Cin, H, W, Cout, k, stride, pad)
    "gemm 4096^3": (16, 4096, 16, 16, 4096, 1, 1, 0),
    "gemm 8192x4096x4096": (32, 4096, 16, 16, 4096, 1, 1, 0),
    "gemm 8192^3": (32, 8192, 16, 16, 8192, 1, 1, 0),
    "l3.c2 3x3 256 @14": (256, 256, 14, 14, 256, 3, 1, 1),
    "l3.c1b 1x1 1024->256": (256, 1024, 14, 14, 256, 1, 1, 0),
    "l3.c3 1x1 256->1024": (256, 256, 14, 14, 1024, 1, 1, 0),
    "l4.c2 3x3 512 @7": (256, 512, 7, 7, 512, 3, 1, 1),
    "l4.c1b 1x1 2048->512": (256, 2048, 7, 7, 512, 1, 1, 0),
    "l4.c3 1x1 512->2048": (256, 512, 7, 7, 2048, 1, 1, 0),
}
op = sys.argv[1] if len(sys.argv) > 1 else "fwd"
fn = lib.mmskin_conv2d_dgrad_time if op == "dgrad" else lib.mmskin_conv2d_time
ws = torch.zeros(3 << 30, dtype=torch.uint8, device="cuda")
torch.manual_seed(0)
hi = torch.randint(0x3c, 0x40, (1 << 29,), dtype=torch.int16, device="cuda") << 8
lo = torch.randint(0, 256, (1 << 29,), dtype=torch.int16, device="cuda")
sign = torch.randint(0, 2, (1 << 29,), dtype=torch.int16, device="cuda") << 15
ws[: 1 << 30].view(torch.int16).copy_(hi | lo | sign)
del hi, lo, sign
has = hasattr(lib, "mmskin_conv_pipe_launches")
print(f"[{op}] lib {os.path.basename(_lib.LIB_PATH)} force {os.environ.get('MMSKIN_CONV_PIPE_FORCE', '0')} tile {os.environ.get('MMSKIN_CONV_PIPE_TILE', 'model')}")
for name, (N, Cin, H, W, Cout, k, s, p) in LAYERS.items():
    OH = (H + 2 * p - k) // s + 1
    flops = 2.0 * N * OH * OH * Cout * Cin * k * k
    n0 = lib.mmskin_conv_pipe_launches() if has else 0
    us = min(fn(N, Cin, H, W, Cout, k, k, s, p, _lib.BF16, 20, ptr(ws), stream()) for _ in range(3))
    n1 = lib.mmskin_conv_pipe_launches() if has else 0
    print(f"{name:24s} {us:8.1f} us {flops / us / 1e6:6.0f} TF/s  {'pipe' if n1 > n0 else '-'}", flush=True)

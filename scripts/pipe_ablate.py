"""GPU box: the 8-phase pipelined conv kernel under ablations (MMSKIN_CONV_ABLATE bits: 1 no gather DMA, 2 no weight DMA, 4 no MFMA,
32 no fragment reads, 64 no vmcnt waits) on the `make ablate` library, every eligible launch forced onto it.
Usage: [MMSKIN_CONV_PIPE_TILE=256|224|196] python scripts/pipe_ablate.py [fwd|dgrad]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
os.environ.setdefault("MMSKIN_CONV_PIPE_FORCE", "1")
import torch
from mmskin import _lib
from mmskin._lib import ptr, stream
_lib.LIB_PATH = os.path.join(ROOT, "build_ab", "libmmskin_hip_ablate.so")
lib = _lib.load()
LAYERS = {  # name: (N, Cin, H, W, Cout, k, stride, pad)
    "gemm 4096^3": (16, 4096, 16, 16, 4096, 1, 1, 0),
    "gemm 8192x4096x4096": (32, 4096, 16, 16, 4096, 1, 1, 0),
    "l3.c2 3x3 256 @14": (256, 256, 14, 14, 256, 3, 1, 1),
    "l3.c1b 1x1 1024->256": (256, 1024, 14, 14, 256, 1, 1, 0),
    "l3.c3 1x1 256->1024": (256, 256, 14, 14, 1024, 1, 1, 0),
    "l4.c2 3x3 512 @7": (256, 512, 7, 7, 512, 3, 1, 1),
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
configs = [0, 4, 3, 32, 7, 36, 35, 39, 64 + 39]
print(f"[{op}] tile pin {os.environ.get('MMSKIN_CONV_PIPE_TILE', 'model')}")
print(f"{'layer':24s}" + "".join(f"{'abl ' + str(a):>10s}" for a in configs) + "   (us; TF/s for unablated)")
for name, (N, Cin, H, W, Cout, k, s, p) in LAYERS.items():
    OH = (H + 2 * p - k) // s + 1
    flops = 2.0 * N * OH * OH * Cout * Cin * k * k
    row = f"{name:24s}"
    for abl in configs:
        os.environ["MMSKIN_CONV_ABLATE"] = str(abl)
        n0 = lib.mmskin_conv_pipe_launches()
        us = fn(N, Cin, H, W, Cout, k, k, s, p, _lib.BF16, 20, ptr(ws), stream())
        assert lib.mmskin_conv_pipe_launches() > n0, "not the pipelined kernel"
        row += f"{us:10.1f}"
        if abl == 0:
            row += f"({flops / us / 1e6:4.0f})"
    print(row, flush=True)

"""GPU box: phase stamps (s_memtime) of the pipelined conv kernel's workgroups on the `make ablate` library -> where a workgroup's life
goes and the shader clock the chip holds during the launch (ticks per microsecond of the launch span).
Phases: 0 start | 1 tables + barrier | 2 first six units issued | 3 first units landed + barrier | 4 K loop done | 5 accumulators staged |
6 rows streamed out | 7 end.   usage: [MMSKIN_CONV_PIPE_TILE=..] pipe_stamps.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
os.environ.setdefault("MMSKIN_CONV_PIPE_FORCE", "1")
import torch
from mmskin import _lib
from mmskin._lib import ptr, stream
_lib.LIB_PATH = os.path.join(ROOT, "build_ab", "libmmskin_hip_ablate.so")
lib = _lib.load()
lib.mmskin_debug_set_conv_stamps.argtypes = [ctypes.c_void_p]
lib.mmskin_debug_set_conv_stamps.restype = None
LAYERS = {
    "gemm 4096^3": (16, 4096, 16, 16, 4096, 1, 1, 0),
    "gemm 8192^3": (32, 8192, 16, 16, 8192, 1, 1, 0),
    "l3.c2 3x3 256 @14": (256, 256, 14, 14, 256, 3, 1, 1),
    "l3.c1b 1x1 1024->256": (256, 1024, 14, 14, 256, 1, 1, 0),
    "l3.c3 1x1 256->1024": (256, 256, 14, 14, 1024, 1, 1, 0),
}
ws = torch.zeros(3 << 30, dtype=torch.uint8, device="cuda")
torch.manual_seed(0)
hi = torch.randint(0x3c, 0x40, (1 << 29,), dtype=torch.int16, device="cuda") << 8
lo = torch.randint(0, 256, (1 << 29,), dtype=torch.int16, device="cuda")
sign = torch.randint(0, 2, (1 << 29,), dtype=torch.int16, device="cuda") << 15
ws[: 1 << 30].view(torch.int16).copy_(hi | lo | sign)
del hi, lo, sign
names = ["prologue", "first issue", "first landed", "K loop", "acc -> lds", "rows out", "stats/end"]
MAXWG = 4096
for name, (N, Cin, H, W, Cout, k, s, p) in LAYERS.items():
    stamps = torch.zeros(MAXWG * 8, dtype=torch.int64, device="cuda")
    lib.mmskin_debug_set_conv_stamps(None)
    for _ in range(3):
        us = lib.mmskin_conv2d_time(N, Cin, H, W, Cout, k, k, s, p, _lib.BF16, 20, ptr(ws), stream())   # warm the clocks
    lib.mmskin_debug_set_conv_stamps(ptr(stamps))
    lib.mmskin_conv2d_time(N, Cin, H, W, Cout, k, k, s, p, _lib.BF16, 1, ptr(ws), stream())
    lib.mmskin_debug_set_conv_stamps(None)
    torch.cuda.synchronize()
    t = stamps.reshape(MAXWG, 8).cpu().double()
    t = t[t[:, 7] > 0]
    d = t[:, 1:] - t[:, :-1]
    life = t[:, 7] - t[:, 0]
    span = float(t[:, 7].max() - t[:, 0].min())
    print(f"{name}: {us:.1f} us per launch, {t.shape[0]} workgroups; launch span {span:.0f} ticks -> {span / us:.0f} ticks per us")
    print(f"  workgroup life: median {life.median():.0f}  p90 {life.quantile(0.9):.0f} ticks;  sum of lives / span = {float(life.sum()) / span:.1f} workgroups in flight")
    for i, nm in enumerate(names):
        print(f"  {nm:12s} median {d[:, i].median():8.0f}  p90 {d[:, i].quantile(0.9):8.0f}  ({100 * float(d[:, i].sum()) / float(life.sum()):4.1f} % of life)")

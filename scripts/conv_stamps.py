"""GPU box: where a conv_gemm workgroup spends its life -- s_memtime stamps at the phase boundaries of every workgroup of one launch
(`make -C multimodal-model-skin-lesion-classifier_amd/csrc ablate` library; the production library has no stamps).
Phases: 0 start | 1 tables + barrier | 2 operand DMA issued | 3 landed + barrier | 4 MFMA loop done | 5 accumulators staged + barrier |
6 rows streamed out (stores issued) | 7 end.  Prints the median / p90 length of each phase in cycles and the whole-launch picture.
usage: conv_stamps.py [layer-substring]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch
from mmskin import _lib
from mmskin._lib import ptr, stream
_lib.LIB_PATH = os.path.join(ROOT, "build_ab", "libmmskin_hip_ablate.so")
lib = _lib.load()
lib.mmskin_debug_set_conv_stamps.argtypes = [ctypes.c_void_p]
lib.mmskin_debug_set_conv_stamps.restype = None
LAYERS = {  # name: (N, Cin, H, W, Cout, k, stride, pad)
    "l1.c3 1x1 64->256 @56": (256, 64, 56, 56, 256, 1, 1, 0),
    "l1.c1 1x1 256->64 @56": (256, 256, 56, 56, 64, 1, 1, 0),
    "l2.c3 1x1 128->512 @28": (256, 128, 28, 28, 512, 1, 1, 0),
    "l1.c2 3x3 64->64 @56": (256, 64, 56, 56, 64, 3, 1, 1),
    "l3.c2 3x3 256->256 @14": (256, 256, 14, 14, 256, 3, 1, 1),
}
sel = sys.argv[1] if len(sys.argv) > 1 else ""
ws = torch.zeros(2 << 30, dtype=torch.uint8, device="cuda")
ws[: 1 << 30].copy_(torch.randint(0, 255, (1 << 30,), dtype=torch.uint8, device="cuda") & 0x3F)
names = ["prologue", "dma issue", "dma landed", "mfma loop", "acc -> lds", "rows out", "stats/end"]
for name, (N, Cin, H, W, Cout, k, s, p) in LAYERS.items():
    if sel not in name:
        continue
    OH = (H + 2 * p - k) // s + 1
    nwg = ((N * OH * OH + 127) // 128) * (Cout // (128 if Cout % 128 == 0 else 64))
    stamps = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    lib.mmskin_debug_set_conv_stamps(None)
    us = lib.mmskin_conv2d_time(N, Cin, H, W, Cout, k, k, s, p, _lib.BF16, 10, ptr(ws), stream())   # warm + reference time
    lib.mmskin_debug_set_conv_stamps(ptr(stamps))
    lib.mmskin_conv2d_time(N, Cin, H, W, Cout, k, k, s, p, _lib.BF16, 1, ptr(ws), stream())          # 3 warm-up launches + 1: the last one's stamps stay
    lib.mmskin_debug_set_conv_stamps(None)
    torch.cuda.synchronize()
    t = stamps.reshape(nwg, 8).cpu().double()
    d = t[:, 1:] - t[:, :-1]
    life = t[:, 7] - t[:, 0]
    span = float(t[:, 7].max() - t[:, 0].min())
    print(f"{name}: {us:.1f} us per launch, {nwg} workgroups; launch span {span:.0f} ticks -> {span / us:.1f} ticks per us")
    print(f"  workgroup life: median {life.median():.0f}  p90 {life.quantile(0.9):.0f} ticks;  sum of lives / span = {float(life.sum()) / span:.1f} workgroups in flight (of 1024 slots)")
    for i, nm in enumerate(names):
        print(f"  {nm:12s} median {d[:, i].median():7.0f}  p90 {d[:, i].quantile(0.9):7.0f}  ({100 * float(d[:, i].sum()) / float(life.sum()):4.1f} % of life)")

"""GPU box: time the fused attention kernel alone on the BERT-base (B 128, H 12, L 512, Dh 64, key mask, dropout 0.1) and
BEiT-large (B 128, H 16, L 197, Dh 64, relative-position bias) shapes of BASELINE configs[4].  usage: flash_bench.py [iters] [bf16|fp32]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch
from mmskin import ops
ops.set_linear_dtype("bf16")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
print("tensor dtype", dt)
for name, B, H, L, D, p, kind in (("bert", 128, 12, 512, 64, 0.1, "mask"), ("bert-nodrop", 128, 12, 512, 64, 0.0, "mask"),
                                  ("beit", 128, 16, 197, 64, 0.0, "bias"), ("davit-win", 4096, 3, 49, 32, 0.0, "plain")):
    qkv = torch.randn(B, L, 3, H, D, device="cuda").to(dt)
    mask = torch.zeros(B, L, device="cuda") if kind == "mask" else None
    bias = torch.randn(H, L, L, device="cuda") if kind == "bias" else None
    with torch.no_grad():
        for _ in range(3):
            ops.attention_blhd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], p, True, mask_add=mask, bias=bias)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ops.attention_blhd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], p, True, mask_add=mask, bias=bias)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / iters * 1e6
    fl = 4.0 * B * H * L * L * D
    print(f"{name:12s} B={B} H={H} L={L} D={D}: {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s")

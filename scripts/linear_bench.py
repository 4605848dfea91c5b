"""GPU box: time the bf16 Linear GEMM alone (mmskin_linear_forward_ex, bf16 in -> bf16 out, bias + GELU in the epilogue) on the
BEiT-large / BERT-base shapes of BASELINE configs[4] and check it against a float64 product of the same bf16-rounded operands.
MMSKIN_GEMM_BIG_MINBLOCKS selects the 256 x 256 tile (0 = off).  usage: linear_bench.py [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch
from mmskin import ops
ops.set_linear_dtype("bf16")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print("MMSKIN_GEMM_BIG_MINBLOCKS =", os.environ.get("MMSKIN_GEMM_BIG_MINBLOCKS"))
for name, M, K, N, act in (("beit qkv", 128 * 197, 1024, 3072, 0), ("beit fc1", 128 * 197, 1024, 4096, 2), ("beit fc2", 128 * 197, 4096, 1024, 0),
                           ("beit proj", 128 * 197, 1024, 1024, 0), ("bert qkv", 128 * 512, 768, 2304, 0), ("bert fc1", 128 * 512, 768, 3072, 2),
                           ("bert fc2", 128 * 512, 3072, 768, 0), ("ragged", 1000, 512, 768, 1)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    with torch.no_grad():
        y = ops._linear_ex(x, w, b, act, torch.bfloat16)
        rows = torch.randint(0, M, (64,), device="cuda")
        ref = x[rows].double() @ w.bfloat16().double().t() + b.double()
        ref = torch.relu(ref) if act == 1 else (torch.nn.functional.gelu(ref) if act == 2 else ref)
        err = float((y[rows].double() - ref).abs().max() / ref.abs().max())
        # the last rows too (ragged tail of the row blocks)
        ref_t = x[-3:].double() @ w.bfloat16().double().t() + b.double()
        ref_t = torch.relu(ref_t) if act == 1 else (torch.nn.functional.gelu(ref_t) if act == 2 else ref_t)
        err_t = float((y[-3:].double() - ref_t).abs().max() / ref_t.abs().max())
        for _ in range(3):
            ops._linear_ex(x, w, b, act, torch.bfloat16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ops._linear_ex(x, w, b, act, torch.bfloat16)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / iters * 1e6
    print(f"{name:10s} M={M:6d} K={K:5d} N={N:5d}: {us:8.1f} us  {2.0 * M * K * N / us / 1e6:7.1f} TF/s   err {err:.2e} tail {err_t:.2e}")

print("fused tails (mmskin_linear_lane): y = residual + gamma * dropout(x W^T + b), fp32 residual stream")
for name, M, K, N, use_g, p in (("beit proj", 128 * 197, 1024, 1024, True, 0.0), ("beit fc2", 128 * 197, 4096, 1024, True, 0.0),
                                ("bert attn.out", 128 * 512, 768, 768, False, 0.1), ("bert out", 128 * 512, 3072, 768, False, 0.1),
                                ("bert out p=0", 128 * 512, 3072, 768, False, 0.0)):
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda") * K ** -0.5
    b = torch.randn(N, device="cuda")
    gam = torch.rand(N, device="cuda") if use_g else None
    res = torch.randn(M, N, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            ops.linear_lane(x, w, b, 0, gam, res, p, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ops.linear_lane(x, w, b, 0, gam, res, p, True)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / iters * 1e6
    print(f"{name:14s} M={M:6d} K={K:5d} N={N:5d}: {us:8.1f} us  {2.0 * M * K * N / us / 1e6:7.1f} TF/s")

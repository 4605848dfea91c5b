"""-m gpu: algebraic BatchNorm backward of an expanding 1x1 convolution (csrc/abn.hip, the two-source data-gradient GEMM of
csrc/conv_gemm.hip and the Gram tiles of csrc/wgrad_ring.hip) against an fp64 reference of what it replaces:

    x = y W^T;  dz = cA g + cB x + cC;  dy = dz W;  dW = dz^T y          (autograd through nn.BatchNorm2d + nn.Conv2d of
                                                                         torchvision's Bottleneck.conv3, train_pad_20.py:112)

g, y are bf16-representable and W is rounded to bf16 on both sides, so the reference multiplies what the kernels multiply.  What is
left: the folded weights cA (.) W and Q = W^T diag(cB) W are rounded to bf16 (2^-9 relative, per term), dy is stored in bf16, the
sums run in fp32.  dy is therefore checked twice: against a torch EMULATION of exactly those roundings (relative L2 <= 1e-3: only
bf16 rounding ties of the stored result may differ) and against the fp64 reference (relative L2 <= 6e-3, worst element <= 5e-2 of the
rms: the bf16 kernel bound).  dW has no bf16-rounded factor of its own (S, the Gram matrix and the column sums are fp32): worst
element <= 1e-3 of its rms."""
import json
import os

import pytest
import torch

from gpu_util import DEV, rel_err, ws
from mmskin import _lib
from mmskin._lib import call, ptr, stream

pytestmark = pytest.mark.gpu
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")

CASES = [
    # N, Cw, C4, H, W
    (2, 64, 256, 14, 14),     # layer1 shape: 256x64 tiles (two pixel groups), one Gram tile with 192 spare columns
    (3, 128, 512, 9, 11),     # layer2 shape: 256x128 tiles, odd sizes, ragged last row block / stage
    (20, 64, 256, 14, 14),    # several splits
    (6, 128, 512, 14, 14),
]


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_algebraic_bn_backward_matches_fp64_reference(case):
    N, Cw, C4, H, W = case
    lib = _lib.load()
    gen = torch.Generator().manual_seed(sum(case))
    y = torch.relu(torch.randn(N, Cw, H, W, generator=gen)).bfloat16().float()          # post-ReLU activations
    g = (torch.randn(N, C4, H, W, generator=gen) * (torch.rand(N, C4, H, W, generator=gen) > 0.4)).bfloat16().float()   # masked gradient
    w = (torch.randn(C4, Cw, generator=gen) / Cw ** 0.5)
    wb = w.bfloat16().double()
    cA = (torch.rand(C4, generator=gen) + 0.5).double()
    cB = (torch.randn(C4, generator=gen) * 0.05).double()
    cC = (torch.randn(C4, generator=gen) * 0.02).double()
    M = N * H * W
    ym = y.double().permute(0, 2, 3, 1).reshape(M, Cw)
    gm = g.double().permute(0, 2, 3, 1).reshape(M, C4)
    x = ym @ wb.t()
    dz = cA * gm + cB * x + cC
    dy_ref = (dz @ wb).reshape(N, H, W, Cw).permute(0, 3, 1, 2)
    dw_ref = dz.t() @ ym
    # the algebraic form with the kernels' roundings: folded weights in bf16, exact sums, result stored in bf16
    w1 = (cA[:, None] * wb).float().bfloat16().double()                       # [C4][Cw]
    q = (wb.t() @ (cB[:, None] * wb)).float().bfloat16().double()             # [Cw][Cw]
    r = (cC @ wb)
    # (the conv kernels stage their accumulators through LDS in bf16 BEFORE the epilogue adds the bias: two roundings)
    dy_emu = ((gm @ w1 + ym @ q).float().bfloat16().double() + r).float().bfloat16().double().reshape(N, H, W, Cw).permute(0, 3, 1, 2)
    dy = torch.empty(N, Cw, H, W, device=DEV)
    dw = torch.empty(C4, Cw, device=DEV)
    wsp = ws(lib.mmskin_abn_workspace_bytes(N, Cw, C4, H, W))
    n0 = lib.mmskin_wgrad_ring_launches()
    dev = [t.float().to(DEV) for t in (g, y, w, cA, cB, cC)]     # kept alive until the synchronize below
    call("mmskin_abn_backward", *[ptr(t) for t in dev], N, Cw, C4, H, W, ptr(dy), ptr(dw), ptr(wsp), stream())
    torch.cuda.synchronize()
    assert lib.mmskin_wgrad_ring_launches() == n0 + 1
    e_dy, e_dw = rel_err(dy, dy_ref), rel_err(dw, dw_ref)
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(dict(test="abn_backward", case=list(case), dy_max_err_over_rms=e_dy, dw_max_err_over_rms=e_dw)) + "\n")
    def rel_l2(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return float((a - b).norm() / b.norm())
    l2_ref, l2_emu = rel_l2(dy, dy_ref), rel_l2(dy, dy_emu)
    with open(REPORT, "a") as f:
        f.write(json.dumps(dict(test="abn_backward_dy", case=list(case), rel_l2_vs_fp64=l2_ref, rel_l2_vs_emulation=l2_emu,
                                emulation_vs_fp64=rel_l2(dy_emu, dy_ref))) + "\n")
    assert l2_emu < 1e-3, ("dy vs emulation", l2_emu)
    assert l2_ref < 6e-3 and e_dy < 5e-2, ("dy", l2_ref, e_dy)
    assert e_dw < 1e-3, ("dw", e_dw)

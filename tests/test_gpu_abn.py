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


@pytest.mark.parametrize("case", CASES, ids=[f"c{i}" for i in range(len(CASES))])
def test_kept_gram_order_of_the_two_pass_forward(case):
    """The production order of the two-pass units: y^T y + colsum(y) by their own launch (forward), g^T y alone later (backward)
    -- same dW as the one-launch form to fp32 summation order, same fp64 bound."""
    N, Cw, C4, H, W = case
    lib = _lib.load()
    gen = torch.Generator().manual_seed(7 + sum(case))
    y = torch.relu(torch.randn(N, Cw, H, W, generator=gen)).bfloat16().float()
    g = (torch.randn(N, C4, H, W, generator=gen) * (torch.rand(N, C4, H, W, generator=gen) > 0.4)).bfloat16().float()
    w = torch.randn(C4, Cw, generator=gen) / Cw ** 0.5
    wb = w.bfloat16().double()
    cA = (torch.rand(C4, generator=gen) + 0.5).double(); cB = (torch.randn(C4, generator=gen) * 0.05).double(); cC = (torch.randn(C4, generator=gen) * 0.02).double()
    M = N * H * W
    ym = y.double().permute(0, 2, 3, 1).reshape(M, Cw); gm = g.double().permute(0, 2, 3, 1).reshape(M, C4)
    dw_ref = (cA * gm + cB * (ym @ wb.t()) + cC).t() @ ym
    dev = [t.float().to(DEV) for t in (g, y, w, cA, cB, cC)]
    wsp = ws(lib.mmskin_abn_workspace_bytes(N, Cw, C4, H, W))
    out = []
    for fn, launches in (("mmskin_abn_backward", 1), ("mmskin_abn_backward_kept_gram", 2)):
        dy = torch.empty(N, Cw, H, W, device=DEV); dw = torch.empty(C4, Cw, device=DEV)
        n0 = lib.mmskin_wgrad_ring_launches()
        call(fn, *[ptr(t) for t in dev], N, Cw, C4, H, W, ptr(dy), ptr(dw), ptr(wsp), stream())
        torch.cuda.synchronize()
        assert lib.mmskin_wgrad_ring_launches() == n0 + launches
        out.append((dy.cpu(), dw.cpu()))
    assert torch.equal(out[0][0], out[1][0])                       # dy does not involve the Gram matrix
    e = rel_err(out[1][1], dw_ref)
    with open(REPORT, "a") as f:
        f.write(json.dumps(dict(test="abn_backward_kept_gram", case=list(case), dw_max_err_over_rms=e,
                                dw_vs_one_launch=rel_err(out[1][1], out[0][1].double()))) + "\n")
    assert e < 1e-3, e


@pytest.mark.parametrize("case", CASES + [(256, 64, 256, 56, 56)], ids=[f"c{i}" for i in range(len(CASES) + 1)])
def test_conv1x1_statistics_from_the_gram_matrix(case):
    """sum x and sum x^2 per channel of x = conv1x1(y, w) from y^T y and colsum(y) (first pass of the two-pass BatchNorm forward) against
    the fp64 sums of the fp64 product: relative error of the mean and of the variance <= 2e-5 (fp32 Gram entries, double contraction).
    The last case is layer1's production shape (802 816 rows)."""
    N, Cw, C4, H, W = case
    lib = _lib.load()
    gen = torch.Generator().manual_seed(11 + sum(case))
    y = (torch.relu(torch.randn(N, Cw, H, W, generator=gen) + 0.3)).bfloat16().float()
    w = torch.randn(C4, Cw, generator=gen) / Cw ** 0.5
    wb = w.bfloat16().double()
    M = N * H * W
    ym = y.permute(0, 2, 3, 1).reshape(M, Cw)
    s_ref = torch.zeros(C4, dtype=torch.float64); q_ref = torch.zeros(C4, dtype=torch.float64)
    for i in range(0, M, 65536):
        x = ym[i:i + 65536].double() @ wb.t()
        s_ref += x.sum(0); q_ref += (x * x).sum(0)
    ssum = torch.empty(C4, device=DEV); ssq = torch.empty(C4, device=DEV)
    wsp = ws(lib.mmskin_abn_workspace_bytes(N, Cw, C4, H, W))
    yd, wd = y.to(DEV), w.to(DEV)
    call("mmskin_conv1x1_gram_stats", ptr(yd), ptr(wd), N, Cw, C4, H, W, ptr(ssum), ptr(ssq), ptr(wsp), stream())
    torch.cuda.synchronize()
    mean_ref, var_ref = s_ref / M, q_ref / M - (s_ref / M) ** 2
    mean = ssum.double().cpu() / M
    var = ssq.double().cpu() / M - mean ** 2
    e_mean = float(((mean - mean_ref).abs() / var_ref.sqrt()).max())
    e_var = float(((var - var_ref).abs() / var_ref).max())
    with open(REPORT, "a") as f:
        f.write(json.dumps(dict(test="conv1x1_gram_stats", case=list(case), mean_err_over_std=e_mean, var_rel_err=e_var)) + "\n")
    assert e_mean < 2e-5 and e_var < 2e-5, (e_mean, e_var)

"""-m gpu: the DaViT-specific HIP ops against plain PyTorch fp32 on the same inputs -- depthwise 3x3 position encoding (forward, data
and weight gradient; the row-walking weight-gradient kernel and the tapped one), window attention that finds each window's tokens in
the image-major grid (timm davit.py window_partition -> WindowAttention -> window_reverse, reached through
loadImageModelClassifier.py:117-131), and the row-in-registers LayerNorm at the DaViT / BEiT / BERT widths.
Tolerances: fp32 kernels, 1e-4 relative to the reference's max (summation order differs)."""
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV
from mmskin import ops

pytestmark = pytest.mark.gpu


def _chk(a, b, tol=1e-4):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item()), (a - b).abs().max().item()


@pytest.mark.parametrize("N,H,W,C", [(2, 14, 14, 96), (3, 7, 9, 192), (1, 56, 56, 96), (5, 3, 4, 384), (2, 1, 1, 768), (2, 28, 28, 100),
                                     # EfficientNet-B0 / B7 expanded widths of the stride-1 3x3 depthwise layers (C/4 = 36, 60, 336, 960, 16): ADVICE r03
                                     (2, 9, 11, 144), (1, 12, 12, 240), (1, 7, 7, 1344), (1, 5, 6, 3840), (2, 10, 10, 64)])
def test_dwconv3_forward_backward(N, H, W, C):
    g = torch.Generator().manual_seed(N * 1000 + H + C)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) * 0.3
    dy = torch.randn(N, H, W, C, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr.permute(0, 3, 1, 2), wr, padding=1, groups=C).permute(0, 2, 3, 1)
    y_ref.backward(dy)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = ops.dwconv3(xd, wd)
    y.backward(dy.to(DEV))
    _chk(y, y_ref); _chk(xd.grad, xr.grad); _chk(wd.grad, wr.grad, 2e-4)


def test_dwconv3_weight_gradient_tapped_kernel_knob():
    """MMSKIN_DWW_ROWS=0 keeps the tapped weight-gradient kernel reachable (A/B knob): same cases in a fresh interpreter."""
    env = dict(os.environ, MMSKIN_DWW_ROWS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k", "forward_backward"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def _window_reference(qkv, ws):
    B, Hp, Wp, _, H, Dh = qkv.shape
    win = qkv.reshape(B, Hp // ws, ws, Wp // ws, ws, 3, H, Dh).permute(0, 1, 3, 2, 4, 5, 6, 7).reshape(-1, ws * ws, 3, H, Dh)
    q, k, v = (win[:, :, i].permute(0, 2, 1, 3) for i in range(3))            # [nW, H, L, Dh]
    p = torch.softmax(q @ k.transpose(-1, -2) * Dh ** -0.5, dim=-1)
    o = (p @ v).permute(0, 2, 1, 3)                                            # [nW, L, H, Dh]
    return o.reshape(B, Hp // ws, Wp // ws, ws, ws, H, Dh).permute(0, 1, 3, 2, 4, 5, 6).reshape(B, Hp, Wp, H, Dh)


@pytest.mark.parametrize("B,Hp,Wp,ws,H,Dh", [(2, 14, 14, 7, 3, 32), (1, 7, 21, 7, 2, 32), (3, 8, 4, 4, 1, 64), (2, 56, 56, 7, 3, 32), (1, 6, 9, 3, 5, 32)])
def test_window_attention_in_place(B, Hp, Wp, ws, H, Dh):
    g = torch.Generator().manual_seed(B + Hp * 7 + Wp)
    qkv = torch.randn(B, Hp, Wp, 3, H, Dh, generator=g)
    dO = torch.randn(B, Hp, Wp, H, Dh, generator=g)
    ref_in = qkv.double().requires_grad_(True)
    o_ref = _window_reference(ref_in, ws)
    o_ref.backward(dO.double())
    dev = qkv.to(DEV).requires_grad_(True)
    assert ops.window_attention_ok(dev, ws)
    o = ops.window_attention(dev, ws)
    o.backward(dO.to(DEV))
    _chk(o, o_ref); _chk(dev.grad, ref_in.grad, 2e-4)


def test_window_attention_dropout_mask_is_regenerated_in_backward():
    """With dropout the backward regenerates the forward's mask from (seed, offset): the gradient of sum(o * c) w.r.t. v is P'^T c with the
    same P' the forward used, so <dv, v> == sum(o * c) (o is linear in v)."""
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(2, 14, 7, 3, 2, 32, generator=g).to(DEV).requires_grad_(True)
    c = torch.randn(2, 14, 7, 2, 32, generator=g).to(DEV)
    o = ops.window_attention(qkv, 7, dropout_p=0.3, training=True)
    (o * c).sum().backward()
    lhs = (qkv.grad[:, :, :, 2] * qkv.detach()[:, :, :, 2]).sum().item()
    rhs = (o.detach() * c).sum().item()
    assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(rhs)), (lhs, rhs)


@pytest.mark.parametrize("M,N", [(1000, 192), (515, 384), (777, 768), (300, 1024), (130, 1536), (3, 128), (70, 256), (12544, 96)])
def test_layernorm_row_kernels(M, N):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g) * 2 - 0.5; w = torch.rand(N, generator=g) + 0.5; b = torch.randn(N, generator=g) * 0.2
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    dy = torch.randn(M, N, generator=g)
    F.layer_norm(xr, (N,), wr, br).backward(dy.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.layernorm(xd, wd, bd, 1e-5)
    y.backward(dy.to(DEV))
    _chk(y, F.layer_norm(x.double(), (N,), w.double(), b.double())); _chk(xd.grad, xr.grad); _chk(wd.grad, wr.grad, 2e-4); _chk(bd.grad, br.grad, 2e-4)


def _channel_reference(qkv, scale):
    B, N, _, G, Dh = qkv.shape
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))            # [B, G, N, Dh]
    a = torch.softmax((q * scale).transpose(-1, -2) @ k, dim=-1)               # [B, G, Dh, Dh]
    x = (a @ v.transpose(-1, -2)).transpose(-1, -2)                            # [B, G, N, Dh]
    return x.permute(0, 2, 1, 3)                                               # [B, N, G, Dh]


@pytest.mark.parametrize("B,N,G", [(2, 49, 3), (1, 196, 12), (3, 7, 1), (2, 3136, 3), (1, 1, 2), (2, 50, 24), (1, 300, 2), (2, 257, 1), (1, 512, 3)])
def test_channel_attention_token_major(B, N, G):
    """timm davit.py ChannelAttention.forward (dynamic_scale: q * N^-0.5) on the packed [B, N, 3, G, 32] qkv, against the same math in
    fp64 with explicit permutes: output, and d(qkv) through softmax and both products.  Odd token counts (the MFMA walks token pairs)."""
    g = torch.Generator().manual_seed(B * 100 + N + G)
    qkv = torch.randn(B, N, 3, G, 32, generator=g)
    dO = torch.randn(B, N, G, 32, generator=g)
    scale = N ** -0.5
    ref_in = qkv.double().requires_grad_(True)
    x_ref = _channel_reference(ref_in, scale)
    x_ref.backward(dO.double())
    dev = qkv.to(DEV).requires_grad_(True)
    assert ops.channel_attention_ok(dev)
    x = ops.channel_attention(dev, scale)
    x.backward(dO.to(DEV))
    _chk(x, x_ref); _chk(dev.grad, ref_in.grad, 2e-4)


@pytest.mark.parametrize("N,H,W,C", [(2, 14, 14, 96), (1, 56, 56, 96), (3, 7, 5, 192), (2, 2, 3, 768)])
def test_conv_pos_enc_fused(N, H, W, C):
    """x + dwconv3(x, w) + b (timm davit.py ConvPosEnc) as one op: output, dx (gradient through both branches), dw and db."""
    g = torch.Generator().manual_seed(N + H * 3 + C)
    x = torch.randn(N, H, W, C, generator=g); w = torch.randn(C, 1, 3, 3, generator=g) * 0.3; b = torch.randn(C, generator=g)
    dy = torch.randn(N, H, W, C, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    y_ref = xr + F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1, groups=C).permute(0, 2, 3, 1)
    y_ref.backward(dy.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.conv_pos_enc(xd, wd, bd)
    y.backward(dy.to(DEV))
    _chk(y, y_ref); _chk(xd.grad, xr.grad); _chk(wd.grad, wr.grad, 2e-4); _chk(bd.grad, br.grad, 2e-4)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("M,K,N", [(4096, 96, 384), (2304, 192, 768), (100, 64, 256)])
def test_linear_gelu_backward_in_one_call(mode, M, K, N):
    """gelu(x w^T + b) with gradients (timm Mlp.fc1 -> act): the backward applies gelu'(z) inside the dy conversion pass in bf16-operand
    mode (padded 96-wide and 64-multiple shapes) and in a pass of its own otherwise.  Reference: F.gelu(F.linear) in fp64; tolerance
    1e-4 of the maximum in fp32, 2e-2 relative L2 with bf16 operands."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g) * 0.1
    dh = torch.randn(M, N, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    F.gelu(F.linear(xr, wr, br)).backward(dh.double())
    prev = ops.get_linear_dtype()
    ops.set_linear_dtype(mode)
    try:
        xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        h = ops.linear_gelu(xd, wd, bd)
        h.backward(dh.to(DEV))
    finally:
        ops.set_linear_dtype(prev)
    ref = F.gelu(F.linear(x.double(), w.double(), b.double()))
    for got, want in ((h, ref), (xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        got, want = got.detach().cpu().double(), want.detach()
        if mode == "fp32" or M < 2048:
            assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
        else:   # bf16 operands: relative L2 of the whole tensor
            assert ((got - want).norm() / want.norm()).item() < 2e-2


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("M,C", [(4096, 96), (2304, 192), (300, 64)])
def test_mlp_fc1_gelu_fc2(mode, M, C):
    """timm Mlp.forward (fc1 -> GELU -> fc2) as ops.mlp: in bf16-operand mode on the large-GEMM shapes MlpFn keeps the hidden
    activation as pre-activation + bf16 operand only; elsewhere it is linear_gelu + linear.  Output and all five gradients against
    fp64 torch; tolerance 1e-4 of the maximum in fp32, 2e-2 relative L2 with bf16 operands."""
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g)
    w1 = torch.randn(4 * C, C, generator=g) / C ** 0.5; b1 = torch.randn(4 * C, generator=g) * 0.1
    w2 = torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5; b2 = torch.randn(C, generator=g) * 0.1
    dy = torch.randn(M, C, generator=g)
    ref = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y_ref = F.linear(F.gelu(F.linear(ref[0], ref[1], ref[2])), ref[3], ref[4])
    y_ref.backward(dy.double())
    prev = ops.get_linear_dtype()
    ops.set_linear_dtype(mode)
    try:
        dev = [t.to(DEV).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        y = ops.mlp(*dev)
        y.backward(dy.to(DEV))
    finally:
        ops.set_linear_dtype(prev)
    pairs = [(y, y_ref)] + [(d.grad, r.grad) for d, r in zip(dev, ref)]
    for got, want in pairs:
        got, want = got.detach().cpu().double(), want.detach()
        if mode == "fp32" or M < 2048:
            assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
        else:
            assert ((got - want).norm() / want.norm()).item() < 2e-2


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
@pytest.mark.parametrize("M,K", [(4096, 96), (2304, 192), (2048, 384), (100, 64)])
def test_linear_and_mlp_with_fused_residual(mode, M, K):
    """residual + x w^T + b and residual + fc2(gelu(fc1(x))) (a transformer block's skip connections riding on the output pass of the
    bf16 large-GEMM path -- padded 96-wide, 192-wide (separate add: the residual epilogue stores 128-column tiles) and 384-wide --
    and ops.add elsewhere): outputs and the gradients w.r.t. the residual and x against fp64 torch."""
    g = torch.Generator().manual_seed(M + K + 1)
    x = torch.randn(M, K, generator=g); res = torch.randn(M, K, generator=g)
    w = torch.randn(K, K, generator=g) / K ** 0.5; b = torch.randn(K, generator=g) * 0.1
    w1 = torch.randn(4 * K, K, generator=g) / K ** 0.5; b1 = torch.randn(4 * K, generator=g) * 0.1
    w2 = torch.randn(K, 4 * K, generator=g) / (4 * K) ** 0.5; b2 = torch.randn(K, generator=g) * 0.1
    dy = torch.randn(M, K, generator=g)
    prev = ops.get_linear_dtype()
    ops.set_linear_dtype(mode)
    try:
        for case in ("linear", "mlp"):
            rx, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
            dx_, dr_ = x.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
            if case == "linear":
                y_ref = rr + F.linear(rx, w.double(), b.double())
                y = ops.linear(dx_, w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True), residual=dr_)
            else:
                y_ref = rr + F.linear(F.gelu(F.linear(rx, w1.double(), b1.double())), w2.double(), b2.double())
                y = ops.mlp(dx_, *(t.to(DEV).requires_grad_(True) for t in (w1, b1, w2, b2)), residual=dr_)
            y_ref.backward(dy.double()); y.backward(dy.to(DEV))
            for got, want in ((y, y_ref), (dx_.grad, rx.grad), (dr_.grad, rr.grad)):
                got, want = got.detach().cpu().double(), want.detach()
                if mode == "fp32" or M < 2048:
                    assert (got - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item()), case
                else:
                    assert ((got - want).norm() / want.norm()).item() < 2e-2, case
    finally:
        ops.set_linear_dtype(prev)

"""CPU emulation of the package's bf16-OPERAND mode (`mmskin.ops.set_linear_dtype("bf16")`) for the transformer image encoders, applied
to the fp32 oracle models: the yardstick the bf16 gradient-parity tests hold the HIP path against, exactly as
tests/test_gpu_model.py holds the bf16 ResNet step against a bf16-storage emulation.

What the HIP path rounds in that mode (csrc/head.hip, mmskin_linear_forward / _backward): a Linear over >= 2048 rows whose widths are
multiples of 8 (>= 32) runs on the bf16 MFMA GEMM kernels -- forward  y = bf16(x) bf16(W)^T + b  with fp32 accumulation and an fp32
result (rounded to bf16 before the bias when a width is not a multiple of 64: the zero-padded operand path); backward  dx = bf16( bf16(dy) bf16(W) ),  dW = bf16(dy)^T bf16(x)  (fp32 result),  db = sum dy (fp32).  Patch-embedding
convolutions (kernel = stride, no padding) are that Linear on im2col columns.  Everything else -- LayerNorm, softmax attention with
gradients, depthwise convolutions, Linears over fewer rows -- stays fp32 in both.  The emulation rounds exactly those operands on
the CPU and leaves the summation order to torch, so a HIP gradient may differ from it by summation order only."""
import contextlib

import torch
import torch.nn.functional as F

MIN_ROWS = 2048


def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _takes_bf16(rows, K, N):
    return rows >= MIN_ROWS and K % 8 == 0 and N % 8 == 0 and K >= 32 and N >= 32


class _Bf16Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x16, w16 = _rb(x), _rb(w)
        ctx.save_for_backward(x16, w16)
        ctx.has_b = b is not None
        y = x16.reshape(-1, x.shape[-1]) @ w16.t()
        if not (w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0):
            y = _rb(y)      # widths that are not multiples of 64 run on zero-padded operand copies and come back through a bf16 tile
        if b is not None:
            y = y + b
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x16, w16 = ctx.saved_tensors
        d2 = dy.reshape(-1, dy.shape[-1])
        d16 = _rb(d2)
        dx = _rb(d16 @ w16).reshape(x16.shape)
        dw = d16.t() @ x16.reshape(-1, x16.shape[-1])
        return dx, dw, (d2.sum(0) if ctx.has_b else None)


class _Bf16PatchConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride):
        x16, w16 = _rb(x), _rb(w)
        ctx.save_for_backward(x16, w16)
        ctx.stride, ctx.has_b = stride, b is not None
        return _orig_conv2d(x16, w16, b, stride)

    @staticmethod
    def backward(ctx, dy):
        x16, w16 = ctx.saved_tensors
        d16 = _rb(dy)
        dx = _rb(torch.nn.grad.conv2d_input(x16.shape, w16, d16, stride=ctx.stride))
        dw = torch.nn.grad.conv2d_weight(x16, w16.shape, d16, stride=ctx.stride)
        return dx, dw, (dy.sum((0, 2, 3)) if ctx.has_b else None), None


_orig_linear, _orig_conv2d = F.linear, F.conv2d


@contextlib.contextmanager
def bf16_operand_emulation():
    """Inside the block torch.nn.functional.linear / conv2d round what the HIP bf16-operand mode rounds (see the module docstring)."""
    def linear(x, w, b=None):
        rows = x.numel() // x.shape[-1]
        if x.dtype == torch.float32 and _takes_bf16(rows, w.shape[1], w.shape[0]):
            return _Bf16Linear.apply(x, w, b)
        return _orig_linear(x, w, b)

    def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        st = (stride, stride) if isinstance(stride, int) else tuple(stride)
        pd = (padding, padding) if isinstance(padding, int) else padding
        if (x.dtype == torch.float32 and groups == 1 and pd in ((0, 0), "valid") and st == tuple(w.shape[2:]) and dilation in (1, (1, 1))
                and x.shape[2] % st[0] == 0 and x.shape[3] % st[1] == 0):
            rows = x.shape[0] * (x.shape[2] // st[0]) * (x.shape[3] // st[1])
            if _takes_bf16(rows, w.shape[1] * w.shape[2] * w.shape[3], w.shape[0]):
                return _Bf16PatchConv.apply(x, w, b, st)
        return _orig_conv2d(x, w, b, stride, padding, dilation, groups)

    F.linear, F.conv2d = linear, conv2d
    torch.nn.functional.linear, torch.nn.functional.conv2d = linear, conv2d
    try:
        yield
    finally:
        F.linear, F.conv2d = _orig_linear, _orig_conv2d
        torch.nn.functional.linear, torch.nn.functional.conv2d = _orig_linear, _orig_conv2d


def grad_distance_report(g_ref, g_hip, g_emu):
    """Per parameter: relative L2 distance and cosine to the fp32 oracle's gradient, for the HIP path and for the emulation."""
    rows = {}
    for k, r in g_ref.items():
        n = float(r.norm()) + 1e-30
        def dist(g):
            return float((g - r).norm()) / n, float((g * r).sum() / ((float(g.norm()) + 1e-30) * n))
        rows[k] = dist(g_hip[k]) + dist(g_emu[k])      # (l2_hip, cos_hip, l2_emu, cos_emu)
    return rows


def assert_grads_not_worse_than_emulation(rows, slack=1.5, floor=2e-3, cos_slack=0.02):
    bad = {k: v for k, v in rows.items() if v[0] > slack * v[2] + floor or v[1] < v[3] - cos_slack}
    assert not bad, {k: tuple(round(x, 5) for x in v) for k, v in list(bad.items())[:8]}

"""Helpers for the -m gpu parity tests: call the C ABI with torch CUDA tensors."""
import torch

from mmskin import _lib
from mmskin._lib import call, ptr, stream

DEV = "cuda:0"
DT = {"fp32": _lib.F32, "bf16": _lib.BF16}


def ws(nbytes):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=DEV)


def rel_err(got, want):
    """max |got-want| relative to the rms of `want` (scale-aware absolute error)."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    scale = float(want.pow(2).mean().sqrt()) + 1e-30
    return float((got - want).abs().max()) / scale


def conv_forward(x, w, stride, pad, dtype):
    N, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    lib = _lib.load()
    wsp = ws(lib.mmskin_conv2d_workspace_bytes(N, Cin, H, W, Cout, kh, kw, stride, pad))
    y = torch.empty((N, Cout, OH, OW), device=DEV)
    call("mmskin_conv2d_forward", ptr(x), ptr(w), ptr(y), N, Cin, H, W, Cout, kh, kw, stride, pad, DT[dtype],
         ptr(wsp), stream())
    torch.cuda.synchronize()
    return y


def conv_backward(dy, x, w, stride, pad, dtype):
    N, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    lib = _lib.load()
    wsp = ws(lib.mmskin_conv2d_workspace_bytes(N, Cin, H, W, Cout, kh, kw, stride, pad))
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    call("mmskin_conv2d_backward", ptr(dy), ptr(x), ptr(w), ptr(dx), ptr(dw), N, Cin, H, W, Cout, kh, kw, stride, pad,
         DT[dtype], ptr(wsp), stream())
    torch.cuda.synchronize()
    return dx, dw

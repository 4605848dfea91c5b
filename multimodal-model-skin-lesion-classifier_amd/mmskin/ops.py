"""autograd.Function wrappers over the C ABI (fp32 head operators + custom-cnn pieces).

Each Function allocates its outputs with torch (device memory plumbing), launches
the HIP kernels on torch's current stream through ctypes and keeps what the
matching backward entry point needs.  No arithmetic happens in PyTorch here.
"""
import ctypes

import torch

from . import _lib
from ._autograd import no_second_order
from ._lib import call, ptr, stream


def _need_gpu(t, what):
    if not t.is_cuda:
        raise _lib.MMSkinError(
            f"mmskin.{what}: tensors must live on a HIP device (got {t.device}); the MI355X path has "
            "no CPU fallback")


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


@no_second_order
class LinearFn(torch.autograd.Function):
    """y = [res +] x @ w.T + b, optional fused ReLU (nn.Linear [+ nn.ReLU] [+ the block's residual add]).  When the bf16-operand
    large-GEMM path runs and w trains, the bf16 copy of x the forward GEMM consumed is what is saved for the weight gradient (half the
    bytes, no second conversion); the residual is added in the pass that writes y (only offered on that path: see linear())."""

    @staticmethod
    def forward(ctx, x, w, b, relu, res=None):
        _need_gpu(x, "linear")
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        w = _f32c(w)
        M, K = x2.shape
        N = w.shape[0]
        y = torch.empty((M, N), device=x.device, dtype=torch.float32)
        need_w = ctx.needs_input_grad[1]
        pitch = _lib.load().mmskin_linear_x16_pitch(M, K, N) if (need_w or res is not None) else 0
        if res is not None and (not pitch or relu):
            raise RuntimeError("linear: the fused residual needs the bf16-operand large-GEMM path and no ReLU")
        if pitch:
            x16 = torch.empty((M, pitch), device=x.device, dtype=torch.bfloat16)
            r2 = _f32c(res).reshape(M, N) if res is not None else None
            call("mmskin_linear_forward_keep", ptr(x2), ptr(w), ptr(b), ptr(r2), ptr(y), ptr(x16), M, K, N, int(relu), stream())
            ctx.save_for_backward(x16 if need_w else None, w, y if relu else None)
        else:
            call("mmskin_linear_forward", ptr(x2), ptr(w), ptr(b), ptr(y), M, K, N, int(relu), stream())
            ctx.save_for_backward(x2 if need_w else None, w, y if relu else None)
        ctx.kept = bool(pitch) and need_w
        ctx.mk = (M, K)
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        ctx.res_shape = None if res is None else res.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xs, w, y = ctx.saved_tensors
        M, K = ctx.mk
        N = w.shape[0]
        dy2 = _f32c(dy).reshape(M, N)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty((M, K), device=dy.device, dtype=torch.float32) if need_x else None
        dw = torch.empty_like(w) if need_w else None
        db = torch.empty(N, device=dy.device, dtype=torch.float32) if need_b else None
        scratch = torch.empty_like(dy2) if y is not None else None
        if ctx.kept:
            call("mmskin_linear_backward_keep", ptr(dy2), ptr(xs), ptr(w), ptr(y), None, ptr(scratch), ptr(dx), ptr(dw), ptr(db),
                 M, K, N, stream())
        else:
            call("mmskin_linear_backward", ptr(dy2), ptr(xs), ptr(w), ptr(y), ptr(scratch), ptr(dx), ptr(dw), ptr(db),
                 M, K, N, stream())
        dres = dy2.reshape(ctx.res_shape) if ctx.res_shape is not None and ctx.needs_input_grad[4] else None   # relu is never combined with res
        return (dx.reshape(ctx.xshape) if need_x else None), dw, db, None, dres


class LinearRowsFn(torch.autograd.Function):
    """y = x @ w[r0:r1].T + b[r0:r1] for a ROW BLOCK of a stacked parameter (the value projection of nn.MultiheadAttention's
    in_proj_weight / in_proj_bias when a single key makes the q / k projections dead, multimodalIntraInterModal.py:190-197).  The
    backward hands autograd the gradients of the WHOLE parameters: one zero fill for both, the row block written in place by the
    Linear-backward launch -- autograd's own slice backward costs two fills and two copies per call (16 launches per head step)."""

    @staticmethod
    def forward(ctx, x, w, b, r0, r1):
        _need_gpu(x, "linear_rows")
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        w = _f32c(w)
        b = _f32c(b) if b is not None else None
        M, K = x2.shape
        N = r1 - r0
        y = torch.empty((M, N), device=x.device, dtype=torch.float32)
        call("mmskin_linear_forward", ptr(x2), w.data_ptr() + 4 * r0 * K, (b.data_ptr() + 4 * r0) if b is not None else None, ptr(y),
             M, K, N, 0, stream())
        ctx.save_for_backward(x2 if ctx.needs_input_grad[1] else None, w)
        ctx.rows, ctx.mk, ctx.xshape = (r0, r1), (M, K), x.shape
        ctx.bias = None if b is None else (b.shape, b.dtype)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xs, w = ctx.saved_tensors
        (r0, r1), (M, K) = ctx.rows, ctx.mk
        N = r1 - r0
        dy2 = _f32c(dy).reshape(M, N)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.bias is not None and ctx.needs_input_grad[2]
        dx = torch.empty((M, K), device=dy.device, dtype=torch.float32) if need_x else None
        nw, nb = (w.numel() if need_w else 0), (ctx.bias[0].numel() if need_b else 0)
        flat = torch.zeros(nw + nb, device=dy.device, dtype=torch.float32) if nw + nb else None
        dw = flat[:nw].view_as(w) if need_w else None
        db = flat[nw:].view(ctx.bias[0]) if need_b else None
        call("mmskin_linear_backward", ptr(dy2), ptr(xs), w.data_ptr() + 4 * r0 * K, None, None, ptr(dx),
             (dw.data_ptr() + 4 * r0 * K) if need_w else None, (db.data_ptr() + 4 * r0) if need_b else None, M, K, N, stream())
        return (dx.reshape(ctx.xshape) if need_x else None), dw, db, None, None


def linear_rows(x, w, b, r0, r1):
    """nn.functional.linear(x, w[r0:r1], b[r0:r1]) with the gradients of w / b produced whole (exact zeros outside the row block)."""
    if x.dtype != torch.float32 or not x.is_cuda:
        return linear(x, w[r0:r1], b[r0:r1] if b is not None else None)
    return LinearRowsFn.apply(x, w, b, r0, r1)


def lane_dtype(x, module=None, *tensors):
    """torch.bfloat16 when the INFERENCE LANE applies to a layer -- bf16-operand mode and no gradient flows through it (input,
    the module's parameters and `tensors` all without grad, or grad mode off): activations are then handed between the layer's
    Linear / LayerNorm / attention ops as bf16 tensors, without fp32 <-> bf16 conversion passes.  None otherwise: the layer runs
    its ordinary (differentiable, fp32-boundary) ops."""
    if not x.is_cuda or get_linear_dtype() != "bf16":
        return None
    params = tuple(module.parameters()) if module is not None else ()
    return None if _needs_grad(x, *params, *tensors) else torch.bfloat16


def _linear_ex(x, w, b, act, out_dtype):
    """no-grad Linear with bf16 and / or fp32 tensors at the boundary (mmskin_linear_forward_ex)."""
    _need_gpu(x, "linear")
    if _needs_grad(x, w, b):
        raise _lib.MMSkinError("mmskin.linear: bf16 activations are an inference-lane feature (no gradient may flow through them)")
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    if x2.dtype not in (torch.float32, torch.bfloat16):
        x2 = x2.float()
    w = _f32c(w)
    M, K = x2.shape
    N = w.shape[0]
    od = out_dtype or torch.float32
    y = torch.empty((M, N), device=x.device, dtype=od)
    call("mmskin_linear_forward_ex", ptr(x2), _lib.BF16 if x2.dtype == torch.bfloat16 else _lib.F32, ptr(w), ptr(b), ptr(y),
         _lib.BF16 if od == torch.bfloat16 else _lib.F32, M, K, N, int(act), stream())
    return y.reshape(*x.shape[:-1], N)


_frozen_cache = {}   # (data_ptr, ...) -> (versions, weakrefs, bf16 tensor)


def frozen_bf16(*ws):
    """bf16 copy of one weight (or of several stacked along dim 0: the q / k / v projections of a BERT layer as ONE [3E, E] operand)
    that no gradient flows into, converted once and reused until a source tensor changes (its version counter moves: optimizer
    step, load_state_dict, .copy_) or is replaced.  Saves the per-step fp32 -> bf16 pass over every frozen encoder weight."""
    import weakref
    key = tuple(w.data_ptr() for w in ws)
    vers = tuple(w._version for w in ws)
    hit = _frozen_cache.get(key)
    if hit is not None and hit[0] == vers and all(r() is w for r, w in zip(hit[1], ws)):
        return hit[2]
    with torch.no_grad():
        w16 = (ws[0] if len(ws) == 1 else torch.cat([w.reshape(w.shape[0], -1) for w in ws], 0)).to(torch.bfloat16).contiguous()
    if len(_frozen_cache) > 4096:
        _frozen_cache.clear()
    drop = lambda _ref, key=key: _frozen_cache.pop(key, None)       # a source tensor died (model deleted / moved): free its bf16 copy too
    _frozen_cache[key] = (vers, tuple(weakref.ref(w, drop) for w in ws), w16)
    return w16


def invalidate_frozen_cache():
    """Drop every cached bf16 weight copy.  Anything that rewrites parameters through `.data` (which does not move the version
    counter) must call this; in-place ops on the parameter or a detach() view are seen by the version check on their own."""
    _frozen_cache.clear()


def lane_ok(x, N, K, fused_tail=False):
    """does mmskin_linear_lane take this shape?  (rows >= 2048, 64-multiple widths; 128-multiple N with a fused residual tail)"""
    M = x.numel() // x.shape[-1]
    return x.is_cuda and get_linear_dtype() == "bf16" and M >= 2048 and K % 64 == 0 and N % (128 if fused_tail else 64) == 0


def linear_lane(x, ws, b=None, act=0, gamma=None, residual=None, drop_p=0.0, training=False, out_dtype=None):
    """y = residual + gamma * dropout(act(x @ W.T + b)) as ONE GEMM launch on the inference lane (no gradient flows: frozen
    encoders / evaluation; bf16-operand mode).  ws: a weight or a tuple of weights stacked along the output dimension; their
    bf16 copy is cached (frozen_bf16).  act 0 none / 1 ReLU / 2 exact GELU.  With residual / gamma / dropout the result is the
    fp32 residual stream; otherwise out_dtype (fp32 default).  Shapes the fused kernel does not take are composed from the
    separate ops (same arithmetic, more launches)."""
    ws = ws if isinstance(ws, (tuple, list)) else (ws,)
    _need_gpu(x, "linear_lane")
    if _needs_grad(x, *ws, b, gamma, residual):
        raise _lib.MMSkinError("mmskin.linear_lane: an inference-lane op (no gradient may flow through it)")
    p = float(drop_p) if training else 0.0
    tail = gamma is not None or residual is not None or p > 0.0
    K = x.shape[-1]
    N = sum(w.shape[0] for w in ws)
    if not lane_ok(x, N, K, tail):
        w = ws[0] if len(ws) == 1 else torch.cat(list(ws), 0)
        h = _linear_ex(x, w, b, act, None if tail else out_dtype)
        if p > 0.0:
            h = dropout(h, p, True)
        if gamma is not None:
            return scale_add(residual if residual is not None else torch.zeros_like(h), h, gamma)
        return add(h, residual) if residual is not None else h
    x2 = x.reshape(-1, K).contiguous()
    if x2.dtype not in (torch.float32, torch.bfloat16):
        x2 = x2.float()
    M = x2.shape[0]
    od = torch.float32 if tail else (out_dtype or torch.float32)
    y = torch.empty((M, N), device=x.device, dtype=od)
    res = None if residual is None else _f32c(residual).reshape(M, N)
    seed, offset = _dropout_state(p, M * N)
    call("mmskin_linear_lane", ptr(x2), _lib.BF16 if x2.dtype == torch.bfloat16 else _lib.F32, ptr(frozen_bf16(*ws)), _lib.BF16,
         ptr(None if b is None else _f32c(b)), ptr(None if gamma is None else _f32c(gamma)), ptr(res), p, int(seed), int(offset),
         ptr(y), _lib.BF16 if od == torch.bfloat16 else _lib.F32, M, K, N, int(act), stream())
    return y.reshape(*x.shape[:-1], N)


def linear(x, w, b=None, relu=False, out_dtype=None, residual=None):
    """nn.Linear (+ ReLU).  residual: y = residual + x @ w.T + b -- added in the pass that writes y when the bf16-operand large-GEMM
    path takes the shape (a transformer block's skip connection), by ops.add otherwise."""
    if x.dtype == torch.bfloat16 or out_dtype == torch.bfloat16:
        y = linear_lane(x, w, b, 1 if relu else 0, out_dtype=out_dtype)
        return y if residual is None else add(y, residual.reshape(y.shape))
    if residual is not None:
        M = x.numel() // x.shape[-1]
        if (not relu and x.is_cuda and x.dtype == torch.float32 and residual.dtype == torch.float32
                and _lib.load().mmskin_linear_x16_pitch(M, x.shape[-1], w.shape[0])):
            return LinearFn.apply(x, w, b, False, residual)
        y = LinearFn.apply(x, w, b, relu)
        return add(y, residual.reshape(y.shape))
    return LinearFn.apply(x, w, b, relu)


@no_second_order
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim, optional fused ReLU."""

    @staticmethod
    def forward(ctx, x, g, b, eps, relu):
        _need_gpu(x, "layernorm")
        N = x.shape[-1]
        x2 = _f32c(x).reshape(-1, N)
        M = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        call("mmskin_layernorm_forward", ptr(x2), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(rstd), M, N, float(eps),
             int(relu), stream())
        ctx.save_for_backward(x2, g, b, mean, rstd)
        ctx.relu = relu
        ctx.xshape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g, b, mean, rstd = ctx.saved_tensors
        M, N = x2.shape
        dy2 = _f32c(dy).reshape(M, N)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        dg = torch.empty_like(g) if ctx.needs_input_grad[1] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[2] else None
        call("mmskin_layernorm_backward", ptr(dy2), ptr(x2), ptr(g), ptr(b), ptr(mean), ptr(rstd), ptr(dx), ptr(dg),
             ptr(db), M, N, int(ctx.relu), stream())
        return (dx.reshape(ctx.xshape) if dx is not None else None), dg, db, None, None


def layernorm(x, g, b, eps=1e-5, relu=False, out_dtype=None, keep_f32=False):
    """nn.LayerNorm.  out_dtype=torch.bfloat16 (inference lane): the result as a bf16 tensor -- and with keep_f32 also the fp32
    one (post-LN residual streams), returned as (fp32, bf16) -- from one pass over the row."""
    N = x.shape[-1]
    if out_dtype == torch.bfloat16 and not relu and N % 4 == 0 and N <= 2048 and not _needs_grad(x, g, b):
        _need_gpu(x, "layernorm")
        x2 = _f32c(x).reshape(-1, N)
        y16 = torch.empty(x2.shape, device=x.device, dtype=torch.bfloat16)
        y32 = torch.empty_like(x2) if keep_f32 else None
        call("mmskin_layernorm_forward_mixed", ptr(x2), ptr(_f32c(g)), ptr(_f32c(b)), ptr(y32), ptr(y16), x2.shape[0], N, float(eps), stream())
        return (y32.reshape(x.shape), y16.reshape(x.shape)) if keep_f32 else y16.reshape(x.shape)
    y = LayerNormFn.apply(x, g, b, eps, relu)
    return (y, y) if keep_f32 else y


@no_second_order
class SigmoidGateFn(torch.autograd.Function):
    """sigmoid(z) * v."""

    @staticmethod
    def forward(ctx, z, v):
        _need_gpu(z, "sigmoid_gate")
        z, v = _f32c(z), _f32c(v)
        out = torch.empty_like(z)
        call("mmskin_sigmoid_gate_forward", ptr(z), ptr(v), ptr(out), z.numel(), stream())
        ctx.save_for_backward(z, v)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, v = ctx.saved_tensors
        dout = _f32c(dout)
        dz, dv = torch.empty_like(z), torch.empty_like(v)
        call("mmskin_sigmoid_gate_backward", ptr(dout), ptr(z), ptr(v), ptr(dz), ptr(dv), z.numel(), stream())
        return dz, dv


sigmoid_gate = SigmoidGateFn.apply


@no_second_order
class GatedMixFn(torch.autograd.Function):
    """g*a + (1-g)*q with g = sigmoid(z)."""

    @staticmethod
    def forward(ctx, z, a, q):
        _need_gpu(z, "gated_mix")
        z, a, q = _f32c(z), _f32c(a), _f32c(q)
        out = torch.empty_like(z)
        call("mmskin_gated_mix_forward", ptr(z), ptr(a), ptr(q), ptr(out), z.numel(), stream())
        ctx.save_for_backward(z, a, q)
        return out

    @staticmethod
    def backward(ctx, dout):
        z, a, q = ctx.saved_tensors
        dout = _f32c(dout)
        dz, da, dq = torch.empty_like(z), torch.empty_like(a), torch.empty_like(q)
        call("mmskin_gated_mix_backward", ptr(dout), ptr(z), ptr(a), ptr(q), ptr(dz), ptr(da), ptr(dq), z.numel(),
             stream())
        return dz, da, dq


gated_mix = GatedMixFn.apply


@no_second_order
class MetaBlockGateFn(torch.autograd.Function):
    """sigmoid(tanh(V*t1) + t2)."""

    @staticmethod
    def forward(ctx, V, t1, t2):
        _need_gpu(V, "metablock_gate")
        V, t1, t2 = _f32c(V), _f32c(t1), _f32c(t2)
        out = torch.empty_like(V)
        call("mmskin_metablock_gate_forward", ptr(V), ptr(t1), ptr(t2), ptr(out), V.numel(), stream())
        ctx.save_for_backward(V, t1, t2)
        return out

    @staticmethod
    def backward(ctx, dout):
        V, t1, t2 = ctx.saved_tensors
        dout = _f32c(dout)
        dV, d1, d2 = torch.empty_like(V), torch.empty_like(t1), torch.empty_like(t2)
        call("mmskin_metablock_gate_backward", ptr(dout), ptr(V), ptr(t1), ptr(t2), ptr(dV), ptr(d1), ptr(d2),
             V.numel(), stream())
        return dV, d1, d2


metablock_gate = MetaBlockGateFn.apply


@no_second_order
class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset):
        _need_gpu(x, "dropout")
        x = _f32c(x)
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        call("mmskin_dropout_forward", ptr(x), ptr(y), ptr(mask), x.numel(), float(p), int(seed), int(offset), stream())
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        call("mmskin_dropout_backward", ptr(dy), ptr(mask), ptr(dx), dy.numel(), float(ctx.p), stream())
        return dx, None, None, None


_dropout_counter = [0]


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    # seed from torch's generator state so torch.manual_seed controls the masks; the call consumes counters [offset, offset + numel)
    seed, offset = _dropout_state(p, x.numel())
    return DropoutFn.apply(x, p, seed, offset)


@no_second_order
class Concat2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, "concat2")
        a, b = _f32c(a), _f32c(b)
        M, Na = a.shape
        Nb = b.shape[1]
        out = torch.empty((M, Na + Nb), device=a.device, dtype=torch.float32)
        call("mmskin_concat2_forward", ptr(a), ptr(b), ptr(out), M, Na, Nb, stream())
        ctx.dims = (M, Na, Nb)
        return out

    @staticmethod
    def backward(ctx, dout):
        M, Na, Nb = ctx.dims
        dout = _f32c(dout)
        da = torch.empty((M, Na), device=dout.device, dtype=torch.float32)
        db = torch.empty((M, Nb), device=dout.device, dtype=torch.float32)
        call("mmskin_concat2_backward", ptr(dout), ptr(da), ptr(db), M, Na, Nb, stream())
        return da, db


concat2 = Concat2Fn.apply


@no_second_order
class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(Dh)) v on [B, H, L, Dh] tensors; optional dropout on the probabilities."""

    @staticmethod
    def forward(ctx, q, k, v, drop_p, seed, offset):
        _need_gpu(q, "attention")
        q, k, v = _f32c(q), _f32c(k), _f32c(v)
        B, H, L, Dh = q.shape
        o = torch.empty_like(q)
        ctx.rng = (float(drop_p), int(seed), int(offset))
        ctx.rows = _rows_ok(L, Dh)
        if ctx.rows:   # one wave per head, row log-sum-exp instead of the [B, H, L, L] probabilities
            lse = torch.empty((B, H, L), device=q.device, dtype=torch.float32)
            st = _i64x3(H * L * Dh, L * Dh, Dh)
            call("mmskin_attention_rows_forward", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), B, H, L, Dh, st, st, Dh ** -0.5, *ctx.rng, stream())
            ctx.save_for_backward(q, k, v, o, lse)
            return o
        p = torch.empty((B, H, L, L), device=q.device, dtype=torch.float32)
        call("mmskin_attention_forward", ptr(q), ptr(k), ptr(v), ptr(o), ptr(p), B, H, L, Dh, float(drop_p), int(seed),
             int(offset), stream())
        ctx.save_for_backward(q, k, v, p)
        return o

    @staticmethod
    def backward(ctx, dO):
        dO = _f32c(dO)
        if ctx.rows:
            q, k, v, o, lse = ctx.saved_tensors
            B, H, L, Dh = q.shape
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            st = _i64x3(H * L * Dh, L * Dh, Dh)
            call("mmskin_attention_rows_backward", ptr(dO), ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(dq), ptr(dk), ptr(dv), B, H, L, Dh,
                 st, st, Dh ** -0.5, *ctx.rng, stream())
            return dq, dk, dv, None, None, None
        q, k, v, p = ctx.saved_tensors
        B, H, L, Dh = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        call("mmskin_attention_backward", ptr(dO), ptr(q), ptr(k), ptr(v), ptr(p), ptr(dq), ptr(dk), ptr(dv), B, H, L,
             Dh, *ctx.rng, stream())
        return dq, dk, dv, None, None, None


def _rows_ok(L, Dh):
    """shapes of the one-wave-per-head attention kernels (mmskin_attention_rows_*)"""
    return L <= 64 and Dh in (32, 64)


def _i64x3(a, b, c):
    import ctypes
    return (ctypes.c_int64 * 3)(int(a), int(b), int(c))


@no_second_order
class AttentionPackedFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(Dh)) v straight on the packed [B, L, 3, H, Dh] output of a fused qkv Linear -> [B, L, H, Dh] (token-major,
    what the output projection reads); the backward writes d(qkv) in the packed layout.  No permute / contiguous copies in either
    direction (timm Attention.forward / DaViT WindowAttention with gradients: window attention of 49 tokens, Dh 32)."""

    @staticmethod
    def forward(ctx, qkv, drop_p, seed, offset):
        _need_gpu(qkv, "attention_packed")
        qkv = _f32c(qkv)
        B, L, three, H, Dh = qkv.shape
        o = torch.empty((B, L, H, Dh), device=qkv.device, dtype=torch.float32)
        lse = torch.empty((B, H, L), device=qkv.device, dtype=torch.float32)
        ctx.rng = (float(drop_p), int(seed), int(offset))
        step = H * Dh * 4
        qs, os_ = _i64x3(L * 3 * H * Dh, Dh, 3 * H * Dh), _i64x3(L * H * Dh, Dh, H * Dh)
        base = qkv.data_ptr()
        import ctypes
        call("mmskin_attention_rows_forward", ctypes.c_void_p(base), ctypes.c_void_p(base + step), ctypes.c_void_p(base + 2 * step), ptr(o),
             ptr(lse), B, H, L, Dh, qs, os_, Dh ** -0.5, *ctx.rng, stream())
        ctx.save_for_backward(qkv, o, lse)
        return o

    @staticmethod
    def backward(ctx, dO):
        import ctypes
        qkv, o, lse = ctx.saved_tensors
        B, L, three, H, Dh = qkv.shape
        dO = _f32c(dO)
        dqkv = torch.empty_like(qkv)
        step = H * Dh * 4
        qs, os_ = _i64x3(L * 3 * H * Dh, Dh, 3 * H * Dh), _i64x3(L * H * Dh, Dh, H * Dh)
        base, dbase = qkv.data_ptr(), dqkv.data_ptr()
        call("mmskin_attention_rows_backward", ptr(dO), ctypes.c_void_p(base), ctypes.c_void_p(base + step), ctypes.c_void_p(base + 2 * step),
             ptr(o), ptr(lse), ctypes.c_void_p(dbase), ctypes.c_void_p(dbase + step), ctypes.c_void_p(dbase + 2 * step), B, H, L, Dh,
             qs, os_, Dh ** -0.5, *ctx.rng, stream())
        return dqkv, None, None, None


@no_second_order
class WindowAttentionFn(torch.autograd.Function):
    """Window attention on the packed qkv of an image-major token grid, qkv [B, Hp, Wp, 3, H, Dh] -> [B, Hp, Wp, H, Dh]: every ws x ws
    window attends over its own tokens where they sit (mmskin_window_attention_*), so timm's window_partition / window_reverse
    (davit.py SpatialBlock.forward) cost no copy in either direction."""

    @staticmethod
    def forward(ctx, qkv, ws, drop_p, seed, offset):
        import ctypes
        _need_gpu(qkv, "window_attention")
        qkv = _f32c(qkv)
        B, Hp, Wp, three, H, Dh = qkv.shape
        nwy, nwx = Hp // ws, Wp // ws
        o = torch.empty((B, Hp, Wp, H, Dh), device=qkv.device, dtype=torch.float32)
        lse = torch.empty((B * nwy * nwx, H, ws * ws), device=qkv.device, dtype=torch.float32)
        ctx.rng = (float(drop_p), int(seed), int(offset))
        ctx.geom = (B, nwy, nwx, ws, H, Dh)
        step = H * Dh * 4
        base = qkv.data_ptr()
        call("mmskin_window_attention_forward", ctypes.c_void_p(base), ctypes.c_void_p(base + step), ctypes.c_void_p(base + 2 * step), ptr(o),
             ptr(lse), B, nwy, nwx, ws, H, Dh, 3 * H * Dh, Dh, H * Dh, Dh, Dh ** -0.5, *ctx.rng, stream())
        ctx.save_for_backward(qkv, o, lse)
        return o

    @staticmethod
    def backward(ctx, dO):
        import ctypes
        qkv, o, lse = ctx.saved_tensors
        B, nwy, nwx, ws, H, Dh = ctx.geom
        dO = _f32c(dO)
        dqkv = torch.empty_like(qkv)
        step = H * Dh * 4
        base, dbase = qkv.data_ptr(), dqkv.data_ptr()
        call("mmskin_window_attention_backward", ptr(dO), ctypes.c_void_p(base), ctypes.c_void_p(base + step), ctypes.c_void_p(base + 2 * step),
             ptr(o), ptr(lse), ctypes.c_void_p(dbase), ctypes.c_void_p(dbase + step), ctypes.c_void_p(dbase + 2 * step), B, nwy, nwx, ws, H, Dh,
             3 * H * Dh, Dh, H * Dh, Dh, Dh ** -0.5, *ctx.rng, stream())
        return dqkv, None, None, None, None


def window_attention_ok(qkv, ws):
    """shapes mmskin_window_attention_* takes: fp32 packed qkv [B, Hp, Wp, 3, H, Dh] on the GPU, Hp / Wp multiples of ws, ws*ws <= 64, Dh 32 / 64"""
    return (qkv.is_cuda and qkv.dtype == torch.float32 and qkv.dim() == 6 and qkv.shape[3] == 3 and qkv.is_contiguous()
            and qkv.shape[1] % ws == 0 and qkv.shape[2] % ws == 0 and _rows_ok(ws * ws, qkv.shape[5]))


def window_attention(qkv, ws, dropout_p=0.0, training=False):
    """softmax(q k^T / sqrt(Dh)) v inside every ws x ws window of the token grid; qkv [B, Hp, Wp, 3, H, Dh] -> [B, Hp, Wp, H, Dh]."""
    B, Hp, Wp, _, H, Dh = qkv.shape
    p = dropout_p if training else 0.0
    nw = B * (Hp // ws) * (Wp // ws)
    seed, offset = _dropout_state(p, nw * H * (ws * ws) ** 2)
    return WindowAttentionFn.apply(qkv, ws, p, seed, offset)


@no_second_order
class ChannelAttentionFn(torch.autograd.Function):
    """DaViT channel attention on the packed qkv of a fused Linear, qkv [B, N, 3, G, 32] -> [B, N, G, 32] (token-major, what the output
    projection reads): A = softmax(scale q^T k) over each group's 32 channels, x = (A v^T)^T (timm davit.py ChannelAttention.forward).
    No permute / contiguous copies; the backward writes d(qkv) in the packed layout."""

    @staticmethod
    def forward(ctx, qkv, scale):
        import ctypes
        _need_gpu(qkv, "channel_attention")
        qkv = _f32c(qkv)
        B, N, three, G, Dh = qkv.shape
        x = torch.empty((B, N, G, Dh), device=qkv.device, dtype=torch.float32)
        attn = torch.empty((B * G, Dh, Dh), device=qkv.device, dtype=torch.float32)
        step = G * Dh * 4
        base = qkv.data_ptr()
        ctx.scale = float(scale)
        ns = _lib.load().mmskin_channel_attention_scratch_floats(B, G, N)
        scratch = torch.empty(ns, device=qkv.device, dtype=torch.float32) if ns else None
        call("mmskin_channel_attention_forward", ctypes.c_void_p(base), ctypes.c_void_p(base + step), ctypes.c_void_p(base + 2 * step), ptr(x),
             ptr(attn), ptr(scratch) if scratch is not None else None, B, G, N, Dh, 3 * G * Dh, N * 3 * G * Dh, G * Dh, N * G * Dh, ctx.scale, stream())
        ctx.save_for_backward(qkv, attn)
        return x

    @staticmethod
    def backward(ctx, dO):
        import ctypes
        qkv, attn = ctx.saved_tensors
        B, N, three, G, Dh = qkv.shape
        dO = _f32c(dO)
        dqkv = torch.empty_like(qkv)
        step = G * Dh * 4
        base, dbase = qkv.data_ptr(), dqkv.data_ptr()
        ns = _lib.load().mmskin_channel_attention_scratch_floats(B, G, N)
        scratch = torch.empty(ns, device=qkv.device, dtype=torch.float32) if ns else None
        call("mmskin_channel_attention_backward", ptr(dO), ctypes.c_void_p(base), ctypes.c_void_p(base + step), ctypes.c_void_p(base + 2 * step),
             ptr(attn), ctypes.c_void_p(dbase), ctypes.c_void_p(dbase + step), ctypes.c_void_p(dbase + 2 * step),
             ptr(scratch) if scratch is not None else None, B, G, N, Dh,
             3 * G * Dh, N * 3 * G * Dh, G * Dh, N * G * Dh, ctx.scale, stream())
        return dqkv, None


def channel_attention_ok(qkv):
    """shapes mmskin_channel_attention_* takes: fp32 packed qkv [B, N, 3, G, 32] on the GPU"""
    return qkv.is_cuda and qkv.dtype == torch.float32 and qkv.dim() == 5 and qkv.shape[2] == 3 and qkv.shape[4] == 32 and qkv.is_contiguous()


def channel_attention(qkv, scale):
    return ChannelAttentionFn.apply(qkv, scale)


def attention_packed(qkv, dropout_p=0.0, training=False, mask_add=None, bias=None, causal=False):
    """Attention on the packed output of a fused qkv Linear, qkv [B, L, 3, H, Dh] -> [B, L, H, Dh].  Picks, in order: the fused bf16
    kernel (inference lane), the one-wave-per-head fp32 kernels reading the packed tensor in place (L <= 64, Dh 32 / 64, no mask /
    bias: window attention with gradients), else attention_blhd on the three views."""
    B, L, _, H, Dh = qkv.shape
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    if (qkv.is_cuda and qkv.dtype == torch.float32 and mask_add is None and bias is None and not causal and _rows_ok(L, Dh)
            and not _flash_ok(q, k, v, mask_add, bias, B * H) and qkv.is_contiguous()):
        p = dropout_p if training else 0.0
        seed, offset = _dropout_state(p, B * H * L * L)
        return AttentionPackedFn.apply(qkv, p, seed, offset)
    return attention_blhd(q, k, v, dropout_p, training, mask_add, bias, causal)


def _bmm(a, b, c, batch, M, N, K, sam, sak, sab, sbn, sbk, sbb, ldc, scb):
    call("mmskin_bmm", ptr(a), ptr(b), ptr(c), batch, M, N, K, sam, sak, sab, sbn, sbk, sbb, ldc, scb, stream())


@no_second_order
class LongAttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(Dh) + mask) v for sequences whose score matrix does not fit one workgroup's LDS
    (BERT: L = 512): strided batched GEMMs + a row-softmax kernel, the probabilities kept for backward.
    q, k, v [B, H, L, Dh]; mask_add [B, L] additive key mask or None; dropout on the probabilities when drop_p > 0."""

    @staticmethod
    def forward(ctx, q, k, v, mask_add, drop_p, seed, offset, bias=None, causal=False):
        _need_gpu(q, "attention")
        q, k, v = _f32c(q), _f32c(k), _f32c(v)
        B, H, L, Dh = q.shape
        BH = B * H
        scores = torch.empty((B, H, L, L), device=q.device, dtype=torch.float32)
        _bmm(q, k, scores, BH, L, L, Dh, Dh, 1, L * Dh, Dh, 1, L * Dh, L, L * L)
        probs = torch.empty_like(scores)
        m = _f32c(mask_add) if mask_add is not None else None
        bs = _f32c(bias) if bias is not None else None                              # [H, L, L], shared by the batch
        call("mmskin_softmax_forward", ptr(scores), ptr(m) if m is not None else None, ptr(bs) if bs is not None else None,
             ptr(probs), BH * L, L, H * L, 1.0 / Dh ** 0.5, int(causal), stream())
        ctx.has_bias = bias is not None
        del scores
        dmask = None
        pd = probs
        if drop_p > 0.0:
            pd = torch.empty_like(probs)
            dmask = torch.empty(probs.shape, device=q.device, dtype=torch.uint8)
            call("mmskin_attn_dropout_forward", ptr(probs), ptr(pd), ptr(dmask), probs.numel(), L, float(drop_p), int(seed), int(offset), stream())
        o = torch.empty_like(q)
        _bmm(pd, v, o, BH, L, Dh, L, L, 1, L * L, 1, Dh, L * Dh, Dh, L * Dh)          # o[i][d] = sum_j pd[i][j] v[j][d]
        ctx.save_for_backward(q, k, v, probs, pd if drop_p > 0.0 else None, dmask)
        ctx.drop_p = float(drop_p)
        return o

    @staticmethod
    def backward(ctx, dO):
        q, k, v, probs, pd, dmask = ctx.saved_tensors
        B, H, L, Dh = q.shape
        BH = B * H
        dO = _f32c(dO)
        pdrop = pd if pd is not None else probs
        dv = torch.empty_like(v)
        _bmm(pdrop, dO, dv, BH, L, Dh, L, 1, L, L * L, 1, Dh, L * Dh, Dh, L * Dh)        # dv[j][d] = sum_i pd[i][j] dO[i][d]
        dp = torch.empty_like(probs)
        _bmm(dO, v, dp, BH, L, L, Dh, Dh, 1, L * Dh, Dh, 1, L * Dh, L, L * L)           # dp[i][j] = sum_d dO[i][d] v[j][d]
        if dmask is not None:
            dp2 = torch.empty_like(dp)
            call("mmskin_dropout_backward", ptr(dp), ptr(dmask), ptr(dp2), dp.numel(), ctx.drop_p, stream())
            dp = dp2
        ds = torch.empty_like(dp)
        call("mmskin_softmax_backward", ptr(dp), ptr(probs), ptr(ds), BH * L, L, 1.0 / Dh ** 0.5, stream())
        del dp
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        _bmm(ds, k, dq, BH, L, Dh, L, L, 1, L * L, 1, Dh, L * Dh, Dh, L * Dh)            # dq[i][d] = sum_j ds[i][j] k[j][d]
        _bmm(ds, q, dk, BH, L, Dh, L, 1, L, L * L, 1, Dh, L * Dh, Dh, L * Dh)            # dk[j][d] = sum_i ds[i][j] q[i][d]
        dbias = None
        if ctx.has_bias and ctx.needs_input_grad[7]:       # d/d(bias) = d/d(scaled scores) summed over the batch = ds / scale
            dbias = torch.empty((H, L, L), device=q.device, dtype=torch.float32)
            call("mmskin_colsum", ptr(ds), ptr(dbias), B, H * L * L, stream())
            dbias.mul_(Dh ** 0.5)
        return dq, dk, dv, None, None, None, None, dbias, None


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


def _flash_ok(q, k, v, mask_add, bias, bh):
    """The fused bf16 attention kernel (csrc/flash_attn.hip) serves bf16-operand mode whenever no gradient has to flow
    through the attention (frozen encoders -- the reference's default `frozen_weights` -- and inference); trainable blocks
    keep the unfused path, which saves the probabilities for its backward."""
    if not (get_linear_dtype() == "bf16" and q.shape[-1] in (32, 64) and q.is_cuda and q.shape == k.shape == v.shape
            and q.dtype == k.dtype == v.dtype and q.dtype in (torch.float32, torch.bfloat16)
            and not _needs_grad(q, k, v, bias) and all(t.stride(-1) == 1 for t in (q, k, v))):
        return False
    # the kernel's grid is (query tiles, batch * heads): batch * heads <= 65535 (flash_attn.hip ARG_CHECK); larger launches (DaViT
    # window attention on >= 342 images: 64 windows x 3 heads each) take the rows / unfused path instead of raising
    return bh <= 65535


def _flash_forward(q, k, v, out, dims, strides, mask_add, bias, causal, p, seed, offset):
    B, H, L, Dh = dims
    if bias is not None and tuple(bias.shape) != (H, L, L):
        raise _lib.MMSkinError(f"mmskin.attention: bias must be [H, L, L] = {(H, L, L)}, got {tuple(bias.shape)}")
    if mask_add is not None and tuple(mask_add.shape) != (B, L):
        raise _lib.MMSkinError(f"mmskin.attention: mask_add must be [B, L] = {(B, L)}, got {tuple(mask_add.shape)}")
    st = (ctypes.c_int64 * 12)(*strides)
    m = _f32c(mask_add) if mask_add is not None else None
    bs = _f32c(bias) if bias is not None else None
    call("mmskin_flash_attention_forward", ptr(q), ptr(k), ptr(v), ptr(m) if m is not None else None,
         ptr(bs) if bs is not None else None, ptr(out), None, B, H, L, Dh, st,
         _lib.BF16 if q.dtype == torch.bfloat16 else _lib.F32, 1.0 / Dh ** 0.5, int(causal), float(p), int(seed), int(offset), stream())
    return out


def _flash_train_ok(q, k, v, bias, bh):
    """Trainable attention on the fused kernels (forward with the row log-sum-exp kept, backward recomputing the probabilities:
    csrc/flash_attn_bwd.hip) -- bf16-operand mode, gradients required, head dim 32 / 64, more than 64 tokens (shorter sequences have
    the one-wave-per-head fp32 kernels).  MMSKIN_FLASH_BWD=0 keeps the unfused fp32 chain."""
    import os
    return (get_linear_dtype() == "bf16" and os.environ.get("MMSKIN_FLASH_BWD", "1") != "0" and q.is_cuda and q.shape[-1] in (32, 64)
            and q.shape == k.shape == v.shape and q.shape[-2] > 64 and bh <= 65535 and _needs_grad(q, k, v, bias))


@no_second_order
class FlashAttnFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(Dh) + bias + mask) v with gradients, fused: q, k, v [B, H, L, Dh] fp32 (rounded to bf16 operands inside),
    mask_add [B, L] or None, bias [H, L, L] or None (its gradient is the sum of dS over the batch), dropout on the probabilities."""

    @staticmethod
    def forward(ctx, q, k, v, mask_add, bias, causal, p, seed, offset):
        q, k, v = _f32c(q), _f32c(k), _f32c(v)
        B, H, L, Dh = q.shape
        out = torch.empty_like(q)
        lse = torch.empty((B, H, L), device=q.device, dtype=torch.float32)
        if bias is not None and tuple(bias.shape) != (H, L, L):
            raise _lib.MMSkinError(f"mmskin.attention: bias must be [H, L, L] = {(H, L, L)}, got {tuple(bias.shape)}")
        if mask_add is not None and tuple(mask_add.shape) != (B, L):
            raise _lib.MMSkinError(f"mmskin.attention: mask_add must be [B, L] = {(B, L)}, got {tuple(mask_add.shape)}")
        m = _f32c(mask_add) if mask_add is not None else None
        bs = _f32c(bias) if bias is not None else None
        st = (ctypes.c_int64 * 12)(*([q.stride(0), q.stride(1), q.stride(2)] * 4))
        call("mmskin_flash_attention_forward", ptr(q), ptr(k), ptr(v), ptr(m) if m is not None else None, ptr(bs) if bs is not None else None,
             ptr(out), ptr(lse), B, H, L, Dh, st, _lib.F32, 1.0 / Dh ** 0.5, int(causal), float(p), int(seed), int(offset), stream())
        ctx.save_for_backward(q, k, v, out, lse, m, bs)
        ctx.cfg = (bool(causal), float(p), int(seed), int(offset))
        return out

    @staticmethod
    def backward(ctx, dO):
        q, k, v, out, lse, m, bs = ctx.saved_tensors
        causal, p, seed, offset = ctx.cfg
        B, H, L, Dh = q.shape
        dO = _f32c(dO)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        delta = torch.empty((B, H, L), device=q.device, dtype=torch.float32)
        bT = bs.transpose(1, 2).contiguous() if bs is not None else None
        ds = torch.empty((B, H, L, L), device=q.device, dtype=torch.float32) if bs is not None and ctx.needs_input_grad[4] else None
        st = (ctypes.c_int64 * 15)(*([q.stride(0), q.stride(1), q.stride(2)] * 5))
        call("mmskin_flash_attention_backward", ptr(q), ptr(k), ptr(v), ptr(out), ptr(dO), ptr(lse), ptr(m) if m is not None else None,
             ptr(bs) if bs is not None else None, ptr(bT) if bT is not None else None, ptr(delta), ptr(dq), ptr(dk), ptr(dv),
             ptr(ds) if ds is not None else None, B, H, L, Dh, st, 1.0 / Dh ** 0.5, int(causal), p, seed, offset, stream())
        dbias = None
        if ds is not None:       # d(bias) = dS summed over the batch (deterministic column sum, as the unfused path does)
            dbias = torch.empty((H, L, L), device=q.device, dtype=torch.float32)
            call("mmskin_colsum", ptr(ds), ptr(dbias), B, H * L * L, stream())
        return dq, dk, dv, None, dbias, None, None, None, None


def _dropout_state(p, n):
    if p <= 0.0:
        return 0, 0
    off = _dropout_counter[0]          # this call consumes counters [off, off + n): ranges of successive calls never overlap
    _dropout_counter[0] += n
    return torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, off


def attention_blhd(q, k, v, dropout_p=0.0, training=False, mask_add=None, bias=None, causal=False):
    """Attention on token-major views: q, k, v [B, L, H, Dh] (any strides with a contiguous last dim, e.g. slices of the
    [B, L, 3, H, Dh] output of a fused qkv Linear) -> [B, L, H, Dh] contiguous.  In bf16-operand mode without gradients this
    is ONE fused kernel reading the qkv tensor in place; otherwise the views are permuted into the [B, H, L, Dh] ops."""
    B, L, H, Dh = q.shape
    p = dropout_p if training else 0.0
    per16 = 16 // q.element_size()
    if _flash_ok(q, k, v, mask_add, bias, B * H) and all(t.data_ptr() % 16 == 0 and all(s % per16 == 0 for s in t.stride()[:3])
                                                   for t in (q, k, v)):
        seed, offset = _dropout_state(p, B * H * L * L)
        out = torch.empty((B, L, H, Dh), device=q.device, dtype=q.dtype)
        strides = []
        for t in (q, k, v, out):
            strides += [t.stride(0), t.stride(2), t.stride(1)]        # (batch, head, token)
        return _flash_forward(q, k, v, out, (B, H, L, Dh), strides, mask_add, bias, causal, p, seed, offset)
    if q.dtype == torch.bfloat16:     # bf16 views only exist on the inference lane; off the fused kernel's shapes go through fp32
        q, k, v = q.float(), k.float(), v.float()
    o = attention(q.permute(0, 2, 1, 3).contiguous(), k.permute(0, 2, 1, 3).contiguous(), v.permute(0, 2, 1, 3).contiguous(),
                  dropout_p, training, mask_add, bias, causal)
    return o.permute(0, 2, 1, 3).contiguous()


def attention(q, k, v, dropout_p=0.0, training=False, mask_add=None, bias=None, causal=False):
    B, H, L, _ = q.shape
    p = dropout_p if training else 0.0
    if _flash_train_ok(q, k, v, bias, B * H):
        seed, offset = _dropout_state(p, B * H * L * L)
        return FlashAttnFn.apply(q, k, v, mask_add, bias, causal, p, seed, offset)
    if _flash_ok(q, k, v, mask_add, bias, B * H):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        seed, offset = _dropout_state(p, B * H * L * L)
        out = torch.empty_like(q)
        strides = []
        for t in (q, k, v, out):
            strides += [t.stride(0), t.stride(1), t.stride(2)]
        return _flash_forward(q, k, v, out, tuple(q.shape), strides, mask_add, bias, causal, p, seed, offset)
    seed = offset = 0
    if p > 0.0:
        seed, offset = _dropout_state(p, B * H * L * L)
    # the one-workgroup-per-head kernel keeps L x L scores in LDS and walks the feature dimension serially: long sequences
    # and long feature dimensions (DaViT's channel attention: feature = tokens) go through the batched-GEMM path
    if mask_add is not None or bias is not None or causal or L * L * 4 > 64 * 1024 or q.shape[3] > 256:
        return LongAttentionFn.apply(q, k, v, mask_add, p, seed, offset, bias, causal)
    if p <= 0.0:
        return AttentionFn.apply(q, k, v, 0.0, 0, 0)
    return AttentionFn.apply(q, k, v, p, seed, offset)


@no_second_order
class MDNetFuseFn(torch.autograd.Function):
    """mean_hw(sigmoid(z) * f + sigmoid(tanh(f * t1) + t2)) for f [N, C, H, W]; z, t1, t2 [N, C] -> [N, C]."""

    @staticmethod
    def forward(ctx, feat, z, t1, t2):
        _need_gpu(feat, "mdnet_fuse")
        feat, z, t1, t2 = _f32c(feat), _f32c(z), _f32c(t1), _f32c(t2)
        N, C, H, W = feat.shape
        pooled = torch.empty((N, C), device=feat.device, dtype=torch.float32)
        call("mmskin_mdnet_fuse_forward", ptr(feat), ptr(z), ptr(t1), ptr(t2), ptr(pooled), N * C, H * W, stream())
        ctx.save_for_backward(feat, z, t1, t2)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        feat, z, t1, t2 = ctx.saved_tensors
        N, C, H, W = feat.shape
        dp = _f32c(dp)
        dfeat = torch.empty_like(feat) if ctx.needs_input_grad[0] else None
        dz, dt1, dt2 = torch.empty_like(z), torch.empty_like(t1), torch.empty_like(t2)
        call("mmskin_mdnet_fuse_backward", ptr(dp), ptr(feat), ptr(z), ptr(t1), ptr(t2),
             ptr(dfeat) if dfeat is not None else None, ptr(dz), ptr(dt1), ptr(dt2), N * C, H * W, stream())
        return dfeat, dz, dt1, dt2


mdnet_fuse = MDNetFuseFn.apply


@no_second_order
class AddFn(torch.autograd.Function):
    """a + b with b broadcast over a's leading dimensions (residual sums, position embeddings)."""

    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, "add")
        a, b = _f32c(a), _f32c(b)
        y = torch.empty_like(a)
        call("mmskin_add", ptr(a), ptr(b), ptr(y), a.numel(), b.numel(), stream())
        ctx.bshape = b.shape
        ctx.lead = a.numel() // b.numel()
        return y

    @staticmethod
    def backward(ctx, dy):
        db = dy if ctx.lead == 1 else dy.reshape(ctx.lead, -1).sum(0).reshape(ctx.bshape)
        return dy, db


add = AddFn.apply


@no_second_order
class ScaleAddFn(torch.autograd.Function):
    """x + gamma[c] * b   (timm LayerScale residual: x + gamma_1 * attn(norm1(x)))."""

    @staticmethod
    def forward(ctx, x, b, gamma):
        _need_gpu(x, "scale_add")
        x, b, gamma = _f32c(x), _f32c(b), _f32c(gamma)
        y = torch.empty_like(x)
        call("mmskin_scale_add_forward", ptr(x), ptr(b), ptr(gamma), ptr(y), x.numel(), gamma.numel(), stream())
        ctx.save_for_backward(b, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        b, gamma = ctx.saved_tensors
        dy = _f32c(dy)
        C = gamma.numel()
        db = torch.empty_like(dy)
        call("mmskin_scale_mul", ptr(dy), ptr(gamma), ptr(db), dy.numel(), C, 1, stream())
        prod = torch.empty_like(dy)
        call("mmskin_scale_mul", ptr(dy), ptr(b), ptr(prod), dy.numel(), C, 0, stream())
        dgamma = torch.empty_like(gamma)
        call("mmskin_colsum", ptr(prod), ptr(dgamma), dy.numel() // C, C, stream())
        return dy, db, dgamma


scale_add = ScaleAddFn.apply


@no_second_order
class TokenMeanFn(torch.autograd.Function):
    """mean over tokens [start, L) of x [B, L, E]   (timm global_pool='avg' over the patch tokens)."""

    @staticmethod
    def forward(ctx, x, start):
        _need_gpu(x, "token_mean")
        x = _f32c(x)
        B, L, E = x.shape
        out = torch.empty((B, E), device=x.device, dtype=torch.float32)
        call("mmskin_token_mean_forward", ptr(x), ptr(out), B, L, E, int(start), stream())
        ctx.dims = (B, L, E, int(start))
        return out

    @staticmethod
    def backward(ctx, dout):
        B, L, E, start = ctx.dims
        dout = _f32c(dout)
        dx = torch.empty((B, L, E), device=dout.device, dtype=torch.float32)
        call("mmskin_token_mean_backward", ptr(dout), ptr(dx), B, L, E, start, stream())
        return dx, None


token_mean = TokenMeanFn.apply


def set_linear_dtype(name):
    """'fp32' (exact-f32 MFMA, default) or 'bf16' (bf16 operands, fp32 accumulate) for the large Linear GEMMs."""
    call("mmskin_set_linear_dtype", {"fp32": _lib.F32, "float32": _lib.F32, "bf16": _lib.BF16, "bfloat16": _lib.BF16}[name.lower()])


def get_linear_dtype():
    return "bf16" if _lib.load().mmskin_get_linear_dtype() == 1 else "fp32"


@no_second_order
class PatchColsFn(torch.autograd.Function):
    """im2col for the patch-embedding GEMMs: x [N, C, H, W] (channels_last=False) or [N, H, W, C] (True), fp32 ->
    cols [N*OH*OW, C*k*k] with columns ordered (c, ky, kx) like conv.weight.flatten(1); zero padding."""

    @staticmethod
    def forward(ctx, x, k, stride, pad, channels_last):
        _need_gpu(x, "patch_cols")
        x = _f32c(x)
        if channels_last:
            N, H, W, C = x.shape
        else:
            N, C, H, W = x.shape
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        cols = torch.empty((N * OH * OW, C * k * k), device=x.device, dtype=torch.float32)
        call("mmskin_im2col_forward", ptr(x), N, C, H, W, k, stride, pad, int(channels_last), ptr(cols), stream())
        ctx.geom = (N, C, H, W, k, stride, pad, int(channels_last), tuple(x.shape))
        return cols

    @staticmethod
    def backward(ctx, dcols):
        N, C, H, W, k, stride, pad, cl, shape = ctx.geom
        dcols = _f32c(dcols)
        dx = torch.empty(shape, device=dcols.device, dtype=torch.float32)
        call("mmskin_im2col_backward", ptr(dcols), N, C, H, W, k, stride, pad, cl, ptr(dx), stream())
        return dx, None, None, None, None


def patch_cols(x, k, stride, pad=0, channels_last=False):
    return PatchColsFn.apply(x, k, stride, pad, channels_last)


@no_second_order
class DwConv3Fn(torch.autograd.Function):
    """Depthwise 3x3 (stride 1, pad 1) on NHWC fp32: x [N, H, W, C], w [C, 1, 3, 3] -> [N, H, W, C]."""

    @staticmethod
    def forward(ctx, x, w):
        _need_gpu(x, "dwconv3")
        x, w = _f32c(x), _f32c(w)
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        stage = torch.empty(9 * C, device=x.device, dtype=torch.float32)
        call("mmskin_dwconv3_forward", ptr(x), ptr(w), ptr(stage), ptr(y), N, H, W, C, stream())
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, H, W, C = x.shape
        dy = _f32c(dy)
        stage = torch.empty(9 * C, device=x.device, dtype=torch.float32)
        scratch = torch.empty(_lib.load().mmskin_dwconv3_scratch_floats(N, H, W, C), device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        call("mmskin_dwconv3_backward", ptr(dy), ptr(x), ptr(w), ptr(stage), ptr(scratch), ptr(dx) if dx is not None else None,
             ptr(dw) if dw is not None else None, N, H, W, C, stream())
        return dx, dw


dwconv3 = DwConv3Fn.apply


@no_second_order
class ConvPosEncFn(torch.autograd.Function):
    """x + dwconv3(x, w) + b on NHWC fp32 (timm davit.py ConvPosEnc.forward) in one kernel; the backward's dx = dy + dgrad(dy) in one
    kernel and dw, db from one pass over dy."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_gpu(x, "conv_pos_enc")
        x, w, b = _f32c(x), _f32c(w), _f32c(b)
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        stage = torch.empty(9 * C, device=x.device, dtype=torch.float32)
        call("mmskin_conv_pos_enc_forward", ptr(x), ptr(w), ptr(b), ptr(stage), ptr(y), N, H, W, C, stream())
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, H, W, C = x.shape
        dy = _f32c(dy)
        stage = torch.empty(9 * C, device=x.device, dtype=torch.float32)
        scratch = torch.empty(_lib.load().mmskin_dwconv3_scratch_floats(N, H, W, C), device=x.device, dtype=torch.float32)
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if need_w else None
        db = torch.empty(C, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        call("mmskin_conv_pos_enc_backward", ptr(dy), ptr(x), ptr(w), ptr(stage), ptr(scratch), ptr(dx), ptr(dw), ptr(db), N, H, W, C, stream())
        return dx, (dw if ctx.needs_input_grad[1] else None), db


conv_pos_enc = ConvPosEncFn.apply


@no_second_order
class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_gpu(x, "gelu")
        x = _f32c(x)
        y = torch.empty_like(x)
        call("mmskin_gelu_forward", ptr(x), ptr(y), x.numel(), stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        call("mmskin_gelu_backward", ptr(dy), ptr(x), ptr(dx), x.numel(), stream())
        return dx


gelu = GeluFn.apply


@no_second_order
class LinearGeluFn(torch.autograd.Function):
    """h = gelu(x @ w.T + b) with gradients (first half of a transformer MLP).  Forward: Linear, then GELU (the pre-activation z is kept
    for the backward); backward: ONE C call -- gelu'(z) is applied inside the pass that converts dh for the bf16 GEMMs and sums it for
    db (mmskin_linear_gelu_backward), so d(z) is never written in fp32.  The bf16 operand copy of x is kept as in LinearFn."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_gpu(x, "linear_gelu")
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        w = _f32c(w)
        M, K = x2.shape
        N = w.shape[0]
        z = torch.empty((M, N), device=x.device, dtype=torch.float32)
        need_w = ctx.needs_input_grad[1]
        pitch = _lib.load().mmskin_linear_x16_pitch(M, K, N) if need_w else 0
        if pitch:
            x16 = torch.empty((M, pitch), device=x.device, dtype=torch.bfloat16)
            call("mmskin_linear_forward_keep", ptr(x2), ptr(w), ptr(b), None, ptr(z), ptr(x16), M, K, N, 0, stream())
            ctx.save_for_backward(x16, w, z)
        else:
            call("mmskin_linear_forward", ptr(x2), ptr(w), ptr(b), ptr(z), M, K, N, 0, stream())
            ctx.save_for_backward(x2 if need_w else None, w, z)
        h = torch.empty_like(z)
        call("mmskin_gelu_forward", ptr(z), ptr(h), z.numel(), stream())
        ctx.kept = bool(pitch)
        ctx.mk = (M, K)
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        return h.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dh):
        xs, w, z = ctx.saved_tensors
        M, K = ctx.mk
        N = w.shape[0]
        dh2 = _f32c(dh).reshape(M, N)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty((M, K), device=dh.device, dtype=torch.float32) if need_x else None
        dw = torch.empty_like(w) if need_w else None
        db = torch.empty(N, device=dh.device, dtype=torch.float32) if need_b else None
        # off the bf16-operand large-GEMM path (head.hip: linear_big / linear_big_padded) the C side applies gelu'(z) in a pass of its own
        # (mmskin_linear_x16_pitch(M, K, N) > 0 IS that predicate, asked of the library instead of re-derived here: ADVICE r03)
        fused = _lib.load().mmskin_linear_x16_pitch(M, K, N) > 0
        scratch = None if fused else torch.empty_like(dh2)
        if ctx.kept:
            call("mmskin_linear_backward_keep", ptr(dh2), ptr(xs), ptr(w), None, ptr(z), ptr(scratch), ptr(dx), ptr(dw), ptr(db), M, K, N, stream())
        else:
            call("mmskin_linear_gelu_backward", ptr(dh2), ptr(xs), ptr(w), ptr(z), ptr(scratch), ptr(dx), ptr(dw), ptr(db), M, K, N, stream())
        return (dx.reshape(ctx.xshape) if need_x else None), dw, db


@no_second_order
class MlpFn(torch.autograd.Function):
    """[res +] fc2(gelu(fc1(x))) with gradients (timm Mlp.forward and the block's skip connection) on the bf16-operand large-GEMM path:
    the hidden activation exists only as the pre-activation z (fp32, for GELU's derivative) and as the bf16 operand of fc2 -- gelu(z)
    is never written in fp32; the residual is added in the pass that writes the output.  Backward: two C calls (fc2's, then fc1's with
    gelu'(z) inside the conversion pass), each on the bf16 operand copy kept from the forward."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res, p1, p2):
        _need_gpu(x, "mlp")
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        w1, w2 = _f32c(w1), _f32c(w2)
        M, K = x2.shape
        Hd, N = w1.shape[0], w2.shape[0]
        z = torch.empty((M, Hd), device=x.device, dtype=torch.float32)
        x16 = torch.empty((M, p1), device=x.device, dtype=torch.bfloat16)
        call("mmskin_linear_forward_keep", ptr(x2), ptr(w1), ptr(b1), None, ptr(z), ptr(x16), M, K, Hd, 0, stream())
        h16 = torch.empty((M, p2), device=x.device, dtype=torch.bfloat16)
        call("mmskin_gelu_forward_bf16", ptr(z), ptr(h16), M, Hd, p2, stream())
        y = torch.empty((M, N), device=x.device, dtype=torch.float32)
        r2 = _f32c(res).reshape(M, N) if res is not None else None
        call("mmskin_linear_forward_x16", ptr(h16), ptr(w2), ptr(b2), ptr(r2), ptr(y), M, Hd, N, 0, stream())
        ctx.save_for_backward(x16, w1, z, h16, w2)
        ctx.dims = (M, K, Hd, N)
        ctx.bias = (b1 is not None, b2 is not None)
        ctx.xshape = x.shape
        ctx.res_shape = None if res is None else res.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x16, w1, z, h16, w2 = ctx.saved_tensors
        M, K, Hd, N = ctx.dims
        dy2 = _f32c(dy).reshape(M, N)
        ng = ctx.needs_input_grad
        dev = dy.device
        dh = torch.empty((M, Hd), device=dev, dtype=torch.float32)
        dw2 = torch.empty_like(w2) if ng[3] else None
        db2 = torch.empty(N, device=dev, dtype=torch.float32) if ctx.bias[1] and ng[4] else None
        call("mmskin_linear_backward_keep", ptr(dy2), ptr(h16), ptr(w2), None, None, None, ptr(dh), ptr(dw2), ptr(db2), M, Hd, N, stream())
        dx = torch.empty((M, K), device=dev, dtype=torch.float32) if ng[0] else None
        dw1 = torch.empty_like(w1) if ng[1] else None
        db1 = torch.empty(Hd, device=dev, dtype=torch.float32) if ctx.bias[0] and ng[2] else None
        call("mmskin_linear_backward_keep", ptr(dh), ptr(x16), ptr(w1), None, ptr(z), None, ptr(dx), ptr(dw1), ptr(db1), M, K, Hd, stream())
        dres = dy2.reshape(ctx.res_shape) if ctx.res_shape is not None and ng[5] else None
        return (dx.reshape(ctx.xshape) if dx is not None else None), dw1, db1, dw2, db2, dres, None, None


def mlp(x, w1, b1, w2, b2, residual=None):
    """[residual +] fc2(gelu(fc1(x))).  With gradients, fp32 tensors on the GPU and both Linears on the bf16-operand large-GEMM path:
    MlpFn (no fp32 gelu(z), kept bf16 operands, residual added by the output pass); otherwise linear_gelu, linear and add."""
    if (_needs_grad(x, w1, b1, w2, b2) and x.is_cuda and x.dtype == torch.float32 and w1.dtype == torch.float32 and w2.dtype == torch.float32
            and w1.requires_grad and w2.requires_grad and (residual is None or residual.dtype == torch.float32)):
        M = x.numel() // x.shape[-1]
        lib = _lib.load()
        p1 = lib.mmskin_linear_x16_pitch(M, x.shape[-1], w1.shape[0])
        p2 = lib.mmskin_linear_x16_pitch(M, w1.shape[0], w2.shape[0])
        if p1 and p2:
            return MlpFn.apply(x, w1, b1, w2, b2, residual, p1, p2)
    return linear(linear_gelu(x, w1, b1), w2, b2, residual=residual)


def linear_gelu(x, w, b=None, out_dtype=None):
    """gelu(x @ w.T + b) -- the first half of a transformer MLP.  Without gradients (frozen encoders, inference) bias and the
    exact GELU run in the GEMM epilogue (one launch, the pre-activation never reaches memory); with gradients the two ops
    stay separate because GELU's backward needs the pre-activation."""
    if _needs_grad(x, w, b):
        if x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32:
            return LinearGeluFn.apply(x, w, b)
        return gelu(linear(x, w, b))
    if x.dtype == torch.bfloat16 or out_dtype == torch.bfloat16:
        return linear_lane(x, w, b, 2, out_dtype=out_dtype)
    with torch.no_grad():
        return LinearFn.apply(x, w, b, 2)


@no_second_order
class GeluTanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_gpu(x, "gelu_tanh")
        x = _f32c(x)
        y = torch.empty_like(x)
        call("mmskin_gelu_tanh_forward", ptr(x), ptr(y), x.numel(), stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        call("mmskin_gelu_tanh_backward", ptr(dy), ptr(x), ptr(dx), x.numel(), stream())
        return dx


gelu_tanh = GeluTanhFn.apply


@no_second_order
class EmbeddingFn(torch.autograd.Function):
    """table [ncols, card, E], ids [B, ncols] -> [B, ncols, E]."""

    @staticmethod
    def forward(ctx, table, ids):
        _need_gpu(table, "embedding")
        table = _f32c(table)
        ids = ids.long().contiguous()
        ncols, card, E = table.shape
        B = ids.shape[0]
        out = torch.empty((B, ncols, E), device=table.device, dtype=torch.float32)
        call("mmskin_embedding_forward", ptr(table), ptr(ids), ptr(out), B, ncols, card, E, stream())
        ctx.save_for_backward(ids)
        ctx.dims = (B, ncols, card, E)
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        B, ncols, card, E = ctx.dims
        dout = _f32c(dout)
        dtable = torch.empty((ncols, card, E), device=dout.device, dtype=torch.float32)
        call("mmskin_embedding_backward", ptr(dout), ptr(ids), ptr(dtable), B, ncols, card, E, stream())
        return dtable, None


embedding = EmbeddingFn.apply


@no_second_order
class DirectConvFn(torch.autograd.Function):
    """Small direct NCHW conv (+bias, optional ReLU) for the `custom-cnn` stem (no input gradient)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, relu):
        _need_gpu(x, "direct_conv2d")
        x, w = _f32c(x), _f32c(w)
        N, Cin, H, W = x.shape
        Cout, _, kh, kw = w.shape
        OH, OW = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        y = torch.empty((N, Cout, OH, OW), device=x.device, dtype=torch.float32)
        call("mmskin_direct_conv2d_forward", ptr(x), ptr(w), ptr(b), ptr(y), N, Cin, H, W, Cout, kh, kw, stride, pad,
             int(relu), stream())
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.cfg = (stride, pad, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        dy = _f32c(dy)
        N, Cin, H, W = x.shape
        Cout, _, kh, kw = w.shape
        dw = torch.empty_like(w)
        db = torch.empty(Cout, device=dy.device, dtype=torch.float32) if has_bias else None
        call("mmskin_direct_conv2d_backward", ptr(dy), ptr(x), ptr(y), ptr(dw), ptr(db), N, Cin, H, W, Cout, kh, kw,
             stride, pad, stream())
        return None, dw, db, None, None, None


direct_conv2d = DirectConvFn.apply


@no_second_order
class PoolGapFn(torch.autograd.Function):
    """MaxPool2d(k) followed by AdaptiveAvgPool2d(1)+Flatten: [N,C,H,W] -> [N,C]."""

    @staticmethod
    def forward(ctx, x, k):
        _need_gpu(x, "pool_gap")
        x = _f32c(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, C, H // k, W // k), device=x.device, dtype=torch.int32)
        call("mmskin_pool_gap_forward", ptr(x), ptr(y), ptr(idx), N, C, H, W, k, stream())
        ctx.save_for_backward(idx)
        ctx.cfg = (N, C, H, W, k)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, C, H, W, k = ctx.cfg
        dy = _f32c(dy)
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
        call("mmskin_pool_gap_backward", ptr(dy), ptr(idx), ptr(dx), N, C, H, W, k, stream())
        return dx, None


pool_gap = PoolGapFn.apply

"""nn.Module parameter holders whose forward runs on the HIP kernels.

They subclass the torch modules the reference uses (nn.Linear, nn.LayerNorm,
nn.MultiheadAttention, ...) only to inherit parameter names, shapes and default
initialisation (so state_dict keys and seeded init match the reference); every
forward is overridden to call the C ABI through mmskin.ops.
"""
import torch
import torch.nn as nn

import os

from . import ops

_MHA_ROWS = os.environ.get("MMSKIN_MHA_ROWS", "1") != "0"


class HipLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, fuse_relu=False):
        super().__init__(in_features, out_features, bias=bias)
        self.fuse_relu = fuse_relu

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias, self.fuse_relu)

    def extra_repr(self):
        return super().extra_repr() + (", fused_relu=True" if self.fuse_relu else "")


class HipLayerNorm(nn.LayerNorm):
    def __init__(self, normalized_shape, fuse_relu=False):
        super().__init__(normalized_shape)
        self.fuse_relu = fuse_relu

    def forward(self, x):
        return ops.layernorm(x, self.weight, self.bias, self.eps, self.fuse_relu)


class FusedAway(nn.Module):
    """Placeholder that keeps nn.Sequential indices (=> state_dict keys) identical to the reference
    where an activation was fused into the preceding HIP kernel."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def forward(self, x):
        return x

    def extra_repr(self):
        return f"{self.what} fused into previous kernel"


class HipDropout(nn.Dropout):
    def forward(self, x):
        return ops.dropout(x, self.p, self.training)


class HipMultiheadAttention(nn.MultiheadAttention):
    """nn.MultiheadAttention(embed_dim, num_heads, batch_first=False) semantics on HIP kernels.

    The reference only ever calls it with one key (multimodalIntraInterModal.py:190-197): softmax over a
    single key is exactly 1, so the output is out_proj(v_proj(value)) and the q/k rows of in_proj receive
    exact-zero gradients -- that case skips the dead q/k projections.  Longer
    sequences go through the softmax-attention kernel.
    """

    def forward(self, query, key, value, need_weights=False, **_unused):
        D, H = self.embed_dim, self.num_heads
        W, Bv = self.in_proj_weight, self.in_proj_bias
        if self.batch_first:
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        Lq, B, _ = query.shape
        Lk = key.shape[0]
        if Lk == 1:
            if _MHA_ROWS:
                v = ops.linear_rows(value.reshape(B, D), W, Bv, 2 * D, 3 * D)
            else:   # autograd's slice backward: two fills + two copies per call (A/B: MMSKIN_MHA_ROWS=0)
                v = ops.linear(value.reshape(B, D), W[2 * D:], Bv[2 * D:] if Bv is not None else None)
            o = ops.linear(v, self.out_proj.weight, self.out_proj.bias).reshape(1, B, D)
            if Lq != 1:
                o = o.expand(Lq, B, D)
            # keep q/k rows of in_proj on the graph with exact-zero gradients, as autograd does for
            # the reference (their softmax gradient is identically zero)
        else:
            q = ops.linear(query.reshape(Lq * B, D), W[:D], Bv[:D] if Bv is not None else None)
            k = ops.linear(key.reshape(Lk * B, D), W[D:2 * D], Bv[D:2 * D] if Bv is not None else None)
            v = ops.linear(value.reshape(Lk * B, D), W[2 * D:], Bv[2 * D:] if Bv is not None else None)
            split = lambda t, L: t.reshape(L, B, H, D // H).permute(1, 2, 0, 3).contiguous()
            a = ops.attention(split(q, Lq), split(k, Lk), split(v, Lk), self.dropout, self.training)   # [B,H,Lq,Dh]
            a = a.permute(2, 0, 1, 3).reshape(Lq * B, D)
            o = ops.linear(a, self.out_proj.weight, self.out_proj.bias).reshape(Lq, B, D)
        if self.batch_first:
            o = o.transpose(0, 1)
        return o, None

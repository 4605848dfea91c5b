"""Gated residual attention block on the HIP path -- drop-in for models/gatedResidualBlock.py:4-42.

LN( g * Drop(MHA8(q,k,v)) + (1-g) * q ),  g = sigmoid(W_g q + b_g).
"""
import os
import sys

import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from mmskin import ops  # noqa: E402
from mmskin.nn import HipDropout, HipLayerNorm, HipLinear, HipMultiheadAttention  # noqa: E402


class GatedAlteredResidualBlock(nn.Module):
    def __init__(self, dim, dropout=0.1):
        super().__init__()
        self.norm = HipLayerNorm(dim)
        self.attn = HipMultiheadAttention(embed_dim=dim, num_heads=8, batch_first=False)
        self.dropout = HipDropout(dropout)
        self.gate_linear = HipLinear(dim, dim)

    def forward(self, q, k, v):
        a = self.dropout(self.attn(q, k, v)[0])
        z = self.gate_linear(q)
        return self.norm(ops.gated_mix(z, a.expand_as(q) if a.shape != q.shape else a, q))


class StackedGatedResidualBlock(nn.Module):
    def __init__(self, dim, depth=4, dropout=0.1):
        super().__init__()
        self.blocks = nn.ModuleList(
            [GatedAlteredResidualBlock(dim=dim, dropout=dropout) for _ in range(depth)])

    def forward(self, q, k=None, v=None):
        k = q if k is None else k
        v = q if v is None else v
        for block in self.blocks:
            q = block(q, k, v)
        return q

"""MetaBlock on the HIP path -- drop-in for the reference's models/metablock.py:4-32.

out = sigmoid(tanh(V * LN(W_f U + b_f)) + LN(W_g U + b_g)); the two Linear+LayerNorm branches run
as MFMA GEMMs + wave-shuffle LayerNorm kernels, the gate as one fused pointwise kernel.
"""
import os
import sys

import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from mmskin import ops  # noqa: E402
from mmskin.nn import HipLayerNorm, HipLinear  # noqa: E402


class MetaBlock(nn.Module):
    def __init__(self, V_dim, U_dim):
        super().__init__()
        self.fb = nn.Sequential(HipLinear(U_dim, V_dim), HipLayerNorm(V_dim))
        self.gb = nn.Sequential(HipLinear(U_dim, V_dim), HipLayerNorm(V_dim))

    def forward(self, V, U):
        return ops.metablock_gate(V, self.fb(U), self.gb(U))

"""TabTransformer metadata encoder on the HIP path -- drop-in for models/tab_transformer.py:6-60.

Per-column embedding gather -> 2 post-norm encoder layers (d=32, 4 heads, ff=128, ReLU) -> flatten,
concatenate the projected continuous columns -> MLP.  Parameters are held by the same torch modules
the reference builds (nn.Embedding list, nn.TransformerEncoder, nn.Linear) so state_dict keys and
seeded initialisation agree; the arithmetic runs on the HIP kernels.
"""
import os
import sys

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from mmskin import ops  # noqa: E402
from mmskin.nn import FusedAway, HipDropout, HipLinear  # noqa: E402


class TabTransformer(nn.Module):
    def __init__(self, categorical_cardinalities, num_continuous, embed_dim=32, num_heads=4,
                 num_transformer_layers=2, hidden_dim=128, output_dim=1, dropout=0.3):
        super().__init__()
        self.embeddings = nn.ModuleList(
            [nn.Embedding(card, embed_dim) for card in categorical_cardinalities])
        self.num_categorical = len(categorical_cardinalities)
        self.cardinalities = list(categorical_cardinalities)
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.drop_p = dropout
        layer = nn.TransformerEncoderLayer(d_model=embed_dim, nhead=num_heads, dim_feedforward=hidden_dim,
                                           activation="relu", dropout=dropout, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_transformer_layers)
        self.numeric_projection = HipLinear(num_continuous, embed_dim) if num_continuous > 0 else None
        width = self.num_categorical * embed_dim + (embed_dim if num_continuous > 0 else 0)
        self.fc = nn.Sequential(HipLinear(width, hidden_dim, fuse_relu=True), FusedAway("ReLU"),
                                HipDropout(dropout), HipLinear(hidden_dim, output_dim))

    def _encoder_layer(self, layer, x):
        B, L, E = x.shape
        H = self.num_heads
        sa = layer.self_attn
        qkv = ops.linear(x.reshape(B * L, E), sa.in_proj_weight, sa.in_proj_bias)        # [B*L, 3E]
        qkv = qkv.reshape(B, L, 3, H, E // H).permute(2, 0, 3, 1, 4).contiguous()       # [3,B,H,L,Dh]
        a = ops.attention(qkv[0], qkv[1], qkv[2], sa.dropout, self.training).permute(0, 2, 1, 3).reshape(B * L, E)
        a = ops.linear(a, sa.out_proj.weight, sa.out_proj.bias)
        a = ops.dropout(a, self.drop_p, self.training)
        x2 = ops.layernorm((x.reshape(B * L, E) + a), layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
        f = ops.linear(x2, layer.linear1.weight, layer.linear1.bias, relu=True)
        f = ops.dropout(f, self.drop_p, self.training)
        f = ops.linear(f, layer.linear2.weight, layer.linear2.bias)
        f = ops.dropout(f, self.drop_p, self.training)
        x3 = ops.layernorm(x2 + f, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
        return x3.reshape(B, L, E)

    def forward(self, x_categorical, x_numerical):
        if len(set(self.cardinalities)) == 1:
            table = torch.stack([e.weight for e in self.embeddings], dim=0)              # [ncols, card, E]
            tokens = ops.embedding(table, x_categorical)
        else:
            tokens = torch.stack([ops.embedding(e.weight.unsqueeze(0), x_categorical[:, i:i + 1]).squeeze(1)
                                  for i, e in enumerate(self.embeddings)], dim=1)
        for layer in self.transformer_encoder.layers:
            tokens = self._encoder_layer(layer, tokens)
        feats = tokens.flatten(start_dim=1)
        if self.numeric_projection is not None:
            feats = ops.concat2(feats, self.numeric_projection(x_numerical))
        return self.fc(feats)

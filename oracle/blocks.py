"""Oracle fusion blocks (torch CPU, fp32).  TEST INFRASTRUCTURE ONLY.

Restates gatedResidualBlock.py:4-17, metablock.py:4-32 and
tab_transformer.py:6-60 of the reference.  Pinned against the reference import
by tests/golden/blocks.json (see oracle/gen_golden.py).
"""
import torch
import torch.nn as nn


class OracleGatedResidual(nn.Module):
    """LN( g * Drop(MHA8(q,k,v)) + (1-g) * q ),  g = sigmoid(W_g q + b)."""

    def __init__(self, dim, dropout=0.1):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.attn = nn.MultiheadAttention(embed_dim=dim, num_heads=8, batch_first=False)
        self.dropout = nn.Dropout(dropout)
        self.gate_linear = nn.Linear(dim, dim)

    def forward(self, q, k, v):
        a = self.dropout(self.attn(q, k, v)[0])
        g = torch.sigmoid(self.gate_linear(q))
        return self.norm(g * a + (1.0 - g) * q)


class OracleMetaBlock(nn.Module):
    """sigmoid( tanh(V * LN(W_f U)) + LN(W_g U) )."""

    def __init__(self, V_dim, U_dim):
        super().__init__()
        self.fb = nn.Sequential(nn.Linear(U_dim, V_dim), nn.LayerNorm(V_dim))
        self.gb = nn.Sequential(nn.Linear(U_dim, V_dim), nn.LayerNorm(V_dim))

    def forward(self, V, U):
        return torch.sigmoid(torch.tanh(V * self.fb(U)) + self.gb(U))


class OracleTabTransformer(nn.Module):
    def __init__(self, categorical_cardinalities, num_continuous, embed_dim=32, num_heads=4,
                 num_transformer_layers=2, hidden_dim=128, output_dim=1, dropout=0.3):
        super().__init__()
        self.embeddings = nn.ModuleList(
            [nn.Embedding(card, embed_dim) for card in categorical_cardinalities])
        self.num_categorical = len(categorical_cardinalities)
        self.embed_dim = embed_dim
        layer = nn.TransformerEncoderLayer(d_model=embed_dim, nhead=num_heads,
                                           dim_feedforward=hidden_dim, activation="relu",
                                           dropout=dropout, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_transformer_layers)
        self.numeric_projection = nn.Linear(num_continuous, embed_dim) if num_continuous > 0 else None
        width = self.num_categorical * embed_dim + (embed_dim if num_continuous > 0 else 0)
        self.fc = nn.Sequential(nn.Linear(width, hidden_dim), nn.ReLU(), nn.Dropout(dropout),
                                nn.Linear(hidden_dim, output_dim))

    def forward(self, x_categorical, x_numerical):
        tokens = torch.stack(
            [emb(x_categorical[:, i]) for i, emb in enumerate(self.embeddings)], dim=1)
        feats = self.transformer_encoder(tokens).flatten(start_dim=1)
        if self.numeric_projection is not None:
            feats = torch.cat([feats, self.numeric_projection(x_numerical)], dim=1)
        return self.fc(feats)

"""GPT-2 decoder stack on the HIP ops -- what `AutoModel.from_pretrained("gpt2")` gives the reference
(loadImageModelClassifier.py:170-181); the model then reads `last_hidden_state[:, 0, :]`.

transformers' `GPT2Model` module tree / keys: wte, wpe, h.N.{ln_1, attn.{c_attn, c_proj}, ln_2, mlp.{c_fc, c_proj}}, ln_f.
c_attn / c_proj / c_fc are HuggingFace `Conv1D` layers: weight [in, out] (transposed w.r.t. nn.Linear).  Causal
self-attention with the padding mask added, "gelu_new" (tanh) activation, pre-LayerNorm residual blocks.
"""
import os
import sys
from types import SimpleNamespace

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from mmskin import ops  # noqa: E402
from mmskin.nn import HipDropout, HipLayerNorm  # noqa: E402

GPT2_SMALL = dict(vocab_size=50257, n_positions=1024, n_embd=768, n_layer=12, n_head=12, layer_norm_epsilon=1e-5,
                  resid_pdrop=0.1, embd_pdrop=0.1, attn_pdrop=0.1, initializer_range=0.02)


class Conv1D(nn.Module):
    """HuggingFace Conv1D: y = x @ weight + bias with weight [in, out]."""

    def __init__(self, nf, nx):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.empty(nx, nf))
        self.bias = nn.Parameter(torch.zeros(nf))
        nn.init.normal_(self.weight, std=0.02)

    def forward(self, x):
        return ops.linear(x, self.weight.t().contiguous(), self.bias)


def _ln(dim, eps):
    m = HipLayerNorm(dim)
    m.eps = eps
    return m


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.num_heads = c.n_head
        self.c_attn = Conv1D(3 * c.n_embd, c.n_embd)
        self.c_proj = Conv1D(c.n_embd, c.n_embd)
        self.attn_dropout = HipDropout(c.attn_pdrop)
        self.resid_dropout = HipDropout(c.resid_pdrop)

    def forward(self, x, B, L, mask_add):
        H, E = self.num_heads, x.shape[1]
        qkv = self.c_attn(x).reshape(B, L, 3, H, E // H)
        o = ops.attention_blhd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], self.attn_dropout.p, self.training, mask_add=mask_add,
                               causal=True)
        return self.resid_dropout(self.c_proj(o.reshape(B * L, E)))


class _MLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c_fc = Conv1D(4 * c.n_embd, c.n_embd)
        self.c_proj = Conv1D(c.n_embd, 4 * c.n_embd)
        self.dropout = HipDropout(c.resid_pdrop)

    def forward(self, x):
        return self.dropout(self.c_proj(ops.gelu_tanh(self.c_fc(x))))


class _Block(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.ln_1 = _ln(c.n_embd, c.layer_norm_epsilon)
        self.attn = _Attention(c)
        self.ln_2 = _ln(c.n_embd, c.layer_norm_epsilon)
        self.mlp = _MLP(c)

    def forward(self, x, B, L, mask_add):
        x = ops.add(x, self.attn(ops.layernorm(x, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps), B, L, mask_add))
        return ops.add(x, self.mlp(ops.layernorm(x, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)))


class HipGPT2Model(nn.Module):
    def __init__(self, config=None, **overrides):
        super().__init__()
        cfg = dict(GPT2_SMALL)
        if isinstance(config, dict):
            cfg.update(config)
        cfg.update(overrides)
        cfg["hidden_size"] = cfg["n_embd"]
        self.config = SimpleNamespace(**cfg)
        c = self.config
        self.wte = nn.Embedding(c.vocab_size, c.n_embd)
        self.wpe = nn.Embedding(c.n_positions, c.n_embd)
        self.drop = HipDropout(c.embd_pdrop)
        self.h = nn.ModuleList([_Block(c) for _ in range(c.n_layer)])
        self.ln_f = _ln(c.n_embd, c.layer_norm_epsilon)
        for m in self.modules():
            if isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=c.initializer_range)

    def forward(self, input_ids=None, attention_mask=None, **_unused):
        B, L = input_ids.shape
        E = self.config.n_embd
        tok = ops.embedding(self.wte.weight.unsqueeze(0), input_ids.reshape(B * L, 1)).reshape(B, L, E)
        x = self.drop(ops.add(tok, self.wpe.weight[:L]).reshape(B * L, E))
        mask_add = None
        if attention_mask is not None:
            mask_add = (1.0 - attention_mask.to(torch.float32)) * torch.finfo(torch.float32).min
        for blk in self.h:
            x = blk(x, B, L, mask_add)
        x = ops.layernorm(x, self.ln_f.weight, self.ln_f.bias, self.ln_f.eps)
        return SimpleNamespace(last_hidden_state=x.reshape(B, L, E))

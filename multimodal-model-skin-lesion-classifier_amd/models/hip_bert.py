"""BERT encoder on the HIP ops -- what `AutoModel.from_pretrained("bert-base-uncased")` gives the reference
(loadImageModelClassifier.py:170-181), which then reads `outputs.last_hidden_state[:, 0, :]`
(multimodalIntraInterModal.py:180-183).

Same module tree as transformers' `BertModel` (embeddings.{word,position,token_type}_embeddings, embeddings.LayerNorm,
encoder.layer.N.attention.self.{query,key,value}, attention.output.{dense,LayerNorm}, intermediate.dense,
output.{dense,LayerNorm}, pooler.dense), so a HuggingFace state_dict loads with strict=True.  Weights are randomly
initialised (normal(0, 0.02), as HF); this package never downloads.  Arithmetic: Linear layers on the exact-f32
implicit-GEMM kernels (batch x tokens rows), attention through strided batched GEMMs + a row-softmax kernel, LayerNorm /
GELU / dropout / embedding kernels from head.hip.
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
from mmskin.nn import HipDropout, HipLayerNorm, HipLinear  # noqa: E402

BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                 max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, initializer_range=0.02)


def _ln(dim, eps):
    m = HipLayerNorm(dim)
    m.eps = eps
    return m


class _Embeddings(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.word_embeddings = nn.Embedding(c.vocab_size, c.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type_embeddings = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.LayerNorm = _ln(c.hidden_size, c.layer_norm_eps)
        self.dropout = HipDropout(c.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids):
        B, L = input_ids.shape
        E = self.word_embeddings.embedding_dim
        w = ops.embedding(self.word_embeddings.weight.unsqueeze(0), input_ids.reshape(B * L, 1)).reshape(B, L, E)
        t = ops.embedding(self.token_type_embeddings.weight.unsqueeze(0), token_type_ids.reshape(B * L, 1)).reshape(B, L, E)
        x = ops.add(ops.add(w, t), self.position_embeddings.weight[:L])           # positions 0..L-1 broadcast over the batch
        x = ops.layernorm(x.reshape(B * L, E), self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps)
        return self.dropout(x)                                                     # [B*L, E]


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.num_attention_heads = c.num_attention_heads
        self.query = HipLinear(c.hidden_size, c.hidden_size)
        self.key = HipLinear(c.hidden_size, c.hidden_size)
        self.value = HipLinear(c.hidden_size, c.hidden_size)
        self.dropout = HipDropout(c.attention_probs_dropout_prob)                  # applied inside ops.attention

    def forward(self, x, B, L, mask_add, od=None):
        H = self.num_attention_heads
        E = x.shape[1]
        if od is not None:   # lane: the three projections as ONE GEMM over the stacked (cached bf16) weights
            qkv = ops.linear_lane(x, (self.query.weight, self.key.weight, self.value.weight),
                                  torch.cat([self.query.bias, self.key.bias, self.value.bias]), out_dtype=od).reshape(B, L, 3, H, E // H)
            o = ops.attention_blhd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], self.dropout.p, self.training, mask_add=mask_add)
            return o.reshape(B * L, E)
        proj = lambda lin: ops.linear(x, lin.weight, lin.bias, out_dtype=od).reshape(B, L, H, E // H)   # token-major: no permute copies
        o = ops.attention_blhd(proj(self.query), proj(self.key), proj(self.value), self.dropout.p, self.training, mask_add=mask_add)
        return o.reshape(B * L, E)


class _SelfOutput(nn.Module):
    def __init__(self, c, in_dim):
        super().__init__()
        self.dense = HipLinear(in_dim, c.hidden_size)
        self.LayerNorm = _ln(c.hidden_size, c.layer_norm_eps)
        self.dropout = HipDropout(c.hidden_dropout_prob)

    def forward(self, h, residual, od=None):
        """-> (fp32, bf16-or-same): the fp32 result is the residual stream, the second what the next Linear reads on the lane"""
        if od is not None:   # lane: bias, hidden dropout and the residual add run in the GEMM epilogue
            h = ops.linear_lane(h, self.dense.weight, self.dense.bias, residual=residual, drop_p=self.dropout.p, training=self.training)
        else:
            h = ops.add(self.dropout(self.dense(h)), residual)
        return ops.layernorm(h, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps, out_dtype=od, keep_f32=True)


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _SelfOutput(c, c.hidden_size)


class _Intermediate(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = HipLinear(c.hidden_size, c.intermediate_size)


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.attention = _Attention(c)
        self.intermediate = _Intermediate(c)
        self.output = _SelfOutput(c, c.intermediate_size)

    def forward(self, x, B, L, mask_add, x_lane=None):
        """x: fp32 hidden states; x_lane: their bf16 copy when the previous layer ran on the inference lane.  -> (fp32, lane copy)"""
        od = ops.lane_dtype(x, self)
        xin = x_lane if (od is not None and x_lane is not None and x_lane.dtype == od) else x
        a32, a_lane = self.attention.output(self.attention.self(xin, B, L, mask_add, od), x, od)
        m = ops.linear_gelu(a_lane, self.intermediate.dense.weight, self.intermediate.dense.bias, out_dtype=od)
        return self.output(m, a32, od)


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])


class _Pooler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = HipLinear(c.hidden_size, c.hidden_size)


class HipBertModel(nn.Module):
    def __init__(self, config=None, **overrides):
        super().__init__()
        cfg = dict(BERT_BASE)
        if config is not None:
            cfg.update({k: getattr(config, k) for k in BERT_BASE if hasattr(config, k)} if not isinstance(config, dict) else config)
        cfg.update(overrides)
        self.config = SimpleNamespace(**cfg)
        c = self.config
        self.embeddings = _Embeddings(c)
        self.encoder = _Encoder(c)
        self.pooler = _Pooler(c)
        for m in self.modules():                                                    # HF BertPreTrainedModel._init_weights
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=c.initializer_range)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, mean=0.0, std=c.initializer_range)
                if m.padding_idx is not None:
                    with torch.no_grad():
                        m.weight[m.padding_idx].zero_()
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, **_unused):
        B, L = input_ids.shape
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        mask_add = None
        if attention_mask is not None:                                              # HF: (1 - mask) * finfo.min added to the scores
            mask_add = (1.0 - attention_mask.to(torch.float32)) * torch.finfo(torch.float32).min
        x = self.embeddings(input_ids, token_type_ids)
        x_lane = None
        for layer in self.encoder.layer:
            x, x_lane = layer(x, B, L, mask_add, x_lane)
        hidden = x.reshape(B, L, -1)
        pooled = torch.tanh(self.pooler.dense(hidden[:, 0].contiguous()))           # unused by the reference
        return SimpleNamespace(last_hidden_state=hidden, pooler_output=pooled)

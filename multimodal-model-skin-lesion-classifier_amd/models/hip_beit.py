"""BEiT / BEiT-v2 image encoder on the HIP ops -- timm `beitv2_large_patch16_224` / `beitv2_base_patch16_224` as the
reference's generic timm branch builds them (loadImageModelClassifier.py:117-152: `timm.create_model(name)`,
`reset_classifier(0)`, F = `num_features`), BASELINE.json configs[4].

timm's `Beit` module tree / state_dict keys: patch_embed.proj, cls_token, blocks.N.{gamma_1, gamma_2, norm1,
attn.{q_bias, v_bias, relative_position_bias_table, qkv.weight, proj}, norm2, mlp.{fc1, fc2}}, fc_norm; no absolute
position embedding, a per-block relative-position bias (27x27+3 table gathered into [heads, 197, 197]), LayerScale
(init 1e-5), mean pooling over the patch tokens followed by fc_norm.  PARITY UNPINNED against timm (absent).
"""
import os
import sys

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_PKG, os.path.dirname(os.path.abspath(__file__))):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from mmskin import ops  # noqa: E402
from mmskin.nn import HipLayerNorm, HipLinear  # noqa: E402
from hip_vit import _PatchEmbed  # noqa: E402

BEIT_CONFIGS = {   # name: (embed_dim, depth, heads)
    "beitv2_base_patch16_224": (768, 12, 12),
    "beitv2_large_patch16_224": (1024, 24, 16),
    "beitv2_tiny_test": (64, 2, 4),       # not a timm model: small shape for tests
}


def relative_position_index(ws=14):
    """timm.models.beit.gen_relative_position_index((ws, ws)) -> [ws*ws+1, ws*ws+1] long, entries in [0, (2ws-1)^2+3)."""
    num = (2 * ws - 1) * (2 * ws - 1) + 3
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)   # [2, ws*ws]
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    idx = torch.zeros((ws * ws + 1,) * 2, dtype=torch.long)
    idx[1:, 1:] = rel.sum(-1)
    idx[0, 0:] = num - 3
    idx[0:, 0] = num - 2
    idx[0, 0] = num - 1
    return idx


def _ln(dim):
    m = HipLayerNorm(dim)
    m.eps = 1e-6
    return m


class _Attention(nn.Module):
    def __init__(self, dim, heads, ws):
        super().__init__()
        self.num_heads = heads
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2 + 3, heads))
        self.qkv = HipLinear(dim, dim * 3, bias=False)
        self.proj = HipLinear(dim, dim)
        self.register_buffer("k_bias", torch.zeros(dim), persistent=False)
        self.register_buffer("relative_position_index", relative_position_index(ws), persistent=False)

    def forward(self, x, B, L, od=None):
        H = self.num_heads
        E = x.shape[1]
        bias = torch.cat([self.q_bias, self.k_bias, self.v_bias])
        qkv = ops.linear(x, self.qkv.weight, bias, out_dtype=od).reshape(B, L, 3, H, E // H)
        rel = self.relative_position_bias_table[self.relative_position_index.reshape(-1)].reshape(L, L, H).permute(2, 0, 1).contiguous()
        o = ops.attention_blhd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], bias=rel)      # token-major views of the fused qkv output
        return o.reshape(B * L, E)                                                      # the caller applies self.proj


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = HipLinear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = HipLinear(hidden, dim)

    def hidden(self, x, od=None):
        return ops.linear_gelu(x, self.fc1.weight, self.fc1.bias, out_dtype=od)

    def forward(self, x, od=None):
        return self.fc2(self.hidden(x, od))


class _Block(nn.Module):
    def __init__(self, dim, heads, ws, init_values):
        super().__init__()
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))
        self.norm1 = _ln(dim)
        self.attn = _Attention(dim, heads, ws)
        self.norm2 = _ln(dim)
        self.mlp = _Mlp(dim, dim * 4)

    def forward(self, x, B, L):
        od = ops.lane_dtype(x, self)     # bf16 activations between the ops of a block through which no gradient flows
        a = self.attn(ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, out_dtype=od), B, L, od)
        if od is not None:
            # lane: bias, layer scale and the residual add of `x + gamma * proj(a)` / `x + gamma * fc2(h)` run in the GEMM epilogue
            x = ops.linear_lane(a, self.attn.proj.weight, self.attn.proj.bias, gamma=self.gamma_1, residual=x)
            h = self.mlp.hidden(ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, out_dtype=od), od)
            return ops.linear_lane(h, self.mlp.fc2.weight, self.mlp.fc2.bias, gamma=self.gamma_2, residual=x)
        x = ops.scale_add(x, self.attn.proj(a), self.gamma_1)
        m = self.mlp(ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, out_dtype=od), od)
        return ops.scale_add(x, m, self.gamma_2)


class HipBeit(nn.Module):
    def __init__(self, name="beitv2_large_patch16_224", img_size=224, init_values=1e-5):
        super().__init__()
        if name not in BEIT_CONFIGS:
            raise NotImplementedError(f"image encoder '{name}' has no MI355X kernels (available: {sorted(BEIT_CONFIGS)})")
        dim, depth, heads = BEIT_CONFIGS[name]
        self.num_features = self.embed_dim = dim
        ws = img_size // 16
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.patch_embed = _PatchEmbed(dim, 16)
        self.blocks = nn.ModuleList([_Block(dim, heads, ws, init_values) for _ in range(depth)])
        self.fc_norm = _ln(dim)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, _Attention):
                nn.init.trunc_normal_(m.relative_position_bias_table, std=0.02)

    def forward_features(self, x):
        B = x.shape[0]
        tok = self.patch_embed(x.float())
        tok = torch.cat([self.cls_token.expand(B, -1, -1), tok], dim=1).contiguous()
        L, E = tok.shape[1], tok.shape[2]
        h = tok.reshape(B * L, E)
        for blk in self.blocks:
            h = blk(h, B, L)
        return h.reshape(B, L, E)                  # timm: self.norm is Identity when mean pooling with fc_norm is used

    def forward(self, x):
        h = self.forward_features(x)
        pooled = ops.token_mean(h, 1)              # global_pool = "avg": mean over the patch tokens
        return ops.layernorm(pooled, self.fc_norm.weight, self.fc_norm.bias, self.fc_norm.eps)

"""Vision Transformer on the HIP ops -- what `timm.create_model("vit_large_patch16_224", num_classes=0)` gives the
reference's LiwTERM model (liwtermModel.py:26-32), which calls `forward_features(image)` and keeps the CLS token.

timm's VisionTransformer module tree / state_dict keys (cls_token, pos_embed, patch_embed.proj, blocks.N.{norm1,
attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2}, norm): pre-norm blocks, qkv bias, LayerNorm eps 1e-6, exact GELU, no
LayerScale.  The 16x16/16 patch convolution is a Linear over unfolded patches; Linear layers over batch x tokens rows
run on the exact-f32 implicit-GEMM kernels, attention (197 tokens) on the batched-GEMM path.  Random init (timm's
trunc_normal(0.02)); this package never downloads weights.
"""
import os
import sys

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from mmskin import ops  # noqa: E402
from mmskin.nn import HipLayerNorm, HipLinear  # noqa: E402

VIT_CONFIGS = {   # name: (embed_dim, depth, heads)
    "vit_tiny_patch16_224": (192, 12, 3),
    "vit_small_patch16_224": (384, 12, 6),
    "vit_base_patch16_224": (768, 12, 12),
    "vit_large_patch16_224": (1024, 24, 16),
}


def _ln(dim):
    m = HipLayerNorm(dim)
    m.eps = 1e-6
    return m


class _PatchEmbed(nn.Module):
    def __init__(self, dim, patch):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)
        self.patch = patch

    def forward(self, x):
        B, C, H, W = x.shape
        P = self.patch
        gh, gw = H // P, W // P
        patches = ops.patch_cols(x, P, P, 0)          # HIP im2col: rows (b, gy, gx), columns (c, ky, kx); the ragged edge is never read
        y = ops.linear(patches, self.proj.weight.flatten(1), self.proj.bias)
        return y.reshape(B, gh * gw, -1)


class _Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = HipLinear(dim, dim * 3)
        self.proj = HipLinear(dim, dim)

    def forward(self, x, B, L, od=None):                # od = torch.bfloat16 on the inference lane (ops.lane_dtype), else None
        H = self.num_heads
        E = x.shape[1]
        qkv = ops.linear(x, self.qkv.weight, self.qkv.bias, out_dtype=od).reshape(B, L, 3, H, E // H)   # token-major: read in place
        o = ops.attention_blhd(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2])
        return self.proj(o.reshape(B * L, E))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = HipLinear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = HipLinear(hidden, dim)

    def forward(self, x, od=None):
        return self.fc2(ops.linear_gelu(x, self.fc1.weight, self.fc1.bias, out_dtype=od))


class _Block(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = _ln(dim)
        self.attn = _Attention(dim, heads)
        self.norm2 = _ln(dim)
        self.mlp = _Mlp(dim, dim * 4)

    def forward(self, x, B, L):
        od = ops.lane_dtype(x, self)     # bf16 activations between the ops of a block through which no gradient flows
        x = ops.add(x, self.attn(ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, out_dtype=od), B, L, od))
        return ops.add(x, self.mlp(ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, out_dtype=od), od))


class HipVisionTransformer(nn.Module):
    def __init__(self, name="vit_large_patch16_224", img_size=224):
        super().__init__()
        if name not in VIT_CONFIGS:
            raise NotImplementedError(f"image encoder '{name}' has no MI355X kernels (available: {sorted(VIT_CONFIGS)})")
        dim, depth, heads = VIT_CONFIGS[name]
        self.num_features = self.embed_dim = dim
        n_patches = (img_size // 16) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n_patches + 1, dim) * 0.02)
        self.patch_embed = _PatchEmbed(dim, 16)
        self.blocks = nn.Sequential(*[_Block(dim, heads) for _ in range(depth)])
        self.norm = _ln(dim)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward_features(self, x):
        B = x.shape[0]
        tok = self.patch_embed(x.float())
        tok = torch.cat([self.cls_token.expand(B, -1, -1), tok], dim=1).contiguous()
        L, E = tok.shape[1], tok.shape[2]
        tok = ops.add(tok, self.pos_embed[0, :L])
        h = tok.reshape(B * L, E)
        for blk in self.blocks:
            h = blk(h, B, L)
        h = ops.layernorm(h, self.norm.weight, self.norm.bias, self.norm.eps)
        return h.reshape(B, L, E)

    def forward(self, x):      # num_classes = 0, global_pool = "token": the CLS feature
        return self.forward_features(x)[:, 0]

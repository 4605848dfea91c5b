"""DaViT image encoder on the HIP ops -- timm `davit_tiny.msft_in1k` as the reference's generic timm branch builds it
(loadImageModelClassifier.py:117-152), BASELINE.json configs[3] (davit_tiny + tab-transformer + gfcam).

timm's `DaVit` module tree / state_dict keys: stem.{conv,norm}, stages.S.downsample.{norm,conv}, stages.S.blocks.D.0.*
(spatial window-attention block) and stages.S.blocks.D.1.* (channel-attention block), each with cpe1.proj / norm1 /
attn.{qkv,proj} / cpe2.proj / norm2 / mlp.{fc1,fc2}, head.norm.  Activations stay in token (NHWC) layout throughout:
the depthwise 3x3 position-encoding convolutions run on the fp32 NHWC depthwise kernels, the 2x2/2 downsample
convolutions are Linear layers over 2x2 patches, window partitioning is a reshape.  Channel attention follows timm
1.0.x (`dynamic_scale=True`: q scaled by N^-0.5, softmax(q^T k) applied to v^T).  PARITY UNPINNED against timm (absent).
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

DAVIT_CONFIGS = {   # name: (depths, dims, heads)
    "davit_tiny": ((1, 1, 3, 1), (96, 192, 384, 768), (3, 6, 12, 24)),
    "davit_small": ((1, 1, 9, 1), (96, 192, 384, 768), (3, 6, 12, 24)),
    "davit_base": ((1, 1, 9, 1), (128, 256, 512, 1024), (4, 8, 16, 32)),
}


def _ln(dim):
    return HipLayerNorm(dim)          # eps 1e-5 (timm DaViT norm_eps)


def _layernorm(mod, x2d):
    return ops.layernorm(x2d, mod.weight, mod.bias, mod.eps)


class _ConvPosEnc(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)

    def forward(self, x):             # x [B, H, W, C] -> x + dwconv(x) + bias
        if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0:
            return ops.conv_pos_enc(x, self.proj.weight, self.proj.bias)      # one kernel; bias gradient from the weight-gradient pass
        return ops.add(ops.add(x, ops.dwconv3(x, self.proj.weight)), self.proj.bias)


class _Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = HipLinear(dim, dim * 4)
        self.act = nn.GELU()
        self.fc2 = HipLinear(dim * 4, dim)

    def forward(self, x, residual=None):      # [residual +] fc2(gelu(fc1(x)))
        return ops.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, residual=residual)


class _WindowAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = HipLinear(dim, dim * 3)
        self.proj = HipLinear(dim, dim)

    def forward(self, xw):            # [nW*B, 49, C]
        Bw, L, C = xw.shape
        H = self.num_heads
        qkv = self.qkv(xw.reshape(Bw * L, C)).reshape(Bw, L, 3, H, C // H)
        o = ops.attention_packed(qkv)                 # reads the packed qkv tensor in place, writes token-major
        return self.proj(o.reshape(Bw * L, C)).reshape(Bw, L, C)


class _ChannelAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.groups = heads
        self.qkv = HipLinear(dim, dim * 3)
        self.proj = HipLinear(dim, dim)

    def forward(self, x, B, N, residual=None):       # x [B*N, C] -> [residual +] proj(attention)
        C = x.shape[1]
        G, Dh = self.groups, x.shape[1] // self.groups
        qkv = self.qkv(x).reshape(B, N, 3, G, Dh)
        if ops.channel_attention_ok(qkv):     # token-major in and out: no transposing copies around the attention
            return ops.linear(ops.channel_attention(qkv, N ** -0.5).reshape(B * N, C), self.proj.weight, self.proj.bias, residual=residual)
        qkv = qkv.permute(2, 0, 3, 4, 1).contiguous()      # [3, B, G, Dh, N]: channels attend over tokens
        # softmax((q * N^-0.5)^T k) over the key channels, applied to v^T: an attention with "sequence" = the Dh channels of a
        # group and "feature" = the N tokens; ops.attention scales by feature^-0.5 = N^-0.5, exactly timm's dynamic_scale
        o = ops.attention(qkv[0], qkv[1], qkv[2])                                          # [B, G, Dh, N]
        return ops.linear(o.permute(0, 3, 1, 2).reshape(B * N, C), self.proj.weight, self.proj.bias, residual=residual)


class _SpatialBlock(nn.Module):
    def __init__(self, dim, heads, ws=7):
        super().__init__()
        self.ws = ws
        self.cpe1 = _ConvPosEnc(dim)
        self.norm1 = _ln(dim)
        self.attn = _WindowAttention(dim, heads)
        self.cpe2 = _ConvPosEnc(dim)
        self.norm2 = _ln(dim)
        self.mlp = _Mlp(dim)

    def forward(self, x):             # [B, H, W, C]
        B, H, W, C = x.shape
        ws = self.ws
        shortcut = self.cpe1(x)
        h = _layernorm(self.norm1, shortcut.reshape(B * H * W, C)).reshape(B, H, W, C)
        ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
        if ph or pw:
            h = torch.nn.functional.pad(h, (0, 0, 0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        heads = self.attn.num_heads
        # qkv and proj are per-token Linears: they run on the image-major token grid, and the attention kernel finds each window's
        # tokens where they sit (ops.window_attention) -- window_partition / window_reverse of timm's SpatialBlock without copies
        qkv = self.attn.qkv(h.reshape(B * Hp * Wp, C)).reshape(B, Hp, Wp, 3, heads, C // heads)
        if ops.window_attention_ok(qkv, ws):
            o = ops.window_attention(qkv, ws).reshape(B * Hp * Wp, C)
            if not (ph or pw):    # the skip connection rides on the projection's output pass
                return self._tail(ops.linear(o, self.attn.proj.weight, self.attn.proj.bias, residual=shortcut.reshape(B * H * W, C)).reshape(B, H, W, C))
            a = self.attn.proj(o).reshape(B, Hp, Wp, C)
        else:
            win = h.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C).contiguous()
            a = self.attn(win).reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        if ph or pw:
            a = a[:, :H, :W, :].contiguous()
        return self._tail(ops.add(shortcut, a))

    def _tail(self, x):               # x = shortcut + attention
        B, H, W, C = x.shape
        x = self.cpe2(x)
        return self.mlp(_layernorm(self.norm2, x.reshape(B * H * W, C)), residual=x.reshape(B * H * W, C)).reshape(B, H, W, C)


class _ChannelBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.cpe1 = _ConvPosEnc(dim)
        self.norm1 = _ln(dim)
        self.attn = _ChannelAttention(dim, heads)
        self.cpe2 = _ConvPosEnc(dim)
        self.norm2 = _ln(dim)
        self.mlp = _Mlp(dim)

    def forward(self, x):
        B, H, W, C = x.shape
        x = self.cpe1(x)
        x = self.cpe2(self.attn(_layernorm(self.norm1, x.reshape(B * H * W, C)), B, H * W, residual=x.reshape(B * H * W, C)).reshape(B, H, W, C))
        return self.mlp(_layernorm(self.norm2, x.reshape(B * H * W, C)), residual=x.reshape(B * H * W, C)).reshape(B, H, W, C)


class _Stem(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv2d(3, dim, 7, 4, 3)
        self.norm = _ln(dim)          # timm LayerNorm2d: LayerNorm over channels


class _Downsample(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = _ln(cin)
        self.conv = nn.Conv2d(cin, cout, 2, 2)


class _Stage(nn.Module):
    def __init__(self, cin, cout, depth, heads, downsample):
        super().__init__()
        self.downsample = _Downsample(cin, cout) if downsample else nn.Identity()
        self.blocks = nn.Sequential(*[nn.Sequential(_SpatialBlock(cout, heads), _ChannelBlock(cout, heads)) for _ in range(depth)])


class _Head(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = _ln(dim)


class HipDaVit(nn.Module):
    def __init__(self, name="davit_tiny"):
        super().__init__()
        key = name.split(".")[0]
        if key not in DAVIT_CONFIGS:
            raise NotImplementedError(f"image encoder '{name}' has no MI355X kernels (available: {sorted(DAVIT_CONFIGS)})")
        depths, dims, heads = DAVIT_CONFIGS[key]
        self.num_features = dims[-1]
        self.stem = _Stem(dims[0])
        stages, cin = [], dims[0]
        for i in range(4):
            stages.append(_Stage(cin, dims[i], depths[i], heads[i], downsample=i > 0))
            cin = dims[i]
        self.stages = nn.Sequential(*stages)
        self.head = _Head(dims[-1])
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward_features(self, image):                                   # -> [B, H/32, W/32, C]
        x = image.float()
        # 7x7/4 stem conv (pad 3) = HIP patch extraction (zero padding in the kernel) + ONE Linear over the 147-wide rows.
        # timm pads the image to a multiple of 4 first; with pad 3 / stride 4 the output grid ceil(H/4) x ceil(W/4) and every
        # tap value are the same whether those zeros are materialised or come from the kernel's bounds check.
        B, _, Hi, Wi = x.shape
        H, W = (Hi + 3) // 4, (Wi + 3) // 4
        Hi_p, Wi_p = 4 * H, 4 * W
        if (Hi_p, Wi_p) != (Hi, Wi):       # ragged sizes only: the kernel's geometry is defined by the padded extent
            xp = x.new_zeros((B, 3, Hi_p, Wi_p)); xp[:, :, :Hi, :Wi] = x; x = xp
        cols = ops.patch_cols(x, 7, 4, 3)
        x = ops.linear(cols, self.stem.conv.weight.flatten(1), self.stem.conv.bias)
        C = x.shape[1]
        x = _layernorm(self.stem.norm, x).reshape(B, H, W, C)
        for stage in self.stages:
            if not isinstance(stage.downsample, nn.Identity):
                ds = stage.downsample
                B, H, W, C = x.shape
                x = _layernorm(ds.norm, x.reshape(B * H * W, C)).reshape(B, H, W, C)
                if H % 2 or W % 2:
                    x = torch.nn.functional.pad(x, (0, 0, 0, W % 2, 0, H % 2))
                    B, H, W, C = x.shape
                p = ops.patch_cols(x, 2, 2, 0, channels_last=True)                                          # columns ordered (c, kh, kw)
                x = ops.linear(p, ds.conv.weight.flatten(1), ds.conv.bias).reshape(B, H // 2, W // 2, -1)
            for pair in stage.blocks:
                x = pair[1](pair[0](x))
        return x

    def forward(self, image):                                            # reset_classifier(0): avg pool -> head.norm -> flatten
        x = self.forward_features(image)
        B, H, W, C = x.shape
        pooled = ops.token_mean(x.reshape(B, H * W, C), 0)
        return _layernorm(self.head.norm, pooled)

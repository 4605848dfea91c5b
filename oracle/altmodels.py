"""Oracle restatements of the reference's alternate forward(img, meta) models.  TEST INFRASTRUCTURE ONLY.

MD-Net (multimodalMDNet.py:7-102) and MetaNet+ResNet (metanet.py:26-147).  The head logic is pinned by
fixtures generated through the reference's own classes (oracle/gen_golden.py builds them with the
third-party backbone constructors replaced by this package's restated backbones -- torchvision / timm are
absent, so the backbone arithmetic itself stays PARITY UNPINNED, as everywhere else).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .backbones import OracleDenseNet169, OracleResNet


class OracleChannelGate(nn.Module):
    """MetaNet: multimodalMDNet.py:7-29 == metanet.py:26-50."""

    def __init__(self, in_channels, middle_channels, out_channels):
        super().__init__()
        self.metanet = nn.Sequential(nn.Conv2d(in_channels, middle_channels, 1), nn.ReLU(),
                                     nn.Conv2d(middle_channels, out_channels, 1), nn.Sigmoid())

    def forward(self, feat_maps, metadata):
        return self.metanet(metadata[:, :, None, None]) * feat_maps


class OracleSpatialMetaBlock(nn.Module):
    """multimodalMDNet.py:32-55."""

    def __init__(self, V, U):
        super().__init__()
        self.fb = nn.Sequential(nn.Linear(U, V), nn.LayerNorm(V))
        self.gb = nn.Sequential(nn.Linear(U, V), nn.LayerNorm(V))

    def forward(self, img_features, metadata):
        t1 = self.fb(metadata)[:, :, None, None]
        t2 = self.gb(metadata)[:, :, None, None]
        return torch.sigmoid(torch.tanh(img_features * t1) + t2)


class OracleMDNet(nn.Module):
    """multimodalMDNet.py:58-102."""

    def __init__(self, meta_dim=85, num_classes=6, hidden_dim=128, unfreeze_weights=False):
        super().__init__()
        self.num_channels = 1664
        densenet = OracleDenseNet169()
        for p in densenet.parameters():
            p.requires_grad = bool(unfreeze_weights)
        self.feature_extractor = densenet.features
        self.meta_net = OracleChannelGate(meta_dim, hidden_dim, self.num_channels)
        self.meta_block = OracleSpatialMetaBlock(self.num_channels, meta_dim)
        self.avg_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.classifier = nn.Linear(self.num_channels, num_classes)

    def forward(self, image, metadata):
        f = self.feature_extractor(image)
        fused = self.meta_net(f, metadata) + self.meta_block(f, metadata)
        return self.classifier(self.avg_pool(fused).flatten(1))


class OracleResNetFeatureMaps(OracleResNet):
    """timm.create_model("resnet50", num_classes=0, global_pool="") (metanet.py:83-88): layer4 feature maps."""

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        for li in range(1, 5):
            x = getattr(self, f"layer{li}")(x)
        return x


class OracleMetaNetModel(nn.Module):
    """metanet.py:56-147."""

    def __init__(self, meta_dim, num_classes=6, dropout_fraction=0.3, image_encoder="resnet50", unfreeze_weights=False):
        super().__init__()
        self.backbone = OracleResNetFeatureMaps({"resnet50": "resnet-50", "resnet18": "resnet-18"}[image_encoder])
        self.feat_dim = self.backbone.num_features
        if not unfreeze_weights:
            for p in self.backbone.parameters():
                p.requires_grad = False
        self.metanet = OracleChannelGate(meta_dim, 128, self.feat_dim)
        d = self.feat_dim
        self.classifier = nn.Sequential(
            nn.Linear(d, d), nn.LayerNorm(d), nn.ReLU(inplace=True), nn.Dropout(dropout_fraction),
            nn.Linear(d, d // 2), nn.LayerNorm(d // 2), nn.ReLU(inplace=True), nn.Dropout(dropout_fraction),
            nn.Linear(d // 2, num_classes))

    def forward(self, image, metadata):
        f = self.metanet(self.backbone(image), metadata)
        return self.classifier(F.adaptive_avg_pool2d(f, 1).flatten(1))


VIT_CONFIGS = {"vit_tiny_patch16_224": (192, 12, 3), "vit_small_patch16_224": (384, 12, 6),
               "vit_base_patch16_224": (768, 12, 12), "vit_large_patch16_224": (1024, 24, 16)}


class _ViTAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, L, E = x.shape
        qkv = self.qkv(x).reshape(B, L, 3, self.num_heads, E // self.num_heads).permute(2, 0, 3, 1, 4)
        o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(o.transpose(1, 2).reshape(B, L, E))


class _ViTMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _ViTBlock(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _ViTAttention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _ViTMlp(dim, dim * 4)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _ViTPatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, 16, 16)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class OracleViT(nn.Module):
    """timm VisionTransformer (vit_*_patch16_224, num_classes=0) restated: pre-norm blocks, qkv bias, LayerNorm eps 1e-6,
    exact GELU, class token + learned position embedding.  PARITY UNPINNED against timm (absent); keys follow it."""

    def __init__(self, name):
        super().__init__()
        dim, depth, heads = VIT_CONFIGS[name]
        self.num_features = dim
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, 197, dim) * 0.02)
        self.patch_embed = _ViTPatchEmbed(dim)
        self.blocks = nn.Sequential(*[_ViTBlock(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed[:, :x.shape[1] + 1]
        return self.norm(self.blocks(x))

    def forward(self, x):
        return self.forward_features(x)[:, 0]


class OracleLiwTERM(nn.Module):
    """liwtermModel.py:6-102."""

    def __init__(self, num_classes, meta_dim, image_encoder="vit_large_patch16_224", unfreeze_backbone=False, dropout=0.3):
        super().__init__()
        self.visual = OracleViT(image_encoder)
        if not unfreeze_backbone:
            for p in self.visual.parameters():
                p.requires_grad = False
        d = self.visual.num_features
        self.visual_proj = nn.Sequential(nn.Linear(d, 4096), nn.LayerNorm(4096), nn.ReLU(), nn.Dropout(dropout))
        self.meta_fc = nn.Sequential(nn.LayerNorm(meta_dim), nn.Linear(meta_dim, 1024), nn.ReLU())
        c = 4096 + 1024
        self.slm = nn.Sequential(
            nn.LayerNorm(c), nn.ReLU(), nn.Dropout(dropout),
            nn.Linear(c, 2048), nn.LayerNorm(2048), nn.ReLU(), nn.Dropout(dropout),
            nn.Linear(2048, 1024), nn.LayerNorm(1024), nn.ReLU(), nn.Dropout(dropout),
            nn.Linear(1024, 512), nn.LayerNorm(512), nn.ReLU(), nn.Dropout(dropout),
            nn.Linear(512, num_classes))

    def forward(self, image, metadata):
        v = self.visual.forward_features(image)[:, 0]
        return self.slm(torch.cat([self.visual_proj(v), self.meta_fc(metadata)], dim=1))


class _BeitAttention(nn.Module):
    def __init__(self, dim, heads, ws):
        super().__init__()
        self.num_heads = heads
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2 + 3, heads))
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)
        self.register_buffer("k_bias", torch.zeros(dim), persistent=False)
        num = (2 * ws - 1) ** 2 + 3
        coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        idx = torch.zeros((ws * ws + 1,) * 2, dtype=torch.long)
        idx[1:, 1:] = rel.sum(-1)
        idx[0, 0:] = num - 3
        idx[0:, 0] = num - 2
        idx[0, 0] = num - 1
        self.register_buffer("relative_position_index", idx, persistent=False)

    def forward(self, x):
        B, L, E = x.shape
        H = self.num_heads
        qkv = F.linear(x, self.qkv.weight, torch.cat([self.q_bias, self.k_bias.to(x.dtype), self.v_bias]))
        q, k, v = qkv.reshape(B, L, 3, H, E // H).permute(2, 0, 3, 1, 4)
        attn = (q * (E // H) ** -0.5) @ k.transpose(-2, -1)
        attn = attn + self.relative_position_bias_table[self.relative_position_index.view(-1)].view(L, L, H).permute(2, 0, 1)
        return self.proj((attn.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, L, E))


class _BeitBlock(nn.Module):
    def __init__(self, dim, heads, ws, init_values):
        super().__init__()
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _BeitAttention(dim, heads, ws)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _ViTMlp(dim, dim * 4)

    def forward(self, x):
        x = x + self.gamma_1 * self.attn(self.norm1(x))
        return x + self.gamma_2 * self.mlp(self.norm2(x))


BEIT_CONFIGS = {"beitv2_base_patch16_224": (768, 12, 12), "beitv2_large_patch16_224": (1024, 24, 16), "beitv2_tiny_test": (64, 2, 4)}


class OracleBeit(nn.Module):
    """timm Beit (beitv2_*_patch16_224 after reset_classifier(0)) restated: no absolute position embedding, per-block
    relative position bias, q/v bias, LayerScale, mean pooling over patch tokens + fc_norm.  PARITY UNPINNED against timm."""

    def __init__(self, name, init_values=1e-5):
        super().__init__()
        dim, depth, heads = BEIT_CONFIGS[name]
        self.num_features = dim
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.patch_embed = _ViTPatchEmbed(dim)
        self.blocks = nn.ModuleList([_BeitBlock(dim, heads, 14, init_values) for _ in range(depth)])
        self.fc_norm = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1)
        for blk in self.blocks:
            x = blk(x)
        return self.fc_norm(x[:, 1:].mean(dim=1))


class _LN2d(nn.LayerNorm):
    """timm LayerNorm2d: LayerNorm over the channel dim of NCHW."""

    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class _DvCpe(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)

    def forward(self, x):
        return x + self.proj(x)


class _DvWindowAttn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B_, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B_, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        attn = ((q * (C // self.num_heads) ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B_, N, C))


class _DvChannelAttn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.groups = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.groups, C // self.groups).permute(2, 0, 3, 1, 4)
        q = q * N ** -0.5                                   # timm 1.0.x: dynamic_scale=True
        attn = (q.transpose(-1, -2) @ k).softmax(dim=-1)
        x = (attn @ v.transpose(-1, -2)).transpose(-1, -2)
        return self.proj(x.transpose(1, 2).reshape(B, N, C))


class _DvBlock(nn.Module):
    def __init__(self, dim, heads, spatial):
        super().__init__()
        self.spatial = spatial
        self.cpe1 = _DvCpe(dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _DvWindowAttn(dim, heads) if spatial else _DvChannelAttn(dim, heads)
        self.cpe2 = _DvCpe(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _ViTMlp(dim, dim * 4)

    def forward(self, x):
        B, C, H, W = x.shape
        shortcut = self.cpe1(x).flatten(2).transpose(1, 2)
        h = self.norm1(shortcut)
        if self.spatial:
            ws = 7
            h = h.view(B, H, W, C)
            pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
            h = F.pad(h, (0, 0, 0, pr, 0, pb))
            Hp, Wp = H + pb, W + pr
            win = h.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
            a = self.attn(win).view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
            h = a[:, :H, :W, :].reshape(B, H * W, C)
        else:
            h = self.attn(h)
        x = shortcut + h
        x = self.cpe2(x.transpose(1, 2).view(B, C, H, W)).flatten(2).transpose(1, 2)
        x = x + self.mlp(self.norm2(x))
        return x.transpose(1, 2).view(B, C, H, W)


class _DvStem(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv2d(3, dim, 7, 4, 3)
        self.norm = _LN2d(dim)

    def forward(self, x):
        return self.norm(self.conv(x))


class _DvDown(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = _LN2d(cin)
        self.conv = nn.Conv2d(cin, cout, 2, 2)

    def forward(self, x):
        return self.conv(self.norm(x))


class _DvStage(nn.Module):
    def __init__(self, cin, cout, depth, heads, downsample):
        super().__init__()
        self.downsample = _DvDown(cin, cout) if downsample else nn.Identity()
        self.blocks = nn.Sequential(*[nn.Sequential(_DvBlock(cout, heads, True), _DvBlock(cout, heads, False)) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class _DvHead(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = _LN2d(dim)


DAVIT_CONFIGS = {"davit_tiny": ((1, 1, 3, 1), (96, 192, 384, 768), (3, 6, 12, 24))}


class OracleDaVit(nn.Module):
    """timm DaVit (davit_tiny after reset_classifier(0)) restated in timm's NCHW formulation.  PARITY UNPINNED against timm."""

    def __init__(self, name="davit_tiny"):
        super().__init__()
        depths, dims, heads = DAVIT_CONFIGS[name.split(".")[0]]
        self.num_features = dims[-1]
        self.stem = _DvStem(dims[0])
        stages, cin = [], dims[0]
        for i in range(4):
            stages.append(_DvStage(cin, dims[i], depths[i], heads[i], i > 0))
            cin = dims[i]
        self.stages = nn.Sequential(*stages)
        self.head = _DvHead(dims[-1])

    def forward(self, x):
        x = self.stages(self.stem(x))
        return self.head.norm(x.mean(dim=(2, 3), keepdim=True)).flatten(1)

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

"""MetaNet + ResNet on the HIP path -- drop-in for models/metanet.py (MetaNet :26-50, MetaNetModel :56-147),
selected with attention_mecanism == "metanet" (train_pad_20.py:353-359; the script passes
`str(model_name).replace("-", "")`, i.e. "resnet50" / "resnet18").

The reference gates the (B, C, H, W) feature maps with a per-(sample, channel) weight and then averages
over H x W; the gate is constant over the plane, so mean(feat * g) == g * mean(feat): the HIP path keeps
the backbone's fused global-average-pool and applies the gate to the pooled features (same value up to
fp32 summation order).  state_dict keys match the reference: `backbone.*` (timm's ResNet uses the
torchvision parameter names), `metanet.metanet.{0,2}.*`, `classifier.{0,1,4,5,8}.*`.
"""
import os
import sys

import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from mmskin import ops  # noqa: E402
from mmskin.backbone import HipResNet  # noqa: E402
from mmskin.nn import FusedAway, HipDropout, HipLayerNorm, HipLinear  # noqa: E402

_ENCODERS = {"resnet50": "resnet-50", "resnet18": "resnet-18"}


class MetaNet(nn.Module):
    def __init__(self, in_channels: int, middle_channels: int, out_channels: int):
        super().__init__()
        self.metanet = nn.Sequential(
            nn.Conv2d(in_channels, middle_channels, kernel_size=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(middle_channels, out_channels, kernel_size=1),
            nn.Sigmoid(),
        )

    def gate_logits(self, metadata):
        c1, c2 = self.metanet[0], self.metanet[2]
        h = ops.linear(metadata, c1.weight.flatten(1), c1.bias, True)
        return ops.linear(h, c2.weight.flatten(1), c2.bias)

    def forward(self, feat_maps, metadata):
        z = self.gate_logits(metadata)
        if feat_maps.dim() == 2:
            return ops.sigmoid_gate(z, feat_maps)
        return ops.sigmoid_gate(z[:, :, None, None].expand_as(feat_maps).contiguous(), feat_maps)


class MetaNetModel(nn.Module):
    def __init__(self, meta_dim: int, num_classes: int = 6, dropout_fraction: float = 0.3,
                 image_encoder: str = "resnet50", pretrained: bool = True, unfreeze_weights: bool = False):
        super().__init__()
        self.meta_dim = meta_dim
        self.num_classes = num_classes
        self.dropout_fraction = dropout_fraction
        self.image_encoder = image_encoder
        self.pretrained = pretrained          # accepted for signature parity; this package never downloads weights
        self.unfreeze_weights = unfreeze_weights
        if image_encoder not in _ENCODERS:
            raise NotImplementedError(f"MetaNetModel: image_encoder '{image_encoder}' has no MI355X plan "
                                      f"(available: {sorted(_ENCODERS)})")
        self.backbone = HipResNet(_ENCODERS[image_encoder])
        self.feat_dim = self.backbone.num_features
        if not self.unfreeze_weights:
            for p in self.backbone.parameters():
                p.requires_grad = False
        self.metanet = MetaNet(in_channels=self.meta_dim, middle_channels=128, out_channels=self.feat_dim)
        self.classifier = self.fc_mlp_module(self.feat_dim)

    def fc_mlp_module(self, input_dim: int) -> nn.Module:
        p = self.dropout_fraction
        return nn.Sequential(
            HipLinear(input_dim, input_dim), HipLayerNorm(input_dim, fuse_relu=True), FusedAway("ReLU"), HipDropout(p),
            HipLinear(input_dim, input_dim // 2), HipLayerNorm(input_dim // 2, fuse_relu=True), FusedAway("ReLU"),
            HipDropout(p),
            HipLinear(input_dim // 2, self.num_classes))

    def forward(self, image, metadata):
        pooled = self.backbone(image)                                  # GAP(layer4 feature maps), [B, feat_dim]
        pooled = self.metanet(pooled, metadata.float())                # == GAP(feat_maps * gate)
        return self.classifier(pooled)

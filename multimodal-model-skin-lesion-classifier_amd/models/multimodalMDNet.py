"""MD-Net on the HIP path -- drop-in for models/multimodalMDNet.py (MetaNet :7-29, MetaBlock :32-55,
MDNet :58-102), selected by the train scripts with attention_mecanism == "md-net"
(train_pad_20.py:338-345).

Same constructor arguments, sub-module names and state_dict keys (`feature_extractor.*` = torchvision
`densenet169().features`, `meta_net.metanet.{0,2}.*`, `meta_block.{fb,gb}.{0,1}.*`, `classifier.*`).
`MDNet.forward` runs the DenseNet-169 plan in feature-map mode and ONE fused kernel for
MetaNet gate + spatial MetaBlock + sum + global average pool (mmskin_mdnet_fuse_*).
"""
import os
import sys

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from mmskin import ops  # noqa: E402
from mmskin.backbone import HipDenseNetFeatures  # noqa: E402
from mmskin.nn import HipLayerNorm, HipLinear  # noqa: E402


class MetaNet(nn.Module):
    """metadata [B, in] -> 1x1 conv - ReLU - 1x1 conv - Sigmoid -> channel gate on the feature maps (reference :7-29)."""

    def __init__(self, in_channels, middle_channels, out_channels):
        super().__init__()
        self.metanet = nn.Sequential(
            nn.Conv2d(in_channels, middle_channels, 1),
            nn.ReLU(),
            nn.Conv2d(middle_channels, out_channels, 1),
            nn.Sigmoid(),
        )

    def gate_logits(self, metadata):
        """pre-sigmoid gate [B, out]: the 1x1 convolutions on a 1x1 map are two Linear layers."""
        c1, c2 = self.metanet[0], self.metanet[2]
        h = ops.linear(metadata, c1.weight.flatten(1), c1.bias, True)
        return ops.linear(h, c2.weight.flatten(1), c2.bias)

    def forward(self, feat_maps, metadata):
        z = self.gate_logits(metadata)
        return ops.sigmoid_gate(z[:, :, None, None].expand_as(feat_maps).contiguous(), feat_maps)


class MetaBlock(nn.Module):
    """sigmoid(tanh(feat * fb(meta)) + gb(meta)) per pixel (reference :32-55)."""

    def __init__(self, V, U):
        super().__init__()
        self.fb = nn.Sequential(HipLinear(U, V), HipLayerNorm(V))
        self.gb = nn.Sequential(HipLinear(U, V), HipLayerNorm(V))

    def forward(self, img_features, metadata):
        t1 = self.fb(metadata)[:, :, None, None].expand_as(img_features).contiguous()
        t2 = self.gb(metadata)[:, :, None, None].expand_as(img_features).contiguous()
        return ops.metablock_gate(img_features.contiguous(), t1, t2)


class MDNet(nn.Module):
    def __init__(self, meta_dim=85, num_classes=6, cnn_model_name="densenet169", text_model_name="one-hot-encode",
                 hidden_dim=128, device="cpu", unfreeze_weights=False):
        super().__init__()
        self.device = device
        self.num_channels = 1664
        self.meta_dim = meta_dim
        self.num_classes = num_classes
        self.cnn_model_name = cnn_model_name
        self.text_model_name = text_model_name
        self.feature_extractor = HipDenseNetFeatures()
        for param in self.feature_extractor.parameters():        # reference :73-75
            param.requires_grad = bool(unfreeze_weights)
        self.meta_net = MetaNet(in_channels=meta_dim, middle_channels=hidden_dim, out_channels=self.num_channels)
        self.meta_block = MetaBlock(V=self.num_channels, U=meta_dim)
        self.avg_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.classifier = HipLinear(self.num_channels, self.num_classes)

    def forward(self, image, metadata):
        image_features = self.feature_extractor(image)                         # [B, 1664, H', W']
        metadata = metadata.float()
        z = self.meta_net.gate_logits(metadata)
        t1, t2 = self.meta_block.fb(metadata), self.meta_block.gb(metadata)
        pooled = ops.mdnet_fuse(image_features, z, t1, t2)                      # gate + MetaBlock + sum + GAP
        return self.classifier(pooled)

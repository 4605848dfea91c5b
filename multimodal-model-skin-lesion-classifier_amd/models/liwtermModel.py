"""LiwTERM on the HIP path -- drop-in for models/liwtermModel.py:6-102, selected with attention_mecanism == "liwterm"
(train_pad_20.py:346-352).  Same constructor, sub-module names and Sequential indices (=> state_dict keys):
`visual.*` (timm ViT keys), `visual_proj.{0,1}`, `meta_fc.{0,1}`, `slm.{0,3,4,7,8,11,12,15}`."""
import os
import sys

import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (_PKG, os.path.dirname(os.path.abspath(__file__))):
    if _p not in sys.path:
        sys.path.insert(0, _p)

from mmskin import ops  # noqa: E402
from mmskin.nn import FusedAway, HipDropout, HipLayerNorm, HipLinear  # noqa: E402
from hip_vit import HipVisionTransformer  # noqa: E402


class LiwTERM(nn.Module):
    def __init__(self, num_classes: int, meta_dim: int, image_encoder: str = "vit_large_patch16_224", pretrained: bool = True,
                 unfreeze_backbone: bool = False, dropout: float = 0.3):
        super().__init__()
        self.visual = HipVisionTransformer(image_encoder)            # pretrained weights are loaded by the caller (no network)
        self.visual_dim = self.visual.num_features
        if not unfreeze_backbone:
            for p in self.visual.parameters():
                p.requires_grad = False
        self.visual_proj = nn.Sequential(HipLinear(self.visual_dim, 4096), HipLayerNorm(4096, fuse_relu=True), FusedAway("ReLU"),
                                         HipDropout(dropout))
        self.meta_fc = nn.Sequential(HipLayerNorm(meta_dim), HipLinear(meta_dim, 1024, fuse_relu=True), FusedAway("ReLU"))
        concat_dim = 4096 + 1024
        self.slm = nn.Sequential(
            HipLayerNorm(concat_dim, fuse_relu=True), FusedAway("ReLU"), HipDropout(dropout),
            HipLinear(concat_dim, 2048), HipLayerNorm(2048, fuse_relu=True), FusedAway("ReLU"), HipDropout(dropout),
            HipLinear(2048, 1024), HipLayerNorm(1024, fuse_relu=True), FusedAway("ReLU"), HipDropout(dropout),
            HipLinear(1024, 512), HipLayerNorm(512, fuse_relu=True), FusedAway("ReLU"), HipDropout(dropout),
            HipLinear(512, num_classes))

    def forward(self, image, metadata):
        v = self.visual.forward_features(image)
        if v.dim() == 3:
            v = v[:, 0]                                               # CLS token
        v = self.visual_proj(v.contiguous())
        m = self.meta_fc(metadata.float())
        return self.slm(ops.concat2(v, m))

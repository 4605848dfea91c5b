"""the reference's alternate forward(img, meta) models on the HIP path --
MD-Net (models/multimodalMDNet.py) and MetaNet+ResNet (models/metanet.py) -- against fixtures recorded through
the reference's own classes (tests/golden/alt_models.json) and against the oracle restatements."""
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import check_record_against_golden, golden, summarize, train_step_record
from gpu_util import DEV, rel_err
from oracle.detinit import det_init_, det_inputs, det_tensor

pytestmark = pytest.mark.gpu
BACKBONE = ("feature_extractor.", "backbone.", "visual.")


def _build(which):
    os.environ["MMSKIN_BACKBONE_DTYPE"] = "fp32"
    if which == "mdnet":
        from models.multimodalMDNet import MDNet
        return MDNet(meta_dim=20, num_classes=6, unfreeze_weights=True, device=DEV)
    if which == "liwterm":
        from models.liwtermModel import LiwTERM
        return LiwTERM(num_classes=6, meta_dim=20, image_encoder="vit_tiny_patch16_224", pretrained=False, unfreeze_backbone=True)
    from models.metanet import MetaNetModel
    return MetaNetModel(meta_dim=20, num_classes=6, image_encoder="resnet18", unfreeze_weights=True)


@pytest.mark.parametrize("which", ["mdnet", "metanet", "liwterm"])
def test_alternate_model_matches_reference_fixture(which):
    gold = golden("alt_models")[which]
    model = det_init_(_build(which)).to(DEV)
    img, meta, lab = det_inputs(2, 224, 20, 6) if which == "liwterm" else det_inputs(3, 64, 20, 6)
    rec = train_step_record(model, img.to(DEV), meta.to(DEV), lab.to(DEV), device=DEV)
    check_record_against_golden(rec, gold, 1e-3, 1e-5, skip_prefix=BACKBONE)        # logits, loss, head gradients, Adam
    for k, g in gold["grads"].items():                                            # backbone: noise-aware (see CPU test)
        if k.startswith(BACKBONE):
            assert rec["grads"][k] is not None
            assert abs(summarize(rec["grads"][k])["abs"] - g["abs"]) <= 0.05 * g["abs"] + 1e-7, k


def test_mdnet_fuse_kernel_vs_torch():
    from mmskin import ops
    g = torch.Generator().manual_seed(9)
    N, C, H, W = 5, 96, 7, 7
    f = torch.randn(N, C, H, W, generator=g); z, t1, t2 = (torch.randn(N, C, generator=g) for _ in range(3))
    fr, zr, t1r, t2r = (t.clone().requires_grad_(True) for t in (f, z, t1, t2))
    ref = (torch.sigmoid(zr)[:, :, None, None] * fr
           + torch.sigmoid(torch.tanh(fr * t1r[:, :, None, None]) + t2r[:, :, None, None])).mean(dim=(2, 3))
    dp = torch.randn(N, C, generator=g)
    ref.backward(dp)
    fd, zd, t1d, t2d = (t.to(DEV).requires_grad_(True) for t in (f, z, t1, t2))
    out = ops.mdnet_fuse(fd, zd, t1d, t2d)
    out.backward(dp.to(DEV))
    for a, b in ((out, ref), (fd.grad, fr.grad), (zd.grad, zr.grad), (t1d.grad, t1r.grad), (t2d.grad, t2r.grad)):
        assert rel_err(a, b) < 1e-4


def test_mdnet_submodules_standalone():
    """MetaNet / MetaBlock keep their stand-alone (B, C, H, W) semantics (reference :21-29, :47-55)."""
    from models.multimodalMDNet import MetaBlock, MetaNet
    from oracle.altmodels import OracleChannelGate, OracleSpatialMetaBlock
    feat, meta = det_tensor("alt.f", (2, 32, 5, 5)), det_tensor("alt.m", (2, 12))
    for hip, ora in ((MetaNet(12, 16, 32), OracleChannelGate(12, 16, 32)), (MetaBlock(32, 12), OracleSpatialMetaBlock(32, 12))):
        det_init_(ora); hip.load_state_dict(ora.state_dict()); hip = hip.to(DEV)
        assert rel_err(hip(feat.to(DEV), meta.to(DEV)), ora(feat, meta)) < 1e-4


def test_mdnet_bf16_and_frozen_backbone():
    """bf16 feature-map plan runs a train step; frozen backbone (reference default) gets no gradients."""
    os.environ["MMSKIN_BACKBONE_DTYPE"] = "bf16"
    from models.multimodalMDNet import MDNet
    model = MDNet(meta_dim=20, num_classes=6, device=DEV).to(DEV).train()
    img, meta, lab = det_inputs(4, 96, 20, 6)
    loss = F.cross_entropy(model(img.to(DEV), meta.to(DEV)), lab.to(DEV))
    loss.backward()
    assert torch.isfinite(loss)
    assert all(p.grad is None for p in model.feature_extractor.parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in model.named_parameters()
               if not n.startswith("feature_extractor."))


def test_vit_large_backbone_runs_frozen():
    """LiwTERM's default: frozen vit_large_patch16_224 (304 M parameters), trainable projection + SLM head."""
    from models.liwtermModel import LiwTERM
    model = LiwTERM(num_classes=6, meta_dim=20).to(DEV).train()
    assert sum(p.numel() for p in model.visual.parameters()) == 303301632          # timm vit_large_patch16_224, num_classes=0
    img, meta, lab = det_inputs(2, 224, 20, 6)
    loss = F.cross_entropy(model(img.to(DEV), meta.to(DEV)), lab.to(DEV))
    loss.backward()
    assert torch.isfinite(loss)
    assert all(p.grad is None for p in model.visual.parameters())
    assert all(p.grad is not None for n, p in model.named_parameters() if not n.startswith("visual."))

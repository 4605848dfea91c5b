"""-m gpu: BASELINE.json configs[2..4] at their STATED per-GPU sizes.  Full-batch oracle runs of these
models do not fit a test budget on the CPU, so each config is checked through size-independent properties at full size --
(a) a few samples of the eval-mode batch against the CPU oracle run on those samples alone (eval-mode normalisation makes
samples independent), (b) batch-permutation equivariance, (c) bit-exact repeatability, (d) one training step with finite
gradients for every live parameter -- and, for the transformer backbones, in both operand modes of the large Linear GEMMs
(fp32 = parity mode; bf16 = the dtype configs[3] is quoted in), through the MODEL, not a single op.

Per-GPU batches: configs[2] 256 on one GPU; configs[3] 512 over DP=8 -> 64; configs[4] 1024 over DP=8 -> 128."""
import os

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, rel_err
from helpers import SMALL, disable_dropout
from mmskin import ops
from models import multimodalIntraInterModal as M
from oracle.detinit import det_init_, det_tensor
from oracle.model import OracleMultimodalModel

pytestmark = pytest.mark.gpu


def _tab_meta(B, g):
    cat = torch.randint(0, 10, (B, 82), generator=g).float()
    return torch.cat([cat, torch.randn(B, 4, generator=g)], dim=1)


def _props(model, img, meta, lab, atol_perm):
    """(b) permutation equivariance, (c) repeatability, (d) finite train step; returns the eval logits."""
    B = img.shape[0]
    g = torch.Generator().manual_seed(7)
    model.eval()
    with torch.no_grad():
        full = model(img, meta).float().cpu()
        again = model(img, meta).float().cpu()
        perm = torch.randperm(B, generator=g)
        pm = {k: v[perm.to(v.device)] for k, v in meta.items()} if isinstance(meta, dict) else meta[perm.to(meta.device)]
        permuted = model(img[perm.to(img.device)], pm).float().cpu()
    assert torch.equal(full, again)
    assert torch.allclose(permuted, full[perm], atol=atol_perm), (permuted - full[perm]).abs().max()
    model.train(); disable_dropout(model)
    model.zero_grad(set_to_none=True)
    loss = F.cross_entropy(model(img, meta), lab)
    loss.backward()
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert torch.isfinite(loss) and grads and all(torch.isfinite(gr).all() for gr in grads)
    return full


def test_config3_densenet_tabtransformer_metablock_b256():
    """configs[2]: DenseNet-169 + tab-transformer + metablock, batch 256 @ 224^2, bf16 backbone compute."""
    os.environ["MMSKIN_BACKBONE_DTYPE"] = "bf16"
    kw = dict(SMALL, cnn_model_name="densenet169", text_model_name="tab-transformer", attention_mecanism="metablock",
              vocab_size=86, common_dim=512, text_encoder_dim_output=512, unfreeze_weights="unfrozen_weights")
    cpu = det_init_(OracleMultimodalModel(**dict(kw, device="cpu"))).eval()
    hip = M.MultimodalModel(**dict(kw, device=DEV))
    hip.load_state_dict(cpu.state_dict(), strict=True)
    hip = hip.to(DEV)
    B = 256
    g = torch.Generator().manual_seed(0)
    img, meta, lab = torch.randn(B, 3, 224, 224, generator=g), _tab_meta(B, g), torch.randint(0, 6, (B,), generator=g)
    full = _props(hip, img.to(DEV), meta.to(DEV), lab.to(DEV), atol_perm=2e-3)
    pick = [0, 85, 170, 255]
    with torch.no_grad():
        want = cpu(img[pick], meta[pick])
    err = float((full[pick] - want).abs().max())
    assert err < 1e-2, err                                                         # north_star: 1e-2 bf16


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_config4_davit_tabtransformer_gfcam_b64(mode):
    """configs[3]: davit_tiny.msft_in1k + tab-transformer + gfcam at the per-GPU batch 64 (512 over DP=8); the image
    encoder's eval features of four samples against the oracle restatement (timm absent: parity unpinned)."""
    from oracle.altmodels import OracleDaVit
    prev = ops.get_linear_dtype()
    ops.set_linear_dtype(mode)
    try:
        kw = dict(SMALL, cnn_model_name="davit_tiny.msft_in1k", text_model_name="tab-transformer", attention_mecanism="gfcam",
                  vocab_size=86, common_dim=512, text_encoder_dim_output=512, unfreeze_weights="unfrozen_weights", device=DEV)
        model = M.MultimodalModel(**kw)
        ref = det_init_(OracleDaVit()).eval()
        model.image_encoder.load_state_dict(ref.state_dict(), strict=True)
        model = model.to(DEV)
        B = 64
        g = torch.Generator().manual_seed(1)
        img, meta, lab = torch.randn(B, 3, 224, 224, generator=g), _tab_meta(B, g), torch.randint(0, 6, (B,), generator=g)
        _props(model, img.to(DEV), meta.to(DEV), lab.to(DEV), atol_perm=1e-4 if mode == "fp32" else 5e-3)
        pick = [0, 21, 42, 63]
        model.eval()
        with torch.no_grad():
            feats = model.image_encoder(img.to(DEV)).cpu()
            want = ref(img[pick])
        err = rel_err(feats[pick], want)
        assert err < (5e-4 if mode == "fp32" else 3e-2), (mode, err)                  # bf16 operands: ~2^-8 per GEMM over 20 layers
    finally:
        ops.set_linear_dtype(prev)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_config5_beitv2_large_bert_rgatt_b128(mode):
    """configs[4]: beitv2_large_patch16_224 + bert-base-uncased (512 tokens) + the RG-ATT string at the per-GPU batch 128
    (1024 over DP=8), last-block fine-tuning ('partial'); BEiT-large eval features of two samples against the oracle."""
    from oracle.altmodels import OracleBeit
    prev = ops.get_linear_dtype()
    ops.set_linear_dtype(mode)
    try:
        model = M.MultimodalModel(num_classes=6, num_heads=8, device=DEV, cnn_model_name="beitv2_large_patch16_224",
                                  text_model_name="bert-base-uncased", common_dim=512, vocab_size=20, unfreeze_weights="partial",
                                  attention_mecanism="att-intramodal+residual+cross-attention-metadados", n=2)
        ref = det_init_(OracleBeit("beitv2_large_patch16_224", init_values=0.1)).eval()
        model.image_encoder.load_state_dict(ref.state_dict(), strict=True)
        model = model.to(DEV)
        B = 128
        g = torch.Generator().manual_seed(2)
        img, lab = torch.randn(B, 3, 224, 224, generator=g), torch.randint(0, 6, (B,), generator=g)
        ids = torch.randint(1, 30000, (B, 1, 512), generator=g)
        mask = torch.ones_like(ids)
        mask[::3, :, 300:] = 0                                                         # padded metadata sentences
        meta = {"input_ids": ids.to(DEV), "attention_mask": mask.to(DEV)}
        _props(model, img.to(DEV), meta, lab.to(DEV), atol_perm=1e-4 if mode == "fp32" else 5e-3)
        pick = [0, 127]
        model.eval()
        with torch.no_grad():
            feats = model.image_encoder(img.to(DEV)).cpu()
            want = ref(img[pick])
        err = rel_err(feats[pick], want)
        assert err < (1e-3 if mode == "fp32" else 3e-2), (mode, err)
        live = [n for n, p in model.named_parameters() if p.grad is not None and n.startswith("image_encoder")]
        assert live and all(n.startswith("image_encoder.blocks.23.") or n.startswith("image_encoder.fc_norm") for n in live)
    finally:
        ops.set_linear_dtype(prev)


def test_patch_cols_matches_unfold():
    """HIP im2col (timm PatchEmbed / DaViT stem + downsample) vs F.unfold, forward and backward, NCHW and NHWC inputs."""
    for (N, C, H, W, k, s, p) in [(2, 3, 224, 224, 7, 4, 3), (2, 3, 64, 48, 16, 16, 0), (3, 3, 70, 50, 16, 16, 0), (2, 96, 14, 10, 2, 2, 0)]:
        x = det_tensor("pc.x", (N, C, H, W))
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        xr = x.clone().requires_grad_(True)
        want = F.unfold(xr, kernel_size=k, padding=p, stride=s).transpose(1, 2).reshape(N * OH * OW, C * k * k)   # the ragged edge is never read
        w = det_tensor("pc.w", tuple(want.shape))
        (want * w).sum().backward()
        for cl in (False, True):
            xi = (x.permute(0, 2, 3, 1).contiguous() if cl else x.clone()).to(DEV).requires_grad_(True)
            got = ops.patch_cols(xi, k, s, p, channels_last=cl)
            assert torch.equal(got.detach().cpu(), want.detach()), (N, C, H, W, k, s, p, cl)
            (got * w.to(DEV)).sum().backward()
            gx = xi.grad.cpu()
            gx = gx.permute(0, 3, 1, 2) if cl else gx
            assert torch.allclose(gx, xr.grad, rtol=1e-5, atol=1e-6), (N, C, H, W, k, s, p, cl)

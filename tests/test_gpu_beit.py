"""BEiT-v2 image encoder (the reference's generic timm branch, loadImageModelClassifier.py:117-152; BASELINE.json
configs[4]: beitv2_large_patch16_224 + bert-base-uncased + RG-ATT) on the HIP ops vs the oracle restatement (timm is
absent: parity unpinned, module tree / keys follow timm's Beit)."""
import os

import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, rel_err
from oracle.altmodels import OracleBeit
from oracle.detinit import det_init_, det_tensor

pytestmark = pytest.mark.gpu


def test_beit_matches_oracle():
    from models.hip_beit import HipBeit
    cpu = det_init_(OracleBeit("beitv2_tiny_test", init_values=0.5))     # LayerScale 0.5: the branches matter in the comparison
    hip = HipBeit("beitv2_tiny_test", init_values=0.5)
    hip.load_state_dict(cpu.state_dict(), strict=True)
    hip = hip.to(DEV)
    x = det_tensor("beit.x", (3, 3, 224, 224))
    w = det_tensor("beit.w", (3, 64))
    res = {}
    for name, m, dev in (("cpu", cpu, "cpu"), ("hip", hip, DEV)):
        m.train()
        f = m(x.to(dev))
        (f * w.to(dev)).sum().backward()
        res[name] = (f.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()})
    assert rel_err(res["hip"][0], res["cpu"][0]) < 2e-4
    assert set(res["hip"][1]) == set(res["cpu"][1])
    scale = max(float(v.abs().max()) for v in res["cpu"][1].values())
    for k, g in res["cpu"][1].items():
        err = float((res["hip"][1][k] - g).abs().max())
        assert err < 2e-3 * max(float(g.abs().max()), 1e-3 * scale), (k, err)


def test_beit_bf16_operand_mode_gradients_vs_emulation():
    """A 2-block BEiT with every block trainable in bf16-OPERAND mode, batch 12 (2 364 token rows: every Linear of the blocks and the
    patch embedding take the bf16 GEMM kernels; the trainable attention (L = 197 > 64) runs on the fused flash forward + backward kernels,
    which round q / k / v and the probabilities to bf16 -- the emulation rounds the Linear operands only, so the attention's own rounding is
    inside the 1.5 x slack): per-parameter gradient distance /
    cosine to the fp32 oracle, bounded by the CPU bf16-operand emulation of the oracle (tests/bf16_emulation.py)."""
    from bf16_emulation import assert_grads_not_worse_than_emulation, bf16_operand_emulation, grad_distance_report
    from mmskin import ops
    from models.hip_beit import HipBeit
    x = det_tensor("beit.xb", (12, 3, 224, 224))
    w = det_tensor("beit.wb", (12, 64))

    def run(m, dev):
        m.train()
        f = m(x.to(dev))
        (f * w.to(dev)).sum().backward()
        return f.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()}

    cpu = det_init_(OracleBeit("beitv2_tiny_test", init_values=0.5))
    f_ref, g_ref = run(cpu, "cpu")
    emu = det_init_(OracleBeit("beitv2_tiny_test", init_values=0.5))
    with bf16_operand_emulation():
        f_emu, g_emu = run(emu, "cpu")
    hip = HipBeit("beitv2_tiny_test", init_values=0.5)
    hip.load_state_dict(cpu.state_dict(), strict=True)
    hip = hip.to(DEV)
    prev = ops.get_linear_dtype()
    try:
        ops.set_linear_dtype("bf16")
        f_hip, g_hip = run(hip, DEV)
    finally:
        ops.set_linear_dtype(prev)
    assert rel_err(f_emu, f_ref) > 1e-4
    assert rel_err(f_hip, f_ref) <= 1.5 * rel_err(f_emu, f_ref) + 1e-3, (rel_err(f_hip, f_ref), rel_err(f_emu, f_ref))
    rows = grad_distance_report(g_ref, g_hip, g_emu)
    worst = max(rows.values(), key=lambda v: v[0])
    print("beit bf16 operand mode: features", rel_err(f_hip, f_ref), "emu", rel_err(f_emu, f_ref), "worst grad (l2 hip, cos hip, l2 emu, cos emu)", worst)
    assert_grads_not_worse_than_emulation(rows)


def test_config5_wiring_beitv2_large_bert():
    """BASELINE configs[4]: beitv2_large + bert-base-uncased + the RG-ATT fusion string, one (small-batch) training step."""
    from models import multimodalIntraInterModal as M
    model = M.MultimodalModel(num_classes=6, num_heads=8, device=DEV, cnn_model_name="beitv2_large_patch16_224",
                              text_model_name="bert-base-uncased", common_dim=512, vocab_size=20, unfreeze_weights="partial",
                              attention_mecanism="att-intramodal+residual+cross-attention-metadados", n=2).to(DEV).train()
    assert model.cnn_dim_output == 1024 and model.text_encoder_dim_output == 768
    img = torch.randn(2, 3, 224, 224, device=DEV)
    ids = torch.randint(1, 30000, (2, 1, 512), device=DEV)
    meta = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    out = model(img, meta)
    assert out.shape == (2, 6)
    F.cross_entropy(out, torch.tensor([1, 4], device=DEV)).backward()
    trainable = [n for n, p in model.named_parameters() if p.requires_grad and n.startswith("image_encoder")]
    assert trainable and all(n.startswith("image_encoder.blocks.23.") for n in trainable)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in model.named_parameters()
               if p.requires_grad and n.startswith(("image_encoder.blocks.23.", "fc_fusion.")))
    assert all(p.grad is None for p in model.text_encoder.parameters())

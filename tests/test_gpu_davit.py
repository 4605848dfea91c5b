"""DaViT-tiny image encoder (the reference's generic timm branch; BASELINE.json configs[3]: davit_tiny.msft_in1k +
tab-transformer + gfcam) on the HIP ops vs the oracle restatement in timm's NCHW formulation (timm is absent: parity
unpinned; module tree / keys follow timm's DaVit, channel attention as in timm 1.0.x)."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import DEV, rel_err
from helpers import SMALL, disable_dropout
from oracle.altmodels import OracleDaVit
from oracle.detinit import det_init_, det_inputs, det_tensor

pytestmark = pytest.mark.gpu


def test_davit_matches_oracle():
    from models.hip_davit import HipDaVit
    cpu = det_init_(OracleDaVit())
    hip = HipDaVit("davit_tiny.msft_in1k")
    hip.load_state_dict(cpu.state_dict(), strict=True)
    hip = hip.to(DEV)
    x = det_tensor("davit.x", (2, 3, 224, 224))
    w = det_tensor("davit.w", (2, 768))
    res = {}
    for name, m, dev in (("cpu", cpu, "cpu"), ("hip", hip, DEV)):
        m.train()
        f = m(x.to(dev))
        (f * w.to(dev)).sum().backward()
        res[name] = (f.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()})
    assert rel_err(res["hip"][0], res["cpu"][0]) < 5e-4, rel_err(res["hip"][0], res["cpu"][0])
    assert set(res["hip"][1]) == set(res["cpu"][1])
    scale = max(float(v.abs().max()) for v in res["cpu"][1].values())
    bad = {}
    for k, g in res["cpu"][1].items():
        err = float((res["hip"][1][k] - g).abs().max())
        if err > 5e-3 * max(float(g.abs().max()), 1e-3 * scale):
            bad[k] = (err, float(g.abs().max()))
    assert not bad, bad


def test_davit_bf16_operand_mode_gradients_vs_emulation():
    """BASELINE configs[3] is quoted in bf16 and every DaViT block is trainable there: the backward of the bf16-OPERAND mode needs a
    quantitative check in that dtype.  Batch 4 @ 224^2 puts the Linears of stages 1 - 2 (12 544 / 3 136 token
    rows) and the stage-2 patch embedding on the bf16 GEMM kernels.  Per parameter: relative L2 distance and cosine of the HIP
    gradient to the fp32 oracle's, bounded by the CPU bf16-operand emulation of the oracle (tests/bf16_emulation.py: the same
    operand roundings, torch's summation order) -- at most 1.5 x its distance (+2e-3), cosine no more than 0.02 below its."""
    from bf16_emulation import assert_grads_not_worse_than_emulation, bf16_operand_emulation, grad_distance_report
    from mmskin import ops
    from models.hip_davit import HipDaVit
    x = det_tensor("davit.xb", (4, 3, 224, 224))
    w = det_tensor("davit.wb", (4, 768))

    def run(m, dev):
        m.train()
        f = m(x.to(dev))
        (f * w.to(dev)).sum().backward()
        return f.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()}

    cpu = det_init_(OracleDaVit())
    f_ref, g_ref = run(cpu, "cpu")
    emu = det_init_(OracleDaVit())
    with bf16_operand_emulation():
        f_emu, g_emu = run(emu, "cpu")
    hip = HipDaVit("davit_tiny.msft_in1k")
    hip.load_state_dict(cpu.state_dict(), strict=True)
    hip = hip.to(DEV)
    prev = ops.get_linear_dtype()
    try:
        ops.set_linear_dtype("bf16")
        f_hip, g_hip = run(hip, DEV)
    finally:
        ops.set_linear_dtype(prev)
    assert rel_err(f_emu, f_ref) > 1e-4                       # the emulation really rounds something at this size
    assert rel_err(f_hip, f_ref) <= 1.5 * rel_err(f_emu, f_ref) + 1e-3, (rel_err(f_hip, f_ref), rel_err(f_emu, f_ref))
    rows = grad_distance_report(g_ref, g_hip, g_emu)
    worst = max(rows.values(), key=lambda v: v[0])
    print("davit bf16 operand mode: features", rel_err(f_hip, f_ref), "emu", rel_err(f_emu, f_ref), "worst grad (l2 hip, cos hip, l2 emu, cos emu)", worst)
    assert_grads_not_worse_than_emulation(rows)


def test_davit_odd_input_size_and_config4_wiring():
    """Window padding / crop at a size that is not a multiple of 7*32, and BASELINE configs[3] end to end."""
    from models import multimodalIntraInterModal as M
    from models.hip_davit import HipDaVit
    cpu = det_init_(OracleDaVit()).eval()
    hip = HipDaVit("davit_tiny")
    hip.load_state_dict(cpu.state_dict(), strict=True)
    hip = hip.to(DEV).eval()
    x = det_tensor("davit.x2", (1, 3, 160, 192))
    with torch.no_grad():
        assert rel_err(hip(x.to(DEV)).cpu(), cpu(x)) < 5e-4
    model = M.MultimodalModel(**dict(SMALL, cnn_model_name="davit_tiny.msft_in1k", text_model_name="tab-transformer", vocab_size=86,
                                     attention_mecanism="gfcam", unfreeze_weights="unfrozen_weights", device=DEV)).to(DEV).train()
    disable_dropout(model)
    img, _, lab = det_inputs(2, 224, 20, 6)
    xc = (det_tensor("tt.cat", (2, 82)).abs() * 10).long().clamp_(0, 9)
    meta = torch.cat([xc.float(), det_tensor("tt.num", (2, 4))], dim=1)
    out = model(img.to(DEV), meta.to(DEV))
    F.cross_entropy(out, lab.to(DEV)).backward()
    assert out.shape == (2, 6) and model.cnn_dim_output == 768
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.image_encoder.parameters())

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

"""VGG-16 image encoder (loadImageModelClassifier.py:77-81; used by the reference's experiment lists next to
densenet169 / resnet-50, train_isic_2020.py:341) on the HIP plan executor vs the CPU oracle (torchvision layout,
parity unpinned against torchvision itself)."""
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import SMALL, disable_dropout
from gpu_util import DEV, rel_err
from oracle.backbones import OracleVGG16
from oracle.detinit import det_init_, det_inputs, det_tensor
from oracle.model import OracleMultimodalModel

pytestmark = pytest.mark.gpu


def _l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _rb(t):
    return t.bfloat16().float()


def _bf16_storage_emulation(model):
    """CPU oracle with bf16 STORAGE of conv weights and of every ReLU / pool output (fp32 arithmetic in between)."""
    for m in model.features.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = _rb(m.weight.data)
        if isinstance(m, (torch.nn.ReLU, torch.nn.MaxPool2d)):
            m.register_forward_hook(lambda mod, i, o: _rb(o))
    return model


def _pair(dtype):
    from mmskin.backbone import HipVGG16
    cpu = det_init_(OracleVGG16())
    hip = HipVGG16(compute_dtype=dtype)
    hip.load_state_dict(cpu.state_dict(), strict=True)
    return cpu, hip.to(DEV)


def test_vgg_pieces_vs_torch():
    """2x2 max-pool fwd / fused un-pool + ReLU mask, adaptive average pool, through the features plan at a size whose
    last map is not 7x7 (adaptive bins) and at an odd size (floor pooling)."""
    for hw in (96, 80):
        cpu, hip = _pair("fp32")
        cpu.eval(); hip.eval()
        x = det_tensor("vgg.x%d" % hw, (2, 3, hw, hw))
        with torch.no_grad():
            a, b = cpu.avgpool(cpu.features(x)), hip.features(x.to(DEV)).cpu()
        assert b.shape == (2, 512, 7, 7)
        assert rel_err(b, a) < 1e-4, (hw, rel_err(b, a))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_vgg_train_step_vs_oracle(dtype):
    cpu, hip = _pair(dtype)
    truth = det_init_(OracleVGG16()).double()
    x = det_tensor("vgg.img", (4, 3, 64, 64))
    w = det_tensor("vgg.w", (4, 4096))
    outs, grads = {}, {}
    runs = [("cpu", cpu, x, w), ("truth", truth, x.double(), w.double()), ("hip", hip, x.to(DEV), w.to(DEV))]
    if dtype == "bf16":
        runs.append(("emu", _bf16_storage_emulation(det_init_(OracleVGG16())), _rb(x), w))
    for name, m, xi, wi in runs:
        m.train(); disable_dropout(m)
        f = m(xi)
        (f * wi).sum().backward()
        outs[name] = f.detach().cpu().double()
        grads[name] = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters()}
    assert set(grads["hip"]) == set(grads["truth"])
    keys = list(grads["truth"])
    f_hip, f_cpu = _l2(outs["hip"], outs["truth"]), _l2(outs["cpu"], outs["truth"])
    hip_l2 = {k: _l2(grads["hip"][k], grads["truth"][k]) for k in keys}
    cpu_l2 = {k: _l2(grads["cpu"][k], grads["truth"][k]) for k in keys}
    worst = max(keys, key=lambda k: hip_l2[k])
    print(dtype, "feat", f_hip, f_cpu, "grad worst", worst, hip_l2[worst], cpu_l2[worst])
    assert all(torch.isfinite(g).all() for g in grads["hip"].values())
    if dtype == "fp32":
        assert f_hip < 1e-4
        assert all(hip_l2[k] <= 3 * cpu_l2[k] + 2e-4 for k in keys), worst
    else:
        # early-layer gradients of this deterministic-random-init net are sums with heavy cancellation: merely
        # rounding weights / activations to bf16 on the CPU moves features.0.weight's gradient by ~70 %.  The HIP
        # bf16 path is held to that emulation, layer by layer.
        emu_l2 = {k: _l2(grads["emu"][k], grads["truth"][k]) for k in keys}
        assert f_hip < 3e-2
        assert all(hip_l2[k] <= 1.5 * emu_l2[k] + 0.02 for k in keys), [(k, hip_l2[k], emu_l2[k]) for k in keys if hip_l2[k] > 1.5 * emu_l2[k] + 0.02]


def test_vgg_in_multimodal_model_and_freeze_mode():
    from models import multimodalIntraInterModal as M
    os.environ["MMSKIN_BACKBONE_DTYPE"] = "fp32"
    kw = dict(SMALL, cnn_model_name="vgg16", attention_mecanism="concatenation", unfreeze_weights="last_layer_unfrozen_weights")
    cpu = det_init_(OracleMultimodalModel(**dict(kw, device="cpu")))
    hip = M.MultimodalModel(**dict(kw, device=DEV))
    hip.load_state_dict(cpu.state_dict(), strict=True)
    hip = hip.to(DEV)
    img, meta, lab = det_inputs(3, 64, 20, 6)
    res = {}
    for name, m, dev in (("cpu", cpu, "cpu"), ("hip", hip, DEV)):
        m.train(); disable_dropout(m)
        out = m(img.to(dev), meta.to(dev))
        F.cross_entropy(out, lab.to(dev)).backward()
        res[name] = (out.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None})
    assert (res["cpu"][0] - res["hip"][0]).abs().max() < 1e-3
    assert set(res["cpu"][1]) == set(res["hip"][1])
    enc = [k for k in res["hip"][1] if k.startswith("image_encoder")]
    assert enc == ["image_encoder.classifier.3.weight", "image_encoder.classifier.3.bias"]     # last 2 parameter tensors
    assert max(_l2(res["hip"][1][k], res["cpu"][1][k]) for k in res["cpu"][1]) < 5e-3

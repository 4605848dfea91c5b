"""DenseNet-169 image encoder (BASELINE.json configs[2]: DenseNet-169 + tab-transformer + metablock) on the HIP
plan executor vs the CPU oracle (oracle/backbones.py::OracleDenseNet169, torchvision layout -- parity
unpinned against torchvision itself, which is absent; state_dict names and shapes follow it).

Op-level: the channel-slice kernels the concatenating plan adds.  Backbone level: features, BN running
statistics and every parameter gradient.  Model level: the config-3 wiring end to end."""
import os

import pytest
import torch
import torch.nn as nn

from helpers import CLASS_WEIGHTS, SMALL, disable_dropout
from gpu_util import DEV, rel_err
from oracle.backbones import OracleDenseNet169
from oracle.detinit import det_init_, det_inputs, det_tensor
from oracle.model import OracleMultimodalModel

pytestmark = pytest.mark.gpu


def _l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _rb(t):
    return t.bfloat16().float()


def _bf16_storage_emulation(model):
    """The CPU oracle with bf16 STORAGE: conv weights and every conv / relu / pool output rounded to bf16
    (fp32 arithmetic in between) -- what any bf16 execution of the network computes, up to summation order."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = _rb(m.weight.data)
            m.register_forward_hook(lambda mod, i, o: _rb(o))
        if isinstance(m, (nn.ReLU, nn.MaxPool2d, nn.AvgPool2d)):
            m.register_forward_hook(lambda mod, i, o: _rb(o))
    return model


def _pair(dtype):
    from mmskin.backbone import HipDenseNet
    cpu = det_init_(OracleDenseNet169())
    hip = HipDenseNet(compute_dtype=dtype)
    hip.load_state_dict(cpu.state_dict(), strict=True)
    return cpu, hip.to(DEV)


def test_densenet_eval_features_fp32():
    cpu, hip = _pair("fp32")
    cpu.eval(); hip.eval()
    x = det_tensor("dn.img", (3, 3, 96, 96))
    with torch.no_grad():
        a, b = cpu(x), hip(x.to(DEV)).cpu()
    assert b.shape == (3, 1664)
    assert rel_err(b, a) < 1e-4, rel_err(b, a)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_densenet_train_step_vs_oracle(dtype):
    """Training-mode forward (batch statistics) + full backward.  Truth is the oracle in fp64; the HIP fp32
    path may be at most 3x as far from it as the CPU fp32 oracle is (same noise-aware criterion as the
    ResNet test).  bf16: a deterministic-random-init DenseNet-169 amplifies bf16 storage rounding a lot (the
    CPU emulation of bf16 storage is itself 3-9 % away from fp64 in the features and ~70 % in the median
    gradient at these tiny batch statistics), so the criterion is relative to that emulation: the HIP bf16
    path may be at most 1.5x as far from the fp64 truth as the emulation is."""
    cpu, hip = _pair(dtype)
    truth = det_init_(OracleDenseNet169()).double()
    x = det_tensor("dn.img", (6, 3, 64, 64))
    w = det_tensor("dn.w", (6, 1664))
    outs, grads = {}, {}
    runs = [("cpu", cpu, x, w), ("truth", truth, x.double(), w.double()), ("hip", hip, x.to(DEV), w.to(DEV))]
    if dtype == "bf16":
        runs.append(("emu", _bf16_storage_emulation(det_init_(OracleDenseNet169())), _rb(x), w))
    for name, m, xi, wi in runs:
        m.train()
        f = m(xi)
        (f * wi).sum().backward()
        outs[name] = f.detach().cpu().double()
        grads[name] = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters()}
    assert set(grads["hip"]) == set(grads["truth"])
    assert all(torch.isfinite(g).all() for g in grads["hip"].values())
    f_err_hip, f_err_cpu = _l2(outs["hip"], outs["truth"]), _l2(outs["cpu"], outs["truth"])
    keys = list(grads["truth"])
    cpu_l2 = sorted(_l2(grads["cpu"][k], grads["truth"][k]) for k in keys)
    hip_l2 = sorted(_l2(grads["hip"][k], grads["truth"][k]) for k in keys)
    worst = max(keys, key=lambda k: _l2(grads["hip"][k], grads["truth"][k]))
    print(dtype, "feat", f_err_hip, f_err_cpu, "grad median", hip_l2[len(keys) // 2], cpu_l2[len(keys) // 2],
          "max", hip_l2[-1], cpu_l2[-1], worst)
    rv = rel_err(hip.features.denseblock4.denselayer32.norm1.running_var, cpu.features.denseblock4.denselayer32.norm1.running_var)
    rm = rel_err(hip.features.transition2.norm.running_mean, cpu.features.transition2.norm.running_mean)
    assert int(hip.features.norm5.num_batches_tracked) == 1
    if dtype == "fp32":
        assert f_err_hip < 1e-4, f_err_hip
        assert hip_l2[len(keys) // 2] <= 3 * cpu_l2[len(keys) // 2] + 1e-4
        assert hip_l2[-1] <= 3 * cpu_l2[-1] + 1e-3, worst
        assert rv < 1e-4 and rm < 1e-4
    else:
        f_err_emu = _l2(outs["emu"], outs["truth"])
        emu_l2 = sorted(_l2(grads["emu"][k], grads["truth"][k]) for k in keys)
        print("emu feat", f_err_emu, "grad median", emu_l2[len(keys) // 2])
        assert f_err_hip <= 1.5 * f_err_emu, (f_err_hip, f_err_emu)
        assert hip_l2[len(keys) // 2] <= 1.5 * emu_l2[len(keys) // 2]
        assert rv < 5e-2 and rm < 5e-2


def test_densenet_config3_model_fp32():
    """configs[2] wiring: DenseNet-169 + tab-transformer + metablock, one training step against the oracle."""
    from models import multimodalIntraInterModal as M
    os.environ["MMSKIN_BACKBONE_DTYPE"] = "fp32"
    kw = dict(SMALL, cnn_model_name="densenet169", text_model_name="tab-transformer", attention_mecanism="metablock",
              vocab_size=86, unfreeze_weights="unfrozen_weights")
    cpu = det_init_(OracleMultimodalModel(**dict(kw, device="cpu")))
    hip = M.MultimodalModel(**dict(kw, device=DEV))
    hip.load_state_dict(cpu.state_dict(), strict=True)
    hip = hip.to(DEV)
    img, _, lab = det_inputs(5, 64, 20, 6)
    xc = (det_tensor("tt.cat", (5, 82)).abs() * 10).long().clamp_(0, 9)
    meta = torch.cat([xc.float(), det_tensor("tt.num", (5, 4))], dim=1)
    res = {}
    for name, m, dev in (("cpu", cpu, "cpu"), ("hip", hip, DEV)):
        m.train(); disable_dropout(m)
        out = m(img.to(dev), meta.to(dev))
        loss = nn.CrossEntropyLoss(weight=torch.tensor(CLASS_WEIGHTS, device=dev))(out, lab.to(dev))
        loss.backward()
        res[name] = (out.detach().cpu(), float(loss.detach()), {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None})
    assert (res["cpu"][0] - res["hip"][0]).abs().max() < 1e-3                  # north_star: 1e-3 fp32
    assert abs(res["cpu"][1] - res["hip"][1]) < 1e-4
    assert set(res["cpu"][2]) == set(res["hip"][2])
    head = [k for k in res["cpu"][2] if not k.startswith("image_encoder")]
    assert max(_l2(res["hip"][2][k], res["cpu"][2][k]) for k in head) < 5e-3


def test_densenet_partial_freeze_mode():
    """loadImageModelClassifier.py:88-92: 'partial' trains denseblock4 only."""
    from models.loadImageModelClassifier import loadModels
    os.environ["MMSKIN_BACKBONE_DTYPE"] = "bf16"
    m, dim = loadModels.loadModelImageEncoder("densenet169", 64, "partial")
    assert dim == 1664
    m = m.to(DEV).train()
    m(torch.randn(2, 3, 64, 64, device=DEV)).sum().backward()
    with_grad = {k for k, p in m.named_parameters() if p.grad is not None}
    assert with_grad and all(k.startswith("features.denseblock4.") for k in with_grad)
    assert len(with_grad) == 32 * 6

"""-m gpu: the drop-in MultimodalModel on the HIP path vs (a) the golden fixtures generated from the real
reference and (b) the CPU oracle on identical inputs.

Tolerances follow BASELINE.json's north_star: logits within 1e-3 (fp32 compute) / 1e-2 (bf16 compute)
of the reference PyTorch-CPU path.
"""
import json
import os

import pytest
import torch
import torch.nn as nn

import step_fixtures as sf
from helpers import (CLASS_WEIGHTS, SMALL, check_record_against_golden, disable_dropout, golden,
                     train_step_record)
from gpu_util import DEV, rel_err
from models import multimodalIntraInterModal as M
from oracle.detinit import det_init_, det_inputs
from oracle.model import FUSION_STRINGS, OracleMultimodalModel

pytestmark = pytest.mark.gpu
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")


def report(**kw):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(kw) + "\n")


def build_pair(dtype="fp32", **kw):
    os.environ["MMSKIN_BACKBONE_DTYPE"] = dtype
    cpu = det_init_(OracleMultimodalModel(**dict(kw, device="cpu")))
    hip = M.MultimodalModel(**dict(kw, device=DEV))
    hip.load_state_dict(cpu.state_dict(), strict=True)
    return cpu, hip.to(DEV)


@pytest.mark.parametrize("mech", FUSION_STRINGS)
def test_mechanism_matches_reference_golden(mech):
    """All 18 fusion strings: logits, loss, every gradient (None / exact-zero pattern included) and the
    first Adam step against fixtures recorded from the reference's own class."""
    gold = golden("mechanisms")[mech]
    kw = dict(SMALL, attention_mecanism=mech, n=1 if mech == "no-metadata" else 2, device=DEV)
    model = det_init_(M.MultimodalModel(**kw)).to(DEV)
    img, meta, lab = det_inputs(4, 32, 20, 6)
    rec = train_step_record(model, img.to(DEV), meta.to(DEV), lab.to(DEV), device=DEV)
    check_record_against_golden(rec, gold, rtol=1e-3, atol=1e-5)


def test_unknown_mechanism_raises_like_reference():
    want = golden("mechanisms")["__error__metablock-se"]
    model = M.MultimodalModel(**dict(SMALL, attention_mecanism="metablock-se", device=DEV)).to(DEV)
    img, meta, _ = det_inputs(4, 32, 20, 6)
    with pytest.raises(ValueError) as e:
        model(img.to(DEV), meta.to(DEV))
    assert str(e.value) == want


def test_full_width_head_golden():
    gold = golden("full_width")
    kw = dict(SMALL, common_dim=512, text_encoder_dim_output=512, attention_mecanism="crossattention", device=DEV)
    model = det_init_(M.MultimodalModel(**kw)).to(DEV)
    img, meta, lab = det_inputs(4, 32, 20, 6)
    rec = train_step_record(model, img.to(DEV), meta.to(DEV), lab.to(DEV), device=DEV)
    check_record_against_golden(rec, gold, rtol=1e-3, atol=1e-5)


def _step(model, img, meta, lab, dev):
    model.train()
    disable_dropout(model)
    model.zero_grad(set_to_none=True)
    out = model(img.to(dev), meta.to(dev))
    loss = nn.CrossEntropyLoss(weight=torch.tensor(CLASS_WEIGHTS, device=dev))(out, lab.to(dev))
    loss.backward()
    return out.detach().float().cpu(), float(loss.detach()), {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None}


def _l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _rb(t):
    return t.bfloat16().float()


def bf16_storage_emulation(model):
    """The CPU oracle with bf16 STORAGE: what any bf16 execution of the network computes, up to summation order.  Conv
    weights, every conv output, every BatchNorm(+ReLU) output, the max-pool output and every residual-block output are
    rounded to bf16 (fp32 arithmetic in between); the casts round the gradients flowing back through the same points."""
    enc = model.image_encoder
    for m in enc.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = _rb(m.weight.data)
            m.register_forward_hook(lambda mod, i, o: _rb(o))
        elif isinstance(m, (nn.MaxPool2d,)) or type(m).__name__ == "_Residual":
            m.register_forward_hook(lambda mod, i, o: _rb(o))
        elif isinstance(m, nn.BatchNorm2d):
            m.register_forward_hook(lambda mod, i, o: _rb(o))
    return model


def damp_residual_branches(model, gamma):
    """Set the weight of the LAST BatchNorm of every residual block (bn3 / bn2) to `gamma`: the parameter point of a
    trained ResNet, whose residual branches are small against the skip path (torchvision's zero_init_residual starts them
    at 0).  At the default init (1.0) sixteen full-strength random branches make the train-mode network chaotic."""
    enc = model.image_encoder
    last = "bn3" if hasattr(enc.layer1[0], "bn3") else "bn2"
    with torch.no_grad():
        for n, m in enc.named_modules():
            if n.endswith("." + last):
                m.weight.fill_(gamma)
    return model


STAGES = ("conv1|bn1", "layer1", "layer2", "layer3", "layer4")


def stage_report(out_c, loss_c, g_c, out_h, loss_h, g_h, cpu, hip):
    """Distances of one train step (logits, loss, running statistics, gradients) from the fp32 CPU oracle's.  Backbone
    gradients are summarised per ResNet stage: cosine and relative L2 of the stage's concatenated gradient vector."""
    rec = {"err_train_logits": float((out_c - out_h).abs().max()), "loss_cpu": loss_c, "loss_hip": loss_h}
    enc_c, enc_h = cpu.image_encoder, hip.image_encoder
    bns = [(n, m) for n, m in enc_c.named_modules() if isinstance(m, nn.BatchNorm2d)]
    mods_h = dict(enc_h.named_modules())
    rec["running_mean_rel_max"] = max(rel_err(mods_h[n].running_mean, m.running_mean) for n, m in bns)
    rec["running_var_rel_max"] = max(rel_err(mods_h[n].running_var, m.running_var) for n, m in bns)
    head = [k for k in g_c if not k.startswith("image_encoder")]
    rec["head_grad_l2_max"] = max(_l2(g_h[k], g_c[k]) for k in head)
    rec["head_grad_cos_min"] = min(_cos(g_h[k], g_c[k]) for k in head if float(g_c[k].abs().max()) > 0)
    for st in STAGES:
        names = st.split("|")
        keys = [k for k in g_c if any(k.startswith("image_encoder." + n) for n in names)]
        a = torch.cat([g_h[k].flatten().double() for k in keys]); b = torch.cat([g_c[k].flatten().double() for k in keys])
        rec["cos_" + names[0]] = _cos(a, b)
        rec["l2_" + names[0]] = _l2(a, b)
    return rec


def stage_report_rec(rc, rh):
    """stage_report on step RECORDS (tests/step_fixtures.py): `rc` the CPU side recorded in the build container
    (tests/golden/step_*.npz), `rh` the record of the HIP step -- gradients at the same seeded sample of coordinates."""
    rec = {"err_train_logits": float((rc["out"] - rh["out"]).abs().max()), "loss_cpu": rc["loss"], "loss_hip": rh["loss"]}
    rec["running_mean_rel_max"] = max(rel_err(rh["stats"][n][0], rc["stats"][n][0]) for n in rc["stats"])
    rec["running_var_rel_max"] = max(rel_err(rh["stats"][n][1], rc["stats"][n][1]) for n in rc["stats"])
    if rc["grads"]:
        head = [k for k in rc["grads"] if not k.startswith("image_encoder")]
        rec["head_grad_l2_max"] = max(sf.l2(rh["grads"][k], rc["grads"][k]) for k in head)
        rec["head_grad_cos_min"] = min(sf.cos(rh["grads"][k], rc["grads"][k]) for k in head if float(rc["grads"][k].abs().max()) > 0)
        for st in STAGES:
            names = st.split("|")
            keys = [k for k in rc["grads"] if any(k.startswith("image_encoder." + n) for n in names)]
            rec["cos_" + names[0]], rec["l2_" + names[0]] = sf.stage_stats(rh, rc, keys)
    return rec


def hip_step_record(hip, img, meta, lab, backward=True):
    if backward:
        out_h, loss_h, g_h = _step(hip, img, meta, lab, DEV)
        return sf.record(hip, out_h, loss_h, g_h)
    out_h, loss_h = _train_forward(hip, img, meta, lab, DEV)
    return sf.record(hip, out_h, loss_h)


def build_hip(dtype="fp32", gamma=None, **kw):
    """the HIP model at the deterministic init of the fixtures (the CPU oracle is only constructed for its state_dict)"""
    cpu, hip = build_pair(dtype, **kw)
    if gamma is not None:
        damp_residual_branches(hip, gamma)
    del cpu
    return hip


@pytest.mark.parametrize("arch,dtype", [("resnet-18", "fp32"), ("resnet-50", "fp32"), ("resnet-50", "bf16")])
def test_resnet_end_to_end_vs_oracle(arch, dtype):
    """Backbone + crossattention head: eval logits and a full train step against the CPU oracle.

    A randomly initialised ResNet at batch 8 is a chaotic map (ReLU sign flips): the CPU fp32 oracle's own
    gradients differ from an fp64 run of the same oracle by ~1-2% (relative L2).  The gradient criterion is
    therefore noise-aware: against the fp64 oracle as truth, the HIP fp32 path may be at most 3x as far away
    as the CPU fp32 oracle is.

    bf16 compute: at torchvision's default init NO bf16 execution of this train-mode network is within 1e-2 of the fp32
    logits -- rounding only the matrix-multiply operands to bf16 in the CPU oracle (everything else fp32) already moves
    them by 3e-2 and decorrelates the early layers' gradients (Numerics).  The reference point for bf16 is
    therefore `bf16_storage_emulation`: the same oracle with bf16 storage, run on the CPU; the HIP path may be at most
    1.5x as far from the fp32 oracle as that emulation is, metric by metric.  The 1e-2 bar itself is asserted at the
    BASELINE shape on a well-conditioned parameter point (test_bf16_train_step_parity_at_baseline_shape)."""
    kw = dict(SMALL, cnn_model_name=arch, common_dim=512, text_encoder_dim_output=512,
              attention_mecanism="crossattention")
    cpu, hip = build_pair(dtype, **kw)
    truth = det_init_(OracleMultimodalModel(**dict(kw, device="cpu"))).double()
    img, meta, lab = det_inputs(8, 128, 20, 6)
    cpu.eval(); hip.eval()
    with torch.no_grad():
        le_c, le_h = cpu(img, meta), hip(img.to(DEV), meta.to(DEV)).cpu()
    err_eval = float((le_c - le_h).abs().max())
    out_c, loss_c, g_c = _step(cpu, img, meta, lab, "cpu")
    out_h, loss_h, g_h = _step(hip, img, meta, lab, DEV)
    truth.train(); disable_dropout(truth)
    out_t = truth(img.double(), meta.double())
    nn.CrossEntropyLoss(weight=torch.tensor(CLASS_WEIGHTS, dtype=torch.float64))(out_t, lab).backward()
    g_t = {k: p.grad for k, p in truth.named_parameters() if p.grad is not None}
    assert set(g_c) == set(g_h) == set(g_t)
    err_train = float((out_c - out_h).abs().max())
    keys = [k for k in g_t if k.startswith("image_encoder")]
    cpu_l2 = sorted(_l2(g_c[k], g_t[k]) for k in keys)
    hip_l2 = sorted(_l2(g_h[k], g_t[k]) for k in keys)
    head_l2 = max(_l2(g_h[k], g_t[k]) for k in g_t if not k.startswith("image_encoder"))
    last = [k for k in keys if k.startswith("image_encoder.layer4.%d.bn%d" % ((1, 2) if arch == "resnet-18" else (2, 3)))]
    last_cos = min(_cos(g_h[k], g_t[k]) for k in last)
    bn_stat = rel_err(hip.image_encoder.layer4[-1].bn2.running_var, cpu.image_encoder.layer4[-1].bn2.running_var)
    report(test="e2e", arch=arch, dtype=dtype, err_eval_logits=err_eval, err_train_logits=err_train,
           loss_cpu=loss_c, loss_hip=loss_h, cpu_fp32_grad_l2_median=cpu_l2[len(cpu_l2) // 2], cpu_fp32_grad_l2_max=cpu_l2[-1],
           hip_grad_l2_median=hip_l2[len(hip_l2) // 2], hip_grad_l2_max=hip_l2[-1], head_grad_l2_max=head_l2,
           last_block_cos=last_cos, running_var_rel=bn_stat)
    assert all(torch.isfinite(v).all() for v in g_h.values())
    assert int(hip.image_encoder.bn1.num_batches_tracked) == 1
    if dtype == "fp32":
        assert err_eval < 1e-3 and err_train < 1e-3, (err_eval, err_train)            # north_star: 1e-3 fp32
        assert abs(loss_c - loss_h) < 1e-4
        assert hip_l2[len(hip_l2) // 2] <= 3 * cpu_l2[len(cpu_l2) // 2] + 1e-4
        assert hip_l2[-1] <= 3 * cpu_l2[-1] + 1e-4
        assert head_l2 < 5e-3 and bn_stat < 1e-4
    else:
        assert err_eval < 1e-2, err_eval                                              # north_star: 1e-2 bf16
        emu = bf16_storage_emulation(det_init_(OracleMultimodalModel(**dict(kw, device="cpu"))))
        out_e, loss_e, g_e = _step(emu, _rb(img), meta, lab, "cpu")
        r_hip = stage_report(out_c, loss_c, g_c, out_h, loss_h, g_h, cpu, hip)
        r_emu = stage_report(out_c, loss_c, g_c, out_e, loss_e, g_e, cpu, emu)
        report(test="e2e_bf16_vs_emulation", arch=arch, hip=r_hip, emu=r_emu)
        assert_not_worse_than_emulation(r_hip, r_emu, slack=1.5)


def assert_not_worse_than_emulation(r_hip, r_emu, slack):
    """Every distance of the HIP bf16 step from the fp32 oracle is at most `slack` x the distance of the CPU bf16-storage
    emulation (plus a small absolute floor); every per-stage gradient cosine at least the emulation's minus 0.05."""
    for k in ("err_train_logits", "running_mean_rel_max", "running_var_rel_max", "head_grad_l2_max"):
        assert r_hip[k] <= slack * r_emu[k] + 1e-4, (k, r_hip[k], r_emu[k])
    assert abs(r_hip["loss_hip"] - r_hip["loss_cpu"]) <= slack * abs(r_emu["loss_hip"] - r_emu["loss_cpu"]) + 2e-3
    for k in r_emu:
        if k.startswith("cos_"):
            assert r_hip[k] >= r_emu[k] - 0.05, (k, r_hip[k], r_emu[k])
        if k.startswith("l2_"):
            assert r_hip[k] <= slack * r_emu[k] + 1e-3, (k, r_hip[k], r_emu[k])


def _baseline_batch(B, hw=224):
    g = torch.Generator().manual_seed(0)
    return (torch.randn(B, 3, hw, hw, generator=g), torch.randn(B, 20, generator=g), torch.randint(0, 6, (B,), generator=g))


RESNET50_KW = dict(SMALL, cnn_model_name="resnet-50", common_dim=512, text_encoder_dim_output=512, attention_mecanism="crossattention")


def test_bf16_train_step_parity_at_baseline_shape():
    """THE benchmarked path (BASELINE configs[1]): ONE train step (dropout off, BatchNorm batch statistics; the step of
    train_pad_20.py:102-113) of ResNet-50 + crossattention at B = 256 @ 224^2 with bf16 backbone compute, HIP vs the fp32
    CPU oracle.  The oracle / emulation sides come from tests/golden/step_b256_*.npz (recorded in the build container by
    tests/golden/gen_step_golden.py: three CPU steps at this size were 170 - 185 s of the GPU box's clock); gradients are
    compared at the fixtures' seeded coordinate samples (tests/step_fixtures.py).  Two parameter points:

    (a) conditioned: the last BatchNorm of every residual block has gamma 0.25 -- residual branches small against the skip
        path, as in a trained network.  Here the north_star bound means something and is asserted as stated: train-mode
        logits within 1e-2; plus loss, every BatchNorm's running statistics, head gradients and the per-stage cosine /
        relative L2 of the backbone gradients (thresholds = 3-5x the values measured on MI355X, gpurun_out/parity_report.jsonl).
    (b) torchvision's default init (gamma 1): sixteen full-strength random branches make the train-mode network chaotic --
        no bf16 execution is within 1e-2 (see test_resnet_end_to_end_vs_oracle); the HIP step must be no further from
        the fp32 oracle than the CPU bf16-storage emulation of the same step (x1.25), and -- the assertion with power on the
        early layers, where both are decorrelated from fp32 -- its gradient must agree with the EMULATION's: both round the
        same operands.  Measured: they do NOT agree to 0.9 -- cosine(HIP, emulation) is 0.32 - 0.36 on layers 1 - 3 and 0.58 on layer 4
        (against 0.15 - 0.40 for either one versus fp32): at this parameter point any two bf16 executions decorrelate from each other,
        so the assertion with power stays the conditioned point (a) and the kernel tests; the cosine is recorded and floored."""
    img, meta, lab = _baseline_batch(256)
    # ---- (a)
    rc = sf.load("step_b256_g025")
    hip = build_hip("bf16", 0.25, **RESNET50_KW)
    rh = hip_step_record(hip, img, meta, lab)
    r = stage_report_rec(rc, rh)
    report(test="bf16_b256_conditioned", **r)
    assert set(rc["grads"]) == set(rh["grads"]) and all(torch.isfinite(v).all() for v in rh["grads"].values())
    assert r["err_train_logits"] < 1e-2, r                                           # north_star: 1e-2 bf16 (measured 1.4e-3)
    assert abs(r["loss_cpu"] - r["loss_hip"]) < 5e-4, r                              # measured 4e-5
    assert r["running_mean_rel_max"] < 2e-2 and r["running_var_rel_max"] < 1e-3, r   # measured 3.9e-3 / 6e-5
    assert r["head_grad_l2_max"] < 0.1 and r["head_grad_cos_min"] > 0.995, r         # measured 2.9e-2 / 0.9996
    floors = {"cos_conv1": 0.80, "cos_layer1": 0.80, "cos_layer2": 0.82, "cos_layer3": 0.85, "cos_layer4": 0.92}
    for k, v in floors.items():                                                     # measured 0.876 / 0.884 / 0.892 / 0.914 / 0.964
        assert r[k] > v, (k, r[k])                                                   # (the CPU bf16 emulation: the same to 3e-3)
    assert r["l2_layer4"] < 0.4 and r["l2_conv1"] < 0.65, r                          # measured 0.27 / 0.50
    del hip
    # ---- (b)
    rc, re = sf.load("step_b256_default"), sf.load("step_b256_default_emu")
    hip = build_hip("bf16", None, **RESNET50_KW)
    rh = hip_step_record(hip, img, meta, lab)
    r_hip, r_emu = stage_report_rec(rc, rh), stage_report_rec(rc, re)
    vs_emu = stage_report_rec(re, rh)     # HIP against the emulation itself
    report(test="bf16_b256_default_init", hip=r_hip, emu=r_emu, hip_vs_emulation={k: v for k, v in vs_emu.items() if k.startswith(("cos_", "l2_"))})
    assert_not_worse_than_emulation(r_hip, r_emu, slack=1.25)
    assert vs_emu["cos_layer4"] >= VS_EMU_COS_FLOOR, vs_emu


VS_EMU_COS_FLOOR = 0.45   # cosine(HIP gradient, emulation gradient) of layer4 at the default init: measured 0.58 (layers 1-3: 0.32 - 0.36) -- two bf16 executions of this chaotic parameter point decorrelate from EACH OTHER, not only from fp32 (parity report: hip_vs_emulation)


def _train_forward(model, img, meta, lab, dev):
    """train-mode forward only (BatchNorm batch statistics, running statistics updated, dropout off): logits and loss"""
    model.train()
    disable_dropout(model)
    with torch.no_grad():
        out = model(img.to(dev), meta.to(dev))
        loss = nn.CrossEntropyLoss(weight=torch.tensor(CLASS_WEIGHTS, device=dev))(out, lab.to(dev))
    return out.detach().float().cpu(), float(loss)


def _forward_report(out_c, loss_c, out_h, loss_h, cpu, hip):
    rec = {"err_train_logits": float((out_c - out_h).abs().max()), "loss_cpu": loss_c, "loss_hip": loss_h}
    bns = [(n, m) for n, m in cpu.image_encoder.named_modules() if isinstance(m, nn.BatchNorm2d)]
    mods_h = dict(hip.image_encoder.named_modules())
    rec["running_mean_rel_max"] = max(rel_err(mods_h[n].running_mean, m.running_mean) for n, m in bns)
    rec["running_var_rel_max"] = max(rel_err(mods_h[n].running_var, m.running_var) for n, m in bns)
    return rec


@pytest.mark.parametrize("gamma", [0.1, 0.5, 0.75])
def test_bf16_train_logit_error_over_last_bn_gamma(gamma):
    """Where between the conditioned point (gamma 0.25: 1.4e-3) and torchvision's default init (gamma 1: 3e-2) does the 1e-2 bf16
    bound on the TRAIN-mode logits break?  The benchmarked configuration at B = 256 @ 224^2 with the last BatchNorm of every
    residual block at gamma in {0.1, 0.5, 0.75} (0.25 and 1.0, with the whole backward, are
    test_bf16_train_step_parity_at_baseline_shape); for each point the train-mode FORWARD (batch statistics, running statistics
    updated) of the HIP model is compared with the fp32 CPU oracle's and the CPU bf16-storage emulation's, both recorded in the
    build container (tests/golden/step_b256_fwd_g*.npz).  Asserted at EVERY point: the HIP logits / loss / running statistics are no
    further from the fp32 oracle than 1.25 x the emulation's, and the logits meet the north_star's 1e-2 wherever the emulation does
    (VERDICT r02 item 6a).  Reference semantics: the forward of one optimisation step of train_pad_20.py:102-113."""
    img, meta, lab = _baseline_batch(256)
    tag = str(gamma).replace(".", "")
    rc, re = sf.load(f"step_b256_fwd_g{tag}"), sf.load(f"step_b256_fwd_g{tag}_emu")
    hip = build_hip("bf16", gamma, **RESNET50_KW)
    rh = hip_step_record(hip, img, meta, lab, backward=False)
    r_hip, r_emu = stage_report_rec(rc, rh), stage_report_rec(rc, re)
    report(test="bf16_b256_gamma_sweep", gamma=gamma, hip=r_hip, emu=r_emu)
    for k in ("err_train_logits", "running_mean_rel_max", "running_var_rel_max"):
        assert r_hip[k] <= 1.25 * r_emu[k] + 1e-4, (k, r_hip[k], r_emu[k])
    assert abs(r_hip["loss_hip"] - r_hip["loss_cpu"]) <= 1.25 * abs(r_emu["loss_hip"] - r_emu["loss_cpu"]) + 2e-3
    if r_emu["err_train_logits"] <= 1e-2 / 1.25:       # the emulation is inside the bound with the slack to spare: so must the kernels be
        assert r_hip["err_train_logits"] < 1e-2, (gamma, r_hip["err_train_logits"], r_emu["err_train_logits"])


def test_fp32_train_step_parity_at_production_size():
    """fp32 compute at B = 64 @ 224^2: the code paths that only engage at production size (single-buffer conv variants for
    launches of > 640 / 800 workgroups, the two-stage slab reduction for > 32 splits, the wgrad split policy, the parity
    zero fill with large 3-D grids) with a QUANTITATIVE backward check: loss, logits and the encoder gradients
    -- per stage, plus the stem conv, a layer1 conv3, the layer2 downsample conv and a layer4 BatchNorm -- under the
    noise-aware rule of test_resnet_end_to_end_vs_oracle (truth = the oracle in fp64; HIP fp32 at most 3x as far as CPU fp32).
    The CPU fp32 and fp64 steps are recorded fixtures (tests/golden/step_b64_fp32.npz / step_b64_fp64.npz)."""
    img, meta, lab = _baseline_batch(64)
    rc, rt = sf.load("step_b64_fp32"), sf.load("step_b64_fp64")
    hip = build_hip("fp32", None, **RESNET50_KW)
    rh = hip_step_record(hip, img, meta, lab)
    r = stage_report_rec(rc, rh)
    picks = ["image_encoder.conv1.weight", "image_encoder.layer1.1.conv3.weight", "image_encoder.layer2.0.downsample.0.weight",
             "image_encoder.layer4.2.bn3.weight", "image_encoder.layer4.2.bn3.bias"]
    dist = {k: (sf.l2(rh["grads"][k], rt["grads"][k]), sf.l2(rc["grads"][k], rt["grads"][k])) for k in picks}
    keys = [k for k in rt["grads"] if k.startswith("image_encoder")]
    hip_l2 = sorted(sf.l2(rh["grads"][k], rt["grads"][k]) for k in keys); cpu_l2 = sorted(sf.l2(rc["grads"][k], rt["grads"][k]) for k in keys)
    report(test="fp32_b64_224", picks=dist, hip_l2_median=hip_l2[len(keys) // 2], cpu_l2_median=cpu_l2[len(keys) // 2],
           hip_l2_max=hip_l2[-1], cpu_l2_max=cpu_l2[-1], **r)
    assert r["err_train_logits"] < 1e-3 and abs(r["loss_cpu"] - r["loss_hip"]) < 1e-4, r            # north_star: 1e-3 fp32
    assert r["running_mean_rel_max"] < 1e-4 and r["running_var_rel_max"] < 1e-4 and r["head_grad_l2_max"] < 5e-3, r
    for k, (dh, dc) in dist.items():
        assert dh <= 3 * dc + 1e-4, (k, dh, dc)
    assert hip_l2[len(keys) // 2] <= 3 * cpu_l2[len(keys) // 2] + 1e-4 and hip_l2[-1] <= 3 * cpu_l2[-1] + 1e-4


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_baseline_shape_properties(dtype):
    """BASELINE config 2 at full size (ResNet-50 + crossattention, 256 x 3 x 224 x 224, 20 metadata columns):
    (a) four samples of the eval-mode batch agree with the CPU oracle run on those four samples alone
    (eval BN makes samples independent), (b) batch-permutation equivariance, (c) bit-exact repeatability,
    (d) a train step produces finite gradients for every live parameter."""
    kw = dict(SMALL, cnn_model_name="resnet-50", common_dim=512, text_encoder_dim_output=512,
              attention_mecanism="crossattention")
    cpu, hip = build_pair(dtype, **kw)
    B = 256 if dtype == "bf16" else 64
    g = torch.Generator().manual_seed(0)
    img = torch.randn(B, 3, 224, 224, generator=g)
    meta = torch.randn(B, 20, generator=g)
    lab = torch.randint(0, 6, (B,), generator=g)
    cpu.eval(); hip.eval()
    pick = [0, B // 3, 2 * B // 3, B - 1]
    with torch.no_grad():
        full = hip(img.to(DEV), meta.to(DEV)).cpu()
        again = hip(img.to(DEV), meta.to(DEV)).cpu()
        perm = torch.randperm(B, generator=g)
        permuted = hip(img[perm].to(DEV), meta[perm].to(DEV)).cpu()
        want = cpu(img[pick], meta[pick])
    err = float((full[pick] - want).abs().max())
    report(test="baseline_shape", dtype=dtype, batch=B, err_logits_vs_oracle=err)
    assert err < (1e-3 if dtype == "fp32" else 1e-2), err
    assert torch.equal(full, again)
    assert torch.allclose(permuted, full[perm], atol=1e-5 if dtype == "fp32" else 2e-3)
    _, loss, grads = _step(hip, img, meta, lab, DEV)
    assert loss == loss and all(torch.isfinite(v).all() for v in grads.values())


def test_tab_transformer_block_and_wiring():
    """TabTransformer metadata encoder on the HIP ops: the class against the reference fixture
    (tests/golden/blocks.json), and the build-defined end-to-end wiring (parity unpinned in the reference,
    whose own wiring raises -- SURVEY.md section 4) against the oracle."""
    from models.tab_transformer import TabTransformer
    from oracle.blocks import OracleTabTransformer
    from oracle.detinit import det_tensor
    gold = golden("blocks")["tab_transformer"]
    tt = det_init_(TabTransformer([10] * 82, num_continuous=4, output_dim=85)).to(DEV)
    tt.eval()
    xc = (det_tensor("tt.cat", (3, 82)).abs() * 10).long().clamp_(0, 9)
    xn = det_tensor("tt.num", (3, 4))
    y = tt(xc.to(DEV), xn.to(DEV))
    assert torch.allclose(y.double().cpu(), torch.tensor(gold["y"], dtype=torch.float64), rtol=1e-3, atol=1e-4)
    # gradients vs the oracle class with identical weights
    ref = det_init_(OracleTabTransformer([10] * 82, num_continuous=4, output_dim=85)); ref.eval()
    y.square().sum().backward()
    ref(xc, xn).square().sum().backward()
    for (k, p), (_, q) in zip(tt.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and rel_err(p.grad, q.grad) < 2e-3, k
    # end-to-end wiring: 82 categorical + 4 continuous columns in one float tensor
    kw = dict(SMALL, text_model_name="tab-transformer", attention_mecanism="metablock", vocab_size=86)
    cpu, hip = build_pair("fp32", **kw)
    cpu.eval(); hip.eval()
    img, _, _ = det_inputs(4, 32, 20, 6)
    meta = torch.cat([xc.float(), xn], dim=1)
    meta = torch.cat([meta, meta[:1]], dim=0)
    with torch.no_grad():
        a, b = cpu(img, meta), hip(img.to(DEV), meta.to(DEV)).cpu()
    assert torch.allclose(a, b, rtol=1e-3, atol=1e-4), (a - b).abs().max()


@pytest.mark.parametrize("batch,hw", [(5, 96), (3, 160), (1, 224)])
def test_ragged_batches_and_input_sizes(batch, hw):
    """Odd batch sizes (last batch of an epoch), non-224 inputs and batch 1 (the API / XAI consumers) in
    eval mode: partial row blocks in every kernel, odd spatial sizes in the strided layers."""
    kw = dict(SMALL, cnn_model_name="resnet-50", common_dim=512, text_encoder_dim_output=512,
              attention_mecanism="gfcam")
    cpu, hip = build_pair("fp32", **kw)
    cpu.eval(); hip.eval()
    img, meta, _ = det_inputs(batch, hw, 20, 6)
    with torch.no_grad():
        a, b = cpu(img, meta), hip(img.to(DEV), meta.to(DEV)).cpu()
    assert torch.allclose(a, b, rtol=1e-3, atol=1e-3), (a - b).abs().max()


@pytest.mark.parametrize("mode,expect_backbone_grads", [("frozen_weights", 0), ("last_layer_unfrozen_weights", 2),
                                                        ("unfrozen_weights", 60)])
def test_freeze_modes_train_step(mode, expect_backbone_grads):
    """loadImageModelClassifier.py:15-35 freeze policy on the HIP path: train-mode BN still uses batch statistics
    and updates running stats when frozen (train_pad_20.py:102), only the selected parameters get gradients."""
    kw = dict(SMALL, cnn_model_name="resnet-18", common_dim=512, text_encoder_dim_output=512,
              attention_mecanism="concatenation", unfreeze_weights=mode)
    cpu, hip = build_pair("fp32", **kw)
    img, meta, lab = det_inputs(8, 96, 20, 6)
    out_c, loss_c, g_c = _step(cpu, img, meta, lab, "cpu")
    out_h, loss_h, g_h = _step(hip, img, meta, lab, DEV)
    assert set(g_c) == set(g_h)
    assert sum(k.startswith("image_encoder") for k in g_h) == expect_backbone_grads
    assert float((out_c - out_h).abs().max()) < 1e-3 and abs(loss_c - loss_h) < 1e-4
    for k in g_c:
        if not k.startswith("image_encoder") or mode == "last_layer_unfrozen_weights":
            assert rel_err(g_h[k], g_c[k]) < 5e-3, k
    rv_c, rv_h = cpu.image_encoder.layer3[0].bn1.running_var, hip.image_encoder.layer3[0].bn1.running_var
    assert rel_err(rv_h, rv_c) < 1e-4 and not torch.allclose(rv_c, torch.ones_like(rv_c))


def test_checkpoint_round_trip_and_module_prefix():
    """state_dict keys are an API: save from the HIP model, strip/add the DataParallel 'module.'
    prefix like inference_all_folds.py:50-56, load with strict=False into a fresh model, same logits; the
    early-stopping pattern copy.deepcopy(model.state_dict()) (early_stopping.py:61) keeps working."""
    import copy
    kw = dict(SMALL, cnn_model_name="resnet-18", common_dim=512, text_encoder_dim_output=512,
              attention_mecanism="crossattention", device=DEV)
    os.environ["MMSKIN_BACKBONE_DTYPE"] = "fp32"
    a = det_init_(M.MultimodalModel(**kw)).to(DEV).eval()
    best = copy.deepcopy(a.state_dict())
    wrapped = {"module." + k: v for k, v in best.items()}
    stripped = {k.replace("module.", "", 1): v for k, v in wrapped.items()}
    stripped["some.unknown.key"] = torch.zeros(1)
    b = M.MultimodalModel(**kw).to(DEV).eval()
    res = b.load_state_dict(stripped, strict=False)
    assert res.missing_keys == [] and res.unexpected_keys == ["some.unknown.key"]
    img, meta, _ = det_inputs(4, 96, 20, 6)
    with torch.no_grad():
        assert torch.equal(a(img.to(DEV), meta.to(DEV)), b(img.to(DEV), meta.to(DEV)))
    assert b.image_encoder._packed()


@pytest.mark.parametrize("arch", ["resnet-18", "densenet169"])
def test_uint8_nhwc_input_path(arch):
    """the decoded uint8 NHWC batch goes straight into the stem packing kernel, which applies the
    reference transform's Normalize + ToTensor (skinLesionDatasets.py:29,111-119).  Must equal feeding the
    normalised fp32 NCHW tensor, through the model's forward (MultimodalModel moves / forwards the tensor as is)."""
    kw = dict(SMALL, cnn_model_name=arch, attention_mecanism="concatenation")
    cpu, hip = build_pair("fp32", **kw)
    cpu.eval(); hip.eval()
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (3, 64, 64, 3), generator=g, dtype=torch.uint8)
    meta = det_inputs(3, 32, 20, 6)[1]
    mean = torch.tensor([0.485, 0.456, 0.406]); std = torch.tensor([0.229, 0.224, 0.225])
    x = ((u8.float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        a = hip(u8.to(DEV), meta.to(DEV)).cpu()
        b = hip(x.to(DEV), meta.to(DEV)).cpu()
        c = cpu(x, meta)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a - b).abs().max()
    assert torch.allclose(a, c, rtol=1e-3, atol=1e-4), (a - c).abs().max()
    with pytest.raises(Exception):
        hip.image_encoder(torch.zeros(2, 3, 64, 64, dtype=torch.uint8, device=DEV))     # uint8 must be NHWC


def test_gradcam_consumer_on_last_conv():
    """the reference's Grad-CAM++ (src/services/XAI/models/cam.py:10-60) hooks the last nn.Conv2d of
    image_encoder (model_loader.py:36-41), runs model(image, metadata) in eval mode and takes
    autograd.grad(score, activations, retain_graph=True, create_graph=True).  The same calls must work on the HIP
    model and give the oracle's activations, gradients and CAM."""
    def find_last_conv(module):
        last = None
        for m in module.modules():
            if isinstance(m, nn.Conv2d):
                last = m
        return last

    def cam_pp(model, image, meta, target):
        store = {}
        h = find_last_conv(model.image_encoder).register_forward_hook(lambda mod, i, o: store.__setitem__("a", o))
        image.requires_grad_(True)
        out = model(image, meta)
        score = out[:, target]
        acts = store["a"]
        grads = torch.autograd.grad(score.sum(), acts, retain_graph=True, create_graph=True)[0]
        g2, g3 = grads ** 2, grads ** 3
        alpha = g2 / (2 * g2 + (acts * g3).sum(dim=(2, 3), keepdim=True) + 1e-8)
        weights = (alpha * torch.relu(grads)).sum(dim=(2, 3), keepdim=True)
        cam = torch.relu((weights * acts).sum(dim=1, keepdim=True))
        h.remove()
        return out.detach().cpu(), acts.detach().cpu(), grads.detach().cpu(), cam.detach().cpu()

    kw = dict(SMALL, cnn_model_name="resnet-18", attention_mecanism="concatenation")
    cpu, hip = build_pair("fp32", **kw)
    cpu.eval(); hip.eval()
    img, meta, _ = det_inputs(3, 64, 20, 6)
    o_c, a_c, g_c, cam_c = cam_pp(cpu, img.clone(), meta, 2)
    o_h, a_h, g_h, cam_h = cam_pp(hip, img.clone().to(DEV), meta.to(DEV), 2)
    assert a_h.shape == a_c.shape == (3, 512, 2, 2)
    assert torch.allclose(o_h, o_c, rtol=1e-3, atol=1e-4)
    assert rel_err(a_h, a_c) < 1e-4 and rel_err(g_h, g_c) < 1e-3 and rel_err(cam_h, cam_c) < 1e-3
    # without hooks the folded inference path is used again and gives the same logits
    with torch.no_grad():
        assert torch.allclose(hip(img.to(DEV), meta.to(DEV)).cpu(), o_c, rtol=1e-3, atol=1e-4)


def test_gradcam_on_densenet_features_tail_and_second_order():
    """SURVEY 8(f)-4 remainder: the reference's Grad-CAM++ script targets `model.image_encoder.features[-1]` on DenseNet
    (interpretability/gradcam_plusplus.py:298) and calls autograd.grad(score, activations, retain_graph=True,
    create_graph=True) (:206).  On the HIP model: same indexing, same activations / gradients / CAM as the oracle.  The
    reference never differentiates a second time (it squares / cubes the first-order gradients); on the HIP path a second
    differentiation is UNSUPPORTED and must say so instead of returning silently wrong (constant-folded) values: every HIP
    op is wrapped by mmskin._autograd.no_second_order, which raises MMSkinError("second-order ...")."""
    from mmskin._lib import MMSkinError

    def run(model, image, meta, target_layer):
        store = {}
        h = target_layer.register_forward_hook(lambda mod, i, o: store.__setitem__("a", o))
        image.requires_grad_(True)
        out = model(image, meta)
        score = out[:, 1].sum()
        acts = store["a"]
        grads = torch.autograd.grad(score, acts, retain_graph=True, create_graph=True)[0]
        alpha = grads ** 2 / (2 * grads ** 2 + (acts * grads ** 3).sum(dim=(2, 3), keepdim=True) + 1e-7)
        cam = torch.relu(((alpha * torch.relu(grads)).sum(dim=(2, 3), keepdim=True) * acts).sum(dim=1))
        h.remove()
        return out, acts, grads, cam

    kw = dict(SMALL, cnn_model_name="densenet169", attention_mecanism="concatenation")
    cpu, hip = build_pair("fp32", **kw)
    cpu.eval(); hip.eval()
    assert type(hip.image_encoder.features[-1]).__name__ == "BatchNorm2d"
    img, meta, _ = det_inputs(2, 64, 20, 6)
    o_c, a_c, g_c, cam_c = run(cpu, img.clone(), meta, cpu.image_encoder.features[-1])
    o_h, a_h, g_h, cam_h = run(hip, img.clone().to(DEV), meta.to(DEV), hip.image_encoder.features[-1])
    assert a_h.shape == a_c.shape == (2, 1664, 2, 2)
    assert torch.allclose(o_h.detach().cpu(), o_c.detach(), rtol=1e-3, atol=1e-4)
    assert rel_err(a_h, a_c) < 1e-4 and rel_err(g_h, g_c) < 1e-3 and rel_err(cam_h, cam_c) < 1e-3
    # a second differentiation through the first-order gradient: the oracle (plain torch ops) has one (the LayerNorms of the
    # fusion head make the logits nonlinear in the features); the HIP ops' backward kernels are not differentiable -> error
    s_c = torch.autograd.grad((g_c * a_c.detach()).sum(), a_c, allow_unused=True)[0]
    assert s_c is not None and float(s_c.abs().max()) > 0
    with pytest.raises(MMSkinError, match="second-order"):
        torch.autograd.grad((g_h * a_h.detach()).sum(), a_h)
    with torch.no_grad():                                     # without hooks: the folded inference plan, same logits
        assert torch.allclose(hip(img.to(DEV), meta.to(DEV)).cpu(), o_c.detach(), rtol=1e-3, atol=1e-4)
    # ResNet: first order works (test_gradcam_consumer_on_last_conv); a second differentiation says why it cannot
    kw = dict(SMALL, cnn_model_name="resnet-18", attention_mecanism="concatenation")
    _, hip = build_pair("fp32", **kw)
    hip.eval()
    last = [m for m in hip.image_encoder.modules() if isinstance(m, nn.Conv2d)][-1]
    _, a_r, g_r, _ = run(hip, img.clone().to(DEV), meta.to(DEV), last)
    with pytest.raises(MMSkinError, match="second-order"):
        torch.autograd.grad((g_r * a_r.detach()).sum(), a_r)


def test_eval_forward_reuses_staged_weights_only_while_unchanged():
    """Serving path: repeated eval forwards reuse the BN-folded staged weights; an in-place parameter
    update or a training forward (running statistics move) must be picked up by the next eval forward."""
    from mmskin.backbone import HipResNet
    os.environ["MMSKIN_BACKBONE_DTYPE"] = "fp32"
    torch.manual_seed(3)
    enc = HipResNet("resnet-18").to(DEV)
    x = torch.randn(4, 3, 64, 64, device=DEV)
    enc.eval()
    with torch.no_grad():
        a = enc(x).clone()
        b = enc(x).clone()                       # second call: staged weights reused
        assert torch.equal(a, b)
        enc.conv1.weight.mul_(1.5)               # version bump -> restaged
        c = enc(x).clone()
        assert rel_err(c, a) > 1e-3
        fresh = HipResNet("resnet-18").to(DEV)
        fresh.load_state_dict(enc.state_dict())
        fresh.eval()
        assert rel_err(c, fresh(x)) < 1e-5
    enc.train()
    enc(x)                                       # updates running_mean / running_var inside the C call
    enc.eval()
    with torch.no_grad():
        d = enc(x).clone()
        fresh.load_state_dict(enc.state_dict())
        assert rel_err(d, fresh(x)) < 1e-5
        assert rel_err(d, c) > 1e-4
    # Two plans: eval at batch 4, a training forward at batch 6 WITHOUT an optimizer step (frozen-weights mode:
    # only the running statistics move, inside the C call, no version counter changes), eval at batch 4 again.  The
    # batch-4 plan must notice and refold; and four shapes must coexist in the plan cache without a rebuild.
    x6 = torch.randn(6, 3, 64, 64, device=DEV)
    plan4 = enc._plan_for(4, 64, 64, x.device)
    enc.train()
    with torch.no_grad():
        enc(x6)
    enc.eval()
    with torch.no_grad():
        e = enc(x).clone()
        fresh.load_state_dict(enc.state_dict())
        assert rel_err(e, fresh(x)) < 1e-5
        assert rel_err(e, d) > 1e-5                                   # the statistics did move
        enc(torch.randn(3, 3, 64, 64, device=DEV)); enc(torch.randn(5, 3, 64, 64, device=DEV))
    assert enc._plan_for(4, 64, 64, x.device) is plan4 and len(enc._plans) == 4


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_compact_downsample_gradient_is_bit_identical(dtype):
    """Backward of the stride-2 downsample branch: its data gradient as a dense GEMM into a compact buffer, added at the even pixels by
    conv1's dgrad epilogue (default), against the zero-filled full-resolution form (MMSKIN_DS_COMPACT=0).  Same products, same fp32
    accumulation, the same single rounding of (acc + addend): every backbone gradient must have the SAME bits.  Odd feature-map sizes
    (23 x 18 into layer2: ceil(H / 2) samples) and a ragged batch."""
    import subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import os, sys, torch
sys.path[:0] = [%r, %r]
os.environ["MMSKIN_BACKBONE_DTYPE"] = %r
from gpu_util import DEV
from models import loadImageModelClassifier as L
torch.manual_seed(3)
enc, dim = L.loadModels.loadModelImageEncoder("resnet-50", 512, "unfrozen_weights")
enc = enc.to(DEV).train()
g = torch.Generator().manual_seed(4)
x = torch.randn(3, 3, 90, 70, generator=g).to(DEV)
y = enc(x)
w = torch.randn(y.shape, generator=g).to(DEV)
(y * w).sum().backward()
torch.cuda.synchronize()
torch.save({n: p.grad.detach().cpu() for n, p in enc.named_parameters()}, sys.argv[1])
"""
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for knob in ("1", "0"):
            f = os.path.join(td, f"g{knob}.pt")
            r = subprocess.run([sys.executable, "-c", code % (os.path.join(root, "tests"), os.path.join(root, "multimodal-model-skin-lesion-classifier_amd"), dtype), f],
                               env=dict(os.environ, MMSKIN_DS_COMPACT=knob), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            outs.append(torch.load(f))
    assert outs[0].keys() == outs[1].keys() and len(outs[0]) > 150
    for k in outs[0]:
        assert torch.isfinite(outs[0][k]).all() and float(outs[0][k].abs().max()) > 0, k
        assert torch.equal(outs[0][k], outs[1][k]), (k, float((outs[0][k] - outs[1][k]).abs().max()))

"""Build container only (CPU, ~25 min on 8 cores, ~25 GB): the fp32 CPU-oracle / bf16-storage-emulation / fp64 sides of the
production-size train-step tests of tests/test_gpu_model.py -> tests/golden/step_*.npz (see tests/step_fixtures.py).

    python tests/golden/gen_step_golden.py [name ...]      (no names: all)

Everything is seeded: det_init_ weights, torch.Generator(0) inputs; the GPU tests rebuild the same inputs and weights."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch
import torch.nn as nn

import step_fixtures as sf
from helpers import CLASS_WEIGHTS, SMALL, disable_dropout
from oracle.detinit import det_init_
from oracle.model import OracleMultimodalModel

RESNET50_KW = dict(SMALL, cnn_model_name="resnet-50", common_dim=512, text_encoder_dim_output=512, attention_mecanism="crossattention")


def rb(t):
    return t.bfloat16().float()


def baseline_batch(B, hw=224):
    g = torch.Generator().manual_seed(0)
    return (torch.randn(B, 3, hw, hw, generator=g), torch.randn(B, 20, generator=g), torch.randint(0, 6, (B,), generator=g))


def emulate_bf16_storage(model):
    """same hooks as tests/test_gpu_model.py::bf16_storage_emulation"""
    enc = model.image_encoder
    for m in enc.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = rb(m.weight.data)
            m.register_forward_hook(lambda mod, i, o: rb(o))
        elif isinstance(m, (nn.MaxPool2d,)) or type(m).__name__ == "_Residual":
            m.register_forward_hook(lambda mod, i, o: rb(o))
        elif isinstance(m, nn.BatchNorm2d):
            m.register_forward_hook(lambda mod, i, o: rb(o))
    return model


def damp(model, gamma):
    enc = model.image_encoder
    last = "bn3" if hasattr(enc.layer1[0], "bn3") else "bn2"
    with torch.no_grad():
        for n, m in enc.named_modules():
            if n.endswith("." + last):
                m.weight.fill_(gamma)
    return model


def model(gamma=None, emu=False, double=False):
    m = det_init_(OracleMultimodalModel(**dict(RESNET50_KW, device="cpu")))
    if gamma is not None:
        damp(m, gamma)
    if emu:
        emulate_bf16_storage(m)
    return m.double() if double else m


def step(m, img, meta, lab, backward=True):
    m.train()
    disable_dropout(m)
    m.zero_grad(set_to_none=True)
    dt = next(m.parameters()).dtype
    crit = nn.CrossEntropyLoss(weight=torch.tensor(CLASS_WEIGHTS, dtype=dt))
    if not backward:
        with torch.no_grad():
            out = m(img.to(dt), meta.to(dt))
            return sf.record(m, out, crit(out, lab))
    out = m(img.to(dt), meta.to(dt))
    loss = crit(out, lab)
    loss.backward()
    return sf.record(m, out, loss, {k: p.grad for k, p in m.named_parameters() if p.grad is not None})


def main():
    want = set(sys.argv[1:])
    img, meta, lab = baseline_batch(256)
    jobs = [
        ("step_b256_g025", lambda: step(model(0.25), img, meta, lab)),
        ("step_b256_default", lambda: step(model(), img, meta, lab)),
        ("step_b256_default_emu", lambda: step(model(emu=True), rb(img), meta, lab)),
    ]
    for gm in (0.1, 0.5, 0.75):
        tag = str(gm).replace(".", "")
        jobs.append((f"step_b256_fwd_g{tag}", lambda gm=gm: step(model(gm), img, meta, lab, backward=False)))
        jobs.append((f"step_b256_fwd_g{tag}_emu", lambda gm=gm: step(model(gm, emu=True), rb(img), meta, lab, backward=False)))
    i64, m64, l64 = baseline_batch(64)
    jobs.append(("step_b64_fp32", lambda: step(model(), i64, m64, l64)))
    jobs.append(("step_b64_fp64", lambda: step(model(double=True), i64, m64, l64)))
    for name, fn in jobs:
        if want and name not in want:
            continue
        rec = fn()
        sf.save(name, rec)
        print(name, "loss", rec["loss"], "params", len(rec["grads"]), flush=True)


if __name__ == "__main__":
    main()

"""Shared test helpers: golden-fixture loading and comparison."""
import json
import os

import torch
import torch.nn as nn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLASS_WEIGHTS = [0.6, 1.7, 0.9, 1.2, 0.4, 2.1]
SMALL = dict(num_classes=6, num_heads=8, device="cpu", cnn_model_name="custom-cnn",
             text_model_name="one-hot-encoder", common_dim=64, text_encoder_dim_output=64,
             vocab_size=20, unfreeze_weights="unfrozen_weights")


def golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def disable_dropout(model):
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0                      # attention-probability dropout (a float, not a module)
        if type(m).__name__ == "StochasticDepth":
            m.p = 0.0
        for attr in ("dropout_p", "drop_p"):
            if hasattr(m, attr):
                setattr(m, attr, 0.0)


def summarize(t):
    t = t.detach().double().flatten().cpu()
    return {"sum": float(t.sum()), "abs": float(t.abs().sum()), "head": [float(v) for v in t[:6]]}


def assert_summary_close(got, want, rtol, atol, what=""):
    """Compare {"sum","abs","head"} summaries. `abs` (L1 norm) scales the tolerance for `sum`."""
    scale = max(want["abs"], 1e-30)
    assert abs(got["abs"] - want["abs"]) <= rtol * scale + atol, (what, "abs", got["abs"], want["abs"])
    assert abs(got["sum"] - want["sum"]) <= rtol * scale + atol, (what, "sum", got["sum"], want["sum"])
    for g, w in zip(got["head"], want["head"]):
        assert abs(g - w) <= rtol * max(abs(w), scale / 1e3) + atol, (what, "head", got["head"], want["head"])


def train_step_record(model, img, meta, lab, device="cpu"):
    """Mirror of oracle.gen_golden.step_record for any model with the reference API."""
    model.eval()
    with torch.no_grad():
        logits_eval = model(img, meta).float().cpu()
    model.train()
    disable_dropout(model)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt = torch.optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4)
    crit = nn.CrossEntropyLoss(weight=torch.tensor(CLASS_WEIGHTS, device=device))
    opt.zero_grad()
    out = model(img, meta)
    loss = crit(out, lab.to(device))
    loss.backward()
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}
    opt.step()
    delta = {k: (p.detach() - before[k]) for k, p in model.named_parameters()}
    return {"logits_eval": logits_eval, "logits_train": out.detach().float().cpu(),
            "loss": float(loss.detach()), "grads": grads, "adam_delta": delta}


def check_record_against_golden(rec, gold, rtol, atol, skip_prefix=()):
    want = torch.tensor(gold["logits_eval"], dtype=torch.float64)
    assert torch.allclose(rec["logits_eval"].double(), want, rtol=rtol, atol=atol), \
        (rec["logits_eval"].double() - want).abs().max()
    if "logits_train" in gold:
        want = torch.tensor(gold["logits_train"], dtype=torch.float64)
        assert torch.allclose(rec["logits_train"].double(), want, rtol=rtol, atol=atol)
    assert abs(rec["loss"] - gold["loss"]) <= rtol * abs(gold["loss"]) + atol
    for k, g in gold["grads"].items():
        if any(k.startswith(p) for p in skip_prefix):
            continue
        got = rec["grads"][k]
        if g is None:
            assert got is None, f"{k}: expected grad None"
            continue
        assert got is not None, f"{k}: expected a gradient"
        assert_summary_close(summarize(got), g, rtol, atol, k)
    for k, z in gold.get("qk_rows_exact_zero", {}).items():
        if z:
            D = rec["grads"][k].shape[1]
            assert bool((rec["grads"][k][:2 * D] == 0).all()), f"{k}: q/k rows must be exact zeros"
    for k, d in gold.get("adam_delta", {}).items():
        if any(k.startswith(p) for p in skip_prefix):
            continue
        # Adam's first step is ~ -lr*sign(g): compare L1 norm of the update loosely
        got = summarize(rec["adam_delta"][k])
        assert abs(got["abs"] - d["abs"]) <= 0.02 * d["abs"] + 1e-9, (k, got["abs"], d["abs"])

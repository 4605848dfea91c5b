"""CPU sides of the production-size train-step parity tests, recorded once in the build container.

One optimisation step of the benchmarked configuration at B = 256 @ 224^2 takes the fp32 CPU oracle minutes; three such steps per
test were 60 % of the GPU suite's wall time on the GPU box's 16-core share.  `tests/golden/gen_step_golden.py`
runs those oracle / bf16-emulation / fp64 steps HERE and stores what the assertions consume: logits, loss, every BatchNorm's running
statistics, and -- per parameter -- the gradient at a fixed seeded SAMPLE of at most K coordinates (the whole tensor when it is
smaller).  Cosines and relative L2 distances of a stage's concatenated gradient are then estimated from the samples with every
tensor weighted by numel / samples, which is unbiased for the inner products the full-vector statistics are made of (relative
noise ~ 1 / sqrt(samples per stage) < 1 %).  The GPU test computes the same samples of the HIP gradients and never runs the oracle.
A fixture is data (inputs are regenerated from seeds, outputs stored); the script that made it is committed beside it."""
import os
import zlib

import numpy as np
import torch
import torch.nn as nn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = 4096


def sample_index(name, numel, k=K):
    if numel <= k:
        return None
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return torch.randperm(numel, generator=g)[:k].sort().values


def sample(name, t, k=K):
    flat = t.detach().flatten().cpu()
    idx = sample_index(name, flat.numel(), k)
    return (flat if idx is None else flat[idx]).double()


def record(model, out, loss, grads=None):
    """What the assertions need from one (train-mode) step of `model`: logits, loss, BatchNorm running statistics of the image
    encoder, sampled gradients + the tensors' sizes."""
    rec = {"out": out.detach().double().cpu(), "loss": float(loss), "stats": {}, "grads": {}, "numel": {}}
    for n, m in model.image_encoder.named_modules():
        if isinstance(m, nn.BatchNorm2d):
            rec["stats"][n] = (m.running_mean.detach().double().cpu().clone(), m.running_var.detach().double().cpu().clone())
    for k, v in (grads or {}).items():
        rec["grads"][k] = sample(k, v)
        rec["numel"][k] = int(v.numel())
    return rec


def save(name, rec):
    arrays = {"out": rec["out"].numpy(), "loss": np.array(rec["loss"], dtype=np.float64)}
    for n, (rm, rv) in rec["stats"].items():
        arrays["rm/" + n] = rm.float().numpy()
        arrays["rv/" + n] = rv.float().numpy()
    for k, v in rec["grads"].items():
        arrays["g/" + k] = v.float().numpy()
        arrays["n/" + k] = np.array(rec["numel"][k], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **arrays)


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    rec = {"out": torch.from_numpy(z["out"]).double(), "loss": float(z["loss"]), "stats": {}, "grads": {}, "numel": {}}
    for key in z.files:
        if key.startswith("rm/"):
            n = key[3:]
            rec["stats"][n] = (torch.from_numpy(z[key]).double(), torch.from_numpy(z["rv/" + n]).double())
        elif key.startswith("g/"):
            k = key[2:]
            rec["grads"][k] = torch.from_numpy(z[key]).double()
            rec["numel"][k] = int(z["n/" + k])
    return rec


def have(name):
    return os.path.exists(os.path.join(GOLDEN, name + ".npz"))


# ---- statistics on sampled gradients
def l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def cos(a, b):
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def stage_stats(ra, rb, keys):
    """cosine and relative L2 of the concatenation of the tensors `keys`, every tensor weighted by numel / samples"""
    ab = aa = bb = dd = 0.0
    for k in keys:
        a, b = ra["grads"][k], rb["grads"][k]
        w = rb["numel"][k] / b.numel()
        ab += w * float(a @ b); aa += w * float(a @ a); bb += w * float(b @ b); dd += w * float((a - b) @ (a - b))
    return ab / ((aa * bb) ** 0.5 + 1e-30), (dd / (bb + 1e-30)) ** 0.5

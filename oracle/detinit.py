"""Deterministic, RNG-free parameter / input fill.  TEST INFRASTRUCTURE ONLY.

Golden fixtures must not depend on torch's RNG stream or on storing multi-MB
state_dicts, so every tensor is filled from a splitmix64 hash of
(crc32(name), element index) computed in numpy uint64 arithmetic:

  * >=2-D ``weight``: uniform(-b, b) with b = 1/sqrt(fan_in)
  * LayerNorm/BatchNorm ``weight`` (1-D ``*norm*.weight``, ``*.bn*.weight``,
    ``*.1.weight`` etc. -- any 1-D tensor named ``weight``): 1 + 0.2*u
  * ``bias`` / 1-D in_proj_bias: 0.2*u
  * ``running_mean``: 0.1*u;  ``running_var``: 1 + 0.2*|u|

The same function is applied to the reference model (when generating the
fixtures), to the oracle and to the HIP model (when checking them).
"""
import zlib

import numpy as np
import torch


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def unit_noise(name, numel, salt=0):
    """float64 array of `numel` values uniform in [-1, 1)."""
    with np.errstate(over="ignore"):
        base = np.uint64(zlib.crc32(name.encode()) + 0x100000000 * (salt + 1))
        idx = np.arange(numel, dtype=np.uint64)
        h = _splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + _splitmix64(base))
    return (h >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def det_tensor(name, shape, scale=1.0, salt=0, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    return torch.from_numpy(unit_noise(name, n, salt) * scale).reshape(shape).to(dtype)


@torch.no_grad()
def det_init_(module, salt=0):
    """Fill every parameter and float buffer of `module` in place."""
    for name, t in list(module.named_parameters()) + list(module.named_buffers()):
        if not t.is_floating_point():
            continue
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            v = 1.0 + 0.2 * det_tensor(name, t.shape, salt=salt).abs()
        elif leaf == "running_mean":
            v = 0.1 * det_tensor(name, t.shape, salt=salt)
        elif t.dim() >= 2:
            fan_in = int(np.prod(t.shape[1:]))
            v = det_tensor(name, t.shape, scale=fan_in ** -0.5, salt=salt)
        elif leaf == "weight":
            v = 1.0 + 0.2 * det_tensor(name, t.shape, salt=salt)
        else:
            v = 0.2 * det_tensor(name, t.shape, salt=salt)
        t.copy_(v.to(t.dtype))
    return module


def det_inputs(batch, hw, meta_cols, num_classes, salt=0):
    img = det_tensor("input.image", (batch, 3, hw, hw), scale=1.5, salt=salt)
    meta = det_tensor("input.meta", (batch, meta_cols), salt=salt)
    lab = (unit_noise("input.label", batch, salt) * 0.5 + 0.5) * num_classes
    return img, meta, torch.from_numpy(np.floor(lab).astype(np.int64)).clamp_(0, num_classes - 1)

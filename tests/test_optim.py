"""mmskin.optim.Adam: torch.optim.Adam's update (train_pad_20.py:54,113) with one launch per flat parameter arena.
CPU: the class is a plain drop-in (no arena -> torch's own step).  -m gpu: arena runs against torch.optim.Adam on the same values."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
from mmskin.optim import Adam   # noqa: E402


def test_cpu_parameters_take_torchs_own_step():
    torch.manual_seed(0)
    w1 = [torch.randn(5, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    w2 = [w.detach().clone().requires_grad_(True) for w in w1]
    o1 = torch.optim.Adam(w1, lr=1e-2, weight_decay=1e-3)
    o2 = Adam(w2, lr=1e-2, weight_decay=1e-3)
    for _ in range(3):
        for a, b in zip(w1, w2):
            g = torch.randn_like(a); a.grad = g.clone(); b.grad = g.clone()
        o1.step(); o2.step()
    assert all(torch.equal(a, b) for a, b in zip(w1, w2))
    assert o2.state_dict()["state"].keys() == o1.state_dict()["state"].keys()


def _arena(shapes, dev, seed):
    g = torch.Generator().manual_seed(seed)
    n = sum(int(torch.tensor(s).prod()) for s in shapes)
    flat = torch.randn(n, generator=g).to(dev)
    ps, off = [], 0
    for s in shapes:
        k = int(torch.tensor(s).prod())
        p = torch.nn.Parameter(torch.empty(0, device=dev))
        p.data = flat[off:off + k].view(s)
        ps.append(p); off += k
    return flat, ps


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 1e-4])
def test_arena_runs_match_torch_adam(wd):
    """A 70 k-element arena of five tensors (one fused launch per step) + two free tensors (torch's step) + one arena tensor
    without a gradient in the middle of a second arena (splits it: the 2-element tail stays with torch) -- five steps against
    torch.optim.Adam on clones: parameters and both moments within 2e-6 of their largest element (fp32 elementwise, different association)."""
    dev = "cuda:0"
    shapes = [(64, 64, 3, 3), (64,), (256, 128), (33,), (5, 7, 11)]
    flat, ps = _arena(shapes, dev, 1)
    flat2, qs = _arena([(300, 300), (4,), (2,)], dev, 2)
    free = [torch.nn.Parameter(torch.randn(17, 5, device=dev)), torch.nn.Parameter(torch.randn(9, device=dev))]
    mine = ps + qs + free
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    o_ref = torch.optim.Adam(ref, lr=3e-3, weight_decay=wd)
    o = Adam(mine, lr=3e-3, weight_decay=wd)
    gen = torch.Generator().manual_seed(3)
    v0 = ps[0]._version
    for it in range(5):
        gflat = torch.randn(flat.numel(), generator=gen).to(dev)
        gflat2 = torch.randn(flat2.numel(), generator=gen).to(dev)
        off = 0
        for p in ps:
            p.grad = gflat[off:off + p.numel()].view(p.shape); off += p.numel()
        off = 0
        for i, q in enumerate(qs):
            q.grad = None if i == 1 else gflat2[off:off + q.numel()].view(q.shape)
            off += q.numel()
        for f in free:
            f.grad = torch.randn(f.shape, generator=gen).to(dev)
        for a, b in zip(mine, ref):
            b.grad = None if a.grad is None else a.grad.detach().clone()
        if it == 2:   # a learning-rate schedule between steps
            for grp in o.param_groups + o_ref.param_groups:
                grp["lr"] = 1e-3
        o.step(); o_ref.step()
        assert all(a.grad is None or a.grad is not None for a in mine)
    assert ps[0].grad is not None                       # gradients are handed back after the step
    assert ps[0]._version > v0                          # version counters move like after an in-place update
    assert len(o._runs) == 2                            # the first arena whole, (300, 300) of the second
    for a, b in zip(mine, ref):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), float((a - b).abs().max())
        if a.grad is not None:
            for k in ("exp_avg", "exp_avg_sq"):
                want = o_ref.state[b][k]
                assert float((o.state[a][k] - want).abs().max()) <= 2e-6 * float(want.abs().max()), k
            assert float(o.state[a]["step"]) == float(o_ref.state[b]["step"]) == 5.0
    # state_dict round trip: a fresh optimizer continues from the loaded moments
    sd = o.state_dict()
    o2 = Adam(mine, lr=1e-3, weight_decay=wd)
    o2.load_state_dict(sd)
    sd_ref = o_ref.state_dict()
    for a, b in zip(mine, ref):
        if a.grad is not None:
            g = torch.randn(a.shape, generator=gen).to(dev)
            a.grad.copy_(g); b.grad.copy_(g)
    o2.step(); o_ref.step()
    for a, b in zip(mine, ref):
        assert torch.allclose(a, b, rtol=3e-6, atol=1e-7), float((a - b).abs().max())

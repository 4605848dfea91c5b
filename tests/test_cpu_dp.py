"""CPU, gloo, world_size 2: the data-parallel step -- rank-sharded batch, one gradient
all-reduce, parameters off the fusion branch keep grad None, encoder gradients reduced in place through the
flat buffer."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from helpers import SMALL
from mmskin import dp
from oracle.detinit import det_init_, det_inputs
from oracle.model import OracleMultimodalModel


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _full_batch_grads(mech):
    model = det_init_(OracleMultimodalModel(**dict(SMALL, attention_mecanism=mech)))
    model.eval()                      # no dropout; custom-cnn has no BatchNorm
    img, meta, lab = det_inputs(8, 32, 20, 6)
    nn.functional.cross_entropy(model(img, meta), lab).backward()
    return {k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}


def _worker(rank, world, port, mech, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)     # different init per rank: broadcast must fix it
    model = OracleMultimodalModel(**dict(SMALL, attention_mecanism=mech))
    if rank == 0:
        det_init_(model)
    dp.broadcast_parameters(model)
    model.eval()
    img, meta, lab = det_inputs(8, 32, 20, 6)
    idx = torch.arange(8)[rank * 4:(rank + 1) * 4]
    nn.functional.cross_entropy(model(img[idx], meta[idx]), lab[idx]).backward()
    nbytes = dp.allreduce_gradients(model)
    want = _full_batch_grads(mech)
    bad = []
    for k, p in model.named_parameters():
        if want[k] is None:
            if p.grad is not None:
                bad.append(k + ": expected None")       # off-branch parameters stay None on every rank
        elif p.grad is None or not torch.allclose(p.grad, want[k], rtol=1e-4, atol=1e-6):
            bad.append(k)
    q.put((rank, nbytes, bad))                           # plain Python only: tensors do not survive the exit
    dist.barrier()
    dist.destroy_process_group()


def test_dp_step_matches_full_batch():
    mech = "gfcam"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mech, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nbytes, bad in results:
        assert nbytes > 0
        assert not bad, (rank, bad)


def _flat_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmskin.backbone import HipResNet

    class Wrap(nn.Module):
        def __init__(self):
            super().__init__()
            self.image_encoder = HipResNet("resnet-18")
            self.head = nn.Linear(4, 2)
    m = Wrap()
    enc = m.image_encoder
    n = sum(p.numel() for p in enc.parameters())
    flat = torch.full((n,), float(rank + 1))
    enc.last_flat_grad = flat
    for p, (off, numel, shape) in zip(enc.parameters(), enc._layout):
        p.grad = flat[off:off + numel].view(shape)       # what _BackboneFn.backward does
    m.head.weight.grad = torch.full((2, 4), float(10 * (rank + 1)))
    dp.allreduce_gradients(m)
    ok = bool((flat == 1.5).all()) and bool((enc.conv1.weight.grad == 1.5).all()) \
        and bool((m.head.weight.grad == 15.0).all()) and m.head.bias.grad is None \
        and enc.conv1.weight.grad.data_ptr() == flat.data_ptr()
    q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_encoder_gradients_reduced_in_place():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert all(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


class _FakePlan:
    """grad_segments() of a plan without a GPU: two ranges in completion order."""

    def __init__(self, n):
        self.n = n

    def grad_segments(self):
        return [(self.n // 2, self.n - self.n // 2), (0, self.n // 2)]


def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmskin.backbone import HipResNet

    class Wrap(nn.Module):
        def __init__(self):
            super().__init__()
            self.image_encoder = HipResNet("resnet-18")
            self.head = nn.Linear(4, 2)
            self.late = nn.Linear(3, 1)
    m = Wrap()
    enc = m.image_encoder
    sync = dp.OverlappedGradSync(m)
    n = sum(p.numel() for p in enc.parameters())
    # what autograd + _BackboneFn.backward do, in that order: head grads, hook, encoder backward, hook
    m.head.weight.grad = torch.full((2, 4), float(10 * (rank + 1)))
    m.head.bias.grad = torch.full((2,), float(rank + 1))
    sync.before_encoder_backward()
    m.head.bias.grad.add_(2.0)                            # touched after it was bucketed: must be reduced again from its value
    m.late.weight.grad = torch.full((1, 3), float(rank))  # appears after the head bucket left
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    enc.last_flat_grad = flat
    for p, (off, numel, shape) in zip(enc.parameters(), enc._layout):
        p.grad = flat[off:off + numel].view(shape)
    sync.after_encoder_backward(_FakePlan(n), flat)
    nbytes = sync.finish()
    ok = torch.allclose(flat, torch.arange(n, dtype=torch.float32) * 1.5) \
        and bool((m.head.weight.grad == 15.0).all()) and bool((m.head.bias.grad == 3.5).all()) \
        and bool((m.late.weight.grad == 0.5).all()) and m.late.bias.grad is None \
        and enc.conv1.weight.grad.data_ptr() == flat.data_ptr() and nbytes >= 4 * n
    # a second step without the hooks firing (e.g. frozen encoder): finish() must fall back to the plain path
    for p in m.parameters():
        p.grad = None
    m.head.weight.grad = torch.full((2, 4), float(rank))
    sync.finish()
    ok = ok and bool((m.head.weight.grad == 0.5).all())
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_grad_sync_matches_plain_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert all(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


def test_shard_indices_partition():
    """uniform path: the ranks' slices partition the (wrap-padded) permutation; nothing of range(n) is dropped"""
    a, b = dp.shard_indices(101, 0, 2, epoch_seed=3), dp.shard_indices(101, 1, 2, epoch_seed=3)
    assert len(a) == len(b) == 51
    assert set(a.tolist()) | set(b.tolist()) == set(range(101))
    assert len(set(a.tolist()) & set(b.tolist())) == 1            # the one wrap-around pad index
    assert not torch.equal(a, dp.shard_indices(101, 0, 2, epoch_seed=4))


def test_shard_indices_weighted_matches_single_process_sampler():
    """weighted path = the reference's WeightedRandomSampler(sample_weights, n, replacement=True)
    (train_pad_20.py:290-302): the union over ranks is the single-process draw, classes come out balanced"""
    from torch.utils.data import WeightedRandomSampler
    labels = torch.tensor([0] * 90 + [1] * 9 + [2] * 4)            # imbalanced, n = 103 (not a multiple of 4)
    class_w = 1.0 / torch.bincount(labels).double()
    w = class_w[labels]
    n, world = len(labels), 4
    parts = [dp.shard_indices(n, r, world, epoch_seed=7, weights=w) for r in range(world)]
    assert all(len(p) == 26 for p in parts)                        # ceil(103 / 4): equal step counts on every rank
    merged = torch.stack(parts, dim=1).reshape(-1)                 # undo the strided slicing
    g = torch.Generator().manual_seed(7)
    single = torch.tensor(list(WeightedRandomSampler(w, n, replacement=True, generator=g)))
    assert torch.equal(merged[:n], single)                         # same draw as the single-process sampler
    assert torch.equal(merged[n:], single[:world * 26 - n])        # padding wraps around
    assert len(set(single.tolist())) < n                           # with replacement: indices repeat
    big = torch.cat([dp.shard_indices(n, r, world, epoch_seed=11, weights=w, num_samples=6000) for r in range(world)])
    frac = torch.bincount(labels[big], minlength=3).double() / big.numel()
    assert (frac - 1 / 3).abs().max() < 0.04                       # class-balanced, unlike a uniform permutation


def test_bench_entry_self_launches_two_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run must spawn its own rank processes: the
    2-rank gloo rehearsal of exactly that entry -- launcher, rendezvous on 127.0.0.1, gradient all-reduce, max-over-ranks
    timing, ONE JSON line from rank 0, non-zero exit if a rank fails."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["ranks_in_sync"] is True and d["data"] == "rehearsal"
    # work-skipping switches are refused outright
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--rehearse"],
                         env=dict(env, MMSKIN_CONV_ABLATE="4"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "work-skipping" in bad.stderr

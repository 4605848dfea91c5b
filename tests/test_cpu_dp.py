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
    a, b = dp.shard_indices(101, 0, 2, epoch_seed=3), dp.shard_indices(101, 1, 2, epoch_seed=3)
    assert len(a) == len(b) == 50 and not set(a.tolist()) & set(b.tolist())
    assert not torch.equal(a, dp.shard_indices(101, 0, 2, epoch_seed=4))

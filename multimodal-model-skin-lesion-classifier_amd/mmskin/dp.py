"""Data-parallel K-fold training support: one process per GPU, gradients averaged with RCCL.

The reference is single-process / single-GPU (train_pad_20.py:509); this is the build-side addition
SURVEY.md section 8(e) describes.  Samples are independent given the parameters and BatchNorm uses
per-GPU batch statistics (the reference has no SyncBN), so the only exchange step is ONE gradient
all-reduce per optimisation step over the parameters whose grad is not None (parameters off the chosen
fusion branch keep grad None on every rank, exactly as on one GPU).

Buckets: the image encoder's gradients already live in one flat fp32 buffer (the backward C-ABI call
writes them there and the parameters' .grad are views into it), so that buffer is all-reduced in place
without any flatten/unflatten copies; the head's gradients are flattened into a second small bucket.
"""
import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s parameters and buffers."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src)


def _aliases(flat, grads):
    lo = flat.data_ptr()
    hi = lo + flat.numel() * flat.element_size()
    return all(lo <= g.data_ptr() < hi for g in grads)


@torch.no_grad()
def allreduce_gradients(model, world_size=None, group=None):
    """Average .grad over ranks.  Returns the number of bytes reduced (for logging)."""
    world_size = world_size or dist.get_world_size(group)
    if world_size == 1:
        return 0
    enc = getattr(model, "image_encoder", None)
    flat = getattr(enc, "last_flat_grad", None)
    enc_ids = set()
    nbytes = 0
    if flat is not None:
        enc_grads = [p.grad for p in enc.parameters() if p.grad is not None]
        if enc_grads and _aliases(flat, enc_grads) and len(enc_grads) == len(list(enc.parameters())):
            dist.all_reduce(flat, group=group)
            flat.div_(world_size)
            enc_ids = {id(p) for p in enc.parameters()}
            nbytes += flat.numel() * 4
    rest = [p.grad for p in model.parameters() if p.grad is not None and id(p) not in enc_ids]
    if rest:
        bucket = _flatten_dense_tensors(rest)
        dist.all_reduce(bucket, group=group)
        bucket.div_(world_size)
        for g, r in zip(rest, _unflatten_dense_tensors(bucket, rest)):
            g.copy_(r)
        nbytes += bucket.numel() * 4
    return nbytes


def shard_indices(n, rank, world_size, epoch_seed=0):
    """Rank-sharded random permutation of range(n) (replaces the single-process sampler,
    train_pad_20.py:298-302, under DP): every rank draws the same permutation and keeps its slice."""
    g = torch.Generator().manual_seed(epoch_seed)
    perm = torch.randperm(n, generator=g)
    per = n // world_size
    return perm[rank * per:(rank + 1) * per]

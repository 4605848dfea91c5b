"""Data-parallel K-fold training support: one process per GPU, gradients averaged with RCCL.

The reference is single-process / single-GPU (train_pad_20.py:509); this is the build-side addition
SURVEY.md section 8(e) describes.  Samples are independent given the parameters and BatchNorm uses
per-GPU batch statistics (the reference has no SyncBN), so the only exchange step is ONE gradient
all-reduce per optimisation step over the parameters whose grad is not None (parameters off the chosen
fusion branch keep grad None on every rank, exactly as on one GPU).

Buckets: the image encoder's gradients already live in one flat fp32 buffer (the backward C-ABI call
writes them there and the parameters' .grad are views into it), so that buffer is all-reduced in place
without any flatten/unflatten copies; the head's gradients are flattened into a second small bucket.

Overlap (`OverlappedGradSync`): the fusion head is last in forward, so its gradients are complete when
autograd reaches the image encoder; its bucket is launched then.  The encoder's backward is ONE C call
that enqueues ~300 kernels and returns; the plan records an event pair per gradient segment (ResNet:
layer4, layer3, layer2, layer1 + stem; `mmskin_backbone_wait_grad_segment`), so right after that call
returns one all-reduce per segment is queued behind its events and runs on RCCL's stream under the rest
of backward: layer4's 60 MB start after ~25 % of the encoder's backward.  Only the last 0.9 MB segment is exposed.
"""
import torch
import torch.distributed as dist
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


def max_inflight_segments():
    """How many of the encoder's gradient segments may have their all-reduce queued behind backward at the same time
    (MMSKIN_DP_STREAMS, default 2: layer4 + layer3 / the rest).  Every queued all-reduce is one more HBM consumer beside the main
    chain and the weight-gradient stream (r01_e (8) / (13): a third concurrent consumer slows the main chain), and torch's
    ProcessGroupNCCL runs them one after the other on its own stream anyway -- the wait streams only decide WHEN each may start."""
    import os
    return max(1, int(os.environ.get("MMSKIN_DP_STREAMS", "2")))


def segment_lag():
    """MMSKIN_DP_LAG (default 0): the all-reduce of gradient segment i starts only when segment i + lag has ALSO been completed by
    backward (clamped to the last segment), i.e. it runs `lag` segments later instead of right beside the weight-gradient GEMMs of the
    next segment.  On one GPU every concurrent HBM consumer beside the main chain and the weight-gradient stream costs the main chain
    about what it moves (profiles/r04_experiments.txt (3)); whether RCCL's kernels are such a consumer on the 8-GPU node is what the first
    scaling run has to show -- lag = number of segments (4 for ResNet) queues every all-reduce behind the whole backward."""
    import os
    return max(0, int(os.environ.get("MMSKIN_DP_LAG", "0")))


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s parameters and buffers."""
    # broadcast into detach() views, not `.data`: a detached view shares the parameter's version counter, so caches keyed by it
    # (ops.frozen_bf16: bf16 copies of frozen weights) see the write; a write through `.data` leaves the counter where it was
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.detach(), src)
    from . import ops
    ops.invalidate_frozen_cache()


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


class OverlappedGradSync:
    """Gradient averaging overlapped with the image encoder's backward.

        sync = OverlappedGradSync(model)        # once, after the process group exists
        loss.backward()                          # all-reduces are queued from inside the encoder's backward
        sync.finish()                            # wait + average; falls back to allreduce_gradients() when nothing was queued

    `finish()` leaves every parameter's .grad equal to what `allreduce_gradients` produces (tests/test_dp.py)."""

    def __init__(self, model, world_size=None, group=None, force=False):
        self.model = model
        self.group = group
        self.world_size = world_size or dist.get_world_size(group)
        self.active = self.world_size > 1 or force      # force: exercise the queueing on a single-rank group (tests)
        enc = getattr(model, "image_encoder", None)
        self.enc = enc if hasattr(enc, "last_flat_grad") else getattr(enc, "features", None)
        if self.enc is None or not hasattr(self.enc, "last_flat_grad"):
            raise ValueError("OverlappedGradSync needs a plan-executed image encoder (flat gradient arena)")
        self.enc._grad_sync = self
        self._enc_ids = {id(p) for p in self.enc.parameters()}
        self._streams = []
        self._reset()

    def _reset(self):
        self._works = []
        self._flat = None
        self._head = None        # (bucket, grads, versions, work)

    def detach(self):
        self.enc._grad_sync = None

    # ---- called from _BackboneFn.backward (mmskin/backbone.py)
    @torch.no_grad()
    def before_encoder_backward(self):
        self._reset()
        if not self.active:
            return
        ready = [p.grad for p in self.model.parameters() if p.grad is not None and id(p) not in self._enc_ids]
        if ready:
            bucket = _flatten_dense_tensors(ready)
            work = dist.all_reduce(bucket, group=self.group, async_op=True)
            self._head = (bucket, ready, [g._version for g in ready], work)

    @torch.no_grad()
    def after_encoder_backward(self, plan, flat):
        if not self.active:
            return
        segs = plan.grad_segments()
        if not segs:
            return
        on_gpu = flat.is_cuda
        nstreams = min(len(segs), max_inflight_segments())
        lag = segment_lag()
        while on_gpu and len(self._streams) < nstreams:
            self._streams.append(torch.cuda.Stream(device=flat.device))
        for i, (off, numel) in enumerate(segs):
            view = flat[off:off + numel]
            if on_gpu:
                s = self._streams[i % nstreams]       # segment i + nstreams queues behind segment i on the same wait stream
                with torch.cuda.stream(s):
                    plan.wait_grad_segment(i, s)      # s waits for the segment's events only, not for the rest of backward
                    if lag:                           # ... and, lagged, for a later segment's (segments complete in order)
                        plan.wait_grad_segment(min(i + lag, len(segs) - 1), s)
                    self._works.append(dist.all_reduce(view, group=self.group, async_op=True))
            else:
                self._works.append(dist.all_reduce(view, group=self.group, async_op=True))
        self._flat = flat

    # ---- called by the training loop after loss.backward()
    @torch.no_grad()
    def finish(self):
        if not self.active:
            self._reset()
            return 0
        nbytes = 0
        done_ids = set()
        if self._flat is not None and self.enc.last_flat_grad is self._flat:
            for w in self._works:
                w.wait()
            self._flat.div_(self.world_size)
            done_ids |= self._enc_ids
            nbytes += self._flat.numel() * 4
        if self._head is not None:
            bucket, grads, versions, work = self._head
            work.wait()
            bucket.div_(self.world_size)
            live = {id(p.grad): p for p in self.model.parameters() if p.grad is not None}
            for g, v, r in zip(grads, versions, _unflatten_dense_tensors(bucket, grads)):
                p = live.get(id(g))
                if p is not None and g._version == v:    # untouched since it was bucketed: take the averaged value
                    g.copy_(r)
                    done_ids.add(id(p))
            nbytes += bucket.numel() * 4
        rest = [p.grad for p in self.model.parameters() if p.grad is not None and id(p) not in done_ids]
        if rest:                                          # anything the overlapped path did not cover
            bucket = _flatten_dense_tensors(rest)
            dist.all_reduce(bucket, group=self.group)
            bucket.div_(self.world_size)
            for g, r in zip(rest, _unflatten_dense_tensors(bucket, rest)):
                g.copy_(r)
            nbytes += bucket.numel() * 4
        self._reset()
        return nbytes


def shard_indices(n, rank, world_size, epoch_seed=0, weights=None, num_samples=None):
    """This rank's share of one epoch's sample indices (replaces the single-process sampler of
    train_pad_20.py:297-302 under DP).  Every rank makes the SAME seeded draw and keeps a strided slice of it, so
    the union over ranks is exactly the single-process draw.

    weights given (the reference's case: `WeightedRandomSampler(sample_weights, num_samples=len(sample_weights),
    replacement=True)`, class-balanced, train_pad_20.py:290-302): `torch.multinomial(weights, num_samples,
    replacement=True)` -- the call WeightedRandomSampler itself makes -- so indices repeat and rare classes are drawn as
    often as on one GPU.  weights None: a uniform permutation without replacement.

    The draw is padded by wrapping around (as torch's DistributedSampler does) so that every rank gets
    ceil(num_samples / world_size) indices and no sample of the draw is dropped: the per-rank batch counts stay
    equal, which the per-step gradient all-reduce needs."""
    g = torch.Generator().manual_seed(int(epoch_seed))
    if weights is not None:
        w = torch.as_tensor(weights, dtype=torch.double)
        total = int(num_samples) if num_samples is not None else w.numel()
        draw = torch.multinomial(w, total, replacement=True, generator=g)
    else:
        draw = torch.randperm(n, generator=g)
        if num_samples is not None:
            draw = draw[:int(num_samples)]
    total = draw.numel()
    per = -(-total // world_size)
    pad = per * world_size - total
    if pad:
        draw = torch.cat([draw, draw[:pad]])
    return draw[rank::world_size]

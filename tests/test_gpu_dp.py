"""-m gpu: the overlapped data-parallel gradient all-reduce (mmskin/dp.py OverlappedGradSync) on the
real HIP backward: per-segment events of the plan (`mmskin_backbone_wait_grad_segment`), queued from inside backward.

A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so
 (a) two ranks share the GPU over `gloo` and the overlapped result is compared with the plain all-reduce, and
 (b) a single-rank `nccl` (= RCCL) group runs the same queueing with `force=True` (stream / event plumbing under RCCL).
The 2-, 4- and 8-GPU RCCL runs are the driver's (bench.py --gpus N)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _grads(model):
    return {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}


def _worker(rank, world, port, backend, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MMSKIN_BACKBONE_DTYPE="bf16")
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from helpers import disable_dropout
        from mmskin import dp
        from models import multimodalIntraInterModal as M
        torch.manual_seed(5)
        model = M.MultimodalModel(num_classes=6, num_heads=8, device="cuda:0", cnn_model_name="resnet-50",
                                  text_model_name="one-hot-encoder", vocab_size=20, attention_mecanism="crossattention",
                                  unfreeze_weights="unfrozen_weights").to("cuda:0")
        dp.broadcast_parameters(model)
        model.train()
        disable_dropout(model)
        g = torch.Generator().manual_seed(100 + rank)
        img = torch.randn(16, 3, 224, 224, generator=g).cuda()
        meta = torch.randn(16, 20, generator=g).cuda()
        lab = torch.randint(0, 6, (16,), generator=g).cuda()

        def backward():
            for p in model.parameters():
                p.grad = None
            nn.functional.cross_entropy(model(img, meta), lab).backward()

        backward()
        dp.allreduce_gradients(model, world)
        plain = _grads(model)
        sync = dp.OverlappedGradSync(model, world, force=True)
        segs = next(iter(model.image_encoder._plans.values())).grad_segments()
        n_enc = sum(p.numel() for p in model.image_encoder.parameters())
        ok_segs = len(segs) == 4 and sum(n for _, n in segs) == n_enc and segs[-1][0] == 0
        bad = []
        for _ in range(2):                      # twice: events and streams are reused
            backward()
            queued = len(sync._works)
            sync.finish()
            got = _grads(model)
            for k, w in plain.items():
                if w is None:
                    if got[k] is not None:
                        bad.append(k + ": expected None")
                elif got[k] is None or not torch.allclose(got[k], w, rtol=1e-5, atol=1e-7):
                    bad.append(k)
        torch.cuda.synchronize()
        q.put((rank, ok_segs, queued, bad[:5]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:          # surface the failure instead of a queue timeout
        q.put((rank, False, -1, [repr(e)]))
        raise


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_overlapped_allreduce_on_hip_backward(backend, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=420) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok_segs, queued, bad in results:
        assert ok_segs, (rank, "segment table")
        assert queued == 4, (rank, queued)
        assert not bad, (rank, bad)

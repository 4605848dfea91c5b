#!/usr/bin/env python3
"""Headline benchmark: images/sec of one full training step (zero_grad -> forward -> weighted CE ->
backward -> [gradient all-reduce] -> Adam) of ResNet-50 + one-hot metadata encoder + 'crossattention'
fusion, batch 256 per GPU, bf16 backbone compute, synthetic 224x224x3 images + 20 metadata columns
(BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: spawns its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events recorded on the launch stream
around every kernel of the dominant class (the implicit-GEMM convolution kernel, forward + dgrad);
`cpu_baseline` times the CPU oracle (a port: the reference's Python never travels to the GPU box) on a
bounded sample of the same workload.
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.nn as nn

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md (chip table)
PEAK_F32_TFLOPS = 157.3
FLOP_PER_IMAGE = 24.33e9    # conv fwd 8.174 + bwd 16.113 + head 0.039 GFLOP
CLASS_NAMES = ["conv_fwd", "conv_dgrad", "wgrad", "bn_fwd", "bn_bwd", "stage_weights", "stem_misc"]

# The headline line is configs[1]; configs[2] (DenseNet-169 + tab-transformer + metablock) is available
# with --workload densenet169-metablock for the widened path.
WORKLOADS = {
    "resnet50-crossattention": {
        "metric": "images/sec fwd+bwd, ResNet-50+crossattention bs=256",
        "label": "ResNet-50 + one-hot(20) + crossattention, 224x224, train step incl. Adam",
        "kw": dict(cnn_model_name="resnet-50", text_model_name="one-hot-encoder", common_dim=512, vocab_size=20,
                   attention_mecanism="crossattention"),
        "flop_per_image": FLOP_PER_IMAGE,
    },
    "densenet169-metablock": {
        "metric": "images/sec fwd+bwd, DenseNet-169+tab-transformer+metablock bs=256",
        "label": "DenseNet-169 + tab-transformer(82 cat + 4 cont) + metablock, 224x224, train step incl. Adam",
        "kw": dict(cnn_model_name="densenet169", text_model_name="tab-transformer", common_dim=512, vocab_size=86,
                   attention_mecanism="metablock"),
        "flop_per_image": 3 * 6.72e9,   # torchvision densenet169: 3.36 GMAC forward; backward = 2x forward
    },
    "mobilenetv2-crossattention": {
        "metric": "images/sec fwd+bwd, MobileNet-V2+crossattention bs=256",
        "label": "MobileNet-V2 + one-hot(20) + crossattention, 224x224, train step incl. Adam",
        "kw": dict(cnn_model_name="mobilenet-v2", text_model_name="one-hot-encoder", common_dim=512, vocab_size=20,
                   attention_mecanism="crossattention"),
        "flop_per_image": 3 * 0.6e9,   # torchvision mobilenet_v2: 0.30 GMAC forward
    },
    "efficientnetb0-crossattention": {
        "metric": "images/sec fwd+bwd, EfficientNet-B0+crossattention bs=256",
        "label": "EfficientNet-B0 + one-hot(20) + crossattention, 224x224, train step incl. Adam",
        "kw": dict(cnn_model_name="efficientnet-b0", text_model_name="one-hot-encoder", common_dim=512, vocab_size=20,
                   attention_mecanism="crossattention"),
        "flop_per_image": 3 * 0.78e9,   # torchvision efficientnet_b0: 0.39 GMAC forward
    },
    # BASELINE configs[3]: bs=512 bf16 over DP=8 -> 64 per GPU; large Linear GEMMs with bf16 operands (fp32 accumulate)
    "davit-tiny-gfcam": {
        "metric": "images/sec fwd+bwd, DaViT-tiny+tab-transformer+gfcam bs=64/GPU",
        "label": "davit_tiny.msft_in1k + tab-transformer(82 cat + 4 cont) + gfcam, 224x224, train step incl. Adam (bf16-operand Linear GEMMs)",
        "kw": dict(cnn_model_name="davit_tiny.msft_in1k", text_model_name="tab-transformer", common_dim=512, vocab_size=86,
                   attention_mecanism="gfcam"),
        "flop_per_image": 3 * 9.0e9,   # timm davit_tiny: 4.5 GMAC forward
        "default_batch": 64, "linear_dtype": "bf16",
    },
    # BASELINE configs[4]: bs=1024 over DP=8 -> 128 per GPU; last-block fine-tuning of the image encoder ("partial"), frozen BERT
    "beitv2-large-bert-rgatt": {
        "metric": "images/sec fwd+bwd, BEiTv2-large+bert-base-uncased+RG-ATT bs=128/GPU",
        "label": "beitv2_large_patch16_224 + bert-base-uncased (512 tokens) + att-intramodal+residual+cross-attention-metadados, "
                 "224x224, train step incl. Adam (bf16-operand Linear GEMMs, 'partial' unfreeze)",
        "kw": dict(cnn_model_name="beitv2_large_patch16_224", text_model_name="bert-base-uncased", common_dim=512, vocab_size=20,
                   attention_mecanism="att-intramodal+residual+cross-attention-metadados"),
        "unfreeze": "partial",
        "flop_per_image": 2 * 61.6e9 + 2 * 49.0e9,   # forward only through the frozen parts: BEiT-L 61.6 GMAC + BERT-base 49 GMAC at 512 tokens
        "default_batch": 128, "linear_dtype": "bf16",
    },
    "vgg16-crossattention": {
        "metric": "images/sec fwd+bwd, VGG-16+crossattention bs=256",
        "label": "VGG-16 + one-hot(20) + crossattention, 224x224, train step incl. Adam",
        "kw": dict(cnn_model_name="vgg16", text_model_name="one-hot-encoder", common_dim=512, vocab_size=20,
                   attention_mecanism="crossattention"),
        "flop_per_image": 3 * 30.9e9,   # torchvision vgg16: 15.47 GMAC forward
    },
}


def source_hash():
    """Identity of the kernel sources the loaded library was built from (the GPU box has no .git): sha256 over csrc/."""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def active_knobs():
    """Every MMSKIN_* environment variable in effect: part of `config`, so a tuned or experimental run is visible as one."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("MMSKIN_")}


def refuse_work_skipping():
    """The production library has the ablation switches compiled out (`make ablate` builds a separate one for scripts/);
    a benchmark run must not even be asked to skip work."""
    bad = [k for k in os.environ if k.startswith("MMSKIN_") and "ABLATE" in k]
    if bad:
        raise SystemExit(f"bench.py: refusing to run with work-skipping switches set: {bad}")


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start N fresh rank processes of this same
    script BEFORE this process touches the GPU, relay rank 0's JSON line, fail if any rank fails."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))   # stderr of every rank is inherited
    # poll EVERY child: a rank that dies during init or in a step would leave the others inside a collective until the RCCL
    # timeout; terminate them at once and fail (rank 0's stdout is drained by a reader thread so a long line cannot block it)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
                break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    codes = [p.wait() for p in procs]
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    if any(codes):
        raise SystemExit(f"bench.py: rank exit codes {codes}" + (f" (rank {failed} failed first; the others were terminated)" if failed is not None else ""))


def rehearse(args, rank, world):
    """Everything of main() that is not the HIP path, on CPU over gloo (see --rehearse)."""
    import torch.distributed as dist
    from mmskin import dp
    if world > 1:
        dist.init_process_group("gloo")
    torch.manual_seed(rank)
    model = nn.Sequential(nn.Linear(20, 32), nn.ReLU(), nn.Linear(32, 6))
    if world > 1:
        dp.broadcast_parameters(model)
    g = torch.Generator().manual_seed(1234 + rank)
    x, y = torch.randn(16, 20, generator=g), torch.randint(0, 6, (16,), generator=g)
    opt = torch.optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = nn.functional.cross_entropy(model(x), y)
        loss.backward()
        if world > 1:
            dp.allreduce_gradients(model, world)
        opt.step()
        return loss
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    w0 = model[0].weight.detach().clone()
    same = True
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        ref = w0.clone()
        dist.broadcast(ref, 0)
        ok = torch.tensor([float(torch.equal(ref, w0))])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        same = bool(ok.item())
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL: launcher / collective plumbing only, no HIP path", "value": None, "unit": None,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "data": "rehearsal", "ranks_in_sync": same}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not same:
        raise SystemExit("rehearsal: ranks diverged")


def make_meta(workload, batch, generator):
    if workload == "beitv2-large-bert-rgatt":   # tokenised metadata sentences (skinLesionDatasetsWithBert.py:67-78): ids / mask [B, 1, 512]
        ids = torch.randint(1, 30000, (batch, 1, 512), generator=generator)
        mask = torch.ones_like(ids)
        mask[:, :, 384:] = 0
        return {"input_ids": ids, "attention_mask": mask}
    if workload in ("densenet169-metablock", "davit-tiny-gfcam"):   # 82 categorical codes (cardinality 10) + 4 continuous columns
        cat = torch.randint(0, 10, (batch, 82), generator=generator).float()
        return torch.cat([cat, torch.randn(batch, 4, generator=generator)], dim=1)
    return torch.randn(batch, 20, generator=generator)


def build_model(device, dtype, workload="resnet50-crossattention", cls=None):
    os.environ["MMSKIN_BACKBONE_DTYPE"] = dtype
    if cls is None:
        from models import multimodalIntraInterModal as M
        cls = M.MultimodalModel
    torch.manual_seed(0)
    model = cls(num_classes=6, num_heads=8, device=device, unfreeze_weights=WORKLOADS[workload].get("unfreeze", "unfrozen_weights"), n=2,
                **WORKLOADS[workload]["kw"])
    return model.to(device)


def cpu_baseline(batch, seconds=12.0, workload="resnet50-crossattention"):
    """fwd+bwd of the CPU oracle (fp32, all host threads) on batches of `batch` until ~`seconds` elapsed."""
    from oracle.model import OracleMultimodalModel
    # the GPU box exposes every host core but a 1-GPU job owns a 16-core share: do not oversubscribe
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 16)))
    torch.manual_seed(0)
    model = build_model("cpu", "fp32", workload, cls=OracleMultimodalModel)
    model.train()
    img, meta = torch.randn(batch, 3, 224, 224), make_meta(workload, batch, None)
    lab = torch.randint(0, 6, (batch,))
    crit = nn.CrossEntropyLoss()
    def step():
        model.zero_grad(set_to_none=True)
        crit(model(img, meta), lab).backward()
    step()                                   # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        step(); n += 1
        if time.perf_counter() - t0 > seconds or n >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n * batch / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} fwd+bwd steps of batch {batch} (fp32, torch CPU oracle), {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default 256, or the workload's per-GPU share")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="resnet50-crossattention", choices=sorted(WORKLOADS))
    ap.add_argument("--infer", action="store_true", help="time the eval-mode forward only (secondary line; BN folded into the convs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="force every rank onto this device (rehearsal on a 1-GPU box)")
    ap.add_argument("--rehearse", action="store_true",
                    help="CPU plumbing check of the multi-rank entry (tests/test_cpu_dp.py): launcher, rendezvous, gradient "
                         "all-reduce, max-over-ranks timing and the JSON line, with a toy torch model over gloo -- no HIP "
                         "path, no benchmark number")
    args = ap.parse_args()
    knobs = active_knobs()

    refuse_work_skipping()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse:
        return rehearse(args, rank, world)
    dev_index = local_rank if args.device is None else args.device
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    import torch.distributed as dist
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
    from mmskin import _lib, dp

    wl = WORKLOADS[args.workload]
    if args.batch is None:
        args.batch = wl.get("default_batch", 256)
    if wl.get("linear_dtype") and "MMSKIN_LINEAR_DTYPE" not in os.environ:
        from mmskin import ops as _ops
        _ops.set_linear_dtype(wl["linear_dtype"])
    model = build_model(device, args.dtype, args.workload)
    if world > 1:
        dp.broadcast_parameters(model)
    model.train(not args.infer)
    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    image = torch.randn(B, 3, 224, 224, generator=g).to(device)
    meta = make_meta(args.workload, B, g)
    meta = {k: v.to(device) for k, v in meta.items()} if isinstance(meta, dict) else meta.to(device)
    label = torch.randint(0, 6, (B,), generator=g).to(device)
    crit = nn.CrossEntropyLoss(weight=torch.tensor([0.6, 1.7, 0.9, 1.2, 0.4, 2.1], device=device))
    # train_pad_20.py:54.  mmskin.optim.Adam = torch.optim.Adam with one launch for the backbone's flat parameter arena (the head's
    # parameters take torch's fused step); MMSKIN_TORCH_ADAM=1: torch's optimizer for everything (A/B)
    from mmskin.optim import Adam as ArenaAdam
    adam_cls = torch.optim.Adam if os.environ.get("MMSKIN_TORCH_ADAM", "0") == "1" else ArenaAdam
    opt = adam_cls(model.parameters(), lr=5e-5, weight_decay=1e-4, fused=True)

    def infer_step():
        with torch.no_grad():
            return model(image, meta).sum()

    # gradient all-reduce queued per encoder segment from inside backward (mmskin/dp.py); MMSKIN_DP_OVERLAP=0: one
    # all-reduce after backward
    sync = None
    if world > 1 and not args.infer and os.environ.get("MMSKIN_DP_OVERLAP", "1") != "0":
        try:
            sync = dp.OverlappedGradSync(model, world)
        except ValueError:
            sync = None

    reduced = [0]   # bytes all-reduced by this rank in the last step (diagnosis of the first multi-GPU lines)

    def step():
        if args.infer:
            return infer_step()
        opt.zero_grad(set_to_none=True)
        loss = crit(model(image, meta), label)
        loss.backward()
        if sync is not None:
            reduced[0] = sync.finish()
        elif world > 1:
            reduced[0] = dp.allreduce_gradients(model, world)
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    rank_ms = None
    if world > 1:
        mine = dt
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        tmin = torch.tensor([mine], device=device, dtype=torch.float64)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        rank_ms = {"min": round(float(tmin) / args.steps * 1e3, 3), "max": round(dt / args.steps * 1e3, 3)}
    loss_val = float(loss.detach())

    roofline, plan = None, None
    if not args.no_roofline and not args.infer:
        enc = model.image_encoder.features if hasattr(model.image_encoder, "classifier") and hasattr(model.image_encoder.features, "_plans") else model.image_encoder
        plan = next(iter(enc._plans.values())) if hasattr(enc, "_plans") else None
    # N > 1: the kernel-class times of rank 0 WITH the collectives in flight -- every rank runs the same three extra (untimed) steps, rank 0
    # with the class timers on -- so that a first multi-GPU line separates RCCL contention from straggling (the pass below is the same
    # thing without any collective)
    classes_dp = None
    if world > 1 and plan is not None:
        lib = _lib.load()
        if rank == 0:
            lib.mmskin_backbone_profile_enable(plan.handle, 1)
        for _ in range(3):
            step()
        fence()
        if rank == 0:
            ms, fl, by = (ctypes.c_double * 7)(), (ctypes.c_double * 7)(), (ctypes.c_double * 7)()
            ln = (ctypes.c_int64 * 7)()
            _lib.check(lib.mmskin_backbone_profile_read(plan.handle, ms, fl, by, ln))
            lib.mmskin_backbone_profile_enable(plan.handle, 0)
            classes_dp = {CLASS_NAMES[i]: round(ms[i] / 3, 3) for i in range(7)}
    if rank != 0:
        plan = None
    if plan_missing := (rank == 0 and not args.no_roofline and not args.infer and plan is None):
        roofline = {"note": "this workload's image encoder is a composition of HIP ops without a plan executor: no per-class timers"}
    if rank == 0 and not args.no_roofline and not args.infer and not plan_missing:
        lib = _lib.load()
        lib.mmskin_backbone_profile_enable(plan.handle, 1)
        nprof = 3
        if sync is not None:
            sync.detach()
        for _ in range(nprof):      # rank 0 only: forward + backward WITHOUT any collective (the other ranks wait at the barrier below)
            for prm in model.parameters():
                prm.grad = None
            crit(model(image, meta), label).backward()
        torch.cuda.synchronize()
        ms, fl, by = (ctypes.c_double * 7)(), (ctypes.c_double * 7)(), (ctypes.c_double * 7)()
        ln = (ctypes.c_int64 * 7)()
        _lib.check(lib.mmskin_backbone_profile_read(plan.handle, ms, fl, by, ln))
        lib.mmskin_backbone_profile_enable(plan.handle, 0)
        classes = {CLASS_NAMES[i]: {"ms_per_step": ms[i] / nprof, "launches_per_step": ln[i] // nprof,
                                    "tflops": (fl[i] / nprof) / (ms[i] / nprof * 1e-3) / 1e12 if ms[i] > 0 and fl[i] > 0 else None,
                                    "gbps": (by[i] / nprof) / (ms[i] / nprof * 1e-3) / 1e9 if ms[i] > 0 and by[i] > 0 else None}
                   for i in range(7)}
        # dominant kernel = conv_gemm_kernel (implicit GEMM): forward + dgrad launches
        dom_ms = (ms[0] + ms[1]) / nprof
        dom_fl = (fl[0] + fl[1]) / nprof
        n_launch = (ln[0] + ln[1]) // nprof
        achieved = dom_fl / (dom_ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        dom_by = (by[0] + by[1]) / nprof
        # per-launch HBM bytes from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, scripts/pmc_traffic.py); only
        # quoted when they were taken on exactly these kernel sources, else null
        traffic, traffic_note = None, "no PMC pass committed"
        frac_production = mfma_busy = prod_us = None
        try:
            with open(os.path.join(ROOT, "profiles", "conv_gemm_traffic.json")) as f:
                tj = json.load(f)
            if tj.get("source_hash") == source_hash():
                traffic, traffic_note = tj.get("hbm_bytes_per_launch"), "PMC pass on this build (profiles/conv_gemm_traffic.json)"
                # the same launches in the PRODUCTION step (side stream on: they share HBM with the weight-gradient GEMMs), from
                # the committed un-instrumented kernel trace of these sources; and the matrix-pipe busy share of the PMC pass
                prod_us = (tj.get("production_trace") or {}).get("avg_launch_us")
                if prod_us:
                    frac_production = round(dom_fl / max(n_launch, 1) / (prod_us * 1e-6) / 1e12 / peak, 4)
                mfma_busy = (tj.get("mfma_busy") or {}).get("mfma_busy_share")
            else:
                traffic_note = f"committed PMC pass is for sources {tj.get('source_hash')}, this build is {source_hash()}"
        except OSError:
            pass
        # whole-step HBM bytes (every kernel, PMC) of exactly these sources: profiles/step_traffic.json, else null
        step_bytes = step_gbps = wgrad_prod_ms = None
        try:
            with open(os.path.join(ROOT, "profiles", "step_traffic.json")) as f:
                sj = json.load(f)
            if sj.get("source_hash") == source_hash():
                step_bytes = sj.get("step_bytes")
                step_gbps = round(step_bytes / (dt / args.steps) / 1e9, 1)
        except OSError:
            pass
        if prod_us:
            wgrad_prod_ms = (tj.get("production_trace") or {}).get("wgrad_gemm_ms_per_step")
        classes["wgrad"]["production_ms_per_step"] = wgrad_prod_ms
        # frac = the PRODUCTION figure (the dominant kernel's launches as they run in the timed step, beside the weight-gradient stream:
        # committed un-instrumented trace of exactly these sources) when one is committed, else the live isolated one (side stream off)
        frac_live = round(achieved / peak, 4)
        roofline = {"bound": "mfma", "kernel": "conv_gemm_kernel + conv3x3_c64_kernel + stem7x7_kernel (convolution forward + dgrad launches)",
                    "achieved": round(achieved if frac_production is None else frac_production * peak, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": frac_live if frac_production is None else frac_production,
                    "frac_basis": "isolated (live HIP events, side stream off)" if frac_production is None else "production (committed kernel trace of these sources)",
                    "achieved_isolated": round(achieved, 2),
                    "frac_isolated": frac_live, "frac_production": frac_production,
                    "step_bytes": step_bytes, "step_gbps": step_gbps,
                    "production_avg_launch_us": prod_us, "mfma_busy_share": mfma_busy,
                    "traffic": traffic, "traffic_note": traffic_note, "launches_per_step": int(n_launch),
                    "algorithmic_gbytes_per_launch": round(dom_by / max(n_launch, 1) / 1e9, 4),
                    "algorithmic_gbps": round(dom_by / (dom_ms * 1e-3) / 1e9, 1),
                    # the same launches against the OTHER roof (their algorithmic bytes exceed what their FLOPs take at peak: hbm_floor_ms >
                    # mfma_floor_ms): algorithmic bytes per launch / launch time over the 8 TB/s HBM3E peak
                    "frac_hbm_isolated": round(dom_by / (dom_ms * 1e-3) / 8.0e12, 4),
                    "frac_hbm_production": (round(dom_by / max(n_launch, 1) / (prod_us * 1e-6) / 8.0e12, 4) if prod_us else None),
                    "hbm_floor_ms": round(dom_by / 6.3e12 * 1e3, 3), "mfma_floor_ms": round(dom_fl / (peak * 1e12) * 1e3, 3),
                    "measured_ms": round(dom_ms, 3),
                    "avg_launch_us": round(dom_ms * 1e3 / max(n_launch, 1), 2),
                    "algorithmic_gflop_per_launch": round(dom_fl / max(n_launch, 1) / 1e9, 3),
                    "classes": classes}
    if world > 1:
        dist.barrier()

    if rank == 0:
        ips = world * B * args.steps / dt
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        out = {
            "metric": wl["metric"] if not args.infer else wl["metric"].replace("fwd+bwd", "eval forward"),
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": wl["label"],
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "weights": "random init (torchvision / timm / transformers layouts)",
                       "unfreeze_weights": wl.get("unfreeze", "unfrozen_weights"), "linear_dtype": wl.get("linear_dtype", "fp32"),
                       "knobs": knobs, "source_hash": source_hash()},
            "step_tflops_per_gpu": round(ips / world * wl["flop_per_image"] / 1e12, 1),
            "step_frac_of_peak": round(ips / world * wl["flop_per_image"] / 1e12 / peak, 4),
            "loss": round(loss_val, 4),
            "roofline": roofline,
        }
        if world > 1:   # what a first multi-GPU line needs to be diagnosable: the spread over ranks and the exchange volume
            out["dp"] = {"ms_per_step_over_ranks": rank_ms, "allreduce_bytes_per_step_per_rank": int(reduced[0]),
                         "overlap": sync is not None, "grad_segments_in_flight": dp.max_inflight_segments() if sync is not None else 0,
                         "segment_lag": dp.segment_lag() if sync is not None else None,
                         # rank 0's kernel-class ms per step with the all-reduces in flight (class timers serialise the weight-gradient
                         # stream onto the main one) against roofline.classes, the same pass without any collective
                         "class_ms_with_collectives": classes_dp,
                         "backend": args.backend}
        if world == 1 and not args.no_cpu_baseline and not args.infer:
            try:
                out["cpu_baseline"] = cpu_baseline(32, workload=args.workload)   # batches of 32, not 256: bounded sample
            except ValueError as e:   # the CPU oracle model covers the torchvision backbones + one-hot / tab-transformer metadata only
                out["cpu_baseline"] = {"value": None, "note": f"no CPU oracle for this workload ({e})"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

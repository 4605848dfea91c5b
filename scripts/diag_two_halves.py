"""Would two half-batch pipelines fill the launch ramps/tails of one full-batch pipeline?  Timing experiment only (the two
halves keep separate BatchNorm statistics here): encoder forward+backward of 2 x 128 images on two streams / two host
threads against 1 x 256 on one."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
os.environ.setdefault("MMSKIN_BACKBONE_DTYPE", "bf16")
import torch
from mmskin.backbone import HipResNet
dev = "cuda:0"
K = 20

def make(B):
    enc = HipResNet("resnet-50").to(dev).train()
    x = torch.randn(B, 3, 224, 224, device=dev)
    g = torch.randn(B, 2048, device=dev)
    return enc, x, g

def run(enc, x, g, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(n):
            for p in enc.parameters():
                p.grad = None
            enc(x).backward(g)

def timed(jobs):
    for j in jobs: run(*j, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(*j, K)) for j in jobs]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

full = make(256)
ms_full = timed([(*full, torch.cuda.Stream())])
h1, h2 = make(128), make(128)
ms_one_half = timed([(*h1, torch.cuda.Stream())])
ms_two = timed([(*h1, torch.cuda.Stream()), (*h2, torch.cuda.Stream())])
print(f"encoder fwd+bwd: 1 x 256: {ms_full:.2f} ms | 1 x 128: {ms_one_half:.2f} ms | 2 x 128 concurrently: {ms_two:.2f} ms")

"""GPU box: where the HOST spends its time enqueueing one training step (cProfile over a few steps; the GPU runs ahead-of-queue, so
wall time here is host time for launch-bound workloads).  usage: host_profile.py [workload] [steps]"""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch, torch.nn as nn
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "davit-tiny-gfcam"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
W = bench.WORKLOADS[wl]
if W.get("linear_dtype"):
    from mmskin import ops
    ops.set_linear_dtype(W["linear_dtype"])
model = bench.build_model(dev, "bf16", wl).train()
B = W.get("default_batch", 256)
g = torch.Generator().manual_seed(0)
image = torch.randn(B, 3, 224, 224, generator=g).to(dev)
meta = bench.make_meta(wl, B, g)
meta = {k: v.to(dev) for k, v in meta.items()} if isinstance(meta, dict) else meta.to(dev)
label = torch.randint(0, 6, (B,), generator=g).to(dev)
crit = nn.CrossEntropyLoss()
opt = torch.optim.Adam(model.parameters(), lr=5e-5, fused=True)
def step():
    opt.zero_grad(set_to_none=True)
    crit(model(image, meta), label).backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())

"""How far ahead of the GPU does the host run?  Per-call host time of bench.step() vs synced step time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch, torch.nn as nn
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "resnet50-crossattention"
dev = "cuda:0"
model = bench.build_model(dev, "bf16", wl).train()
B = 256
g = torch.Generator().manual_seed(0)
image = torch.randn(B, 3, 224, 224, generator=g).to(dev); meta = bench.make_meta(wl, B, g).to(dev)
label = torch.randint(0, 6, (B,), generator=g).to(dev)
crit = nn.CrossEntropyLoss(weight=torch.tensor([0.6, 1.7, 0.9, 1.2, 0.4, 2.1], device=dev))
opt = torch.optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4, fused=True)
def step(marks=None):
    t = time.perf_counter
    t0 = t(); opt.zero_grad(set_to_none=True)
    t1 = t(); out = model(image, meta)
    t2 = t(); loss = crit(out, label)
    t3 = t(); loss.backward()
    t4 = t(); opt.step()
    t5 = t()
    if marks is not None: marks.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))])
for _ in range(3): step()
torch.cuda.synchronize()
marks = []
t0 = time.perf_counter()
for _ in range(8): step(marks)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{wl}: host enqueue {t_host/8*1e3:.2f} ms/step, synced {t_all/8*1e3:.2f} ms/step")
for m in marks: print("  zero_grad %.2f fwd %.2f loss %.2f bwd %.2f adam %.2f" % tuple(m))
# forward split: backbone vs head
import torch.autograd.profiler as prof

"""GPU box: upper bound of folding BatchNorm apply passes into their consumers, measured by NOT launching them (results are wrong, the
timing of everything else is what a perfect, free fusion would leave).  `make -C multimodal-model-skin-lesion-classifier_amd/csrc ablate`
library; MMSKIN_BN_ABLATE (read once per process): bit 0 = plain forward BN + ReLU applies (bn1 / bn2 of every bottleneck), bit 1 = the
BatchNorm-backward applies.  usage: MMSKIN_BN_ABLATE=v python scripts/bn_fusion_bound.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch, torch.nn as nn
from mmskin import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_ab", "libmmskin_hip_ablate.so")
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
model = bench.build_model(dev, "bf16", "resnet50-crossattention").train()
g = torch.Generator().manual_seed(1234)
image = torch.randn(256, 3, 224, 224, generator=g).to(dev)
meta = bench.make_meta("resnet50-crossattention", 256, g).to(dev)
label = torch.randint(0, 6, (256,), generator=g).to(dev)
crit = nn.CrossEntropyLoss()
opt = torch.optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4, fused=True)
def step():
    opt.zero_grad(set_to_none=True)
    crit(model(image, meta), label).backward()
    opt.step()
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"MMSKIN_BN_ABLATE={os.environ.get('MMSKIN_BN_ABLATE', '0')}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step", flush=True)

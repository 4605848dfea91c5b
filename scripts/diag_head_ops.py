"""GPU box: which autograd / aten ops launch the ~120 fill and copy kernels of one ResNet-50 + crossattention step (torch.profiler, one step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch, torch.nn as nn
from torch.profiler import profile, ProfilerActivity
os.environ.setdefault("MMSKIN_BACKBONE_DTYPE", "bf16")
from models import multimodalIntraInterModal as M
dev = "cuda:0"
model = M.MultimodalModel(num_classes=6, num_heads=8, device=dev, cnn_model_name="resnet-50", text_model_name="one-hot-encoder", common_dim=512,
                          vocab_size=20, unfreeze_weights="unfrozen_weights", attention_mecanism="crossattention").to(dev)
model.train()
opt = torch.optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4, fused=True)
crit = nn.CrossEntropyLoss()
img = torch.randn(256, 3, 224, 224, device=dev); meta = torch.randn(256, 20, device=dev); lab = torch.randint(0, 6, (256,), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model(img, meta), lab)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.count > 0]
rows.sort(key=lambda e: -e.count)
for e in rows[:45]:
    print(f"{e.key[:70]:70s} calls {e.count:5d}  cpu {e.cpu_time_total:9.0f} us  device {getattr(e, 'device_time_total', 0):9.0f} us")

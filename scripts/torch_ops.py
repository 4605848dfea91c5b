"""GPU box: which torch-side ops launch the small fill/copy kernels inside one bench step?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch, torch.nn as nn
from torch.profiler import profile, ProfilerActivity
import bench
model = bench.build_model("cuda:0", "bf16"); model.train()
B = 64
image = torch.randn(B, 3, 224, 224, device="cuda"); meta = torch.randn(B, 20, device="cuda"); label = torch.randint(0, 6, (B,), device="cuda")
crit = nn.CrossEntropyLoss(weight=torch.ones(6, device="cuda"))
opt = torch.optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4, fused=True)
def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model(image, meta), label); loss.backward(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))

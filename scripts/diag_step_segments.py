"""GPU box: where one ResNet-50 + crossattention training step (batch 256, bf16) spends its wall time on the GPU timeline, UNPROFILED:
events on the step's stream at the backbone's forward / backward boundaries (module hooks), averaged over 20 steps.
  fwd_backbone | head forward + loss + head backward (up to the backbone's backward) | bwd_backbone | rest of backward + Adam | zero_grad + gap"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch, torch.nn as nn
os.environ.setdefault("MMSKIN_BACKBONE_DTYPE", "bf16")
from models import multimodalIntraInterModal as M
dev = "cuda:0"
model = M.MultimodalModel(num_classes=6, num_heads=8, device=dev, cnn_model_name="resnet-50", text_model_name="one-hot-encoder", common_dim=512,
                          vocab_size=20, unfreeze_weights="unfrozen_weights", attention_mecanism="crossattention").to(dev)
model.train()
from mmskin.optim import Adam
opt = Adam(model.parameters(), lr=5e-5, weight_decay=1e-4, fused=True)   # as bench.py
crit = nn.CrossEntropyLoss()
img = torch.randn(256, 3, 224, 224, device=dev); meta = torch.randn(256, 20, device=dev); lab = torch.randint(0, 6, (256,), device=dev)
bb = model.image_encoder
ev = {}
def mark(k):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(k, []).append(e)
bb.register_forward_pre_hook(lambda m, a: mark("f0"))
bb.register_forward_hook(lambda m, a, o: mark("f1"))
bb.register_full_backward_pre_hook(lambda m, g: mark("b0"))
def step():
    mark("s0")
    opt.zero_grad(set_to_none=True)
    loss = crit(model(img, meta), lab)
    loss.backward()
    mark("b1")
    opt.step()
    mark("s1")
for _ in range(5): step()
torch.cuda.synchronize(); ev.clear()
import time
t0 = time.perf_counter()
for _ in range(20): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
def avg(a, b, shift=0):
    xs = [x.elapsed_time(y) for x, y in zip(ev[a][: len(ev[a]) - shift], ev[b][shift:])]
    return sum(xs) / len(xs)
print(f"wall {tw / 20 * 1e3:.2f} ms/step, host enqueue {th / 20 * 1e3:.2f} ms/step")
print(f"s0->f0 (zero_grad, metadata side) {avg('s0', 'f0'):.3f} | backbone fwd {avg('f0', 'f1'):.3f} | head fwd+loss+head bwd {avg('f1', 'b0'):.3f} | "
      f"backbone bwd + rest of autograd {avg('b0', 'b1'):.3f} | Adam {avg('b1', 's1'):.3f} | s1->next s0 {avg('s1', 's0', 1):.3f}")

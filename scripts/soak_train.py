"""GPU box: 300 training steps of ResNet-50 + one-hot + crossattention (bf16 backbone) on ONE fixed batch of 256 synthetic images: the loss must fall
(the model memorises the batch), stay finite, and the run with mmskin.optim.Adam must track the run with torch.optim.Adam (same init, same data).
Exercises every round-4 path for many steps: Gram statistics, algebraic BatchNorm backward, pooled-side stem sums, compact downsample gradient, arena Adam."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd")]
import torch, torch.nn as nn
os.environ.setdefault("MMSKIN_BACKBONE_DTYPE", "bf16")
from models import multimodalIntraInterModal as M
from mmskin.optim import Adam
dev = "cuda:0"
def run(opt_cls, steps=300):
    torch.manual_seed(0)
    model = M.MultimodalModel(num_classes=6, num_heads=8, device=dev, cnn_model_name="resnet-50", text_model_name="one-hot-encoder", common_dim=512,
                              vocab_size=20, unfreeze_weights="unfrozen_weights", attention_mecanism="crossattention").to(dev)
    model.train()
    opt = opt_cls(model.parameters(), lr=3e-4, weight_decay=1e-4, fused=True)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(256, 3, 224, 224, generator=g).to(dev); meta = torch.randn(256, 20, generator=g).to(dev); lab = torch.randint(0, 6, (256,), generator=g).to(dev)
    crit = nn.CrossEntropyLoss()
    out = []
    for it in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = crit(model(img, meta), lab)
        loss.backward()
        opt.step()
        if it % 25 == 0 or it == steps - 1:
            out.append((it, float(loss)))
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    return out, bad
a, bad_a = run(Adam)
b, bad_b = run(torch.optim.Adam)
for (i, x), (_, y) in zip(a, b):
    print(f"step {i:4d}  loss mmskin.optim.Adam {x:.4f}   torch.optim.Adam {y:.4f}")
print("non-finite parameters:", bad_a, bad_b)
assert not bad_a and not bad_b
assert a[-1][1] < 0.5 * a[0][1] and b[-1][1] < 0.5 * b[0][1], (a[0], a[-1], b[0], b[-1])
print("soak OK")

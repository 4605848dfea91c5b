"""GPU box: time bert-base (random init) forward and forward+backward on the HIP ops (B x 512 tokens)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd"), os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd", "models")]
import torch
from hip_bert import HipBertModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = HipBertModel().cuda()
ids = torch.randint(1, 30000, (B, 512), device="cuda"); mask = torch.ones_like(ids)
def run(train):
    m.train(train)
    for p in m.parameters(): p.requires_grad = train
    def step():
        if train:
            m.zero_grad(set_to_none=True)
            m(input_ids=ids, attention_mask=mask).last_hidden_state[:, 0].sum().backward()
        else:
            with torch.no_grad(): m(input_ids=ids, attention_mask=mask)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 3
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    flops = B * 512 * (12 * (4 * 768 * 768 + 2 * 768 * 3072) * 2 + 12 * 4 * 512 * 768) * (3 if train else 1)
    print(f"bert-base B={B} L=512 {'train' if train else 'eval '}: {dt*1e3:8.1f} ms  {B/dt:8.1f} seq/s  {flops/dt/1e12:6.1f} TFLOP/s", flush=True)
run(False); run(True)

"""Diagnostic (GPU box): per-parameter gradient error of the HIP ResNet path vs the CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd"), os.path.join(ROOT, "tests")]
import torch
from test_gpu_model import build_pair, _step, SMALL
from gpu_util import rel_err, DEV
from oracle.detinit import det_inputs

arch, dtype, B, HW = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
kw = dict(SMALL, cnn_model_name=arch, common_dim=512, text_encoder_dim_output=512, attention_mecanism="crossattention")
cpu, hip = build_pair(dtype, **kw)
img, meta, lab = det_inputs(B, HW, 20, 6)
out_c, loss_c, g_c = _step(cpu, img, meta, lab, "cpu")
out_h, loss_h, g_h = _step(hip, img, meta, lab, DEV)
print("logit err", float((out_c - out_h).abs().max()), "loss", loss_c, loss_h)
for k in g_c:
    e = rel_err(g_h[k], g_c[k])
    a, b = g_h[k].double().flatten(), g_c[k].double().flatten()
    l2 = float((a - b).norm() / (b.norm() + 1e-30))
    cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
    flag = " <<<<" if l2 > (2e-3 if dtype == "fp32" else 0.1) else ""
    print(f"max/rms {e:10.3e}  l2 {l2:10.3e}  cos {cos:.6f}  {k}  {tuple(g_c[k].shape)}{flag}")

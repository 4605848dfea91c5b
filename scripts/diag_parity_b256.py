"""GPU box: ONE train step (dropout off, BN batch statistics) of ResNet-50 + crossattention on the HIP path vs the fp32 CPU
oracle at a chosen batch / size / dtype; prints train-logit error, loss, running statistics, head gradients and per-stage
backbone gradient cosine / relative L2.  usage: diag_parity_b256.py [dtype=bf16] [B=256] [HW=224] [emu=0]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd"), os.path.join(ROOT, "tests")]
import torch
from test_gpu_model import build_pair, _step, _l2, _cos, SMALL, stage_report, bf16_storage_emulation
from gpu_util import rel_err, DEV
from oracle.detinit import det_init_
from oracle.model import OracleMultimodalModel

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 224
emu = len(sys.argv) > 4 and sys.argv[4] == "1"
g3 = float(sys.argv[5]) if len(sys.argv) > 5 else None     # gamma of every block's last BatchNorm (None: default init, 1.0)
kw = dict(SMALL, cnn_model_name="resnet-50", common_dim=512, text_encoder_dim_output=512, attention_mecanism="crossattention")
cpu, hip = build_pair(dtype, **kw)
if g3 is not None:
    from test_gpu_model import damp_residual_branches
    damp_residual_branches(cpu, g3); damp_residual_branches(hip, g3)
g = torch.Generator().manual_seed(0)
img, meta, lab = torch.randn(B, 3, HW, HW, generator=g), torch.randn(B, 20, generator=g), torch.randint(0, 6, (B,), generator=g)
t0 = time.time()
out_c, loss_c, g_c = _step(cpu, img, meta, lab, "cpu")
t1 = time.time()
out_h, loss_h, g_h = _step(hip, img, meta, lab, DEV)
rec = stage_report(out_c, loss_c, g_c, out_h, loss_h, g_h, cpu, hip)
rec.update(dtype=dtype, batch=B, hw=HW, g3=g3, cpu_seconds=round(t1 - t0, 1))
if emu:
    e = det_init_(OracleMultimodalModel(**dict(kw, device="cpu")))
    if g3 is not None:
        damp_residual_branches(e, g3)
    e = bf16_storage_emulation(e)
    out_e, loss_e, g_e = _step(e, img.bfloat16().float(), meta, lab, "cpu")
    rec["emu"] = stage_report(out_c, loss_c, g_c, out_e, loss_e, g_e, cpu, e)
print(json.dumps(rec, indent=1))

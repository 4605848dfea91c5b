"""Diagnostic (GPU box): compare saved activations / BN coefficients in the plan workspace with the oracle."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multimodal-model-skin-lesion-classifier_amd"), os.path.join(ROOT, "tests")]
import torch
import torch.nn as nn
from test_gpu_model import build_pair, _step, SMALL
from gpu_util import rel_err, DEV
from oracle.detinit import det_inputs
from mmskin import _lib

arch, dtype, B, HW = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
kw = dict(SMALL, cnn_model_name=arch, common_dim=512, text_encoder_dim_output=512, attention_mecanism="crossattention")
cpu, hip = build_pair(dtype, **kw)
img, meta, lab = det_inputs(B, HW, 20, 6)
acts, gacts = {}, {}
def hook(name):
    def f(mod, inp, out):
        acts[name] = out.detach()
        out.register_hook(lambda g: gacts.__setitem__(name, g.detach()))
    return f
for n, m in cpu.image_encoder.named_modules():
    if isinstance(m, nn.Conv2d):
        m.register_forward_hook(hook(n))
feat = {}
def _fh(m, i, o):
    o.register_hook(lambda g: feat.__setitem__("dfeat", g.detach()))
cpu.image_encoder.register_forward_hook(_fh)
out_c, loss_c, g_c = _step(cpu, img, meta, lab, "cpu")
out_h, loss_h, g_h = _step(hip, img, meta, lab, DEV)
torch.cuda.synchronize()
enc = hip.image_encoder
plan = next(iter(enc._plans.values()))
lib = _lib.load()
tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
es = 4 if dtype == "fp32" else 2
for i in range(lib.mmskin_backbone_num_units(plan.handle)):
    name = ctypes.create_string_buffer(128); info = (ctypes.c_int64 * 12)()
    lib.mmskin_backbone_unit_info(plan.handle, i, name, 128, info)
    x_off, y_off, coef_off, rows, C, OH, OW = [info[j] for j in range(7)]
    nm = name.value.decode().replace(".weight", "")
    x = plan.workspace[x_off:x_off + rows * C * es].view(tdt).view(B, OH, OW, C).permute(0, 3, 1, 2).float().cpu()
    print(f"{rel_err(x, acts[nm]):10.3e} x   {nm} rows={rows} C={C}")
print("dfeat rms", float(feat["dfeat"].pow(2).mean().sqrt()))

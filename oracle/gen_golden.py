"""Generate tests/golden/*.json from the REAL reference (build container only).

    python -m oracle.gen_golden

The reference holds no tests, fixtures or golden vectors,
so the pin is: import the reference's own classes from /root/reference, give
them deterministic RNG-free weights/inputs (oracle/detinit.py), run them on
CPU and record logits, loss, gradient summaries (incl. the None / exact-zero
pattern) and the first Adam step.  Fixtures are data only.
"""
import json
import os

import torch
import torch.nn as nn

from . import ref_import
from .detinit import det_init_, det_inputs, det_tensor
from .model import FUSION_STRINGS

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SMALL = dict(num_classes=6, num_heads=8, device="cpu", cnn_model_name="custom-cnn",
             text_model_name="one-hot-encoder", common_dim=64, text_encoder_dim_output=64,
             vocab_size=20, unfreeze_weights="unfrozen_weights")
CLASS_WEIGHTS = [0.6, 1.7, 0.9, 1.2, 0.4, 2.1]


def disable_dropout(model):
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0


def summarize(t):
    t = t.detach().double().flatten()
    return {"sum": float(t.sum()), "abs": float(t.abs().sum()), "head": [float(v) for v in t[:6]]}


def step_record(model, img, meta, lab):
    """eval logits, then a train step (dropout p=0) with CE(weight) + Adam."""
    model.eval()
    with torch.no_grad():
        logits_eval = model(img, meta)
    model.train()
    disable_dropout(model)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt = torch.optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4)   # train_pad_20.py:54
    crit = nn.CrossEntropyLoss(weight=torch.tensor(CLASS_WEIGHTS))            # train_pad_20.py:52
    opt.zero_grad()
    out = model(img, meta)
    loss = crit(out, lab)
    loss.backward()
    grads = {k: (None if p.grad is None else summarize(p.grad)) for k, p in model.named_parameters()}
    zero_rows = {}
    for k, p in model.named_parameters():
        if k.endswith("in_proj_weight") and p.grad is not None:
            D = p.shape[1]
            zero_rows[k] = bool((p.grad[: 2 * D] == 0).all())
    opt.step()
    delta = {k: summarize(p.detach() - before[k]) for k, p in model.named_parameters()}
    return {
        "logits_eval": logits_eval.double().tolist(),
        "logits_train": out.detach().double().tolist(),
        "loss": float(loss),
        "grads": grads,
        "qk_rows_exact_zero": zero_rows,
        "adam_delta": delta,
    }


def gen_mechanisms(ref):
    out = {}
    for mech in FUSION_STRINGS:
        kw = dict(SMALL, attention_mecanism=mech, n=1 if mech == "no-metadata" else 2)
        model = det_init_(ref["MultimodalModel"](**kw))
        img, meta, lab = det_inputs(4, 32, 20, 6)
        out[mech] = step_record(model, img, meta, lab)
    # error behaviour: ctor accepts the string, forward raises (:413-416)
    model = ref["MultimodalModel"](**dict(SMALL, attention_mecanism="metablock-se"))
    try:
        model(*det_inputs(4, 32, 20, 6)[:2])
        out["__error__metablock-se"] = None
    except ValueError as e:
        out["__error__metablock-se"] = str(e)
    return out


def gen_full_width(ref):
    """D=512 / 8 heads (the BASELINE config-2 head) on crossattention, logits only."""
    kw = dict(SMALL, common_dim=512, text_encoder_dim_output=512, attention_mecanism="crossattention")
    model = det_init_(ref["MultimodalModel"](**kw))
    img, meta, lab = det_inputs(4, 32, 20, 6)
    rec = step_record(model, img, meta, lab)
    return {"logits_eval": rec["logits_eval"], "loss": rec["loss"],
            "grads": {k: v for k, v in rec["grads"].items() if not k.startswith("image_encoder")}}


def gen_blocks(ref):
    out = {}
    mb = det_init_(ref["MetaBlock"](48, 24))
    V = det_tensor("mb.V", (5, 48))
    U = det_tensor("mb.U", (5, 24))
    V.requires_grad_(True); U.requires_grad_(True)
    y = mb(V, U); y.square().sum().backward()
    out["metablock"] = {"y": y.detach().double().tolist(), "dV": summarize(V.grad), "dU": summarize(U.grad),
                        "grads": {k: summarize(p.grad) for k, p in mb.named_parameters()}}

    g = det_init_(ref["GatedAlteredResidualBlock"](64)); g.eval()
    q = det_tensor("g.q", (1, 5, 64)).requires_grad_(True)
    k = det_tensor("g.k", (1, 5, 64)).requires_grad_(True)
    y = g(q, k, k); y.square().sum().backward()
    out["gated_residual"] = {"y": y.detach().double().tolist(), "dq": summarize(q.grad), "dk": summarize(k.grad),
                             "grads": {n: summarize(p.grad) for n, p in g.named_parameters()}}

    tt = det_init_(ref["TabTransformer"]([10] * 82, num_continuous=4, output_dim=85)); tt.eval()
    xc = (det_tensor("tt.cat", (3, 82)).abs() * 10).long().clamp_(0, 9)
    xn = det_tensor("tt.num", (3, 4))
    y = tt(xc, xn); y.square().sum().backward()
    out["tab_transformer"] = {"y": y.detach().double().tolist(),
                              "grads": {n: summarize(p.grad) for n, p in tt.named_parameters()
                                        if p.grad is not None and not n.startswith("embeddings.")},
                              "emb0": summarize(tt.embeddings[0].weight.grad)}
    return out


def gen_seed_equivalence(ref):
    """Same torch seed => same default init, key by key (construction order is API)."""
    torch.manual_seed(1234)
    m = ref["MultimodalModel"](**dict(SMALL, attention_mecanism="crossattention"))
    return {"keys": list(m.state_dict().keys()),
            "sums": {k: float(v.double().sum()) for k, v in m.state_dict().items()}}


def gen_alt_models(ref):
    """MD-Net (multimodalMDNet.py) and MetaNet+ResNet (metanet.py) through the reference's own classes.  Their
    constructors fetch torchvision / timm backbones by name (absent here): for this fixture those two
    constructor calls return this package's restated backbones, so the fixture pins the reference's head /
    fusion code and the state_dict key layout, not the third-party backbone arithmetic."""
    import importlib
    import sys
    from .altmodels import OracleResNetFeatureMaps, OracleViT
    from .backbones import OracleDenseNet169
    sys.modules["torchvision.models"].densenet169 = lambda pretrained=True: OracleDenseNet169()
    def create_model(name, pretrained=True, num_classes=0, global_pool=""):
        if name.startswith("vit_"):
            return OracleViT(name)
        return OracleResNetFeatureMaps({"resnet50": "resnet-50", "resnet18": "resnet-18"}[name])
    sys.modules["timm"].create_model = create_model
    out = {}
    img, meta, lab = det_inputs(3, 64, 20, 6)
    mdnet = importlib.import_module("multimodalMDNet").MDNet(meta_dim=20, num_classes=6, unfreeze_weights=True)
    out["mdnet"] = step_record(det_init_(mdnet), img, meta, lab)
    out["mdnet"]["keys"] = list(mdnet.state_dict().keys())
    mn = importlib.import_module("metanet").MetaNetModel(meta_dim=20, num_classes=6, image_encoder="resnet18",
                                                         unfreeze_weights=True)
    out["metanet"] = step_record(det_init_(mn), img, meta, lab)
    out["metanet"]["keys"] = list(mn.state_dict().keys())
    img224, meta, lab = det_inputs(2, 224, 20, 6)
    lw = importlib.import_module("liwtermModel").LiwTERM(num_classes=6, meta_dim=20, image_encoder="vit_tiny_patch16_224",
                                                         pretrained=False, unfreeze_backbone=True)
    out["liwterm"] = step_record(det_init_(lw), img224, meta, lab)
    out["liwterm"]["keys"] = list(lw.state_dict().keys())
    return out


def main():
    assert ref_import.available(), "reference not mounted"
    ref = ref_import.load()
    torch.set_num_threads(1)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, fn in (("mechanisms", gen_mechanisms), ("full_width", gen_full_width),
                     ("blocks", gen_blocks), ("seed_equivalence", gen_seed_equivalence),
                     ("alt_models", gen_alt_models)):
        with open(os.path.join(GOLDEN_DIR, name + ".json"), "w") as f:
            json.dump(fn(ref), f, indent=0)
        print("wrote", name)


if __name__ == "__main__":
    main()

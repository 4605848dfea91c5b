"""BERT text encoder (loadImageModelClassifier.py:170-181: AutoModel.from_pretrained("bert-base-uncased"); the model
reads last_hidden_state[:, 0, :], multimodalIntraInterModal.py:180-183) on the HIP ops vs transformers' own BertModel
on CPU (the reference's dependency; the installed version is 5.x, the reference pins 4.46.3 -- same architecture).
Random init everywhere: no checkpoint can be fetched."""
import os

import pytest
import torch

from gpu_util import DEV, rel_err
from helpers import disable_dropout
from oracle.detinit import det_init_

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")


def _pair(**cfg):
    from models.hip_bert import HipBertModel
    hf = transformers.BertModel(transformers.BertConfig(**cfg))
    det_init_(hf)
    hip = HipBertModel(**cfg)
    hip.load_state_dict(hf.state_dict(), strict=True)
    return hf, hip.to(DEV)


SMALL_CFG = dict(vocab_size=120, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 max_position_embeddings=192)


@pytest.mark.parametrize("B,L", [(3, 16), (16, 160)])     # short: LDS attention kernel; long: batched-GEMM attention + big-M Linear path
def test_bert_matches_transformers(B, L):
    hf, hip = _pair(**SMALL_CFG)
    g = torch.Generator().manual_seed(B * L)
    ids = torch.randint(1, 120, (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    mask[1, L // 2:] = 0
    hf.eval(); hip.eval()
    with torch.no_grad():
        a = hf(input_ids=ids, attention_mask=mask).last_hidden_state
        b = hip(input_ids=ids.to(DEV), attention_mask=mask.to(DEV)).last_hidden_state.cpu()
    assert rel_err(b, a) < 2e-4, rel_err(b, a)
    # training step (dropout off): gradients of every parameter the CLS feature depends on
    w = torch.randn(B, 64, generator=g)
    res = {}
    for name, m, dev in (("hf", hf, "cpu"), ("hip", hip, DEV)):
        m.train(); disable_dropout(m)
        for mod in m.modules():
            if hasattr(mod, "dropout") and isinstance(getattr(mod, "dropout"), float):
                mod.dropout = 0.0
        m.zero_grad()
        cls = m(input_ids=ids.to(dev), attention_mask=mask.to(dev)).last_hidden_state[:, 0, :]
        (cls * w.to(dev)).sum().backward()
        res[name] = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
    assert set(res["hf"]) == set(res["hip"])
    # key.bias gradients are exactly zero in exact arithmetic (a per-query constant added to every score leaves the
    # softmax unchanged): both sides hold rounding noise there, checked in absolute terms
    scale = max(float(v.abs().max()) for v in res["hf"].values())
    bad = {k: rel_err(res["hip"][k], res["hf"][k]) for k in res["hf"]
           if not k.endswith("key.bias") and rel_err(res["hip"][k], res["hf"][k]) > 2e-3}
    assert not bad, bad
    assert all(float(res["hip"][k].abs().max()) < 1e-5 * scale for k in res["hip"] if k.endswith("key.bias"))


def test_bert_factory_and_multimodal_wiring():
    from models.loadImageModelClassifier import loadModels
    model, d1, d2 = loadModels.loadTextModelEncoder("bert-base-uncased", "frozen_weights")
    assert (d1, d2) == (768, 768) and not any(p.requires_grad for p in model.parameters())
    assert sum(p.numel() for p in model.parameters()) == 109482240                     # bert-base-uncased
    model = model.to(DEV).eval()
    ids = torch.randint(1, 30000, (2, 512), device=DEV)
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=torch.ones_like(ids)).last_hidden_state
    assert out.shape == (2, 512, 768) and torch.isfinite(out).all()
    g2, d1, _ = loadModels.loadTextModelEncoder("gpt2", "unfrozen_weights")
    assert d1 == 768 and sum(p.numel() for p in g2.parameters()) == 124439808 and all(p.requires_grad for p in g2.parameters())


def test_gpt2_matches_transformers():
    """GPT-2 (HF Conv1D layout, causal attention + padding mask, gelu_new) vs transformers' GPT2Model on CPU."""
    from models.hip_gpt2 import HipGPT2Model
    cfg = dict(vocab_size=120, n_positions=192, n_embd=64, n_layer=2, n_head=4)
    hf = transformers.GPT2Model(transformers.GPT2Config(**cfg, bos_token_id=0, eos_token_id=0))
    det_init_(hf)
    hip = HipGPT2Model(**cfg)
    hip.load_state_dict(hf.state_dict(), strict=True)
    hip = hip.to(DEV)
    g = torch.Generator().manual_seed(7)
    B, L = 16, 160
    ids = torch.randint(1, 120, (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    mask[2, 100:] = 0
    w = torch.randn(B, L, 64, generator=g)
    res = {}
    for name, m, dev in (("hf", hf, "cpu"), ("hip", hip, DEV)):
        m.train(); disable_dropout(m)
        for mod in m.modules():
            for attr in ("attn_dropout", "resid_dropout", "dropout", "drop"):
                d = getattr(mod, attr, None)
                if isinstance(d, torch.nn.Dropout):
                    d.p = 0.0
        out = m(input_ids=ids.to(dev), attention_mask=mask.to(dev)).last_hidden_state
        (out * w.to(dev)).sum().backward()
        res[name] = (out.detach().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None})
    assert rel_err(res["hip"][0], res["hf"][0]) < 2e-4
    assert set(res["hip"][1]) == set(res["hf"][1])
    bad = {k: rel_err(res["hip"][1][k], v) for k, v in res["hf"][1].items() if rel_err(res["hip"][1][k], v) > 2e-3}
    assert not bad, bad

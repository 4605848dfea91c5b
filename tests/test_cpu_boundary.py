"""CPU: the C-ABI library loads and exports everything include/mmskin.h declares; the drop-in
modules mirror the reference's constructor / state_dict / error behaviour (no compute calls)."""
import ctypes

import pytest
import torch

from helpers import SMALL, golden
from mmskin import _lib
from models import multimodalIntraInterModal as M
from models.loadImageModelClassifier import loadModels
from oracle.backbones import OracleResNet
from oracle.model import FUSION_STRINGS, OracleMultimodalModel


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mmskin.h but not exported"
    assert set(names) == set(_lib._SIGNATURES), set(names) ^ set(_lib._SIGNATURES)
    assert lib.mmskin_version() >= 100


def test_plan_layout_without_gpu():
    """backbone_create needs no GPU: the flat parameter layout is the torchvision named_parameters order."""
    from mmskin.backbone import HipResNet, _Plan
    for arch, feat in (("resnet-18", 512), ("resnet-50", 2048)):
        plan = _Plan(arch, 2, 64, 64, _lib.BF16, None)
        assert plan.feat_dim == feat
        net = HipResNet(arch)
        table = plan.tensor_table(0)
        assert [t[0] for t in table] == [n for n, _ in net.named_parameters()]
        assert [t[3] for t in table] == [tuple(p.shape) for p in net.parameters()]
        assert plan.param_numel == sum(p.numel() for p in net.parameters())
        # same names / shapes as the oracle restatement of torchvision's ResNet
        ora = OracleResNet(arch)
        assert [n for n, _ in ora.named_parameters()] == [t[0] for t in table]
        bufs = [t[0] for t in plan.tensor_table(1)]
        assert bufs == [n for n, _ in ora.named_buffers() if not n.endswith("num_batches_tracked")]
        assert plan.ws_bytes > 0
    with pytest.raises(_lib.MMSkinError):
        _Plan("vgg16", 2, 64, 64, _lib.BF16, None)


def test_state_dict_keys_and_seeded_init_match_reference():
    gold = golden("seed_equivalence")
    torch.manual_seed(1234)
    m = M.MultimodalModel(**dict(SMALL, attention_mecanism="crossattention"))
    sd = m.state_dict()
    assert list(sd.keys()) == gold["keys"]
    for k, s in gold["sums"].items():
        assert abs(float(sd[k].double().sum()) - s) <= 1e-6 * max(1.0, abs(s)), k


@pytest.mark.parametrize("mech", FUSION_STRINGS)
def test_constructor_matches_oracle_for_every_mechanism(mech):
    kw = dict(SMALL, attention_mecanism=mech, n=1 if mech == "no-metadata" else 2)
    a, b = M.MultimodalModel(**kw), OracleMultimodalModel(**kw)
    assert [(k, tuple(v.shape)) for k, v in a.state_dict().items()] == \
           [(k, tuple(v.shape)) for k, v in b.state_dict().items()]


def test_resnet50_state_dict_matches_oracle_and_survives_to():
    kw = dict(SMALL, cnn_model_name="resnet-50", common_dim=512, text_encoder_dim_output=512,
              attention_mecanism="crossattention")
    torch.manual_seed(7)
    a = M.MultimodalModel(**kw)
    torch.manual_seed(7)
    b = OracleMultimodalModel(**kw)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k          # same seed => same init
    assert sum(p.numel() for p in a.parameters()) == 36543320 + 0
    enc = a.image_encoder
    assert enc._packed()
    a.double().float()                                 # _apply round trip keeps the flat arena
    assert enc._packed()
    a.load_state_dict(sb, strict=True)
    assert enc._packed()
    # positional ctor order (train_isic_2020.py:268 passes the first five positionally)
    M.MultimodalModel(6, 8, "cpu", "custom-cnn", "one-hot-encoder")


def test_error_conventions():
    with pytest.raises(ValueError, match="Backbone 'nope' não implementado."):
        loadModels.loadModelImageEncoder("nope", 64, "frozen_weights")
    with pytest.raises(ValueError, match="Invalid backbone_train_mode: false"):
        loadModels.loadModelImageEncoder("custom-cnn", 64, "false")
    with pytest.raises(ValueError, match="Text encoder 'foo' não suportado."):
        loadModels.loadTextModelEncoder("foo")
    m = M.MultimodalModel(**dict(SMALL, attention_mecanism="concatenation"))
    with pytest.raises(_lib.MMSkinError, match="no CPU fallback"):
        m(torch.zeros(2, 3, 32, 32), torch.zeros(2, 20))


def test_freeze_policy():
    for mode, expect in (("frozen_weights", 0), ("unfrozen_weights", 159), ("last_layer_unfrozen_weights", 2)):
        net, dim = loadModels.loadModelImageEncoder("resnet-50", 512, mode)
        assert dim == 2048
        assert sum(p.requires_grad for p in net.parameters()) == expect


def test_alternate_model_drop_ins_key_layout():
    """models/multimodalMDNet.py and models/metanet.py: same state_dict keys / shapes as the reference classes
    (fixture keys) and as the oracle restatements; constructible without a GPU."""
    from helpers import golden
    from models.metanet import MetaNetModel
    from models.multimodalMDNet import MDNet
    from oracle.altmodels import OracleMDNet, OracleMetaNetModel
    gold = golden("alt_models")
    hip = MDNet(meta_dim=20, num_classes=6, unfreeze_weights=True)
    ora = OracleMDNet(meta_dim=20, num_classes=6, unfreeze_weights=True)
    assert list(hip.state_dict().keys()) == gold["mdnet"]["keys"]
    assert {k: tuple(v.shape) for k, v in hip.state_dict().items()} == {k: tuple(v.shape) for k, v in ora.state_dict().items()}
    hip.load_state_dict(ora.state_dict(), strict=True)
    assert all(p.requires_grad for p in hip.parameters())
    assert not any(p.requires_grad for p in MDNet(meta_dim=20).feature_extractor.parameters())      # reference :73-75
    hip = MetaNetModel(meta_dim=20, num_classes=6, image_encoder="resnet18", unfreeze_weights=True)
    ora = OracleMetaNetModel(meta_dim=20, num_classes=6, image_encoder="resnet18", unfreeze_weights=True)
    assert list(hip.state_dict().keys()) == gold["metanet"]["keys"]
    hip.load_state_dict(ora.state_dict(), strict=True)
    import pytest
    with pytest.raises(NotImplementedError):
        MetaNetModel(meta_dim=20, image_encoder="vit_large_patch16_224")


def test_vgg16_key_layout_and_plan():
    """vgg16 branch of the factory (reference :77-81): 4096 features, torchvision key layout, last_layer freeze mode."""
    from models.loadImageModelClassifier import loadModels
    from oracle.backbones import OracleVGG16
    m, dim = loadModels.loadModelImageEncoder("vgg16", 64, "last_layer_unfrozen_weights")
    assert dim == 4096
    assert list(m.state_dict().keys()) == list(OracleVGG16().state_dict().keys())
    assert [k for k, p in m.named_parameters() if p.requires_grad] == ["classifier.3.weight", "classifier.3.bias"]
    plan = m.features._plan_for(2, 64, 64, None)          # layout check against the C plan happens inside
    assert plan.out_hw == (7, 7) and plan.feat_dim == 512


def test_mobilenet_v2_key_layout_and_plan():
    from models.loadImageModelClassifier import loadModels
    from oracle.backbones import OracleMobileNetV2
    m, dim = loadModels.loadModelImageEncoder("mobilenet-v2", 64, "last_layer_unfrozen_weights")
    assert dim == 1280
    ref = OracleMobileNetV2()
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert [k for k, p in m.named_parameters() if p.requires_grad] == ["features.18.1.weight", "features.18.1.bias"]
    assert m._plan_for(2, 96, 96, None).feat_dim == 1280      # parameter / buffer layout is checked against the C plan inside


def test_efficientnet_key_layout_and_plan():
    from models.loadImageModelClassifier import loadModels
    from oracle.backbones import OracleEfficientNet
    for name, dim in (("efficientnet-b0", 1280), ("efficientnet-b7", 2560)):
        m, d = loadModels.loadModelImageEncoder(name, 64, "frozen_weights")
        assert d == dim
        ref = OracleEfficientNet(name)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
        assert not any(p.requires_grad for p in m.parameters())
        assert m._plan_for(2, 64, 64, None).feat_dim == dim      # parameter / buffer layout checked against the C plan inside


def test_bert_key_layout_matches_transformers():
    transformers = __import__("pytest").importorskip("transformers")
    from models.hip_bert import HipBertModel
    cfg = dict(vocab_size=50, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64, max_position_embeddings=16)
    hf = transformers.BertModel(transformers.BertConfig(**cfg))
    hip = HipBertModel(**cfg)
    assert {k: tuple(v.shape) for k, v in hip.state_dict().items()} == {k: tuple(v.shape) for k, v in hf.state_dict().items()}
    hip.load_state_dict(hf.state_dict(), strict=True)


def test_gpt2_key_layout_matches_transformers():
    transformers = __import__("pytest").importorskip("transformers")
    from models.hip_gpt2 import HipGPT2Model
    cfg = dict(vocab_size=50, n_positions=16, n_embd=32, n_layer=2, n_head=2)
    hf = transformers.GPT2Model(transformers.GPT2Config(**cfg))
    hip = HipGPT2Model(**cfg)
    assert {k: tuple(v.shape) for k, v in hip.state_dict().items()} == {k: tuple(v.shape) for k, v in hf.state_dict().items()}
    hip.load_state_dict(hf.state_dict(), strict=True)
    assert hip.config.hidden_size == 32


def test_liwterm_drop_in_key_layout():
    from helpers import golden
    from models.liwtermModel import LiwTERM
    from oracle.altmodels import OracleLiwTERM
    hip = LiwTERM(num_classes=6, meta_dim=20, image_encoder="vit_tiny_patch16_224", pretrained=False, unfreeze_backbone=True)
    ora = OracleLiwTERM(num_classes=6, meta_dim=20, image_encoder="vit_tiny_patch16_224", unfreeze_backbone=True)
    assert list(hip.state_dict().keys()) == golden("alt_models")["liwterm"]["keys"]
    hip.load_state_dict(ora.state_dict(), strict=True)
    assert not any(p.requires_grad for p in LiwTERM(6, 20, image_encoder="vit_tiny_patch16_224").visual.parameters())


def test_timm_branch_beit_and_vit_layout():
    from models.loadImageModelClassifier import loadModels
    from oracle.altmodels import OracleBeit, OracleViT
    m, dim = loadModels.loadModelImageEncoder("beitv2_base_patch16_224", 64, "partial")
    assert dim == 768 and list(m.state_dict().keys()) == list(OracleBeit("beitv2_base_patch16_224").state_dict().keys())
    assert all(k.startswith("blocks.11.") for k, p in m.named_parameters() if p.requires_grad)
    m, dim = loadModels.loadModelImageEncoder("vit_small_patch16_224", 64, "frozen_weights")
    assert dim == 384 and list(m.state_dict().keys()) == list(OracleViT("vit_small_patch16_224").state_dict().keys())


def test_timm_branch_davit_layout():
    from models.loadImageModelClassifier import loadModels
    from oracle.altmodels import OracleDaVit
    m, dim = loadModels.loadModelImageEncoder("davit_tiny.msft_in1k", 64, "partial")
    assert dim == 768 and list(m.state_dict().keys()) == list(OracleDaVit().state_dict().keys())
    assert sum(p.numel() for p in m.parameters()) == 27591168                       # timm davit_tiny without its classifier
    assert all(k.startswith("stages.3.") for k, p in m.named_parameters() if p.requires_grad)
    import pytest
    with pytest.raises(ValueError):
        loadModels.loadModelImageEncoder("mvitv2_small.fb_in1k", 64, "frozen_weights")


def test_second_order_through_a_hip_op_raises_instead_of_folding_constants():
    """mmskin._autograd.no_second_order (wraps every HIP autograd.Function): first-order gradients under create_graph=True
    are returned unchanged; differentiating through them again raises MMSkinError -- torch would otherwise treat the opaque
    backward's result as a constant and return a silently wrong second-order gradient (host logic, no GPU needed)."""
    import pytest
    import torch
    from mmskin._autograd import no_second_order
    from mmskin._lib import MMSkinError

    @no_second_order
    class Cube(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return w * x ** 3

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            return (3 * w * x ** 2 * g).detach(), (x ** 3 * g).detach()     # opaque: no graph, like a kernel launch

    x = torch.tensor([1.0, -2.0], requires_grad=True)
    w = torch.tensor([0.5, 2.0], requires_grad=True)
    y = Cube.apply(x, w).sum()
    gx, gw = torch.autograd.grad(y, (x, w), create_graph=True)
    assert torch.allclose(gx, 3 * w * x ** 2) and torch.allclose(gw, x ** 3)
    for target in (x, w):                                     # anchored on EVERY differentiable input
        with pytest.raises(MMSkinError, match="second-order"):
            torch.autograd.grad(gx.sum(), target, retain_graph=True)
    Cube.apply(x, w).sum().backward()                         # the ordinary path is untouched
    assert torch.allclose(x.grad, (3 * w * x ** 2).detach()) and torch.allclose(w.grad, (x ** 3).detach())

"""-m gpu: the fused bf16 attention kernel (csrc/flash_attn.hip, C ABI mmskin_flash_attention_forward) against the unfused fp32
ops (QK^T GEMM -> softmax(+bias/mask/causal) -> dropout -> PV GEMM) and against plain torch math on the CPU.  Tolerance: bf16
operands (8 significant bits) with fp32 accumulation -- within 2e-2 (of the output rms, worst element) of an fp64 computation that rounds exactly
the kernel's bf16 operands, and within 6e-2 of exact attention on N(0,1) inputs; the dropout mask must be IDENTICAL to the
unfused path's (same counter-based generator on the [B, H, L, L] element index), which is checked by comparing both with p > 0."""
import math

import pytest
import torch

from gpu_util import DEV, rel_err
from mmskin import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_mode():
    prev = ops.get_linear_dtype()
    yield
    ops.set_linear_dtype(prev)


def _inputs(B, H, L, D, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, H, L, D, generator=g) for _ in range(3)]


def _rb(t):
    return t.bfloat16().float()


def _torch_ref(q, k, v, mask_add=None, bias=None, causal=False, bf16=False):
    """exact attention in fp64; bf16=True rounds exactly what the kernel rounds (scaled Q, K, V and the un-normalised
    probabilities relative to the row maximum) and keeps everything else exact -- the kernel must match THAT to ~1e-3."""
    if bf16:
        return _torch_ref_bf16(q, k, v, mask_add, bias, causal)
    s = (q.double() @ k.double().transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if bias is not None:
        s = s + bias.double()[None]
    if mask_add is not None:
        s = s + mask_add.double()[:, None, None, :]
    if causal:
        L = q.shape[2]
        s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float("-inf"))
    return (torch.softmax(s, -1) @ v.double()).float()


def _torch_ref_bf16(q, k, v, mask_add, bias, causal):
    s = (_rb(q).double() @ _rb(k).double().transpose(-1, -2)) / math.sqrt(q.shape[-1])      # the scale is applied to S in fp32
    if bias is not None:
        s = s + bias.double()[None]
    if mask_add is not None:
        s = s + mask_add.double()[:, None, None, :]
    if causal:
        L = q.shape[2]
        s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float("-inf"))
    e = torch.exp(s - s.amax(-1, keepdim=True))
    return ((_rb(e.float()).double() @ _rb(v).double()) / e.sum(-1, keepdim=True)).float()


@pytest.mark.parametrize("B,H,L,D,kind", [(2, 4, 197, 64, "bias"), (2, 3, 512, 64, "mask"), (3, 2, 49, 32, "plain"),
                                          (1, 2, 130, 64, "causal"), (2, 2, 77, 32, "bias"), (1, 1, 64, 64, "plain"),
                                          (2, 2, 65, 32, "mask+causal")])
def test_flash_forward_matches_unfused_and_torch(B, H, L, D, kind):
    q, k, v = _inputs(B, H, L, D, L * D)
    g = torch.Generator().manual_seed(1)
    bias = torch.randn(H, L, L, generator=g) if "bias" in kind else None
    mask = None
    if "mask" in kind:
        mask = torch.zeros(B, L)
        mask[0, L // 2:] = -10000.0                       # BERT's extended attention mask
        mask[-1, L - 3:] = float("-inf")
    causal = "causal" in kind
    want = _torch_ref(q, k, v, mask, bias, causal)
    dev = lambda t: None if t is None else t.to(DEV)
    with torch.no_grad():
        ops.set_linear_dtype("fp32")
        unfused = ops.attention(dev(q), dev(k), dev(v), mask_add=dev(mask), bias=dev(bias), causal=causal).cpu()
        ops.set_linear_dtype("bf16")
        fused = ops.attention(dev(q), dev(k), dev(v), mask_add=dev(mask), bias=dev(bias), causal=causal).cpu()
    assert rel_err(unfused, want) < 1e-4
    assert torch.isfinite(fused).all()
    emu = _torch_ref(q, k, v, mask, bias, causal, bf16=True)
    # (the online softmax rounds probabilities relative to the RUNNING maximum, the emulation relative to the final one: measured up to 1.2e-2 of the rms at its worst element, 3x closer than to exact attention)
    assert rel_err(fused, emu) < 2e-2, rel_err(fused, emu)
    assert rel_err(fused, want) < 6e-2, rel_err(fused, want)      # bf16 operands: max deviation 2-4 % of the output rms


def test_flash_fully_masked_row_is_the_uniform_average_like_torch():
    """A batch entry whose keys ALL carry the finite HuggingFace mask (finfo.min) -- torch's softmax gives the uniform average of V
    there (every score is the same huge negative number); the fused kernel's running maximum starts at -FLT_MAX so that it does too,
    and so does the unfused path (they used to disagree, the fused kernel wrote zeros)."""
    B, H, L, D = 2, 2, 130, 64
    q, k, v = _inputs(B, H, L, D, 21)
    mask = torch.zeros(B, L)
    mask[1, :] = torch.finfo(torch.float32).min
    want = _torch_ref(q, k, v, mask)
    assert rel_err(want[1], v[1].mean(1, keepdim=True).expand_as(want[1])) < 1e-5      # the reference IS the uniform average
    dev = lambda t: t.to(DEV)
    with torch.no_grad():
        ops.set_linear_dtype("fp32")
        unfused = ops.attention(dev(q), dev(k), dev(v), mask_add=dev(mask)).cpu()
        ops.set_linear_dtype("bf16")
        fused = ops.attention(dev(q), dev(k), dev(v), mask_add=dev(mask)).cpu()
    assert rel_err(unfused, want) < 1e-4
    assert rel_err(fused[1], want[1]) < 2e-2 and rel_err(fused[0], want[0]) < 6e-2


def test_flash_grid_limit_falls_back_instead_of_raising():
    """batch x heads > 65535 exceeds the fused kernel's grid (flash_attn.hip ARG_CHECK): the router must take the rows / unfused
    path (DaViT stage-1 window attention on >= 342 images is 64 windows x 3 heads per image) and give the same numbers."""
    B, H, L, D = 22000, 3, 16, 32
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B, L, 3, H, D, generator=g)
    want = _torch_ref(*(qkv[:8, :, i].permute(0, 2, 1, 3) for i in range(3))).permute(0, 2, 1, 3)
    qd = qkv.to(DEV)
    with torch.no_grad():
        ops.set_linear_dtype("bf16")
        assert not ops._flash_ok(qd[:, :, 0], qd[:, :, 1], qd[:, :, 2], None, None, B * H)
        got = ops.attention_packed(qd)
        small = ops.attention_packed(qd[:8].contiguous())       # 24 (batch, head) pairs: the fused kernel
    assert got.shape == (B, L, H, D)
    assert rel_err(got[:8].cpu(), want) < 1e-3                  # the fp32 rows kernel
    assert rel_err(small.cpu(), want) < 6e-2


def test_flash_rejects_a_misshapen_bias_or_mask():
    q, k, v = (t.to(DEV) for t in _inputs(2, 2, 70, 64, 4))
    ops.set_linear_dtype("bf16")
    from mmskin import _lib
    with torch.no_grad():
        with pytest.raises(_lib.MMSkinError):
            ops.attention(q, k, v, bias=torch.zeros(2, 70, 69, device=DEV))
        with pytest.raises(_lib.MMSkinError):
            ops.attention(q, k, v, mask_add=torch.zeros(3, 70, device=DEV))


def test_dropout_counter_ranges_of_successive_calls_do_not_overlap():
    """Call i consumes counters [c, c + n_i): a second, smaller call must not re-use a slice of the first one's range."""
    ops._dropout_counter[0] = 500
    torch.manual_seed(7)
    s1, o1 = ops._dropout_state(0.1, 1000)
    s2, o2 = ops._dropout_state(0.1, 10)
    assert (o1, o2) == (500, 1500) and ops._dropout_counter[0] == 1510 and s1 == s2
    x = torch.ones(4096, device=DEV)
    ops._dropout_counter[0] = 0
    a = ops.dropout(x, 0.5, True)
    b = ops.dropout(x[:1024].contiguous(), 0.5, True)
    ops._dropout_counter[0] = 4096
    c = ops.dropout(x[:1024].contiguous(), 0.5, True)
    assert torch.equal(b, c) and not torch.equal(b, a[:1024])    # the second call continues where the first one ended


def test_flash_dropout_drops_the_same_elements_as_the_unfused_path():
    B, H, L, D = 2, 3, 200, 64
    q, k, v = (t.to(DEV) for t in _inputs(B, H, L, D, 9))
    outs = {}
    with torch.no_grad():
        for mode in ("fp32", "bf16"):
            ops.set_linear_dtype(mode)
            torch.manual_seed(1234)
            ops._dropout_counter[0] = 1000                 # same generator position for both calls
            outs[mode] = ops.attention(q, k, v, 0.3, True).cpu()
        ops.set_linear_dtype("bf16")
        torch.manual_seed(1234)
        ops._dropout_counter[0] = 77777
        other = ops.attention(q, k, v, 0.3, True).cpu()
        nodrop = ops.attention(q, k, v, 0.0, True).cpu()
    assert rel_err(outs["bf16"], outs["fp32"]) < 6e-2      # identical mask, bf16 operands
    assert rel_err(other, outs["fp32"]) > 0.2               # another generator position gives another mask
    assert rel_err(nodrop, outs["fp32"]) > 0.2


def _attn_grads(fn, q, k, v, w, bias):
    qs = [t.clone().requires_grad_(True) for t in (q, k, v)]
    bs = bias.clone().requires_grad_(True) if bias is not None else None
    (fn(*qs, bs) * w).sum().backward()
    return [t.grad.detach().double().cpu() for t in qs] + ([bs.grad.detach().double().cpu()] if bs is not None else [])


@pytest.mark.parametrize("B,H,L,D,kind", [(2, 3, 197, 64, "bias"), (2, 2, 512, 64, "mask"), (1, 2, 130, 32, "causal"), (2, 2, 100, 64, "plain"),
                                          (1, 2, 77, 32, "mask+causal+bias"), (1, 1, 65, 64, "plain")])
def test_flash_backward_matches_fp64_autograd_and_the_unfused_path(B, H, L, D, kind, monkeypatch):
    """Trainable attention in bf16-operand mode: forward with the row log-sum-exp kept + the recomputing backward (csrc/flash_attn_bwd.hip)
    against (a) torch autograd through exact attention in fp64 -- bf16 operands: every gradient within 8e-2 of its rms at the worst
    element and 2e-2 in relative L2 -- and (b) the unfused fp32 chain of this package (MMSKIN_FLASH_BWD=0), which is (a) to 1e-3.  Covers the relative
    position bias and its gradient (BEiT: summed over the batch), key masks with -inf tails, the causal mask, ragged last tiles and a
    sequence one token past the tile (65)."""
    q, k, v = _inputs(B, H, L, D, 3 * L + D)
    g = torch.Generator().manual_seed(2)
    w = torch.randn(B, H, L, D, generator=g)
    bias = torch.randn(H, L, L, generator=g) if "bias" in kind else None
    mask = None
    if "mask" in kind:
        mask = torch.zeros(B, L)
        mask[0, L // 2:] = -10000.0
        mask[-1, L - 3:] = float("-inf")
    causal = "causal" in kind
    want = _attn_grads(lambda a, b_, c, bb: _torch_ref64(a, b_, c, mask, bb, causal), q.double(), k.double(), v.double(), w.double(),
                       bias.double() if bias is not None else None)
    dev = lambda t: None if t is None else t.to(DEV)
    ops.set_linear_dtype("bf16")
    fused = _attn_grads(lambda a, b_, c, bb: ops.attention(a, b_, c, mask_add=dev(mask), bias=bb, causal=causal), dev(q), dev(k), dev(v), dev(w), dev(bias))
    monkeypatch.setenv("MMSKIN_FLASH_BWD", "0")
    unfused = _attn_grads(lambda a, b_, c, bb: ops.attention(a, b_, c, mask_add=dev(mask), bias=bb, causal=causal), dev(q), dev(k), dev(v), dev(w), dev(bias))
    for name, gf, gu, gw in zip(("dq", "dk", "dv", "dbias"), fused, unfused, want):
        assert torch.isfinite(gf).all(), name
        assert rel_err(gu, gw) < 2e-3, (name, "unfused", rel_err(gu, gw))
        if name == "dbias":      # a sum of dS over the batch: heavy-tailed (a few elements are 10 - 30 x the rms), so the bf16-operand
            # error is bounded against the largest element and in relative L2 instead of max-error / rms
            assert float((gf - gw).abs().max()) < 2e-2 * float(gw.abs().max()), (name, float((gf - gw).abs().max()), float(gw.abs().max()))
            assert float((gf - gw).norm() / gw.norm()) < 3e-2, (name, float((gf - gw).norm() / gw.norm()))
        else:                    # two more bf16-rounded MFMA operands than the forward (dS and P^T): 8e-2 of the rms at the worst element
            # (measured 4e-2 - 6.3e-2), relative L2 within 2e-2
            assert rel_err(gf, gw) < 8e-2, (name, "fused", rel_err(gf, gw))
            assert float((gf - gw).norm() / gw.norm()) < 2e-2, (name, float((gf - gw).norm() / gw.norm()))


def _torch_ref64(q, k, v, mask_add, bias, causal):
    s = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if bias is not None:
        s = s + bias[None]
    if mask_add is not None:
        s = s + mask_add.double()[:, None, None, :]
    if causal:
        L = q.shape[2]
        s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float("-inf"))
    return torch.softmax(s, -1) @ v


def test_flash_backward_regenerates_the_forward_dropout_mask(monkeypatch):
    """Dropout on the probabilities: the fused backward must rebuild exactly the mask the fused forward drew (counter-based generator on
    the [B, H, L, L] element index) -- its gradients equal the unfused path's at the same generator position to bf16 accuracy, and a
    different position gives different gradients."""
    B, H, L, D = 2, 2, 200, 64
    q, k, v = (t.to(DEV) for t in _inputs(B, H, L, D, 11))
    w = torch.randn(B, H, L, D, generator=torch.Generator().manual_seed(4)).to(DEV)
    ops.set_linear_dtype("bf16")

    def grads(pos):
        torch.manual_seed(99)
        ops._dropout_counter[0] = pos
        return _attn_grads(lambda a, b_, c, bb: ops.attention(a, b_, c, 0.25, True), q, k, v, w, None)

    fused = grads(4096)
    other = grads(123456)
    monkeypatch.setenv("MMSKIN_FLASH_BWD", "0")
    unfused = grads(4096)
    for name, gf, gu, go in zip(("dq", "dk", "dv"), fused, unfused, other):
        assert rel_err(gf, gu) < 6e-2, (name, rel_err(gf, gu))
        assert rel_err(go, gu) > 0.2, name


def test_flash_reads_a_fused_qkv_tensor_in_place():
    """attention_blhd on slices of the [B, L, 3, H, Dh] output of a qkv Linear: no permute copies, token-major output."""
    B, L, H, D = 3, 197, 4, 64
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, L, 3, H, D, generator=g)
    want = _torch_ref(*(qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))).permute(0, 2, 1, 3)
    qd = qkv.to(DEV)
    with torch.no_grad():
        ops.set_linear_dtype("bf16")
        got = ops.attention_blhd(qd[:, :, 0], qd[:, :, 1], qd[:, :, 2])
        assert got.shape == (B, L, H, D) and got.is_contiguous()
        assert rel_err(got.cpu(), want) < 6e-2
        ops.set_linear_dtype("fp32")                                        # unfused fallback of the same entry
        assert rel_err(ops.attention_blhd(qd[:, :, 0], qd[:, :, 1], qd[:, :, 2]).cpu(), want) < 1e-4


def test_trainable_attention_keeps_the_differentiable_path():
    """Gradients must keep flowing: with requires_grad inputs bf16 mode uses the unfused ops (which save the probabilities)."""
    ops.set_linear_dtype("bf16")
    q, k, v = (t.to(DEV).requires_grad_(True) for t in _inputs(1, 2, 70, 64, 3))
    o = ops.attention(q, k, v)
    o.sum().backward()
    assert q.grad is not None and torch.isfinite(q.grad).all() and float(q.grad.abs().max()) > 0


@pytest.mark.parametrize("M,K,N", [(4096, 256, 512), (2500, 128, 64), (64, 96, 40)])
def test_linear_gelu_epilogue(M, K, N):
    """gelu(x W^T + b) fused into the GEMM epilogue (no-grad path) = the separate ops, in both operand modes and on the small
    fp32 GEMM path; with gradients the unfused pair runs and differentiates."""
    g = torch.Generator().manual_seed(M)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    want = torch.nn.functional.gelu(x.double() @ w.double().T + b.double()).float()
    for mode, tol in (("fp32", 1e-4), ("bf16", 3e-2)):
        ops.set_linear_dtype(mode)
        with torch.no_grad():
            fused = ops.linear_gelu(x.to(DEV), w.to(DEV), b.to(DEV)).cpu()
            pair = ops.gelu(ops.linear(x.to(DEV), w.to(DEV), b.to(DEV))).cpu()
        assert rel_err(fused, want) < tol, (mode, rel_err(fused, want))
        assert rel_err(fused, pair) < (1e-5 if mode == "fp32" else 1e-2)     # bf16: the unfused pair rounds the pre-activation to bf16 too
    ops.set_linear_dtype("fp32")
    xg = x.to(DEV).requires_grad_(True)
    ops.linear_gelu(xg, w.to(DEV), b.to(DEV)).sum().backward()
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.gelu(xr @ w.T + b).sum().backward()
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-3


def test_flash_bf16_tensors_and_lane_ops():
    """The inference lane's bf16 tensors: attention on bf16 q / k / v views (bf16 out), Linear with bf16 in / out and LayerNorm with a
    bf16 (and fp32) result -- each against fp64 math on the values those tensors actually hold."""
    ops.set_linear_dtype("bf16")
    B, L, H, D = 2, 197, 4, 64
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B, L, 3, H, D, generator=g).bfloat16()
    want = _torch_ref(*(qkv[:, :, i].float().permute(0, 2, 1, 3) for i in range(3))).permute(0, 2, 1, 3)
    qd = qkv.to(DEV)
    with torch.no_grad():
        got = ops.attention_blhd(qd[:, :, 0], qd[:, :, 1], qd[:, :, 2])
    assert got.dtype == torch.bfloat16 and got.shape == (B, L, H, D)
    assert rel_err(got.float().cpu(), want) < 3e-2
    # Linear: bf16 in -> bf16 out with GELU, bf16 in -> fp32 out, small shape through the fallback
    for M, K, N in ((4096, 256, 512), (96, 64, 40)):
        x = torch.randn(M, K, generator=g).bfloat16()
        w, b = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
        ref = x.double() @ w.double().T + b.double()
        with torch.no_grad():
            y32 = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV))
            y16 = ops.linear_gelu(x.to(DEV), w.to(DEV), b.to(DEV), out_dtype=torch.bfloat16)
        assert y32.dtype == torch.float32 and y16.dtype == torch.bfloat16
        assert rel_err(y32.cpu(), ref.float()) < 2e-2
        assert rel_err(y16.float().cpu(), torch.nn.functional.gelu(ref).float()) < 3e-2
    # LayerNorm: one pass, fp32 + bf16 results
    x = torch.randn(300, 768, generator=g)
    gam, bet = torch.rand(768, generator=g) + 0.5, torch.randn(768, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (768,), gam.double(), bet.double(), 1e-6).float()
    with torch.no_grad():
        y32, y16 = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), 1e-6, out_dtype=torch.bfloat16, keep_f32=True)
    assert rel_err(y32.cpu(), ref) < 1e-5 and y16.dtype == torch.bfloat16 and rel_err(y16.float().cpu(), ref) < 2.5e-2   # 2^-9 of |y| <= 4
    # a gradient-carrying input is refused on the lane instead of silently detached
    with pytest.raises(Exception):
        ops.linear(x.to(DEV).bfloat16().requires_grad_(True), w.to(DEV)[:, :768] if w.shape[1] >= 768 else torch.randn(8, 768, device=DEV))


def test_bert_bf16_lane_matches_transformers():
    """HipBertModel on the inference lane (bf16-operand mode, frozen: bf16 activations between Linear / LayerNorm / fused attention)
    against transformers' BertModel in fp32 -- 512 tokens, padded rows."""
    transformers = pytest.importorskip("transformers")
    from models.hip_bert import HipBertModel
    from oracle.detinit import det_init_
    cfg = dict(vocab_size=200, hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512, max_position_embeddings=512)
    hf = det_init_(transformers.BertModel(transformers.BertConfig(**cfg))).eval()
    hip = HipBertModel(**cfg)
    hip.load_state_dict(hf.state_dict(), strict=True)
    hip = hip.to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1, 200, (6, 512), generator=g)
    mask = torch.ones(6, 512, dtype=torch.long); mask[1, 300:] = 0; mask[4, 17:] = 0
    with torch.no_grad():
        want = hf(input_ids=ids, attention_mask=mask).last_hidden_state[:, 0]
        ops.set_linear_dtype("bf16")
        got = hip(input_ids=ids.to(DEV), attention_mask=mask.to(DEV)).last_hidden_state[:, 0].cpu()
        ops.set_linear_dtype("fp32")
        exact = hip(input_ids=ids.to(DEV), attention_mask=mask.to(DEV)).last_hidden_state[:, 0].cpu()
    assert rel_err(exact, want) < 5e-4
    assert rel_err(got, want) < 5e-2, rel_err(got, want)


def test_linear_lane_fused_tail_matches_separate_ops():
    """mmskin_linear_lane: y = residual + gamma * dropout(act(x W^T + b)) in the GEMM epilogue against the same chain of separate
    ops (Linear on the lane -> dropout op -> scale_add): identical dropout mask (same generator, same element index), bf16-rounded
    GEMM result in both, so the two agree to fp32 rounding; stacked weights = one GEMM over the concatenation; the cached bf16
    weight follows in-place updates of its source; shapes off the fused path are composed from the separate ops."""
    ops.set_linear_dtype("bf16")
    g = torch.Generator().manual_seed(21)
    M, K, N = 4096 + 40, 256, 384                       # ragged last row block
    x = torch.randn(M, K, generator=g).bfloat16().to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b, gam = torch.randn(N, generator=g).to(DEV), (torch.rand(N, generator=g) + 0.5).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    with torch.no_grad():
        for act, p, use_g, use_r in ((0, 0.0, True, True), (2, 0.0, False, True), (0, 0.1, False, True), (1, 0.25, True, True), (0, 0.1, False, False)):
            state = list(ops._dropout_counter)
            y = ops.linear_lane(x, w, b, act, gam if use_g else None, res if use_r else None, p, True)
            ops._dropout_counter[:] = state               # replay the same dropout call for the separate ops
            h = ops.linear_lane(x, w, b, act)                                  # fp32 out, plain epilogue
            if p > 0:
                h = ops.dropout(h, p, True)
            want = h * (gam if use_g else 1.0) + (res if use_r else 0.0)
            assert y.dtype == torch.float32 and y.shape == (M, N)
            assert rel_err(y.cpu(), want.cpu()) < 1e-6, (act, p, use_g, use_r, rel_err(y.cpu(), want.cpu()))
            if p > 0 and act == 0:
                assert abs(float((h == 0).float().mean()) - p) < 0.01
        # against fp64 math on the bf16 operands
        ref = res.double().cpu() + gam.double().cpu() * (x.double().cpu() @ w.bfloat16().double().cpu().T + b.double().cpu())
        assert rel_err(ops.linear_lane(x, w, b, 0, gam, res).cpu(), ref.float()) < 2e-2
        # stacked weights: one GEMM over the concatenation
        w2 = (torch.randn(128, K, generator=g) / K ** 0.5).to(DEV)
        b2 = torch.randn(N + 128, generator=g).to(DEV)
        ys = ops.linear_lane(x, (w, w2), b2, out_dtype=torch.bfloat16)
        yc = ops.linear_lane(x, torch.cat([w, w2]), b2, out_dtype=torch.bfloat16)
        assert ys.dtype == torch.bfloat16 and torch.equal(ys, yc)
        # the cache follows an in-place update of the source weight
        y0 = ops.linear_lane(x, w, b)
        w.mul_(2.0)
        y1 = ops.linear_lane(x, w, b)
        assert rel_err((y1 - b).cpu(), (2.0 * (y0 - b)).cpu()) < 1e-2
        # off the fused path (few rows; N not a multiple of 128 with a tail): composed from the separate ops
        xs = torch.randn(96, 64, generator=g).to(DEV)
        ws, rs = torch.randn(40, 64, generator=g).to(DEV), torch.randn(96, 40, generator=g).to(DEV)
        ysm = ops.linear_lane(xs, ws, None, 0, None, rs)
        assert rel_err(ysm.cpu(), (xs @ ws.T + rs).cpu()) < 2e-2
    with pytest.raises(Exception):
        ops.linear_lane(x.float().requires_grad_(True), w, b)


@pytest.mark.parametrize("B,H,L,Dh,p", [(3, 3, 49, 32, 0.0), (2, 2, 64, 64, 0.0), (5, 1, 7, 32, 0.0), (2, 3, 49, 32, 0.2)])
def test_attention_rows_kernels_match_torch(B, H, L, Dh, p):
    """One-wave-per-head fp32 attention (mmskin_attention_rows_*): forward and all three gradients against float64 torch math, through
    both layouts -- separate [B, H, L, Dh] tensors (ops.attention) and the packed [B, L, 3, H, Dh] qkv tensor read in place
    (ops.attention_packed).  With dropout: same mask as the unfused path (compared through the long-sequence ops on the same call)."""
    g = torch.Generator().manual_seed(31 + L)
    qkv = torch.randn(B, L, 3, H, Dh, generator=g)
    dO = torch.randn(B, L, H, Dh, generator=g)

    def ref(qkv64):
        q, k, v = (qkv64[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        pr = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(Dh), -1)
        return (pr @ v).permute(0, 2, 1, 3)
    if p == 0.0:
        r = qkv.double().requires_grad_(True)
        want = ref(r)
        want.backward(dO.double())
        # packed layout
        x = qkv.to(DEV).requires_grad_(True)
        got = ops.attention_packed(x)
        assert got.shape == (B, L, H, Dh)
        got.backward(dO.to(DEV))
        assert rel_err(got.detach().cpu(), want.detach().float()) < 1e-5
        assert rel_err(x.grad.cpu(), r.grad.float()) < 2e-5
        # separate [B, H, L, Dh] tensors
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).contiguous().to(DEV).requires_grad_(True) for i in range(3))
        o2 = ops.attention(q, k, v)
        o2.backward(dO.permute(0, 2, 1, 3).contiguous().to(DEV))
        assert rel_err(o2.detach().permute(0, 2, 1, 3).cpu(), want.detach().float()) < 1e-5
        for i, t in enumerate((q, k, v)):
            assert rel_err(t.grad.permute(0, 2, 1, 3).cpu(), r.grad[:, :, i].float()) < 2e-5
    else:
        # dropout: the packed path and the [B, H, L, Dh] path draw element ((b*H + h)*L + i)*L + j of their call -- replay one call for both
        state = list(ops._dropout_counter)
        x = qkv.to(DEV).requires_grad_(True)
        got = ops.attention_packed(x, p, True)
        got.backward(dO.to(DEV))
        ops._dropout_counter[:] = state
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).contiguous().to(DEV).requires_grad_(True) for i in range(3))
        seed, offset = ops._dropout_state(p, B * H * L * L)
        o2 = ops.LongAttentionFn.apply(q, k, v, None, p, seed, offset, None, False)      # unfused GEMM -> softmax -> dropout -> GEMM chain
        o2.backward(dO.permute(0, 2, 1, 3).contiguous().to(DEV))
        assert rel_err(got.detach().cpu(), o2.detach().permute(0, 2, 1, 3).cpu()) < 1e-5
        for i, t in enumerate((q, k, v)):
            assert rel_err(x.grad[:, :, i].cpu(), t.grad.permute(0, 2, 1, 3).cpu()) < 2e-5
        frac = float((got.detach() != ops.attention_packed(x.detach())).float().mean())
        assert frac > 0.5      # dropout really changed the result


@pytest.mark.parametrize("M,K,N", [(4096 + 24, 96, 288), (2048, 288, 96), (3000, 384, 96), (2048, 40, 72)])
def test_linear_bf16_padded_widths(M, K, N):
    """bf16-operand Linear with widths that are not multiples of 64 (DaViT stage 1: 96 / 288): zero-padded bf16 operand copies on the
    MFMA GEMM kernels, un-padded while widening -- forward (+ bias, ReLU) and dx / dw / db against float64 math on the bf16-rounded
    operands (the pad columns must contribute exact zeros)."""
    ops.set_linear_dtype("bf16")
    g = torch.Generator().manual_seed(M + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xr, wr = x.bfloat16().double(), w.bfloat16().double()
    for relu in (False, True):
        xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        y = ops.linear(xd, wd, bd, relu)
        y.backward(dy.to(DEV))
        ref = xr @ wr.T + b.double()
        gg = dy.double() * ((y.detach().cpu() > 0) if relu else 1.0)     # the ReLU mask the backward really used (sign flips next to 0 are rounding)
        if relu:
            ref = ref.clamp_min(0)
        assert rel_err(y.detach().cpu(), ref.float()) < 2.5e-2     # the GEMM result is staged in bf16 before bias / ReLU: 2^-9 of |y| <= 4
        gr = gg.bfloat16().double()                      # the gradient operand is rounded to bf16 as well
        assert rel_err(xd.grad.cpu(), (gr @ wr).float()) < 2e-2
        assert rel_err(wd.grad.cpu(), (gr.T @ xr).float()) < 2e-2
        assert rel_err(bd.grad.cpu(), gg.sum(0).float()) < 1e-4

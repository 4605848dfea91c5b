// ResNet image-encoder plan executor (resnet-18 / resnet-50, torchvision v1.5 layout).
//
// One C-ABI call runs the whole backbone forward (or backward) by launching the gfx950 kernels of
// conv_gemm.hip / wgrad.hip / ops.hip back to back on the caller's stream: no Python per layer, no
// allocation, graph-capturable.  Replaces `self.image_encoder(image)` of the reference
// (multimodalIntraInterModal.py:167; factory loadImageModelClassifier.py:65-75) and its autograd
// backward.  Parameters stay fp32 masters in one flat buffer laid out in torchvision's
// named_parameters() order; they are re-staged to the compute dtype (K-contiguous GEMM operands) at
// the start of every forward.
#include "plan.h"

namespace {

struct Unit {  // conv + batch-norm
  ConvShape s;
  bool stem = false;
  int64_t w_off, g_off, b_off;   // flat param offsets
  int64_t rm_off, rv_off;        // flat buffer offsets
  int64_t wf_off, wd_off;        // staged weight element offsets
  size_t x_off;                  // raw conv output (bytes in workspace)
  size_t y_off;                  // post-activation output (bytes); for the last unit of a block = block output
  size_t coef_off;               // floats: scale, shift, mean, invstd (4*Cout)
  bool abn = false;              // algebraic BatchNorm backward (abn.hip): expanding 1x1 conv3 of a bottleneck, bf16 plans
  size_t gram_off = 0;           // two-pass unit with Gram statistics: y^T y [128][Cin] + colsum(y) [Cin] of its input, forward -> backward
  bool fwd2p = false;            // ... and its raw output x never stored: two-pass forward (statistics, then conv + BatchNorm + residual + ReLU in the
                                 // epilogue), BatchNorm-backward x sums from the weight-gradient GEMM (blocks without a downsample branch)
  size_t abn_coef_off = 0;       // this unit's copy of cA | cB | cC (3*Cout floats) for the weight-gradient stream
  size_t rows() const { return (size_t)s.N * s.OH() * s.OW(); }
};

struct Block {
  std::vector<int> units;
  int ds = -1;
  size_t mask_off = 0;   // 1-bit ReLU mask of the block output (one byte per 16-byte chunk)
  size_t in_off;   // block input activation (bytes)
  int in_C, in_H, in_W;
  int seg_done = -1;   // gradient segment that is complete once this block's backward has been enqueued
};

struct Plan : PlanBase {
  int arch;
  int Hp, Wp, OH0, OW0, PH, PW;
  std::vector<Unit> units;
  std::vector<Block> blocks;
  int64_t staged_elems = 0;
  // workspace layout (bytes)
  size_t off_img4, off_wf, off_wd, off_stat, off_pool, off_idx, off_scratch[7], off_slab, off_partial,
      off_coefbwd, off_coefbwd_b, off_dwv, off_red, off_partial_b, off_stat_b, off_red_b;
  size_t off_abn_wd = 0, off_abn_bias = 0, off_abn_S = 0, off_abn_cs = 0;   // algebraic BatchNorm backward scratch
  size_t off_abn_wd2 = 0, off_abn_bias2 = 0;                                 // ... of a stride-1 downsample convolution (its folded weights live beside conv3's)
  size_t off_abn_sgx = 0, off_abn_slab2 = 0, off_abn_S2 = 0, off_abn_cs2 = 0;   // two-pass units: main-stream weight-gradient scratch
  size_t maxact_bytes = 0, stat_bytes = 0;

  int forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
              float* features, bool training, hipStream_t st) override;
  int backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) override;
  int last_conv_shape(int* C, int* OH, int* OW) const override;
  int last_conv_export(const unsigned char* ws, float* x_nchw, hipStream_t st) override;
  int last_conv_grad(const float* dfeat, const unsigned char* ws, float* dx_nchw, hipStream_t st) override;
  int num_units() const override { return (int)units.size(); }
  int unit_info(int index, std::string* name, int64_t* info12) const override;
};

int add_unit(Plan& p, const std::string& conv_name, const std::string& bn_name, ConvShape s) {
  Unit u;
  u.s = s;
  u.w_off = add_tensor(p.params, p.param_numel, conv_name + ".weight", {s.Cout, s.Cin, s.kh, s.kw});
  u.g_off = add_tensor(p.params, p.param_numel, bn_name + ".weight", {s.Cout});
  u.b_off = add_tensor(p.params, p.param_numel, bn_name + ".bias", {s.Cout});
  u.rm_off = add_tensor(p.buffers, p.buffer_numel, bn_name + ".running_mean", {s.Cout});
  u.rv_off = add_tensor(p.buffers, p.buffer_numel, bn_name + ".running_var", {s.Cout});
  p.units.push_back(u);
  return (int)p.units.size() - 1;
}

int build_plan(Plan& p) {
  const bool bottleneck = p.arch == 50;
  const int depths18[4] = {2, 2, 2, 2}, depths50[4] = {3, 4, 6, 3};
  const int* depths = bottleneck ? depths50 : depths18;
  const int expansion = bottleneck ? 4 : 1;
  // stem
  ConvShape s0 = {p.N, p.H, p.W, 3, 64, 7, 7, 2, 3};
  int u0 = add_unit(p, "conv1", "bn1", s0);
  p.units[u0].stem = true;
  p.OH0 = s0.OH(); p.OW0 = s0.OW();
  p.Hp = 2 * p.OH0 + 8; p.Wp = 2 * p.OW0 + 8;
  if (p.Hp < p.H + 6) p.Hp = p.H + 6;
  if (p.Wp < p.W + 6) p.Wp = p.W + 6;
  p.Wp = (p.Wp + 1) / 2 * 2;
  p.PH = (p.OH0 + 2 - 3) / 2 + 1; p.PW = (p.OW0 + 2 - 3) / 2 + 1;
  int cin = 64, h = p.PH, w = p.PW;
  const int widths[4] = {64, 128, 256, 512};
  std::vector<int64_t> seg_starts;   // first parameter of layer2, layer3, layer4
  for (int li = 0; li < 4; ++li) {
    for (int b = 0; b < depths[li]; ++b) {
      const int stride = (b == 0 && li > 0) ? 2 : 1;
      const int width = widths[li], cout = width * expansion;
      std::string base = "layer" + std::to_string(li + 1) + "." + std::to_string(b);
      Block blk;
      blk.in_C = cin; blk.in_H = h; blk.in_W = w;
      if (bottleneck) {
        blk.units.push_back(add_unit(p, base + ".conv1", base + ".bn1", {p.N, h, w, cin, width, 1, 1, 1, 0}));
        blk.units.push_back(add_unit(p, base + ".conv2", base + ".bn2", {p.N, h, w, width, width, 3, 3, stride, 1}));
        int h2 = (h + 2 - 3) / stride + 1, w2 = (w + 2 - 3) / stride + 1;
        blk.units.push_back(add_unit(p, base + ".conv3", base + ".bn3", {p.N, h2, w2, width, cout, 1, 1, 1, 0}));
      } else {
        blk.units.push_back(add_unit(p, base + ".conv1", base + ".bn1", {p.N, h, w, cin, width, 3, 3, stride, 1}));
        int h2 = (h + 2 - 3) / stride + 1, w2 = (w + 2 - 3) / stride + 1;
        blk.units.push_back(add_unit(p, base + ".conv2", base + ".bn2", {p.N, h2, w2, width, cout, 3, 3, 1, 1}));
      }
      if (stride != 1 || cin != cout)
        blk.ds = add_unit(p, base + ".downsample.0", base + ".downsample.1", {p.N, h, w, cin, cout, 1, 1, stride, 0});
      if (b == 0 && li > 0) {   // layerN's parameters start here: everything from this offset on completes first in backward
        blk.seg_done = 3 - li;
        seg_starts.push_back(p.units[blk.units[0]].w_off);
      }
      p.blocks.push_back(blk);
      cin = cout;
      h = (h + 2 - 3) / stride + 1;
      w = (w + 2 - 3) / stride + 1;
    }
  }
  p.feat_dim = cin;
  {   // gradient segments in backward completion order: layer4, layer3, layer2, layer1 + stem
    int64_t end = p.param_numel;
    for (int i = 2; i >= 0; --i) {
      PlanBase::GradSegment g;
      g.offset = seg_starts[i]; g.numel = end - seg_starts[i];
      p.segments.push_back(g);
      end = seg_starts[i];
    }
    PlanBase::GradSegment g;
    g.offset = 0; g.numel = end;
    p.segments.push_back(g);
  }

  // ---- staged weights + stage table
  std::vector<StageDesc> table;
  int64_t wf = 0, wd = 0;
  for (Unit& u : p.units) {
    StageDesc d = {};
    d.src_off = u.w_off;
    d.Cout = u.s.Cout; d.Cin = u.s.Cin; d.taps = u.s.kh * u.s.kw; d.stem = u.stem ? 1 : 0;
    u.wf_off = wf; u.wd_off = wd;
    d.fwd_off = wf; d.dgrad_off = wd;
    int64_t n = u.stem ? 64 * 256 : (int64_t)d.Cout * d.Cin * d.taps;
    wf += n;
    if (!u.stem) wd += n;
    if (d.Cout * d.Cin * d.taps > p.max_stage_elems) p.max_stage_elems = d.Cout * d.Cin * d.taps;
    table.push_back(d);
  }
  p.staged_elems = wf;
  p.table_host = table;  // uploaded by ensure_table() on the first forward (create() needs no GPU)

  // ---- workspace layout
  const size_t es = p.esz();
  size_t cur = 0;
  p.off_img4 = carve(cur, (size_t)p.N * p.Hp * p.Wp * 4 * es);
  p.off_wf = carve(cur, (size_t)wf * es);
  p.off_wd = carve(cur, (size_t)wd * es);
  size_t stat_rows_max = 0, maxact = 0, slab_max = stem_wgrad_slab_bytes(p.N, p.OH0, p.OW0), partial_max = 0;
  int maxC = 64;
  for (Unit& u : p.units) {
    size_t rows = u.rows();
    size_t sr = (size_t)ceil_div((int)rows, 128) * u.s.Cout;
    if (sr > stat_rows_max) stat_rows_max = sr;
    if (rows * u.s.Cout > maxact) maxact = rows * u.s.Cout;
    if ((size_t)u.s.N * u.s.H * u.s.W * u.s.Cin > maxact && !u.stem) maxact = (size_t)u.s.N * u.s.H * u.s.W * u.s.Cin;
    if (!u.stem) { size_t sb = conv_wgrad_slab_bytes(u.s); if (sb > slab_max) slab_max = sb; }
    size_t pr = (size_t)bn_bwd_partial_rows(rows, u.s.Cout) * 2 * u.s.Cout * sizeof(float);
    if (pr > partial_max) partial_max = pr;
    // partial rows written by a dgrad epilogue that produces this unit's output-shaped gradient
    pr = ((rows + 127) / 128 + 4) * 2 * u.s.Cout * sizeof(float);
    if (pr > partial_max) partial_max = pr;
    if (u.s.Cout > maxC) maxC = u.s.Cout;
  }
  p.stat_bytes = stat_rows_max * sizeof(float);
  p.off_stat = carve(cur, 2 * p.stat_bytes);
  p.off_stat_b = carve(cur, 2 * p.stat_bytes);   // downsample branch (runs on the side stream)
  p.maxact_bytes = maxact * es;
  for (Unit& u : p.units) {
    u.coef_off = carve(cur, 4 * (size_t)u.s.Cout * sizeof(float));
    u.x_off = carve(cur, u.rows() * u.s.Cout * es);
  }
  for (size_t i = 0; i < p.units.size(); ++i) {   // BatchNorm behind every conv: inference folding / eval coefficients
    StageDesc& d = p.table_host[i];
    const Unit& u = p.units[i];
    d.has_bn = 1;
    d.bn_g_off = u.g_off; d.bn_b_off = u.b_off; d.bn_rm_off = u.rm_off; d.bn_rv_off = u.rv_off;
    d.coef_off = (int64_t)u.coef_off;
  }
  p.off_pool = carve(cur, (size_t)p.N * p.PH * p.PW * 64 * es);
  p.off_idx = carve(cur, (size_t)p.N * p.PH * p.PW * 64);
  // post-activation outputs
  size_t prev = p.off_pool;
  for (Block& b : p.blocks) {
    b.in_off = prev;
    for (int ui : b.units) {
      Unit& u = p.units[ui];
      u.y_off = carve(cur, u.rows() * u.s.Cout * es);
    }
    prev = p.units[b.units.back()].y_off;
    Unit& bl = p.units[b.units.back()];
    b.mask_off = carve(cur, bl.rows() * bl.s.Cout * es / 16);
  }
  for (int i = 0; i < 7; ++i) p.off_scratch[i] = carve(cur, p.maxact_bytes);
  {   // algebraic BatchNorm backward of the bottlenecks' expanding 1x1 convolutions (bf16 plans; MMSKIN_ABN=0 switches it off,
      // MMSKIN_ABN_MAXC bounds the input width it takes: the data gradient of wider layers belongs to the pipelined kernel)
    static const int abn_on = [] { const char* v = getenv("MMSKIN_ABN"); return v ? atoi(v) : 1; }();
    static const int abn_maxc = [] { const char* v = getenv("MMSKIN_ABN_MAXC"); return v ? atoi(v) : 128; }();
    size_t wd_max = 0, s_max = 0;
    int cw_max = 0;
    std::vector<int> cand;
    for (Block& b : p.blocks) { cand.push_back(b.units.back()); if (b.ds >= 0) cand.push_back(b.ds); }
    for (int ui : cand) {
      Unit& u = p.units[ui];
      WgradRingPlan rp;
      int sh = 0;
      while ((u.s.Cin << sh) < u.s.Cout) ++sh;
      if (!abn_on || p.dtype != 1 || !bottleneck || u.s.kh != 1 || u.s.stride != 1 || u.s.Cin > abn_maxc || u.s.Cin % 64 || (u.s.Cin << sh) != u.s.Cout ||
          !wgrad_gram_plan((int)u.rows(), u.s.Cout, u.s.Cin, rp) || rp.gram_tiles != 1)
        continue;
      u.abn = true;
      u.abn_coef_off = carve(cur, 3 * (size_t)u.s.Cout * sizeof(float));
      const size_t sb = wgrad_gram_slab_bytes((int)u.rows(), u.s.Cout, u.s.Cin);
      if (sb > slab_max) slab_max = sb;
      const size_t wdb = (size_t)u.s.Cin * (u.s.Cout + u.s.Cin) * 2, sf = ((size_t)u.s.Cout + 64 * rp.wo) * u.s.Cin * sizeof(float);
      if (wdb > wd_max) wd_max = wdb;
      if (sf > s_max) s_max = sf;
      if (u.s.Cin > cw_max) cw_max = u.s.Cin;
    }
    if (wd_max) {
      p.off_abn_wd = carve(cur, wd_max);
      p.off_abn_bias = carve(cur, (size_t)cw_max * sizeof(float));
      p.off_abn_S = carve(cur, s_max);
      p.off_abn_cs = carve(cur, (size_t)cw_max * sizeof(float));
      p.off_abn_wd2 = carve(cur, wd_max);
      p.off_abn_bias2 = carve(cur, (size_t)cw_max * sizeof(float));
      static const int fwd2p_on = [] { const char* v = getenv("MMSKIN_FWD2P"); return v ? atoi(v) : 1; }();
      size_t slab2 = 0;
      for (Block& b : p.blocks) {
        Unit& u = p.units[b.units.back()];
        if (!fwd2p_on || !u.abn || b.ds >= 0) continue;
        u.fwd2p = true;
        size_t sb = wgrad_gram_slab_bytes((int)u.rows(), u.s.Cout, u.s.Cin);
        if (sb > slab2) slab2 = sb;
        // MMSKIN_FWDG (default 1): the first pass is not the convolution again but the Gram matrix of its INPUT (gram_stats, abn.hip);
        // the backward pass reuses that matrix and runs g^T y alone
        static const int fwdg_on = [] { const char* v = getenv("MMSKIN_FWDG"); return v ? atoi(v) : 1; }();
        WgradRingPlan rg;
        if (fwdg_on && wgrad_gram_plan((int)u.rows(), 0, u.s.Cin, rg, 2) && rg.gram_tiles == 1 &&
            wgrad_gram_slab_bytes((int)u.rows(), u.s.Cout, u.s.Cin, 1)) {
          u.gram_off = carve(cur, ((size_t)128 + 1) * u.s.Cin * sizeof(float));
          sb = wgrad_gram_slab_bytes((int)u.rows(), 0, u.s.Cin, 2);
          if (sb > slab2) slab2 = sb;
          sb = wgrad_gram_slab_bytes((int)u.rows(), u.s.Cout, u.s.Cin, 1);
          if (sb > slab2) slab2 = sb;
        }
      }
      if (slab2) {
        p.off_abn_sgx = carve(cur, 1024 * sizeof(float));
        p.off_abn_slab2 = carve(cur, slab2);
        p.off_abn_S2 = carve(cur, s_max);
        p.off_abn_cs2 = carve(cur, (size_t)cw_max * sizeof(float));
      }
    }
  }
  p.off_slab = carve(cur, slab_max);
  p.off_partial = carve(cur, partial_max);
  p.off_partial_b = carve(cur, partial_max);
  p.off_coefbwd = carve(cur, 3 * (size_t)maxC * sizeof(float));
  p.off_coefbwd_b = carve(cur, 3 * (size_t)maxC * sizeof(float));   // the downsample branch's own coefficients (its stream in backward)
  p.off_dwv = carve(cur, 64 * 256 * sizeof(float));
  p.off_red = carve(cur, bn_reduce_scratch_bytes(maxC));
  p.off_red_b = carve(cur, bn_reduce_scratch_bytes(maxC));
  p.ws_bytes = cur;
  return MMSKIN_OK;
}

// Inference: eval-mode BatchNorm is folded into the convolutions -- the staged weights carry gamma/sqrt(var+eps)
// (stage_weights above), the conv epilogue adds the shift, the residual and the ReLU, so no BatchNorm pass and
// no raw conv output exists (loadImageModelClassifier.py backbones under model.eval(): model_metrics.py:50-62).
template <typename T>
int forward_eval(Plan& p, const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
                 float* features, hipStream_t st, bool reuse_table) {
  const float eps = 1e-5f;
  T* wf = reinterpret_cast<T*>(ws + p.off_wf);
  int rc, maxC = 64;
  for (const Unit& u : p.units) maxC = u.s.Cout > maxC ? u.s.Cout : maxC;
  if (!reuse_table) PROF(K_BN_FWD, 0.0, 0.0, bn_eval_table(p.table_dev, (int)p.units.size(), maxC, params, buffers, ws, eps, st));
  auto shift_of = [&](const Unit& u) { return reinterpret_cast<const float*>(ws + u.coef_off) + u.s.Cout; };
  // stem: the 7x7 conv keeps its own BN + ReLU + max-pool kernel (one pass over the largest activation)
  Unit& u0 = p.units[0];
  T* img4 = reinterpret_cast<T*>(ws + p.off_img4);
  if (norm6) PROF(K_STEM_MISC, 0.0, 0.0, stem_pack_u8<T>((const uint8_t*)image, p.N, p.H, p.W, p.Hp, p.Wp, norm6, img4, st));
  else PROF(K_STEM_MISC, 0.0, 0.0, stem_pack<T>((const float*)image, p.N, p.H, p.W, p.Hp, p.Wp, img4, st));
  T* x0 = reinterpret_cast<T*>(ws + u0.x_off);
  PROF(K_CONV_FWD, conv_flops(u0.s), conv_bytes(u0.s, sizeof(T)),
       launch_stem_conv_fwd<T>(p.N, p.OH0, p.OW0, p.Hp, p.Wp, img4, wf + u0.wf_off, x0, nullptr, nullptr, st));
  const float* c0 = reinterpret_cast<const float*>(ws + u0.coef_off);
  T* pool = reinterpret_cast<T*>(ws + p.off_pool);
  PROF(K_STEM_MISC, 0.0, 0.0, stem_bn_relu_pool<T>(x0, c0, c0 + 64, p.N, p.OH0, p.OW0, 64, pool, ws + p.off_idx, st));
  for (Block& b : p.blocks) {
    const T* in = reinterpret_cast<const T*>(ws + b.in_off);
    const T* cur = in;
    const int nu = (int)b.units.size();
    const T* skip = in;
    if (b.ds >= 0) {
      Unit& d = p.units[b.ds];
      FwdFuse f; f.bias = shift_of(d);
      T* xd = reinterpret_cast<T*>(ws + d.x_off);
      PROF(K_CONV_FWD, conv_flops(d.s), conv_bytes(d.s, sizeof(T)), launch_conv_fwd<T>(d.s, in, wf + d.wf_off, xd, nullptr, nullptr, st, &f));
      skip = xd;
    }
    for (int i = 0; i < nu; ++i) {
      Unit& u = p.units[b.units[i]];
      T* y = reinterpret_cast<T*>(ws + u.y_off);
      FwdFuse f; f.bias = shift_of(u); f.relu = true;
      if (i + 1 == nu) f.addend = skip;
      PROF(K_CONV_FWD, conv_flops(u.s), conv_bytes(u.s, sizeof(T), 0), launch_conv_fwd<T>(u.s, cur, wf + u.wf_off, y, nullptr, nullptr, st, &f));
      cur = y;
    }
  }
  Unit& last = p.units[p.blocks.back().units.back()];
  return avgpool_fwd<T>(reinterpret_cast<const T*>(ws + last.y_off), p.N, last.s.OH() * last.s.OW(), last.s.Cout, features, st);
}

template <typename T>
int forward_impl(Plan& p, const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
                 float* features, bool training, hipStream_t st) {
  const float eps = 1e-5f, mom = 0.1f;
  T* wf = reinterpret_cast<T*>(ws + p.off_wf);
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  float* stat_sum = reinterpret_cast<float*>(ws + p.off_stat);
  float* stat_sq = reinterpret_cast<float*>(ws + p.off_stat + p.stat_bytes);
  int rc;
  if ((rc = p.ensure_table())) return rc;
  const bool folded_eval = !training && !p.keep_raw_eval;
  const bool reuse = folded_eval && p.reuse_staged && p.staged_eval_ws == ws;
  p.staged_eval_ws = folded_eval ? ws : nullptr;
  // stage weights (stem region needs zeros in its padding taps)
  if (!reuse) HIP_CHECK_RET(hipMemsetAsync(wf + p.units[0].wf_off, 0, 64 * 256 * sizeof(T), st));
  static const bool side_off_f = [] { const char* v = getenv("MMSKIN_NO_SIDE_STREAM"); return v && atoi(v) != 0; }();
  const bool use_side = !side_off_f && !p.prof.on;
  if (use_side && (rc = p.side.init())) return rc;
  const bool stage_aside = use_side && training;   // the stem only needs its own weights: the other layers are staged beside it
  const float* fold = (training || p.keep_raw_eval) ? nullptr : buffers;
  if (stage_aside) {
    if ((rc = stage_weights<T>(p.table_dev, 1, p.max_stage_elems, params, wf, wd, training, st, fold, eps))) return rc;
    HIP_CHECK_RET(hipEventRecord(p.side.f_ready, st));
    HIP_CHECK_RET(hipStreamWaitEvent(p.side.s, p.side.f_ready, 0));
    if ((rc = stage_weights<T>(p.table_dev + 1, (int)p.units.size() - 1, p.max_stage_elems, params, wf, wd, training, p.side.s,
                               fold, eps))) return rc;
    HIP_CHECK_RET(hipEventRecord(p.side.f_staged, p.side.s));
  } else if (!reuse) {
    PROF(K_STAGE, 0.0, 0.0, stage_weights<T>(p.table_dev, (int)p.units.size(), p.max_stage_elems, params, wf, wd, training, st, fold, eps));
  }
  if (folded_eval) return forward_eval<T>(p, image, norm6, params, buffers, ws, features, st, reuse);

  auto bn_coeffs_on = [&](Unit& u, int stat_rows, float* ssum, float* ssq, double* red, hipStream_t s2) -> int {
    float* coef = reinterpret_cast<float*>(ws + u.coef_off);
    const int C = u.s.Cout;
    p.prof.begin(K_BN_FWD, s2);
    struct End { Profiler& pr; hipStream_t s; ~End() { pr.end(s); } } end_guard{p.prof, s2};
    if (training)
      return bn_finalize(ssum, ssq, stat_rows, C, (double)u.rows(), params + u.g_off, params + u.b_off, eps,
                         mom, buffers + u.rm_off, buffers + u.rv_off, coef, coef + C, coef + 2 * C, coef + 3 * C, red, s2);
    return bn_eval_coeffs(C, params + u.g_off, params + u.b_off, buffers + u.rm_off, buffers + u.rv_off, eps, coef,
                          coef + C, s2);
  };
  auto bn_coeffs = [&](Unit& u, int stat_rows) -> int {
    return bn_coeffs_on(u, stat_rows, stat_sum, stat_sq, reinterpret_cast<double*>(ws + p.off_red), st);
  };
  float* stat_b_sum = reinterpret_cast<float*>(ws + p.off_stat_b);
  float* stat_b_sq = reinterpret_cast<float*>(ws + p.off_stat_b + p.stat_bytes);

  // ---- stem
  Unit& u0 = p.units[0];
  T* img4 = reinterpret_cast<T*>(ws + p.off_img4);
  if (norm6) PROF(K_STEM_MISC, 0.0, 0.0, stem_pack_u8<T>((const uint8_t*)image, p.N, p.H, p.W, p.Hp, p.Wp, norm6, img4, st));
  else PROF(K_STEM_MISC, 0.0, 0.0, stem_pack<T>((const float*)image, p.N, p.H, p.W, p.Hp, p.Wp, img4, st));
  T* x0 = reinterpret_cast<T*>(ws + u0.x_off);
  int stem_rows = 0;
  PROF(K_CONV_FWD, conv_flops(u0.s), conv_bytes(u0.s, sizeof(T)),
       launch_stem_conv_fwd<T>(p.N, p.OH0, p.OW0, p.Hp, p.Wp, img4, wf + u0.wf_off, x0,
                               training ? stat_sum : nullptr, training ? stat_sq : nullptr, st, &stem_rows));
  if ((rc = bn_coeffs(u0, stem_rows))) return rc;
  float* c0 = reinterpret_cast<float*>(ws + u0.coef_off);
  T* pool = reinterpret_cast<T*>(ws + p.off_pool);
  PROF(K_STEM_MISC, 0.0, 0.0, stem_bn_relu_pool<T>(x0, c0, c0 + 64, p.N, p.OH0, p.OW0, 64, pool, ws + p.off_idx, st));

  // ---- residual stages
  if (stage_aside) HIP_CHECK_RET(hipStreamWaitEvent(st, p.side.f_staged, 0));
  for (Block& b : p.blocks) {
    const T* in = reinterpret_cast<const T*>(ws + b.in_off);
    const T* cur = in;
    const int nu = (int)b.units.size();
    if (b.ds >= 0) {
      // the downsample conv + its BN statistics only meet the main branch at the block's final BN-apply:
      // they run on the side stream (own stat slab / reduction scratch) beside conv1..conv3
      Unit& d = p.units[b.ds];
      if (use_side) {
        HIP_CHECK_RET(hipEventRecord(p.side.f_ready, st));
        HIP_CHECK_RET(hipStreamWaitEvent(p.side.s, p.side.f_ready, 0));
        int nrows_d = 0;
        if ((rc = launch_conv_fwd<T>(d.s, in, wf + d.wf_off, reinterpret_cast<T*>(ws + d.x_off),
                                     training ? stat_b_sum : nullptr, training ? stat_b_sq : nullptr, p.side.s, nullptr, &nrows_d))) return rc;
        if ((rc = bn_coeffs_on(d, nrows_d, stat_b_sum, stat_b_sq,
                               reinterpret_cast<double*>(ws + p.off_red_b), p.side.s))) return rc;
        HIP_CHECK_RET(hipEventRecord(p.side.f_done, p.side.s));
      } else {
        int nrows_d = 0;
        PROF(K_CONV_FWD, conv_flops(d.s), conv_bytes(d.s, sizeof(T)),
             launch_conv_fwd<T>(d.s, in, wf + d.wf_off, reinterpret_cast<T*>(ws + d.x_off),
                                training ? stat_sum : nullptr, training ? stat_sq : nullptr, st, nullptr, &nrows_d));
        if ((rc = bn_coeffs(d, nrows_d))) return rc;
      }
    }
    for (int i = 0; i < nu; ++i) {
      Unit& u = p.units[b.units[i]];
      T* x = reinterpret_cast<T*>(ws + u.x_off);
      T* y = reinterpret_cast<T*>(ws + u.y_off);
      int nrows_u = 0;   // statistics row blocks this launch wrote (the launcher picks the tile height)
      const bool two_pass = training && u.fwd2p && i + 1 == nu && sizeof(T) == 2;
      const bool gram_pass = two_pass && u.gram_off != 0;
      if (gram_pass) {
        if constexpr (sizeof(T) == 2) {
          float* gm = reinterpret_cast<float*>(ws + u.gram_off);
          float* gcs = gm + (size_t)128 * u.s.Cin;
          PROF(K_BN_FWD, 0.0, (double)u.rows() * u.s.Cin * sizeof(T),
               launch_wgrad_gram(u.s.N, u.s.OH(), u.s.OW(), u.s.Cin, u.s.Cout, nullptr, cur, reinterpret_cast<float*>(ws + p.off_abn_slab2), gm, gcs, st, 2));
          float* cf = reinterpret_cast<float*>(ws + u.coef_off);
          const int Cq = u.s.Cout;
          PROF(K_BN_FWD, 0.0, 0.0, gram_stats_finalize(gm, gcs, params + u.w_off, Cq, u.s.Cin, (double)u.rows(), params + u.g_off, params + u.b_off, eps, mom,
                                                       buffers + u.rm_off, buffers + u.rv_off, cf, cf + Cq, cf + 2 * Cq, cf + 3 * Cq, st));
          nrows_u = -1;   // coefficients done
        }
      } else {
        PROF(K_CONV_FWD, conv_flops(u.s), conv_bytes(u.s, sizeof(T)),
             launch_conv_fwd<T>(u.s, cur, wf + u.wf_off, two_pass ? (T*)nullptr : x, training ? stat_sum : nullptr,
                                training ? stat_sq : nullptr, st, nullptr, &nrows_u));
      }
      if (nrows_u >= 0 && (rc = bn_coeffs(u, nrows_u))) return rc;
      float* coef = reinterpret_cast<float*>(ws + u.coef_off);
      const int C = u.s.Cout;
      if (two_pass) {
        // second pass: the conv again with scale * acc + shift + identity + ReLU (+ the mask byte) in its epilogue -- the raw output
        // (the block's widest tensor) is neither written nor read: 2T + 2t bytes instead of 4T + t, and the normalisation multiplies the
        // fp32 accumulator, not a bf16-rounded copy of it
        FwdFuse f2; f2.mul = coef; f2.bias = coef + C; f2.addend = in; f2.relu = true; f2.mask_out = ws + b.mask_off;
        // (class accounting: the recomputed MACs are overhead, not algorithmic work -- the layer's FLOPs were counted with the first pass)
        PROF(K_CONV_FWD, gram_pass ? conv_flops(u.s) : 0.0, conv_bytes(u.s, sizeof(T), 0) + (double)u.rows() * C * sizeof(T), launch_conv_fwd<T>(u.s, cur, wf + u.wf_off, y, nullptr, nullptr, st, &f2));
      } else if (i + 1 < nu) {
        PROF(K_BN_FWD, 0.0, 2.0 * u.rows() * C * sizeof(T), bn_apply<T>(x, nullptr, coef, coef + C, nullptr, nullptr, y, u.rows(), C, true, st));
      } else if (b.ds >= 0) {
        Unit& d = p.units[b.ds];
        float* dc = reinterpret_cast<float*>(ws + d.coef_off);
        if (use_side) HIP_CHECK_RET(hipStreamWaitEvent(st, p.side.f_done, 0));
        PROF(K_BN_FWD, 0.0, 3.0 * u.rows() * C * sizeof(T),
             bn_apply<T>(x, reinterpret_cast<const T*>(ws + d.x_off), coef, coef + C, dc, dc + C, y, u.rows(), C, true, st,
                         training ? ws + b.mask_off : nullptr));
      } else {
        PROF(K_BN_FWD, 0.0, 3.0 * u.rows() * C * sizeof(T),
             bn_apply<T>(x, in, coef, coef + C, nullptr, nullptr, y, u.rows(), C, true, st, training ? ws + b.mask_off : nullptr));
      }
      cur = y;
    }
  }
  Unit& last = p.units[p.blocks.back().units.back()];
  return avgpool_fwd<T>(reinterpret_cast<const T*>(ws + last.y_off), p.N, last.s.OH() * last.s.OW(), last.s.Cout,
                        features, st);
}

template <typename T>
int backward_impl(Plan& p, const float* dfeat, const float* params, unsigned char* ws, float* grads,
                  hipStream_t st) {
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  float* slab = reinterpret_cast<float*>(ws + p.off_slab);
  float* partial = reinterpret_cast<float*>(ws + p.off_partial);
  float* cA = reinterpret_cast<float*>(ws + p.off_coefbwd);
  T* S[7];
  for (int i = 0; i < 7; ++i) S[i] = reinterpret_cast<T*>(ws + p.off_scratch[i]);
  int rc;

  // BN backward of unit u given dy: fills dx (and optionally dz)
  auto bn_backward = [&](Unit& u, const T* dy, const T* ymask, int mode, T* dx, T* dz) -> int {
    const int C = u.s.Cout;
    float* coef = reinterpret_cast<float*>(ws + u.coef_off);
    const T* x = reinterpret_cast<const T*>(ws + u.x_off);
    float* cB = cA + C; float* cC = cA + 2 * C;
    int r, nr = 0;
    p.prof.begin(K_BN_BWD, st);
    struct End { Profiler& pr; hipStream_t s; ~End() { pr.end(s); } } end_guard{p.prof, st};
    if (p.prof.on) p.prof.bytes[K_BN_BWD] += (mode == MASK_FROM_Y ? 7.0 : 5.0) * u.rows() * C * sizeof(T);
    if ((r = bn_bwd_reduce<T>(dy, x, ymask, coef, coef + C, mode, u.rows(), C, partial, &nr, st))) return r;
    if ((r = bn_bwd_finalize(partial, nr, C, (double)u.rows(), params + u.g_off,
                             coef + 2 * C, coef + 3 * C, grads + u.g_off, grads + u.b_off, cA, cB, cC,
                             reinterpret_cast<double*>(ws + p.off_red), st))) return r;
    return bn_bwd_apply<T>(dy, x, ymask, coef, coef + C, mode, cA, cB, cC, dx, dz, u.rows(), C, st);
  };

  float* partial_b = reinterpret_cast<float*>(ws + p.off_partial_b);
  // BN backward of unit u when `dz` is ALREADY masked and its partial sums (sum dz, sum dz*x) were
  // produced by the epilogue of the dgrad launch that wrote dz: finalize + one apply pass.
  // on_branch: the downsample branch on its own stream (own coefficient and reduction scratch)
  // dx == nullptr: finalize only (gamma / beta gradients and the coefficients cA, cB, cC) -- the algebraic path folds the apply
  // into its two GEMMs
  auto bn_backward_fused = [&](Unit& u, const T* dz, const float* part, int nrows, T* dx, bool on_branch = false, const float* sum_dz_x = nullptr) -> int {
    const int C = u.s.Cout;
    float* coef = reinterpret_cast<float*>(ws + u.coef_off);
    const T* x = reinterpret_cast<const T*>(ws + u.x_off);
    float* kA = on_branch ? reinterpret_cast<float*>(ws + p.off_coefbwd_b) : cA;
    float* cB = kA + C; float* cC = kA + 2 * C;
    hipStream_t bs = on_branch ? p.side.s2 : st;
    double* red = reinterpret_cast<double*>(ws + (on_branch ? p.off_red_b : p.off_red));
    int r;
    p.prof.begin(K_BN_BWD, st);
    struct End { Profiler& pr; hipStream_t s; ~End() { pr.end(s); } } end_guard{p.prof, st};
    if (p.prof.on && dx) p.prof.bytes[K_BN_BWD] += 3.0 * u.rows() * C * sizeof(T);
    if ((r = bn_bwd_finalize(part, nrows, C, (double)u.rows(), params + u.g_off, coef + 2 * C, coef + 3 * C,
                             grads + u.g_off, grads + u.b_off, kA, cB, cC, red, bs, -1, false, sum_dz_x))) return r;
    if (!dx) return MMSKIN_OK;
    return bn_bwd_apply<T>(dz, x, nullptr, coef, coef + C, MASK_NONE, kA, cB, cC, dx, nullptr, u.rows(), C, bs);
  };

  Unit& last = p.units[p.blocks.back().units.back()];
  T* g = S[0];
  T* gin = S[1];
  if ((rc = avgpool_bwd<T>(dfeat, p.N, last.s.OH() * last.s.OW(), last.s.Cout, g, st))) return rc;

  // ---- side stream for the weight-gradient GEMMs (buffers: 0/1 = alternating dX, 2 = downsample dX)
  static const bool side_off = [] { const char* v = getenv("MMSKIN_NO_SIDE_STREAM"); return v && atoi(v) != 0; }();
  const bool use_side = !side_off && !p.prof.on;
  if (use_side && (rc = p.side.init())) return rc;
  for (int i = 0; i < 3; ++i) p.side.done_valid[i] = false;
  // measured: 19.99 / 20.06 ms per step off, 20.05 / 20.12 ms on (same box, profiles/r03_experiments.txt (9)) -- the chip has no idle
  // resource for the branch to use, the launches only move.  Off by default; kept as an A/B knob.
  static const bool ds_stream = [] { const char* v = getenv("MMSKIN_BWD_DS_STREAM"); return v && atoi(v) != 0; }();
  T* DX[3] = {S[2], S[6], S[5]};
  // main stream may overwrite buffer i only after the wgrad that reads it has finished
  auto acquire = [&](int i) -> int {
    if (use_side && p.side.done_valid[i]) HIP_CHECK_RET(hipStreamWaitEvent(st, p.side.done[i], 0));
    return MMSKIN_OK;
  };
  // from: the stream that produced DX[i] (the main stream, or the downsample branch's)
  auto wgrad_async = [&](Unit& u, int i, const T* uin, hipStream_t from = nullptr) -> int {
    hipStream_t ws_st = st;
    if (use_side) {
      HIP_CHECK_RET(hipEventRecord(p.side.ready[i], from ? from : st));
      HIP_CHECK_RET(hipStreamWaitEvent(p.side.s, p.side.ready[i], 0));
      ws_st = p.side.s;
    }
    p.prof.begin(K_WGRAD, st);
    int r = launch_conv_wgrad<T>(u.s, DX[i], uin, slab, grads + u.w_off, ws_st);
    p.prof.end(st);
    if (p.prof.on) { p.prof.flops[K_WGRAD] += conv_flops(u.s); p.prof.bytes[K_WGRAD] += conv_bytes(u.s, sizeof(T)); }
    if (r) return r;
    if (use_side) {
      HIP_CHECK_RET(hipEventRecord(p.side.done[i], p.side.s));
      p.side.done_valid[i] = true;
    }
    return MMSKIN_OK;
  };
  int dxi = 0;   // index of the buffer holding the current dX
  for (int i = 0; i < 2; ++i) p.side.g_done_valid[i] = false;
  // algebraic path: the weight-gradient stream reads the block-output gradient buffer (S[0] / S[1]) itself; the main stream may write
  // that buffer again (two blocks later) only after that launch has finished
  auto g_index = [&](const T* buf) { return buf == S[0] ? 0 : 1; };
  auto g_acquire = [&](const T* buf, hipStream_t writer = nullptr) -> int {
    const int i = g_index(buf);
    if (use_side && p.side.g_done_valid[i]) HIP_CHECK_RET(hipStreamWaitEvent(writer ? writer : st, p.side.g_done[i], 0));
    return MMSKIN_OK;
  };
  // dW = cA (.) (g^T y) + cB (.) (W (y^T y)) + cC (x) colsum(y) on the weight-gradient stream
  auto wgrad_abn_async = [&](Unit& u, const T* gbuf, const T* uin) -> int {
    if constexpr (sizeof(T) == 2) {
      hipStream_t ws_st = st;
      if (use_side) {
        HIP_CHECK_RET(hipEventRecord(p.side.g_ready, st));
        HIP_CHECK_RET(hipStreamWaitEvent(p.side.s, p.side.g_ready, 0));
        ws_st = p.side.s;
      }
      float* S_out = reinterpret_cast<float*>(ws + p.off_abn_S);
      float* cs_out = reinterpret_cast<float*>(ws + p.off_abn_cs);
      p.prof.begin(K_WGRAD, st);
      int r = launch_wgrad_gram(u.s.N, u.s.OH(), u.s.OW(), u.s.Cin, u.s.Cout, gbuf, uin, slab, S_out, cs_out, ws_st);
      if (!r) r = abn_wgrad_finalize(S_out, cs_out, params + u.w_off, reinterpret_cast<const float*>(ws + u.abn_coef_off), u.s.Cout, u.s.Cin,
                                     grads + u.w_off, ws_st);
      p.prof.end(st);
      if (p.prof.on) { p.prof.flops[K_WGRAD] += conv_flops(u.s); p.prof.bytes[K_WGRAD] += conv_bytes(u.s, sizeof(T)); }
      if (r) return r;
      if (use_side) {
        const int gi = g_index(gbuf);
        HIP_CHECK_RET(hipEventRecord(p.side.g_done[gi], p.side.s));
        p.side.g_done_valid[gi] = true;
      }
      return MMSKIN_OK;
    } else {
      return MMSKIN_ERR_UNSUPPORTED;
    }
  };

  bool fused_ready = false;   // g already holds the masked dz of this block's final unit, partials in the slabs
  int fused_rows = 0;
  for (int bi = (int)p.blocks.size() - 1; bi >= 0; --bi) {
    Block& b = p.blocks[bi];
    const int nu = (int)b.units.size();
    const T* in = reinterpret_cast<const T*>(ws + b.in_off);
    Unit& ul = p.units[b.units[nu - 1]];
    const T* out = reinterpret_cast<const T*>(ws + ul.y_off);
    T* dY = S[3];
    T* dZ = S[4];
    T* dXd = DX[2];
    const bool has_ds = b.ds >= 0;
    // The downsample branch (BatchNorm-backward apply, dgrad) only has to be done when conv1's dgrad takes its result as the
    // addend: with MMSKIN_BWD_DS_STREAM=1 it runs on its own stream beside conv3 .. conv2 (as in the forward; ~0.9 ms of launches
    // per ResNet-50 step leave the main chain).  The main stream hands it g + partial_b (d_ready) and takes gin back (d_done).
    const bool ds_branch = has_ds && use_side && ds_stream && fused_ready;
    const T* dz_final;   // masked gradient of the block output (residual branch addend)
    dxi ^= 1;
    if ((rc = acquire(dxi))) return rc;
    if (has_ds && !ds_branch && (rc = acquire(2))) return rc;
    T* dX = DX[dxi];
    const bool abn = fused_ready && ul.abn && sizeof(T) == 2;
    if (fused_ready) {
      if (ds_branch) {
        HIP_CHECK_RET(hipEventRecord(p.side.d_ready, st));
        HIP_CHECK_RET(hipStreamWaitEvent(p.side.s2, p.side.d_ready, 0));
        if (p.side.done_valid[2]) HIP_CHECK_RET(hipStreamWaitEvent(p.side.s2, p.side.done[2], 0));   // the last wgrad that read dXd
        if ((rc = bn_backward_fused(p.units[b.ds], g, partial_b, fused_rows, dXd, true))) return rc;
      }
      const T* uin3 = nu > 1 ? reinterpret_cast<const T*>(ws + p.units[b.units[nu - 2]].y_off) : in;
      const float* sgx = nullptr;
      if (abn && ul.fwd2p) {
        // two-pass unit: x was never stored.  g^T y first (main stream: the BatchNorm-backward sum of g x is W . (g^T y) row by row)
        if constexpr (sizeof(T) == 2) {
          float* S2 = reinterpret_cast<float*>(ws + p.off_abn_S2);
          p.prof.begin(K_WGRAD, st);
          rc = launch_wgrad_gram(ul.s.N, ul.s.OH(), ul.s.OW(), ul.s.Cin, ul.s.Cout, g, uin3, reinterpret_cast<float*>(ws + p.off_abn_slab2), S2,
                                 reinterpret_cast<float*>(ws + p.off_abn_cs2), st, ul.gram_off ? 1 : 0);
          p.prof.end(st);
          if (p.prof.on) { p.prof.flops[K_WGRAD] += conv_flops(ul.s); p.prof.bytes[K_WGRAD] += conv_bytes(ul.s, sizeof(T)); }
          if (rc) return rc;
          if ((rc = abn_sgx(S2, params + ul.w_off, ul.s.Cout, ul.s.Cin, reinterpret_cast<float*>(ws + p.off_abn_sgx), st))) return rc;
          sgx = reinterpret_cast<const float*>(ws + p.off_abn_sgx);
        }
      }
      if ((rc = bn_backward_fused(ul, g, partial, fused_rows, abn ? nullptr : dX, false, sgx))) return rc;
      if (abn) {   // fold the coefficients into this block's conv3 data-gradient weights; keep a copy for the weight-gradient fix-up
        if constexpr (sizeof(T) == 2) {
          if ((rc = abn_prep(params + ul.w_off, cA, cA + ul.s.Cout, cA + 2 * ul.s.Cout, ul.s.Cout, ul.s.Cin, reinterpret_cast<bf16_t*>(ws + p.off_abn_wd),
                             reinterpret_cast<float*>(ws + p.off_abn_bias), reinterpret_cast<float*>(ws + ul.abn_coef_off), st))) return rc;
          const float* gkept = ul.gram_off ? reinterpret_cast<const float*>(ws + ul.gram_off) : nullptr;   // y^T y | colsum(y) from the forward pass
          if (ul.fwd2p && (rc = abn_wgrad_finalize(reinterpret_cast<const float*>(ws + p.off_abn_S2),
                                                   gkept ? gkept + (size_t)128 * ul.s.Cin : reinterpret_cast<const float*>(ws + p.off_abn_cs2), params + ul.w_off,
                                                   reinterpret_cast<const float*>(ws + ul.abn_coef_off), ul.s.Cout, ul.s.Cin, grads + ul.w_off, st, gkept))) return rc;
        }
      }
      if (has_ds && !ds_branch) {
        Unit& d = p.units[b.ds];
        const bool abn_d = d.abn && sizeof(T) == 2;
        if ((rc = bn_backward_fused(d, g, partial_b, fused_rows, abn_d ? nullptr : dXd))) return rc;
        if (abn_d) {
          if constexpr (sizeof(T) == 2) {
            if ((rc = abn_prep(params + d.w_off, cA, cA + d.s.Cout, cA + 2 * d.s.Cout, d.s.Cout, d.s.Cin, reinterpret_cast<bf16_t*>(ws + p.off_abn_wd2),
                               reinterpret_cast<float*>(ws + p.off_abn_bias2), reinterpret_cast<float*>(ws + d.abn_coef_off), st))) return rc;
          }
        }
      }
      dz_final = g;
    } else {
      if ((rc = bn_backward(ul, g, out, MASK_FROM_Y, dX, has_ds ? nullptr : dZ))) return rc;
      if (has_ds)
        if ((rc = bn_backward(p.units[b.ds], g, out, MASK_FROM_Y, dXd, nullptr))) return rc;
      dz_final = dZ;
    }
    for (int i = nu - 1; i >= 0; --i) {
      Unit& u = p.units[b.units[i]];
      const T* uin = i == 0 ? in : reinterpret_cast<const T*>(ws + p.units[b.units[i - 1]].y_off);
      const bool abn_u = abn && i == nu - 1;
      if (abn_u) { if (!u.fwd2p && (rc = wgrad_abn_async(u, g, uin))) return rc; }   // (two-pass units: done above, on this stream)
      else if ((rc = wgrad_async(u, dxi, uin))) return rc;
      if (i > 0) {
        // dgrad writes the gradient of unit i-1's ReLU output; its epilogue applies that ReLU's mask and
        // accumulates unit i-1's BN-backward sums, so the stand-alone reduce pass is gone.
        Unit& up = p.units[b.units[i - 1]];
        float* cup = reinterpret_cast<float*>(ws + up.coef_off);
        DgradFuse f;
        f.x = ws + up.x_off; f.scale = cup; f.shift = cup + up.s.Cout; f.partial = partial;
        if (abn_u) {   // dY = [g | y] [cA (.) W ; Q] + r: no dz in memory
          f.in2 = uin; f.k2 = u.s.Cin; f.bias = reinterpret_cast<const float*>(ws + p.off_abn_bias);
          PROF(K_CONV_DGRAD, conv_flops(u.s), conv_bytes(u.s, sizeof(T), 1), launch_conv_dgrad<T>(u.s, g, reinterpret_cast<const T*>(ws + p.off_abn_wd), dY, (const T*)nullptr, st, &f));
        } else
        PROF(K_CONV_DGRAD, conv_flops(u.s), conv_bytes(u.s, sizeof(T), 1), launch_conv_dgrad<T>(u.s, dX, wd + u.wd_off, dY, (const T*)nullptr, st, &f));
        dxi ^= 1;                              // the next dX goes to the other buffer: wgrad(u) may still read this one
        if ((rc = acquire(dxi))) return rc;
        dX = DX[dxi];
        if ((rc = bn_backward_fused(up, dY, partial, f.rows_written, dX))) return rc;
      } else {
        const T* addend = dz_final;
        bool ds_addend_compact = false;
        static const bool ds_compact = [] { const char* v = getenv("MMSKIN_DS_COMPACT"); return !v || atoi(v) != 0; }();
        if (has_ds) {
          Unit& d = p.units[b.ds];
          if ((rc = g_acquire(gin, ds_branch ? p.side.s2 : nullptr))) return rc;
          if (ds_branch) {
            if ((rc = wgrad_async(d, 2, in, p.side.s2))) return rc;
            if ((rc = launch_conv_dgrad<T>(d.s, dXd, wd + d.wd_off, gin, (const T*)nullptr, p.side.s2))) return rc;
            HIP_CHECK_RET(hipEventRecord(p.side.d_done, p.side.s2));
            HIP_CHECK_RET(hipStreamWaitEvent(st, p.side.d_done, 0));
          } else if (fused_ready && d.abn && sizeof(T) == 2) {   // stride-1 downsample convolution on the algebraic path, as conv3 above
            if ((rc = wgrad_abn_async(d, g, in))) return rc;
            DgradFuse fd;
            fd.in2 = in; fd.k2 = d.s.Cin; fd.bias = reinterpret_cast<const float*>(ws + p.off_abn_bias2);
            PROF(K_CONV_DGRAD, conv_flops(d.s), conv_bytes(d.s, sizeof(T)), launch_conv_dgrad<T>(d.s, g, reinterpret_cast<const T*>(ws + p.off_abn_wd2), gin, (const T*)nullptr, st, &fd));
          } else if (ds_compact && bi > 0 && d.s.kh == 1 && d.s.stride == 2 && d.s.pad == 0 && u.s.kh == 1 && u.s.stride == 1) {
            // Stride-2 1x1 downsample: its data gradient lives on the even pixels only.  It is computed as the dense GEMM over the pixels the
            // convolution read (a 1x1 / stride 1 launch on the OH x OW grid) into a COMPACT buffer, and conv1's dgrad adds it at the even
            // pixels in its epilogue: the zero fill of the other three quarters (154 + 77 + 38 MB written, then read back as the addend)
            // and its launches are gone.
            if ((rc = wgrad_async(d, 2, in))) return rc;
            ConvShape dc = d.s;
            dc.H = d.s.OH(); dc.W = d.s.OW(); dc.stride = 1;
            PROF(K_CONV_DGRAD, conv_flops(d.s), conv_bytes(d.s, sizeof(T)), launch_conv_dgrad<T>(dc, dXd, wd + d.wd_off, S[3], (const T*)nullptr, st));
            ds_addend_compact = true;
          } else {
            if ((rc = wgrad_async(d, 2, in))) return rc;
            PROF(K_CONV_DGRAD, conv_flops(d.s), conv_bytes(d.s, sizeof(T)), launch_conv_dgrad<T>(d.s, dXd, wd + d.wd_off, gin, (const T*)nullptr, st));
          }
          addend = ds_addend_compact ? S[3] : gin;   // main-branch dgrad accumulates on top (in place unless the branch's gradient is compact)
        }
        if ((rc = g_acquire(gin))) return rc;
        DgradFuse f;
        DgradFuse* fp = nullptr;
        if (bi > 0) {   // gin is the gradient of the previous block's output: fuse that block's final BN reduce
          Block& pb = p.blocks[bi - 1];
          Unit& pu = p.units[pb.units.back()];
          f.mask_bits = ws + pb.mask_off; f.x = pu.fwd2p ? nullptr : ws + pu.x_off; f.partial = partial;   // two-pass unit: no raw output to take sums against
          if (pb.ds >= 0) { f.x2 = ws + p.units[pb.ds].x_off; f.partial_b = partial_b; }
          f.addend_s2 = ds_addend_compact;
          fp = &f;
        }
        PROF(K_CONV_DGRAD, conv_flops(u.s), conv_bytes(u.s, sizeof(T), 1 + (fp ? (f.x2 ? 3 : 2) : 0)), launch_conv_dgrad<T>(u.s, dX, wd + u.wd_off, gin, addend, st, fp));
        fused_ready = fp != nullptr;
        fused_rows = f.rows_written;
      }
    }
    T* t = g; g = gin; gin = t;
    if (b.seg_done >= 0 && (rc = p.segment_done(b.seg_done, st, use_side))) return rc;
  }

  // ---- stem: g = grad wrt pooled activation.  Its BatchNorm backward touches none of the buffers the side stream's
  // weight-gradient GEMMs are still reading (dX ring, forward activations) or writing (slab), so it runs on the main
  // stream BESIDE the tail of layer1's wgrads; the stem's own wgrad needs the slab and queues behind them on the side stream.
  Unit& u0 = p.units[0];
  T* dx0 = S[3];
  {   // max-pool + ReLU + BatchNorm backward without materialising the full-resolution pooled gradient
    float* c0 = reinterpret_cast<float*>(ws + u0.coef_off);
    const T* x0 = reinterpret_cast<const T*>(ws + u0.x_off);
    float* cB = cA + 64; float* cC = cA + 128;
    int nr = 0;
    p.prof.begin(K_BN_BWD, st);
    // MMSKIN_STEM_SUMS_POOLED=0: the sums from the conv output and the routed gradient (565 MB read instead of 206 MB)
    static const bool pooled_sums = [] { const char* v = getenv("MMSKIN_STEM_SUMS_POOLED"); return !v || atoi(v) != 0; }();
    if (pooled_sums) rc = stem_pool_bwd_sums<T>(g, reinterpret_cast<const T*>(ws + p.off_pool), ws + p.off_idx, x0, c0, c0 + 64, p.N, p.OH0, p.OW0, 64, partial, &nr, st);
    else rc = stem_pool_bn_bwd_reduce<T>(g, ws + p.off_idx, x0, c0, c0 + 64, p.N, p.OH0, p.OW0, 64, partial, &nr, st);
    if (!rc) rc = bn_bwd_finalize(partial, nr, 64, (double)u0.rows(), params + u0.g_off, c0 + 128, c0 + 192, grads + u0.g_off,
                                  grads + u0.b_off, cA, cB, cC, reinterpret_cast<double*>(ws + p.off_red), st);
    if (!rc) rc = stem_pool_bn_bwd_apply<T>(g, ws + p.off_idx, x0, c0, c0 + 64, cA, cB, cC, p.N, p.OH0, p.OW0, 64, dx0, st);
    p.prof.end(st);
    if (p.prof.on) p.prof.bytes[K_BN_BWD] += 3.5 * u0.rows() * 64 * sizeof(T);
    if (rc) return rc;
  }
  float* dwv = reinterpret_cast<float*>(ws + p.off_dwv);
  hipStream_t wst = st;
  if (use_side) {
    HIP_CHECK_RET(hipEventRecord(p.side.ready[0], st));
    HIP_CHECK_RET(hipStreamWaitEvent(p.side.s, p.side.ready[0], 0));
    wst = p.side.s;
  }
  p.prof.begin(K_WGRAD, st);
  rc = launch_stem_conv_wgrad<T>(p.N, p.OH0, p.OW0, p.Hp, p.Wp, dx0, reinterpret_cast<const T*>(ws + p.off_img4), slab, dwv, wst);
  p.prof.end(st);
  if (p.prof.on) p.prof.flops[K_WGRAD] += conv_flops(u0.s);
  if (rc) return rc;
  if ((rc = stem_wgrad_unpack(dwv, grads + u0.w_off, wst))) return rc;
  if (use_side) {   // final join: everything the side stream was given has finished before backward's last event
    HIP_CHECK_RET(hipEventRecord(p.side.done[0], p.side.s));
    HIP_CHECK_RET(hipStreamWaitEvent(st, p.side.done[0], 0));
  }
  return p.segment_done((int)p.segments.size() - 1, st, false);
}

int Plan::forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
                  float* features, bool training, hipStream_t st) {
  if (dtype == 1) return forward_impl<bf16_t>(*this, image, norm6, params, buffers, ws, features, training, st);
  return forward_impl<float>(*this, image, norm6, params, buffers, ws, features, training, st);
}
int Plan::backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) {
  if (dtype == 1) return backward_impl<bf16_t>(*this, dfeat, params, ws, grads, st);
  return backward_impl<float>(*this, dfeat, params, ws, grads, st);
}
// torchvision's last nn.Conv2d in module order is the final block's last conv (layerN.k.conv3 / conv2)
int Plan::last_conv_shape(int* C, int* OH, int* OW) const {
  const Unit& u = units[blocks.back().units.back()];
  *C = u.s.Cout; *OH = u.s.OH(); *OW = u.s.OW();
  return MMSKIN_OK;
}
int Plan::last_conv_export(const unsigned char* ws, float* x_nchw, hipStream_t st) {
  const Unit& u = units[blocks.back().units.back()];
  if (dtype == 1) return nhwc_to_nchw<bf16_t>(reinterpret_cast<const bf16_t*>(ws + u.x_off), N, u.s.Cout, u.s.OH(), u.s.OW(), x_nchw, st);
  return nhwc_to_nchw<float>(reinterpret_cast<const float*>(ws + u.x_off), N, u.s.Cout, u.s.OH(), u.s.OW(), x_nchw, st);
}
int Plan::last_conv_grad(const float* dfeat, const unsigned char* ws, float* dx_nchw, hipStream_t st) {
  const Unit& u = units[blocks.back().units.back()];
  const float* scale = reinterpret_cast<const float*>(ws + u.coef_off);
  const int HW = u.s.OH() * u.s.OW();
  if (dtype == 1) return gap_relu_bn_grad<bf16_t>(dfeat, reinterpret_cast<const bf16_t*>(ws + u.y_off), scale, N, HW, u.s.Cout, dx_nchw, st);
  return gap_relu_bn_grad<float>(dfeat, reinterpret_cast<const float*>(ws + u.y_off), scale, N, HW, u.s.Cout, dx_nchw, st);
}
int Plan::unit_info(int index, std::string* name, int64_t* info12) const {
  const Unit& u = units[index];
  // the conv weight is the unit's first parameter; find its name through the param table
  for (const TensorInfo& t : params)
    if (t.offset == u.w_off && name) *name = t.name;
  const int64_t v[12] = {(int64_t)u.x_off, (int64_t)u.y_off, (int64_t)u.coef_off, (int64_t)u.rows(), u.s.Cout,
                         u.s.OH(), u.s.OW(), u.s.Cin, u.s.H, u.s.W, (int64_t)off_pool, (int64_t)off_scratch[0]};
  for (int i = 0; i < 12; ++i) info12[i] = v[i];
  return MMSKIN_OK;
}

}  // namespace

PlanBase* make_resnet_plan(int arch, int N, int H, int W, int dtype, int* rc) {
  Plan* p = new Plan();
  p->arch = arch; p->N = N; p->H = H; p->W = W; p->dtype = dtype;
  *rc = build_plan(*p);
  if (*rc) { delete p; return nullptr; }
  return p;
}

// ------------------------------------------------------------------------------------------ C ABI
#include "../../include/mmskin.h"

struct mmskin_backbone { PlanBase* plan = nullptr; };

extern "C" {

int mmskin_backbone_create(const char* arch, int batch, int height, int width, int dtype, mmskin_backbone_t* out) {
  ARG_CHECK(arch && out, "backbone_create: null argument");
  int a = 0;
  if (!strcmp(arch, "resnet-18")) a = 18;
  else if (!strcmp(arch, "resnet-50")) a = 50;
  else if (!strcmp(arch, "densenet169")) a = 169;
  else if (!strcmp(arch, "densenet169-features")) a = 1690;   // norm5 feature map, no ReLU / pool (MDNet)
  else if (!strcmp(arch, "vgg16-features")) a = 16;           // vgg16().features + avgpool -> [N][512][7][7]
  else if (!strcmp(arch, "mobilenet-v2")) a = 2;
  else if (!strcmp(arch, "efficientnet-b0")) a = 100;
  else if (!strcmp(arch, "efficientnet-b7")) a = 107;
  else { mmskin_set_error("backbone_create: Backbone '%s' has no HIP plan", arch); return MMSKIN_ERR_UNSUPPORTED; }
  ARG_CHECK(batch > 0 && height >= 32 && width >= 32, "backbone_create: bad shape %dx%dx%d", batch, height, width);
  ARG_CHECK(dtype == MMSKIN_F32 || dtype == MMSKIN_BF16, "backbone_create: dtype %d", dtype);
  int rc = MMSKIN_OK;
  PlanBase* p = (a == 169 || a == 1690) ? make_densenet_plan(batch, height, width, dtype, a == 1690, &rc)
                : a == 16               ? make_vgg_plan(batch, height, width, dtype, &rc)
                : a == 2                ? make_mobilenet_plan(batch, height, width, dtype, &rc)
                : a >= 100              ? make_efficientnet_plan(a - 100, batch, height, width, dtype, &rc)
                                        : make_resnet_plan(a, batch, height, width, dtype, &rc);
  if (!p) return rc ? rc : MMSKIN_ERR_ARG;
  mmskin_backbone* h = new mmskin_backbone();
  h->plan = p;
  *out = h;
  return MMSKIN_OK;
}

void mmskin_backbone_destroy(mmskin_backbone_t h) {
  if (!h) return;
  delete h->plan;
  delete h;
}

int mmskin_backbone_num_tensors(mmskin_backbone_t h, int kind) {
  return (int)(kind == 0 ? h->plan->params.size() : h->plan->buffers.size());
}

int mmskin_backbone_tensor_info(mmskin_backbone_t h, int kind, int index, char* name, int name_cap,
                                int64_t* offset, int64_t* numel, int* ndim, int64_t* shape4) {
  const std::vector<TensorInfo>& v = kind == 0 ? h->plan->params : h->plan->buffers;
  ARG_CHECK(index >= 0 && index < (int)v.size(), "tensor_info: index %d out of range", index);
  const TensorInfo& t = v[index];
  if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (offset) *offset = t.offset;
  if (numel) *numel = t.numel;
  if (ndim) *ndim = t.ndim;
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
  return MMSKIN_OK;
}

int64_t mmskin_backbone_param_numel(mmskin_backbone_t h) { return h->plan->param_numel; }
int64_t mmskin_backbone_buffer_numel(mmskin_backbone_t h) { return h->plan->buffer_numel; }
int64_t mmskin_backbone_workspace_bytes(mmskin_backbone_t h) { return (int64_t)h->plan->ws_bytes; }
int mmskin_backbone_feature_dim(mmskin_backbone_t h) { return h->plan->feat_dim; }
int mmskin_backbone_feature_hw(mmskin_backbone_t h, int* out_h, int* out_w) {
  ARG_CHECK(h && out_h && out_w, "backbone_feature_hw: null argument");
  *out_h = h->plan->out_h; *out_w = h->plan->out_w;
  return MMSKIN_OK;
}

int mmskin_backbone_profile_enable(mmskin_backbone_t h, int on) {
  h->plan->prof.on = on != 0;
  h->plan->prof.reset();
  return MMSKIN_OK;
}

int mmskin_backbone_profile_read(mmskin_backbone_t h, double* ms7, double* flops7, double* bytes7, int64_t* launches7) {
  Profiler& pr = h->plan->prof;
  HIP_CHECK_RET(hipDeviceSynchronize());
  for (int c = 0; c < K_NCLASS; ++c) { ms7[c] = 0; flops7[c] = pr.flops[c]; bytes7[c] = pr.bytes[c]; launches7[c] = 0; }
  for (size_t i = 0; i < pr.cls.size(); ++i) {
    float ms = 0.f;
    HIP_CHECK_RET(hipEventElapsedTime(&ms, pr.pool[2 * i], pr.pool[2 * i + 1]));
    ms7[pr.cls[i]] += ms;
    launches7[pr.cls[i]] += 1;
  }
  pr.reset();
  return MMSKIN_OK;
}

int mmskin_backbone_num_units(mmskin_backbone_t h) { return h->plan->num_units(); }

int mmskin_backbone_unit_info(mmskin_backbone_t h, int index, char* name, int name_cap, int64_t* info12) {
  ARG_CHECK(index >= 0 && index < h->plan->num_units(), "unit_info: index %d out of range", index);
  std::string nm;
  int rc = h->plan->unit_info(index, &nm, info12);
  if (rc) return rc;
  if (name && name_cap > 0) { strncpy(name, nm.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  return MMSKIN_OK;
}

int mmskin_backbone_forward(mmskin_backbone_t h, const float* image_nchw, const float* params, float* buffers,
                            void* workspace, float* features, int training, void* stream) {
  ARG_CHECK(h && image_nchw && params && (buffers || h->plan->buffer_numel == 0) && workspace && features, "backbone_forward: null argument");
  return h->plan->forward(image_nchw, nullptr, params, buffers, (unsigned char*)workspace, features, training != 0,
                          (hipStream_t)stream);
}

int mmskin_backbone_forward_u8(mmskin_backbone_t h, const uint8_t* image_nhwc, const float* mean_std6, const float* params,
                               float* buffers, void* workspace, float* features, int training, void* stream) {
  ARG_CHECK(h && image_nhwc && mean_std6 && params && (buffers || h->plan->buffer_numel == 0) && workspace && features, "backbone_forward_u8: null argument");
  return h->plan->forward(image_nhwc, mean_std6, params, buffers, (unsigned char*)workspace, features, training != 0,
                          (hipStream_t)stream);
}

int mmskin_backbone_set_option(mmskin_backbone_t h, const char* key, int value) {
  ARG_CHECK(h && key, "backbone_set_option: null argument");
  if (!strcmp(key, "keep_raw_eval")) { h->plan->keep_raw_eval = value != 0; return MMSKIN_OK; }
  if (!strcmp(key, "reuse_staged")) { h->plan->reuse_staged = value != 0; return MMSKIN_OK; }
  mmskin_set_error("backbone_set_option: unknown option '%s'", key);
  return MMSKIN_ERR_ARG;
}
int mmskin_backbone_num_grad_segments(mmskin_backbone_t h, int* count) {
  ARG_CHECK(h && count, "backbone_num_grad_segments: null argument");
  *count = (int)h->plan->segments.size();   // 0: this plan reports no segments (all-reduce the arena after backward)
  return MMSKIN_OK;
}
int mmskin_backbone_grad_segment(mmskin_backbone_t h, int index, int64_t* offset, int64_t* numel) {
  ARG_CHECK(h && offset && numel, "backbone_grad_segment: null argument");
  ARG_CHECK(index >= 0 && index < (int)h->plan->segments.size(), "backbone_grad_segment: index %d", index);
  *offset = h->plan->segments[index].offset;
  *numel = h->plan->segments[index].numel;
  return MMSKIN_OK;
}
int mmskin_backbone_wait_grad_segment(mmskin_backbone_t h, int index, void* stream) {
  ARG_CHECK(h, "backbone_wait_grad_segment: null argument");
  ARG_CHECK(index >= 0 && index < (int)h->plan->segments.size(), "backbone_wait_grad_segment: index %d", index);
  int rc = h->plan->segment_wait(index, (hipStream_t)stream);
  if (rc == MMSKIN_ERR_ARG) mmskin_set_error("backbone_wait_grad_segment: no backward has been enqueued on this plan yet");
  return rc;
}
int mmskin_backbone_set_pointer(mmskin_backbone_t h, const char* key, const void* device_ptr) {
  ARG_CHECK(h && key, "backbone_set_pointer: null argument");
  int rc = h->plan->set_pointer(key, device_ptr);
  if (rc) mmskin_set_error("backbone_set_pointer: '%s' is not a pointer this plan takes", key);
  return rc;
}
int mmskin_backbone_last_conv_shape(mmskin_backbone_t h, int* C, int* OH, int* OW) {
  ARG_CHECK(h && C && OH && OW, "backbone_last_conv_shape: null argument");
  int rc = h->plan->last_conv_shape(C, OH, OW);
  if (rc == MMSKIN_ERR_UNSUPPORTED) mmskin_set_error("backbone_last_conv_*: this plan has no Grad-CAM export");
  return rc;
}
int mmskin_backbone_last_conv_export(mmskin_backbone_t h, const void* workspace, float* x_nchw, void* stream) {
  ARG_CHECK(h && workspace && x_nchw, "backbone_last_conv_export: null argument");
  ARG_CHECK(h->plan->keep_raw_eval, "backbone_last_conv_export: set option keep_raw_eval before the forward");
  int rc = h->plan->last_conv_export((const unsigned char*)workspace, x_nchw, (hipStream_t)stream);
  if (rc == MMSKIN_ERR_UNSUPPORTED) mmskin_set_error("backbone_last_conv_*: this plan has no Grad-CAM export");
  return rc;
}
int mmskin_backbone_last_conv_grad(mmskin_backbone_t h, const float* dfeatures, const void* workspace, float* dx_nchw,
                                   void* stream) {
  ARG_CHECK(h && dfeatures && workspace && dx_nchw, "backbone_last_conv_grad: null argument");
  int rc = h->plan->last_conv_grad(dfeatures, (const unsigned char*)workspace, dx_nchw, (hipStream_t)stream);
  if (rc == MMSKIN_ERR_UNSUPPORTED) mmskin_set_error("backbone_last_conv_*: this plan has no Grad-CAM export");
  return rc;
}

int mmskin_backbone_backward(mmskin_backbone_t h, const float* dfeatures, const float* params, void* workspace,
                             float* param_grads, void* stream) {
  ARG_CHECK(h && dfeatures && params && workspace && param_grads, "backbone_backward: null argument");
  return h->plan->backward(dfeatures, params, (unsigned char*)workspace, param_grads, (hipStream_t)stream);
}

}  // extern "C"

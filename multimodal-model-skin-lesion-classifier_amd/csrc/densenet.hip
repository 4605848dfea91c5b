// DenseNet-169 image-encoder plan executor (torchvision layout: growth 32, blocks 6/12/32/32,
// bn_size 4, 64 stem features, 1664 output features).  Replaces `self.image_encoder(image)` of the
// reference for cnn_model_name == "densenet169" (loadImageModelClassifier.py:84-92 builds
// torchvision's densenet169 and drops its classifier) with the same kernels the ResNet plan uses.
//
// Data layout.  Each dense block owns ONE concatenated activation `cat` [rows][Ctot] (NHWC, compute
// dtype): layer i reads channels [0, Cin_i) and appends its 32 new channels at [Cin_i, Cin_i+32), so
// torch.cat never copies.  BatchNorm statistics of a cat channel are computed once, when the channel
// is written (conv2 epilogue partial sums / one slice_stats pass), into a per-block mean/var table
// every consumer (norm1 of later layers, the transition norm, norm5) reads.  The GEMM kernels want
// compact operands whose channel count is a multiple of 64, so norm1+ReLU writes a compact,
// zero-padded copy t_i [rows][Cp_i] (also what the weight-gradient GEMM needs later), and conv2 runs
// with its 32 output channels padded to 64 (zero weight rows) into a temp that is scattered into cat.
// Backward keeps ONE gradient buffer dcat per block; every layer accumulates its BN-backward result
// into the channel prefix it consumed.
#include "plan.h"

namespace {

struct BNRef {
  int64_t g_off, b_off, rm_off, rv_off;
};

struct DLayer {
  int Cin, Cp;
  BNRef n1, n2;
  int64_t w1_off, w2_off;        // flat param offsets
  int64_t wf1, wd1, wf2, wd2;    // staged element offsets
  size_t t_off, a_off, u_off;    // saved activations (bytes)
  size_t coef1_off, coef2_off;   // floats: 5*Cp | 4*128
  int tab1;                      // stage-table index of conv1 (norm2 folds into it for inference)
};

struct DBlock {
  int H, W, C0, Ctot;
  size_t rows;
  std::vector<DLayer> layers;
  size_t cat_off, dcat_off, tab_off;   // tab: mean[Ctot] | var[Ctot]
};

struct DTrans {
  int C;
  BNRef n;
  int64_t w_off, wf, wd;
  size_t tt_off, coef_off;   // coef: 5*C floats
};

constexpr int GROWTH = 32, BOTTLE = 128, G_PAD = 64;

struct DensePlan : PlanBase {
  bool fmap = false;   // "densenet169-features": output = norm5 feature map [N][C][H/32][W/32] fp32 (MDNet), no ReLU / pool
  int Hp, Wp, OH0, OW0, PH, PW;
  // stem
  int64_t w0_off; BNRef n0; int64_t wf0;
  size_t x0_off, coef0_off, off_pool, off_idx, off_img4;
  std::vector<DBlock> blocks;
  DTrans trans[3];
  BNRef n5; size_t coef5_off, y5_off;
  size_t off_wf, off_wd, off_stat, off_partial, off_coefbwd, off_defer, off_dwv, off_red, off_slab;
  size_t off_sB, off_sB2, off_sU, off_sA, off_sA2, off_sX, off_sZ, off_sC;
  size_t stat_bytes = 0;
  // weight-gradient GEMMs on the side stream: slots 0/1 = conv2 operand (sB) of even/odd layers, 2/3 = conv1
  // operand (sA) of even/odd layers, 4 = transition operand (sC)
  hipEvent_t ev_ready[5] = {}, ev_done[5] = {};
  bool ev_valid[5] = {};
  ~DensePlan() override {
    for (int i = 0; i < 5; ++i) {
      if (ev_ready[i]) (void)hipEventDestroy(ev_ready[i]);
      if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
    }
  }
  int init_events() {
    if (ev_ready[0]) return MMSKIN_OK;
    for (int i = 0; i < 5; ++i) {
      HIP_CHECK_RET(hipEventCreateWithFlags(&ev_ready[i], hipEventDisableTiming));
      HIP_CHECK_RET(hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming));
    }
    return MMSKIN_OK;
  }

  int forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
              float* features, bool training, hipStream_t st) override;
  int backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) override;
};

BNRef add_bn(DensePlan& p, const std::string& name, int C) {
  BNRef r;
  r.g_off = add_tensor(p.params, p.param_numel, name + ".weight", {C});
  r.b_off = add_tensor(p.params, p.param_numel, name + ".bias", {C});
  r.rm_off = add_tensor(p.buffers, p.buffer_numel, name + ".running_mean", {C});
  r.rv_off = add_tensor(p.buffers, p.buffer_numel, name + ".running_var", {C});
  return r;
}

int build_dense_plan(DensePlan& p) {
  const int depths[4] = {6, 12, 32, 32};
  // ---- parameters in torchvision's named_parameters() order (MDNet holds densenet.features itself: no prefix)
  const std::string pre = p.fmap ? "" : "features.";
  p.w0_off = add_tensor(p.params, p.param_numel, pre + "conv0.weight", {64, 3, 7, 7});
  p.n0 = add_bn(p, pre + "norm0", 64);
  ConvShape s0 = {p.N, p.H, p.W, 3, 64, 7, 7, 2, 3};
  p.OH0 = s0.OH(); p.OW0 = s0.OW();
  p.Hp = 2 * p.OH0 + 8; p.Wp = 2 * p.OW0 + 8;
  if (p.Hp < p.H + 6) p.Hp = p.H + 6;
  if (p.Wp < p.W + 6) p.Wp = p.W + 6;
  p.Wp = (p.Wp + 1) / 2 * 2;
  p.PH = (p.OH0 + 2 - 3) / 2 + 1; p.PW = (p.OW0 + 2 - 3) / 2 + 1;
  int c = 64, h = p.PH, w = p.PW;
  for (int bi = 0; bi < 4; ++bi) {
    ARG_CHECK(h >= 1 && w >= 1, "densenet169: input %dx%d too small", p.H, p.W);
    DBlock b;
    b.H = h; b.W = w; b.C0 = c; b.Ctot = c + GROWTH * depths[bi];
    b.rows = (size_t)p.N * h * w;
    for (int i = 0; i < depths[bi]; ++i) {
      DLayer l;
      l.Cin = c + GROWTH * i;
      l.Cp = (l.Cin + 63) / 64 * 64;
      std::string base = pre + "denseblock" + std::to_string(bi + 1) + ".denselayer" + std::to_string(i + 1);
      l.n1 = add_bn(p, base + ".norm1", l.Cin);
      l.w1_off = add_tensor(p.params, p.param_numel, base + ".conv1.weight", {BOTTLE, l.Cin, 1, 1});
      l.n2 = add_bn(p, base + ".norm2", BOTTLE);
      l.w2_off = add_tensor(p.params, p.param_numel, base + ".conv2.weight", {GROWTH, BOTTLE, 3, 3});
      b.layers.push_back(l);
    }
    c = b.Ctot;
    p.blocks.push_back(b);
    if (bi < 3) {
      DTrans& t = p.trans[bi];
      std::string base = pre + "transition" + std::to_string(bi + 1);
      t.C = c;
      t.n = add_bn(p, base + ".norm", c);
      t.w_off = add_tensor(p.params, p.param_numel, base + ".conv.weight", {c / 2, c, 1, 1});
      c /= 2; h /= 2; w /= 2;
    }
  }
  p.n5 = add_bn(p, pre + "norm5", c);
  p.feat_dim = c;
  if (p.fmap) { p.out_h = h; p.out_w = w; }

  // ---- staged weights + stage table (stem first: its slot needs zeroed padding taps)
  int64_t wf = 0, wd = 0;
  auto stage = [&](int64_t src, int Cout, int Cin, int taps, int Cop, int Cip, int64_t& wf_off, int64_t& wd_off) {
    StageDesc d = {};
    d.src_off = src; d.Cout = Cout; d.Cin = Cin; d.taps = taps; d.Cout_pad = Cop; d.Cin_pad = Cip;
    wf_off = wf; wd_off = wd;
    d.fwd_off = wf; d.dgrad_off = wd;
    const int64_t n = (int64_t)Cop * Cip * taps;
    wf += n; wd += n;
    if (n > p.max_stage_elems) p.max_stage_elems = (int)n;
    p.table_host.push_back(d);
  };
  {
    StageDesc d = {};
    d.src_off = p.w0_off; d.Cout = 64; d.Cin = 3; d.taps = 49; d.stem = 1; d.fwd_off = 0;
    p.wf0 = 0; wf = 64 * 256;
    p.max_stage_elems = 64 * 3 * 49;
    p.table_host.push_back(d);
  }
  for (DBlock& b : p.blocks)
    for (DLayer& l : b.layers) {
      l.tab1 = (int)p.table_host.size();
      stage(l.w1_off, BOTTLE, l.Cin, 1, BOTTLE, l.Cp, l.wf1, l.wd1);
      stage(l.w2_off, GROWTH, BOTTLE, 9, G_PAD, BOTTLE, l.wf2, l.wd2);
    }
  for (int i = 0; i < 3; ++i) stage(p.trans[i].w_off, p.trans[i].C / 2, p.trans[i].C, 1, p.trans[i].C / 2, p.trans[i].C, p.trans[i].wf, p.trans[i].wd);

  // ---- workspace
  const size_t es = p.esz();
  size_t cur = 0;
  p.off_img4 = carve(cur, (size_t)p.N * p.Hp * p.Wp * 4 * es);
  p.off_wf = carve(cur, (size_t)wf * es);
  p.off_wd = carve(cur, (size_t)wd * es);
  const size_t rows0 = (size_t)p.N * p.OH0 * p.OW0;
  p.x0_off = carve(cur, rows0 * 64 * es);
  p.coef0_off = carve(cur, 4 * 64 * sizeof(float));
  p.off_pool = carve(cur, (size_t)p.N * p.PH * p.PW * 64 * es);
  p.off_idx = carve(cur, (size_t)p.N * p.PH * p.PW * 64);

  size_t stat_floats = (size_t)stem_conv_stat_rows(p.N, p.OH0, p.OW0) * 64;
  size_t partial_bytes = (size_t)bn_bwd_partial_rows(rows0, 64) * 2 * 64 * sizeof(float);
  size_t slab = stem_wgrad_slab_bytes(p.N, p.OH0, p.OW0);
  size_t small_elems = 0, big_elems = rows0 * 64;
  int maxC = BOTTLE;
  auto need_stat = [&](size_t floats) { if (floats > stat_floats) stat_floats = floats; };
  auto need_partial = [&](size_t rows, int C) {
    size_t a = (size_t)bn_bwd_partial_rows(rows, C) * 2 * C * sizeof(float);
    size_t b2 = ((rows + 127) / 128 + 4) * 2 * (size_t)C * sizeof(float);
    if (a > partial_bytes) partial_bytes = a;
    if (b2 > partial_bytes) partial_bytes = b2;
    if (C > maxC) maxC = C;
  };
  auto need_slab = [&](const ConvShape& s) { size_t v = conv_wgrad_slab_bytes(s); if (v > slab) slab = v; };
  for (int bi = 0; bi < 4; ++bi) {
    DBlock& b = p.blocks[bi];
    b.cat_off = carve(cur, b.rows * b.Ctot * es);
    b.dcat_off = carve(cur, b.rows * b.Ctot * es);
    b.tab_off = carve(cur, 2 * (size_t)b.Ctot * sizeof(float));
    need_stat((size_t)column_stats_rows(b.rows, b.C0) * 2 * b.C0);
    if (b.rows * BOTTLE > small_elems) small_elems = b.rows * BOTTLE;
    for (DLayer& l : b.layers) {
      l.t_off = carve(cur, b.rows * l.Cp * es);
      l.a_off = carve(cur, b.rows * BOTTLE * es);
      l.u_off = carve(cur, b.rows * BOTTLE * es);
      l.coef1_off = carve(cur, 5 * (size_t)l.Cp * sizeof(float));
      l.coef2_off = carve(cur, 4 * (size_t)BOTTLE * sizeof(float));
      {
        StageDesc& d = p.table_host[l.tab1];   // inference: conv1 carries norm2's scale, its epilogue adds shift + ReLU
        d.has_bn = 1;
        d.bn_g_off = l.n2.g_off; d.bn_b_off = l.n2.b_off; d.bn_rm_off = l.n2.rm_off; d.bn_rv_off = l.n2.rv_off;
        d.coef_off = (int64_t)l.coef2_off;
      }
      ConvShape c1 = {p.N, b.H, b.W, l.Cp, BOTTLE, 1, 1, 1, 0}, c2 = {p.N, b.H, b.W, BOTTLE, G_PAD, 3, 3, 1, 1};
      need_stat((size_t)conv_fwd_stat_rows(c1) * BOTTLE);
      need_stat((size_t)conv_fwd_stat_rows(c2) * G_PAD);
      need_partial(b.rows, l.Cp);
      need_partial(b.rows, BOTTLE);
      need_slab(c1); need_slab(c2);
      if (b.rows * l.Cp > big_elems) big_elems = b.rows * l.Cp;
    }
    if (b.rows * b.Ctot > big_elems) big_elems = b.rows * b.Ctot;
    if (bi < 3) {
      DTrans& t = p.trans[bi];
      t.tt_off = carve(cur, b.rows * t.C * es);
      t.coef_off = carve(cur, 5 * (size_t)t.C * sizeof(float));
      ConvShape ct = {p.N, b.H, b.W, t.C, t.C / 2, 1, 1, 1, 0};
      need_partial(b.rows, t.C);
      need_slab(ct);
    }
  }
  DBlock& lb = p.blocks[3];
  p.coef5_off = carve(cur, 5 * (size_t)lb.Ctot * sizeof(float));
  p.y5_off = carve(cur, lb.rows * lb.Ctot * es);
  need_partial(lb.rows, lb.Ctot);
  if (rows0 * 64 > big_elems) big_elems = rows0 * 64;   // stem backward: full-resolution dy / dx
  p.stat_bytes = align_up(stat_floats * sizeof(float), 256);
  p.off_stat = carve(cur, 2 * p.stat_bytes);
  p.off_partial = carve(cur, partial_bytes);
  p.off_coefbwd = carve(cur, 3 * (size_t)maxC * sizeof(float));
  p.off_defer = carve(cur, 2 * (size_t)maxC * sizeof(float));   // per block: running sums of the consumers' cB / cC (backward)
  p.off_dwv = carve(cur, 64 * 256 * sizeof(float));
  p.off_red = carve(cur, bn_reduce_scratch_bytes(maxC));
  p.off_slab = carve(cur, slab);
  p.off_sB = carve(cur, p.blocks[0].rows * G_PAD * es);
  p.off_sB2 = carve(cur, p.blocks[0].rows * G_PAD * es);
  p.off_sU = carve(cur, small_elems * es);
  p.off_sA = carve(cur, small_elems * es);
  p.off_sA2 = carve(cur, small_elems * es);
  p.off_sX = carve(cur, big_elems * es);
  p.off_sZ = carve(cur, big_elems * es);
  p.off_sC = carve(cur, big_elems * es);
  p.ws_bytes = cur;
  return MMSKIN_OK;
}

template <typename T>
int dense_forward(DensePlan& p, const void* image, const float* norm6, const float* params, float* buffers,
                  unsigned char* ws, float* features, bool training, hipStream_t st) {
  const float eps = 1e-5f, mom = 0.1f;
  T* wf = reinterpret_cast<T*>(ws + p.off_wf);
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  float* stat_sum = reinterpret_cast<float*>(ws + p.off_stat);
  float* stat_sq = reinterpret_cast<float*>(ws + p.off_stat + p.stat_bytes);
  double* red = reinterpret_cast<double*>(ws + p.off_red);
  int rc;
  if ((rc = p.ensure_table())) return rc;
  HIP_CHECK_RET(hipMemsetAsync(wf + p.wf0, 0, 64 * 256 * sizeof(T), st));
  PROF(K_STAGE, 0.0, 0.0, stage_weights<T>(p.table_dev, (int)p.table_host.size(), p.max_stage_elems, params, wf, wd, training, st,
                                           training ? nullptr : buffers, eps));
  if (!training) PROF(K_BN_FWD, 0.0, 0.0, bn_eval_table(p.table_dev, (int)p.table_host.size(), BOTTLE, params, buffers, ws, eps, st));

  // batch statistics of cat channels [c0, c0+C) of block b -> its mean/var table
  auto table_from_slice = [&](DBlock& b, int c0, int C) -> int {
    float* tab = reinterpret_cast<float*>(ws + b.tab_off);
    int nr = 0, r;
    p.prof.begin(K_BN_FWD, st);
    struct End { Profiler& pr; hipStream_t s; ~End() { pr.end(s); } } end_guard{p.prof, st};
    if (p.prof.on) p.prof.bytes[K_BN_FWD] += (double)b.rows * C * sizeof(T);
    if ((r = slice_stats<T>(reinterpret_cast<const T*>(ws + b.cat_off) + c0, b.Ctot, C, b.rows, stat_sum, stat_sum + C, &nr, st))) return r;
    return bn_table_finalize(stat_sum, stat_sum + C, nr, 2 * C, C, (double)b.rows, tab + c0, tab + b.Ctot + c0, red, st);
  };

  // ---- stem: conv0 (7x7 s2) -> norm0 -> relu -> maxpool 3x3 s2
  T* img4 = reinterpret_cast<T*>(ws + p.off_img4);
  if (norm6) PROF(K_STEM_MISC, 0.0, 0.0, stem_pack_u8<T>((const uint8_t*)image, p.N, p.H, p.W, p.Hp, p.Wp, norm6, img4, st));
  else PROF(K_STEM_MISC, 0.0, 0.0, stem_pack<T>((const float*)image, p.N, p.H, p.W, p.Hp, p.Wp, img4, st));
  T* x0 = reinterpret_cast<T*>(ws + p.x0_off);
  ConvShape s0 = {p.N, p.H, p.W, 3, 64, 7, 7, 2, 3};
  int stem_rows = 0;
  PROF(K_CONV_FWD, conv_flops(s0), conv_bytes(s0, sizeof(T)),
       launch_stem_conv_fwd<T>(p.N, p.OH0, p.OW0, p.Hp, p.Wp, img4, wf + p.wf0, x0, training ? stat_sum : nullptr,
                               training ? stat_sq : nullptr, st, &stem_rows));
  float* c0 = reinterpret_cast<float*>(ws + p.coef0_off);
  if (training) {
    PROF(K_BN_FWD, 0.0, 0.0, bn_finalize(stat_sum, stat_sq, stem_rows, 64, (double)p.N * p.OH0 * p.OW0,
                     params + p.n0.g_off, params + p.n0.b_off, eps, mom, buffers + p.n0.rm_off, buffers + p.n0.rv_off,
                     c0, c0 + 64, c0 + 128, c0 + 192, red, st));
  } else {
    PROF(K_BN_FWD, 0.0, 0.0, bn_eval_coeffs(64, params + p.n0.g_off, params + p.n0.b_off, buffers + p.n0.rm_off, buffers + p.n0.rv_off, eps, c0, c0 + 64, st));
  }
  T* pool = reinterpret_cast<T*>(ws + p.off_pool);
  PROF(K_STEM_MISC, 0.0, 0.0, stem_bn_relu_pool<T>(x0, c0, c0 + 64, p.N, p.OH0, p.OW0, 64, pool, ws + p.off_idx, st));
  {
    DBlock& b = p.blocks[0];
    PROF(K_STEM_MISC, 0.0, 0.0, slice_scatter<T>(pool, 64, 64, reinterpret_cast<T*>(ws + b.cat_off), b.Ctot, b.rows, st));
    if (training && (rc = table_from_slice(b, 0, 64))) return rc;
  }

  T* sB = reinterpret_cast<T*>(ws + p.off_sB);
  T* sC = reinterpret_cast<T*>(ws + p.off_sC);
  for (int bi = 0; bi < 4; ++bi) {
    DBlock& b = p.blocks[bi];
    T* cat = reinterpret_cast<T*>(ws + b.cat_off);
    float* tab = reinterpret_cast<float*>(ws + b.tab_off);
    const double count = (double)b.rows;
    for (DLayer& l : b.layers) {
      float* k1 = reinterpret_cast<float*>(ws + l.coef1_off);
      float* k2 = reinterpret_cast<float*>(ws + l.coef2_off);
      T* t = reinterpret_cast<T*>(ws + l.t_off);
      T* a = reinterpret_cast<T*>(ws + l.a_off);
      T* u = reinterpret_cast<T*>(ws + l.u_off);
      // norm1 + relu over the channel prefix -> compact padded operand
      PROF(K_BN_FWD, 0.0, 0.0, bn_coef_from_table(tab, tab + b.Ctot, l.Cin, l.Cp, params + l.n1.g_off, params + l.n1.b_off, eps, mom,
                         count, buffers + l.n1.rm_off, buffers + l.n1.rv_off, training, k1, st));
      PROF(K_BN_FWD, 0.0, (double)b.rows * (l.Cin + l.Cp) * sizeof(T),
           slice_pack<T>(cat, b.Ctot, l.Cin, l.Cp, b.rows, k1, k1 + l.Cp, t, st));
      // conv1 1x1 -> norm2 -> relu
      ConvShape c1 = {p.N, b.H, b.W, l.Cp, BOTTLE, 1, 1, 1, 0};
      if (!training) {   // norm2 folded: u = relu(conv1'(t) + shift2) straight from the conv epilogue
        FwdFuse f; f.bias = k2 + BOTTLE; f.relu = true;
        PROF(K_CONV_FWD, conv_flops(c1), conv_bytes(c1, sizeof(T)), launch_conv_fwd<T>(c1, t, wf + l.wf1, u, nullptr, nullptr, st, &f));
      } else {
      PROF(K_CONV_FWD, conv_flops(c1), conv_bytes(c1, sizeof(T)),
           launch_conv_fwd<T>(c1, t, wf + l.wf1, a, training ? stat_sum : nullptr, training ? stat_sq : nullptr, st));
      }
      if (training) {
        PROF(K_BN_FWD, 0.0, 0.0, bn_finalize(stat_sum, stat_sq, conv_fwd_stat_rows(c1), BOTTLE, count, params + l.n2.g_off, params + l.n2.b_off, eps, mom,
                         buffers + l.n2.rm_off, buffers + l.n2.rv_off, k2, k2 + BOTTLE, k2 + 2 * BOTTLE, k2 + 3 * BOTTLE, red, st));
        PROF(K_BN_FWD, 0.0, 2.0 * b.rows * BOTTLE * sizeof(T),
             bn_apply<T>(a, nullptr, k2, k2 + BOTTLE, nullptr, nullptr, u, b.rows, BOTTLE, true, st));
      }
      // conv2 3x3 (32 outputs padded to 64) -> new cat channels + their batch statistics
      ConvShape c2 = {p.N, b.H, b.W, BOTTLE, G_PAD, 3, 3, 1, 1};
      PROF(K_CONV_FWD, conv_flops(c2) / 2, conv_bytes(c2, sizeof(T)),
           launch_conv_fwd<T>(c2, u, wf + l.wf2, sB, training ? stat_sum : nullptr, training ? stat_sq : nullptr, st));
      PROF(K_BN_FWD, 0.0, 2.0 * b.rows * GROWTH * sizeof(T), slice_scatter<T>(sB, G_PAD, GROWTH, cat + l.Cin, b.Ctot, b.rows, st));
      if (training)
        PROF(K_BN_FWD, 0.0, 0.0, bn_table_finalize(stat_sum, stat_sq, conv_fwd_stat_rows(c2), G_PAD, GROWTH, count, tab + l.Cin, tab + b.Ctot + l.Cin, red, st));
    }
    if (bi < 3) {
      // transition: norm -> relu -> conv 1x1 (C -> C/2) -> avgpool 2x2 into the next block's cat prefix
      DTrans& t = p.trans[bi];
      DBlock& nb = p.blocks[bi + 1];
      float* k = reinterpret_cast<float*>(ws + t.coef_off);
      T* tt = reinterpret_cast<T*>(ws + t.tt_off);
      PROF(K_BN_FWD, 0.0, 0.0, bn_coef_from_table(tab, tab + b.Ctot, t.C, t.C, params + t.n.g_off, params + t.n.b_off, eps, mom, count,
                         buffers + t.n.rm_off, buffers + t.n.rv_off, training, k, st));
      PROF(K_BN_FWD, 0.0, 2.0 * b.rows * t.C * sizeof(T), bn_apply<T>(cat, nullptr, k, k + t.C, nullptr, nullptr, tt, b.rows, t.C, true, st));
      ConvShape ct = {p.N, b.H, b.W, t.C, t.C / 2, 1, 1, 1, 0};
      PROF(K_CONV_FWD, conv_flops(ct), conv_bytes(ct, sizeof(T)), launch_conv_fwd<T>(ct, tt, wf + t.wf, sC, nullptr, nullptr, st));
      PROF(K_STEM_MISC, 0.0, 0.0, avgpool2_fwd<T>(sC, p.N, b.H, b.W, t.C / 2, reinterpret_cast<T*>(ws + nb.cat_off), nb.Ctot, st));
      if (training && (rc = table_from_slice(nb, 0, t.C / 2))) return rc;
    }
  }
  // ---- norm5 -> relu -> global average pool
  DBlock& lb = p.blocks[3];
  float* k5 = reinterpret_cast<float*>(ws + p.coef5_off);
  float* tab = reinterpret_cast<float*>(ws + lb.tab_off);
  T* y5 = reinterpret_cast<T*>(ws + p.y5_off);
  const int C5 = lb.Ctot;
  PROF(K_BN_FWD, 0.0, 0.0, bn_coef_from_table(tab, tab + C5, C5, C5, params + p.n5.g_off, params + p.n5.b_off, eps, mom, (double)lb.rows,
                     buffers + p.n5.rm_off, buffers + p.n5.rv_off, training, k5, st));
  PROF(K_BN_FWD, 0.0, 2.0 * lb.rows * C5 * sizeof(T),
       bn_apply<T>(reinterpret_cast<const T*>(ws + lb.cat_off), nullptr, k5, k5 + C5, nullptr, nullptr, y5, lb.rows, C5, !p.fmap, st));
  if (p.fmap) return nhwc_to_nchw<T>(y5, p.N, C5, lb.H, lb.W, features, st);
  return avgpool_fwd<T>(y5, p.N, lb.H * lb.W, C5, features, st);
}

template <typename T>
int dense_backward(DensePlan& p, const float* dfeat, const float* params, unsigned char* ws, float* grads,
                   hipStream_t st) {
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  float* slab = reinterpret_cast<float*>(ws + p.off_slab);
  float* partial = reinterpret_cast<float*>(ws + p.off_partial);
  float* cA = reinterpret_cast<float*>(ws + p.off_coefbwd);
  double* red = reinterpret_cast<double*>(ws + p.off_red);
  T* sBq[2] = {reinterpret_cast<T*>(ws + p.off_sB), reinterpret_cast<T*>(ws + p.off_sB2)};
  T* sAq[2] = {reinterpret_cast<T*>(ws + p.off_sA), reinterpret_cast<T*>(ws + p.off_sA2)};
  T* sU = reinterpret_cast<T*>(ws + p.off_sU);
  T* sX = reinterpret_cast<T*>(ws + p.off_sX);
  T* sZ = reinterpret_cast<T*>(ws + p.off_sZ);
  T* sC = reinterpret_cast<T*>(ws + p.off_sC);
  int rc;

  // Weight-gradient GEMMs only feed the optimizer: they run on the side stream beside the dgrad -> BN-backward
  // chain (one slab, the side stream is in order).  Operand buffers alternate between consecutive layers; the main
  // stream re-acquires a buffer (waits for the wgrad that read it) before overwriting it.
  static const bool side_off = [] { const char* v = getenv("MMSKIN_NO_SIDE_STREAM"); return v && atoi(v) != 0; }();
  const bool use_side = !side_off && !p.prof.on;
  if (use_side) {
    if ((rc = p.side.init())) return rc;
    if ((rc = p.init_events())) return rc;
  }
  for (int i = 0; i < 5; ++i) p.ev_valid[i] = false;
  auto acquire = [&](int slot) -> int {
    if (use_side && p.ev_valid[slot]) HIP_CHECK_RET(hipStreamWaitEvent(st, p.ev_done[slot], 0));
    return MMSKIN_OK;
  };
  auto wgrad_async = [&](int slot, const ConvShape& cs, double flops, const T* dout, const T* in, float* dw, int cov,
                         int civ) -> int {
    hipStream_t wst = st;
    if (use_side) {
      HIP_CHECK_RET(hipEventRecord(p.ev_ready[slot], st));
      HIP_CHECK_RET(hipStreamWaitEvent(p.side.s, p.ev_ready[slot], 0));
      wst = p.side.s;
    }
    p.prof.begin(K_WGRAD, st);
    int r = launch_conv_wgrad<T>(cs, dout, in, slab, dw, wst, cov, civ);
    p.prof.end(st);
    if (p.prof.on) { p.prof.flops[K_WGRAD] += flops; p.prof.bytes[K_WGRAD] += conv_bytes(cs, sizeof(T)); }
    if (r) return r;
    if (use_side) {
      HIP_CHECK_RET(hipEventRecord(p.ev_done[slot], p.side.s));
      p.ev_valid[slot] = true;
    }
    return MMSKIN_OK;
  };
  int layer_no = 0;

  // ---- global average pool <- relu <- norm5: writes the whole of dcat4
  {
    DBlock& lb = p.blocks[3];
    const int C5 = lb.Ctot;
    float* k5 = reinterpret_cast<float*>(ws + p.coef5_off);
    const T* x = reinterpret_cast<const T*>(ws + lb.cat_off);
    const T* y5 = reinterpret_cast<const T*>(ws + p.y5_off);
    float* cB = cA + C5; float* cC = cA + 2 * C5;
    int nr = 0;
    const int mode = p.fmap ? MASK_NONE : MASK_FROM_Y;
    if (p.fmap) rc = nchw_to_nhwc<T>(dfeat, p.N, C5, lb.H, lb.W, sZ, st);
    else rc = avgpool_bwd<T>(dfeat, p.N, lb.H * lb.W, C5, sZ, st);
    if (rc) return rc;
    p.prof.begin(K_BN_BWD, st);
    rc = bn_bwd_reduce<T>(sZ, x, y5, k5, k5 + C5, mode, lb.rows, C5, partial, &nr, st);
    if (!rc) rc = bn_bwd_finalize(partial, nr, C5, (double)lb.rows, params + p.n5.g_off, k5 + 2 * C5, k5 + 3 * C5,
                                  grads + p.n5.g_off, grads + p.n5.b_off, cA, cB, cC, red, st);
    if (!rc) rc = bn_bwd_apply<T>(sZ, x, y5, k5, k5 + C5, mode, cA, cB, cC, reinterpret_cast<T*>(ws + lb.dcat_off),
                                  nullptr, lb.rows, C5, st);
    p.prof.end(st);
    if (rc) return rc;
  }

  // Every later layer of a block adds cA*g + cB*x + cC to the channel prefix it consumed, and x (the concatenated activation) is the
  // same tensor for all of them: the x / constant terms are summed as COEFFICIENTS (sB, sC: bn_bwd_finalize adds into them) and
  // applied once, when a channel's gradient is consumed (its own layer's slice, or the block input at the block's end) -- the
  // per-layer pass then reads g and read-modify-writes dcat only (3 passes over the prefix instead of 4).  MMSKIN_DN_DEFER=0: the old form.
  static const bool defer = [] { const char* v = getenv("MMSKIN_DN_DEFER"); return !v || atoi(v) != 0; }();
  float* sB_ = reinterpret_cast<float*>(ws + p.off_defer);
  for (int bi = 3; bi >= 0; --bi) {
    DBlock& b = p.blocks[bi];
    const T* cat = reinterpret_cast<const T*>(ws + b.cat_off);
    T* dcat = reinterpret_cast<T*>(ws + b.dcat_off);
    const double count = (double)b.rows;
    float* sC_ = sB_ + b.Ctot;
    if (defer) HIP_CHECK_RET(hipMemsetAsync(sB_, 0, 2 * (size_t)b.Ctot * sizeof(float), st));
    for (int li = (int)b.layers.size() - 1; li >= 0; --li) {
      DLayer& l = b.layers[li];
      float* k1 = reinterpret_cast<float*>(ws + l.coef1_off);
      float* k2 = reinterpret_cast<float*>(ws + l.coef2_off);
      const T* t = reinterpret_cast<const T*>(ws + l.t_off);
      const T* a = reinterpret_cast<const T*>(ws + l.a_off);
      const T* u = reinterpret_cast<const T*>(ws + l.u_off);
      ConvShape c1 = {p.N, b.H, b.W, l.Cp, BOTTLE, 1, 1, 1, 0}, c2 = {p.N, b.H, b.W, BOTTLE, G_PAD, 3, 3, 1, 1};
      const int q = layer_no++ & 1;
      T* sB = sBq[q];
      T* sA = sAq[q];
      // gradient of this layer's 32 output channels, padded to the GEMM's 64
      if ((rc = acquire(q))) return rc;
      if (defer)
        PROF(K_BN_BWD, 0.0, 3.0 * b.rows * GROWTH * sizeof(T),
             slice_pack_deferred<T>(dcat + l.Cin, cat + l.Cin, b.Ctot, GROWTH, G_PAD, b.rows, sB_ + l.Cin, sC_ + l.Cin, sB, st));
      else
        PROF(K_BN_BWD, 0.0, 2.0 * b.rows * GROWTH * sizeof(T), slice_pack<T>(dcat + l.Cin, b.Ctot, GROWTH, G_PAD, b.rows, nullptr, nullptr, sB, st));
      // conv2: weight gradient (first 32 rows are real) and data gradient with norm2's mask + sums fused
      if ((rc = wgrad_async(q, c2, conv_flops(c2) / 2, sB, u, grads + l.w2_off, GROWTH, 0))) return rc;
      DgradFuse f2;
      f2.x = a; f2.scale = k2; f2.shift = k2 + BOTTLE; f2.partial = partial;
      PROF(K_CONV_DGRAD, conv_flops(c2) / 2, conv_bytes(c2, sizeof(T), 1), launch_conv_dgrad<T>(c2, sB, wd + l.wd2, sU, (const T*)nullptr, st, &f2));
      {
        float* cB = cA + BOTTLE; float* cC = cA + 2 * BOTTLE;
        p.prof.begin(K_BN_BWD, st);
        if ((rc = acquire(2 + q))) return rc;
        rc = bn_bwd_finalize(partial, f2.rows_written, BOTTLE, count, params + l.n2.g_off, k2 + 2 * BOTTLE, k2 + 3 * BOTTLE,
                             grads + l.n2.g_off, grads + l.n2.b_off, cA, cB, cC, red, st);
        if (!rc) rc = bn_bwd_apply<T>(sU, a, nullptr, k2, k2 + BOTTLE, MASK_NONE, cA, cB, cC, sA, nullptr, b.rows, BOTTLE, st);
        p.prof.end(st);
        if (p.prof.on) p.prof.bytes[K_BN_BWD] += 3.0 * b.rows * BOTTLE * sizeof(T);
        if (rc) return rc;
      }
      // conv1: weight gradient, padded input channels dropped by the reduction
      if ((rc = wgrad_async(2 + q, c1, conv_flops(c1), sA, t, grads + l.w1_off, 0, l.Cin))) return rc;
      // conv1 data gradient with norm1's mask + sums fused: the epilogue reads the raw channel prefix straight from
      // cat (row pitch Ctot); padded channels [Cin, Cp) have scale = shift = 0, so their mask is false
      DgradFuse f1;
      f1.x = cat; f1.x_pitch = b.Ctot; f1.scale = k1; f1.shift = k1 + l.Cp; f1.partial = partial;
      PROF(K_CONV_DGRAD, conv_flops(c1), conv_bytes(c1, sizeof(T), 1), launch_conv_dgrad<T>(c1, sA, wd + l.wd1, sZ, (const T*)nullptr, st, &f1));
      {
        float* cB = cA + l.Cp; float* cC = cA + 2 * l.Cp;
        p.prof.begin(K_BN_BWD, st);
        if (defer) {   // padded channels [Cin, Cp) have gamma = 0: they add zeros to sB / sC
          rc = bn_bwd_finalize(partial, f1.rows_written, l.Cp, count, k1 + 4 * l.Cp, k1 + 2 * l.Cp, k1 + 3 * l.Cp,
                               grads + l.n1.g_off, grads + l.n1.b_off, cA, sB_, sC_, red, st, l.Cin, true);
          if (!rc) rc = slice_accumulate_scaled<T>(dcat, b.Ctot, l.Cin, sZ, l.Cp, cA, b.rows, st);
        } else {
          rc = bn_bwd_finalize(partial, f1.rows_written, l.Cp, count, k1 + 4 * l.Cp, k1 + 2 * l.Cp, k1 + 3 * l.Cp,
                               grads + l.n1.g_off, grads + l.n1.b_off, cA, cB, cC, red, st, l.Cin);
          if (!rc) rc = slice_bn_bwd_accumulate<T>(dcat, cat, b.Ctot, l.Cin, sZ, l.Cp, cA, cB, cC, b.rows, st);
        }
        p.prof.end(st);
        if (p.prof.on) p.prof.bytes[K_BN_BWD] += (defer ? 3.0 : 4.0) * b.rows * l.Cin * sizeof(T);
        if (rc) return rc;
      }
    }
    if (defer)   // the block-input channels [0, C0): every layer of the block consumed them
      PROF(K_BN_BWD, 0.0, 3.0 * b.rows * b.C0 * sizeof(T), slice_affine_inplace<T>(dcat, cat, b.Ctot, b.C0, b.rows, sB_, sC_, st));
    if (bi > 0) {
      // transition bi-1: avgpool <- conv 1x1 <- relu <- norm; writes the whole of the previous block's dcat
      DTrans& tr = p.trans[bi - 1];
      DBlock& pb = p.blocks[bi - 1];
      float* k = reinterpret_cast<float*>(ws + tr.coef_off);
      const T* tt = reinterpret_cast<const T*>(ws + tr.tt_off);
      const T* pcat = reinterpret_cast<const T*>(ws + pb.cat_off);
      ConvShape ct = {p.N, pb.H, pb.W, tr.C, tr.C / 2, 1, 1, 1, 0};
      if ((rc = acquire(4))) return rc;
      PROF(K_STEM_MISC, 0.0, 0.0, avgpool2_bwd<T>(dcat, b.Ctot, p.N, pb.H, pb.W, tr.C / 2, sC, st));
      if ((rc = wgrad_async(4, ct, conv_flops(ct), sC, tt, grads + tr.w_off, 0, 0))) return rc;
      DgradFuse f;
      f.x = pcat; f.scale = k; f.shift = k + tr.C; f.partial = partial;
      PROF(K_CONV_DGRAD, conv_flops(ct), conv_bytes(ct, sizeof(T), 1), launch_conv_dgrad<T>(ct, sC, wd + tr.wd, sZ, (const T*)nullptr, st, &f));
      float* cB = cA + tr.C; float* cC = cA + 2 * tr.C;
      p.prof.begin(K_BN_BWD, st);
      rc = bn_bwd_finalize(partial, f.rows_written, tr.C, (double)pb.rows, params + tr.n.g_off, k + 2 * tr.C, k + 3 * tr.C,
                           grads + tr.n.g_off, grads + tr.n.b_off, cA, cB, cC, red, st);
      if (!rc) rc = bn_bwd_apply<T>(sZ, pcat, nullptr, k, k + tr.C, MASK_NONE, cA, cB, cC, reinterpret_cast<T*>(ws + pb.dcat_off),
                                    nullptr, pb.rows, tr.C, st);
      p.prof.end(st);
      if (p.prof.on) p.prof.bytes[K_BN_BWD] += 3.0 * pb.rows * tr.C * sizeof(T);
      if (rc) return rc;
    }
  }

  // join: the stem reuses the slab and scratch buffers the side stream has been working on
  for (int i = 0; i < 5; ++i)
    if ((rc = acquire(i))) return rc;

  // ---- stem: maxpool <- relu <- norm0 <- conv0
  {
    T* sB = sBq[0];
    DBlock& b = p.blocks[0];
    const size_t rows0 = (size_t)p.N * p.OH0 * p.OW0;
    float* c0 = reinterpret_cast<float*>(ws + p.coef0_off);
    const T* x0 = reinterpret_cast<const T*>(ws + p.x0_off);
    float* cB = cA + 64; float* cC = cA + 128;
    int nr = 0;
    PROF(K_STEM_MISC, 0.0, 0.0, slice_pack<T>(reinterpret_cast<const T*>(ws + b.dcat_off), b.Ctot, 64, 64, b.rows, nullptr, nullptr, sB, st));
    p.prof.begin(K_BN_BWD, st);   // max-pool + ReLU + BatchNorm backward straight from the pooled gradient
    rc = stem_pool_bn_bwd_reduce<T>(sB, ws + p.off_idx, x0, c0, c0 + 64, p.N, p.OH0, p.OW0, 64, partial, &nr, st);
    if (!rc) rc = bn_bwd_finalize(partial, nr, 64, (double)rows0, params + p.n0.g_off, c0 + 128, c0 + 192, grads + p.n0.g_off,
                                  grads + p.n0.b_off, cA, cB, cC, red, st);
    if (!rc) rc = stem_pool_bn_bwd_apply<T>(sB, ws + p.off_idx, x0, c0, c0 + 64, cA, cB, cC, p.N, p.OH0, p.OW0, 64, sX, st);
    p.prof.end(st);
    if (rc) return rc;
    float* dwv = reinterpret_cast<float*>(ws + p.off_dwv);
    ConvShape s0 = {p.N, p.H, p.W, 3, 64, 7, 7, 2, 3};
    PROF(K_WGRAD, conv_flops(s0), 0.0,
         launch_stem_conv_wgrad<T>(p.N, p.OH0, p.OW0, p.Hp, p.Wp, sX, reinterpret_cast<const T*>(ws + p.off_img4), slab, dwv, st));
    return stem_wgrad_unpack(dwv, grads + p.w0_off, st);
  }
}

int DensePlan::forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
                       float* features, bool training, hipStream_t st) {
  if (dtype == 1) return dense_forward<bf16_t>(*this, image, norm6, params, buffers, ws, features, training, st);
  return dense_forward<float>(*this, image, norm6, params, buffers, ws, features, training, st);
}
int DensePlan::backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) {
  if (dtype == 1) return dense_backward<bf16_t>(*this, dfeat, params, ws, grads, st);
  return dense_backward<float>(*this, dfeat, params, ws, grads, st);
}

}  // namespace

PlanBase* make_densenet_plan(int N, int H, int W, int dtype, bool feature_map, int* rc) {
  DensePlan* p = new DensePlan();
  p->N = N; p->H = H; p->W = W; p->dtype = dtype; p->fmap = feature_map;
  *rc = build_dense_plan(*p);
  if (*rc) { delete p; return nullptr; }
  return p;
}

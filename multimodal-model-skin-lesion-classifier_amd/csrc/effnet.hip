// EfficientNet-B0 / B7 image-encoder plan executor (torchvision layout: 3x3/2 stem, MBConv stages with squeeze-
// excitation, SiLU, 3x3 / 5x5 depthwise convolutions, stochastic depth on the residual branches, 1x1 head;
// `classifier = Identity`).  Replaces `self.image_encoder(image)` for cnn_model_name == "efficientnet-b0" /
// "efficientnet-b7" (loadImageModelClassifier.py:102-112).
//
// Same construction as the MobileNet-V2 plan (mobilenet.hip): 1x1 convolutions on the implicit-GEMM kernels with
// channels padded to 64, depthwise convolutions as HBM-bound elementwise kernels, BatchNorm through the statistics
// table.  Additions: SiLU (apply / backward-derivative variants of the BatchNorm kernels), squeeze-excitation (global
// average pool -> two small fp32 Linear layers on the head GEMM -> per-(sample, channel) gate) and row-mode
// stochastic depth (the per-sample keep/scale mask is handed in by the host each training step).
#include <math.h>

#include "plan.h"
#include "../../include/mmskin.h"

namespace {

enum UKind { U_FIRST = 0, U_PW = 1, U_DW = 2 };

struct BNRef {
  int64_t g_off, b_off, rm_off, rv_off;
};

struct SEBlock {   // squeeze-excitation behind a depthwise unit
  int C, Cp, Csq;
  int64_t w1_off, b1_off, w2_off, b2_off;            // flat params: fc1 [Csq][C], fc2 [C][Csq]
  size_t w1p_off, w2p_off, b2p_off;                  // padded fp32 copies (bytes in ws): [Csq][Cp], [Cp][Csq], [Cp]
  size_t s_off, z1_off, a1_off, z2_off, g_off;       // fp32 activations [N][Cp] / [N][Csq]
  size_t yse_off;                                     // gated activation (T) -- the project conv's input
};

struct EUnit {
  int kind, ksize;
  int Cin, Cout, Cinp, Coutp;
  int H, W, stride, OH, OW;
  bool act;                     // SiLU after the BatchNorm
  bool res_last, res_first;
  int se;                       // index into ses (depthwise units) or -1
  int sd;                       // residual-block index for stochastic depth (res_last units) or -1
  int64_t w_off;
  BNRef bn;
  int64_t wf, wd;
  size_t x_off, y_off, coef_off, in_off, res_off;
};

inline int pad64(int c) { return (c + 63) / 64 * 64; }
inline int make_divisible(double v, int divisor = 8) {
  int nv = (int)(v + divisor / 2.0) / divisor * divisor;
  if (nv < divisor) nv = divisor;
  if (nv < 0.9 * v) nv += divisor;
  return nv;
}

struct EffPlan : PlanBase {
  int variant = 0;                  // 0 = B0, 7 = B7
  float eps = 1e-5f, mom = 0.1f;
  std::vector<EUnit> units;
  std::vector<SEBlock> ses;
  int n_res = 0;
  const float* sd_mask = nullptr;   // [n_res][N] keep/scale factors for this training step (null: no stochastic depth)
  int Hp, Wp, stemC;
  size_t off_img8, off_wf, off_wd, off_stat, off_tab, off_partial, off_coefbwd, off_red, off_slab, off_dwv, off_dwpart,
      off_setmp, off_g[4];
  size_t stat_bytes = 0;

  int set_pointer(const char* key, const void* ptr) override {
    if (!strcmp(key, "sd_mask")) { sd_mask = reinterpret_cast<const float*>(ptr); return MMSKIN_OK; }
    return MMSKIN_ERR_ARG;
  }
  int forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
              float* features, bool training, hipStream_t st) override;
  int backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) override;
};

BNRef add_bn(EffPlan& p, const std::string& name, int C) {
  BNRef r;
  r.g_off = add_tensor(p.params, p.param_numel, name + ".weight", {C});
  r.b_off = add_tensor(p.params, p.param_numel, name + ".bias", {C});
  r.rm_off = add_tensor(p.buffers, p.buffer_numel, name + ".running_mean", {C});
  r.rv_off = add_tensor(p.buffers, p.buffer_numel, name + ".running_var", {C});
  return r;
}

int build_eff_plan(EffPlan& p) {
  const double width = p.variant == 7 ? 2.0 : 1.0, depth = p.variant == 7 ? 3.1 : 1.0;
  if (p.variant == 7) { p.eps = 1e-3f; p.mom = 0.01f; }   // torchvision: BatchNorm2d(eps=0.001, momentum=0.01) for B5-B7
  auto adj = [&](int c) { return make_divisible(c * width); };
  auto add_unit = [&](int kind, const std::string& conv_name, const std::string& bn_name, int cin, int cout, int h, int w,
                      int stride, int ksize, bool act) -> int {
    EUnit u = {};
    u.kind = kind; u.ksize = ksize; u.Cin = cin; u.Cout = cout; u.Cinp = kind == U_FIRST ? 3 : pad64(cin); u.Coutp = pad64(cout);
    u.H = h; u.W = w; u.stride = stride; u.se = -1; u.sd = -1;
    const int pad = ksize / 2;
    u.OH = kind == U_PW ? h : (h + 2 * pad - ksize) / stride + 1;
    u.OW = kind == U_PW ? w : (w + 2 * pad - ksize) / stride + 1;
    u.act = act;
    if (kind == U_DW) u.w_off = add_tensor(p.params, p.param_numel, conv_name + ".weight", {cout, 1, ksize, ksize});
    else u.w_off = add_tensor(p.params, p.param_numel, conv_name + ".weight", {cout, cin, kind == U_FIRST ? 3 : 1, kind == U_FIRST ? 3 : 1});
    u.bn = add_bn(p, bn_name, cout);
    p.units.push_back(u);
    return (int)p.units.size() - 1;
  };
  int h = p.H, w = p.W;
  p.stemC = adj(32);
  ARG_CHECK(p.stemC <= 64, "efficientnet: stem width %d", p.stemC);
  {
    int i = add_unit(U_FIRST, "features.0.0", "features.0.1", 3, p.stemC, h, w, 2, 3, true);
    h = p.units[i].OH; w = p.units[i].OW;
  }
  // expand, kernel, stride, in, out, layers  (torchvision _efficientnet_conf)
  const int cfg[7][6] = {{1, 3, 1, 32, 16, 1}, {6, 3, 2, 16, 24, 2}, {6, 5, 2, 24, 40, 2}, {6, 3, 2, 40, 80, 3},
                         {6, 5, 1, 80, 112, 3}, {6, 5, 2, 112, 192, 4}, {6, 3, 1, 192, 320, 1}};
  int last_out = 0;
  for (int si = 0; si < 7; ++si) {
    const int layers = (int)ceil(cfg[si][5] * depth);
    for (int li = 0; li < layers; ++li) {
      const int expand = cfg[si][0], ks = cfg[si][1];
      const int cout = adj(cfg[si][4]);
      const int cin = li == 0 ? adj(cfg[si][3]) : cout;
      const int stride = li == 0 ? cfg[si][2] : 1;
      const int hidden = make_divisible((double)cin * expand);
      const bool res = stride == 1 && cin == cout;
      const std::string base = "features." + std::to_string(si + 1) + "." + std::to_string(li) + ".block.";
      const size_t first = p.units.size();
      int k = 0;
      if (hidden != cin) { add_unit(U_PW, base + "0.0", base + "0.1", cin, hidden, h, w, 1, 1, true); k = 1; }
      const int di = add_unit(U_DW, base + std::to_string(k) + ".0", base + std::to_string(k) + ".1", hidden, hidden, h, w, stride, ks, true);
      h = p.units[di].OH; w = p.units[di].OW;
      SEBlock se = {};
      se.C = hidden; se.Cp = pad64(hidden); se.Csq = cin / 4 > 1 ? cin / 4 : 1;
      const std::string sn = base + std::to_string(k + 1);
      se.w1_off = add_tensor(p.params, p.param_numel, sn + ".fc1.weight", {se.Csq, hidden, 1, 1});
      se.b1_off = add_tensor(p.params, p.param_numel, sn + ".fc1.bias", {se.Csq});
      se.w2_off = add_tensor(p.params, p.param_numel, sn + ".fc2.weight", {hidden, se.Csq, 1, 1});
      se.b2_off = add_tensor(p.params, p.param_numel, sn + ".fc2.bias", {hidden});
      p.units[di].se = (int)p.ses.size();
      p.ses.push_back(se);
      const int pi = add_unit(U_PW, base + std::to_string(k + 2) + ".0", base + std::to_string(k + 2) + ".1", hidden, cout, h, w, 1, 1, false);
      if (res) { p.units[first].res_first = true; p.units[pi].res_last = true; p.units[pi].sd = p.n_res++; }
      ARG_CHECK(h >= 1 && w >= 1, "efficientnet: input %dx%d too small", p.H, p.W);
      last_out = cout;
    }
  }
  const int headC = 4 * last_out;
  add_unit(U_PW, "features.8.0", "features.8.1", last_out, headC, h, w, 1, 1, true);
  p.feat_dim = headC;
  ARG_CHECK(headC % 64 == 0, "efficientnet: head width %d", headC);
  ARG_CHECK(p.units[0].OH <= 240 && p.units[0].OW <= 240, "efficientnet: input %dx%d too large for the weight-gradient kernel", p.H, p.W);
  p.Hp = p.H + 2; p.Wp = (p.W + 4 + 1) / 2 * 2;

  // ---- staged weights
  int64_t wf = 64 * 128, wd = 0;
  p.units[0].wf = 0;
  for (size_t i = 1; i < p.units.size(); ++i) {
    EUnit& u = p.units[i];
    if (u.kind == U_DW) { u.wf = wf; wf += (int64_t)u.ksize * u.ksize * u.Coutp; continue; }
    StageDesc d = {};
    d.src_off = u.w_off; d.Cout = u.Cout; d.Cin = u.Cin; d.taps = 1; d.Cout_pad = u.Coutp; d.Cin_pad = u.Cinp;
    u.wf = wf; u.wd = wd;
    d.fwd_off = wf; d.dgrad_off = wd;
    const int64_t n = (int64_t)u.Coutp * u.Cinp;
    wf += n; wd += n;
    if (n > p.max_stage_elems) p.max_stage_elems = (int)n;
    p.table_host.push_back(d);
  }

  // ---- workspace
  const size_t es = p.esz();
  size_t cur = 0;
  p.off_img8 = carve(cur, (size_t)p.N * p.Hp * p.Wp * 8 * es);
  p.off_wf = carve(cur, (size_t)wf * es);
  p.off_wd = carve(cur, (size_t)(wd > 0 ? wd : 1) * es);
  size_t maxact = 0, stat_floats = 0, partial = 0, dwpart = 0, setmp = 0, slab = vgg_first_wgrad_slab_bytes(p.N, p.units[0].OH, p.units[0].OW);
  int maxCp = 64;
  size_t prev_y = 0, block_in = 0;
  for (size_t i = 0; i < p.units.size(); ++i) {
    EUnit& u = p.units[i];
    const size_t rows = (size_t)p.N * u.OH * u.OW, in_rows = (size_t)p.N * u.H * u.W;
    u.in_off = prev_y;
    if (u.res_first) block_in = prev_y;
    if (u.res_last) u.res_off = block_in;
    u.x_off = carve(cur, rows * u.Coutp * es);
    u.y_off = carve(cur, rows * u.Coutp * es);
    u.coef_off = carve(cur, 5 * (size_t)u.Coutp * sizeof(float));
    prev_y = u.y_off;
    if (u.se >= 0) {
      SEBlock& se = p.ses[u.se];
      se.yse_off = carve(cur, rows * u.Coutp * es);
      se.w1p_off = carve(cur, (size_t)se.Csq * se.Cp * 4);
      se.w2p_off = carve(cur, (size_t)se.Cp * se.Csq * 4);
      se.b2p_off = carve(cur, (size_t)se.Cp * 4);
      se.s_off = carve(cur, (size_t)p.N * se.Cp * 4);
      se.z1_off = carve(cur, (size_t)p.N * se.Csq * 4);
      se.a1_off = carve(cur, (size_t)p.N * se.Csq * 4);
      se.z2_off = carve(cur, (size_t)p.N * se.Cp * 4);
      se.g_off = carve(cur, (size_t)p.N * se.Cp * 4);
      prev_y = se.yse_off;
      // backward temporaries: dgate/dz2 [N][Cp] x2, da1/dz1 [N][Csq] x2, ds [N][Cp], dW1p, dW2p, db2p
      size_t t = (size_t)3 * p.N * se.Cp + (size_t)2 * p.N * se.Csq + (size_t)2 * se.Csq * se.Cp + se.Cp + (size_t)p.N * se.Csq;
      if (t > setmp) setmp = t;
    }
    if (rows * u.Coutp > maxact) maxact = rows * u.Coutp;
    if (u.kind != U_FIRST && in_rows * u.Cinp > maxact) maxact = in_rows * u.Cinp;
    size_t sf = u.kind == U_DW ? (size_t)column_stats_rows(rows, u.Coutp) * u.Coutp : (size_t)((rows + 127) / 128) * u.Coutp;
    if (sf > stat_floats) stat_floats = sf;
    size_t pb = (size_t)bn_bwd_partial_rows(rows, u.Coutp) * 2 * u.Coutp * sizeof(float);
    if (pb > partial) partial = pb;
    if (u.Coutp > maxCp) maxCp = u.Coutp;
    if (u.kind == U_PW) {
      ConvShape s = {p.N, u.H, u.W, u.Cinp, u.Coutp, 1, 1, 1, 0};
      size_t sb = conv_wgrad_slab_bytes(s);
      if (sb > slab) slab = sb;
    }
    if (u.kind == U_DW) {
      size_t f = dwconv3_wgrad_partial_floats(p.N, u.H, u.W, u.Coutp, u.stride, u.ksize);
      if (f > dwpart) dwpart = f;
    }
  }
  p.stat_bytes = align_up(stat_floats * sizeof(float), 256);
  p.off_stat = carve(cur, 2 * p.stat_bytes);
  p.off_tab = carve(cur, 2 * (size_t)maxCp * sizeof(float));
  p.off_partial = carve(cur, partial);
  p.off_coefbwd = carve(cur, 3 * (size_t)maxCp * sizeof(float));
  p.off_red = carve(cur, bn_reduce_scratch_bytes(maxCp));
  p.off_slab = carve(cur, slab);
  p.off_dwv = carve(cur, 64 * 128 * sizeof(float));
  p.off_dwpart = carve(cur, (dwpart > 0 ? dwpart : 1) * sizeof(float));
  p.off_setmp = carve(cur, (setmp > 0 ? setmp : 1) * sizeof(float));
  for (int i = 0; i < 4; ++i) p.off_g[i] = carve(cur, maxact * es);
  p.ws_bytes = cur;
  return MMSKIN_OK;
}

template <typename T>
int eff_forward(EffPlan& p, const void* image, const float* norm6, const float* params, float* buffers,
                unsigned char* ws, float* features, bool training, hipStream_t st) {
  const float eps = p.eps, mom = p.mom;
  T* wf = reinterpret_cast<T*>(ws + p.off_wf);
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  float* stat_sum = reinterpret_cast<float*>(ws + p.off_stat);
  float* stat_sq = reinterpret_cast<float*>(ws + p.off_stat + p.stat_bytes);
  float* tab = reinterpret_cast<float*>(ws + p.off_tab);
  double* red = reinterpret_cast<double*>(ws + p.off_red);
  const float* sd = training ? p.sd_mask : nullptr;
  int rc;
  if ((rc = p.ensure_table())) return rc;
  PROF(K_STAGE, 0.0, 0.0, stage_weights<T>(p.table_dev, (int)p.table_host.size(), p.max_stage_elems, params, wf, wd, training, st));
  PROF(K_STAGE, 0.0, 0.0, vgg_stage_first<T>(params + p.units[0].w_off, wf, st, p.stemC));
  for (EUnit& u : p.units)
    if (u.kind == U_DW) PROF(K_STAGE, 0.0, 0.0, dw_stage_weights<T>(params + u.w_off, u.Cout, u.Coutp, wf + u.wf, st, u.ksize));
  for (SEBlock& se : p.ses) {
    PROF(K_STAGE, 0.0, 0.0, pad_matrix(params + se.w1_off, se.Csq, se.C, se.Csq, se.Cp, reinterpret_cast<float*>(ws + se.w1p_off), st));
    PROF(K_STAGE, 0.0, 0.0, pad_matrix(params + se.w2_off, se.C, se.Csq, se.Cp, se.Csq, reinterpret_cast<float*>(ws + se.w2p_off), st));
    PROF(K_STAGE, 0.0, 0.0, pad_matrix(params + se.b2_off, 1, se.C, 1, se.Cp, reinterpret_cast<float*>(ws + se.b2p_off), st));
  }
  T* img8 = reinterpret_cast<T*>(ws + p.off_img8);
  PROF(K_STEM_MISC, 0.0, 0.0, pack_nhwc8<T>(image, norm6, p.N, p.H, p.W, p.Hp, p.Wp, img8, st));

  for (EUnit& u : p.units) {
    const size_t rows = (size_t)p.N * u.OH * u.OW;
    const T* in = reinterpret_cast<const T*>(ws + u.in_off);
    T* x = reinterpret_cast<T*>(ws + u.x_off);
    T* y = reinterpret_cast<T*>(ws + u.y_off);
    float* k = reinterpret_cast<float*>(ws + u.coef_off);
    const int Cp = u.Coutp;
    int nrows = 0;
    if (u.kind == U_FIRST) {
      ConvShape s = {p.N, p.H, p.W, 3, 64, 3, 3, 2, 1};
      PROF(K_CONV_FWD, conv_flops(s) / 2, conv_bytes(s, sizeof(T)),
           launch_vgg_first_conv_fwd<T>(p.N, p.H, p.W, p.Hp, p.Wp, img8, wf, x, nullptr, st, 2, training ? stat_sum : nullptr,
                                        training ? stat_sq : nullptr));
      nrows = (int)((rows + 127) / 128);
    } else if (u.kind == U_PW) {
      ConvShape s = {p.N, u.H, u.W, u.Cinp, u.Coutp, 1, 1, 1, 0};
      PROF(K_CONV_FWD, conv_flops(s), conv_bytes(s, sizeof(T)),
           launch_conv_fwd<T>(s, in, wf + u.wf, x, training ? stat_sum : nullptr, training ? stat_sq : nullptr, st));
      nrows = conv_fwd_stat_rows(s);
    } else {
      PROF(K_CONV_FWD, 2.0 * u.ksize * u.ksize * rows * Cp, (double)((size_t)p.N * u.H * u.W + rows) * Cp * sizeof(T),
           dwconv3_fwd<T>(in, wf + u.wf, p.N, u.H, u.W, Cp, u.stride, x, st, u.ksize));
      if (training) PROF(K_BN_FWD, 0.0, (double)rows * Cp * sizeof(T), column_stats<T>(x, rows, Cp, stat_sum, stat_sq, &nrows, st));
    }
    if (training)
      PROF(K_BN_FWD, 0.0, 0.0, bn_table_finalize(stat_sum, stat_sq, nrows, Cp, Cp, (double)rows, tab, tab + Cp, red, st));
    PROF(K_BN_FWD, 0.0, 0.0, bn_coef_from_table(tab, tab + Cp, u.Cout, Cp, params + u.bn.g_off, params + u.bn.b_off, eps, mom,
                       (double)rows, buffers + u.bn.rm_off, buffers + u.bn.rv_off, training, k, st));
    const T* res = u.res_last ? reinterpret_cast<const T*>(ws + u.res_off) : nullptr;
    if (res && sd) {   // stochastic depth: y = bn(x) * mask[n] + block input
      PROF(K_BN_FWD, 0.0, 2.0 * rows * Cp * sizeof(T), bn_apply<T>(x, nullptr, k, k + Cp, nullptr, nullptr, y, rows, Cp, false, st));
      PROF(K_BN_FWD, 0.0, 3.0 * rows * Cp * sizeof(T),
           sd_residual_add<T>(y, res, sd + (size_t)u.sd * p.N, p.N, (size_t)u.OH * u.OW * Cp, y, st));
    } else {
      PROF(K_BN_FWD, 0.0, (res ? 3.0 : 2.0) * rows * Cp * sizeof(T),
           bn_apply<T>(x, res, k, k + Cp, nullptr, nullptr, y, rows, Cp, u.act, st, nullptr, -1.f));
    }
    if (u.se >= 0) {   // squeeze-excitation on the depthwise output
      SEBlock& se = p.ses[u.se];
      float* s = reinterpret_cast<float*>(ws + se.s_off);
      float* z1 = reinterpret_cast<float*>(ws + se.z1_off);
      float* a1 = reinterpret_cast<float*>(ws + se.a1_off);
      float* z2 = reinterpret_cast<float*>(ws + se.z2_off);
      float* g = reinterpret_cast<float*>(ws + se.g_off);
      if ((rc = gap_reduce<T>(y, nullptr, p.N, u.OH * u.OW, Cp, 1.f / (float)(u.OH * u.OW), s, st))) return rc;
      if ((rc = mmskin_linear_forward(s, reinterpret_cast<const float*>(ws + se.w1p_off), params + se.b1_off, z1, p.N, Cp, se.Csq, 0, st))) return rc;
      if ((rc = ew_act_fwd(z1, a1, (int64_t)p.N * se.Csq, 0, st))) return rc;
      if ((rc = mmskin_linear_forward(a1, reinterpret_cast<const float*>(ws + se.w2p_off), reinterpret_cast<const float*>(ws + se.b2p_off),
                                      z2, p.N, se.Csq, Cp, 0, st))) return rc;
      if ((rc = ew_act_fwd(z2, g, (int64_t)p.N * Cp, 1, st))) return rc;
      PROF(K_BN_FWD, 0.0, 2.0 * rows * Cp * sizeof(T), se_scale_fwd<T>(y, g, p.N, u.OH * u.OW, Cp, reinterpret_cast<T*>(ws + se.yse_off), st));
    }
  }
  EUnit& last = p.units.back();
  return avgpool_fwd<T>(reinterpret_cast<const T*>(ws + last.y_off), p.N, last.OH * last.OW, last.Coutp, features, st);
}

template <typename T>
int eff_backward(EffPlan& p, const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) {
  T* wf = reinterpret_cast<T*>(ws + p.off_wf);
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  float* slab = reinterpret_cast<float*>(ws + p.off_slab);
  float* partial = reinterpret_cast<float*>(ws + p.off_partial);
  float* cA = reinterpret_cast<float*>(ws + p.off_coefbwd);
  double* red = reinterpret_cast<double*>(ws + p.off_red);
  float* setmp = reinterpret_cast<float*>(ws + p.off_setmp);
  const float* sd = p.sd_mask;
  T* B[4];
  for (int i = 0; i < 4; ++i) B[i] = reinterpret_cast<T*>(ws + p.off_g[i]);
  int rc, cur = 0, reserved = -1;
  auto take = [&](int a, int b) { for (int i = 0; i < 4; ++i) if (i != a && i != b && i != reserved) return i; return -1; };
  EUnit& last = p.units.back();
  if ((rc = avgpool_bwd<T>(dfeat, p.N, last.OH * last.OW, last.Coutp, B[cur], st))) return rc;

  for (int ui = (int)p.units.size() - 1; ui >= 0; --ui) {
    EUnit& u = p.units[ui];
    const size_t rows = (size_t)p.N * u.OH * u.OW;
    const int Cp = u.Coutp, HW = u.OH * u.OW;
    const T* x = reinterpret_cast<const T*>(ws + u.x_off);
    const T* y = reinterpret_cast<const T*>(ws + u.y_off);
    const T* in = reinterpret_cast<const T*>(ws + u.in_off);
    float* k = reinterpret_cast<float*>(ws + u.coef_off);
    float* cB = cA + Cp; float* cC = cA + 2 * Cp;
    if (u.se >= 0) {
      // ---- squeeze-excitation backward: B[cur] = d(y_se) -> d(y_dw) = dyse * g + ds / HW
      SEBlock& se = p.ses[u.se];
      const float* s = reinterpret_cast<const float*>(ws + se.s_off);
      const float* z1 = reinterpret_cast<const float*>(ws + se.z1_off);
      const float* a1 = reinterpret_cast<const float*>(ws + se.a1_off);
      const float* z2 = reinterpret_cast<const float*>(ws + se.z2_off);
      const float* g = reinterpret_cast<const float*>(ws + se.g_off);
      float* t = setmp;
      float* dgate = t; t += (size_t)p.N * Cp;
      float* dz2 = t; t += (size_t)p.N * Cp;
      float* ds = t; t += (size_t)p.N * Cp;
      float* da1 = t; t += (size_t)p.N * se.Csq;
      float* dz1 = t; t += (size_t)p.N * se.Csq;
      float* dw1p = t; t += (size_t)se.Csq * Cp;
      float* dw2p = t; t += (size_t)Cp * se.Csq;
      float* db2p = t; t += Cp;
      if ((rc = gap_reduce<T>(B[cur], y, p.N, HW, Cp, 1.f, dgate, st))) return rc;
      if ((rc = ew_act_bwd(dgate, z2, dz2, (int64_t)p.N * Cp, 1, st))) return rc;
      if ((rc = mmskin_linear_backward(dz2, a1, reinterpret_cast<const float*>(ws + se.w2p_off), nullptr, nullptr, da1, dw2p, db2p,
                                       p.N, se.Csq, Cp, st))) return rc;
      if ((rc = ew_act_bwd(da1, z1, dz1, (int64_t)p.N * se.Csq, 0, st))) return rc;
      if ((rc = mmskin_linear_backward(dz1, s, reinterpret_cast<const float*>(ws + se.w1p_off), nullptr, nullptr, ds, dw1p,
                                       grads + se.b1_off, p.N, Cp, se.Csq, st))) return rc;
      HIP_CHECK_RET(hipMemcpy2DAsync(grads + se.w1_off, (size_t)se.C * 4, dw1p, (size_t)Cp * 4, (size_t)se.C * 4, se.Csq,
                                     hipMemcpyDeviceToDevice, st));
      HIP_CHECK_RET(hipMemcpyAsync(grads + se.w2_off, dw2p, (size_t)se.C * se.Csq * 4, hipMemcpyDeviceToDevice, st));
      HIP_CHECK_RET(hipMemcpyAsync(grads + se.b2_off, db2p, (size_t)se.C * 4, hipMemcpyDeviceToDevice, st));
      const int nb = take(cur, -1);
      PROF(K_BN_BWD, 0.0, 2.0 * rows * Cp * sizeof(T), se_dx<T>(B[cur], g, ds, p.N, HW, Cp, B[nb], st));
      cur = nb;
    }
    if (u.res_last) {
      reserved = cur;   // this gradient is also the residual branch's: keep it until the block's first unit
      if (sd) {         // branch gradient = dy * mask[n]
        const int nb = take(cur, -1);
        PROF(K_BN_BWD, 0.0, 2.0 * rows * Cp * sizeof(T), sd_row_scale<T>(B[cur], sd + (size_t)u.sd * p.N, p.N, (size_t)HW * Cp, B[nb], st));
        cur = nb;
      }
    }
    // ---- BatchNorm (+ SiLU) backward: dy -> dx
    const int a = take(cur, -1);
    const int mode = u.act ? MASK_SILU_X : MASK_NONE;
    int nr = 0;
    p.prof.begin(K_BN_BWD, st);
    rc = bn_bwd_reduce<T>(B[cur], x, y, k, k + Cp, mode, rows, Cp, partial, &nr, st);
    if (!rc) rc = bn_bwd_finalize(partial, nr, Cp, (double)rows, k + 4 * Cp, k + 2 * Cp, k + 3 * Cp, grads + u.bn.g_off,
                                  grads + u.bn.b_off, cA, cB, cC, red, st, u.Cout);
    if (!rc) rc = bn_bwd_apply<T>(B[cur], x, y, k, k + Cp, mode, cA, cB, cC, B[a], nullptr, rows, Cp, st);
    p.prof.end(st);
    if (p.prof.on) p.prof.bytes[K_BN_BWD] += 6.0 * rows * Cp * sizeof(T);
    if (rc) return rc;
    const T* dx = B[a];
    if (u.kind == U_FIRST) {
      float* dwv = reinterpret_cast<float*>(ws + p.off_dwv);
      ConvShape s = {p.N, p.H, p.W, 3, 64, 3, 3, 2, 1};
      PROF(K_WGRAD, conv_flops(s) / 2, 0.0,
           launch_vgg_first_conv_wgrad<T>(p.N, p.H, p.W, p.Hp, p.Wp, dx, reinterpret_cast<const T*>(ws + p.off_img8), slab, dwv, st, 2));
      return vgg_wgrad_unpack_first(dwv, grads + u.w_off, st, p.stemC);
    }
    int b;
    if (u.kind == U_PW) {
      ConvShape s = {p.N, u.H, u.W, u.Cinp, u.Coutp, 1, 1, 1, 0};
      PROF(K_WGRAD, conv_flops(s), conv_bytes(s, sizeof(T)), launch_conv_wgrad<T>(s, dx, in, slab, grads + u.w_off, st, u.Cout, u.Cin));
      if (u.res_first) {
        b = reserved;
        PROF(K_CONV_DGRAD, conv_flops(s), conv_bytes(s, sizeof(T), 1), launch_conv_dgrad<T>(s, dx, wd + u.wd, B[b], B[b], st));
        reserved = -1;
      } else {
        b = take(a, -1);
        PROF(K_CONV_DGRAD, conv_flops(s), conv_bytes(s, sizeof(T)), launch_conv_dgrad<T>(s, dx, wd + u.wd, B[b], (const T*)nullptr, st));
      }
    } else {
      b = take(a, -1);
      PROF(K_WGRAD, 2.0 * u.ksize * u.ksize * rows * Cp, 0.0,
           dwconv3_wgrad<T>(dx, in, p.N, u.H, u.W, Cp, u.stride, reinterpret_cast<float*>(ws + p.off_dwpart), grads + u.w_off, u.Cout, st, u.ksize));
      PROF(K_CONV_DGRAD, 2.0 * u.ksize * u.ksize * rows * Cp, 0.0, dwconv3_dgrad<T>(dx, wf + u.wf, p.N, u.H, u.W, Cp, u.stride, B[b], st, u.ksize));
      if (u.res_first) {   // block whose first unit is the depthwise conv (expand ratio 1, B7 stage 1): add the residual gradient
        PROF(K_CONV_DGRAD, 0.0, 0.0, ew_add<T>(B[b], B[reserved], B[reserved], (size_t)p.N * u.H * u.W * Cp, st));
        b = reserved;
        reserved = -1;
      }
    }
    cur = b;
  }
  return MMSKIN_OK;
}

int EffPlan::forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
                     float* features, bool training, hipStream_t st) {
  if (dtype == 1) return eff_forward<bf16_t>(*this, image, norm6, params, buffers, ws, features, training, st);
  return eff_forward<float>(*this, image, norm6, params, buffers, ws, features, training, st);
}
int EffPlan::backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) {
  if (dtype == 1) return eff_backward<bf16_t>(*this, dfeat, params, ws, grads, st);
  return eff_backward<float>(*this, dfeat, params, ws, grads, st);
}

}  // namespace

PlanBase* make_efficientnet_plan(int variant, int N, int H, int W, int dtype, int* rc) {
  EffPlan* p = new EffPlan();
  p->variant = variant; p->N = N; p->H = H; p->W = W; p->dtype = dtype;
  *rc = build_eff_plan(*p);
  if (*rc) { delete p; return nullptr; }
  return p;
}

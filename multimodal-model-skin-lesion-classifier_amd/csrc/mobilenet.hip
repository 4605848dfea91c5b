// MobileNet-V2 image-encoder plan executor (torchvision layout: 3x3/2 stem, 17 inverted-residual blocks with
// expansion 6, 1x1 head to 1280 channels, global average pool; `classifier = Identity`).  Replaces
// `self.image_encoder(image)` for cnn_model_name == "mobilenet-v2" (loadImageModelClassifier.py:96-100).
//
// The 1x1 convolutions (97 % of the MACs) run on the implicit-GEMM kernels; every activation is kept NHWC with its
// channel count padded to a multiple of 64 (zero weights / zero BatchNorm gain on the padding, so padded channels
// stay exactly zero) and the weight-gradient reductions drop the padding.  The 3x3 depthwise convolutions are
// HBM-bound elementwise-style kernels (ops.hip: dwconv3_*).  BatchNorm uses the shared statistics-table kernels
// (batch statistics from the conv epilogue / one column pass), ReLU6 is the clamp variant of the apply / mask kernels.
#include "plan.h"

namespace {

enum UKind { U_FIRST = 0, U_PW = 1, U_DW = 2 };

struct BNRef {
  int64_t g_off, b_off, rm_off, rv_off;
};

struct MUnit {
  int kind;
  int Cin, Cout, Cinp, Coutp;   // real / padded channels
  int H, W, stride, OH, OW;     // input and output spatial size
  bool relu6;                   // activation after the BatchNorm
  bool res_last;                // last unit of a block with a residual connection: y = bn(x) + block input
  bool res_first;               // first unit of such a block: its data gradient adds the residual branch's gradient
  int64_t w_off;
  BNRef bn;
  int64_t wf, wd;               // staged weights (element offsets; depthwise: [9][Cp] inside the forward buffer)
  size_t x_off, y_off, coef_off;
  size_t in_off;                // input activation (bytes); block input for res_last's residual = res_off
  size_t res_off;
};

inline int pad64(int c) { return (c + 63) / 64 * 64; }

struct MobilePlan : PlanBase {
  std::vector<MUnit> units;
  int Hp, Wp;
  size_t off_img8, off_wf, off_wd, off_stat, off_tab, off_partial, off_coefbwd, off_red, off_slab, off_dwv, off_dwpart,
      off_g[4];
  size_t stat_bytes = 0;

  int forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
              float* features, bool training, hipStream_t st) override;
  int backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) override;
};

BNRef add_bn(MobilePlan& p, const std::string& name, int C) {
  BNRef r;
  r.g_off = add_tensor(p.params, p.param_numel, name + ".weight", {C});
  r.b_off = add_tensor(p.params, p.param_numel, name + ".bias", {C});
  r.rm_off = add_tensor(p.buffers, p.buffer_numel, name + ".running_mean", {C});
  r.rv_off = add_tensor(p.buffers, p.buffer_numel, name + ".running_var", {C});
  return r;
}

int build_mobile_plan(MobilePlan& p) {
  auto add_unit = [&](int kind, const std::string& conv_name, const std::string& bn_name, int cin, int cout, int h, int w,
                      int stride, bool relu6) -> MUnit& {
    MUnit u = {};
    u.kind = kind; u.Cin = cin; u.Cout = cout; u.Cinp = kind == U_FIRST ? 3 : pad64(cin); u.Coutp = pad64(cout);
    u.H = h; u.W = w; u.stride = stride;
    u.OH = kind == U_PW ? h : (h + 2 - 3) / stride + 1;
    u.OW = kind == U_PW ? w : (w + 2 - 3) / stride + 1;
    u.relu6 = relu6;
    if (kind == U_DW) u.w_off = add_tensor(p.params, p.param_numel, conv_name + ".weight", {cout, 1, 3, 3});
    else u.w_off = add_tensor(p.params, p.param_numel, conv_name + ".weight", {cout, cin, kind == U_FIRST ? 3 : 1, kind == U_FIRST ? 3 : 1});
    u.bn = add_bn(p, bn_name, cout);
    p.units.push_back(u);
    return p.units.back();
  };
  int h = p.H, w = p.W;
  {
    MUnit& u = add_unit(U_FIRST, "features.0.0", "features.0.1", 3, 32, h, w, 2, true);
    h = u.OH; w = u.OW;
  }
  const int cfg[7][4] = {{1, 16, 1, 1}, {6, 24, 2, 2}, {6, 32, 3, 2}, {6, 64, 4, 2}, {6, 96, 3, 1}, {6, 160, 3, 2}, {6, 320, 1, 1}};
  int cin = 32, fi = 1;
  for (const auto& c : cfg) {
    for (int i = 0; i < c[2]; ++i, ++fi) {
      const int t = c[0], oup = c[1], stride = i == 0 ? c[3] : 1, hidden = cin * t;
      const bool res = stride == 1 && cin == oup;
      const std::string base = "features." + std::to_string(fi) + ".conv.";
      const size_t first = p.units.size();
      int k = 0;
      if (t != 1) {
        add_unit(U_PW, base + "0.0", base + "0.1", cin, hidden, h, w, 1, true);
        k = 1;
      }
      {
        MUnit& d = add_unit(U_DW, base + std::to_string(k) + ".0", base + std::to_string(k) + ".1", hidden, hidden, h, w, stride, true);
        h = d.OH; w = d.OW;
      }
      add_unit(U_PW, base + std::to_string(k + 1), base + std::to_string(k + 2), hidden, oup, h, w, 1, false);
      if (res) { p.units[first].res_first = true; p.units.back().res_last = true; }
      ARG_CHECK(h >= 1 && w >= 1, "mobilenet-v2: input %dx%d too small", p.H, p.W);
      cin = oup;
    }
  }
  add_unit(U_PW, "features.18.0", "features.18.1", cin, 1280, h, w, 1, true);
  p.feat_dim = 1280;
  ARG_CHECK(p.units[0].OH <= 240 && p.units[0].OW <= 240, "mobilenet-v2: input %dx%d too large for the weight-gradient kernel", p.H, p.W);
  p.Hp = p.H + 2; p.Wp = (p.W + 4 + 1) / 2 * 2;

  // ---- staged weights
  int64_t wf = 64 * 128, wd = 0;   // slot 0: first conv's virtual operand
  p.units[0].wf = 0;
  for (size_t i = 1; i < p.units.size(); ++i) {
    MUnit& u = p.units[i];
    if (u.kind == U_DW) { u.wf = wf; wf += 9 * (int64_t)u.Coutp; continue; }
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
  size_t maxact = 0, stat_floats = 0, partial = 0, dwpart = 0, slab = vgg_first_wgrad_slab_bytes(p.N, p.units[0].OH, p.units[0].OW);
  int maxCp = 64;
  size_t prev_y = 0, block_in = 0;
  for (size_t i = 0; i < p.units.size(); ++i) {
    MUnit& u = p.units[i];
    const size_t rows = (size_t)p.N * u.OH * u.OW, in_rows = (size_t)p.N * u.H * u.W;
    u.in_off = prev_y;
    if (u.res_first) block_in = prev_y;
    if (u.res_last) u.res_off = block_in;
    u.x_off = carve(cur, rows * u.Coutp * es);
    u.y_off = carve(cur, rows * u.Coutp * es);
    u.coef_off = carve(cur, 5 * (size_t)u.Coutp * sizeof(float));
    prev_y = u.y_off;
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
      size_t f = dwconv3_wgrad_partial_floats(p.N, u.H, u.W, u.Coutp, u.stride);
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
  for (int i = 0; i < 4; ++i) p.off_g[i] = carve(cur, maxact * es);
  p.ws_bytes = cur;
  return MMSKIN_OK;
}

template <typename T>
int mobile_forward(MobilePlan& p, const void* image, const float* norm6, const float* params, float* buffers,
                   unsigned char* ws, float* features, bool training, hipStream_t st) {
  const float eps = 1e-5f, mom = 0.1f;
  T* wf = reinterpret_cast<T*>(ws + p.off_wf);
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  float* stat_sum = reinterpret_cast<float*>(ws + p.off_stat);
  float* stat_sq = reinterpret_cast<float*>(ws + p.off_stat + p.stat_bytes);
  float* tab = reinterpret_cast<float*>(ws + p.off_tab);
  double* red = reinterpret_cast<double*>(ws + p.off_red);
  int rc;
  if ((rc = p.ensure_table())) return rc;
  PROF(K_STAGE, 0.0, 0.0, stage_weights<T>(p.table_dev, (int)p.table_host.size(), p.max_stage_elems, params, wf, wd, training, st));
  PROF(K_STAGE, 0.0, 0.0, vgg_stage_first<T>(params + p.units[0].w_off, wf, st, 32));
  for (MUnit& u : p.units)
    if (u.kind == U_DW) PROF(K_STAGE, 0.0, 0.0, dw_stage_weights<T>(params + u.w_off, u.Cout, u.Coutp, wf + u.wf, st));
  T* img8 = reinterpret_cast<T*>(ws + p.off_img8);
  PROF(K_STEM_MISC, 0.0, 0.0, pack_nhwc8<T>(image, norm6, p.N, p.H, p.W, p.Hp, p.Wp, img8, st));

  for (MUnit& u : p.units) {
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
      PROF(K_CONV_FWD, 18.0 * rows * Cp, (double)((size_t)p.N * u.H * u.W + rows) * Cp * sizeof(T),
           dwconv3_fwd<T>(in, wf + u.wf, p.N, u.H, u.W, Cp, u.stride, x, st));
      if (training) PROF(K_BN_FWD, 0.0, (double)rows * Cp * sizeof(T), column_stats<T>(x, rows, Cp, stat_sum, stat_sq, &nrows, st));
    }
    if (training)
      PROF(K_BN_FWD, 0.0, 0.0, bn_table_finalize(stat_sum, stat_sq, nrows, Cp, Cp, (double)rows, tab, tab + Cp, red, st));
    PROF(K_BN_FWD, 0.0, 0.0, bn_coef_from_table(tab, tab + Cp, u.Cout, Cp, params + u.bn.g_off, params + u.bn.b_off, eps, mom,
                       (double)rows, buffers + u.bn.rm_off, buffers + u.bn.rv_off, training, k, st));
    const T* res = u.res_last ? reinterpret_cast<const T*>(ws + u.res_off) : nullptr;
    PROF(K_BN_FWD, 0.0, (res ? 3.0 : 2.0) * rows * Cp * sizeof(T),
         bn_apply<T>(x, res, k, k + Cp, nullptr, nullptr, y, rows, Cp, u.relu6, st, nullptr, 6.f));
  }
  MUnit& last = p.units.back();
  return avgpool_fwd<T>(reinterpret_cast<const T*>(ws + last.y_off), p.N, last.OH * last.OW, last.Coutp, features, st);
}

template <typename T>
int mobile_backward(MobilePlan& p, const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) {
  T* wf = reinterpret_cast<T*>(ws + p.off_wf);
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  float* slab = reinterpret_cast<float*>(ws + p.off_slab);
  float* partial = reinterpret_cast<float*>(ws + p.off_partial);
  float* cA = reinterpret_cast<float*>(ws + p.off_coefbwd);
  double* red = reinterpret_cast<double*>(ws + p.off_red);
  T* B[4];
  for (int i = 0; i < 4; ++i) B[i] = reinterpret_cast<T*>(ws + p.off_g[i]);
  int rc, cur = 0, reserved = -1;
  auto take = [&](int a, int b) { for (int i = 0; i < 4; ++i) if (i != a && i != b && i != reserved) return i; return -1; };
  MUnit& last = p.units.back();
  if ((rc = avgpool_bwd<T>(dfeat, p.N, last.OH * last.OW, last.Coutp, B[cur], st))) return rc;

  for (int ui = (int)p.units.size() - 1; ui >= 0; --ui) {
    MUnit& u = p.units[ui];
    const size_t rows = (size_t)p.N * u.OH * u.OW;
    const int Cp = u.Coutp;
    const T* x = reinterpret_cast<const T*>(ws + u.x_off);
    const T* y = reinterpret_cast<const T*>(ws + u.y_off);
    const T* in = reinterpret_cast<const T*>(ws + u.in_off);
    float* k = reinterpret_cast<float*>(ws + u.coef_off);
    float* cB = cA + Cp; float* cC = cA + 2 * Cp;
    if (u.res_last) reserved = cur;   // this gradient is also the residual branch's: keep it until the block's first unit
    // ---- BatchNorm (+ ReLU6) backward: dy -> dx
    const int a = take(cur, -1);
    const int mode = u.relu6 ? MASK_FROM_Y6 : MASK_NONE;
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
    // ---- convolution backward
    if (u.kind == U_FIRST) {
      float* dwv = reinterpret_cast<float*>(ws + p.off_dwv);
      ConvShape s = {p.N, p.H, p.W, 3, 64, 3, 3, 2, 1};
      PROF(K_WGRAD, conv_flops(s) / 2, 0.0,
           launch_vgg_first_conv_wgrad<T>(p.N, p.H, p.W, p.Hp, p.Wp, dx, reinterpret_cast<const T*>(ws + p.off_img8), slab, dwv, st, 2));
      return vgg_wgrad_unpack_first(dwv, grads + u.w_off, st, 32);
    }
    int b;
    if (u.kind == U_PW) {
      ConvShape s = {p.N, u.H, u.W, u.Cinp, u.Coutp, 1, 1, 1, 0};
      PROF(K_WGRAD, conv_flops(s), conv_bytes(s, sizeof(T)), launch_conv_wgrad<T>(s, dx, in, slab, grads + u.w_off, st, u.Cout, u.Cin));
      if (u.res_first) {   // add the residual branch's gradient in the epilogue, in place on the buffer that holds it
        b = reserved;
        PROF(K_CONV_DGRAD, conv_flops(s), conv_bytes(s, sizeof(T), 1), launch_conv_dgrad<T>(s, dx, wd + u.wd, B[b], B[b], st));
        reserved = -1;
      } else {
        b = take(a, -1);
        PROF(K_CONV_DGRAD, conv_flops(s), conv_bytes(s, sizeof(T)), launch_conv_dgrad<T>(s, dx, wd + u.wd, B[b], (const T*)nullptr, st));
      }
    } else {
      b = take(a, -1);
      PROF(K_WGRAD, 18.0 * rows * Cp, 0.0,
           dwconv3_wgrad<T>(dx, in, p.N, u.H, u.W, Cp, u.stride, reinterpret_cast<float*>(ws + p.off_dwpart), grads + u.w_off, u.Cout, st));
      PROF(K_CONV_DGRAD, 18.0 * rows * Cp, 0.0, dwconv3_dgrad<T>(dx, wf + u.wf, p.N, u.H, u.W, Cp, u.stride, B[b], st));
    }
    cur = b;
  }
  return MMSKIN_OK;
}

int MobilePlan::forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
                        float* features, bool training, hipStream_t st) {
  if (dtype == 1) return mobile_forward<bf16_t>(*this, image, norm6, params, buffers, ws, features, training, st);
  return mobile_forward<float>(*this, image, norm6, params, buffers, ws, features, training, st);
}
int MobilePlan::backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) {
  if (dtype == 1) return mobile_backward<bf16_t>(*this, dfeat, params, ws, grads, st);
  return mobile_backward<float>(*this, dfeat, params, ws, grads, st);
}

}  // namespace

PlanBase* make_mobilenet_plan(int N, int H, int W, int dtype, int* rc) {
  MobilePlan* p = new MobilePlan();
  p->N = N; p->H = H; p->W = W; p->dtype = dtype;
  *rc = build_mobile_plan(*p);
  if (*rc) { delete p; return nullptr; }
  return p;
}

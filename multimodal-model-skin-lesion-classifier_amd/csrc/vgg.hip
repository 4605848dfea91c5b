// VGG-16 convolutional trunk plan executor: torchvision `vgg16().features` (13 x [conv3x3 + bias + ReLU], five
// 2x2 max-pools) followed by `avgpool` = AdaptiveAvgPool2d(7).  Replaces that part of `self.image_encoder(image)`
// for cnn_model_name == "vgg16" (loadImageModelClassifier.py:77-81); the two remaining classifier Linear layers run
// on the head GEMM from Python.  Output: [N][512][7][7] fp32 (NCHW, so `flatten(1)` matches the reference).
//
// No BatchNorm here: every conv writes relu(acc + bias) straight from its epilogue (the only activation pass),
// the dgrad epilogue applies the producer's ReLU mask (y > 0) and accumulates the bias-gradient column sums, and
// a pool boundary uses one fused max-pool-backward + ReLU-mask kernel.
#include "plan.h"

namespace {

struct VConv {
  int Cin, Cout, H, W;       // input = output spatial size (3x3 s1 p1)
  bool pool_after;
  int64_t w_off, b_off;      // flat params
  int64_t wf, wd;            // staged
  size_t y_off;              // relu(conv + bias), NHWC (bytes)
  size_t p_off, idx_off;     // pooled output + argmax bytes when pool_after
};

struct VggPlan : PlanBase {
  std::vector<VConv> convs;
  int Hp, Wp;                 // padded NHWC8 image
  int fH, fW;                 // spatial size after the five pools
  size_t off_img8, off_wf, off_wd, off_partial, off_red, off_slab, off_dwv, off_g[2];

  int forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
              float* features, bool training, hipStream_t st) override;
  int backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) override;
};

int build_vgg_plan(VggPlan& p) {
  const int cfg[] = {64, 64, -1, 128, 128, -1, 256, 256, 256, -1, 512, 512, 512, -1, 512, 512, 512, -1};
  int cin = 3, h = p.H, w = p.W, idx = 0;
  for (int v : cfg) {
    if (v < 0) {
      p.convs.back().pool_after = true;
      h /= 2; w /= 2;
      ARG_CHECK(h >= 1 && w >= 1, "vgg16: input %dx%d too small", p.H, p.W);
      ++idx;
      continue;
    }
    VConv c = {};
    c.Cin = cin; c.Cout = v; c.H = h; c.W = w; c.pool_after = false;
    const std::string name = std::to_string(idx);   // index inside torchvision's features Sequential
    c.w_off = add_tensor(p.params, p.param_numel, name + ".weight", {v, cin, 3, 3});
    c.b_off = add_tensor(p.params, p.param_numel, name + ".bias", {v});
    p.convs.push_back(c);
    cin = v;
    idx += 2;   // conv, relu
  }
  p.fH = h; p.fW = w;
  p.feat_dim = 512; p.out_h = 7; p.out_w = 7;
  ARG_CHECK(p.H <= 240 && p.W <= 240, "vgg16: input %dx%d too large for the weight-gradient kernel's pixel stepping", p.H, p.W);
  p.Hp = p.H + 2; p.Wp = (p.W + 4 + 1) / 2 * 2;

  int64_t wf = 64 * 128, wd = 0;   // slot 0 of wf: the first conv's virtual operand
  for (size_t i = 1; i < p.convs.size(); ++i) {
    VConv& c = p.convs[i];
    StageDesc d = {};
    d.src_off = c.w_off; d.Cout = c.Cout; d.Cin = c.Cin; d.taps = 9;
    c.wf = wf; c.wd = wd;
    d.fwd_off = wf; d.dgrad_off = wd;
    const int64_t n = (int64_t)c.Cout * c.Cin * 9;
    wf += n; wd += n;
    if (n > p.max_stage_elems) p.max_stage_elems = (int)n;
    p.table_host.push_back(d);
  }
  p.convs[0].wf = 0;

  const size_t es = p.esz();
  size_t cur = 0;
  p.off_img8 = carve(cur, (size_t)p.N * p.Hp * p.Wp * 8 * es);
  p.off_wf = carve(cur, (size_t)wf * es);
  p.off_wd = carve(cur, (size_t)wd * es);
  size_t maxact = 0, partial = 0, slab = vgg_first_wgrad_slab_bytes(p.N, p.H, p.W);
  int maxC = 64;
  for (VConv& c : p.convs) {
    const size_t rows = (size_t)p.N * c.H * c.W;
    c.y_off = carve(cur, rows * c.Cout * es);
    if (c.pool_after) {
      const size_t prow = (size_t)p.N * (c.H / 2) * (c.W / 2);
      c.p_off = carve(cur, prow * c.Cout * es);
      c.idx_off = carve(cur, prow * c.Cout);
    }
    if (rows * c.Cout > maxact) maxact = rows * c.Cout;
    size_t pr = ((rows + 127) / 128 + 4) * 2 * (size_t)c.Cout * sizeof(float);
    size_t pr2 = (size_t)column_stats_rows(rows, c.Cout) * 2 * c.Cout * sizeof(float);
    if (pr > partial) partial = pr;
    if (pr2 > partial) partial = pr2;
    if (c.Cout > maxC) maxC = c.Cout;
    if (c.Cin != 3) {
      ConvShape s = {p.N, c.H, c.W, c.Cin, c.Cout, 3, 3, 1, 1};
      size_t sb = conv_wgrad_slab_bytes(s);
      if (sb > slab) slab = sb;
    }
  }
  p.off_partial = carve(cur, partial);
  p.off_red = carve(cur, bn_reduce_scratch_bytes(maxC));
  p.off_slab = carve(cur, slab);
  p.off_dwv = carve(cur, 64 * 128 * sizeof(float));
  p.off_g[0] = carve(cur, maxact * es);
  p.off_g[1] = carve(cur, maxact * es);
  p.ws_bytes = cur;
  return MMSKIN_OK;
}

template <typename T>
int vgg_forward(VggPlan& p, const void* image, const float* norm6, const float* params, unsigned char* ws, float* features,
                hipStream_t st) {
  T* wf = reinterpret_cast<T*>(ws + p.off_wf);
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  int rc;
  if ((rc = p.ensure_table())) return rc;
  PROF(K_STAGE, 0.0, 0.0, stage_weights<T>(p.table_dev, (int)p.table_host.size(), p.max_stage_elems, params, wf, wd, true, st));
  PROF(K_STAGE, 0.0, 0.0, vgg_stage_first<T>(params + p.convs[0].w_off, wf, st));
  T* img8 = reinterpret_cast<T*>(ws + p.off_img8);
  PROF(K_STEM_MISC, 0.0, 0.0, pack_nhwc8<T>(image, norm6, p.N, p.H, p.W, p.Hp, p.Wp, img8, st));
  const T* cur = nullptr;
  for (size_t i = 0; i < p.convs.size(); ++i) {
    VConv& c = p.convs[i];
    T* y = reinterpret_cast<T*>(ws + c.y_off);
    FwdFuse f; f.bias = params + c.b_off; f.relu = true;
    ConvShape s = {p.N, c.H, c.W, c.Cin, c.Cout, 3, 3, 1, 1};
    if (i == 0) {
      PROF(K_CONV_FWD, conv_flops(s), conv_bytes(s, sizeof(T)),
           launch_vgg_first_conv_fwd<T>(p.N, p.H, p.W, p.Hp, p.Wp, img8, wf, y, &f, st));
    } else {
      PROF(K_CONV_FWD, conv_flops(s), conv_bytes(s, sizeof(T)), launch_conv_fwd<T>(s, cur, wf + c.wf, y, nullptr, nullptr, st, &f));
    }
    cur = y;
    if (c.pool_after) {
      T* pl = reinterpret_cast<T*>(ws + c.p_off);
      PROF(K_STEM_MISC, 0.0, 0.0, maxpool2_fwd<T>(y, p.N, c.H, c.W, c.Cout, pl, ws + c.idx_off, st));
      cur = pl;
    }
  }
  return adaptive_avgpool_fwd<T>(cur, p.N, p.fH, p.fW, 512, 7, 7, features, st);
}

template <typename T>
int vgg_backward(VggPlan& p, const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) {
  T* wd = reinterpret_cast<T*>(ws + p.off_wd);
  float* slab = reinterpret_cast<float*>(ws + p.off_slab);
  float* partial = reinterpret_cast<float*>(ws + p.off_partial);
  double* red = reinterpret_cast<double*>(ws + p.off_red);
  T* G[2] = {reinterpret_cast<T*>(ws + p.off_g[0]), reinterpret_cast<T*>(ws + p.off_g[1])};
  int rc, gi = 0;
  // gradient of the last pooled map
  PROF(K_STEM_MISC, 0.0, 0.0, adaptive_avgpool_bwd<T>(dfeat, p.N, p.fH, p.fW, 512, 7, 7, G[gi], st));
  bool have_dz = false;     // G[gi] holds dz of conv i (masked, with bias partials in `partial`) instead of a pooled-map gradient
  int part_rows = 0, part_stride = 0;
  for (int i = (int)p.convs.size() - 1; i >= 0; --i) {
    VConv& c = p.convs[i];
    const T* y = reinterpret_cast<const T*>(ws + c.y_off);
    const size_t rows = (size_t)p.N * c.H * c.W;
    if (!have_dz) {   // G[gi] = gradient of this conv's pooled output: un-pool + ReLU mask in one pass, then column sums
      T* dz = G[gi ^ 1];
      PROF(K_BN_BWD, 0.0, 0.0, maxpool2_bwd_relu<T>(G[gi], ws + c.idx_off, y, p.N, c.H, c.W, c.Cout, dz, st));
      gi ^= 1;
      PROF(K_BN_BWD, 0.0, 0.0, column_stats<T>(G[gi], rows, c.Cout, partial, partial + (size_t)column_stats_rows(rows, c.Cout) * c.Cout, &part_rows, st));
      part_stride = c.Cout;
    }
    const T* dz = G[gi];
    PROF(K_BN_BWD, 0.0, 0.0, bias_grad_finalize(partial, part_rows, part_stride, c.Cout, grads + c.b_off, red, st));
    ConvShape s = {p.N, c.H, c.W, c.Cin, c.Cout, 3, 3, 1, 1};
    if (i == 0) {
      float* dwv = reinterpret_cast<float*>(ws + p.off_dwv);
      PROF(K_WGRAD, conv_flops(s), 0.0,
           launch_vgg_first_conv_wgrad<T>(p.N, p.H, p.W, p.Hp, p.Wp, dz, reinterpret_cast<const T*>(ws + p.off_img8), slab, dwv, st));
      return vgg_wgrad_unpack_first(dwv, grads + c.w_off, st);
    }
    VConv& prev = p.convs[i - 1];
    const T* in = prev.pool_after ? reinterpret_cast<const T*>(ws + prev.p_off) : reinterpret_cast<const T*>(ws + prev.y_off);
    PROF(K_WGRAD, conv_flops(s), conv_bytes(s, sizeof(T)), launch_conv_wgrad<T>(s, dz, in, slab, grads + c.w_off, st));
    T* din = G[gi ^ 1];
    if (prev.pool_after) {   // input was a pooled map: plain data gradient; the pool / ReLU are undone next iteration
      PROF(K_CONV_DGRAD, conv_flops(s), conv_bytes(s, sizeof(T)), launch_conv_dgrad<T>(s, dz, wd + c.wd, din, (const T*)nullptr, st));
      have_dz = false;
    } else {                 // input was relu(conv_{i-1}): mask + bias-gradient column sums in the epilogue
      DgradFuse f;
      f.mask_y = ws + prev.y_off; f.partial = partial;
      PROF(K_CONV_DGRAD, conv_flops(s), conv_bytes(s, sizeof(T), 1), launch_conv_dgrad<T>(s, dz, wd + c.wd, din, (const T*)nullptr, st, &f));
      have_dz = true;
      part_rows = f.rows_written; part_stride = 2 * c.Cin;
    }
    gi ^= 1;
  }
  return MMSKIN_OK;
}

int VggPlan::forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
                     float* features, bool training, hipStream_t st) {
  (void)buffers; (void)training;   // no BatchNorm: train and eval run the same kernels
  if (dtype == 1) return vgg_forward<bf16_t>(*this, image, norm6, params, ws, features, st);
  return vgg_forward<float>(*this, image, norm6, params, ws, features, st);
}
int VggPlan::backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) {
  if (dtype == 1) return vgg_backward<bf16_t>(*this, dfeat, params, ws, grads, st);
  return vgg_backward<float>(*this, dfeat, params, ws, grads, st);
}

}  // namespace

PlanBase* make_vgg_plan(int N, int H, int W, int dtype, int* rc) {
  VggPlan* p = new VggPlan();
  p->N = N; p->H = H; p->W = W; p->dtype = dtype;
  *rc = build_vgg_plan(*p);
  if (*rc) { delete p; return nullptr; }
  return p;
}

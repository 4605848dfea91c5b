// Shared scaffolding of the image-encoder plan executors (backbone.hip: ResNet, densenet.hip: DenseNet).
// A plan is built once per (architecture, batch, H, W, dtype): it fixes the flat parameter / buffer
// layout (torchvision named_parameters() order), the workspace carve-up and the launch sequence, so
// that one C-ABI call runs a whole forward or backward on the caller's stream.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "conv.h"
#include "ops.h"

struct TensorInfo {
  std::string name;
  int64_t offset, numel;
  int ndim;
  int64_t shape[4];
};

enum KClass { K_CONV_FWD = 0, K_CONV_DGRAD, K_WGRAD, K_BN_FWD, K_BN_BWD, K_STAGE, K_STEM_MISC, K_NCLASS };

// Optional per-kernel-class timing with HIP events recorded on the launch stream (bench.py's live
// roofline).  Off by default: the timed region of a benchmark never pays for it.
struct Profiler {
  bool on = false;
  std::vector<hipEvent_t> pool;
  std::vector<int> cls;       // class of event pair i (events 2i, 2i+1)
  size_t used = 0;
  double flops[K_NCLASS] = {0};
  double bytes[K_NCLASS] = {0};
  hipEvent_t get() {
    if (used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
    return pool[used++];
  }
  void begin(int c, hipStream_t st) { if (on) { cls.push_back(c); (void)hipEventRecord(get(), st); } }
  void end(hipStream_t st) { if (on) (void)hipEventRecord(get(), st); }
  void reset() { used = 0; cls.clear(); for (int i = 0; i < K_NCLASS; ++i) { flops[i] = 0; bytes[i] = 0; } }
};

// Weight-gradient GEMMs only feed the optimizer, so they run on a second HIP stream beside the
// dgrad -> BN-backward chain of the main stream (fills launch tails and latency-bound phases).
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t ready[3] = {nullptr, nullptr, nullptr};   // main: buffer i holds a fresh dX
  hipEvent_t done[3] = {nullptr, nullptr, nullptr};    // side: the wgrad reading buffer i has finished
  bool done_valid[3] = {false, false, false};
  hipEvent_t f_ready = nullptr, f_done = nullptr;      // forward: block input ready / downsample branch finished
  hipEvent_t f_staged = nullptr;                       // forward: weights of the residual stages staged (beside the stem)
  // backward: the downsample branch of a block (BatchNorm-backward apply + dgrad) on a stream of its own, beside conv3 .. conv2
  hipStream_t s2 = nullptr;
  hipEvent_t d_ready = nullptr, d_done = nullptr;      // main: block-output gradient + its partial sums ready / s2: branch gradient written
  // algebraic BatchNorm backward (backbone.hip): the weight-gradient stream reads the block-output gradient buffer itself
  hipEvent_t g_ready = nullptr, g_done[2] = {nullptr, nullptr};
  bool g_done_valid[2] = {false, false};
  int init() {
    if (s) return MMSKIN_OK;
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    const char* v = getenv("MMSKIN_SIDE_PRIORITY");   // "normal" (default) | "high" | "low": no measurable difference
    int prio = 0;
    if (v && !strcmp(v, "high")) prio = greatest;
    if (v && !strcmp(v, "low")) prio = least;
    HIP_CHECK_RET(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
    for (int i = 0; i < 3; ++i) {
      HIP_CHECK_RET(hipEventCreateWithFlags(&ready[i], hipEventDisableTiming));
      HIP_CHECK_RET(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    HIP_CHECK_RET(hipEventCreateWithFlags(&f_ready, hipEventDisableTiming));
    HIP_CHECK_RET(hipEventCreateWithFlags(&f_done, hipEventDisableTiming));
    HIP_CHECK_RET(hipEventCreateWithFlags(&f_staged, hipEventDisableTiming));
    HIP_CHECK_RET(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, prio));
    HIP_CHECK_RET(hipEventCreateWithFlags(&d_ready, hipEventDisableTiming));
    HIP_CHECK_RET(hipEventCreateWithFlags(&d_done, hipEventDisableTiming));
    HIP_CHECK_RET(hipEventCreateWithFlags(&g_ready, hipEventDisableTiming));
    HIP_CHECK_RET(hipEventCreateWithFlags(&g_done[0], hipEventDisableTiming));
    HIP_CHECK_RET(hipEventCreateWithFlags(&g_done[1], hipEventDisableTiming));
    return MMSKIN_OK;
  }
  void destroy() {
    if (!s) return;
    for (int i = 0; i < 3; ++i) { (void)hipEventDestroy(ready[i]); (void)hipEventDestroy(done[i]); }
    (void)hipEventDestroy(f_ready); (void)hipEventDestroy(f_done); (void)hipEventDestroy(f_staged);
    if (d_ready) (void)hipEventDestroy(d_ready);
    if (d_done) (void)hipEventDestroy(d_done);
    if (g_ready) { (void)hipEventDestroy(g_ready); (void)hipEventDestroy(g_done[0]); (void)hipEventDestroy(g_done[1]); }
    if (s2) (void)hipStreamDestroy(s2);
    s2 = nullptr;
    (void)hipStreamDestroy(s);
    s = nullptr;
  }
};

struct PlanBase {
  Profiler prof;
  SideStream side;
  int N = 0, H = 0, W = 0, dtype = 0;
  int feat_dim = 0;
  int out_h = 1, out_w = 1;   // spatial size of the output (1x1 = pooled features; >1 for feature-map plans)
  std::vector<TensorInfo> params, buffers;
  int64_t param_numel = 0, buffer_numel = 0;
  StageDesc* table_dev = nullptr;
  std::vector<StageDesc> table_host;
  int max_stage_elems = 0;
  size_t ws_bytes = 0;
  size_t esz() const { return dtype == 1 ? 2 : 4; }

  // Gradient segments: contiguous ranges of the flat gradient arena in the order backward COMPLETES them,
  // each with an event pair (main stream: BN gamma/beta gradients; side stream: weight gradients), so the caller can
  // start the data-parallel all-reduce of a finished range while the rest of backward is still running.
  struct GradSegment {
    int64_t offset = 0, numel = 0;
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    bool side_valid = false;
  };
  std::vector<GradSegment> segments;
  int segment_done(int i, hipStream_t st, bool use_side) {   // called by backward once segment i is fully enqueued
    GradSegment& g = segments[i];
    if (!g.ev_main) {
      HIP_CHECK_RET(hipEventCreateWithFlags(&g.ev_main, hipEventDisableTiming));
      HIP_CHECK_RET(hipEventCreateWithFlags(&g.ev_side, hipEventDisableTiming));
    }
    HIP_CHECK_RET(hipEventRecord(g.ev_main, st));
    g.side_valid = use_side;
    if (use_side) HIP_CHECK_RET(hipEventRecord(g.ev_side, side.s));
    return MMSKIN_OK;
  }
  int segment_wait(int i, hipStream_t waiter) {
    GradSegment& g = segments[i];
    if (!g.ev_main) return MMSKIN_ERR_ARG;   // no backward has run yet
    HIP_CHECK_RET(hipStreamWaitEvent(waiter, g.ev_main, 0));
    if (g.side_valid) HIP_CHECK_RET(hipStreamWaitEvent(waiter, g.ev_side, 0));
    return MMSKIN_OK;
  }

  virtual ~PlanBase() {
    if (table_dev) (void)hipFree(table_dev);
    for (GradSegment& g : segments)
      if (g.ev_main) { (void)hipEventDestroy(g.ev_main); (void)hipEventDestroy(g.ev_side); }
    side.destroy();
  }
  // image: fp32 NCHW, or (norm6 != nullptr) uint8 NHWC normalised on the fly with the 6 host floats mean rgb | std rgb
  virtual int forward(const void* image, const float* norm6, const float* params, float* buffers, unsigned char* ws,
                      float* features, bool training, hipStream_t st) = 0;
  virtual int backward(const float* dfeat, const float* params, unsigned char* ws, float* grads, hipStream_t st) = 0;
  // Grad-CAM support: the raw output of the network's last nn.Conv2d and d(features)/d(that output)
  // for an eval-mode forward that kept raw conv outputs (option "keep_raw_eval"); plans without it: unsupported
  bool keep_raw_eval = false;
  // Serving: option "reuse_staged" = the caller vouches that parameters and BatchNorm buffers are unchanged
  // since the previous folded eval forward on this workspace, so the staged (BN-folded) weights and the coefficient
  // table already in it are reused instead of rebuilt.  Plans that do not implement it simply restage.
  bool reuse_staged = false;
  const void* staged_eval_ws = nullptr;
  virtual int last_conv_shape(int*, int*, int*) const { return MMSKIN_ERR_UNSUPPORTED; }
  virtual int last_conv_export(const unsigned char*, float*, hipStream_t) { return MMSKIN_ERR_UNSUPPORTED; }
  virtual int last_conv_grad(const float*, const unsigned char*, float*, hipStream_t) { return MMSKIN_ERR_UNSUPPORTED; }
  // per-step device pointers handed in by the host (e.g. "sd_mask": stochastic-depth keep/scale factors)
  virtual int set_pointer(const char*, const void*) { return MMSKIN_ERR_UNSUPPORTED; }
  // per-unit introspection (tests / diagnostics); plans without it report zero units
  virtual int num_units() const { return 0; }
  virtual int unit_info(int, std::string*, int64_t*) const { return MMSKIN_ERR_UNSUPPORTED; }

  int ensure_table() {   // uploaded on the first forward (create() needs no GPU)
    if (table_dev) return MMSKIN_OK;
    HIP_CHECK_RET(hipMalloc((void**)&table_dev, table_host.size() * sizeof(StageDesc)));
    HIP_CHECK_RET(hipMemcpy(table_dev, table_host.data(), table_host.size() * sizeof(StageDesc),
                            hipMemcpyHostToDevice));
    return MMSKIN_OK;
  }
};

static inline int64_t add_tensor(std::vector<TensorInfo>& v, int64_t& total, const std::string& name,
                                 std::initializer_list<int64_t> shape) {
  TensorInfo t;
  t.name = name;
  t.offset = total;
  t.ndim = (int)shape.size();
  t.numel = 1;
  int i = 0;
  for (int64_t d : shape) { t.shape[i++] = d; t.numel *= d; }
  for (; i < 4; ++i) t.shape[i] = 1;
  total += t.numel;
  v.push_back(t);
  return t.offset;
}

static inline size_t carve(size_t& cursor, size_t bytes) {
  size_t o = cursor;
  cursor = align_up(cursor + bytes, 256);
  return o;
}

#define PROF(cls_, flops_, bytes_, call_)                      \
  do {                                                        \
    p.prof.begin((cls_), st);                                 \
    rc = (call_);                                             \
    p.prof.end(st);                                           \
    if (p.prof.on) { p.prof.flops[(cls_)] += (flops_); p.prof.bytes[(cls_)] += (bytes_); } \
    if (rc) return rc;                                        \
  } while (0)

static inline double conv_flops(const ConvShape& s) {
  return 2.0 * s.N * s.OH() * s.OW() * (double)s.Cout * s.Cin * s.kh * s.kw;
}
// algorithmic HBM bytes of one conv GEMM pass: input once, output once, weights once, plus `extra`
// input-shaped tensors read by a fused dgrad epilogue (addend, mask source, BN inputs)
static inline double conv_bytes(const ConvShape& s, size_t es, int extra_in_shaped = 0) {
  double in = (double)s.N * s.H * s.W * s.Cin, out = (double)s.N * s.OH() * s.OW() * s.Cout;
  return (in * (1 + extra_in_shaped) + out) * es + (double)s.Cout * s.Cin * s.kh * s.kw * es;
}

// plan factories (create() needs no GPU); return nullptr and set *rc on failure
PlanBase* make_resnet_plan(int arch, int N, int H, int W, int dtype, int* rc);
PlanBase* make_densenet_plan(int N, int H, int W, int dtype, bool feature_map, int* rc);
PlanBase* make_vgg_plan(int N, int H, int W, int dtype, int* rc);
PlanBase* make_mobilenet_plan(int N, int H, int W, int dtype, int* rc);
PlanBase* make_efficientnet_plan(int variant, int N, int H, int W, int dtype, int* rc);

// C-ABI glue: error reporting + the op-level convolution / batch-norm / stem entry points.
// These wrap exactly the kernels the backbone plan launches (conv_gemm.hip, wgrad.hip, ops.hip) with
// NCHW fp32 tensors at the boundary, so every kernel can be parity-tested in isolation.
#include <stdarg.h>
#include <string.h>

#include "../../include/mmskin.h"
#include "conv.h"
#include "ops.h"

static thread_local char g_err[1024] = "";

void mmskin_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

struct Carver {
  unsigned char* base;
  size_t cur = 0;
  explicit Carver(void* p) : base((unsigned char*)p) {}
  template <typename U> U* take(size_t count) {
    size_t o = cur;
    cur = align_up(cur + count * sizeof(U), 256);
    return reinterpret_cast<U*>(base + o);
  }
};

__global__ void coef_from_saved_kernel(int C, const float* gamma, const float* beta, const float* mean,
                                       const float* invstd, float* scale, float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float sc = gamma[c] * invstd[c];
    scale[c] = sc;
    shift[c] = beta[c] - mean[c] * sc;
  }
}

template <typename T>
int stage_one(const float* w, int Cout, int Cin, int taps, bool stem, T* wfwd, T* wdgrad, StageDesc* table_dev,
              hipStream_t st) {
  StageDesc d = {};
  d.src_off = 0; d.fwd_off = 0; d.dgrad_off = 0;
  d.Cout = Cout; d.Cin = Cin; d.taps = taps; d.stem = stem ? 1 : 0;
  HIP_CHECK_RET(hipMemcpyAsync(table_dev, &d, sizeof(d), hipMemcpyHostToDevice, st));
  HIP_CHECK_RET(hipStreamSynchronize(st));  // `d` lives on this stack frame
  if (stem) HIP_CHECK_RET(hipMemsetAsync(wfwd, 0, 64 * 256 * sizeof(T), st));
  return stage_weights<T>(table_dev, 1, Cout * Cin * taps, w, wfwd, wdgrad, wdgrad != nullptr, st);
}

template <typename T>
int conv_fwd_op(const float* x, const float* w, float* y, const ConvShape& s, void* ws, hipStream_t st) {
  Carver c(ws);
  StageDesc* table = c.take<StageDesc>(1);
  T* xh = c.take<T>((size_t)s.N * s.H * s.W * s.Cin);
  T* wf = c.take<T>((size_t)s.Cout * s.Cin * s.kh * s.kw);
  T* yh = c.take<T>((size_t)s.N * s.OH() * s.OW() * s.Cout);
  int rc;
  if ((rc = nchw_to_nhwc<T>(x, s.N, s.Cin, s.H, s.W, xh, st))) return rc;
  if ((rc = stage_one<T>(w, s.Cout, s.Cin, s.kh * s.kw, false, wf, (T*)nullptr, table, st))) return rc;
  if ((rc = launch_conv_fwd<T>(s, xh, wf, yh, nullptr, nullptr, st))) return rc;
  return nhwc_to_nchw<T>(yh, s.N, s.Cout, s.OH(), s.OW(), y, st);
}

template <typename T>
int conv_bwd_op(const float* dy, const float* x, const float* w, float* dx, float* dw, const ConvShape& s, void* ws,
                hipStream_t st) {
  Carver c(ws);
  StageDesc* table = c.take<StageDesc>(1);
  T* xh = c.take<T>((size_t)s.N * s.H * s.W * s.Cin);
  T* wf = c.take<T>((size_t)s.Cout * s.Cin * s.kh * s.kw);
  T* yh = c.take<T>((size_t)s.N * s.OH() * s.OW() * s.Cout);  // dy in NHWC
  T* wd = c.take<T>((size_t)s.Cout * s.Cin * s.kh * s.kw);
  T* dxh = c.take<T>((size_t)s.N * s.H * s.W * s.Cin);
  float* slab = c.take<float>(conv_wgrad_slab_bytes(s) / sizeof(float));
  int rc;
  if ((rc = nchw_to_nhwc<T>(dy, s.N, s.Cout, s.OH(), s.OW(), yh, st))) return rc;
  if (dx) {
    if ((rc = stage_one<T>(w, s.Cout, s.Cin, s.kh * s.kw, false, wf, wd, table, st))) return rc;
    if ((rc = launch_conv_dgrad<T>(s, yh, wd, dxh, (const T*)nullptr, st))) return rc;
    if ((rc = nhwc_to_nchw<T>(dxh, s.N, s.Cin, s.H, s.W, dx, st))) return rc;
  }
  if (dw) {
    if ((rc = nchw_to_nhwc<T>(x, s.N, s.Cin, s.H, s.W, xh, st))) return rc;
    if ((rc = launch_conv_wgrad<T>(s, yh, xh, slab, dw, st))) return rc;
  }
  return MMSKIN_OK;
}

template <typename T>
int bn_fwd_op(const float* x, const float* gamma, const float* beta, float* rm, float* rv, float* y, float* save_mean,
              float* save_invstd, int N, int C, int H, int W, float eps, float mom, int relu, void* ws, hipStream_t st) {
  Carver c(ws);
  const size_t rows = (size_t)N * H * W;
  T* xh = c.take<T>(rows * C);
  T* yh = c.take<T>(rows * C);
  float* ssum = c.take<float>((size_t)column_stats_rows(rows, C) * C);
  float* ssq = c.take<float>((size_t)column_stats_rows(rows, C) * C);
  float* coef = c.take<float>(2 * (size_t)C);
  int rc, nr = 0;
  if ((rc = nchw_to_nhwc<T>(x, N, C, H, W, xh, st))) return rc;
  if ((rc = column_stats<T>(xh, rows, C, ssum, ssq, &nr, st))) return rc;
  if ((rc = bn_finalize(ssum, ssq, nr, C, (double)rows, gamma, beta, eps, mom, rm, rv, coef, coef + C, save_mean,
                        save_invstd, nullptr, st))) return rc;
  if ((rc = bn_apply<T>(xh, nullptr, coef, coef + C, nullptr, nullptr, yh, rows, C, relu != 0, st))) return rc;
  return nhwc_to_nchw<T>(yh, N, C, H, W, y, st);
}

template <typename T>
int bn_bwd_op(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
              const float* save_invstd, float* dx, float* dgamma, float* dbeta, int N, int C, int H, int W, int relu,
              void* ws, hipStream_t st) {
  Carver c(ws);
  const size_t rows = (size_t)N * H * W;
  T* xh = c.take<T>(rows * C);
  T* dyh = c.take<T>(rows * C);
  float* partial = c.take<float>((size_t)bn_bwd_partial_rows(rows, C) * 2 * C);
  float* tail = c.take<float>((size_t)bn_bwd_partial_rows(rows, C) * C);  // keep layout equal to bn_fwd_op's
  (void)tail;
  float* coef = c.take<float>(5 * (size_t)C);
  T* dxh = c.take<T>(rows * C);
  int rc;
  if ((rc = nchw_to_nhwc<T>(x, N, C, H, W, xh, st))) return rc;
  if ((rc = nchw_to_nhwc<T>(dy, N, C, H, W, dyh, st))) return rc;
  hipLaunchKernelGGL(coef_from_saved_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, C, gamma, beta, save_mean,
                     save_invstd, coef, coef + C);
  HIP_CHECK_RET(hipGetLastError());
  const int mode = relu ? MASK_FROM_X : MASK_NONE;
  int nr = 0;
  if ((rc = bn_bwd_reduce<T>(dyh, xh, nullptr, coef, coef + C, mode, rows, C, partial, &nr, st))) return rc;
  if ((rc = bn_bwd_finalize(partial, nr, C, (double)rows, gamma, save_mean, save_invstd,
                            dgamma, dbeta, coef + 2 * C, coef + 3 * C, coef + 4 * C, nullptr, st))) return rc;
  if ((rc = bn_bwd_apply<T>(dyh, xh, nullptr, coef, coef + C, mode, coef + 2 * C, coef + 3 * C, coef + 4 * C, dxh,
                            (T*)nullptr, rows, C, st))) return rc;
  return nhwc_to_nchw<T>(dxh, N, C, H, W, dx, st);
}

struct StemGeom {
  int OH, OW, Hp, Wp, PH, PW;
  StemGeom(int H, int W) {
    OH = (H + 6 - 7) / 2 + 1; OW = (W + 6 - 7) / 2 + 1;
    Hp = 2 * OH + 8; Wp = 2 * OW + 8;
    if (Hp < H + 6) Hp = H + 6;
    if (Wp < W + 6) Wp = W + 6;
    Wp = (Wp + 1) / 2 * 2;
    PH = (OH + 2 - 3) / 2 + 1; PW = (OW + 2 - 3) / 2 + 1;
  }
};

template <typename T>
struct StemWs {
  StageDesc* table; T* img4; T* wv; T* x0; T* pool; uint8_t* idx; float* ssum; float* ssq; float* coef;
  T* dpool; T* dyfull; T* dx0; float* partial; float* slab; float* dwv; double* red;
  StemWs(void* ws, int N, int H, int W) {
    StemGeom g(H, W);
    Carver c(ws);
    const size_t rows = (size_t)N * g.OH * g.OW;
    table = c.take<StageDesc>(1);
    img4 = c.take<T>((size_t)N * g.Hp * g.Wp * 4);
    wv = c.take<T>(64 * 256);
    x0 = c.take<T>(rows * 64);
    pool = c.take<T>((size_t)N * g.PH * g.PW * 64);
    idx = c.take<uint8_t>((size_t)N * g.PH * g.PW * 64);
    ssum = c.take<float>((size_t)stem_conv_stat_rows(N, g.OH, g.OW) * 64);
    ssq = c.take<float>((size_t)stem_conv_stat_rows(N, g.OH, g.OW) * 64);
    coef = c.take<float>(7 * 64);
    dpool = c.take<T>((size_t)N * g.PH * g.PW * 64);
    dyfull = c.take<T>(rows * 64);
    dx0 = c.take<T>(rows * 64);
    partial = c.take<float>((size_t)bn_bwd_partial_rows(rows, 64) * 2 * 64);
    slab = c.take<float>(stem_wgrad_slab_bytes(N, g.OH, g.OW) / sizeof(float));
    dwv = c.take<float>(64 * 256);
    red = c.take<double>(2 * 64 * 64);
    total = c.cur;
  }
  size_t total;
};

template <typename T>
int stem_fwd_core(StemWs<T>& s, const float* x, const float* w, const float* gamma, const float* beta, int N, int H,
                  int W, float eps, hipStream_t st) {
  StemGeom g(H, W);
  int rc;
  if ((rc = stem_pack<T>(x, N, H, W, g.Hp, g.Wp, s.img4, st))) return rc;
  if ((rc = stage_one<T>(w, 64, 3, 49, true, s.wv, (T*)nullptr, s.table, st))) return rc;
  int stem_rows = 0;
  if ((rc = launch_stem_conv_fwd<T>(N, g.OH, g.OW, g.Hp, g.Wp, s.img4, s.wv, s.x0, s.ssum, s.ssq, st, &stem_rows))) return rc;
  if ((rc = bn_finalize(s.ssum, s.ssq, stem_rows, 64, (double)N * g.OH * g.OW, gamma, beta, eps,
                        0.1f, nullptr, nullptr, s.coef, s.coef + 64, s.coef + 128, s.coef + 192, s.red, st))) return rc;
  return stem_bn_relu_pool<T>(s.x0, s.coef, s.coef + 64, N, g.OH, g.OW, 64, s.pool, s.idx, st);
}

template <typename T>
int stem_fwd_op(const float* x, const float* w, const float* gamma, const float* beta, float* y, int N, int H,
                       int W, float eps, void* ws, hipStream_t st) {
  StemWs<T> s(ws, N, H, W);
  StemGeom g(H, W);
  int rc;
  if ((rc = stem_fwd_core<T>(s, x, w, gamma, beta, N, H, W, eps, st))) return rc;
  return nhwc_to_nchw<T>(s.pool, N, 64, g.PH, g.PW, y, st);
}

template <typename T>
int stem_bwd_op(const float* dy, const float* x, const float* w, const float* gamma, const float* beta,
                       float* dw, float* dgamma, float* dbeta, int N, int H, int W, float eps, void* ws, hipStream_t st) {
  StemWs<T> s(ws, N, H, W);
  StemGeom g(H, W);
  const size_t rows = (size_t)N * g.OH * g.OW;
  int rc;
  if ((rc = stem_fwd_core<T>(s, x, w, gamma, beta, N, H, W, eps, st))) return rc;
  if ((rc = nchw_to_nhwc<T>(dy, N, 64, g.PH, g.PW, s.dpool, st))) return rc;
  // the plans' fused path: max-pool + ReLU + BatchNorm backward on pooled cells, no full-resolution pooled gradient
  int nr = 0;
  static const bool pooled_sums = [] { const char* v = getenv("MMSKIN_STEM_SUMS_POOLED"); return !v || atoi(v) != 0; }();   // as the ResNet plan
  if (pooled_sums) rc = stem_pool_bwd_sums<T>(s.dpool, s.pool, s.idx, s.x0, s.coef, s.coef + 64, N, g.OH, g.OW, 64, s.partial, &nr, st);
  else rc = stem_pool_bn_bwd_reduce<T>(s.dpool, s.idx, s.x0, s.coef, s.coef + 64, N, g.OH, g.OW, 64, s.partial, &nr, st);
  if (rc) return rc;
  if ((rc = bn_bwd_finalize(s.partial, nr, 64, (double)rows, gamma, s.coef + 128, s.coef + 192,
                            dgamma, dbeta, s.coef + 256, s.coef + 320, s.coef + 384, s.red, st))) return rc;
  if ((rc = stem_pool_bn_bwd_apply<T>(s.dpool, s.idx, s.x0, s.coef, s.coef + 64, s.coef + 256, s.coef + 320, s.coef + 384, N, g.OH,
                                      g.OW, 64, s.dx0, st))) return rc;
  if ((rc = launch_stem_conv_wgrad<T>(N, g.OH, g.OW, g.Hp, g.Wp, s.dx0, s.img4, s.slab, s.dwv, st))) return rc;
  return stem_wgrad_unpack(s.dwv, dw, st);
}

}  // namespace

extern "C" {

const char* mmskin_last_error(void) { return g_err; }
int mmskin_version(void) { return 100; }

int64_t mmskin_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad) {
  ConvShape s = {N, H, W, Cin, Cout, kh, kw, stride, pad};
  size_t in = (size_t)N * H * W * Cin * 4, out = (size_t)N * s.OH() * s.OW() * Cout * 4;
  size_t wgt = (size_t)Cout * Cin * kh * kw * 4;
  size_t slab = (Cout % 64 == 0 && (Cin * kh * kw) % 64 == 0) ? conv_wgrad_slab_bytes(s) : 0;
  return (int64_t)(2 * in + out + 2 * wgt + slab + 16 * 256);
}

#define DISPATCH(dtype, call_f32, call_bf16)                            \
  do {                                                                  \
    if ((dtype) == MMSKIN_F32) return call_f32;                         \
    if ((dtype) == MMSKIN_BF16) return call_bf16;                       \
    mmskin_set_error("unknown dtype %d", (dtype));                      \
    return MMSKIN_ERR_ARG;                                              \
  } while (0)

int mmskin_conv2d_forward(const float* x, const float* w, float* y, int N, int Cin, int H, int W, int Cout, int kh,
                          int kw, int stride, int pad, int dtype, void* workspace, void* stream) {
  ARG_CHECK(x && w && y && workspace, "conv2d_forward: null argument");
  ConvShape s = {N, H, W, Cin, Cout, kh, kw, stride, pad};
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype, conv_fwd_op<float>(x, w, y, s, workspace, st), conv_fwd_op<bf16_t>(x, w, y, s, workspace, st));
}

int mmskin_conv2d_backward(const float* dy, const float* x, const float* w, float* dx, float* dw, int N, int Cin,
                           int H, int W, int Cout, int kh, int kw, int stride, int pad, int dtype, void* workspace,
                           void* stream) {
  ARG_CHECK(dy && x && w && workspace, "conv2d_backward: null argument");
  ConvShape s = {N, H, W, Cin, Cout, kh, kw, stride, pad};
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype, conv_bwd_op<float>(dy, x, w, dx, dw, s, workspace, st),
           conv_bwd_op<bf16_t>(dy, x, w, dx, dw, s, workspace, st));
}

/* Data gradient with the CONSUMER's BatchNorm-backward prologue fused into the epilogue (DgradFuse, profile 3: what the plan launches
 * for every conv -> BatchNorm -> ReLU unit): dz = dgrad(dy) * (xc * scale + shift > 0), plus the partial sums of dz and dz * xc per
 * row block.  xc [N,Cin,H,W] is the raw BatchNorm input of the unit that produced this conv's input; dz [N,Cin,H,W];
 * partial [rows][2][Cin] (rows <= mmskin_conv2d_dgrad_fused_rows(...)), *rows_written = rows the launch produced. */
int mmskin_conv2d_dgrad_fused_rows(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad) {
  ConvShape s = {N, H, W, Cin, Cout, kh, kw, stride, pad};
  const int a = conv_dgrad_partial_rows(s);
  return a > N ? a : N;
}
int mmskin_conv2d_dgrad_fused(const float* dy, const float* w, const float* xc, const float* scale, const float* shift, float* dz,
                              float* partial, int* rows_written, int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride,
                              int pad, void* workspace, void* stream) {
  ARG_CHECK(dy && w && xc && scale && shift && dz && partial && rows_written && workspace, "conv2d_dgrad_fused: null argument");
  ConvShape s = {N, H, W, Cin, Cout, kh, kw, stride, pad};
  hipStream_t st = (hipStream_t)stream;
  Carver c(workspace);
  typedef bf16_t T;
  StageDesc* table = c.take<StageDesc>(1);
  T* xh = c.take<T>((size_t)N * H * W * Cin);
  T* wf = c.take<T>((size_t)Cout * Cin * kh * kw);
  T* yh = c.take<T>((size_t)N * s.OH() * s.OW() * Cout);
  T* wd = c.take<T>((size_t)Cout * Cin * kh * kw);
  T* dxh = c.take<T>((size_t)N * H * W * Cin);
  int rc;
  if ((rc = nchw_to_nhwc<T>(dy, N, Cout, s.OH(), s.OW(), yh, st))) return rc;
  if ((rc = nchw_to_nhwc<T>(xc, N, Cin, H, W, xh, st))) return rc;
  if ((rc = stage_one<T>(w, Cout, Cin, kh * kw, false, wf, wd, table, st))) return rc;
  DgradFuse f;
  f.x = xh; f.scale = scale; f.shift = shift; f.partial = partial;
  if ((rc = launch_conv_dgrad<T>(s, yh, wd, dxh, (const T*)nullptr, st, &f))) return rc;
  *rows_written = f.rows_written;
  return nhwc_to_nchw<T>(dxh, N, Cin, H, W, dz, st);
}

/* timing helper: runs the forward conv kernel `iters` times on NHWC buffers already resident in the
 * workspace (contents irrelevant) and returns the average microseconds per launch. */
double mmskin_conv2d_time(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dtype,
                          int iters, void* workspace, void* stream) {
  ConvShape s = {N, H, W, Cin, Cout, kh, kw, stride, pad};
  hipStream_t st = (hipStream_t)stream;
  Carver c(workspace);
  size_t es = dtype == MMSKIN_BF16 ? 2 : 4;
  unsigned char* xh = c.take<unsigned char>((size_t)N * H * W * Cin * es);
  unsigned char* wf = c.take<unsigned char>((size_t)Cout * Cin * kh * kw * es);
  unsigned char* yh = c.take<unsigned char>((size_t)N * s.OH() * s.OW() * Cout * es);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  auto run = [&]() -> int {
    if (dtype == MMSKIN_BF16) return launch_conv_fwd<bf16_t>(s, (bf16_t*)xh, (bf16_t*)wf, (bf16_t*)yh, nullptr, nullptr, st);
    return launch_conv_fwd<float>(s, (float*)xh, (float*)wf, (float*)yh, nullptr, nullptr, st);
  };
  for (int i = 0; i < 3; ++i) if (run()) return -1.0;
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) if (run()) return -1.0;
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return (double)ms * 1e3 / iters;
}

/* same for the data-gradient launch (no fused epilogue; stride-2 layers run as parity classes) */
double mmskin_conv2d_dgrad_time(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dtype,
                                int iters, void* workspace, void* stream) {
  ConvShape s = {N, H, W, Cin, Cout, kh, kw, stride, pad};
  hipStream_t st = (hipStream_t)stream;
  Carver c(workspace);
  size_t es = dtype == MMSKIN_BF16 ? 2 : 4;
  unsigned char* xh = c.take<unsigned char>((size_t)N * H * W * Cin * es);
  unsigned char* wt = c.take<unsigned char>((size_t)Cout * Cin * kh * kw * es);
  unsigned char* yh = c.take<unsigned char>((size_t)N * s.OH() * s.OW() * Cout * es);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  auto run = [&]() -> int {
    if (dtype == MMSKIN_BF16) return launch_conv_dgrad<bf16_t>(s, (bf16_t*)yh, (bf16_t*)wt, (bf16_t*)xh, (const bf16_t*)nullptr, st);
    return launch_conv_dgrad<float>(s, (float*)yh, (float*)wt, (float*)xh, (const float*)nullptr, st);
  };
  for (int i = 0; i < 3; ++i) if (run()) return -1.0;
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) if (run()) return -1.0;
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return (double)ms * 1e3 / iters;
}

/* Algebraic BatchNorm backward of an expanding 1x1 convolution (abn.hip), op level: g [N,C4,H,W] is the (masked) gradient of the
 * BatchNorm output, y [N,Cw,H,W] the convolution's input, w [C4,Cw] its weight, cA / cB / cC [C4] the BatchNorm-backward coefficients
 * (dz = cA g + cB x + cC, x = conv(y)).  Returns dy = dz W [N,Cw,H,W] and dw = dz^T y [C4,Cw] without forming dz or x. */
int64_t mmskin_abn_workspace_bytes(int N, int Cw, int C4, int H, int W) {
  const size_t M = (size_t)N * H * W;
  size_t b = 4096 + M * C4 * 2 + 2 * M * Cw * 2 + (size_t)Cw * (C4 + Cw) * 2 + 4 * (size_t)Cw + 12 * (size_t)C4;
  b += 2 * wgrad_gram_slab_bytes((int)M, C4, Cw) + wgrad_gram_slab_bytes((int)M, 0, Cw, 2) + ((size_t)C4 + 512) * Cw * 4 + 32 * 256;
  return (int64_t)b;
}
static int abn_backward_impl(const float* g, const float* y, const float* w, const float* cA, const float* cB, const float* cC, int N, int Cw,
                             int C4, int H, int W, float* dy, float* dw, void* workspace, hipStream_t st, bool kept_gram) {
  Carver c(workspace);
  const size_t M = (size_t)N * H * W;
  ARG_CHECK(wgrad_gram_slab_bytes((int)M, C4, Cw) > 0 && (!kept_gram || (wgrad_gram_slab_bytes((int)M, C4, Cw, 1) > 0 && wgrad_gram_slab_bytes((int)M, 0, Cw, 2) > 0)),
            "abn_backward: shape C4=%d Cw=%d unsupported", C4, Cw);
  bf16_t* gh = c.take<bf16_t>(M * C4);
  bf16_t* yh = c.take<bf16_t>(M * Cw);
  bf16_t* dyh = c.take<bf16_t>(M * Cw);
  bf16_t* wd = c.take<bf16_t>((size_t)Cw * (C4 + Cw));
  float* bias = c.take<float>(Cw);
  float* coef = c.take<float>(3 * (size_t)C4);
  float* S = c.take<float>(((size_t)C4 + 256) * Cw);
  float* cs = c.take<float>(256);
  float* gram = c.take<float>((size_t)256 * Cw);
  size_t sb = wgrad_gram_slab_bytes((int)M, C4, Cw);
  if (wgrad_gram_slab_bytes((int)M, C4, Cw, 1) > sb) sb = wgrad_gram_slab_bytes((int)M, C4, Cw, 1);
  if (wgrad_gram_slab_bytes((int)M, 0, Cw, 2) > sb) sb = wgrad_gram_slab_bytes((int)M, 0, Cw, 2);
  float* slab = c.take<float>(sb / sizeof(float));
  int rc;
  if ((rc = nchw_to_nhwc<bf16_t>(g, N, C4, H, W, gh, st))) return rc;
  if ((rc = nchw_to_nhwc<bf16_t>(y, N, Cw, H, W, yh, st))) return rc;
  if ((rc = abn_prep(w, cA, cB, cC, C4, Cw, wd, bias, coef, st))) return rc;
  ConvShape s = {N, H, W, Cw, C4, 1, 1, 1, 0};
  DgradFuse f;
  f.in2 = yh; f.k2 = Cw; f.bias = bias;
  if ((rc = launch_conv_dgrad<bf16_t>(s, gh, wd, dyh, (const bf16_t*)nullptr, st, &f))) return rc;
  if ((rc = nhwc_to_nchw<bf16_t>(dyh, N, Cw, H, W, dy, st))) return rc;
  if (kept_gram) {   // the two-pass forward's order: y^T y + colsum(y) first (forward), g^T y alone later
    if ((rc = launch_wgrad_gram(N, H, W, Cw, C4, nullptr, yh, slab, gram, cs, st, 2))) return rc;
    if ((rc = launch_wgrad_gram(N, H, W, Cw, C4, gh, yh, slab, S, nullptr, st, 1))) return rc;
    return abn_wgrad_finalize(S, cs, w, coef, C4, Cw, dw, st, gram);
  }
  if ((rc = launch_wgrad_gram(N, H, W, Cw, C4, gh, yh, slab, S, cs, st))) return rc;
  return abn_wgrad_finalize(S, cs, w, coef, C4, Cw, dw, st);
}
int mmskin_abn_backward(const float* g, const float* y, const float* w, const float* cA, const float* cB, const float* cC, int N, int Cw,
                        int C4, int H, int W, float* dy, float* dw, void* workspace, void* stream) {
  return abn_backward_impl(g, y, w, cA, cB, cC, N, Cw, C4, H, W, dy, dw, workspace, (hipStream_t)stream, false);
}
/* same result through the two-pass forward's kernels: Gram matrix + column sums of y in their own launch, g^T y alone */
int mmskin_abn_backward_kept_gram(const float* g, const float* y, const float* w, const float* cA, const float* cB, const float* cC, int N,
                                  int Cw, int C4, int H, int W, float* dy, float* dw, void* workspace, void* stream) {
  return abn_backward_impl(g, y, w, cA, cB, cC, N, Cw, C4, H, W, dy, dw, workspace, (hipStream_t)stream, true);
}
/* Per-channel sum and sum of squares of x = conv1x1(y, w) (y [N,Cw,H,W], w [C4,Cw], bf16 operands) WITHOUT forming x: from the Gram
 * matrix y^T y and the column sums of y (abn.hip gram_stats) -- the first pass of the two-pass BatchNorm forward.  Workspace as
 * mmskin_abn_workspace_bytes. */
int mmskin_conv1x1_gram_stats(const float* y, const float* w, int N, int Cw, int C4, int H, int W, float* stat_sum, float* stat_sq,
                              void* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  Carver c(workspace);
  const size_t M = (size_t)N * H * W;
  ARG_CHECK(wgrad_gram_slab_bytes((int)M, 0, Cw, 2) > 0, "conv1x1_gram_stats: Cw=%d unsupported", Cw);
  bf16_t* yh = c.take<bf16_t>(M * Cw);
  float* gram = c.take<float>((size_t)256 * Cw);
  float* cs = c.take<float>(256);
  float* slab = c.take<float>(wgrad_gram_slab_bytes((int)M, 0, Cw, 2) / sizeof(float));
  int rc;
  if ((rc = nchw_to_nhwc<bf16_t>(y, N, Cw, H, W, yh, st))) return rc;
  if ((rc = launch_wgrad_gram(N, H, W, Cw, C4, nullptr, yh, slab, gram, cs, st, 2))) return rc;
  return gram_stats(gram, cs, w, C4, Cw, stat_sum, stat_sq, st);
}

/* same for the weight-gradient kernel (+ its slab reduction) */
double mmskin_conv2d_wgrad_time(int N, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dtype,
                                int iters, void* workspace, void* stream) {
  ConvShape s = {N, H, W, Cin, Cout, kh, kw, stride, pad};
  hipStream_t st = (hipStream_t)stream;
  Carver c(workspace);
  size_t es = dtype == MMSKIN_BF16 ? 2 : 4;
  unsigned char* xh = c.take<unsigned char>((size_t)N * H * W * Cin * es);
  unsigned char* yh = c.take<unsigned char>((size_t)N * s.OH() * s.OW() * Cout * es);
  float* dw = c.take<float>((size_t)Cout * Cin * kh * kw);
  float* slab = c.take<float>(conv_wgrad_slab_bytes(s) / sizeof(float));
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.0;
  auto run = [&]() -> int {
    if (dtype == MMSKIN_BF16) return launch_conv_wgrad<bf16_t>(s, (bf16_t*)yh, (bf16_t*)xh, slab, dw, st);
    return launch_conv_wgrad<float>(s, (float*)yh, (float*)xh, slab, dw, st);
  };
  for (int i = 0; i < 3; ++i) if (run()) return -1.0;
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) if (run()) return -1.0;
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return (double)ms * 1e3 / iters;
}

int64_t mmskin_batchnorm_workspace_bytes(int N, int C, int H, int W) {
  size_t rows = (size_t)N * H * W;
  return (int64_t)(3 * rows * C * 4 + (size_t)bn_bwd_partial_rows(rows, C) * 3 * C * 4 + 8 * (size_t)C * 4 + 16 * 256);
}

int mmskin_batchnorm_forward(const float* x, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float* y, float* save_mean, float* save_invstd, int N, int C, int H,
                             int W, float eps, float momentum, int relu, int dtype, void* workspace, void* stream) {
  ARG_CHECK(x && gamma && beta && y && save_mean && save_invstd && workspace, "batchnorm_forward: null argument");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           bn_fwd_op<float>(x, gamma, beta, running_mean, running_var, y, save_mean, save_invstd, N, C, H, W, eps, momentum, relu, workspace, st),
           bn_fwd_op<bf16_t>(x, gamma, beta, running_mean, running_var, y, save_mean, save_invstd, N, C, H, W, eps, momentum, relu, workspace, st));
}

int mmskin_batchnorm_backward(const float* dy, const float* x, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta,
                              int N, int C, int H, int W, int relu, int dtype, void* workspace, void* stream) {
  ARG_CHECK(dy && x && gamma && beta && save_mean && save_invstd && dx && workspace, "batchnorm_backward: null argument");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype,
           bn_bwd_op<float>(dy, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, N, C, H, W, relu, workspace, st),
           bn_bwd_op<bf16_t>(dy, x, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, N, C, H, W, relu, workspace, st));
}

int64_t mmskin_stem_workspace_bytes(int N, int H, int W) {
  StemWs<float> s(nullptr, N, H, W);
  return (int64_t)s.total + 4096;
}

int mmskin_stem_forward(const float* x, const float* w, const float* gamma, const float* beta, float* y, int N, int H,
                        int W, float eps, int dtype, void* workspace, void* stream) {
  ARG_CHECK(x && w && gamma && beta && y && workspace, "stem_forward: null argument");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype, stem_fwd_op<float>(x, w, gamma, beta, y, N, H, W, eps, workspace, st),
           stem_fwd_op<bf16_t>(x, w, gamma, beta, y, N, H, W, eps, workspace, st));
}

int mmskin_stem_backward(const float* dy, const float* x, const float* w, const float* gamma, const float* beta,
                         float* dw, float* dgamma, float* dbeta, int N, int H, int W, float eps, int dtype,
                         void* workspace, void* stream) {
  ARG_CHECK(dy && x && w && gamma && beta && dw && workspace, "stem_backward: null argument");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH(dtype, stem_bwd_op<float>(dy, x, w, gamma, beta, dw, dgamma, dbeta, N, H, W, eps, workspace, st),
           stem_bwd_op<bf16_t>(dy, x, w, gamma, beta, dw, dgamma, dbeta, N, H, W, eps, workspace, st));
}

/* fp32 NHWC depthwise 3x3 (stride 1, pad 1) for token-layout models (DaViT's convolutional position encoding):
 * w is the nn.Conv2d(groups = C) weight [C][1][3][3]; w_stage holds 9*C floats; backward also needs
 * mmskin_dwconv3_scratch_floats(...) floats of scratch. */
int64_t mmskin_dwconv3_scratch_floats(int N, int H, int W, int C) { return (int64_t)dwconv3_wgrad_partial_floats(N, H, W, C, 1, 3); }
int mmskin_dwconv3_forward(const float* x, const float* w, float* w_stage, float* y, int N, int H, int W, int C, void* stream) {
  ARG_CHECK(x && w && w_stage && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "dwconv3_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int rc = dw_stage_weights<float>(w, C, C, w_stage, st, 3);
  if (rc) return rc;
  return dwconv3_fwd<float>(x, w_stage, N, H, W, C, 1, y, st, 3);
}
int mmskin_dwconv3_backward(const float* dy, const float* x, const float* w, float* w_stage, float* scratch, float* dx, float* dw,
                            int N, int H, int W, int C, void* stream) {
  ARG_CHECK(dy && x && w && w_stage && scratch && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "dwconv3_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int rc = dw_stage_weights<float>(w, C, C, w_stage, st, 3);
  if (rc) return rc;
  if (dx && (rc = dwconv3_dgrad<float>(dy, w_stage, N, H, W, C, 1, dx, st, 3))) return rc;
  if (dw && (rc = dwconv3_wgrad<float>(dy, x, N, H, W, C, 1, scratch, dw, C, st, 3))) return rc;
  return MMSKIN_OK;
}

/* DaViT's convolutional position encoding in one pass each way (timm davit.py ConvPosEnc.forward: x + proj(x), proj = depthwise 3x3 with
 * bias; loadImageModelClassifier.py:117-131):  y = x + dwconv3(x, w) + b.  Backward: dx = dy + dgrad(dy), dw, and db = sum(dy) from the
 * weight-gradient pass.  Same staging / scratch as mmskin_dwconv3_*. */
int mmskin_conv_pos_enc_forward(const float* x, const float* w, const float* b, float* w_stage, float* y, int N, int H, int W, int C,
                                void* stream) {
  ARG_CHECK(x && w && w_stage && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "conv_pos_enc_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int rc = dw_stage_weights<float>(w, C, C, w_stage, st, 3);
  if (rc) return rc;
  return dwconv3_fwd<float>(x, w_stage, N, H, W, C, 1, y, st, 3, b, true);
}
int mmskin_conv_pos_enc_backward(const float* dy, const float* x, const float* w, float* w_stage, float* scratch, float* dx, float* dw,
                                 float* db, int N, int H, int W, int C, void* stream) {
  ARG_CHECK(dy && x && w && w_stage && scratch && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "conv_pos_enc_backward: bad argument");
  ARG_CHECK(dw || !db, "conv_pos_enc_backward: db is produced by the weight-gradient pass (dw required)");
  hipStream_t st = (hipStream_t)stream;
  int rc = dw_stage_weights<float>(w, C, C, w_stage, st, 3);
  if (rc) return rc;
  if (dx && (rc = dwconv3_dgrad<float>(dy, w_stage, N, H, W, C, 1, dx, st, 3, true))) return rc;
  if (dw && (rc = dwconv3_wgrad<float>(dy, x, N, H, W, C, 1, scratch, dw, C, st, 3, db))) return rc;
  return MMSKIN_OK;
}

}  // extern "C"

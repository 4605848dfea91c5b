// HBM-bound NHWC kernels (see ops.h).  Replaces the ATen batch_norm / relu / add / max_pool2d /
// adaptive_avg_pool2d forward+backward kernels the reference reaches through torchvision's ResNet
// (multimodalIntraInterModal.py:167).
#include <stdlib.h>

#include "ops.h"

#define EW_BLOCK 256
static inline int ew_grid(size_t work_items) {
  size_t b = (work_items + EW_BLOCK - 1) / EW_BLOCK;
  // One 16-byte chunk per thread, no grid-stride cap in practice: on MI355X a 2-reads-1-write pass over 1.2 GB ran at
  // 4.9 TB/s with 4 096 workgroups and 5.9 TB/s with 65 536 (scripts/bench/membench.hip) -- many short workgroups
  // keep more loads in flight than few long-running ones.
  if (b > (size_t)1 << 20) b = (size_t)1 << 20;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------ column reduction geometry
struct ColGeom {
  int CPR;   // 16-byte chunks per row
  int CW;    // chunk columns per block
  int RL;    // row lanes per block
  int RB;    // rows per block
  int gx, gy;
};
static ColGeom col_geom(size_t rows, int C, int EPC) {
  ColGeom g;
  g.CPR = C / EPC;
  g.CW = g.CPR >= 256 ? 256 : g.CPR;
  g.RL = 256 / g.CW;
  size_t rb = (rows + 1023) / 1024;
  if (rb < (size_t)g.RL * 4) rb = (size_t)g.RL * 4;
  rb = (rb + g.RL - 1) / g.RL * g.RL;
  g.RB = (int)rb;
  g.gx = (int)((rows + rb - 1) / rb);
  g.gy = (g.CPR + g.CW - 1) / g.CW;
  return g;
}

// Reduce NQ per-thread EPC-wide accumulators over the row lanes of a block and write them to
// partial[(blockIdx.x*NQ + q)*C + channel].
template <int EPC, int NQ>
__device__ __forceinline__ void block_col_reduce(float (&acc)[NQ][EPC], int cx, int ry, int CW, int RL,
                                                 int col, int CPR, int C, float* partial, float* red) {
  // red: [NQ][RL][CW*EPC]
  const bool active = ry < RL;
  if (active) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < EPC; ++e) red[(q * RL + ry) * CW * EPC + cx * EPC + e] = acc[q][e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NQ * CW * EPC; i += blockDim.x) {
    int q = i / (CW * EPC), ce = i - q * CW * EPC;
    int ch = blockIdx.y * CW * EPC + ce;
    if (ch < C) {
      float s = 0.f;
      for (int r = 0; r < RL; ++r) s += red[(q * RL + r) * CW * EPC + ce];
      partial[((size_t)blockIdx.x * NQ + q) * C + ch] = s;
    }
  }
}

// ------------------------------------------------------------------ row-partial pre-reduction
// in[nrows][cols] -> out[G][cols]: group g sums rows [g*per, (g+1)*per).  Conv epilogues / wgrad splits
// leave up to ~25k partial rows; reducing them in one finalize block per 64 channels was latency-bound
// (200 us per BatchNorm), so a wide first stage brings the row count down to <= 64 first.
template <typename OUT>
__global__ __launch_bounds__(512) void partial_reduce_kernel(const float* __restrict__ in0,
                                                             const float* __restrict__ in1, int nrows, int cols,
                                                             int G, OUT* __restrict__ out) {
  __shared__ OUT red[8][64];
  const float* in = blockIdx.z ? in1 : in0;
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int per = (nrows + G - 1) / G;
  const int r0 = blockIdx.y * per;
  const int r1 = min(nrows, r0 + per);
  OUT a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (c < cols) {
    int r = r0 + ry;
    for (; r + 24 < r1; r += 32) {
      a0 += (OUT)in[(size_t)r * cols + c];
      a1 += (OUT)in[(size_t)(r + 8) * cols + c];
      a2 += (OUT)in[(size_t)(r + 16) * cols + c];
      a3 += (OUT)in[(size_t)(r + 24) * cols + c];
    }
    for (; r < r1; r += 8) a0 += (OUT)in[(size_t)r * cols + c];
  }
  red[ry][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ry == 0 && c < cols) {
    OUT s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += red[i][cx];
    out[((size_t)blockIdx.z * G + blockIdx.y) * cols + c] = s;
  }
}

// float4 variant for wide matrices (wgrad split slabs): 64 threads cover 256 columns, 8 row lanes
__global__ __launch_bounds__(512) void partial_reduce4_kernel(const float* __restrict__ in, int nrows, int cols,
                                                              int G, float* __restrict__ out) {
  __shared__ float4 red[8][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cx) * 4;
  const int per = (nrows + G - 1) / G;
  const int r0 = blockIdx.y * per;
  const int r1 = min(nrows, r0 + per);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  if (c < cols) {
    int r = r0 + ry;
    for (; r + 8 < r1; r += 16) {
      float4 u = *reinterpret_cast<const float4*>(in + (size_t)r * cols + c);
      float4 v = *reinterpret_cast<const float4*>(in + (size_t)(r + 8) * cols + c);
      a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
      a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
    }
    for (; r < r1; r += 8) {
      float4 u = *reinterpret_cast<const float4*>(in + (size_t)r * cols + c);
      a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
    }
  }
  red[ry][cx] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  __syncthreads();
  if (ry == 0 && c < cols) {
    float4 s = red[0][cx];
#pragma unroll
    for (int i = 1; i < 8; ++i) { s.x += red[i][cx].x; s.y += red[i][cx].y; s.z += red[i][cx].z; s.w += red[i][cx].w; }
    *reinterpret_cast<float4*>(out + (size_t)blockIdx.y * cols + c) = s;
  }
}

template <typename OUT>
int partial_reduce(const float* in0, const float* in1, int nrows, int cols, int G, OUT* out, hipStream_t st) {
  if constexpr (sizeof(OUT) == 4) {
    if (!in1 && cols % 4 == 0) {
      hipLaunchKernelGGL(partial_reduce4_kernel, dim3(ceil_div(cols, 256), G), dim3(512), 0, st, in0, nrows, cols, G,
                         reinterpret_cast<float*>(out));
      HIP_CHECK_RET(hipGetLastError());
      return MMSKIN_OK;
    }
  }
  hipLaunchKernelGGL(partial_reduce_kernel<OUT>, dim3(ceil_div(cols, 64), G, in1 ? 2 : 1), dim3(512), 0, st, in0, in1,
                     nrows, cols, G, out);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template int partial_reduce<float>(const float*, const float*, int, int, int, float*, hipStream_t);
template int partial_reduce<double>(const float*, const float*, int, int, int, double*, hipStream_t);

static inline int reduce_groups(int nrows) {
  int g = (nrows + 31) / 32;
  return g > 64 ? 64 : (g < 1 ? 1 : g);
}

// partial-row counts up to this are reduced by the finalize kernel itself (one launch instead of two)
#ifndef BN_SINGLE_STAGE_ROWS
#define BN_SINGLE_STAGE_ROWS bn_single_stage_rows()
#endif
// MMSKIN_BN_SINGLE_ROWS (default 512): layers 3 - 4 of ResNet-50 at batch 256 leave 98 - 392 partial rows (64 - 256 with the
// pipelined conv kernel's tiles) -- one 1024-thread finalize launch (16 row lanes x 2 chains) instead of pre-reduction + finalize
static inline int bn_single_stage_rows() {
  static const int v = [] { const char* e = getenv("MMSKIN_BN_SINGLE_ROWS"); return e ? atoi(e) : 512; }();
  return v;
}

// ------------------------------------------------------------------ BN forward
template <typename IN, int RL = 4>
__global__ __launch_bounds__(64 * RL) void bn_finalize_kernel(const IN* __restrict__ ssum, const IN* __restrict__ ssq, int nrows,
                                   int C, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum,
                                   float* running_mean, float* running_var, float* scale, float* shift,
                                   float* save_mean, float* save_invstd) {
  __shared__ double red[2][RL][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  double s = 0.0, q = 0.0;
  if (c < C) {
    int r = ry;
    double s1 = 0.0, q1 = 0.0, s2 = 0.0, q2 = 0.0, s3 = 0.0, q3 = 0.0;
    for (; r + 3 * RL < nrows; r += 4 * RL) {     // four independent chains per lane (eight loads in flight)
      s += (double)ssum[(size_t)r * C + c]; q += (double)ssq[(size_t)r * C + c];
      s1 += (double)ssum[(size_t)(r + RL) * C + c]; q1 += (double)ssq[(size_t)(r + RL) * C + c];
      s2 += (double)ssum[(size_t)(r + 2 * RL) * C + c]; q2 += (double)ssq[(size_t)(r + 2 * RL) * C + c];
      s3 += (double)ssum[(size_t)(r + 3 * RL) * C + c]; q3 += (double)ssq[(size_t)(r + 3 * RL) * C + c];
    }
    for (; r < nrows; r += RL) { s += (double)ssum[(size_t)r * C + c]; q += (double)ssq[(size_t)r * C + c]; }
    s = (s + s1) + (s2 + s3); q = (q + q1) + (q2 + q3);
  }
  red[0][ry][cx] = s; red[1][ry][cx] = q;
  __syncthreads();
  if (ry == 0 && c < C) {
    s = 0.0; q = 0.0;
#pragma unroll
    for (int i = 0; i < RL; ++i) { s += red[0][i][cx]; q += red[1][i][cx]; }
    bn_fwd_coeffs(c, s, q, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd);
  }
}

int bn_finalize(const float* stat_sum, const float* stat_sq, int nrows, int C, double count,
                const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                double* scratch, hipStream_t st) {
  if (scratch && nrows > BN_SINGLE_STAGE_ROWS) {   // (both stages in one launch behind a ticket per column block: +1.3 ms per step, profiles/r04_experiments.txt (11))
    const int G = reduce_groups(nrows);
    int rc = partial_reduce<double>(stat_sum, stat_sq, nrows, C, G, scratch, st);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3(ceil_div(C, 64)), dim3(256), 0, st, scratch,
                       scratch + (size_t)G * C, G, C, count, gamma, beta, eps, momentum, running_mean, running_var,
                       scale, shift, save_mean, save_invstd);
  } else if (nrows > 64) {   // up to BN_SINGLE_STAGE_ROWS partial rows in ONE launch: 16 row lanes x 2 chains each
    hipLaunchKernelGGL((bn_finalize_kernel<float, 16>), dim3(ceil_div(C, 64)), dim3(1024), 0, st, stat_sum, stat_sq, nrows,
                       C, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean,
                       save_invstd);
  } else {
    hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3(ceil_div(C, 64)), dim3(256), 0, st, stat_sum, stat_sq, nrows,
                       C, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean,
                       save_invstd);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float eps, float* scale, float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}
int bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, float eps, float* scale, float* shift, hipStream_t st) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, C, gamma, beta,
                     running_mean, running_var, eps, scale, shift);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T, int RELU, int RES, bool NT = false>  // RELU: 0 none, 1 relu / capped relu, 2 SiLU; RES: 0 none, 1 plain residual, 2 residual*rscale + rshift
__global__ __launch_bounds__(EW_BLOCK) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ rscale,
                                                            const float* __restrict__ rshift, T* __restrict__ y,
                                                            uint8_t* __restrict__ mask_bits, size_t nchunks, int CPR,
                                                            float relu_cap) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    int c0 = (int)(i % CPR) * EPC;
    Chunk<T> v;
    if (NT) v.load_nt(x + i * EPC); else v.load(x + i * EPC);     // the raw conv output is read once more only in backward
    Chunk<T> r;
    if (RES) { if (NT) r.load_nt(res + i * EPC); else r.load(res + i * EPC); }
    uint32_t bits = 0;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float t = v.v[e] * scale[c0 + e] + shift[c0 + e];
      if (RES == 1) t += r.v[e];
      if (RES == 2) t += r.v[e] * rscale[c0 + e] + rshift[c0 + e];
      // SiLU is its own instantiation: as a run-time branch its exp + divide were if-converted into the ReLU path
      // and made the (HBM-bound) pass VALU-bound -- 3x slower BatchNorm-apply on every backbone
      if (RELU == 2) t = t / (1.f + __expf(-t));
      else if (RELU == 1) { t = fmaxf(t, 0.f); if (relu_cap > 0.f) t = fminf(t, relu_cap); }
      v.v[e] = t;
      bits |= (from_f32<T>(t) != 0 && t > 0.f ? 1u : 0u) << e;   // bit = (stored y > 0)
    }
    v.store(y + i * EPC);
    if (mask_bits) mask_bits[i] = (uint8_t)bits;   // one byte per 16-byte chunk: the ReLU mask for backward
  }
}

template <typename T>
int bn_apply(const T* x, const T* res, const float* scale, const float* shift, const float* rscale,
             const float* rshift, T* y, size_t rows, int C, bool relu, hipStream_t st, uint8_t* mask_bits,
             float relu_cap) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0, "bn_apply: C=%d", C);
  size_t nch = rows * (C / EPC);
  int grid = ew_grid(nch);
  int mode = res ? (rscale ? 2 : 1) : 0;
#ifdef MMSKIN_ABLATE   // `make ablate` only: upper bound of folding the plain BN + ReLU apply into its consumers (wrong results, valid timing)
  { static const int abl = [] { const char* v = getenv("MMSKIN_BN_ABLATE"); return v ? atoi(v) : 0; }(); if ((abl & 1) && mode == 0 && relu) return MMSKIN_OK; }
#endif
#define LAUNCH(R, H) hipLaunchKernelGGL((bn_apply_kernel<T, R, H>), dim3(grid), dim3(EW_BLOCK), 0, st, x, res, scale, shift, rscale, rshift, y, mask_bits, nch, C / EPC, relu_cap)
  if (relu && relu_cap < 0.f) { if (mode == 2) LAUNCH(2, 2); else if (mode == 1) LAUNCH(2, 1); else LAUNCH(2, 0); }
  else if (relu) {
    static const bool nt = [] { const char* v = getenv("MMSKIN_EW_NT"); return !v || atoi(v) != 0; }();
#define LAUNCH_NT(R, H) hipLaunchKernelGGL((bn_apply_kernel<T, R, H, true>), dim3(grid), dim3(EW_BLOCK), 0, st, x, res, scale, shift, rscale, rshift, y, mask_bits, nch, C / EPC, relu_cap)
    if (nt) { if (mode == 2) LAUNCH_NT(1, 2); else if (mode == 1) LAUNCH_NT(1, 1); else LAUNCH_NT(1, 0); }
    else { if (mode == 2) LAUNCH(1, 2); else if (mode == 1) LAUNCH(1, 1); else LAUNCH(1, 0); }
#undef LAUNCH_NT
  }
  else { if (mode == 2) LAUNCH(0, 2); else if (mode == 1) LAUNCH(0, 1); else LAUNCH(0, 0); }
#undef LAUNCH
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void column_stats_kernel(const T* __restrict__ x, size_t rows, int C,
                                                           ColGeom g, float* partial_sum, float* partial_sq) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[2 * 256 * EPC];
  const int cx = threadIdx.x % g.CW, ry = threadIdx.x / g.CW;
  const int col = blockIdx.y * g.CW + cx;
  float acc[2][EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  if (ry < g.RL && col < g.CPR) {
    size_t r_end = (size_t)(blockIdx.x + 1) * g.RB;
    if (r_end > rows) r_end = rows;
    for (size_t r = (size_t)blockIdx.x * g.RB + ry; r < r_end; r += g.RL) {
      Chunk<T> v;
      v.load(x + (r * g.CPR + col) * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) { acc[0][e] += v.v[e]; acc[1][e] += v.v[e] * v.v[e]; }
    }
  }
  // two quantities into two separate slabs: reuse the NQ=1 reducer twice
  float a1[1][EPC], a2[1][EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { a1[0][e] = acc[0][e]; a2[0][e] = acc[1][e]; }
  block_col_reduce<EPC, 1>(a1, cx, ry, g.CW, g.RL, col, g.CPR, C, partial_sum, red);
  __syncthreads();
  block_col_reduce<EPC, 1>(a2, cx, ry, g.CW, g.RL, col, g.CPR, C, partial_sq, red);
}

int column_stats_rows(size_t rows, int C) {
  ColGeom a = col_geom(rows, C, 4), b = col_geom(rows, C, 8);
  return a.gx > b.gx ? a.gx : b.gx;
}
template <typename T>
int column_stats(const T* x, size_t rows, int C, float* stat_sum, float* stat_sq, int* nrows_out,
                 hipStream_t st) {
  ARG_CHECK(C % DT<T>::EPC == 0, "column_stats: C=%d", C);
  ColGeom g = col_geom(rows, C, DT<T>::EPC);
  hipLaunchKernelGGL(column_stats_kernel<T>, dim3(g.gx, g.gy), dim3(256), 0, st, x, rows, C, g, stat_sum, stat_sq);
  HIP_CHECK_RET(hipGetLastError());
  *nrows_out = g.gx;
  return MMSKIN_OK;
}

// ------------------------------------------------------------------ BN backward
template <typename T, int MODE>
__device__ __forceinline__ void masked_dy(Chunk<T>& dz, const Chunk<T>& xv, const T* ymask, size_t off,
                                          const float* scale, const float* shift, int c0) {
  constexpr int EPC = DT<T>::EPC;
  if (MODE == MASK_FROM_X) {
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      if (!(xv.v[e] * scale[c0 + e] + shift[c0 + e] > 0.f)) dz.v[e] = 0.f;
  } else if (MODE == MASK_FROM_Y) {
    Chunk<T> yv;
    yv.load(ymask + off);
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      if (!(yv.v[e] > 0.f)) dz.v[e] = 0.f;
  } else if (MODE == MASK_FROM_Y6) {
    Chunk<T> yv;
    yv.load(ymask + off);
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      if (!(yv.v[e] > 0.f && yv.v[e] < 6.f)) dz.v[e] = 0.f;
  } else if (MODE == MASK_SILU_X) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float t = xv.v[e] * scale[c0 + e] + shift[c0 + e];
      const float sg = 1.f / (1.f + __expf(-t));
      dz.v[e] *= sg * (1.f + t * (1.f - sg));
    }
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const T* __restrict__ ymask,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, size_t rows, int C,
                                                            ColGeom g, float* partial) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[2 * 256 * EPC];
  const int cx = threadIdx.x % g.CW, ry = threadIdx.x / g.CW;
  const int col = blockIdx.y * g.CW + cx;
  float acc[2][EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  if (ry < g.RL && col < g.CPR) {
    size_t r_end = (size_t)(blockIdx.x + 1) * g.RB;
    if (r_end > rows) r_end = rows;
    const int c0 = col * EPC;
    for (size_t r = (size_t)blockIdx.x * g.RB + ry; r < r_end; r += g.RL) {
      size_t off = (r * g.CPR + col) * EPC;
      Chunk<T> dz, xv;
      dz.load(dy + off);
      xv.load(x + off);
      masked_dy<T, MODE>(dz, xv, ymask, off, scale, shift, c0);
#pragma unroll
      for (int e = 0; e < EPC; ++e) { acc[0][e] += dz.v[e]; acc[1][e] += dz.v[e] * xv.v[e]; }
    }
  }
  block_col_reduce<EPC, 2>(acc, cx, ry, g.CW, g.RL, col, g.CPR, C, partial, red);
}

int bn_bwd_partial_rows(size_t rows, int C) { return column_stats_rows(rows, C); }

template <typename T>
int bn_bwd_reduce(const T* dy, const T* x, const T* ymask, const float* scale, const float* shift,
                  int mask_mode, size_t rows, int C, float* partial, int* nrows_out, hipStream_t st) {
  ARG_CHECK(C % DT<T>::EPC == 0, "bn_bwd_reduce: C=%d", C);
  ColGeom g = col_geom(rows, C, DT<T>::EPC);
#define LAUNCH(M) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, M>), dim3(g.gx, g.gy), dim3(256), 0, st, dy, x, ymask, scale, shift, rows, C, g, partial)
  if (mask_mode == MASK_FROM_X) LAUNCH(MASK_FROM_X);
  else if (mask_mode == MASK_FROM_Y) LAUNCH(MASK_FROM_Y);
  else if (mask_mode == MASK_FROM_Y6) LAUNCH(MASK_FROM_Y6);
  else if (mask_mode == MASK_SILU_X) LAUNCH(MASK_SILU_X);
  else LAUNCH(MASK_NONE);
#undef LAUNCH
  HIP_CHECK_RET(hipGetLastError());
  *nrows_out = g.gx;
  return MMSKIN_OK;
}

// channel c's BatchNorm-backward coefficients (dz = cA g + cB x + cC) and parameter gradients from s1 = sum dz, s2 = sum dz x
__device__ __forceinline__ void bn_bwd_coeffs(int c, double s1, double s2, double count, const float* __restrict__ gamma, const float* __restrict__ mean,
                                              const float* __restrict__ invstd, float* dgamma, float* dbeta, float* cA, float* cB, float* cC, int n_grad,
                                              int acc_bc, const float* __restrict__ s2_override) {
  if (s2_override) s2 = (double)s2_override[c];
  double mu = mean[c], is = invstd[c], g = gamma ? gamma[c] : 1.0;
  double dg = is * (s2 - mu * s1);   // sum dz * xhat
  if (dgamma && c < n_grad) dgamma[c] = (float)dg;
  if (dbeta && c < n_grad) dbeta[c] = (float)s1;
  double A = g * is;
  cA[c] = (float)A;
  const float vb = (float)(-A * is * dg / count), vc = (float)(A * (-s1 / count + mu * is * dg / count));
  if (acc_bc) { cB[c] += vb; cC[c] += vc; }   // running sums over the consumers of a shared input (DenseNet's deferred x / constant terms)
  else { cB[c] = vb; cC[c] = vc; }
}
// CB columns x RL row lanes per block.  CB = 16 (256 threads, 4 KB of LDS) is the backward pass's form: a block that small fits on a CU
// beside a weight-gradient ring workgroup of the other stream (208 VGPRs x 8 waves, 121 - 132 KB of LDS), where the 1024-thread form
// waited for a ring workgroup to retire -- i.e. for the whole weight-gradient launch (profiles/r04_experiments.txt (8)).
template <typename IN, int RL = 4, int CB = 64>
__global__ __launch_bounds__(CB * RL) void bn_bwd_finalize_kernel(const IN* __restrict__ partial, int nrows, int C, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* dgamma, float* dbeta,
                                       float* cA, float* cB, float* cC, int n_grad, int acc_bc, const float* __restrict__ s2_override = nullptr) {
  __shared__ double red[2][RL][CB];
  const int cx = threadIdx.x % CB, ry = threadIdx.x / CB;
  const int c = blockIdx.x * CB + cx;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    int r = ry;
    double t1 = 0.0, t2 = 0.0, u1 = 0.0, u2 = 0.0, v1 = 0.0, v2 = 0.0;
    for (; r + 3 * RL < nrows; r += 4 * RL) {     // four independent chains per lane (eight loads in flight)
      s1 += (double)partial[((size_t)r * 2) * C + c];
      s2 += (double)partial[((size_t)r * 2 + 1) * C + c];
      t1 += (double)partial[((size_t)(r + RL) * 2) * C + c];
      t2 += (double)partial[((size_t)(r + RL) * 2 + 1) * C + c];
      u1 += (double)partial[((size_t)(r + 2 * RL) * 2) * C + c];
      u2 += (double)partial[((size_t)(r + 2 * RL) * 2 + 1) * C + c];
      v1 += (double)partial[((size_t)(r + 3 * RL) * 2) * C + c];
      v2 += (double)partial[((size_t)(r + 3 * RL) * 2 + 1) * C + c];
    }
    for (; r < nrows; r += RL) { s1 += (double)partial[((size_t)r * 2) * C + c]; s2 += (double)partial[((size_t)r * 2 + 1) * C + c]; }
    s1 = (s1 + t1) + (u1 + v1); s2 = (s2 + t2) + (u2 + v2);
  }
  red[0][ry][cx] = s1; red[1][ry][cx] = s2;
  __syncthreads();
  if (ry == 0 && c < C) {
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int i = 0; i < RL; ++i) { s1 += red[0][i][cx]; s2 += red[1][i][cx]; }
    bn_bwd_coeffs(c, s1, s2, count, gamma, mean, invstd, dgamma, dbeta, cA, cB, cC, n_grad, acc_bc, s2_override);
  }
}
int bn_bwd_finalize(const float* partial, int nrows, int C, double count, const float* gamma,
                    const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                    float* cA, float* cB, float* cC, double* scratch, hipStream_t st, int n_grad, bool accumulate_bc, const float* sum_dz_x) {
  if (n_grad < 0) n_grad = C;
  const int acc_bc = accumulate_bc ? 1 : 0;
  if (scratch && nrows > BN_SINGLE_STAGE_ROWS) {
    const int G = reduce_groups(nrows);
    int rc = partial_reduce<double>(partial, nullptr, nrows, 2 * C, G, scratch, st);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3(ceil_div(C, 64)), dim3(256), 0, st, scratch, G, C, count,
                       gamma, save_mean, save_invstd, dgamma, dbeta, cA, cB, cC, n_grad, acc_bc, sum_dz_x);
  } else if (nrows > 64) {
    static const int small_blocks = [] { const char* e = getenv("MMSKIN_BNB_SMALL_BLOCKS"); return e ? atoi(e) : 1; }();
    if (small_blocks)
      hipLaunchKernelGGL((bn_bwd_finalize_kernel<float, 16, 16>), dim3(ceil_div(C, 16)), dim3(256), 0, st, partial, nrows, C, count,
                         gamma, save_mean, save_invstd, dgamma, dbeta, cA, cB, cC, n_grad, acc_bc, sum_dz_x);
    else
      hipLaunchKernelGGL((bn_bwd_finalize_kernel<float, 16>), dim3(ceil_div(C, 64)), dim3(1024), 0, st, partial, nrows, C, count,
                         gamma, save_mean, save_invstd, dgamma, dbeta, cA, cB, cC, n_grad, acc_bc, sum_dz_x);
  } else {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<float>, dim3(ceil_div(C, 64)), dim3(256), 0, st, partial, nrows, C, count,
                       gamma, save_mean, save_invstd, dgamma, dbeta, cA, cB, cC, n_grad, acc_bc, sum_dz_x);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T, int MODE, bool WRITE_DZ, bool NT = false>
__global__ __launch_bounds__(EW_BLOCK) void bn_bwd_apply_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ ymask,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ cA,
    const float* __restrict__ cB, const float* __restrict__ cC, T* __restrict__ dx, T* __restrict__ dz_out,
    size_t nchunks, int CPR) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const int c0 = (int)(i % CPR) * EPC;
    const size_t off = i * EPC;
    Chunk<T> dz, xv;
    if (NT) { dz.load_nt(dy + off); xv.load_nt(x + off); } else { dz.load(dy + off); xv.load(x + off); }
    masked_dy<T, MODE>(dz, xv, ymask, off, scale, shift, c0);
    if (WRITE_DZ) dz.store(dz_out + off);
#pragma unroll
    for (int e = 0; e < EPC; ++e) xv.v[e] = cA[c0 + e] * dz.v[e] + cB[c0 + e] * xv.v[e] + cC[c0 + e];
    xv.store(dx + off);
  }
}

template <typename T>
int bn_bwd_apply(const T* dy, const T* x, const T* ymask, const float* scale, const float* shift,
                 int mask_mode, const float* cA, const float* cB, const float* cC, T* dx, T* dz_out,
                 size_t rows, int C, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0, "bn_bwd_apply: C=%d", C);
  size_t nch = rows * (C / EPC);
  int grid = ew_grid(nch);
#ifdef MMSKIN_ABLATE   // `make ablate` only: upper bound of folding the BN-backward apply into the dgrad / weight-gradient operand loads
  { static const int abl = [] { const char* v = getenv("MMSKIN_BN_ABLATE"); return v ? atoi(v) : 0; }(); if ((abl & 2) && mask_mode == MASK_NONE && !dz_out) return MMSKIN_OK; }
#endif
#define LAUNCH(M, W) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, M, W>), dim3(grid), dim3(EW_BLOCK), 0, st, dy, x, ymask, scale, shift, cA, cB, cC, dx, dz_out, nch, C / EPC)
  if (mask_mode == MASK_FROM_X) { if (dz_out) LAUNCH(MASK_FROM_X, true); else LAUNCH(MASK_FROM_X, false); }
  else if (mask_mode == MASK_FROM_Y) { if (dz_out) LAUNCH(MASK_FROM_Y, true); else LAUNCH(MASK_FROM_Y, false); }
  else if (mask_mode == MASK_FROM_Y6) { if (dz_out) LAUNCH(MASK_FROM_Y6, true); else LAUNCH(MASK_FROM_Y6, false); }
  else if (mask_mode == MASK_SILU_X) { if (dz_out) LAUNCH(MASK_SILU_X, true); else LAUNCH(MASK_SILU_X, false); }
  else {
    // nontemporal loads of dz / x (each is read for the last time here; dx stays cacheable: the dgrad and the weight-gradient
    // GEMM read it next): same-box A/B 20.74 -> 20.43 ms per step (MMSKIN_EW_NT=0/1, profiles/r02_experiments.txt)
    static const bool nt = [] { const char* v = getenv("MMSKIN_EW_NT"); return !v || atoi(v) != 0; }();
    if (dz_out) LAUNCH(MASK_NONE, true);
    else if (nt) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, MASK_NONE, false, true>), dim3(grid), dim3(EW_BLOCK), 0, st, dy, x, ymask, scale, shift, cA, cB, cC, dx, dz_out, nch, C / EPC);
    else LAUNCH(MASK_NONE, false);
  }
#undef LAUNCH
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// ------------------------------------------------------------------ stem
template <typename T>
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ img, int N, int H, int W,
                                                        int Hp, int Wp, T* __restrict__ out) {
  // one padded pixel per thread, (column block, padded row, image) from the 3-D grid: the 1-D form spent its time in two 64-bit
  // divisions per pixel (168 us for a 256 x 3 x 224 x 224 batch = 1.2 TB/s; the pass moves 154 MB in and 110 MB out)
  const int wp = blockIdx.x * 256 + threadIdx.x, hp = blockIdx.y, n = blockIdx.z;
  if (wp >= Wp) return;
  const int h = hp - 3, w = wp - 3;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
  if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
    const float* src = img + ((size_t)n * 3 * H + h) * W + w;
    const size_t plane = (size_t)H * W;
    v0 = __builtin_nontemporal_load(src); v1 = __builtin_nontemporal_load(src + plane); v2 = __builtin_nontemporal_load(src + 2 * plane);
  }
  const size_t i = ((size_t)n * Hp + hp) * Wp + wp;
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(v0, v1, v2, 0.f);
  } else {
    uint32_t lo = f32_to_bf16_bits(v0) | (f32_to_bf16_bits(v1) << 16);
    uint32_t hi = f32_to_bf16_bits(v2);
    *reinterpret_cast<uint2*>(out + i * 4) = make_uint2(lo, hi);
  }
}
// W % 4 == 0, bf16: a thread moves FOUR source pixels -- three aligned 16-byte loads (one per colour plane), four 8-byte stores -- instead
// of one pixel with three 4-byte loads (256 B per wave-instruction: 164 us for the 154 MB + 108 MB this pass moves = 1.6 TB/s).  Threads
// W/4 and W/4 + 1 of a row write the left / right zero borders; rows outside the image are zeros throughout.  Block = 64 x 4 rows.
__global__ __launch_bounds__(256) void stem_pack4_kernel(const float* __restrict__ img, int H, int W, int Hp, int Wp, bf16_t* __restrict__ out) {
  const int t = blockIdx.x * 64 + (threadIdx.x & 63), hp = blockIdx.y * 4 + (threadIdx.x >> 6), n = blockIdx.z;
  const int nq = W / 4;
  if (hp >= Hp || t >= nq + 2) return;
  uint2* row = reinterpret_cast<uint2*>(out) + ((size_t)n * Hp + hp) * Wp;
  const int h = hp - 3;
  const uint2 z = make_uint2(0u, 0u);
  if (t >= nq) {   // borders: padded columns 0..2, or W + 3 .. Wp - 1
    const int b = t == nq ? 0 : W + 3, e = t == nq ? 3 : Wp;
    for (int wp = b; wp < e; ++wp) row[wp] = z;
    return;
  }
  uint2 o[4] = {z, z, z, z};
  if ((unsigned)h < (unsigned)H) {
    const float* src = img + ((size_t)n * 3 * H + h) * W + 4 * t;
    const size_t plane = (size_t)H * W;
    const f32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src));
    const f32x4_t g = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + plane));
    const f32x4_t b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + 2 * plane));
    o[0] = make_uint2(f32_to_bf16_bits(r.x) | (f32_to_bf16_bits(g.x) << 16), f32_to_bf16_bits(b.x));
    o[1] = make_uint2(f32_to_bf16_bits(r.y) | (f32_to_bf16_bits(g.y) << 16), f32_to_bf16_bits(b.y));
    o[2] = make_uint2(f32_to_bf16_bits(r.z) | (f32_to_bf16_bits(g.z) << 16), f32_to_bf16_bits(b.z));
    o[3] = make_uint2(f32_to_bf16_bits(r.w) | (f32_to_bf16_bits(g.w) << 16), f32_to_bf16_bits(b.w));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) row[4 * t + 3 + i] = o[i];
}
template <typename T>
int stem_pack(const float* img, int N, int H, int W, int Hp, int Wp, T* img4, hipStream_t st) {
  ARG_CHECK(Hp >= H + 6 && Wp >= W + 6 && Wp % 2 == 0, "stem_pack: bad padded size");
  ARG_CHECK(Hp <= 65535 && N <= 65535, "stem_pack: grid %d x %d", Hp, N);
  if constexpr (sizeof(T) == 2) {
    static const bool pack4 = [] { const char* v = getenv("MMSKIN_STEM_PACK4"); return !v || atoi(v) != 0; }();
    if (pack4 && W % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0) {
      hipLaunchKernelGGL(stem_pack4_kernel, dim3(ceil_div(W / 4 + 2, 64), ceil_div(Hp, 4), N), dim3(256), 0, st, img, H, W, Hp, Wp, img4);
      HIP_CHECK_RET(hipGetLastError());
      return MMSKIN_OK;
    }
  }
  hipLaunchKernelGGL(stem_pack_kernel<T>, dim3(ceil_div(Wp, 256), Hp, N), dim3(256), 0, st, img, N, H, W, Hp, Wp, img4);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

struct Norm6 { float a[3], b[3]; };   // v = u8 * a[c] + b[c]
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void stem_pack_u8_kernel(const uint8_t* __restrict__ img, int N, int H, int W,
                                                               int Hp, int Wp, Norm6 nm, T* __restrict__ out) {
  const size_t total = (size_t)N * Hp * Wp;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < total; i += (size_t)gridDim.x * EW_BLOCK) {
    int wp = (int)(i % Wp);
    size_t t = i / Wp;
    int hp = (int)(t % Hp), n = (int)(t / Hp);
    int h = hp - 3, w = wp - 3;
    float v[3] = {0.f, 0.f, 0.f};
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
      const uint8_t* px = img + (((size_t)n * H + h) * W + w) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (float)px[c] * nm.a[c] + nm.b[c];
    }
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(out + i * 4) = make_float4(v[0], v[1], v[2], 0.f);
    } else {
      uint32_t lo = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
      uint32_t hi = f32_to_bf16_bits(v[2]);
      *reinterpret_cast<uint2*>(out + i * 4) = make_uint2(lo, hi);
    }
  }
}
template <typename T>
int stem_pack_u8(const uint8_t* img_nhwc, int N, int H, int W, int Hp, int Wp, const float* norm6, T* img4,
                 hipStream_t st) {
  ARG_CHECK(Hp >= H + 6 && Wp >= W + 6 && Wp % 2 == 0, "stem_pack_u8: bad padded size");
  Norm6 nm;
  for (int c = 0; c < 3; ++c) {
    ARG_CHECK(norm6[3 + c] > 0.f, "stem_pack_u8: std[%d] = %f", c, norm6[3 + c]);
    nm.a[c] = 1.f / (255.f * norm6[3 + c]);
    nm.b[c] = -norm6[c] / norm6[3 + c];
  }
  hipLaunchKernelGGL(stem_pack_u8_kernel<T>, dim3(ew_grid((size_t)N * Hp * Wp)), dim3(EW_BLOCK), 0, st, img_nhwc, N, H, W, Hp,
                     Wp, nm, img4);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// grid (ceil(PW*CPR / 256), PH, N): the pooled row and the image come from the block index, so a thread needs one 32-bit
// division (64-bit div/mod chains per chunk made these stem kernels instruction-bound at ~2 TB/s)
template <typename T>
__global__ __launch_bounds__(256) void stem_bn_relu_pool_kernel(const T* __restrict__ x,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int N, int H,
                                                                int W, int C, int PH, int PW,
                                                                T* __restrict__ y, uint8_t* __restrict__ idx) {
  constexpr int EPC = DT<T>::EPC;
  const unsigned CPR = C / EPC;
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  if (t >= (unsigned)PW * CPR) return;
  const int pw = (int)(t / CPR), cj = (int)(t - (unsigned)pw * CPR);
  const int ph = blockIdx.y, n = blockIdx.z;
  const int c0 = cj * EPC;
  float sc[EPC], sh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; }
  float best[EPC];
  int bi[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { best[e] = -INFINITY; bi[e] = -1; }
  const T* xn = x + (size_t)n * H * W * C + c0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      int h = 2 * ph - 1 + r, w = 2 * pw - 1 + s;
      if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
        Chunk<T> v;
        v.load(xn + (size_t)(h * W + w) * C);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float a = fmaxf(v.v[e] * sc[e] + sh[e], 0.f);
          if (a > best[e] || bi[e] < 0 || a != a) { best[e] = a; bi[e] = r * 3 + s; }
        }
      }
    }
  const size_t i = ((size_t)(n * PH + ph) * PW + pw) * CPR + cj;
  Chunk<T> o;
#pragma unroll
  for (int e = 0; e < EPC; ++e) o.v[e] = best[e];
  o.store(y + i * EPC);
  if constexpr (EPC == 8) {
    uint32_t lo = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    uint32_t hi = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    *reinterpret_cast<uint2*>(idx + i * 8) = make_uint2(lo, hi);
  } else {
    *reinterpret_cast<uint32_t*>(idx + i * 4) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
  }
}
template <typename T>
int stem_bn_relu_pool(const T* x, const float* scale, const float* shift, int N, int H, int W, int C, T* y,
                      uint8_t* idx, hipStream_t st) {
  int PH = (H + 2 - 3) / 2 + 1, PW = (W + 2 - 3) / 2 + 1;
  ARG_CHECK(PH <= 65535 && N <= 65535, "stem_bn_relu_pool: grid %dx%d", PH, N);
  hipLaunchKernelGGL(stem_bn_relu_pool_kernel<T>, dim3(ceil_div(PW * (C / DT<T>::EPC), 256), PH, N), dim3(256), 0, st, x, scale,
                     shift, N, H, W, C, PH, PW, y, idx);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// Stem backward works on pooled CELLS: the 2x2 input pixels (2ph..2ph+1, 2pw..2pw+1) receive gradient from the four
// pooling windows (ph..ph+1, pw..pw+1) only, so one thread loads those four windows once (gradient + argmax bytes) and
// the four pixels' x -- 12 independent 16-byte loads in flight instead of a dependent gather per pixel.
//   pixel (2ph  ,2pw  ): window (ph,pw) tap 4
//   pixel (2ph  ,2pw+1): (ph,pw) tap 5, (ph,pw+1) tap 3
//   pixel (2ph+1,2pw  ): (ph,pw) tap 7, (ph+1,pw) tap 1
//   pixel (2ph+1,2pw+1): (ph,pw) tap 8, (ph,pw+1) tap 6, (ph+1,pw) tap 2, (ph+1,pw+1) tap 0      (tap = 3*r + s)
// Sums run in ascending (ph, pw) window order (what a per-pixel gather over its windows does too).
template <typename T>
__device__ __forceinline__ void stem_cell_grad(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ x,
                                               const float* __restrict__ scale, const float* __restrict__ shift, size_t n, int ph,
                                               int pw, int cj, int H, int W, int C, int PH, int PW,
                                               float (&d)[4][DT<T>::EPC], float (&xo)[4][DT<T>::EPC], bool (&ok)[4]) {
  constexpr int EPC = DT<T>::EPC;
  const int c0 = cj * EPC;
  const bool wv[4] = {true, pw + 1 < PW, ph + 1 < PH, (pw + 1 < PW) && (ph + 1 < PH)};
  const size_t pbase = ((n * PH + ph) * PW + pw) * C + c0;
  const size_t poff[4] = {0, (size_t)C, (size_t)PW * C, (size_t)(PW + 1) * C};
  Chunk<T> g[4];
  uint32_t iw[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    iw[q][0] = 0xffffffffu; iw[q][1] = 0xffffffffu;       // no tap matches a window that does not exist
    if (wv[q]) {
      g[q].load(dpool + pbase + poff[q]);
      if constexpr (EPC == 8) {
        uint2 t = *reinterpret_cast<const uint2*>(idx + pbase + poff[q]);
        iw[q][0] = t.x; iw[q][1] = t.y;
      } else {
        iw[q][0] = *reinterpret_cast<const uint32_t*>(idx + pbase + poff[q]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) g[q].v[e] = 0.f;
    }
  }
  const int h0 = 2 * ph, w0 = 2 * pw;
  ok[0] = true; ok[1] = w0 + 1 < W; ok[2] = h0 + 1 < H; ok[3] = ok[1] && ok[2];
  const size_t xbase = ((n * H + h0) * W + w0) * C + c0;
  const size_t xoff[4] = {0, (size_t)C, (size_t)W * C, (size_t)(W + 1) * C};
  Chunk<T> xv[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (ok[p]) xv[p].load(x + xbase + xoff[p]);
    else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) xv[p].v[e] = 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int t0 = (iw[0][e >> 2] >> (8 * (e & 3))) & 0xff, t1 = (iw[1][e >> 2] >> (8 * (e & 3))) & 0xff;
    const int t2 = (iw[2][e >> 2] >> (8 * (e & 3))) & 0xff, t3 = (iw[3][e >> 2] >> (8 * (e & 3))) & 0xff;
    const float g0 = g[0].v[e], g1 = g[1].v[e], g2 = g[2].v[e], g3 = g[3].v[e];
    float a;
    d[0][e] = (t0 == 4) ? g0 : 0.f;
    a = (t0 == 5) ? g0 : 0.f; if (t1 == 3) a += g1; d[1][e] = a;
    a = (t0 == 7) ? g0 : 0.f; if (t2 == 1) a += g2; d[2][e] = a;
    a = (t0 == 8) ? g0 : 0.f; if (t1 == 6) a += g1; if (t2 == 2) a += g2; if (t3 == 0) a += g3; d[3][e] = a;
    const float sc = scale[c0 + e], sh = shift[c0 + e];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      xo[p][e] = xv[p].v[e];
      if (!(xv[p].v[e] * sc + sh > 0.f)) d[p][e] = 0.f;     // ReLU mask recomputed from x
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bn_bwd_reduce_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                     const T* __restrict__ x, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, int H, int W, int C, int PH,
                                                                     int PW, size_t cells, ColGeom g, float* partial) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[2 * 256 * EPC];
  const int cx = threadIdx.x % g.CW, ry = threadIdx.x / g.CW;
  const int col = blockIdx.y * g.CW + cx;
  float acc[2][EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  if (ry < g.RL && col < g.CPR) {
    size_t r_end = (size_t)(blockIdx.x + 1) * g.RB;
    if (r_end > cells) r_end = cells;
    for (size_t r = (size_t)blockIdx.x * g.RB + ry; r < r_end; r += g.RL) {
      const unsigned r32 = (unsigned)r, t = r32 / (unsigned)PW;     // cells < 2^32 (checked by the launcher)
      const int pw = (int)(r32 - t * (unsigned)PW);
      const unsigned n = t / (unsigned)PH;
      const int ph = (int)(t - n * (unsigned)PH);
      float d[4][EPC], xv[4][EPC];
      bool ok[4];
      stem_cell_grad<T>(dpool, idx, x, scale, shift, n, ph, pw, col, H, W, C, PH, PW, d, xv, ok);
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (ok[p]) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) { acc[0][e] += d[p][e]; acc[1][e] += d[p][e] * xv[p][e]; }
        }
    }
  }
  block_col_reduce<EPC, 2>(acc, cx, ry, g.CW, g.RL, col, g.CPR, C, partial, red);
}
template <typename T>
int stem_pool_bn_bwd_reduce(const T* dpool, const uint8_t* idx, const T* x, const float* scale, const float* shift, int N,
                            int H, int W, int C, float* partial, int* nrows_out, hipStream_t st) {
  ARG_CHECK(C % DT<T>::EPC == 0, "stem_pool_bn_bwd_reduce: C=%d", C);
  const int PH = (H + 2 - 3) / 2 + 1, PW = (W + 2 - 3) / 2 + 1;
  const size_t cells = (size_t)N * PH * PW;
  ARG_CHECK(cells < ((size_t)1 << 32), "stem_pool_bn_bwd_reduce: %zu cells", cells);
  ColGeom g = col_geom(cells, C, DT<T>::EPC);
  // col_geom aims at ~1000 blocks (25 cells per lane, 12 loads each, two integer divisions per cell): 294 us for the 540 MB the pass
  // reads.  Four cells per lane instead (6 272 blocks at batch 256): the partial rows go through the two-stage reduction like the
  // dgrad epilogues' (MMSKIN_STEM_REDUCE_CELLS: cells per lane)
  static const int cells_per_lane = [] { const char* v = getenv("MMSKIN_STEM_REDUCE_CELLS"); return v ? atoi(v) : 4; }();
  if (cells_per_lane > 0 && (size_t)g.RL * cells_per_lane < (size_t)g.RB) {
    g.RB = g.RL * cells_per_lane;
    g.gx = (int)((cells + g.RB - 1) / g.RB);
  }
  hipLaunchKernelGGL(stem_pool_bn_bwd_reduce_kernel<T>, dim3(g.gx, g.gy), dim3(256), 0, st, dpool, idx, x, scale, shift, H, W, C, PH,
                     PW, cells, g, partial);
  HIP_CHECK_RET(hipGetLastError());
  *nrows_out = g.gx;
  return MMSKIN_OK;
}
// The same two sums from the POOLED side: every pooled cell hands its gradient to exactly one conv cell (its argmax), that cell's ReLU
// is open exactly when the pooled value is positive, and its pre-BatchNorm value follows from the pooled value, x = (y - shift) / scale
// (y is the rounded relu(scale x + shift): the reconstruction differs from the stored x by y's rounding over scale, the size of x's own
// rounding).  So  sum dz = sum_pooled [y > 0] dpool,  sum dz x = sum_pooled [y > 0] dpool x(y)  -- one pass over the two pooled tensors
// (206 MB at batch 256) instead of the conv output + pooled gradient + argmax bytes (565 MB, 12 loads and two divisions per cell).
// A channel whose scale is exactly zero (gamma = 0) cannot be inverted: it reads x at the argmax the forward recorded (never taken otherwise).
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bwd_sums_kernel(const T* __restrict__ dpool, const T* __restrict__ ypool, const uint8_t* __restrict__ idx,
                                                                const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                int H, int W, int PH, int PW, size_t cells, int C, ColGeom g, float* partial) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[2 * 256 * EPC];
  const int cx = threadIdx.x % g.CW, ry = threadIdx.x / g.CW;
  const int col = blockIdx.y * g.CW + cx;
  float acc[2][EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  if (ry < g.RL && col < g.CPR) {
    size_t r_end = (size_t)(blockIdx.x + 1) * g.RB;
    if (r_end > cells) r_end = cells;
    const int c0 = col * EPC;
    float inv[EPC], sh[EPC];
    bool degenerate = false;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float sc = scale[c0 + e];
      sh[e] = shift[c0 + e];
      inv[e] = sc != 0.f ? 1.f / sc : 0.f;
      degenerate |= sc == 0.f;
    }
    for (size_t r = (size_t)blockIdx.x * g.RB + ry; r < r_end; r += g.RL) {
      const size_t off = (r * g.CPR + col) * EPC;
      Chunk<T> dz, yv;
      dz.load(dpool + off);
      yv.load(ypool + off);
      float xr[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) xr[e] = (yv.v[e] - sh[e]) * inv[e];
      if (degenerate) {
        const unsigned r32 = (unsigned)r, t = r32 / (unsigned)PW;
        const int pw = (int)(r32 - t * (unsigned)PW);
        const unsigned n = t / (unsigned)PH;
        const int ph = (int)(t - n * (unsigned)PH);
        for (int e = 0; e < EPC; ++e)
          if (inv[e] == 0.f) {
            const int tap = idx[off + e], hh = 2 * ph - 1 + tap / 3, ww = 2 * pw - 1 + tap % 3;
            xr[e] = ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) ? to_f32(x[(((size_t)n * H + hh) * W + ww) * C + c0 + e]) : 0.f;
          }
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float v = yv.v[e] > 0.f ? dz.v[e] : 0.f;
        acc[0][e] += v; acc[1][e] += v * xr[e];
      }
    }
  }
  block_col_reduce<EPC, 2>(acc, cx, ry, g.CW, g.RL, col, g.CPR, C, partial, red);
}
template <typename T>
int stem_pool_bwd_sums(const T* dpool, const T* ypool, const uint8_t* idx, const T* x, const float* scale, const float* shift, int N, int H, int W,
                       int C, float* partial, int* nrows_out, hipStream_t st) {
  ARG_CHECK(C % DT<T>::EPC == 0, "stem_pool_bwd_sums: C=%d", C);
  const int PH = (H + 2 - 3) / 2 + 1, PW = (W + 2 - 3) / 2 + 1;
  const size_t cells = (size_t)N * PH * PW;
  ARG_CHECK(cells < ((size_t)1 << 32), "stem_pool_bwd_sums: %zu cells", cells);
  ColGeom g = col_geom(cells, C, DT<T>::EPC);
  hipLaunchKernelGGL(stem_pool_bwd_sums_kernel<T>, dim3(g.gx, g.gy), dim3(256), 0, st, dpool, ypool, idx, x, scale, shift, H, W, PH, PW, cells, C, g, partial);
  HIP_CHECK_RET(hipGetLastError());
  *nrows_out = g.gx;
  return MMSKIN_OK;
}
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_bn_bwd_apply_kernel(
    const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC,
    int H, int W, int C, int PH, int PW, T* __restrict__ dx) {
  constexpr int EPC = DT<T>::EPC;
  const unsigned CPR = C / EPC;
  const unsigned t = blockIdx.x * 256u + threadIdx.x;     // grid (ceil(PW*CPR / 256), PH, N): one thread per pooled cell and chunk
  if (t >= (unsigned)PW * CPR) return;
  const int pw = (int)(t / CPR), cj = (int)(t - (unsigned)pw * CPR), c0 = cj * EPC;
  const int ph = blockIdx.y;
  const size_t n = blockIdx.z;
  float d[4][EPC], xv[4][EPC];
  bool ok[4];
  stem_cell_grad<T>(dpool, idx, x, scale, shift, n, ph, pw, cj, H, W, C, PH, PW, d, xv, ok);
  const size_t xbase = ((n * H + 2 * ph) * W + 2 * pw) * C + c0;
  const size_t xoff[4] = {0, (size_t)C, (size_t)W * C, (size_t)(W + 1) * C};
#pragma unroll
  for (int p = 0; p < 4; ++p)
    if (ok[p]) {
      Chunk<T> o;
#pragma unroll
      for (int e = 0; e < EPC; ++e) o.v[e] = cA[c0 + e] * d[p][e] + cB[c0 + e] * xv[p][e] + cC[c0 + e];
      o.store(dx + xbase + xoff[p]);
    }
}
template <typename T>
int stem_pool_bn_bwd_apply(const T* dpool, const uint8_t* idx, const T* x, const float* scale, const float* shift,
                           const float* cA, const float* cB, const float* cC, int N, int H, int W, int C, T* dx, hipStream_t st) {
  ARG_CHECK(C % DT<T>::EPC == 0, "stem_pool_bn_bwd_apply: C=%d", C);
  const int PH = (H + 2 - 3) / 2 + 1, PW = (W + 2 - 3) / 2 + 1;
  ARG_CHECK(PH <= 65535 && N <= 65535, "stem_pool_bn_bwd_apply: grid %dx%d", PH, N);
  hipLaunchKernelGGL(stem_pool_bn_bwd_apply_kernel<T>, dim3(ceil_div(PW * (C / DT<T>::EPC), 256), PH, N), dim3(256), 0, st, dpool, idx,
                     x, scale, shift, cA, cB, cC, H, W, C, PH, PW, dx);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// ------------------------------------------------------------------ global average pool
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void avgpool_fwd_kernel(const T* __restrict__ x, int N, int HW, int C,
                                                              float* __restrict__ feat) {
  constexpr int EPC = DT<T>::EPC;
  const int CPR = C / EPC;
  const int total = N * CPR;
  for (int i = blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += gridDim.x * EW_BLOCK) {
    int n = i / CPR, cj = i - n * CPR;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int p = 0; p < HW; ++p) {
      Chunk<T> v;
      v.load(x + ((size_t)n * HW + p) * C + cj * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += v.v[e];
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < EPC; ++e) feat[(size_t)n * C + cj * EPC + e] = acc[e] * inv;
  }
}
template <typename T>
int avgpool_fwd(const T* x, int N, int HW, int C, float* feat, hipStream_t st) {
  hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(ew_grid((size_t)N * (C / DT<T>::EPC))), dim3(EW_BLOCK), 0, st, x, N, HW, C, feat);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void avgpool_bwd_kernel(const float* __restrict__ dfeat, int N, int HW,
                                                              int C, T* __restrict__ dx) {
  constexpr int EPC = DT<T>::EPC;
  const int CPR = C / EPC;
  const size_t total = (size_t)N * HW * CPR;
  const float inv = 1.f / (float)HW;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < total; i += (size_t)gridDim.x * EW_BLOCK) {
    int cj = (int)(i % CPR);
    int n = (int)(i / ((size_t)HW * CPR));
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = dfeat[(size_t)n * C + cj * EPC + e] * inv;
    o.store(dx + i * EPC);
  }
}
template <typename T>
int avgpool_bwd(const float* dfeat, int N, int HW, int C, T* dx, hipStream_t st) {
  hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(ew_grid((size_t)N * HW * (C / DT<T>::EPC))), dim3(EW_BLOCK), 0, st, dfeat, N, HW, C, dx);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// ------------------------------------------------------------------ weight staging
// Destination-major: every thread produces 1 (stem) or 8 consecutive destination elements, so the bf16
// stores are 16-byte coalesced and only the fp32 reads are strided (they hit L2: one layer's weights are
// re-read by its neighbours).  blockIdx.y = layer; blockIdx.z = 0 forward layout, 1 dgrad layout.
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void stage_weights_kernel(const StageDesc* __restrict__ table,
                                                                const float* __restrict__ params,
                                                                T* __restrict__ wfwd, T* __restrict__ wdgrad,
                                                                int need_dgrad, const float* __restrict__ fold_buffers,
                                                                float eps) {
  constexpr int EPC = DT<T>::EPC;
  const StageDesc d = table[blockIdx.y];
  const float* src = params + d.src_off;
  const int CT = d.Cin * d.taps;
  if (d.stem) {
    if (blockIdx.z) return;
    const int total = d.Cout * d.Cin * d.taps;
    for (int i = blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += gridDim.x * EW_BLOCK) {
      int o = i / CT, rem = i - o * CT;
      int c = rem / d.taps, t = rem - c * d.taps;
      int r = t / 7, s = t - r * 7;  // 7x7 taps, Cin = 3 -> wv[o][r][s*4 + c]
      wfwd[d.fwd_off + ((size_t)o * 8 + r) * 32 + s * 4 + c] = from_f32<T>(src[i]);
    }
    return;
  }
  if (blockIdx.z && !need_dgrad) return;
  const int Cop = d.Cout_pad ? d.Cout_pad : d.Cout, Cip = d.Cin_pad ? d.Cin_pad : d.Cin;
  const int nchunks = Cop * Cip * d.taps / EPC;   // staged Cin and Cout are multiples of EPC for every GEMM conv
  for (int i = blockIdx.x * EW_BLOCK + threadIdx.x; i < nchunks; i += gridDim.x * EW_BLOCK) {
    Chunk<T> v;
    if (blockIdx.z == 0) {         // dst [o][t][c0..c0+EPC)
      const int cpr = Cip / EPC;
      int c0 = (i % cpr) * EPC, ot = i / cpr;
      int t = ot % d.taps, o = ot / d.taps;
      const float* s0 = src + ((size_t)o * d.Cin + c0) * d.taps + t;
      float fold = 1.f;
      if (fold_buffers && d.has_bn && o < d.Cout) fold = params[d.bn_g_off + o] / sqrtf(fold_buffers[d.bn_rv_off + o] + eps);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = (o < d.Cout && c0 + e < d.Cin) ? s0[(size_t)e * d.taps] * fold : 0.f;
      v.store(wfwd + d.fwd_off + (size_t)i * EPC);
    } else {                       // dst [c][t][o0..o0+EPC)
      // consecutive lanes walk (c, t) -- the SOURCE's contiguous index inside a row o -- so each of the EPC loads of a wave is one
      // coalesced run; with lanes along o every 4-byte read pulled its own 128-byte line (1.0 GB of HBM reads per step for 102 MB of
      // parameters, profiles/r03_step_traffic.txt).  The stores become 16-byte pieces Cop elements apart: 4x their bytes at worst.
      const int ctn = Cip * d.taps;
      const int ct = i % ctn, o0 = (i / ctn) * EPC;
      const int t = ct % d.taps, c = ct / d.taps;
      const float* s0 = src + (size_t)o0 * CT + (size_t)c * d.taps + t;
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = (c < d.Cin && o0 + e < d.Cout) ? s0[(size_t)e * CT] : 0.f;
      v.store(wdgrad + d.dgrad_off + (size_t)ct * Cop + o0);
    }
  }
}
template <typename T>
int stage_weights(const StageDesc* table_dev, int nlayers, int max_elems, const float* params, T* wfwd,
                  T* wdgrad, bool need_dgrad, hipStream_t st, const float* fold_buffers, float eps) {
  int gx = ceil_div(max_elems / DT<T>::EPC, EW_BLOCK * 2);
  if (gx > 2048) gx = 2048;   // short workgroups (one or two chunks per thread): 0.30 -> see DESIGN.md
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(stage_weights_kernel<T>, dim3(gx, nlayers, need_dgrad ? 2 : 1), dim3(EW_BLOCK), 0, st, table_dev,
                     params, wfwd, wdgrad, need_dgrad ? 1 : 0, fold_buffers, eps);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

__global__ void bn_eval_table_kernel(const StageDesc* __restrict__ table, const float* __restrict__ params,
                                     const float* __restrict__ buffers, unsigned char* ws, float eps) {
  const StageDesc d = table[blockIdx.y];
  if (!d.has_bn) return;
  float* coef = reinterpret_cast<float*>(ws + d.coef_off);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < d.Cout; c += gridDim.x * blockDim.x) {
    float sc = params[d.bn_g_off + c] / sqrtf(buffers[d.bn_rv_off + c] + eps);
    coef[c] = sc;
    coef[d.Cout + c] = params[d.bn_b_off + c] - buffers[d.bn_rm_off + c] * sc;
  }
}
int bn_eval_table(const StageDesc* table_dev, int nlayers, int maxC, const float* params, const float* buffers,
                  unsigned char* ws, float eps, hipStream_t st) {
  hipLaunchKernelGGL(bn_eval_table_kernel, dim3(ceil_div(maxC, 256), nlayers), dim3(256), 0, st, table_dev, params, buffers, ws, eps);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

__global__ void stem_wgrad_unpack_kernel(const float* __restrict__ dwv, float* __restrict__ dw) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;  // over 64*3*7*7
  if (i < 64 * 147) {
    int o = i / 147, rem = i - o * 147;
    int c = rem / 49, t = rem - c * 49;
    int r = t / 7, s = t - r * 7;
    dw[i] = dwv[((size_t)o * 8 + r) * 32 + s * 4 + c];
  }
}
int stem_wgrad_unpack(const float* dwv, float* dw, hipStream_t st) {
  hipLaunchKernelGGL(stem_wgrad_unpack_kernel, dim3(ceil_div(64 * 147, 256)), dim3(256), 0, st, dwv, dw);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// ------------------------------------------------------------------ layout converters
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, int N, int C, int H, int W, T* __restrict__ dst) {
  const size_t total = (size_t)N * C * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    size_t t = i / C;
    int w = (int)(t % W); t /= W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    dst[i] = from_f32<T>(src[(((size_t)n * C + c) * H + h) * W + w]);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, int N, int C, int H, int W, float* __restrict__ dst) {
  const size_t total = (size_t)N * C * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int w = (int)(i % W);
    size_t t = i / W;
    int h = (int)(t % H); t /= H;
    int c = (int)(t % C);
    int n = (int)(t / C);
    dst[i] = to_f32(src[(((size_t)n * H + h) * W + w) * C + c]);
  }
}
template <typename T>
int nchw_to_nhwc(const float* src, int N, int C, int H, int W, T* dst, hipStream_t st) {
  hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(ew_grid((size_t)N * C * H * W)), dim3(256), 0, st, src, N, C, H, W, dst);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template <typename T>
int nhwc_to_nchw(const T* src, int N, int C, int H, int W, float* dst, hipStream_t st) {
  hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(ew_grid((size_t)N * C * H * W)), dim3(256), 0, st, src, N, C, H, W, dst);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

#define INST(T)                                                                                               \
  template int bn_apply<T>(const T*, const T*, const float*, const float*, const float*, const float*, T*, size_t, int, bool, hipStream_t, uint8_t*, float); \
  template int column_stats<T>(const T*, size_t, int, float*, float*, int*, hipStream_t);                       \
  template int bn_bwd_reduce<T>(const T*, const T*, const T*, const float*, const float*, int, size_t, int, float*, int*, hipStream_t); \
  template int bn_bwd_apply<T>(const T*, const T*, const T*, const float*, const float*, int, const float*, const float*, const float*, T*, T*, size_t, int, hipStream_t); \
  template int stem_pack<T>(const float*, int, int, int, int, int, T*, hipStream_t);                            \
  template int stem_pack_u8<T>(const uint8_t*, int, int, int, int, int, const float*, T*, hipStream_t);        \
  template int stem_bn_relu_pool<T>(const T*, const float*, const float*, int, int, int, int, T*, uint8_t*, hipStream_t); \
  template int stem_pool_bn_bwd_reduce<T>(const T*, const uint8_t*, const T*, const float*, const float*, int, int, int, int, float*, int*, hipStream_t); \
  template int stem_pool_bwd_sums<T>(const T*, const T*, const uint8_t*, const T*, const float*, const float*, int, int, int, int, float*, int*, hipStream_t); \
  template int stem_pool_bn_bwd_apply<T>(const T*, const uint8_t*, const T*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, T*, hipStream_t); \
  template int avgpool_fwd<T>(const T*, int, int, int, float*, hipStream_t);                                    \
  template int avgpool_bwd<T>(const float*, int, int, int, T*, hipStream_t);                                    \
  template int stage_weights<T>(const StageDesc*, int, int, const float*, T*, T*, bool, hipStream_t, const float*, float); \
  template int nchw_to_nhwc<T>(const float*, int, int, int, int, T*, hipStream_t);                              \
  template int nhwc_to_nchw<T>(const T*, int, int, int, int, float*, hipStream_t);
INST(float)
INST(bf16_t)

// ------------------------------------------------------------------ channel-slice kernels (DenseNet)
template <typename T, bool BN>
__global__ __launch_bounds__(EW_BLOCK) void slice_pack_kernel(const T* __restrict__ in, int pitch, int CPRin, int CPRout,
                                                              size_t nchunks, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, T* __restrict__ out) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const size_t r = i / CPRout;
    const int cc = (int)(i - r * CPRout);
    Chunk<T> v;
    if (cc < CPRin) {
      v.load(in + r * pitch + (size_t)cc * EPC);
      if (BN) {
        const int c0 = cc * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] = fmaxf(v.v[e] * scale[c0 + e] + shift[c0 + e], 0.f);
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = 0.f;
    }
    v.store(out + i * EPC);
  }
}
template <typename T>
int slice_pack(const T* in, int pitch, int C, int Cp, size_t rows, const float* scale, const float* shift, T* out,
               hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && Cp % EPC == 0 && pitch % EPC == 0 && C <= Cp, "slice_pack: C=%d Cp=%d pitch=%d", C, Cp, pitch);
  const size_t nch = rows * (Cp / EPC);
  if (scale)
    hipLaunchKernelGGL((slice_pack_kernel<T, true>), dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, in, pitch, C / EPC,
                       Cp / EPC, nch, scale, shift, out);
  else
    hipLaunchKernelGGL((slice_pack_kernel<T, false>), dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, in, pitch, C / EPC,
                       Cp / EPC, nch, scale, shift, out);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void slice_scatter_kernel(const T* __restrict__ src, int srcC, int CPR,
                                                                 T* __restrict__ dst, int pitch, size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const size_t r = i / CPR;
    const int cc = (int)(i - r * CPR);
    *reinterpret_cast<uint4*>(dst + r * pitch + (size_t)cc * EPC) =
        *reinterpret_cast<const uint4*>(src + r * srcC + (size_t)cc * EPC);
  }
}
template <typename T>
int slice_scatter(const T* src, int srcC, int C, T* dst, int pitch, size_t rows, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && srcC % EPC == 0 && pitch % EPC == 0 && C <= srcC, "slice_scatter: C=%d srcC=%d pitch=%d", C, srcC, pitch);
  const size_t nch = rows * (C / EPC);
  hipLaunchKernelGGL(slice_scatter_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, src, srcC, C / EPC, dst, pitch, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void slice_bn_bwd_accumulate_kernel(
    T* __restrict__ dcat, const T* __restrict__ x, int pitch, int CPR, const T* __restrict__ dz, int Cp,
    const float* __restrict__ cA, const float* __restrict__ cB, const float* __restrict__ cC, size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const size_t r = i / CPR;
    const int cc = (int)(i - r * CPR), c0 = cc * EPC;
    const size_t off = r * pitch + c0;
    Chunk<T> g, xv, dv;
    g.load(dcat + off);
    xv.load(x + off);
    dv.load(dz + r * Cp + c0);
#pragma unroll
    for (int e = 0; e < EPC; ++e) g.v[e] += cA[c0 + e] * dv.v[e] + cB[c0 + e] * xv.v[e] + cC[c0 + e];
    g.store(dcat + off);
  }
}
template <typename T>
int slice_bn_bwd_accumulate(T* dcat, const T* x, int pitch, int C, const T* dz, int Cp, const float* cA,
                            const float* cB, const float* cC, size_t rows, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && Cp % EPC == 0 && pitch % EPC == 0 && C <= Cp && C <= pitch, "slice_bn_bwd_accumulate: C=%d Cp=%d pitch=%d", C, Cp, pitch);
  const size_t nch = rows * (C / EPC);
  hipLaunchKernelGGL(slice_bn_bwd_accumulate_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, dcat, x, pitch,
                     C / EPC, dz, Cp, cA, cB, cC, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// Deferred form of the accumulation above (DenseNet: every later layer of a block adds cA*g + cB*x + cC to the prefix it consumed;
// x is the SAME for all of them, so sum(cB) * x + sum(cC) is added once, when a channel's gradient is consumed):
//   slice_accumulate_scaled : dcat[r*pitch + c] += cA[c] * dz[r*Cp + c]                        (no read of x: 3 passes instead of 4)
//   slice_pack_deferred     : out[r][c < C] = d[r*pitch + c] + sB[c] * x[r*pitch + c] + sC[c], zero for C <= c < Cp
//   slice_affine_inplace    : d[r*pitch + c] += sB[c] * x[r*pitch + c] + sC[c]                  for c < C
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void slice_accumulate_scaled_kernel(T* __restrict__ dcat, int pitch, unsigned CPR, const T* __restrict__ dz,
                                                                          int Cp, const float* __restrict__ cA, size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const unsigned r = (unsigned)(i / CPR);
    const int c0 = (int)(i - (size_t)r * CPR) * EPC;
    const size_t off = (size_t)r * pitch + c0;
    Chunk<T> g, dv;
    g.load(dcat + off);
    dv.load(dz + (size_t)r * Cp + c0);
#pragma unroll
    for (int e = 0; e < EPC; ++e) g.v[e] += cA[c0 + e] * dv.v[e];
    g.store(dcat + off);
  }
}
template <typename T>
int slice_accumulate_scaled(T* dcat, int pitch, int C, const T* dz, int Cp, const float* cA, size_t rows, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && Cp % EPC == 0 && pitch % EPC == 0 && C <= Cp && C <= pitch && rows < ((size_t)1 << 32), "slice_accumulate_scaled: C=%d Cp=%d pitch=%d", C, Cp, pitch);
  const size_t nch = rows * (C / EPC);
  hipLaunchKernelGGL(slice_accumulate_scaled_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, dcat, pitch, (unsigned)(C / EPC), dz, Cp, cA, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template <typename T, bool INPLACE>
__global__ __launch_bounds__(EW_BLOCK) void slice_deferred_kernel(T* __restrict__ d, const T* __restrict__ x, int pitch, unsigned CPRin, unsigned CPRout,
                                                                 const float* __restrict__ sB, const float* __restrict__ sC, T* __restrict__ out,
                                                                 size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const unsigned r = (unsigned)(i / CPRout);
    const unsigned cc = (unsigned)(i - (size_t)r * CPRout);
    const int c0 = (int)cc * EPC;
    Chunk<T> v;
    if (cc < CPRin) {
      Chunk<T> xv;
      v.load(d + (size_t)r * pitch + c0);
      xv.load(x + (size_t)r * pitch + c0);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] += sB[c0 + e] * xv.v[e] + sC[c0 + e];
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = 0.f;
    }
    if (INPLACE) v.store(d + (size_t)r * pitch + c0);
    else v.store(out + i * EPC);
  }
}
template <typename T>
int slice_pack_deferred(const T* d, const T* x, int pitch, int C, int Cp, size_t rows, const float* sB, const float* sC, T* out, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && Cp % EPC == 0 && pitch % EPC == 0 && C <= Cp && rows < ((size_t)1 << 32), "slice_pack_deferred: C=%d Cp=%d pitch=%d", C, Cp, pitch);
  const size_t nch = rows * (Cp / EPC);
  hipLaunchKernelGGL((slice_deferred_kernel<T, false>), dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, const_cast<T*>(d), x, pitch, (unsigned)(C / EPC),
                     (unsigned)(Cp / EPC), sB, sC, out, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template <typename T>
int slice_affine_inplace(T* d, const T* x, int pitch, int C, size_t rows, const float* sB, const float* sC, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && pitch % EPC == 0 && C <= pitch && rows < ((size_t)1 << 32), "slice_affine_inplace: C=%d pitch=%d", C, pitch);
  const size_t nch = rows * (C / EPC);
  hipLaunchKernelGGL((slice_deferred_kernel<T, true>), dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, d, x, pitch, (unsigned)(C / EPC), (unsigned)(C / EPC),
                     sB, sC, (T*)nullptr, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void slice_stats_kernel(const T* __restrict__ x, int pitch, size_t rows, int C,
                                                          ColGeom g, float* partial_sum, float* partial_sq) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[2 * 256 * EPC];
  const int cx = threadIdx.x % g.CW, ry = threadIdx.x / g.CW;
  const int col = blockIdx.y * g.CW + cx;
  float acc[2][EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  if (ry < g.RL && col < g.CPR) {
    size_t r_end = (size_t)(blockIdx.x + 1) * g.RB;
    if (r_end > rows) r_end = rows;
    for (size_t r = (size_t)blockIdx.x * g.RB + ry; r < r_end; r += g.RL) {
      Chunk<T> v;
      v.load(x + r * pitch + (size_t)col * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) { acc[0][e] += v.v[e]; acc[1][e] += v.v[e] * v.v[e]; }
    }
  }
  block_col_reduce<EPC, 2>(acc, cx, ry, g.CW, g.RL, col, g.CPR, C, partial_sum, red);
}
// NOTE: partial layout here is the interleaved [row][2][C] of block_col_reduce<.,2>; stat_sum points at
// it and stat_sq is unused by the kernel -- bn_table_finalize is told through stride / offsets.
template <typename T>
int slice_stats(const T* x, int pitch, int C, size_t rows, float* stat_sum, float* stat_sq, int* nrows_out,
                hipStream_t st) {
  ARG_CHECK(C % DT<T>::EPC == 0 && pitch % DT<T>::EPC == 0, "slice_stats: C=%d pitch=%d", C, pitch);
  ARG_CHECK(stat_sq == stat_sum + C, "slice_stats: stat_sq must be stat_sum + C (interleaved [row][2][C] slab)");
  ColGeom g = col_geom(rows, C, DT<T>::EPC);
  hipLaunchKernelGGL(slice_stats_kernel<T>, dim3(g.gx, g.gy), dim3(256), 0, st, x, pitch, rows, C, g, stat_sum, stat_sq);
  HIP_CHECK_RET(hipGetLastError());
  *nrows_out = g.gx;
  return MMSKIN_OK;
}

template <typename IN>
__global__ __launch_bounds__(1024) void bn_table_finalize_kernel(const IN* __restrict__ ssum, const IN* __restrict__ ssq, int nrows, int stride,
                                         int C, double count, float* __restrict__ mean, float* __restrict__ var) {
  // 64 columns x 16 row lanes (a DenseNet growth slice has 32 channels: with 4 lanes one workgroup walked up to 512 rows in 18 us)
  __shared__ double red[2][16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  double s = 0.0, q = 0.0;
  if (c < C)
    for (int r = ry; r < nrows; r += 16) { s += (double)ssum[(size_t)r * stride + c]; q += (double)ssq[(size_t)r * stride + c]; }
  red[0][ry][cx] = s; red[1][ry][cx] = q;
  __syncthreads();
  if (ry == 0 && c < C) {
    s = 0.0; q = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { s += red[0][k][cx]; q += red[1][k][cx]; }
    double m = s / count, v = q / count - m * m;
    mean[c] = (float)m;
    var[c] = (float)(v < 0.0 ? 0.0 : v);
  }
}
int bn_table_finalize(const float* stat_sum, const float* stat_sq, int nrows, int stride, int C, double count,
                      float* mean, float* var, double* scratch, hipStream_t st) {
  const bool interleaved = stat_sq == stat_sum + C && stride == 2 * C;   // slice_stats slab
  if (scratch && nrows > BN_SINGLE_STAGE_ROWS) {
    const int G = reduce_groups(nrows);
    int rc;
    if (interleaved) {
      if ((rc = partial_reduce<double>(stat_sum, nullptr, nrows, stride, G, scratch, st))) return rc;
      hipLaunchKernelGGL(bn_table_finalize_kernel<double>, dim3(ceil_div(C, 64)), dim3(1024), 0, st, scratch, scratch + C, G,
                         stride, C, count, mean, var);
    } else {
      if ((rc = partial_reduce<double>(stat_sum, stat_sq, nrows, stride, G, scratch, st))) return rc;
      hipLaunchKernelGGL(bn_table_finalize_kernel<double>, dim3(ceil_div(C, 64)), dim3(1024), 0, st, scratch,
                         scratch + (size_t)G * stride, G, stride, C, count, mean, var);
    }
  } else {
    hipLaunchKernelGGL(bn_table_finalize_kernel<float>, dim3(ceil_div(C, 64)), dim3(1024), 0, st, stat_sum, stat_sq, nrows,
                       stride, C, count, mean, var);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

__global__ void bn_coef_from_table_kernel(const float* __restrict__ mean_tab, const float* __restrict__ var_tab, int C,
                                          int Cp, const float* __restrict__ gamma, const float* __restrict__ beta,
                                          float eps, float momentum, double count, float* running_mean,
                                          float* running_var, int training, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f, g = 0.f;
  if (c < C) {
    g = gamma[c];
    float var;
    if (training) {
      mu = mean_tab[c]; var = var_tab[c];
      double unbiased = count > 1.0 ? (double)var * count / (count - 1.0) : (double)var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    } else {
      mu = running_mean[c]; var = running_var[c];
    }
    is = (float)(1.0 / sqrt((double)var + (double)eps));
    sc = g * is;
    sh = beta[c] - mu * sc;
  }
  coef[c] = sc; coef[Cp + c] = sh; coef[2 * Cp + c] = mu; coef[3 * Cp + c] = is; coef[4 * Cp + c] = g;
}
int bn_coef_from_table(const float* mean_tab, const float* var_tab, int C, int Cp, const float* gamma,
                       const float* beta, float eps, float momentum, double count, float* running_mean,
                       float* running_var, bool training, float* coef, hipStream_t st) {
  hipLaunchKernelGGL(bn_coef_from_table_kernel, dim3(ceil_div(Cp, 256)), dim3(256), 0, st, mean_tab, var_tab, C, Cp,
                     gamma, beta, eps, momentum, count, running_mean, running_var, training ? 1 : 0, coef);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void avgpool2_fwd_kernel(const T* __restrict__ x, int H, int W, int CPR,
                                                                T* __restrict__ dst, int pitch, size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  const int OH = H / 2, OW = W / 2;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const int cc = (int)(i % CPR);
    size_t t = i / CPR;          // output pixel (n, oy, ox)
    const int ox = (int)(t % OW); size_t t2 = t / OW;
    const int oy = (int)(t2 % OH);
    const size_t n = t2 / OH;
    const size_t C = (size_t)CPR * EPC;
    const T* p00 = x + (((n * H + 2 * oy) * W) + 2 * ox) * C + (size_t)cc * EPC;
    Chunk<T> a, b, c, d;
    a.load(p00); b.load(p00 + C); c.load(p00 + (size_t)W * C); d.load(p00 + (size_t)W * C + C);
#pragma unroll
    for (int e = 0; e < EPC; ++e) a.v[e] = 0.25f * ((a.v[e] + b.v[e]) + (c.v[e] + d.v[e]));
    a.store(dst + t * pitch + (size_t)cc * EPC);
  }
}
template <typename T>
int avgpool2_fwd(const T* x, int N, int H, int W, int C, T* dst, int pitch, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && pitch % EPC == 0 && H >= 2 && W >= 2, "avgpool2_fwd: C=%d pitch=%d %dx%d", C, pitch, H, W);
  const size_t nch = (size_t)N * (H / 2) * (W / 2) * (C / EPC);
  hipLaunchKernelGGL(avgpool2_fwd_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, x, H, W, C / EPC, dst, pitch, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void avgpool2_bwd_kernel(const T* __restrict__ dpool, int pitch, int H, int W,
                                                                int CPR, T* __restrict__ dx, size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  const int OH = H / 2, OW = W / 2;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const int cc = (int)(i % CPR);
    size_t t = i / CPR;          // input pixel (n, y, x)
    const int xx = (int)(t % W); size_t t2 = t / W;
    const int yy = (int)(t2 % H);
    const size_t n = t2 / H;
    Chunk<T> g;
    if ((yy >> 1) < OH && (xx >> 1) < OW) {   // odd H/W: the last row/column is outside every window
      g.load(dpool + ((n * OH + (yy >> 1)) * OW + (xx >> 1)) * pitch + (size_t)cc * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) g.v[e] *= 0.25f;
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) g.v[e] = 0.f;
    }
    g.store(dx + i * EPC);
  }
}
template <typename T>
int avgpool2_bwd(const T* dpool, int pitch, int N, int H, int W, int C, T* dx, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && pitch % EPC == 0, "avgpool2_bwd: C=%d pitch=%d", C, pitch);
  const size_t nch = (size_t)N * H * W * (C / EPC);
  hipLaunchKernelGGL(avgpool2_bwd_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, dpool, pitch, H, W, C / EPC, dx, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

#define INST_SLICE(T)                                                                                              \
  template int slice_pack<T>(const T*, int, int, int, size_t, const float*, const float*, T*, hipStream_t);       \
  template int slice_scatter<T>(const T*, int, int, T*, int, size_t, hipStream_t);                                \
  template int slice_bn_bwd_accumulate<T>(T*, const T*, int, int, const T*, int, const float*, const float*,      \
                                          const float*, size_t, hipStream_t);                                     \
  template int slice_accumulate_scaled<T>(T*, int, int, const T*, int, const float*, size_t, hipStream_t);       \
  template int slice_pack_deferred<T>(const T*, const T*, int, int, int, size_t, const float*, const float*, T*, hipStream_t); \
  template int slice_affine_inplace<T>(T*, const T*, int, int, size_t, const float*, const float*, hipStream_t); \
  template int slice_stats<T>(const T*, int, int, size_t, float*, float*, int*, hipStream_t);                     \
  template int avgpool2_fwd<T>(const T*, int, int, int, int, T*, int, hipStream_t);                               \
  template int avgpool2_bwd<T>(const T*, int, int, int, int, int, T*, hipStream_t);
INST_SLICE(float)
INST_SLICE(bf16_t)

// ------------------------------------------------------------------ VGG pieces
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void pack_nhwc8_kernel(const void* __restrict__ img, int u8, Norm6 nm, int N, int H,
                                                             int W, int Hp, int Wp, T* __restrict__ out) {
  const size_t total = (size_t)N * Hp * Wp;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < total; i += (size_t)gridDim.x * EW_BLOCK) {
    int wp = (int)(i % Wp);
    size_t t = i / Wp;
    int hp = (int)(t % Hp), n = (int)(t / Hp);
    int h = hp - 1, w = wp - 1;
    float v[3] = {0.f, 0.f, 0.f};
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
      if (u8) {
        const uint8_t* px = reinterpret_cast<const uint8_t*>(img) + (((size_t)n * H + h) * W + w) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (float)px[c] * nm.a[c] + nm.b[c];
      } else {
        const float* f = reinterpret_cast<const float*>(img);
        size_t base = ((size_t)n * 3 * H + h) * W + w;
        v[0] = f[base]; v[1] = f[base + (size_t)H * W]; v[2] = f[base + 2 * (size_t)H * W];
      }
    }
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(out + i * 8) = make_float4(v[0], v[1], v[2], 0.f);
      *reinterpret_cast<float4*>(out + i * 8 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      uint32_t lo = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
      uint32_t hi = f32_to_bf16_bits(v[2]);
      *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(lo, hi, 0u, 0u);
    }
  }
}
template <typename T>
int pack_nhwc8(const void* img, const float* norm6, int N, int H, int W, int Hp, int Wp, T* out, hipStream_t st) {
  ARG_CHECK(Hp >= H + 2 && Wp >= W + 4, "pack_nhwc8: bad padded size");
  Norm6 nm = {};
  if (norm6)
    for (int c = 0; c < 3; ++c) {
      ARG_CHECK(norm6[3 + c] > 0.f, "pack_nhwc8: std[%d] = %f", c, norm6[3 + c]);
      nm.a[c] = 1.f / (255.f * norm6[3 + c]);
      nm.b[c] = -norm6[c] / norm6[3 + c];
    }
  hipLaunchKernelGGL(pack_nhwc8_kernel<T>, dim3(ew_grid((size_t)N * Hp * Wp)), dim3(EW_BLOCK), 0, st, img, norm6 ? 1 : 0, nm, N,
                     H, W, Hp, Wp, out);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ void vgg_stage_first_kernel(const float* __restrict__ w, T* __restrict__ wv, int cout) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;   // over cout*3*3*3 (o, c, r, s)
  if (i < cout * 27) {
    int o = i / 27, rem = i - o * 27;
    int c = rem / 9, t = rem - c * 9;
    int r = t / 3, s2 = t - r * 3;
    wv[((size_t)o * 4 + r) * 32 + s2 * 8 + c] = from_f32<T>(w[i]);
  }
}
template <typename T>
int vgg_stage_first(const float* w, T* wv, hipStream_t st, int cout) {
  HIP_CHECK_RET(hipMemsetAsync(wv, 0, 64 * 128 * sizeof(T), st));
  hipLaunchKernelGGL(vgg_stage_first_kernel<T>, dim3(ceil_div(64 * 27, 256)), dim3(256), 0, st, w, wv, cout);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
__global__ void vgg_wgrad_unpack_first_kernel(const float* __restrict__ dwv, float* __restrict__ dw, int cout) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout * 27) {
    int o = i / 27, rem = i - o * 27;
    int c = rem / 9, t = rem - c * 9;
    int r = t / 3, s2 = t - r * 3;
    dw[i] = dwv[((size_t)o * 4 + r) * 32 + s2 * 8 + c];
  }
}
int vgg_wgrad_unpack_first(const float* dwv, float* dw, hipStream_t st, int cout) {
  hipLaunchKernelGGL(vgg_wgrad_unpack_first_kernel, dim3(ceil_div(64 * 27, 256)), dim3(256), 0, st, dwv, dw, cout);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void maxpool2_fwd_kernel(const T* __restrict__ x, int H, int W, int CPR,
                                                               T* __restrict__ y, uint8_t* __restrict__ idx, size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  const int PH = H / 2, PW = W / 2;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const int cc = (int)(i % CPR);
    size_t t = i / CPR;
    const int pw = (int)(t % PW); size_t t2 = t / PW;
    const int ph = (int)(t2 % PH);
    const size_t n = t2 / PH;
    const size_t C = (size_t)CPR * EPC;
    const T* p00 = x + (((n * H + 2 * ph) * W) + 2 * pw) * C + (size_t)cc * EPC;
    Chunk<T> v[4];
    v[0].load(p00); v[1].load(p00 + C); v[2].load(p00 + (size_t)W * C); v[3].load(p00 + (size_t)W * C + C);
    Chunk<T> best = v[0];
    uint8_t bi[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) bi[e] = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < EPC; ++e)
        if (v[k].v[e] > best.v[e] || v[k].v[e] != v[k].v[e]) { best.v[e] = v[k].v[e]; bi[e] = (uint8_t)k; }
    best.store(y + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) idx[i * EPC + e] = bi[e];
  }
}
template <typename T>
int maxpool2_fwd(const T* x, int N, int H, int W, int C, T* y, uint8_t* idx, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && H >= 2 && W >= 2, "maxpool2_fwd: C=%d %dx%d", C, H, W);
  const size_t nch = (size_t)N * (H / 2) * (W / 2) * (C / EPC);
  hipLaunchKernelGGL(maxpool2_fwd_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, x, H, W, C / EPC, y, idx, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void maxpool2_bwd_relu_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                    const T* __restrict__ y, int H, int W, int CPR,
                                                                    T* __restrict__ dz, size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  const int PH = H / 2, PW = W / 2;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const int cc = (int)(i % CPR);
    size_t t = i / CPR;          // input pixel (n, h, w)
    const int w = (int)(t % W); size_t t2 = t / W;
    const int h = (int)(t2 % H);
    const size_t n = t2 / H;
    Chunk<T> out;
#pragma unroll
    for (int e = 0; e < EPC; ++e) out.v[e] = 0.f;
    const int ph = h >> 1, pw = w >> 1;
    if (ph < PH && pw < PW) {    // odd H/W: the last row / column is outside every window
      const size_t po = (((n * PH + ph) * PW + pw) * CPR + cc) * EPC;
      const int tap = (h & 1) * 2 + (w & 1);
      Chunk<T> g, yv;
      g.load(dpool + po);
      yv.load(y + i * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e)
        if (idx[po + e] == tap && yv.v[e] > 0.f) out.v[e] = g.v[e];
    }
    out.store(dz + i * EPC);
  }
}
template <typename T>
int maxpool2_bwd_relu(const T* dpool, const uint8_t* idx, const T* y, int N, int H, int W, int C, T* dz, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0, "maxpool2_bwd_relu: C=%d", C);
  const size_t nch = (size_t)N * H * W * (C / EPC);
  hipLaunchKernelGGL(maxpool2_bwd_relu_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, dpool, idx, y, H, W, C / EPC, dz, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// adaptive bins as torch: [floor(i*H/OH), ceil((i+1)*H/OH))
template <typename T>
__global__ void adaptive_avgpool_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, int OH, int OW,
                                            float* __restrict__ out) {
  const size_t total = (size_t)N * C * OH * OW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // thread order follows the NHWC source (channel fastest) for coalesced reads
    int c = (int)(i % C); size_t t = i / C;
    int ow = (int)(t % OW); t /= OW;
    int oh = (int)(t % OH);
    int n = (int)(t / OH);
    int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH;
    int w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
    float s = 0.f;
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) s += to_f32(x[(((size_t)n * H + h) * W + w) * C + c]);
    out[(((size_t)n * C + c) * OH + oh) * OW + ow] = s / (float)((h1 - h0) * (w1 - w0));
  }
}
template <typename T>
int adaptive_avgpool_fwd(const T* x, int N, int H, int W, int C, int OH, int OW, float* out_nchw, hipStream_t st) {
  hipLaunchKernelGGL(adaptive_avgpool_fwd_kernel<T>, dim3(ew_grid((size_t)N * C * OH * OW)), dim3(256), 0, st, x, N, H, W, C, OH,
                     OW, out_nchw);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template <typename T>
__global__ void adaptive_avgpool_bwd_kernel(const float* __restrict__ dout, int N, int H, int W, int C, int OH, int OW,
                                            T* __restrict__ dx) {
  const size_t total = (size_t)N * H * W * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); size_t t = i / C;
    int w = (int)(t % W); t /= W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    float s = 0.f;
    // every output bin whose window contains (h, w)
    for (int oh = (h * OH) / H; oh < OH && (oh * H) / OH <= h; ++oh) {
      int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH;
      if (h < h0 || h >= h1) continue;
      for (int ow = (w * OW) / W; ow < OW && (ow * W) / OW <= w; ++ow) {
        int w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
        if (w < w0 || w >= w1) continue;
        s += dout[(((size_t)n * C + c) * OH + oh) * OW + ow] / (float)((h1 - h0) * (w1 - w0));
      }
    }
    dx[i] = from_f32<T>(s);
  }
}
template <typename T>
int adaptive_avgpool_bwd(const float* dout_nchw, int N, int H, int W, int C, int OH, int OW, T* dx, hipStream_t st) {
  hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel<T>, dim3(ew_grid((size_t)N * H * W * C)), dim3(256), 0, st, dout_nchw, N, H, W, C,
                     OH, OW, dx);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename IN>
__global__ void bias_grad_finalize_kernel(const IN* __restrict__ part, int nrows, int stride, int C, float* __restrict__ db) {
  __shared__ double red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  double s = 0.0;
  if (c < C)
    for (int r = ry; r < nrows; r += 4) s += (double)part[(size_t)r * stride + c];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && c < C) db[c] = (float)((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]));
}
int bias_grad_finalize(const float* partial, int nrows, int stride, int C, float* db, double* scratch, hipStream_t st) {
  if (scratch && nrows > BN_SINGLE_STAGE_ROWS) {
    const int G = reduce_groups(nrows);
    int rc = partial_reduce<double>(partial, nullptr, nrows, stride, G, scratch, st);
    if (rc) return rc;
    hipLaunchKernelGGL(bias_grad_finalize_kernel<double>, dim3(ceil_div(C, 64)), dim3(256), 0, st, scratch, G, stride, C, db);
  } else {
    hipLaunchKernelGGL(bias_grad_finalize_kernel<float>, dim3(ceil_div(C, 64)), dim3(256), 0, st, partial, nrows, stride, C, db);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

#define INST_VGG(T)                                                                                         \
  template int pack_nhwc8<T>(const void*, const float*, int, int, int, int, int, T*, hipStream_t);          \
  template int vgg_stage_first<T>(const float*, T*, hipStream_t, int);                                      \
  template int maxpool2_fwd<T>(const T*, int, int, int, int, T*, uint8_t*, hipStream_t);                    \
  template int maxpool2_bwd_relu<T>(const T*, const uint8_t*, const T*, int, int, int, int, T*, hipStream_t); \
  template int adaptive_avgpool_fwd<T>(const T*, int, int, int, int, int, int, float*, hipStream_t);        \
  template int adaptive_avgpool_bwd<T>(const float*, int, int, int, int, int, int, T*, hipStream_t);
INST_VGG(float)
INST_VGG(bf16_t)

template <typename T>
__global__ void gap_relu_bn_grad_kernel(const float* __restrict__ dfeat, const T* __restrict__ y, const float* __restrict__ scale,
                                        int N, int HW, int C, float* __restrict__ dx) {
  const size_t total = (size_t)N * HW * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); size_t t = i / C;      // NHWC source order
    int hw = (int)(t % HW);
    int n = (int)(t / HW);
    float g = to_f32(y[i]) > 0.f ? dfeat[(size_t)n * C + c] * scale[c] / (float)HW : 0.f;
    dx[((size_t)n * C + c) * HW + hw] = g;
  }
}
template <typename T>
int gap_relu_bn_grad(const float* dfeat, const T* y, const float* scale, int N, int HW, int C, float* dx_nchw, hipStream_t st) {
  hipLaunchKernelGGL(gap_relu_bn_grad_kernel<T>, dim3(ew_grid((size_t)N * HW * C)), dim3(256), 0, st, dfeat, y, scale, N, HW, C, dx_nchw);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template int gap_relu_bn_grad<float>(const float*, const float*, const float*, int, int, int, float*, hipStream_t);
template int gap_relu_bn_grad<bf16_t>(const float*, const bf16_t*, const float*, int, int, int, float*, hipStream_t);

// ------------------------------------------------------------------ depthwise k x k (k = 3 MobileNet / 3, 5 EfficientNet)
template <typename T>
__global__ void dw_stage_weights_kernel(const float* __restrict__ w, int C, int Cp, int KK, T* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;   // over KK * Cp: (tap, c)
  if (i < KK * Cp) {
    int t = i / Cp, c = i - t * Cp;
    out[i] = from_f32<T>(c < C ? w[(size_t)c * KK + t] : 0.f);
  }
}
template <typename T>
int dw_stage_weights(const float* w_oihw, int C, int Cp, T* w_tc, hipStream_t st, int ksize) {
  const int KK = ksize * ksize;
  hipLaunchKernelGGL(dw_stage_weights_kernel<T>, dim3(ceil_div(KK * Cp, 256)), dim3(256), 0, st, w_oihw, C, Cp, KK, w_tc);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void dwconv3_fwd_kernel(const T* __restrict__ in, const T* __restrict__ w, int H, int W,
                                                              int CPR, int stride, int K, int OH, int OW, T* __restrict__ out,
                                                              size_t nchunks, const float* __restrict__ bias, int residual) {
  constexpr int EPC = DT<T>::EPC;
  const size_t C = (size_t)CPR * EPC;
  const int pad = K / 2;
  // XCD-aware block order: consecutive workgroups are dealt to different XCDs (each with its own L2), so with the plain order the
  // rows above / below a pixel were fetched from HBM by three XCDs (measured: 4.2 bytes read per byte written,
  // profiles/r03_step_traffic_davit-tiny-gfcam.txt); xcd_remap gives every XCD one contiguous run of the image.
  const size_t bid = (size_t)xcd_remap((int)blockIdx.x, (int)gridDim.x);
  for (size_t i = bid * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const int cc = (int)(i % CPR);
    size_t t = i / CPR;
    const int ox = (int)(t % OW); size_t t2 = t / OW;
    const int oy = (int)(t2 % OH);
    const size_t n = t2 / OH;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int r = 0; r < K; ++r) {
      const int iy = oy * stride - pad + r;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int q = 0; q < K; ++q) {
        const int ix = ox * stride - pad + q;
        if ((unsigned)ix >= (unsigned)W) continue;
        Chunk<T> v, wv;
        v.load(in + ((n * H + iy) * W + ix) * C + (size_t)cc * EPC);
        wv.load(w + (size_t)(r * K + q) * C + (size_t)cc * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += v.v[e] * wv.v[e];
      }
    }
    if (bias) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += bias[(size_t)cc * EPC + e];
    }
    if (residual) {   // y = x + conv(x) + b (stride 1: output pixel = input pixel)
      Chunk<T> v;
      v.load(in + i * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += v.v[e];
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = acc[e];
    o.store(out + i * EPC);
  }
}
template <typename T>
int dwconv3_fwd(const T* in, const T* w_tc, int N, int H, int W, int C, int stride, T* out, hipStream_t st, int ksize, const float* bias,
                bool residual) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && (stride == 1 || stride == 2) && (ksize == 3 || ksize == 5), "dwconv_fwd: C=%d stride=%d k=%d", C, stride, ksize);
  const int pad = ksize / 2;
  const int OH = (H + 2 * pad - ksize) / stride + 1, OW = (W + 2 * pad - ksize) / stride + 1;
  const size_t nch = (size_t)N * OH * OW * (C / EPC);
  ARG_CHECK(!residual || stride == 1, "dwconv_fwd: the residual form needs stride 1");
  hipLaunchKernelGGL(dwconv3_fwd_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, in, w_tc, H, W, C / EPC, stride, ksize, OH, OW, out, nch,
                     bias, residual ? 1 : 0);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void dwconv3_dgrad_kernel(const T* __restrict__ dout, const T* __restrict__ w, int H, int W,
                                                                int CPR, int stride, int K, int OH, int OW, T* __restrict__ din,
                                                                size_t nchunks, int residual) {
  constexpr int EPC = DT<T>::EPC;
  const size_t C = (size_t)CPR * EPC;
  const int pad = K / 2;
  const size_t bid = (size_t)xcd_remap((int)blockIdx.x, (int)gridDim.x);   // as the forward: one contiguous run of the image per XCD
  for (size_t i = bid * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const int cc = (int)(i % CPR);
    size_t t = i / CPR;
    const int ix = (int)(t % W); size_t t2 = t / W;
    const int iy = (int)(t2 % H);
    const size_t n = t2 / H;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int r = 0; r < K; ++r) {
      const int ty = iy + pad - r;
      if (ty < 0 || ty % stride != 0) continue;
      const int oy = ty / stride;
      if (oy >= OH) continue;
      for (int q = 0; q < K; ++q) {
        const int tx = ix + pad - q;
        if (tx < 0 || tx % stride != 0) continue;
        const int ox = tx / stride;
        if (ox >= OW) continue;
        Chunk<T> g, wv;
        g.load(dout + ((n * OH + oy) * OW + ox) * C + (size_t)cc * EPC);
        wv.load(w + (size_t)(r * K + q) * C + (size_t)cc * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += g.v[e] * wv.v[e];
      }
    }
    if (residual) {   // d/dx of x + conv(x): the gradient passes through as well
      Chunk<T> g;
      g.load(dout + i * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += g.v[e];
    }
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = acc[e];
    o.store(din + i * EPC);
  }
}
template <typename T>
int dwconv3_dgrad(const T* dout, const T* w_tc, int N, int H, int W, int C, int stride, T* din, hipStream_t st, int ksize, bool residual) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && (stride == 1 || stride == 2) && (ksize == 3 || ksize == 5), "dwconv_dgrad: C=%d stride=%d k=%d", C, stride, ksize);
  const int pad = ksize / 2;
  const int OH = (H + 2 * pad - ksize) / stride + 1, OW = (W + 2 * pad - ksize) / stride + 1;
  const size_t nch = (size_t)N * H * W * (C / EPC);
  ARG_CHECK(!residual || stride == 1, "dwconv_dgrad: the residual form needs stride 1");
  hipLaunchKernelGGL(dwconv3_dgrad_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, dout, w_tc, H, W, C / EPC, stride, ksize, OH, OW, din, nch,
                     residual ? 1 : 0);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// weight gradient: blockIdx.x = strip of output pixels, blockIdx.y = group of 8 chunk columns, blockIdx.z = kernel row;
// 32 pixel lanes per block (channel counts here are 64..1152 wide: with 32 columns per block most lanes of the narrow
// early layers sat idle and one launch took 1 ms).  One kernel row (<= 5 taps) per block keeps the accumulators in registers.
#define DWW_COLS 8
#define DWW_LANES 32
static inline int dww_strips(size_t opix) {
  size_t s = (opix + 2047) / 2048;
  if (s < 1) s = 1;
  if (s > 4096) s = 4096;
  return (int)s;
}
// 3x3 / stride 1 (the DaViT position encodings, the stride-1 depthwise layers of MobileNet / EfficientNet): one thread owns one
// 16-byte channel chunk of whole image ROWS and walks them with a sliding 3x3 window in registers -- one load of dY and three of
// the input per pixel and chunk (the tapped kernel above: 3 + 9 through L1, two 64-bit divisions per pixel), all nine tap sums
// in registers, one LDS reduction over the block's row lanes at the end.
struct DwwRowsPlan { int CW, gy, lanes, rpl, nb; };
static DwwRowsPlan dww_rows_plan(int N, int H, int CPR) {
  DwwRowsPlan g;
  g.CW = CPR <= 32 ? CPR : (CPR % 32 == 0 ? 32 : (CPR % 24 == 0 ? 24 : 32));
  g.gy = (CPR + g.CW - 1) / g.CW;
  g.lanes = 256 / g.CW;
  const int NR = N * H;
  int want = 512 / g.gy;                      // ~512 workgroups per launch
  if (want < 1) want = 1;
  g.rpl = (NR + want * g.lanes - 1) / (want * g.lanes);
  if (g.rpl < 1) g.rpl = 1;
  g.nb = (NR + g.lanes * g.rpl - 1) / (g.lanes * g.rpl);
  return g;
}
size_t dwconv3_wgrad_partial_floats(int N, int H, int W, int C, int stride, int ksize) {
  const int pad = ksize / 2;
  const int OH = (H + 2 * pad - ksize) / stride + 1, OW = (W + 2 * pad - ksize) / stride + 1;
  size_t strips = (size_t)dww_strips((size_t)N * OH * OW);
  if (stride == 1 && ksize == 3) {            // the row-walking kernel's blocks (either element type)
    for (int epc = 4; epc <= 8; epc *= 2)
      if (C % epc == 0) { const size_t nb = (size_t)dww_rows_plan(N, H, C / epc).nb; if (nb > strips) strips = nb; }
  }
  return strips * (ksize * ksize + 1) * C;    // + one row per strip for the bias gradient of the fused position-encoding form
}
template <typename T>
__global__ __launch_bounds__(256) void dwconv3_wgrad_rows_kernel(const T* __restrict__ dout, const T* __restrict__ in, int H, int W,
                                                                int CPR, int CW, int lanes, int rpl, int NR,
                                                                float* __restrict__ partial, int with_bias) {
  constexpr int EPC = DT<T>::EPC;
  extern __shared__ float dww_red[];          // [lanes][CW * EPC]
  const int cx = threadIdx.x % CW, ly = threadIdx.x / CW;
  const int cc = blockIdx.y * CW + cx;
  const size_t C = (size_t)CPR * EPC;
  float acc[10][EPC];                         // 9 taps + sum of dY (bias gradient)
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[t][e] = 0.f;
  const int NT = with_bias ? 10 : 9;
  if (ly < lanes && cc < CPR) {
    for (int k = 0; k < rpl; ++k) {
      const int row = (blockIdx.x * rpl + k) * lanes + ly;      // = n * H + oy (stride 1, pad 1: the output has the input's geometry)
      if (row >= NR) break;
      const int oy = row % H;
      const T* drow = dout + (size_t)row * W * C + (size_t)cc * EPC;
      const T* irow = in + (size_t)row * W * C + (size_t)cc * EPC;
      const bool up = oy > 0, dn = oy + 1 < H;
      const ptrdiff_t rs = (ptrdiff_t)W * (ptrdiff_t)C;
      Chunk<T> a[3], b[3], c[3], g;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int e = 0; e < EPC; ++e) { a[r].v[e] = 0.f; b[r].v[e] = 0.f; c[r].v[e] = 0.f; }
      if (up) b[0].load(irow - rs);
      b[1].load(irow);
      if (dn) b[2].load(irow + rs);
      for (int ox = 0; ox < W; ++ox) {
        const bool right = ox + 1 < W;
        const T* pr = irow + (size_t)(ox + 1) * C;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int e = 0; e < EPC; ++e) c[r].v[e] = 0.f;
        if (right) {
          if (up) c[0].load(pr - rs);
          c[1].load(pr);
          if (dn) c[2].load(pr + rs);
        }
        g.load(drow + (size_t)ox * C);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            acc[r * 3 + 0][e] += g.v[e] * a[r].v[e];
            acc[r * 3 + 1][e] += g.v[e] * b[r].v[e];
            acc[r * 3 + 2][e] += g.v[e] * c[r].v[e];
          }
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[9][e] += g.v[e];
#pragma unroll
        for (int r = 0; r < 3; ++r) { a[r] = b[r]; b[r] = c[r]; }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 10; ++t) {              // reduce the row lanes, one tap at a time
    if (t >= NT) break;
    if (ly < lanes) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) dww_red[ly * CW * EPC + cx * EPC + e] = acc[t][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CW * EPC; i += 256) {
      const int ch = blockIdx.y * CW * EPC + i;
      if (ch < (int)C) {
        float s2 = 0.f;
        for (int l = 0; l < lanes; ++l) s2 += dww_red[l * CW * EPC + i];
        partial[((size_t)blockIdx.x * NT + t) * C + ch] = s2;
      }
    }
    __syncthreads();
  }
}
template <typename T>
__global__ __launch_bounds__(256) void dwconv3_wgrad_kernel(const T* __restrict__ dout, const T* __restrict__ in, int H, int W,
                                                           int CPR, int stride, int K, int OH, int OW, size_t opix,
                                                           size_t per_strip, float* __restrict__ partial) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[DWW_LANES][DWW_COLS * EPC];
  const size_t C = (size_t)CPR * EPC;
  const int pad = K / 2, r = blockIdx.z, KK = K * K;
  const int cx = threadIdx.x % DWW_COLS, ly = threadIdx.x / DWW_COLS;
  const int cc = blockIdx.y * DWW_COLS + cx;
  float acc[5][EPC];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[t][e] = 0.f;
  if (cc < CPR) {
    size_t p_end = ((size_t)blockIdx.x + 1) * per_strip;
    if (p_end > opix) p_end = opix;
    for (size_t p = (size_t)blockIdx.x * per_strip + ly; p < p_end; p += DWW_LANES) {
      const int ox = (int)(p % OW); size_t t2 = p / OW;
      const int oy = (int)(t2 % OH);
      const size_t n = t2 / OH;
      const int iy = oy * stride - pad + r;
      if ((unsigned)iy >= (unsigned)H) continue;
      Chunk<T> g;
      g.load(dout + p * C + (size_t)cc * EPC);
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        if (q >= K) break;
        const int ix = ox * stride - pad + q;
        if ((unsigned)ix >= (unsigned)W) continue;
        Chunk<T> v;
        v.load(in + ((n * H + iy) * W + ix) * C + (size_t)cc * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[q][e] += g.v[e] * v.v[e];
      }
    }
  }
  for (int q = 0; q < K; ++q) {   // reduce the pixel lanes, one tap at a time
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[ly][cx * EPC + e] = acc[q][e];
    __syncthreads();
    for (int i = threadIdx.x; i < DWW_COLS * EPC; i += 256) {
      const int ch = blockIdx.y * DWW_COLS * EPC + i;
      if (ch < (int)C) {
        float s2 = 0.f;
#pragma unroll
        for (int l = 0; l < DWW_LANES; ++l) s2 += red[l][i];
        partial[((size_t)blockIdx.x * KK + r * K + q) * C + ch] = s2;
      }
    }
    __syncthreads();
  }
}
__global__ void dwconv3_wgrad_finalize_kernel(const float* __restrict__ partial, int nstrips, int C, int Cv, int KK,
                                              float* __restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over KK * C: (tap, c)
  if (i >= KK * C) return;
  const int t = i / C, c = i - t * C;
  double s = 0.0;
  for (int k = 0; k < nstrips; ++k) s += (double)partial[(size_t)k * KK * C + i];
  if (c < Cv) dw[(size_t)c * KK + t] = (float)s;
}
// the same sum with 16 strip lanes per (tap, channel) column: the row-walking kernel leaves a few hundred partial rows
// NT = KK (+ 1: the last row of a strip is the bias gradient -> db)
__global__ __launch_bounds__(1024) void dwconv3_wgrad_finalize_lanes_kernel(const float* __restrict__ partial, int nstrips, int C, int Cv,
                                                                            int KK, int NT, float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double red[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cx;   // over NT * C: (tap, c)
  double s = 0.0;
  if (i < NT * C)
    for (int k = ry; k < nstrips; k += 16) s += (double)partial[(size_t)k * NT * C + i];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && i < NT * C) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cx];
    const int tp = i / C, c = i - tp * C;
    if (c < Cv) {
      if (tp < KK) dw[(size_t)c * KK + tp] = (float)t;
      else if (db) db[c] = (float)t;
    }
  }
}
template <typename T>
int dwconv3_wgrad(const T* dout, const T* in, int N, int H, int W, int C, int stride, float* partial, float* dw,
                  int Cv, hipStream_t st, int ksize, float* db) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0 && (stride == 1 || stride == 2) && (ksize == 3 || ksize == 5), "dwconv_wgrad: C=%d stride=%d k=%d", C, stride, ksize);
  const int pad = ksize / 2;
  const int OH = (H + 2 * pad - ksize) / stride + 1, OW = (W + 2 * pad - ksize) / stride + 1;
  const size_t opix = (size_t)N * OH * OW;
  const int strips = dww_strips(opix);
  const size_t per = (opix + strips - 1) / strips;
  const int CPR = C / EPC, KK = ksize * ksize;
  static const bool rows_on = [] { const char* v = getenv("MMSKIN_DWW_ROWS"); return !v || atoi(v) != 0; }();
  ARG_CHECK(!db || (stride == 1 && ksize == 3), "dwconv_wgrad: the bias gradient comes with the 3x3 / stride 1 kernel only");
  if ((rows_on || db) && stride == 1 && ksize == 3) {
    const DwwRowsPlan g = dww_rows_plan(N, H, CPR);
    const int NT = db ? 10 : 9;
    hipLaunchKernelGGL(dwconv3_wgrad_rows_kernel<T>, dim3(g.nb, g.gy), dim3(256), (size_t)g.lanes * g.CW * EPC * sizeof(float), st, dout, in,
                       H, W, CPR, g.CW, g.lanes, g.rpl, N * H, partial, db ? 1 : 0);
    hipLaunchKernelGGL(dwconv3_wgrad_finalize_lanes_kernel, dim3(ceil_div(NT * C, 64)), dim3(1024), 0, st, partial, g.nb, C, Cv, KK, NT, dw, db);
    HIP_CHECK_RET(hipGetLastError());
    return MMSKIN_OK;
  }
  hipLaunchKernelGGL(dwconv3_wgrad_kernel<T>, dim3(strips, ceil_div(CPR, DWW_COLS), ksize), dim3(256), 0, st, dout, in, H, W, CPR,
                     stride, ksize, OH, OW, opix, per, partial);
  hipLaunchKernelGGL(dwconv3_wgrad_finalize_kernel, dim3(ceil_div(KK * C, 256)), dim3(256), 0, st, partial, strips, C, Cv, KK, dw);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// ------------------------------------------------------------------ squeeze-excitation, stochastic depth (EfficientNet)
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void se_scale_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gate, int HW,
                                                               int CPR, T* __restrict__ y, size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const int cc = (int)(i % CPR);
    const size_t n = i / CPR / HW;
    const float* g = gate + (n * CPR + cc) * EPC;
    Chunk<T> v;
    v.load(x + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v.v[e] *= g[e];
    v.store(y + i * EPC);
  }
}
template <typename T>
int se_scale_fwd(const T* x, const float* gate, int N, int HW, int C, T* y, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0, "se_scale_fwd: C=%d", C);
  const size_t nch = (size_t)N * HW * (C / EPC);
  hipLaunchKernelGGL(se_scale_fwd_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, x, gate, HW, C / EPC, y, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template <typename T, bool PROD>
__global__ __launch_bounds__(256) void gap_reduce_kernel(const T* __restrict__ a, const T* __restrict__ b, int HW, int CPR,
                                                        float scale, float* __restrict__ out) {
  constexpr int EPC = DT<T>::EPC;
  __shared__ float red[32][8 * EPC];
  const int cx = threadIdx.x & 7, ly = threadIdx.x >> 3;   // 8 chunk columns x 32 pixel lanes
  const int cc = blockIdx.x * 8 + cx;
  const size_t n = blockIdx.y;
  float acc[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
  if (cc < CPR)
    for (int p = ly; p < HW; p += 32) {
      const size_t off = ((n * HW + p) * CPR + cc) * EPC;
      Chunk<T> u;
      u.load(a + off);
      if (PROD) {
        Chunk<T> v;
        v.load(b + off);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += u.v[e] * v.v[e];
      } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += u.v[e];
      }
    }
#pragma unroll
  for (int e = 0; e < EPC; ++e) red[ly][cx * EPC + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 8 * EPC) {
    const int ch = blockIdx.x * 8 * EPC + threadIdx.x;
    if (ch < CPR * EPC) {
      float s2 = 0.f;
#pragma unroll
      for (int l = 0; l < 32; ++l) s2 += red[l][threadIdx.x];
      out[n * CPR * EPC + ch] = s2 * scale;
    }
  }
}
template <typename T>
int gap_reduce(const T* a, const T* b, int N, int HW, int C, float scale, float* out, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0, "gap_reduce: C=%d", C);
  const int CPR = C / EPC;
  if (b) hipLaunchKernelGGL((gap_reduce_kernel<T, true>), dim3(ceil_div(CPR, 8), N), dim3(256), 0, st, a, b, HW, CPR, scale, out);
  else hipLaunchKernelGGL((gap_reduce_kernel<T, false>), dim3(ceil_div(CPR, 8), N), dim3(256), 0, st, a, b, HW, CPR, scale, out);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void se_dgate_kernel(const T* __restrict__ dy, const T* __restrict__ x, int N, int HW,
                                                           int CPR, float* __restrict__ dgate) {
  constexpr int EPC = DT<T>::EPC;
  const int total = N * CPR;
  for (int i = blockIdx.x * EW_BLOCK + threadIdx.x; i < total; i += gridDim.x * EW_BLOCK) {
    const int n = i / CPR, cc = i - n * CPR;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int p = 0; p < HW; ++p) {
      Chunk<T> a, b;
      const size_t off = (((size_t)n * HW + p) * CPR + cc) * EPC;
      a.load(dy + off); b.load(x + off);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += a.v[e] * b.v[e];
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) dgate[(size_t)i * EPC + e] = acc[e];
  }
}
template <typename T>
int se_dgate(const T* dy, const T* x, int N, int HW, int C, float* dgate, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0, "se_dgate: C=%d", C);
  hipLaunchKernelGGL(se_dgate_kernel<T>, dim3(ew_grid((size_t)N * (C / EPC))), dim3(EW_BLOCK), 0, st, dy, x, N, HW, C / EPC, dgate);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void se_dx_kernel(const T* __restrict__ dy, const float* __restrict__ gate,
                                                        const float* __restrict__ dpool, int HW, int CPR, T* __restrict__ dx,
                                                        size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  const float inv = 1.f / (float)HW;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const int cc = (int)(i % CPR);
    const size_t n = i / CPR / HW;
    const size_t go = (n * CPR + cc) * EPC;
    Chunk<T> v;
    v.load(dy + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v.v[e] = v.v[e] * gate[go + e] + dpool[go + e] * inv;
    v.store(dx + i * EPC);
  }
}
template <typename T>
int se_dx(const T* dy, const float* gate, const float* dpool, int N, int HW, int C, T* dx, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(C % EPC == 0, "se_dx: C=%d", C);
  const size_t nch = (size_t)N * HW * (C / EPC);
  hipLaunchKernelGGL(se_dx_kernel<T>, dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, dy, gate, dpool, HW, C / EPC, dx, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
__global__ void ew_act_fwd_kernel(const float* __restrict__ z, float* __restrict__ out, int64_t n, int mode) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float t = z[i], sg = 1.f / (1.f + expf(-t));
    out[i] = mode == 0 ? t * sg : sg;
  }
}
__global__ void ew_act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ z, float* __restrict__ dz, int64_t n,
                                  int mode) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float t = z[i], sg = 1.f / (1.f + expf(-t));
    dz[i] = dout[i] * (mode == 0 ? sg * (1.f + t * (1.f - sg)) : sg * (1.f - sg));
  }
}
int ew_act_fwd(const float* z, float* out, int64_t n, int mode, hipStream_t st) {
  hipLaunchKernelGGL(ew_act_fwd_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, st, z, out, n, mode);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int ew_act_bwd(const float* dout, const float* z, float* dz, int64_t n, int mode, hipStream_t st) {
  hipLaunchKernelGGL(ew_act_bwd_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, st, dout, z, dz, n, mode);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
__global__ void pad_matrix_kernel(const float* __restrict__ src, int rows, int cols, int rows_p, int cols_p, float* __restrict__ dst) {
  const int64_t total = (int64_t)rows_p * cols_p;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols_p), c = (int)(i - (int64_t)r * cols_p);
    dst[i] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
  }
}
int pad_matrix(const float* src, int rows, int cols, int rows_p, int cols_p, float* dst, hipStream_t st) {
  hipLaunchKernelGGL(pad_matrix_kernel, dim3(ew_grid((size_t)rows_p * cols_p)), dim3(256), 0, st, src, rows, cols, rows_p, cols_p, dst);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template <typename T, bool ADD>
__global__ __launch_bounds__(EW_BLOCK) void sd_kernel(const T* __restrict__ a, const T* __restrict__ res, const float* __restrict__ mask,
                                                     size_t chunks_per_sample, T* __restrict__ y, size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    const float m = mask[i / chunks_per_sample];
    Chunk<T> v, r;
    v.load(a + i * EPC);
    if (ADD) r.load(res + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v.v[e] = v.v[e] * m + (ADD ? r.v[e] : 0.f);
    v.store(y + i * EPC);
  }
}
template <typename T>
int sd_residual_add(const T* branch, const T* res, const float* mask, int N, size_t per_sample, T* y, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(per_sample % EPC == 0, "sd_residual_add: per-sample size");
  const size_t nch = (size_t)N * (per_sample / EPC);
  hipLaunchKernelGGL((sd_kernel<T, true>), dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, branch, res, mask, per_sample / EPC, y, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
template <typename T>
int sd_row_scale(const T* dy, const float* mask, int N, size_t per_sample, T* out, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(per_sample % EPC == 0, "sd_row_scale: per-sample size");
  const size_t nch = (size_t)N * (per_sample / EPC);
  hipLaunchKernelGGL((sd_kernel<T, false>), dim3(ew_grid(nch)), dim3(EW_BLOCK), 0, st, dy, (const T*)nullptr, mask, per_sample / EPC, out, nch);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void ew_add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                         size_t nchunks) {
  constexpr int EPC = DT<T>::EPC;
  for (size_t i = blockIdx.x * (size_t)EW_BLOCK + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * EW_BLOCK) {
    Chunk<T> u, v;
    u.load(a + i * EPC); v.load(b + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) u.v[e] += v.v[e];
    u.store(out + i * EPC);
  }
}
template <typename T>
int ew_add(const T* a, const T* b, T* out, size_t n, hipStream_t st) {
  constexpr int EPC = DT<T>::EPC;
  ARG_CHECK(n % EPC == 0, "ew_add: size");
  hipLaunchKernelGGL(ew_add_kernel<T>, dim3(ew_grid(n / EPC)), dim3(EW_BLOCK), 0, st, a, b, out, n / EPC);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

#define INST_DW(T)                                                                                        \
  template int dw_stage_weights<T>(const float*, int, int, T*, hipStream_t, int);                         \
  template int dwconv3_fwd<T>(const T*, const T*, int, int, int, int, int, T*, hipStream_t, int, const float*, bool); \
  template int dwconv3_dgrad<T>(const T*, const T*, int, int, int, int, int, T*, hipStream_t, int, bool); \
  template int dwconv3_wgrad<T>(const T*, const T*, int, int, int, int, int, float*, float*, int, hipStream_t, int, float*); \
  template int se_scale_fwd<T>(const T*, const float*, int, int, int, T*, hipStream_t);                   \
  template int se_dgate<T>(const T*, const T*, int, int, int, float*, hipStream_t);                       \
  template int gap_reduce<T>(const T*, const T*, int, int, int, float, float*, hipStream_t);              \
  template int se_dx<T>(const T*, const float*, const float*, int, int, int, T*, hipStream_t);            \
  template int sd_residual_add<T>(const T*, const T*, const float*, int, size_t, T*, hipStream_t);        \
  template int ew_add<T>(const T*, const T*, T*, size_t, hipStream_t);                                    \
  template int sd_row_scale<T>(const T*, const float*, int, size_t, T*, hipStream_t);
INST_DW(float)
INST_DW(bf16_t)

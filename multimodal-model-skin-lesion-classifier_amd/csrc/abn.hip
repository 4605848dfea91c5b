// Algebraic BatchNorm backward for an expanding 1x1 convolution x = y W^T (w -> 4w channels, ResNet bottleneck conv3) followed by a
// training-mode BatchNorm:  dz = cA (.) g + cB (.) x + cC  (g = masked gradient of the BatchNorm output, coefficients per channel from
// bn_bwd_finalize).  Because x IS y W^T, neither the data gradient nor the weight gradient needs dz -- or x -- in memory:
//     dy = dz W        = [g | y] [cA (.) W ; Q] + r          Q = W^T diag(cB) W  (w x w),   r = cC W          (one GEMM, K = 4w + w)
//     dW = dz^T y      = cA (.) (g^T y) + cB (.) (W (y^T y)) + cC (x) colsum(y)                                (one GEMM + a w-deep fix-up)
// so the BatchNorm-backward apply pass (read g, read x, write dz: three passes over the block's widest tensor) disappears and both
// GEMMs read g directly (profiles/r04_experiments.txt).  Reference semantics: autograd through nn.BatchNorm2d + nn.Conv2d of
// torchvision's Bottleneck under model.train() (train_pad_20.py:102,112).  bf16 mode only; W below is the bf16-rounded weight the
// forward multiplied with, so that y W^T reproduces the forward's accumulators.
#include "conv.h"
#include "ops.h"

__device__ __forceinline__ float abn_wb(float w) { return bf16_bits_to_f32(f32_to_bf16_bits(w)); }

// grid = Cw blocks (one per input channel k), 256 threads.  wd [Cw][C4 + Cw] bf16, bias [Cw], coef_copy [3][C4].
// The Q row of this k is a C4-deep sum per (k2): the 256 threads split it as (256 / Cw) parts x Cw columns with eight independent
// loads in flight each (the first version gave every k2 one thread and one dependent load at a time: 75 us on the critical path).
__global__ __launch_bounds__(256) void abn_prep_kernel(const float* __restrict__ W, const float* __restrict__ cA, const float* __restrict__ cB,
                                                       const float* __restrict__ cC, int C4, int Cw, bf16_t* __restrict__ wd,
                                                       float* __restrict__ bias, float* __restrict__ coef_copy) {
  __shared__ float colk[1024];   // cB[o] * W[o][k]
  __shared__ float red[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  bf16_t* row = wd + (size_t)k * (C4 + Cw);
  float bsum = 0.f;
  for (int o = tid; o < C4; o += 256) {
    const float w = abn_wb(W[(size_t)o * Cw + k]);
    row[o] = (bf16_t)f32_to_bf16_bits(cA[o] * w);
    colk[o] = cB[o] * w;
    bsum += cC[o] * w;
  }
  red[tid] = bsum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) bias[k] = red[0];
  __syncthreads();
  const int np = 256 / Cw, k2 = tid % Cw, part = tid / Cw;   // Cw in {64, 128, 256}
  const int per = C4 / np, ob = part * per;
  float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
  const float* wp = W + (size_t)ob * Cw + k2;
  for (int o = 0; o < per; o += 4, wp += (size_t)4 * Cw) {
    q0 += abn_wb(wp[0]) * colk[ob + o];
    q1 += abn_wb(wp[Cw]) * colk[ob + o + 1];
    q2 += abn_wb(wp[2 * Cw]) * colk[ob + o + 2];
    q3 += abn_wb(wp[3 * Cw]) * colk[ob + o + 3];
  }
  red[tid] = (q0 + q1) + (q2 + q3);
  __syncthreads();
  if (part == 0) {
    float q = red[k2];
    for (int j = 1; j < np; ++j) q += red[j * Cw + k2];
    row[C4 + k2] = (bf16_t)f32_to_bf16_bits(q);
  }
  if (k == 0)
    for (int o = tid; o < C4; o += 256) { coef_copy[o] = cA[o]; coef_copy[C4 + o] = cB[o]; coef_copy[2 * C4 + o] = cC[o]; }
}

int abn_prep(const float* W, const float* cA, const float* cB, const float* cC, int C4, int Cw, bf16_t* wd, float* bias, float* coef_copy,
             hipStream_t st) {
  ARG_CHECK((Cw == 64 || Cw == 128 || Cw == 256) && C4 <= 1024 && C4 % (4 * (256 / Cw)) == 0, "abn_prep: C4=%d Cw=%d", C4, Cw);
  hipLaunchKernelGGL(abn_prep_kernel, dim3(Cw), dim3(256), 0, st, W, cA, cB, cC, C4, Cw, wd, bias, coef_copy);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// dW[o][k] = cA[o] S[o][k] + cB[o] sum_k' W[o][k'] Gm[k'][k] + cC[o] colsum[k];  S = [g^T y ; y^T y] as [C4 + gram rows][Cw]
// grid = C4 blocks, Cw threads (Cw <= 256)
__global__ void abn_wgrad_finalize_kernel(const float* __restrict__ S, const float* __restrict__ Gm, const float* __restrict__ colsum,
                                          const float* __restrict__ W, const float* __restrict__ coef, int C4, int Cw, float* __restrict__ dW) {
  const int o = blockIdx.x, k = threadIdx.x;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;   // Cw % 4 == 0
  const float* wr = W + (size_t)o * Cw;
  for (int k2 = 0; k2 < Cw; k2 += 4) {
    t0 += abn_wb(wr[k2]) * Gm[(size_t)k2 * Cw + k];
    t1 += abn_wb(wr[k2 + 1]) * Gm[(size_t)(k2 + 1) * Cw + k];
    t2 += abn_wb(wr[k2 + 2]) * Gm[(size_t)(k2 + 2) * Cw + k];
    t3 += abn_wb(wr[k2 + 3]) * Gm[(size_t)(k2 + 3) * Cw + k];
  }
  const float t = (t0 + t1) + (t2 + t3);
  dW[(size_t)o * Cw + k] = coef[o] * S[(size_t)o * Cw + k] + coef[C4 + o] * t + coef[2 * C4 + o] * colsum[k];
}

int abn_wgrad_finalize(const float* S, const float* colsum, const float* W, const float* coef, int C4, int Cw, float* dW, hipStream_t st,
                       const float* gram) {
  ARG_CHECK(Cw <= 256, "abn_wgrad_finalize: Cw=%d", Cw);
  hipLaunchKernelGGL(abn_wgrad_finalize_kernel, dim3(C4), dim3(Cw), 0, st, S, gram ? gram : S + (size_t)C4 * Cw, colsum, W, coef, C4, Cw, dW);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// ---- second phase: the BatchNorm-backward sum  sum_m g[m][o] x[m][o]  without x:  x = y W^T  =>  it is  sum_k W[o][k] (g^T y)[o][k],
// a row-wise dot product of the weight with the S the weight-gradient GEMM has just produced.  grid = C4 blocks, 64 threads.
__global__ __launch_bounds__(64) void abn_sgx_kernel(const float* __restrict__ S, const float* __restrict__ W, int Cw, float* __restrict__ sgx) {
  const int o = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < Cw; k += 64) s += abn_wb(W[(size_t)o * Cw + k]) * S[(size_t)o * Cw + k];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
  if (threadIdx.x == 0) sgx[o] = s;
}
int abn_sgx(const float* S, const float* W, int C4, int Cw, float* sgx, hipStream_t st) {
  hipLaunchKernelGGL(abn_sgx_kernel, dim3(C4), dim3(64), 0, st, S, W, Cw, sgx);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// ---- forward: the BatchNorm statistics of x = y W^T from the Gram matrix and the column sums of y (one pass over y, the block's
// narrowest tensor, instead of a statistics-only pass of the convolution):
//     sum_m x[m][o] = sum_k W[o][k] colsum[k]            sum_m x[m][o]^2 = w_o^T (y^T y) w_o
// W as multiplied (bf16-rounded); sums in double, stored as the one-row fp32 statistics table bn_finalize reads.  grid = C4 blocks, Cw threads.
struct GramBn {   // stat_sum == nullptr: finalize here -- the BatchNorm coefficients of ops.hip bn_finalize straight from the two sums
  double count; const float* gamma; const float* beta; float eps, momentum;
  float* running_mean; float* running_var; float* scale; float* shift; float* save_mean; float* save_invstd;
};
__global__ __launch_bounds__(256) void gram_stats_kernel(const float* __restrict__ Gm, const float* __restrict__ colsum, const float* __restrict__ W,
                                                         int Cw, float* __restrict__ stat_sum, float* __restrict__ stat_sq, const GramBn bn) {
  __shared__ float wrow[256];
  __shared__ double red[2][256];
  const int o = blockIdx.x, k = threadIdx.x;
  const float wk = abn_wb(W[(size_t)o * Cw + k]);
  wrow[k] = wk;
  __syncthreads();
  double t[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // eight independent loads in flight (Cw % 8 == 0; two chains took 10 us)
  for (int k2 = 0; k2 < Cw; k2 += 8) {
    float gv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[e] = Gm[(size_t)(k2 + e) * Cw + k];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] += (double)wrow[k2 + e] * (double)gv[e];
  }
  red[0][k] = (double)wk * (double)colsum[k];
  red[1][k] = (double)wk * (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7])));
  __syncthreads();
  for (int s = Cw >> 1; s > 0; s >>= 1) {
    if (k < s) { red[0][k] += red[0][k + s]; red[1][k] += red[1][k + s]; }
    __syncthreads();
  }
  if (k == 0) {
    if (stat_sum) { stat_sum[o] = (float)red[0][0]; stat_sq[o] = (float)red[1][0]; }
    else   // (the sums go through fp32 exactly as the statistics table would carry them)
      bn_fwd_coeffs(o, (double)(float)red[0][0], (double)(float)red[1][0], bn.count, bn.gamma, bn.beta, bn.eps, bn.momentum, bn.running_mean, bn.running_var, bn.scale,
                    bn.shift, bn.save_mean, bn.save_invstd);
  }
}
int gram_stats(const float* gram, const float* colsum, const float* W, int C4, int Cw, float* stat_sum, float* stat_sq, hipStream_t st) {
  ARG_CHECK(Cw == 64 || Cw == 128 || Cw == 256, "gram_stats: Cw=%d", Cw);
  hipLaunchKernelGGL(gram_stats_kernel, dim3(C4), dim3(Cw), 0, st, gram, colsum, W, Cw, stat_sum, stat_sq, GramBn{});
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
// ... and the BatchNorm finalize in the same launch (training forward of the two-pass units: one dependent launch less per block)
int gram_stats_finalize(const float* gram, const float* colsum, const float* W, int C4, int Cw, double count, const float* gamma, const float* beta, float eps,
                        float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                        hipStream_t st) {
  ARG_CHECK(Cw == 64 || Cw == 128 || Cw == 256, "gram_stats_finalize: Cw=%d", Cw);
  GramBn bn = {count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd};
  hipLaunchKernelGGL(gram_stats_kernel, dim3(C4), dim3(Cw), 0, st, gram, colsum, W, Cw, (float*)nullptr, (float*)nullptr, bn);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

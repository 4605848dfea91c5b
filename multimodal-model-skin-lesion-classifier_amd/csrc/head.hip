// Fusion-head operators (fp32) for gfx950.  Replaces nn.Linear / nn.LayerNorm / sigmoid gates /
// MetaBlock / gated-residual pointwise math / small softmax attention / embedding gather of
// multimodalIntraInterModal.py:172-412, metablock.py:27-32, gatedResidualBlock.py:12-17,
// tab_transformer.py:40-60.  Dense contractions run on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32);
// row reductions (LayerNorm, softmax) are wavefront-shuffle reductions, one 64-lane wave per row.
#include <stdlib.h>
#include <string.h>

#include "../../include/mmskin.h"
#include "common.h"
#include "conv.h"

#define ST(s) ((hipStream_t)(s))

// Library-owned scratch for split reductions (split-K partial tiles, column-sum partial rows).  The head's
// C entry points carry no workspace argument (they mirror nn.Linear / nn.LayerNorm call sites), so the
// buffer is allocated lazily and only ever grows; every user runs on the caller's stream, in order.
// One buffer PER DEVICE (the process model is one process per GPU, but nothing here may hand device 0's memory to a launch on
// device 1); `slot` separates the regions two nested entry points use at the same time.
static float* head_scratch(size_t bytes, int slot = 0) {
  constexpr int MAXDEV = 16;
  static float* buf[2][MAXDEV] = {};
  static size_t cap[2][MAXDEV] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAXDEV) return nullptr;
  if (bytes > cap[slot][d]) {
    if (buf[slot][d]) (void)hipFree(buf[slot][d]);   // implicit device sync: no kernel still reads the old buffer
    size_t want = bytes < (size_t)(8 << 20) ? (size_t)(8 << 20) : bytes;
    if (hipMalloc((void**)&buf[slot][d], want) != hipSuccess) { buf[slot][d] = nullptr; cap[slot][d] = 0; return nullptr; }
    cap[slot][d] = want;
  }
  return buf[slot][d];
}
// out[i] = sum_s part[s*n + i]
__global__ void split_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int S, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int s2 = 0; s2 < S; ++s2) t += part[(int64_t)s2 * n + i];
    out[i] = t;
  }
}

// ------------------------------------------------------------------ generic strided f32 GEMM
// C[m][n] = sum_k A(m,k) * B(n,k) (+ bias[n]) ; A(m,k) = a[m*sam + k*sak] ; B(n,k) = b[n*sbn + k*sbk]
// The head's GEMMs have M = batch (256) and N, K <= 2048: tiny for a 256-CU chip, so the kernel is built
// for latency, not throughput: 32x32 output tiles (many workgroups), K-steps of 128 (16 + 16 independent loads per thread in flight;
// with K-steps of 32 a K = 512 GEMM was 16 dependent load -> barrier -> multiply rounds, 12 - 16 us for 0.13 GFLOP) with the next
// step's operands prefetched into registers while the current one is multiplied on the exact-f32 MFMA
// (v_mfma_f32_16x16x4_f32, one 16x16 fragment per wave).
#define LG_BK 128
#define LG_T 32
#define LG_PK 130  // pitch of a [row][k] tile (k-contiguous source): 2 row + g distinct over a half wave's 16 rows x 2 k
#define LG_PM 48   // pitch of a [k][row] tile (row-contiguous source)
#define LG_EPT (LG_T * LG_BK / 256)   // elements per thread and operand
template <bool KC> __device__ __forceinline__ int lg_idx(int row, int k) { return KC ? row * LG_PK + k : k * LG_PM + row; }

constexpr int LG_LDS = LG_BK * LG_PM > LG_T * LG_PK ? LG_BK * LG_PM : LG_T * LG_PK;   // floats per operand tile
// one 32 x 32 output tile (bx, by) of C = A B^T (+ bias, ReLU); As / Bs: LG_LDS floats each
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_f32_tile(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c,
                                              const float* __restrict__ bias, int M, int N, int K, int64_t sam, int64_t sak, int64_t sbn,
                                              int64_t sbk, int64_t ldc, int relu, int bx, int by, float* As, float* Bs) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int m0 = by * LG_T, n0 = bx * LG_T;
  const int wm = wid >> 1, wn = wid & 1;
  f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float ra[LG_EPT], rb[LG_EPT];
  // element e = tid + 256*i of a 32 x LG_BK tile: k-contiguous sources walk k fastest, else rows fastest
  auto coords = [&](bool kc, int e, int& row, int& kk) { if (kc) { row = e / LG_BK; kk = e % LG_BK; } else { row = e % LG_T; kk = e / LG_T; } };
#define LG_FETCH(k0)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < LG_EPT; ++i) {                                                  \
    int row, kk;                                                                                        \
    coords(A_KC, tid + 256 * i, row, kk);                                                               \
    ra[i] = (m0 + row < M && (k0) + kk < K) ? a[(int64_t)(m0 + row) * sam + (int64_t)((k0) + kk) * sak] : 0.f; \
    coords(B_KC, tid + 256 * i, row, kk);                                                               \
    rb[i] = (n0 + row < N && (k0) + kk < K) ? b[(int64_t)(n0 + row) * sbn + (int64_t)((k0) + kk) * sbk] : 0.f; \
  }
  LG_FETCH(0)
  for (int k0 = 0; k0 < K; k0 += LG_BK) {
#pragma unroll
    for (int i = 0; i < LG_EPT; ++i) {
      int row, kk;
      coords(A_KC, tid + 256 * i, row, kk);
      As[lg_idx<A_KC>(row, kk)] = ra[i];
      coords(B_KC, tid + 256 * i, row, kk);
      Bs[lg_idx<B_KC>(row, kk)] = rb[i];
    }
    __syncthreads();
    if (k0 + LG_BK < K) { LG_FETCH(k0 + LG_BK) }   // in flight while this step is multiplied
#pragma unroll
    for (int s = 0; s < LG_BK / 4; ++s) {
      float fa = As[lg_idx<A_KC>(wm * 16 + l15, 4 * s + g)];
      float fb = Bs[lg_idx<B_KC>(wn * 16 + l15, 4 * s + g)];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fa, acc, 0, 0, 0);
    }
    __syncthreads();
  }
#undef LG_FETCH
  // D[i = n][j = m]: lane holds m = l15, n = 4g + reg
  const int m = m0 + wm * 16 + l15;
  if (m < M) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int n = n0 + wn * 16 + 4 * g + r;
      if (n < N) {
        float v = acc[r] + (bias ? bias[n] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        c[(int64_t)m * ldc + n] = v;
      }
    }
  }
}
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ c, const float* __restrict__ bias, int M,
                                                       int N, int K, int64_t sam, int64_t sak, int64_t sbn,
                                                       int64_t sbk, int64_t ldc, int relu, int kchunk, int64_t sab,
                                                       int64_t sbb, int64_t scb) {
  __shared__ float As[LG_LDS];
  __shared__ float Bs[LG_LDS];
  if (kchunk < 0) {   // batched: blockIdx.z = batch index (no split-K)
    a += (int64_t)blockIdx.z * sab; b += (int64_t)blockIdx.z * sbb; c += (int64_t)blockIdx.z * scb;
  }
  // split-K (kchunk > 0): slice blockIdx.z covers k in [z*kchunk, (z+1)*kchunk) and writes its own M x ldc
  // partial matrix; the caller sums the slices.  Used by the weight-gradient GEMMs whose K is batch*tokens.
  if (kchunk > 0) {
    const int kb = blockIdx.z * kchunk;
    a += (int64_t)kb * sak; b += (int64_t)kb * sbk;
    c += (int64_t)blockIdx.z * M * ldc;
    K = min(kchunk, K - kb);
  }
  gemm_f32_tile<A_KC, B_KC>(a, b, c, bias, M, N, K, sam, sak, sbn, sbk, ldc, relu, blockIdx.x, blockIdx.y, As, Bs);
}
// Backward of a small Linear (M = batch rows) in ONE launch: the tiles of dx = g w, the tiles of dw = g^T x and the column sums db = sum_m g,
// selected by block index (three dependent-free launches of 5 - 11 us each were 4.5 us of launch latency apiece: 16 Linear layers per head step)
__global__ __launch_bounds__(256) void linear_bwd_small_kernel(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ x,
                                                               float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int M, int K,
                                                               int N, int nx_dx, int n_dx, int nx_dw, int n_dw) {
  __shared__ float As[LG_LDS];
  __shared__ float Bs[LG_LDS];
  int b = blockIdx.x;
  if (b < n_dx) {   // dx[m][k] = sum_n g[m][n] w[n][k]
    gemm_f32_tile<true, false>(g, w, dx, nullptr, M, K, N, N, 1, 1, K, K, 0, b % nx_dx, b / nx_dx, As, Bs);
    return;
  }
  b -= n_dx;
  if (b < n_dw) {   // dw[n][k] = sum_m g[m][n] x[m][k]
    gemm_f32_tile<false, false>(g, x, dw, nullptr, N, K, M, 1, N, 1, K, K, 0, b % nx_dw, b / nx_dw, As, Bs);
    return;
  }
  b -= n_dw;        // db: 32 columns x 8 row lanes, four independent sums per lane
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5, n = b * 32 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    int m = ry;
    for (; m + 24 < M; m += 32) {
      s0 += g[(int64_t)m * N + n]; s1 += g[(int64_t)(m + 8) * N + n]; s2 += g[(int64_t)(m + 16) * N + n]; s3 += g[(int64_t)(m + 24) * N + n];
    }
    for (; m < M; m += 8) s0 += g[(int64_t)m * N + n];
  }
  As[ry * 32 + cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += As[i * 32 + cx];
    db[n] = t;
  }
}

static int gemm_f32(const float* a, const float* b, float* c, const float* bias, int M, int N, int K, int64_t sam,
                    int64_t sak, int64_t sbn, int64_t sbk, int64_t ldc, int relu, hipStream_t st, int batch = 1,
                    int64_t sab = 0, int64_t sbb = 0, int64_t scb = 0) {
  if (M <= 0 || N <= 0) return MMSKIN_OK;
  dim3 grid(ceil_div(N, LG_T), ceil_div(M, LG_T));
  const bool akc = sak == 1, bkc = sbk == 1;
  if (batch > 1) {
    grid.z = batch;
#define LAUNCHB(X, Y) hipLaunchKernelGGL((gemm_f32_kernel<X, Y>), grid, dim3(256), 0, st, a, b, c, bias, M, N, K, sam, sak, sbn, sbk, ldc, relu, -1, sab, sbb, scb)
    if (akc && bkc) LAUNCHB(true, true);
    else if (akc) LAUNCHB(true, false);
    else if (bkc) LAUNCHB(false, true);
    else LAUNCHB(false, false);
#undef LAUNCHB
    HIP_CHECK_RET(hipGetLastError());
    return MMSKIN_OK;
  }
  // few output tiles but a long contraction (dW = dY^T X over batch*tokens rows): split K over blockIdx.z
  int S = 1, kchunk = 0;
  float* out = c;
  const int tiles = grid.x * grid.y;
  if (K >= 4096 && tiles < 256 && !bias && !relu && ldc == N) {
    S = ceil_div(512, tiles);
    if (S > ceil_div(K, 512)) S = ceil_div(K, 512);
    if (S > 1) {
      kchunk = ceil_div(ceil_div(K, S), LG_BK) * LG_BK;
      S = ceil_div(K, kchunk);
      out = head_scratch((size_t)S * M * N * sizeof(float));
      if (!out) { mmskin_set_error("gemm_f32: split-K scratch allocation failed"); return MMSKIN_ERR_HIP; }
      grid.z = S;
    }
  }
#define LAUNCH(X, Y) hipLaunchKernelGGL((gemm_f32_kernel<X, Y>), grid, dim3(256), 0, st, a, b, out, bias, M, N, K, sam, sak, sbn, sbk, ldc, relu, S > 1 ? kchunk : 0, (int64_t)0, (int64_t)0, (int64_t)0)
  if (akc && bkc) LAUNCH(true, true);
  else if (akc) LAUNCH(true, false);
  else if (bkc) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
  HIP_CHECK_RET(hipGetLastError());
  if (S > 1) {
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(split_reduce_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, st, out, c, S, n);
    HIP_CHECK_RET(hipGetLastError());
  }
  return MMSKIN_OK;
}

// out[k][n] = in[n][k]  (rows x cols -> cols x rows), 32x32 tiles through LDS
__global__ void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[i][threadIdx.x] = in[(int64_t)r * cols + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < rows && c < cols) out[(int64_t)c * rows + r] = tile[threadIdx.x][i];
  }
}
// Linear layers over batch x tokens rows with 64-multiple widths run on the exact-f32 implicit-GEMM conv kernels
static inline bool linear_big(int M, int K, int N) { return M >= 2048 && K % 64 == 0 && N % 64 == 0; }
// MMSKIN_LINEAR_DTYPE=bf16: those GEMMs take bf16 operands (fp32 accumulate, fp32 tensors at the boundary): the inputs are
// converted into library scratch, the bf16 MFMA kernels run, the result is converted back.  Default fp32 (parity mode).
// The mode is a process-wide setting: MMSKIN_LINEAR_DTYPE at first use, or mmskin_set_linear_dtype() at any time.
static int g_linear_dtype = -1;   // -1: not read yet; MMSKIN_F32 / MMSKIN_BF16
static inline bool linear_bf16() {
  if (g_linear_dtype < 0) { const char* e = getenv("MMSKIN_LINEAR_DTYPE"); g_linear_dtype = (e && !strcmp(e, "bf16")) ? 1 : 0; }
  return g_linear_dtype == 1;
}
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    reinterpret_cast<uint2*>(out)[i] = make_uint2(f32_to_bf16_bits(v.x) | (f32_to_bf16_bits(v.y) << 16),
                                                  f32_to_bf16_bits(v.z) | (f32_to_bf16_bits(v.w) << 16));
  }
}
__global__ void bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const uint2 v = reinterpret_cast<const uint2*>(in)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(bf16_bits_to_f32(v.x & 0xffffu), bf16_bits_to_f32(v.x >> 16),
                                                    bf16_bits_to_f32(v.y & 0xffffu), bf16_bits_to_f32(v.y >> 16));
  }
}
// out[k][n] (bf16) = in[n][k] (fp32)
__global__ void transpose_f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[i][threadIdx.x] = in[(int64_t)r * cols + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < rows && c < cols) out[(int64_t)c * rows + r] = (bf16_t)f32_to_bf16_bits(tile[threadIdx.x][i]);
  }
}
static int cvt_to_bf16(const float* in, bf16_t* out, int64_t n, hipStream_t st) {
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(256), 0, st, in, out, n4);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
static int cvt_to_f32(const bf16_t* in, float* out, int64_t n, hipStream_t st) {
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(256), 0, st, in, out, n4);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// ---- Linear layers whose widths are not multiples of 64 (DaViT's 96 / 288-wide first stage over 200 704 tokens) on the bf16 GEMM
// kernels: operands are converted into zero-padded bf16 copies (the conversion pass exists anyway), the GEMM runs on the padded
// widths, and the result is un-padded while it is widened (+ bias / activation).  Zero pad columns contribute exact zeros.
static inline int pad64(int v) { return (v + 63) / 64 * 64; }
static inline bool linear_big_padded(int M, int K, int N) {
  return M >= 2048 && K % 8 == 0 && N % 8 == 0 && K >= 32 && N >= 32 && !(K % 64 == 0 && N % 64 == 0);
}
// out [rows_pad][cols_pad] bf16 <- in [rows][cols] fp32, zeros elsewhere; one 16-byte chunk (8 values) per thread
__global__ void f32_to_bf16_pad_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t rows, int cols, int64_t rows_pad,
                                       int cols_pad) {
  const int cpr = cols_pad / 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows_pad * cpr; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (r < rows && c < cols) {
      const float4 a = *reinterpret_cast<const float4*>(in + r * cols + c), b = *reinterpret_cast<const float4*>(in + r * cols + c + 4);
      o = make_uint4(f32_to_bf16_bits(a.x) | (f32_to_bf16_bits(a.y) << 16), f32_to_bf16_bits(a.z) | (f32_to_bf16_bits(a.w) << 16),
                     f32_to_bf16_bits(b.x) | (f32_to_bf16_bits(b.y) << 16), f32_to_bf16_bits(b.z) | (f32_to_bf16_bits(b.w) << 16));
    }
    *reinterpret_cast<uint4*>(out + r * cols_pad + c) = o;
  }
}
// out [rows][cols] fp32 <- act(in [rows][cols_pad] bf16 + bias); act 0 none / 1 ReLU / 2 exact GELU
__global__ void bf16_unpad_bias_act_kernel(const bf16_t* __restrict__ in, const float* __restrict__ bias, float* __restrict__ out,
                                           int64_t rows, int cols, int cols_pad, int act, const float* __restrict__ res) {
  const int cpr = cols / 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows * cpr; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    Chunk<bf16_t> v;
    v.load(in + r * cols_pad + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = v.v[e] + (bias ? bias[c + e] : 0.f);
      if (act == 1) t = fmaxf(t, 0.f);
      else if (act == 2) t = 0.5f * t * (1.f + erff(t * 0.70710678118654752f));
      v.v[e] = t;
    }
    if (res) {   // y = residual + act(...): the block's skip connection in the same pass
      const float4 r0 = *reinterpret_cast<const float4*>(res + r * cols + c), r1 = *reinterpret_cast<const float4*>(res + r * cols + c + 4);
      v.v[0] += r0.x; v.v[1] += r0.y; v.v[2] += r0.z; v.v[3] += r0.w; v.v[4] += r1.x; v.v[5] += r1.y; v.v[6] += r1.z; v.v[7] += r1.w;
    }
    *reinterpret_cast<float4*>(out + r * cols + c) = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]);
    *reinterpret_cast<float4*>(out + r * cols + c + 4) = make_float4(v.v[4], v.v[5], v.v[6], v.v[7]);
  }
}
// out[k][n] (bf16, row pitch out_pitch) = in[n][k] (fp32): the transposed weight inside a zero-filled padded matrix
__global__ void transpose_f32_to_bf16_pitch_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int rows, int cols, int out_pitch) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[i][threadIdx.x] = in[(int64_t)r * cols + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < rows && c < cols) out[(int64_t)c * out_pitch + r] = (bf16_t)f32_to_bf16_bits(tile[threadIdx.x][i]);
  }
}
static int cvt_to_bf16_pad(const float* in, bf16_t* out, int64_t rows, int cols, int64_t rows_pad, int cols_pad, hipStream_t st) {
  const int64_t n = rows_pad * (cols_pad / 8);
  int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(f32_to_bf16_pad_kernel, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(256), 0, st, in, out, rows, cols, rows_pad, cols_pad);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
static int unpad_bias_act(const bf16_t* in, const float* bias, float* out, int64_t rows, int cols, int cols_pad, int act, hipStream_t st,
                          const float* res = nullptr) {
  const int64_t n = rows * (cols / 8);
  int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(bf16_unpad_bias_act_kernel, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(256), 0, st, in, bias, out, rows, cols, cols_pad, act, res);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

__global__ void relu_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// blockIdx.y = row group (rows [y*per, (y+1)*per)) -> out[y*N + n]; one group = plain column sum
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N,
                                                     int per) {
  __shared__ float red[8][32];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;     // 32 columns x 8 row lanes per block
  const int n = blockIdx.x * 32 + cx;
  const int m_end = min(M, ((int)blockIdx.y + 1) * per);
  out += (int64_t)blockIdx.y * N;
  float s = 0.f;
  if (n < N)
    for (int m = blockIdx.y * per + ry; m < m_end; m += 8) s += x[(int64_t)m * N + n];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][cx];
    out[n] = t;
  }
}
static inline int grid1d(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------ LayerNorm
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                            int N, float eps, int relu) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (int64_t)row * N;
  float s = 0.f;
  for (int i = lane; i < N; i += 64) s += xr[i];
  const float mu = wave_sum(s) / (float)N;
  float q = 0.f;
  for (int i = lane; i < N; i += 64) { float d = xr[i] - mu; q += d * d; }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)N + eps);
  for (int i = lane; i < N; i += 64) {
    float v = (xr[i] - mu) * rs * g[i] + b[i];
    if (relu) v = fmaxf(v, 0.f);
    y[(int64_t)row * N + i] = v;
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ g, const float* __restrict__ b,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ dx,
                                                               int M, int N, int relu) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float mu = mean[row], rs = rstd[row];
  const float* xr = x + (int64_t)row * N;
  const float* dr = dy + (int64_t)row * N;
  float c1 = 0.f, c2 = 0.f;
  for (int i = lane; i < N; i += 64) {
    float xh = (xr[i] - mu) * rs;
    float d = dr[i];
    if (relu && !(xh * g[i] + b[i] > 0.f)) d = 0.f;
    float dg = d * g[i];
    c1 += dg; c2 += dg * xh;
  }
  c1 = wave_sum(c1) / (float)N; c2 = wave_sum(c2) / (float)N;
  for (int i = lane; i < N; i += 64) {
    float xh = (xr[i] - mu) * rs;
    float d = dr[i];
    if (relu && !(xh * g[i] + b[i] > 0.f)) d = 0.f;
    dx[(int64_t)row * N + i] = rs * (d * g[i] - c1 - xh * c2);
  }
}
// out[c] = sum_r part[r][c] for c < ncols (c < split -> out0[c], else out1[c - split]; either may be null): 16 columns x 64 row lanes
// per block -- narrow outputs (96 .. 768 columns) still spread over 6 .. 48 workgroups, and a lane adds R / 64 rows
__global__ __launch_bounds__(1024) void rows_sum_kernel(const float* __restrict__ part, int R, int ncols, int split, float* __restrict__ out0,
                                                        float* __restrict__ out1) {
  __shared__ float red[64][17];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  float s0 = 0.f, s1 = 0.f;
  if (c < ncols) {
    int r = ry;
    for (; r + 64 < R; r += 128) { s0 += part[(int64_t)r * ncols + c]; s1 += part[(int64_t)(r + 64) * ncols + c]; }
    if (r < R) s0 += part[(int64_t)r * ncols + c];
  }
  red[ry][cx] = s0 + s1;
  __syncthreads();
  if (ry == 0 && c < ncols) {
    float t = 0.f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) t += red[k][cx];
    float* out = c < split ? out0 : out1;
    if (out) out[c < split ? c : c - split] = t;
  }
}
static int rows_sum(const float* part, int R, int ncols, int split, float* out0, float* out1, hipStream_t st) {
  hipLaunchKernelGGL(rows_sum_kernel, dim3(ceil_div(ncols, 16)), dim3(1024), 0, st, part, R, ncols, split, out0, out1);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
// Column sums with float4 loads: CW chunk columns x (256 / CW) row lanes per block, two rows in flight per lane, one partial row
// per block, rows_sum adds the blocks.  CVT: the same pass also writes the bf16 copy (row pitch cols_pad, zero pad columns) a
// bf16-operand GEMM consumes -- Linear backward needs both from dy (bias gradient and the dgrad / wgrad operand), so dy is read once.
// GELU: x is the gradient w.r.t. gelu(z); it is multiplied by gelu'(z) on the way in (the exact-erf GELU of nn.GELU), so a Linear ->
// GELU pair's backward reads dy and z once and never writes the fp32 gradient of the pre-activation.
__device__ __forceinline__ float gelu_grad(float z) {
  return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}
template <bool CVT, bool GELU = false>
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ x, int64_t M, int N, int rows_per_block, int CW, int lanes,
                                                      float* __restrict__ part, bf16_t* __restrict__ out16, int cols_pad,
                                                      const float* __restrict__ zg = nullptr) {
  extern __shared__ float cs_red[];           // [lanes][CW * 4]
  const int cx = threadIdx.x % CW, ly = threadIdx.x / CW;
  const int c4 = blockIdx.y * CW + cx, col = c4 * 4;
  const int ncol4 = (CVT ? cols_pad : N) / 4;
  const bool real = col < N;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  if (ly < lanes && c4 < ncol4) {
    int64_t r = r0 + ly;
    for (; r + lanes < r1; r += 2 * lanes) {
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (real) { v0 = *reinterpret_cast<const float4*>(x + r * N + col); v1 = *reinterpret_cast<const float4*>(x + (r + lanes) * N + col); }
      if (GELU && real) {
        const float4 z0 = *reinterpret_cast<const float4*>(zg + r * N + col), z1 = *reinterpret_cast<const float4*>(zg + (r + lanes) * N + col);
        v0.x *= gelu_grad(z0.x); v0.y *= gelu_grad(z0.y); v0.z *= gelu_grad(z0.z); v0.w *= gelu_grad(z0.w);
        v1.x *= gelu_grad(z1.x); v1.y *= gelu_grad(z1.y); v1.z *= gelu_grad(z1.z); v1.w *= gelu_grad(z1.w);
      }
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      if (CVT) {
        *reinterpret_cast<uint2*>(out16 + r * cols_pad + col) = make_uint2(f32_to_bf16_bits(v0.x) | (f32_to_bf16_bits(v0.y) << 16), f32_to_bf16_bits(v0.z) | (f32_to_bf16_bits(v0.w) << 16));
        *reinterpret_cast<uint2*>(out16 + (r + lanes) * cols_pad + col) = make_uint2(f32_to_bf16_bits(v1.x) | (f32_to_bf16_bits(v1.y) << 16), f32_to_bf16_bits(v1.z) | (f32_to_bf16_bits(v1.w) << 16));
      }
    }
    if (r < r1) {
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (real) v0 = *reinterpret_cast<const float4*>(x + r * N + col);
      if (GELU && real) {
        const float4 z0 = *reinterpret_cast<const float4*>(zg + r * N + col);
        v0.x *= gelu_grad(z0.x); v0.y *= gelu_grad(z0.y); v0.z *= gelu_grad(z0.z); v0.w *= gelu_grad(z0.w);
      }
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      if (CVT) *reinterpret_cast<uint2*>(out16 + r * cols_pad + col) = make_uint2(f32_to_bf16_bits(v0.x) | (f32_to_bf16_bits(v0.y) << 16), f32_to_bf16_bits(v0.z) | (f32_to_bf16_bits(v0.w) << 16));
    }
    *reinterpret_cast<float4*>(cs_red + ((size_t)ly * CW + cx) * 4) = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CW * 4; i += 256) {
    const int cc = blockIdx.y * CW * 4 + i;
    if (cc < N) {
      float t = 0.f;
      for (int l = 0; l < lanes; ++l) t += cs_red[(size_t)l * CW * 4 + i];
      part[(int64_t)blockIdx.x * N + cc] = t;
    }
  }
}
struct Colsum4Plan { int CW, lanes, gy, nbx, rpb; };
static inline Colsum4Plan colsum4_plan(int64_t M, int ncol4) {
  Colsum4Plan g;
  g.CW = ncol4 <= 32 ? ncol4 : 32;
  g.lanes = 256 / g.CW;
  g.gy = ceil_div(ncol4, g.CW);
  int64_t want = 1024 / g.gy;
  if (want < 1) want = 1;
  int64_t rpb = (M + want - 1) / want;
  const int64_t lo = (int64_t)g.lanes * 4;                  // a lane adds at least four rows
  if (rpb < lo) rpb = lo;
  g.rpb = (int)rpb;
  g.nbx = (int)((M + rpb - 1) / rpb);
  return g;
}
static inline size_t colsum4_part_bytes(int N) { return (size_t)1024 * N * sizeof(float); }
// out[N] = column sums of x [M][N]; out16 != null: also the bf16 copy [M][cols_pad].  part: colsum4_part_bytes(N) of scratch.
static int colsum4(const float* x, float* out, int64_t M, int N, float* part, bf16_t* out16, int cols_pad, hipStream_t st,
                   const float* z_gelu = nullptr) {
  const Colsum4Plan g = colsum4_plan(M, (out16 ? cols_pad : N) / 4);
  const size_t lds = (size_t)g.lanes * g.CW * 4 * sizeof(float);
  if (out16 && z_gelu) hipLaunchKernelGGL((colsum4_kernel<true, true>), dim3(g.nbx, g.gy), dim3(256), lds, st, x, M, N, g.rpb, g.CW, g.lanes, part, out16, cols_pad, z_gelu);
  else if (out16) hipLaunchKernelGGL(colsum4_kernel<true>, dim3(g.nbx, g.gy), dim3(256), lds, st, x, M, N, g.rpb, g.CW, g.lanes, part, out16, cols_pad);
  else hipLaunchKernelGGL(colsum4_kernel<false>, dim3(g.nbx, g.gy), dim3(256), lds, st, x, M, N, g.rpb, g.CW, g.lanes, part, out16, cols_pad);
  HIP_CHECK_RET(hipGetLastError());
  return rows_sum(part, g.nbx, N, N, out, nullptr, st);
}
static int colsum(const float* x, float* out, int M, int N, hipStream_t st) {
  static const bool c4 = [] { const char* v = getenv("MMSKIN_COLSUM4"); return !v || atoi(v) != 0; }();
  if (c4 && N % 4 == 0 && M >= 2048) {
    float* part = head_scratch(colsum4_part_bytes(N));
    if (!part) { mmskin_set_error("colsum: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    return colsum4(x, out, M, N, part, nullptr, 0, st);
  }
  const int G = M >= 2048 ? (M / 256 > 128 ? 128 : M / 256) : 1;
  if (G <= 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 32)), dim3(256), 0, st, x, out, M, N, M);
  } else {
    float* part = head_scratch((size_t)G * N * sizeof(float));
    if (!part) { mmskin_set_error("colsum: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    hipLaunchKernelGGL(colsum_kernel, dim3(ceil_div(N, 32), G), dim3(256), 0, st, x, part, M, N, ceil_div(M, G));
    hipLaunchKernelGGL(split_reduce_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, part, out, G, (int64_t)N);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// gamma/beta gradients: blockIdx.y = row group -> part[(y*2 + {0,1})*N + n]; 64 columns x 4 row lanes per block
__global__ __launch_bounds__(256) void layernorm_bwd_gb_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ g, const float* __restrict__ b,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               float* __restrict__ part, int M, int N, int relu, int per) {
  __shared__ float red[2][4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cx;
  const int m_end = min(M, ((int)blockIdx.y + 1) * per);
  float sg = 0.f, sb = 0.f;
  if (n < N) {
    const float gn = g[n], bn = b[n];
    for (int m = blockIdx.y * per + ry; m < m_end; m += 4) {
      float xh = (x[(int64_t)m * N + n] - mean[m]) * rstd[m];
      float d = dy[(int64_t)m * N + n];
      if (relu && !(xh * gn + bn > 0.f)) d = 0.f;
      sg += d * xh; sb += d;
    }
  }
  red[0][ry][cx] = sg; red[1][ry][cx] = sb;
  __syncthreads();
  if (ry == 0 && n < N) {
    part[((int64_t)blockIdx.y * 2) * N + n] = (red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]);
    part[((int64_t)blockIdx.y * 2 + 1) * N + n] = (red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]);
  }
}
// dg[n] = sum_y part[(2y)*N + n], db[n] = sum_y part[(2y+1)*N + n]
__global__ void layernorm_gb_reduce_kernel(const float* __restrict__ part, float* dg, float* db, int G, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float sg = 0.f, sb = 0.f;
  for (int y = 0; y < G; ++y) { sg += part[((int64_t)y * 2) * N + n]; sb += part[((int64_t)y * 2 + 1) * N + n]; }
  if (dg) dg[n] = sg;
  if (db) db[n] = sb;
}

// Row-in-registers LayerNorm (N % 4 == 0, N <= 2048, no ReLU): LPR lanes own one row (LPR = 32 for N <= 128 so a 64-wide wave
// carries two rows instead of idling 40 lanes), NJ float4 chunks per lane, ONE pass over memory.
//   forward : y as fp32 and / or bf16 (what a bf16-operand Linear consumes), mean / rstd when a backward follows
//   backward: dx AND the lane's running sums of dy * xhat / dy over all the rows it visits -> part[slot][2][N]; a second small
//             kernel adds the slots.  Replaces dx pass + column pass + serial 1-3 block reduction (94 us per LayerNorm of the
//             DaViT stage-1 shape, three reads of dy and x) by one read of each.
template <int LPR> __device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int NJ, int LPR>
__global__ __launch_bounds__(256) void layernorm_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                 const float* __restrict__ b, float* __restrict__ y32,
                                                                 bf16_t* __restrict__ y16, float* __restrict__ mean,
                                                                 float* __restrict__ rstd, int M, int N, float eps) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, sl = lane % LPR;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
  const bool live = row < M;
  const float* xr = x + (int64_t)(live ? row : 0) * N;
  float4 v[NJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = (j * LPR + sl) * 4;
    v[j] = (live && i < N) ? *reinterpret_cast<const float4*>(xr + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  const float mu = row_sum<LPR>(s) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = (j * LPR + sl) * 4;
    if (i < N) { const float a = v[j].x - mu, c = v[j].y - mu, d = v[j].z - mu, e = v[j].w - mu; q += (a * a + c * c) + (d * d + e * e); }
  }
  const float rs = 1.0f / sqrtf(row_sum<LPR>(q) / (float)N + eps);
  if (!live) return;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = (j * LPR + sl) * 4;
    if (i >= N) continue;
    const float4 gg = *reinterpret_cast<const float4*>(g + i), bb = *reinterpret_cast<const float4*>(b + i);
    const float4 o = make_float4((v[j].x - mu) * rs * gg.x + bb.x, (v[j].y - mu) * rs * gg.y + bb.y,
                                 (v[j].z - mu) * rs * gg.z + bb.z, (v[j].w - mu) * rs * gg.w + bb.w);
    if (y32) *reinterpret_cast<float4*>(y32 + (int64_t)row * N + i) = o;
    if (y16) *reinterpret_cast<uint2*>(y16 + (int64_t)row * N + i) = make_uint2(f32_to_bf16_bits(o.x) | (f32_to_bf16_bits(o.y) << 16),
                                                                                f32_to_bf16_bits(o.z) | (f32_to_bf16_bits(o.w) << 16));
  }
  if (sl == 0 && mean) { mean[row] = mu; rstd[row] = rs; }
}

template <int NJ, int LPR>
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ g, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ dx,
                                                                 float* __restrict__ part, int M, int N) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, sl = lane % LPR;
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), nslots = gridDim.x * 4;
  float4 gg[NJ], ag[NJ], ab[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = (j * LPR + sl) * 4;
    gg[j] = i < N ? *reinterpret_cast<const float4*>(g + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invN = 1.0f / (float)N;
  for (int base = slot * RPW; base < M; base += nslots * RPW) {   // wave-uniform trip count: the shuffles below run converged
    const int row = base + sub;
    const bool live = row < M;
    const int64_t off = (int64_t)(live ? row : 0) * N;
    const float mu = mean[live ? row : 0], rs = rstd[live ? row : 0];
    float4 xh[NJ], dg[NJ];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = (j * LPR + sl) * 4;
      const bool ok = live && i < N;
      const float4 xv = ok ? *reinterpret_cast<const float4*>(x + off + i) : make_float4(mu, mu, mu, mu);
      const float4 d = ok ? *reinterpret_cast<const float4*>(dy + off + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      dg[j] = make_float4(d.x * gg[j].x, d.y * gg[j].y, d.z * gg[j].z, d.w * gg[j].w);
      c1 += (dg[j].x + dg[j].y) + (dg[j].z + dg[j].w);
      c2 += (dg[j].x * xh[j].x + dg[j].y * xh[j].y) + (dg[j].z * xh[j].z + dg[j].w * xh[j].w);
      ag[j].x += d.x * xh[j].x; ag[j].y += d.y * xh[j].y; ag[j].z += d.z * xh[j].z; ag[j].w += d.w * xh[j].w;
      ab[j].x += d.x; ab[j].y += d.y; ab[j].z += d.z; ab[j].w += d.w;
    }
    c1 = row_sum<LPR>(c1) * invN; c2 = row_sum<LPR>(c2) * invN;
    if (dx && live) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = (j * LPR + sl) * 4;
        if (i < N)
          *reinterpret_cast<float4*>(dx + off + i) = make_float4(rs * (dg[j].x - c1 - xh[j].x * c2), rs * (dg[j].y - c1 - xh[j].y * c2),
                                                                 rs * (dg[j].z - c1 - xh[j].z * c2), rs * (dg[j].w - c1 - xh[j].w * c2));
      }
    }
  }
  if (!part) return;
  // the block's 4 * RPW row groups meet in LDS (gamma sums, then beta sums through the same buffer): one partial row per block
  __shared__ float red[4 * RPW][NJ * LPR * 4];
  const int wv = (threadIdx.x >> 6) * RPW + sub;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = (j * LPR + sl) * 4;
      if (i < N) *reinterpret_cast<float4*>(&red[wv][i]) = which ? ab[j] : ag[j];
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 4 * RPW; ++r) t += red[r][n];
      part[((int64_t)blockIdx.x * 2 + which) * N + n] = t;
    }
    __syncthreads();
  }
}
// NJ / LPR dispatch: N <= 128 -> two rows per wave; else the smallest NJ with 256 * NJ >= N
#define LN_ROWS_DISPATCH(N, CALL)                                                                   \
  do {                                                                                              \
    if ((N) <= 128) { CALL(1, 32); }                                                                \
    else if ((N) <= 256) { CALL(1, 64); }                                                           \
    else if ((N) <= 512) { CALL(2, 64); }                                                           \
    else if ((N) <= 768) { CALL(3, 64); }                                                           \
    else if ((N) <= 1024) { CALL(4, 64); }                                                          \
    else if ((N) <= 1536) { CALL(6, 64); }                                                          \
    else { CALL(8, 64); }                                                                           \
  } while (0)
static inline bool ln_rows_ok(int N, int relu) { return !relu && N % 4 == 0 && N <= 2048; }

// ------------------------------------------------------------------ pointwise gates
__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + __expf(-z)); }

#define EW_LOOP(n) for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void sigmoid_gate_fwd_kernel(const float* z, const float* v, float* out, int64_t n) {
  EW_LOOP(n) out[i] = (1.f / (1.f + expf(-z[i]))) * v[i];
}
__global__ void sigmoid_gate_bwd_kernel(const float* dout, const float* z, const float* v, float* dz, float* dv, int64_t n) {
  EW_LOOP(n) {
    float s = 1.f / (1.f + expf(-z[i]));
    float d = dout[i];
    dz[i] = d * v[i] * s * (1.f - s);
    dv[i] = d * s;
  }
}
__global__ void gated_mix_fwd_kernel(const float* z, const float* a, const float* q, float* out, int64_t n) {
  EW_LOOP(n) {
    float gt = 1.f / (1.f + expf(-z[i]));
    out[i] = gt * a[i] + (1.f - gt) * q[i];
  }
}
__global__ void gated_mix_bwd_kernel(const float* dout, const float* z, const float* a, const float* q, float* dz,
                                     float* da, float* dq, int64_t n) {
  EW_LOOP(n) {
    float gt = 1.f / (1.f + expf(-z[i]));
    float d = dout[i];
    dz[i] = d * (a[i] - q[i]) * gt * (1.f - gt);
    da[i] = d * gt;
    dq[i] = d * (1.f - gt);
  }
}
__global__ void metablock_gate_fwd_kernel(const float* V, const float* t1, const float* t2, float* out, int64_t n) {
  EW_LOOP(n) out[i] = 1.f / (1.f + expf(-(tanhf(V[i] * t1[i]) + t2[i])));
}
__global__ void metablock_gate_bwd_kernel(const float* dout, const float* V, const float* t1, const float* t2,
                                          float* dV, float* dt1, float* dt2, int64_t n) {
  EW_LOOP(n) {
    float u = tanhf(V[i] * t1[i]);
    float o = 1.f / (1.f + expf(-(u + t2[i])));
    float ds = dout[i] * o * (1.f - o);
    float dp = ds * (1.f - u * u);
    dt2[i] = ds;
    dV[i] = dp * t1[i];
    dt1[i] = dp * V[i];
  }
}

__global__ void dropout_fwd_kernel(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed,
                                   uint64_t offset) {
  const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  EW_LOOP(n) {
    uint64_t h = mix64(mix64(seed) ^ (offset + (uint64_t)i));
    float u = (float)(h >> 40) * (1.0f / 16777216.0f);
    uint8_t keep = u >= p ? 1 : 0;
    mask[i] = keep;
    y[i] = keep ? x[i] * scale : 0.f;
  }
}
// dropout on attention PROBABILITIES laid out [rows][L] (the unfused softmax-attention chain): the (row, key) generator of common.h, so the
// unfused chain drops exactly what the fused kernels drop
__global__ void attn_dropout_fwd_kernel(const float* x, float* y, uint8_t* mask, int64_t n, int L, float p, uint64_t seed, uint64_t offset) {
  const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  const AttnDropKey dk = attn_drop_key(seed, offset, p);
  const int Lh = (L + 1) >> 1;
  EW_LOOP(n) {
    const uint64_t row = (uint64_t)i / (uint64_t)L;
    const int key = (int)((uint64_t)i - row * (uint64_t)L);
    const uint8_t keep = attn_keep(dk, attn_row_base(row, Lh), key) ? 1 : 0;
    mask[i] = keep;
    y[i] = keep ? x[i] * scale : 0.f;
  }
}
__global__ void dropout_bwd_kernel(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p) {
  const float scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  EW_LOOP(n) dx[i] = mask[i] ? dy[i] * scale : 0.f;
}
__global__ void concat2_fwd_kernel(const float* a, const float* b, float* out, int M, int Na, int Nb) {
  const int N = Na + Nb;
  EW_LOOP((int64_t)M * N) {
    int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
    out[i] = n < Na ? a[(int64_t)m * Na + n] : b[(int64_t)m * Nb + n - Na];
  }
}
__global__ void concat2_bwd_kernel(const float* dout, float* da, float* db, int M, int Na, int Nb) {
  const int N = Na + Nb;
  EW_LOOP((int64_t)M * N) {
    int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
    if (n < Na) da[(int64_t)m * Na + n] = dout[i];
    else db[(int64_t)m * Nb + n - Na] = dout[i];
  }
}

// ------------------------------------------------------------------ small softmax attention
// one workgroup per (b, h); L*L scores in LDS; one wave per score row for the softmax.  Training-mode
// dropout on the attention probabilities (nn.MultiheadAttention(dropout=p) inside the TabTransformer's
// encoder layers) uses the same counter-based generator as the dropout op: element gi of call `offset`.
// (the generator: attn_keep / AttnDropKey in common.h, shared with the fused kernels of flash_attn*.hip)
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ o,
                                                            float* __restrict__ p, int L, int Dh, float drop_p,
                                                            uint64_t seed, uint64_t offset) {
  extern __shared__ float sc[];  // [L][L]
  const int64_t base = (int64_t)blockIdx.x * L * Dh;
  const float scale = 1.0f / sqrtf((float)Dh);
  for (int idx = threadIdx.x; idx < L * L; idx += 256) {
    int i = idx / L, j = idx - i * L;
    float s = 0.f;
    for (int d = 0; d < Dh; ++d) s += q[base + i * Dh + d] * k[base + j * Dh + d];
    sc[idx] = s * scale;
  }
  __syncthreads();
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wid; i < L; i += 4) {
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, sc[i * L + j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) { float e = expf(sc[i * L + j] - mx); sc[i * L + j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < L; j += 64) {
      float pv = sc[i * L + j] * inv;
      const int64_t gi = (int64_t)blockIdx.x * L * L + i * L + j;
      p[gi] = pv;   // the pure softmax is what backward needs; the dropout mask is recomputed there
      if (drop_p > 0.f) pv = attn_keep(attn_drop_key(seed, offset, drop_p), attn_row_base((uint64_t)blockIdx.x * L + i, (L + 1) >> 1), j) ? pv * (1.f / (1.f - drop_p)) : 0.f;
      sc[i * L + j] = pv;
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < L * Dh; idx += 256) {
    int i = idx / Dh, d = idx - i * Dh;
    float s = 0.f;
    for (int j = 0; j < L; ++j) s += sc[i * L + j] * v[base + j * Dh + d];
    o[base + idx] = s;
  }
}
__global__ __launch_bounds__(256) void attention_bwd_kernel(const float* __restrict__ dO, const float* __restrict__ q,
                                                            const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ p, float* __restrict__ dq,
                                                            float* __restrict__ dk, float* __restrict__ dv, int L,
                                                            int Dh, float drop_p, uint64_t seed, uint64_t offset) {
  extern __shared__ float ds[];  // [L][L]
  const int64_t base = (int64_t)blockIdx.x * L * Dh;
  const int64_t pbase = (int64_t)blockIdx.x * L * L;
  const float* pp = p + pbase;
  const float scale = 1.0f / sqrtf((float)Dh);
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (int idx = threadIdx.x; idx < L * L; idx += 256) {   // dropout factor of every probability
    const int i = idx / L, j = idx - i * L;
    ds[idx] = (drop_p > 0.f && !attn_keep(attn_drop_key(seed, offset, drop_p), attn_row_base((uint64_t)blockIdx.x * L + i, (L + 1) >> 1), j)) ? 0.f : dscale;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < L * Dh; idx += 256) {  // dV = P'^T dO,  P' = P * factor
    int j = idx / Dh, d = idx - j * Dh;
    float s = 0.f;
    for (int i = 0; i < L; ++i) s += pp[i * L + j] * ds[i * L + j] * dO[base + i * Dh + d];
    dv[base + idx] = s;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < L * L; idx += 256) {  // dP = (dO V^T) * factor
    int i = idx / L, j = idx - i * L;
    float s = 0.f;
    for (int d = 0; d < Dh; ++d) s += dO[base + i * Dh + d] * v[base + j * Dh + d];
    ds[idx] *= s;
  }
  __syncthreads();
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wid; i < L; i += 4) {
    float dot = 0.f;
    for (int j = lane; j < L; j += 64) dot += ds[i * L + j] * pp[i * L + j];
    dot = wave_sum(dot);
    for (int j = lane; j < L; j += 64) ds[i * L + j] = pp[i * L + j] * (ds[i * L + j] - dot) * scale;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < L * Dh; idx += 256) {
    int i = idx / Dh, d = idx - i * Dh;
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < L; ++j) {
      s1 += ds[i * L + j] * k[base + j * Dh + d];   // dq[i]
      s2 += ds[j * L + i] * q[base + j * Dh + d];   // dk[i]
    }
    dq[base + idx] = s1;
    dk[base + idx] = s2;
  }
}

// ------------------------------------------------------------------ small softmax attention, one WAVE per (batch, head)
// L <= 64 tokens, Dh = 32 or 64 (window attention of DaViT / Swin-style encoders, TabTransformer columns): lane i owns query
// row i -- its scores, its softmax and its output row live in that lane's registers, so the softmax needs no cross-lane step --
// and the key / value rows, which are the same for every lane, come through SCALAR loads (one s_load per 16 values, fed to v_fma
// as the SGPR operand): no LDS staging, no barrier.  q / k / v / o are addressed by (batch, head, token) element strides, so the
// kernel reads the packed [B, L, 3, H, Dh] output of a fused qkv Linear in place and writes the token-major [B, L, H, Dh] tensor
// the output projection wants.  The backward recomputes the probabilities from the saved row log-sum-exp (no [B, H, L, L] tensor
// is ever stored), forms dS / P' rows in the lanes that own the query rows, passes them through a wave-private LDS tile and
// re-reads them by COLUMN with lane j owning key row j for dK / dV.  fp32 throughout (exact-parity path: it replaces the
// one-workgroup-per-head kernel above, 507 / 605 us forward / backward on DaViT's 12 288 windows of 49 tokens).
struct AttnRowsArgs {
  int64_t qs_b, qs_h, qs_l;     // element strides of q / k / v (and dq / dk / dv)
  int64_t os_b, os_h, os_l;     // element strides of o (and dO)
  int nheads, H, L, Dh;
  float scale, drop_p;
  uint64_t seed, offset;
  int win_ws, win_nwy, win_nwx;   // > 0: "batch" b is window (image, wy, wx) of a [image][nwy*ws][nwx*ws] token grid, row l its token (l / ws, l % ws)
};
// Base offsets of (batch b, head h), the token offset of the lane's row and the stepping of the wave-uniform walk over the other
// rows (token offset t: ++t, and + wrap after every ws-th row).  Window mode reads / writes the tokens where they sit in the
// image-major activation: window partition / reverse are this index arithmetic, not copies (timm davit.py window_partition / window_reverse).
__device__ __forceinline__ void rows_geom(const AttnRowsArgs& p, int b, int h, int row, int64_t& qbase, int64_t& obase, int& lrow,
                                          int& ws, int& wrap) {
  if (p.win_ws > 0) {
    const int nw = p.win_nwy * p.win_nwx, img = b / nw, w = b - img * nw, wy = w / p.win_nwx, wx = w - wy * p.win_nwx;
    const int Wimg = p.win_nwx * p.win_ws;
    const int64_t t0 = ((int64_t)img * (p.win_nwy * p.win_ws) + wy * p.win_ws) * Wimg + wx * p.win_ws;
    qbase = t0 * p.qs_l + h * p.qs_h; obase = t0 * p.os_l + h * p.os_h;
    lrow = (row / p.win_ws) * Wimg + row % p.win_ws;
    ws = p.win_ws; wrap = Wimg - p.win_ws;
  } else {
    qbase = b * p.qs_b + h * p.qs_h; obase = b * p.os_b + h * p.os_h;
    lrow = row; ws = 0x7fffffff; wrap = 0;
  }
}
#define ROWS_STEP(t, x) do { ++t; if (++x == g_ws) { x = 0; t += g_wrap; } } while (0)

// (pointers are separate __restrict__ kernel parameters, not struct members: the no-alias guarantee is what lets the compiler turn
// the wave-uniform key / value reads into scalar loads.  The loops over the OTHER token are rolled -- unrolled, the scheduler hoists
// every row's s_load and spills 2 000 SGPRs -- so a lane's score row lives in a wave-private LDS column, one ds op per 32 FMAs.)
template <int DH>
__global__ __launch_bounds__(256) void attention_rows_fwd_kernel(const float* __restrict__ gq, const float* __restrict__ gk,
                                                                 const float* __restrict__ gv, float* __restrict__ gout,
                                                                 float* __restrict__ glse, const AttnRowsArgs p) {
  extern __shared__ float rows_lds[];                        // per wave: S [L][64]
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int head = blockIdx.x * 4 + wave;                    // wave-uniform
  if (head >= p.nheads) return;
  const int lane = threadIdx.x & 63;
  const int L = p.L;
  float* Ss = rows_lds + (size_t)wave * L * 64 + lane;
  const int b = head / p.H, h = head - b * p.H;
  const bool act = lane < L;
  const int row = act ? lane : 0;                            // idle lanes shadow row 0; nothing of theirs is stored
  int64_t qbase, obase;
  int lrow, g_ws, g_wrap;
  rows_geom(p, b, h, row, qbase, obase, lrow, g_ws, g_wrap);
  g_ws = __builtin_amdgcn_readfirstlane(g_ws); g_wrap = __builtin_amdgcn_readfirstlane(g_wrap);
  const float* __restrict__ qrow = gq + qbase + lrow * p.qs_l;
  const float* __restrict__ kb = gk + qbase;
  const float* __restrict__ vb = gv + qbase;
  float qv[DH];
#pragma unroll
  for (int d = 0; d < DH; d += 4) {
    const float4 t = *reinterpret_cast<const float4*>(qrow + d);
    qv[d] = t.x * p.scale; qv[d + 1] = t.y * p.scale; qv[d + 2] = t.z * p.scale; qv[d + 3] = t.w * p.scale;
  }
  float mx = -INFINITY;
  int jt = 0, jx = 0;
#pragma unroll 1
  for (int j = 0; j < L; ++j) {
    const float* __restrict__ kr = kb + jt * p.qs_l;         // uniform address: scalar loads
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) a = fmaf(qv[d], kr[d], a);
    Ss[j * 64] = a;
    mx = fmaxf(mx, a);
    ROWS_STEP(jt, jx);
  }
  const float keep_scale = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const AttnDropKey dkey = attn_drop_key(p.seed, p.offset, p.drop_p);
  float ov[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) ov[d] = 0.f;
  float sum = 0.f;
  jt = 0; jx = 0;
#pragma unroll 1
  for (int j = 0; j < L; ++j) {
    const float* __restrict__ vr = vb + jt * p.qs_l;
    float e = expf(Ss[j * 64] - mx);
    sum += e;                                                // the softmax denominator counts dropped probabilities too
    if (p.drop_p > 0.f) e = attn_keep(dkey, attn_row_base((uint64_t)head * L + row, (L + 1) >> 1), j) ? e * keep_scale : 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) ov[d] = fmaf(e, vr[d], ov[d]);
    ROWS_STEP(jt, jx);
  }
  const float inv = 1.f / sum;
  if (act) {
    if (glse) glse[(int64_t)head * L + lane] = mx + logf(sum);
    float* __restrict__ orow = gout + obase + lrow * p.os_l;
#pragma unroll
    for (int d = 0; d < DH; d += 4) *reinterpret_cast<float4*>(orow + d) = make_float4(ov[d] * inv, ov[d + 1] * inv, ov[d + 2] * inv, ov[d + 3] * inv);
  }
}

template <int DH>
__global__ __launch_bounds__(256) void attention_rows_bwd_kernel(const float* __restrict__ gq, const float* __restrict__ gk,
                                                                 const float* __restrict__ gv, const float* __restrict__ go,
                                                                 const float* __restrict__ gdO, const float* __restrict__ glse,
                                                                 float* __restrict__ gdq, float* __restrict__ gdk,
                                                                 float* __restrict__ gdv, const AttnRowsArgs p) {
  extern __shared__ float rows_lds[];                        // per wave: dS [L][L+1] | P' [L][L+1]
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int head = blockIdx.x * 4 + wave;
  if (head >= p.nheads) return;
  const int lane = threadIdx.x & 63;
  const int L = p.L, LP = L + 1;
  float* dSs = rows_lds + (size_t)wave * 2 * L * LP;
  float* Pps = dSs + L * LP;
  const int b = head / p.H, h = head - b * p.H;
  const bool act = lane < L;
  const int row = act ? lane : 0;
  int64_t qbase, obase;
  int lrow, g_ws, g_wrap;
  rows_geom(p, b, h, row, qbase, obase, lrow, g_ws, g_wrap);
  g_ws = __builtin_amdgcn_readfirstlane(g_ws); g_wrap = __builtin_amdgcn_readfirstlane(g_wrap);
  const float* __restrict__ qb = gq + qbase;
  const float* __restrict__ kb = gk + qbase;
  const float* __restrict__ vb = gv + qbase;
  const float* __restrict__ dob = gdO + obase;
  {
    // ---- phase A: lane i = query row i.  p_ij from the saved row log-sum-exp, dP'_ij = dO_i . v_j, delta_i = dO_i . o_i,
    // dS_ij = p_ij (f_ij dP'_ij - delta_i) scale  (f = dropout factor);  dq_i = sum_j dS_ij k_j
    const float* __restrict__ qrow = qb + lrow * p.qs_l;
    const float* __restrict__ dorow = dob + lrow * p.os_l;
    const float* __restrict__ orow = go + obase + lrow * p.os_l;
    float qv[DH], gvv[DH], acc[DH];
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      const float4 t = *reinterpret_cast<const float4*>(qrow + d);
      qv[d] = t.x * p.scale; qv[d + 1] = t.y * p.scale; qv[d + 2] = t.z * p.scale; qv[d + 3] = t.w * p.scale;
      const float4 u = *reinterpret_cast<const float4*>(dorow + d);
      gvv[d] = u.x; gvv[d + 1] = u.y; gvv[d + 2] = u.z; gvv[d + 3] = u.w;
      const float4 w = *reinterpret_cast<const float4*>(orow + d);
      delta = fmaf(u.x, w.x, fmaf(u.y, w.y, fmaf(u.z, w.z, fmaf(u.w, w.w, delta))));
      acc[d] = 0.f; acc[d + 1] = 0.f; acc[d + 2] = 0.f; acc[d + 3] = 0.f;
    }
    const float lse = glse[(int64_t)head * L + row];
    const float keep_scale = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    const AttnDropKey dkey = attn_drop_key(p.seed, p.offset, p.drop_p);
    int jt = 0, jx = 0;
#pragma unroll 1
    for (int j = 0; j < L; ++j) {
      const float* __restrict__ kr = kb + jt * p.qs_l;
      const float* __restrict__ vr = vb + jt * p.qs_l;
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) { a = fmaf(qv[d], kr[d], a); c = fmaf(gvv[d], vr[d], c); }
      const float pj = expf(a - lse);
      float f = 1.f;
      if (p.drop_p > 0.f) f = attn_keep(dkey, attn_row_base((uint64_t)head * L + row, (L + 1) >> 1), j) ? keep_scale : 0.f;
      const float ds = pj * (f * c - delta) * p.scale;
      if (act) { dSs[lane * LP + j] = ds; Pps[lane * LP + j] = pj * f; }
#pragma unroll
      for (int d = 0; d < DH; ++d) acc[d] = fmaf(ds, kr[d], acc[d]);
      ROWS_STEP(jt, jx);
    }
    if (act) {
      float* __restrict__ dqrow = gdq + qbase + lrow * p.qs_l;
#pragma unroll
      for (int d = 0; d < DH; d += 4) *reinterpret_cast<float4*>(dqrow + d) = make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]);
    }
  }
  // ---- phase B: lane j = key row j.  dk_j = sum_i dS_ij q_i, dv_j = sum_i P'_ij dO_i  (columns of the wave's LDS tiles)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float ak[DH], av[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { ak[d] = 0.f; av[d] = 0.f; }
  int it = 0, ix = 0;
#pragma unroll 1
  for (int i = 0; i < L; ++i) {
    const float dsi = dSs[i * LP + row], ppi = Pps[i * LP + row];
    const float* __restrict__ qr = qb + it * p.qs_l;         // uniform: scalar loads
    const float* __restrict__ gr = dob + it * p.os_l;
#pragma unroll
    for (int d = 0; d < DH; ++d) { ak[d] = fmaf(dsi, qr[d], ak[d]); av[d] = fmaf(ppi, gr[d], av[d]); }
    ROWS_STEP(it, ix);
  }
  if (act) {
    float* __restrict__ dkrow = gdk + qbase + lrow * p.qs_l;
    float* __restrict__ dvrow = gdv + qbase + lrow * p.qs_l;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
      *reinterpret_cast<float4*>(dkrow + d) = make_float4(ak[d], ak[d + 1], ak[d + 2], ak[d + 3]);
      *reinterpret_cast<float4*>(dvrow + d) = make_float4(av[d], av[d + 1], av[d + 2], av[d + 3]);
    }
  }
}

// ------------------------------------------------------------------ MD-Net fusion (multimodalMDNet.py:7-55,88-101)
// pooled[n][c] = mean_hw( sigmoid(z[n][c]) * f + sigmoid(tanh(f * t1[n][c]) + t2[n][c]) ),  f = feat[n][c][hw]:
// MetaNet channel gate + spatial MetaBlock + element-wise sum + global average pool in one pass over the
// feature map (NCHW fp32); one 64-lane wave per (n, c) plane.
__global__ __launch_bounds__(256) void mdnet_fuse_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ z,
                                                             const float* __restrict__ t1, const float* __restrict__ t2,
                                                             float* __restrict__ pooled, int64_t NC, int HW) {
  const int64_t nc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (nc >= NC) return;
  const int lane = threadIdx.x & 63;
  const float g = 1.f / (1.f + expf(-z[nc])), a = t1[nc], b = t2[nc];
  float s = 0.f;
  for (int i = lane; i < HW; i += 64) {
    float f = feat[nc * HW + i];
    s += g * f + 1.f / (1.f + expf(-(tanhf(f * a) + b)));
  }
  s = wave_sum(s);
  if (lane == 0) pooled[nc] = s / (float)HW;
}
__global__ __launch_bounds__(256) void mdnet_fuse_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ feat,
                                                             const float* __restrict__ z, const float* __restrict__ t1,
                                                             const float* __restrict__ t2, float* __restrict__ dfeat,
                                                             float* __restrict__ dz, float* __restrict__ dt1,
                                                             float* __restrict__ dt2, int64_t NC, int HW) {
  const int64_t nc = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (nc >= NC) return;
  const int lane = threadIdx.x & 63;
  const float g = 1.f / (1.f + expf(-z[nc])), a = t1[nc], b = t2[nc];
  const float dp = dpooled[nc] / (float)HW;
  float sf = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < HW; i += 64) {
    float f = feat[nc * HW + i];
    float u = tanhf(f * a);
    float o = 1.f / (1.f + expf(-(u + b)));
    float ds = dp * o * (1.f - o);        // d/d(u + b)
    float du = ds * (1.f - u * u);        // d/d(f * a)
    if (dfeat) dfeat[nc * HW + i] = dp * g + du * a;
    sf += f; s1 += du * f; s2 += ds;
  }
  sf = wave_sum(sf); s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) {
    dz[nc] = dp * sf * g * (1.f - g);
    dt1[nc] = s1;
    dt2[nc] = s2;
  }
}

// ------------------------------------------------------------------ embedding
__global__ void embedding_fwd_kernel(const float* table, const int64_t* ids, float* out, int B, int ncols, int card, int E) {
  EW_LOOP((int64_t)B * ncols * E) {
    int e = (int)(i % E);
    int64_t t = i / E;
    int col = (int)(t % ncols);
    int64_t id = ids[t];
    if (id < 0) id = 0;
    if (id >= card) id = card - 1;
    out[i] = table[((int64_t)col * card + id) * E + e];
  }
}
__global__ void embedding_bwd_kernel(const float* dout, const int64_t* ids, float* dtable, int B, int ncols, int card, int E) {
  EW_LOOP((int64_t)ncols * card * E) {  // deterministic: each table element sums its own batch rows
    int e = (int)(i % E);
    int64_t t = i / E;
    int id = (int)(t % card), col = (int)(t / card);
    float s = 0.f;
    for (int bb = 0; bb < B; ++bb)
      if (ids[(int64_t)bb * ncols + col] == id) s += dout[((int64_t)bb * ncols + col) * E + e];
    dtable[i] = s;
  }
}

// ------------------------------------------------------------------ custom-cnn direct kernels (NCHW)
__global__ void direct_conv_fwd_kernel(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H,
                                       int W, int Cout, int kh, int kw, int stride, int pad, int OH, int OW, int relu) {
  EW_LOOP((int64_t)N * Cout * OH * OW) {
    int ow = (int)(i % OW);
    int64_t t = i / OW;
    int oh = (int)(t % OH); t /= OH;
    int co = (int)(t % Cout);
    int n = (int)(t / Cout);
    float s = b ? b[co] : 0.f;
    for (int c = 0; c < Cin; ++c)
      for (int r = 0; r < kh; ++r) {
        int h = oh * stride - pad + r;
        if ((unsigned)h >= (unsigned)H) continue;
        for (int q = 0; q < kw; ++q) {
          int ww = ow * stride - pad + q;
          if ((unsigned)ww >= (unsigned)W) continue;
          s += x[(((int64_t)n * Cin + c) * H + h) * W + ww] * w[((co * Cin + c) * kh + r) * kw + q];
        }
      }
    if (relu) s = fmaxf(s, 0.f);
    y[i] = s;
  }
}
// one workgroup per weight element (and one per bias): block-wide reduction over (n, oh, ow)
__global__ __launch_bounds__(256) void direct_conv_bwd_kernel(const float* dy, const float* x, const float* y_relu,
                                                              float* dw, float* db, int N, int Cin, int H, int W,
                                                              int Cout, int kh, int kw, int stride, int pad, int OH,
                                                              int OW) {
  __shared__ float red[4];
  const int nw = Cout * Cin * kh * kw;
  const int id = blockIdx.x;
  const bool is_bias = id >= nw;
  int co, c = 0, r = 0, q = 0;
  if (is_bias) co = id - nw;
  else { q = id % kw; int t = id / kw; r = t % kh; t /= kh; c = t % Cin; co = t / Cin; }
  float s = 0.f;
  const int64_t total = (int64_t)N * OH * OW;
  for (int64_t i = threadIdx.x; i < total; i += 256) {
    int ow = (int)(i % OW);
    int64_t t = i / OW;
    int oh = (int)(t % OH);
    int n = (int)(t / OH);
    int64_t yo = (((int64_t)n * Cout + co) * OH + oh) * OW + ow;
    float d = dy[yo];
    if (y_relu && !(y_relu[yo] > 0.f)) d = 0.f;
    if (is_bias) s += d;
    else {
      int h = oh * stride - pad + r, ww = ow * stride - pad + q;
      if ((unsigned)h < (unsigned)H && (unsigned)ww < (unsigned)W) s += d * x[(((int64_t)n * Cin + c) * H + h) * W + ww];
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = red[0] + red[1] + red[2] + red[3];
    if (is_bias) { if (db) db[co] = tot; } else dw[id] = tot;
  }
}
__global__ __launch_bounds__(64) void pool_gap_fwd_kernel(const float* x, float* y, int32_t* idx, int C, int H, int W,
                                                          int k, int PH, int PW) {
  const int nc = blockIdx.x;  // n*C + c
  const float* xp = x + (int64_t)nc * H * W;
  float s = 0.f;
  for (int pi = threadIdx.x; pi < PH * PW; pi += 64) {
    int ph = pi / PW, pw = pi - ph * PW;
    float best = -INFINITY;
    int bi = (ph * k) * W + pw * k;
    for (int r = 0; r < k; ++r)
      for (int q = 0; q < k; ++q) {
        int h = ph * k + r, w = pw * k + q;
        float v = xp[h * W + w];
        if (v > best || v != v) { best = v; bi = h * W + w; }
      }
    idx[(int64_t)nc * PH * PW + pi] = bi;
    s += best;
  }
  s = wave_sum(s);
  if (threadIdx.x == 0) y[nc] = s / (float)(PH * PW);
}
__global__ void pool_gap_bwd_kernel(const float* dy, const int32_t* idx, float* dx, int64_t NC, int HW, int PHW) {
  EW_LOOP(NC * PHW) {
    int64_t nc = i / PHW;
    dx[nc * HW + idx[i]] = dy[nc] / (float)PHW;
  }
}

// ------------------------------------------------------------------ row softmax / GELU (BERT-style encoder layers)
// y[r][j] = softmax_j(x[r][j] * scale + mask_add[b][j]), b = r / rows_per_batch; one wave per row
// bias (optional): [rows_per_batch][L] added to every batch's scores (per-head relative-position bias of BEiT)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask_add,
                                                          const float* __restrict__ bias, float* __restrict__ y, int64_t rows,
                                                          int L, int64_t rows_per_batch, float scale, int causal) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + r * L;
  const float* mk = mask_add ? mask_add + (r / rows_per_batch) * L : nullptr;
  const float* bs = bias ? bias + (r % rows_per_batch) * L : nullptr;
  const int jend = causal ? (int)(r % L) + 1 : L;      // causal (GPT-2): query i attends keys 0..i (square score matrices)
  float mx = -INFINITY;
  for (int j = lane; j < jend; j += 64) mx = fmaxf(mx, xr[j] * scale + (mk ? mk[j] : 0.f) + (bs ? bs[j] : 0.f));
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < jend; j += 64) sum += expf(xr[j] * scale + (mk ? mk[j] : 0.f) + (bs ? bs[j] : 0.f) - mx);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  for (int j = lane; j < L; j += 64)
    y[r * L + j] = j < jend ? expf(xr[j] * scale + (mk ? mk[j] : 0.f) + (bs ? bs[j] : 0.f) - mx) * inv : 0.f;
}
// y = x + gamma[c] * b  (LayerScale residual), and its pieces backward
__global__ void scale_add_fwd_kernel(const float* __restrict__ x, const float* __restrict__ b, const float* __restrict__ gamma,
                                     float* __restrict__ y, int64_t n, int C) {
  EW_LOOP(n) y[i] = x[i] + gamma[i % C] * b[i];
}
__global__ void scale_mul_kernel(const float* __restrict__ dy, const float* __restrict__ v, float* __restrict__ out, int64_t n, int C,
                                 int per_channel) {
  EW_LOOP(n) out[i] = dy[i] * (per_channel ? v[i % C] : v[i]);
}
// out[b][e] = mean over tokens [start, L) of x[b][t][e]; backward spreads dout / (L - start)
__global__ void token_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int L, int E, int start) {
  EW_LOOP((int64_t)B * E) {
    const int e = (int)(i % E); const int64_t b = i / E;
    float s2 = 0.f;
    for (int t = start; t < L; ++t) s2 += x[(b * L + t) * E + e];
    out[i] = s2 / (float)(L - start);
  }
}
__global__ void token_mean_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int B, int L, int E, int start) {
  EW_LOOP((int64_t)B * L * E) {
    const int e = (int)(i % E); const int64_t t2 = i / E;
    const int t = (int)(t2 % L); const int64_t b = t2 / L;
    dx[i] = t >= start ? dout[b * E + e] / (float)(L - start) : 0.f;
  }
}
// dx = (dy - sum_j dy*y) * y * scale
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          float* __restrict__ dx, int64_t rows, int L, float scale) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float dot = 0.f;
  for (int j = lane; j < L; j += 64) dot += dy[r * L + j] * y[r * L + j];
  dot = wave_sum(dot);
  for (int j = lane; j < L; j += 64) dx[r * L + j] = (dy[r * L + j] - dot) * y[r * L + j] * scale;
}
// GPT-2's "gelu_new": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
__global__ void gelu_tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  EW_LOOP(n) { const float v = x[i]; y[i] = 0.5f * v * (1.f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v))); }
}
__global__ void gelu_tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t n) {
  EW_LOOP(n) {
    const float v = x[i];
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    const float t = tanhf(u);
    const float du = 0.7978845608028654f * (1.f + 3.f * 0.044715f * v * v);
    dx[i] = dy[i] * (0.5f * (1.f + t) + 0.5f * v * (1.f - t * t) * du);
  }
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n, int64_t nb) {
  EW_LOOP(n) y[i] = a[i] + b[i % nb];   // b broadcasts over the leading dimension when nb < n
}
__global__ void add4_inplace_kernel(float* __restrict__ y, const float* __restrict__ b, int64_t n4) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<float4*>(y)[i];
    const float4 c = reinterpret_cast<const float4*>(b)[i];
    a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
    reinterpret_cast<float4*>(y)[i] = a;
  }
}
__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  EW_LOOP(n) y[i] = 0.5f * x[i] * (1.f + erff(x[i] * 0.70710678118654752f));
}
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
// h16 [rows][cols_pad] (bf16, zero pad columns) = gelu(z [rows][cols]): the operand of the MLP's second Linear straight from the
// pre-activation -- gelu(z) is never written in fp32 (4 values per thread)
__global__ void gelu_to_bf16_kernel(const float* __restrict__ z, bf16_t* __restrict__ h16, int64_t rows, int cols, int cols_pad) {
  const int cpr = cols_pad / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows * cpr; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cpr;
    const int c = (int)(i - r * cpr) * 4;
    uint2 o = make_uint2(0u, 0u);
    if (c < cols) {
      const float4 v = *reinterpret_cast<const float4*>(z + r * cols + c);
      o = make_uint2(f32_to_bf16_bits(gelu_exact(v.x)) | (f32_to_bf16_bits(gelu_exact(v.y)) << 16),
                     f32_to_bf16_bits(gelu_exact(v.z)) | (f32_to_bf16_bits(gelu_exact(v.w)) << 16));
    }
    *reinterpret_cast<uint2*>(h16 + r * cols_pad + c) = o;
  }
}
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t n) {
  EW_LOOP(n) {
    const float v = x[i];
    const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
    dx[i] = dy[i] * (cdf + v * pdf);
  }
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

// x16_keep (optional): the bf16 copy of x the GEMM consumes is written THERE ([M][mmskin_linear_x16_pitch]) instead of into library
// scratch, so the caller can hand it back to the backward's weight-gradient GEMM (no second conversion of x, half the saved bytes).
// x16_in (optional, instead of x): the operand is already there in bf16 ([M][mmskin_linear_x16_pitch], zero pad columns).
// res (optional, bf16 large-GEMM path only): y = res + act(x w^T + b), res fp32 [M][N] -- the residual add of a transformer block.
static int linear_forward_impl(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int relu, void* x16_keep,
                               void* stream, const void* x16_in = nullptr, const float* res = nullptr) {
  ARG_CHECK(!res || (linear_bf16() && (linear_big_padded(M, K, N) || linear_big(M, K, N))),
            "linear_forward: the fused residual needs the bf16 large-GEMM path (shape / mode)");
  ARG_CHECK((x || x16_in) && w && y && M > 0 && K > 0 && N > 0, "linear_forward: bad argument");
  ARG_CHECK(!x16_in || (linear_bf16() && (linear_big_padded(M, K, N) || linear_big(M, K, N))),
            "linear_forward: a bf16 operand needs the bf16 large-GEMM path (shape / mode)");
  ARG_CHECK(relu >= 0 && relu <= 2, "linear_forward: activation %d (0 none, 1 ReLU, 2 exact GELU)", relu);
  if (linear_bf16() && linear_big_padded(M, K, N)) {
    const int Kp = pad64(K), Np = pad64(N);
    ConvShape s = {M, 1, 1, Kp, Np, 1, 1, 1, 0};
    const size_t xb = align_up((size_t)M * Kp * 2, 256), wb = align_up((size_t)Np * Kp * 2, 256), yb = align_up((size_t)M * Np * 2, 256);
    unsigned char* sc = reinterpret_cast<unsigned char*>(head_scratch(xb + wb + yb));
    if (!sc) { mmskin_set_error("linear_forward: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    const bf16_t* x16 = reinterpret_cast<const bf16_t*>(x16_in);
    bf16_t* w16 = reinterpret_cast<bf16_t*>(sc + xb); bf16_t* y16 = reinterpret_cast<bf16_t*>(sc + xb + wb);
    int rc;
    if (!x16) {
      bf16_t* xc = x16_keep ? reinterpret_cast<bf16_t*>(x16_keep) : reinterpret_cast<bf16_t*>(sc);
      if ((rc = cvt_to_bf16_pad(x, xc, M, K, M, Kp, ST(stream)))) return rc;
      x16 = xc;
    }
    if ((rc = cvt_to_bf16_pad(w, w16, N, K, Np, Kp, ST(stream)))) return rc;
    if ((rc = launch_conv_fwd<bf16_t>(s, x16, w16, y16, nullptr, nullptr, ST(stream), nullptr))) return rc;
    return unpad_bias_act(y16, b, y, M, N, Np, relu, ST(stream), res);
  }
  if (linear_big(M, K, N) && linear_bf16()) {
    // bf16 operands, fp32 accumulate; bias + activation + the widening to fp32 all happen in the GEMM epilogue
    ConvShape s = {M, 1, 1, K, N, 1, 1, 1, 0};
    const size_t xb = align_up((size_t)M * K * 2, 256), wb = align_up((size_t)N * K * 2, 256);
    unsigned char* sc = reinterpret_cast<unsigned char*>(head_scratch(xb + wb));
    if (!sc) { mmskin_set_error("linear_forward: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    const bf16_t* x16 = reinterpret_cast<const bf16_t*>(x16_in);
    bf16_t* w16 = reinterpret_cast<bf16_t*>(sc + xb);
    int rc;
    if (!x16) {
      bf16_t* xc = x16_keep ? reinterpret_cast<bf16_t*>(x16_keep) : reinterpret_cast<bf16_t*>(sc);
      if ((rc = cvt_to_bf16(x, xc, (int64_t)M * K, ST(stream)))) return rc;
      x16 = xc;
    }
    if ((rc = cvt_to_bf16(w, w16, (int64_t)N * K, ST(stream)))) return rc;
    const bool res_epi = res && N % 128 == 0;   // the residual epilogue stores whole 128-column tiles; other widths add in a pass of their own
    FwdFuse f; f.bias = b; f.relu = relu == 1; f.gelu = relu == 2; f.out_f32 = y; f.res_f32 = res_epi ? res : nullptr;
    if ((rc = launch_conv_fwd<bf16_t>(s, x16, w16, reinterpret_cast<bf16_t*>(y), nullptr, nullptr, ST(stream), &f))) return rc;
    if (res && !res_epi) {
      const int64_t n4 = (int64_t)M * N / 4;
      int64_t blocks = (n4 + 255) / 256;
      hipLaunchKernelGGL(add4_inplace_kernel, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(256), 0, ST(stream), y, res, n4);
      HIP_CHECK_RET(hipGetLastError());
    }
    return MMSKIN_OK;
  }
  if (linear_big(M, K, N)) {   // tokens x hidden GEMMs of the text encoders: the exact-f32 implicit-GEMM kernel as a 1x1 conv
    ConvShape s = {M, 1, 1, K, N, 1, 1, 1, 0};
    FwdFuse f; f.bias = b; f.relu = relu == 1; f.gelu = relu == 2;
    return launch_conv_fwd<float>(s, x, w, y, nullptr, nullptr, ST(stream), (b || relu) ? &f : nullptr);
  }
  int rc = gemm_f32(x, w, y, b, M, N, K, K, 1, K, 1, N, relu == 1, ST(stream));
  if (rc || relu != 2) return rc;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid1d((int64_t)M * N)), dim3(256), 0, ST(stream), y, y, (int64_t)M * N);   // in place: element i only
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

int mmskin_linear_forward(const float* x, const float* w, const float* b, float* y, int M, int K, int N, int relu,
                          void* stream) {
  return linear_forward_impl(x, w, b, y, M, K, N, relu, nullptr, stream);
}
// Row pitch (elements) of the bf16 operand copy the current mode's large-GEMM path makes of an [M][K] input, 0 when this shape / mode
// does not take that path (then there is nothing to keep).
int mmskin_linear_x16_pitch(int M, int K, int N) {
  if (!linear_bf16() || M <= 0 || K <= 0 || N <= 0) return 0;
  if (linear_big_padded(M, K, N)) return pad64(K);
  if (linear_big(M, K, N)) return K;
  return 0;
}
// y = act(x16 w^T + b) with the operand already in bf16 (e.g. written by mmskin_gelu_forward_bf16): no conversion pass
int mmskin_linear_forward_x16(const void* x16, const float* w, const float* b, const float* res, float* y, int M, int K, int N, int relu,
                              void* stream) {
  ARG_CHECK(x16, "linear_forward_x16: null operand");
  return linear_forward_impl(nullptr, w, b, y, M, K, N, relu, nullptr, stream, x16, res);
}
int mmskin_gelu_forward_bf16(const float* z, void* h16, int64_t rows, int cols, int cols_pad, void* stream) {
  ARG_CHECK(z && h16 && rows > 0 && cols > 0 && cols % 4 == 0 && cols_pad % 4 == 0 && cols_pad >= cols, "gelu_forward_bf16: bad argument");
  const int64_t n = rows * (cols_pad / 4);
  int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(gelu_to_bf16_kernel, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(256), 0, ST(stream), z,
                     reinterpret_cast<bf16_t*>(h16), rows, cols, cols_pad);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_linear_forward_keep(const float* x, const float* w, const float* b, const float* res, float* y, void* x16_keep, int M, int K,
                               int N, int relu, void* stream) {
  ARG_CHECK(x16_keep && mmskin_linear_x16_pitch(M, K, N) > 0, "linear_forward_keep: no bf16 operand copy for this shape / mode");
  return linear_forward_impl(x, w, b, y, M, K, N, relu, x16_keep, stream, nullptr, res);
}

// Linear with bf16 tensors at either end (the inference lane of the transformer encoders in bf16-operand mode): x and / or y may be
// bf16, so consecutive layers hand activations over without fp32 <-> bf16 conversion passes.  Shapes off the large-GEMM path (or
// fp32 operand mode) fall back to the fp32 entry point through scratch conversions.
int mmskin_linear_forward_ex(const void* x, int x_dtype, const float* w, const float* b, void* y, int y_dtype, int M, int K, int N,
                             int act, void* stream) {
  ARG_CHECK(x && w && y && M > 0 && K > 0 && N > 0, "linear_forward_ex: bad argument");
  ARG_CHECK((x_dtype == 0 || x_dtype == 1) && (y_dtype == 0 || y_dtype == 1) && act >= 0 && act <= 2, "linear_forward_ex: dtype / activation");
  hipStream_t st = ST(stream);
  int rc;
  if (linear_big(M, K, N) && linear_bf16()) {
    ConvShape s = {M, 1, 1, K, N, 1, 1, 1, 0};
    const size_t xb = x_dtype == 1 ? 0 : align_up((size_t)M * K * 2, 256), wb = align_up((size_t)N * K * 2, 256);
    unsigned char* sc = reinterpret_cast<unsigned char*>(head_scratch(xb + wb));
    if (!sc) { mmskin_set_error("linear_forward_ex: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    const bf16_t* x16 = reinterpret_cast<const bf16_t*>(x);
    if (x_dtype == 0) {
      if ((rc = cvt_to_bf16(reinterpret_cast<const float*>(x), reinterpret_cast<bf16_t*>(sc), (int64_t)M * K, st))) return rc;
      x16 = reinterpret_cast<const bf16_t*>(sc);
    }
    bf16_t* w16 = reinterpret_cast<bf16_t*>(sc + xb);
    if ((rc = cvt_to_bf16(w, w16, (int64_t)N * K, st))) return rc;
    FwdFuse f; f.bias = b; f.relu = act == 1; f.gelu = act == 2;
    if (y_dtype == 0) f.out_f32 = reinterpret_cast<float*>(y);
    return launch_conv_fwd<bf16_t>(s, x16, w16, reinterpret_cast<bf16_t*>(y), nullptr, nullptr, st, (b || act || y_dtype == 0) ? &f : nullptr);
  }
  const size_t xb = x_dtype == 1 ? align_up((size_t)M * K * 4, 256) : 0, yb = y_dtype == 1 ? align_up((size_t)M * N * 4, 256) : 0;
  // (the fp32 entry point may itself use head_scratch: keep these conversions in a region of their own behind it)
  float* side = head_scratch(xb + yb, 1);
  if (!side) { mmskin_set_error("linear_forward_ex: scratch allocation failed"); return MMSKIN_ERR_HIP; }
  const float* xf = reinterpret_cast<const float*>(x);
  if (x_dtype == 1) {
    if ((rc = cvt_to_f32(reinterpret_cast<const bf16_t*>(x), side, (int64_t)M * K, st))) return rc;
    xf = side;
  }
  float* yf = y_dtype == 1 ? reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(side) + xb) : reinterpret_cast<float*>(y);
  if ((rc = mmskin_linear_forward(xf, w, b, yf, M, K, N, act, stream))) return rc;
  if (y_dtype == 1) return cvt_to_bf16(yf, reinterpret_cast<bf16_t*>(y), (int64_t)M * N, st);
  return MMSKIN_OK;
}

// The lane Linear with everything a frozen transformer block hangs on its GEMMs fused into the epilogue:
//   y = residual + gamma * dropout(act(x w^T + b))          (each of residual / gamma / dropout optional)
// x fp32 or bf16; w fp32 (converted per call) or bf16 (a cached conversion of a frozen weight: no per-step conversion pass);
// residual fp32 [M][N]; with any of the three, y must be fp32 (the residual stream).  bf16-operand mode and the large-GEMM shape
// class only: the op has no fp32 formulation (the caller composes the separate ops instead).
int mmskin_linear_lane(const void* x, int x_dtype, const void* w, int w_dtype, const float* b, const float* gamma,
                       const float* residual, float drop_p, uint64_t seed, uint64_t offset, void* y, int y_dtype, int M, int K,
                       int N, int act, void* stream) {
  ARG_CHECK(x && w && y && M > 0 && K > 0 && N > 0, "linear_lane: bad argument");
  ARG_CHECK((x_dtype == 0 || x_dtype == 1) && (w_dtype == 0 || w_dtype == 1) && (y_dtype == 0 || y_dtype == 1) && act >= 0 && act <= 2,
            "linear_lane: dtype / activation");
  ARG_CHECK(drop_p >= 0.f && drop_p < 1.f, "linear_lane: dropout probability %f", (double)drop_p);
  ARG_CHECK(linear_big(M, K, N) && linear_bf16(), "linear_lane: M=%d K=%d N=%d is off the large bf16 GEMM path (rows >= 2048, 64-multiple widths, "
            "MMSKIN_LINEAR_DTYPE=bf16)", M, K, N);
  const bool tr = gamma || residual || drop_p > 0.f;
  ARG_CHECK(!tr || (y_dtype == 0 && N % 128 == 0), "linear_lane: residual / layer scale / dropout need an fp32 result and N %% 128 == 0");
  hipStream_t st = ST(stream);
  int rc;
  ConvShape s = {M, 1, 1, K, N, 1, 1, 1, 0};
  const size_t xb = x_dtype == 1 ? 0 : align_up((size_t)M * K * 2, 256), wb = w_dtype == 1 ? 0 : align_up((size_t)N * K * 2, 256);
  unsigned char* sc = (xb + wb) ? reinterpret_cast<unsigned char*>(head_scratch(xb + wb)) : nullptr;
  if ((xb + wb) && !sc) { mmskin_set_error("linear_lane: scratch allocation failed"); return MMSKIN_ERR_HIP; }
  const bf16_t* x16 = reinterpret_cast<const bf16_t*>(x);
  if (x_dtype == 0) {
    if ((rc = cvt_to_bf16(reinterpret_cast<const float*>(x), reinterpret_cast<bf16_t*>(sc), (int64_t)M * K, st))) return rc;
    x16 = reinterpret_cast<const bf16_t*>(sc);
  }
  const bf16_t* w16 = reinterpret_cast<const bf16_t*>(w);
  if (w_dtype == 0) {
    if ((rc = cvt_to_bf16(reinterpret_cast<const float*>(w), reinterpret_cast<bf16_t*>(sc + xb), (int64_t)N * K, st))) return rc;
    w16 = reinterpret_cast<const bf16_t*>(sc + xb);
  }
  FwdFuse f; f.bias = b; f.relu = act == 1; f.gelu = act == 2;
  if (y_dtype == 0) f.out_f32 = reinterpret_cast<float*>(y);
  f.gamma = gamma; f.res_f32 = residual; f.drop_p = drop_p; f.seed = seed; f.offset = offset;
  return launch_conv_fwd<bf16_t>(s, x16, w16, reinterpret_cast<bf16_t*>(y), nullptr, nullptr, st, (b || act || y_dtype == 0 || tr) ? &f : nullptr);
}

static int linear_backward_impl(const float* dy, const float* x, const float* w, const float* y_relu, const float* z_gelu, float* dy_scratch,
                                float* dx, float* dw, float* db, int M, int K, int N, void* stream, const void* x16_kept = nullptr) {
  ARG_CHECK(dy && M > 0 && K > 0 && N > 0, "linear_backward: bad argument");
  ARG_CHECK(!(y_relu && z_gelu), "linear_backward: one activation");
  hipStream_t st = ST(stream);
  const float* g = dy;
  const bool bf16_gemm = linear_bf16() && (linear_big_padded(M, K, N) || linear_big(M, K, N));
  ARG_CHECK(!x16_kept || bf16_gemm, "linear_backward: the kept bf16 operand belongs to the bf16 large-GEMM path (mode changed since the forward?)");
  if (z_gelu && !bf16_gemm) {   // no conversion pass to fold the GELU derivative into: its own pass
    ARG_CHECK(dy_scratch, "linear_backward: dy_scratch required with z_gelu");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid1d((int64_t)M * N)), dim3(256), 0, st, dy, z_gelu, dy_scratch, (int64_t)M * N);
    HIP_CHECK_RET(hipGetLastError());
    g = dy_scratch; z_gelu = nullptr;
  }
  if (y_relu) {
    ARG_CHECK(dy_scratch, "linear_backward: dy_scratch required with y_relu");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid1d((int64_t)M * N)), dim3(256), 0, st, dy, y_relu, dy_scratch, (int64_t)M * N);
    HIP_CHECK_RET(hipGetLastError());
    g = dy_scratch;
  }
  int rc;
  if (linear_bf16() && linear_big_padded(M, K, N)) {
    const int Kp = pad64(K), Np = pad64(N);
    ConvShape s = {M, 1, 1, Kp, Np, 1, 1, 1, 0};
    const size_t gb = align_up((size_t)M * Np * 2, 256), xb = align_up((size_t)M * Kp * 2, 256), wb = align_up((size_t)Np * Kp * 2, 256);
    const size_t slb = align_up(conv_wgrad_slab_bytes(s), 256);
    unsigned char* sc = reinterpret_cast<unsigned char*>(head_scratch(gb + xb + wb + slb + colsum4_part_bytes(N)));
    if (!sc) { mmskin_set_error("linear_backward: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    bf16_t* g16 = reinterpret_cast<bf16_t*>(sc); bf16_t* t16 = reinterpret_cast<bf16_t*>(sc + gb); bf16_t* w16 = reinterpret_cast<bf16_t*>(sc + gb + xb);
    float* slab = reinterpret_cast<float*>(sc + gb + xb + wb);
    if (db || z_gelu) {   // bias gradient (and the GELU derivative) from the pass that converts dy
      if ((rc = colsum4(g, db, M, N, reinterpret_cast<float*>(sc + gb + xb + wb + slb), g16, Np, st, z_gelu))) return rc;
      db = nullptr;
    } else if ((rc = cvt_to_bf16_pad(g, g16, M, N, M, Np, st))) return rc;
    if (dx) {
      ARG_CHECK(w, "linear_backward: w required for dx");
      HIP_CHECK_RET(hipMemsetAsync(w16, 0, (size_t)Np * Kp * 2, st));
      hipLaunchKernelGGL(transpose_f32_to_bf16_pitch_kernel, dim3(ceil_div(K, 32), ceil_div(N, 32)), dim3(32, 8), 0, st, w, w16, N, K, Np);
      HIP_CHECK_RET(hipGetLastError());
      if ((rc = launch_conv_dgrad<bf16_t>(s, g16, w16, t16, (const bf16_t*)nullptr, st))) return rc;   // dx [M][Kp] in bf16, un-padded while widened
      if ((rc = unpad_bias_act(t16, nullptr, dx, M, K, Kp, 0, st))) return rc;
    }
    if (dw) {
      ARG_CHECK(x || x16_kept, "linear_backward: x required for dw");
      const bf16_t* xo = reinterpret_cast<const bf16_t*>(x16_kept);
      if (!xo) { if ((rc = cvt_to_bf16_pad(x, t16, M, K, M, Kp, st))) return rc; xo = t16; }
      if ((rc = launch_conv_wgrad<bf16_t>(s, g16, xo, slab, dw, st, N, K))) return rc;   // reduces straight into the unpadded [N][K] gradient
    }
    if (db && (rc = colsum(g, db, M, N, st))) return rc;
    return MMSKIN_OK;
  }
  if (linear_big(M, K, N) && linear_bf16()) {
    ConvShape s = {M, 1, 1, K, N, 1, 1, 1, 0};
    const size_t gb = align_up((size_t)M * N * 2, 256), xb = align_up((size_t)M * K * 2, 256), wb = align_up((size_t)N * K * 2, 256);
    const size_t slb = align_up(conv_wgrad_slab_bytes(s), 256);
    unsigned char* sc = reinterpret_cast<unsigned char*>(head_scratch(gb + xb + wb + slb + colsum4_part_bytes(N)));
    if (!sc) { mmskin_set_error("linear_backward: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    bf16_t* g16 = reinterpret_cast<bf16_t*>(sc); bf16_t* t16 = reinterpret_cast<bf16_t*>(sc + gb); bf16_t* w16 = reinterpret_cast<bf16_t*>(sc + gb + xb);
    float* slab = reinterpret_cast<float*>(sc + gb + xb + wb);
    if (db || z_gelu) {   // bias gradient (and the GELU derivative) from the pass that converts dy
      if ((rc = colsum4(g, db, M, N, reinterpret_cast<float*>(sc + gb + xb + wb + slb), g16, N, st, z_gelu))) return rc;
      db = nullptr;
    } else if ((rc = cvt_to_bf16(g, g16, (int64_t)M * N, st))) return rc;
    if (dx) {
      ARG_CHECK(w, "linear_backward: w required for dx");
      hipLaunchKernelGGL(transpose_f32_to_bf16_kernel, dim3(ceil_div(K, 32), ceil_div(N, 32)), dim3(32, 8), 0, st, w, w16, N, K);
      HIP_CHECK_RET(hipGetLastError());
      // dx = g w as a FORWARD 1x1 conv over the transposed weight (w16 [K][N] is its [Cout][Cin] layout): the light epilogue writes
      // fp32 straight into dx -- no bf16 round trip and widening pass (10 us x 57 per DaViT step)
      static const bool dx_f32 = [] { const char* v = getenv("MMSKIN_LINEAR_DX_F32"); return !v || atoi(v) != 0; }();
      if (dx_f32) {
        ConvShape sd = {M, 1, 1, N, K, 1, 1, 1, 0};
        FwdFuse f; f.out_f32 = dx;
        if ((rc = launch_conv_fwd<bf16_t>(sd, g16, w16, reinterpret_cast<bf16_t*>(dx), nullptr, nullptr, st, &f))) return rc;
      } else {
        if ((rc = launch_conv_dgrad<bf16_t>(s, g16, w16, t16, (const bf16_t*)nullptr, st))) return rc;   // dx in bf16, then widened
        if ((rc = cvt_to_f32(t16, dx, (int64_t)M * K, st))) return rc;
      }
    }
    if (dw) {
      ARG_CHECK(x || x16_kept, "linear_backward: x required for dw");
      const bf16_t* xo = reinterpret_cast<const bf16_t*>(x16_kept);
      if (!xo) { if ((rc = cvt_to_bf16(x, t16, (int64_t)M * K, st))) return rc; xo = t16; }
      if ((rc = launch_conv_wgrad<bf16_t>(s, g16, xo, slab, dw, st))) return rc;
    }
    if (db && (rc = colsum(g, db, M, N, st))) return rc;
    return MMSKIN_OK;
  }
  if (linear_big(M, K, N)) {
    ConvShape s = {M, 1, 1, K, N, 1, 1, 1, 0};
    const size_t wt_bytes = align_up((size_t)N * K * sizeof(float), 256);
    float* scratch = head_scratch(wt_bytes + conv_wgrad_slab_bytes(s));
    if (!scratch) { mmskin_set_error("linear_backward: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    if (dx) {
      ARG_CHECK(w, "linear_backward: w required for dx");
      hipLaunchKernelGGL(transpose_f32_kernel, dim3(ceil_div(K, 32), ceil_div(N, 32)), dim3(32, 8), 0, st, w, scratch, N, K);
      HIP_CHECK_RET(hipGetLastError());
      if ((rc = launch_conv_dgrad<float>(s, g, scratch, dx, (const float*)nullptr, st))) return rc;
    }
    if (dw) {
      ARG_CHECK(x, "linear_backward: x required for dw");
      if ((rc = launch_conv_wgrad<float>(s, g, x, reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(scratch) + wt_bytes), dw, st))) return rc;
    }
    if (db && (rc = colsum(g, db, M, N, st))) return rc;
    return MMSKIN_OK;
  }
  static const bool one_launch = [] { const char* v = getenv("MMSKIN_LINEAR_BWD_ONE"); return !v || atoi(v) != 0; }();
  if (one_launch && M < 4096 && (dx || dw || db)) {   // (a longer contraction takes the split-K form of the separate launches)
    ARG_CHECK((!dx || w) && (!dw || x), "linear_backward: w / x required");
    const int nx_dx = ceil_div(K, LG_T), n_dx = dx ? nx_dx * ceil_div(M, LG_T) : 0;
    const int nx_dw = ceil_div(K, LG_T), n_dw = dw ? nx_dw * ceil_div(N, LG_T) : 0;
    const int n_db = db ? ceil_div(N, 32) : 0;
    hipLaunchKernelGGL(linear_bwd_small_kernel, dim3(n_dx + n_dw + n_db), dim3(256), 0, st, g, w, x, dx, dw, db, M, K, N, nx_dx, n_dx, nx_dw, n_dw);
    HIP_CHECK_RET(hipGetLastError());
    return MMSKIN_OK;
  }
  if (dx) {  // dx[m][k] = sum_n g[m][n] * w[n][k]
    ARG_CHECK(w, "linear_backward: w required for dx");
    if ((rc = gemm_f32(g, w, dx, nullptr, M, K, N, N, 1, 1, K, K, 0, st))) return rc;
  }
  if (dw) {  // dw[n][k] = sum_m g[m][n] * x[m][k]
    ARG_CHECK(x, "linear_backward: x required for dw");
    if ((rc = gemm_f32(g, x, dw, nullptr, N, K, M, 1, N, 1, K, K, 0, st))) return rc;
  }
  if (db) {
    if ((rc = colsum(g, db, M, N, st))) return rc;
  }
  return MMSKIN_OK;
}

int mmskin_linear_backward(const float* dy, const float* x, const float* w, const float* y_relu, float* dy_scratch,
                           float* dx, float* dw, float* db, int M, int K, int N, void* stream) {
  return linear_backward_impl(dy, x, w, y_relu, nullptr, dy_scratch, dx, dw, db, M, K, N, stream);
}
// Backward of h = gelu(x w^T + b) given dh and the saved pre-activation z [M][N]: the GELU derivative is applied inside the pass that
// converts the gradient for the bf16 GEMMs (and sums it for db), so d(z) never exists in fp32.  dy_scratch [M][N]: used off the bf16 path.
int mmskin_linear_gelu_backward(const float* dh, const float* x, const float* w, const float* z, float* dy_scratch, float* dx, float* dw,
                                float* db, int M, int K, int N, void* stream) {
  ARG_CHECK(z, "linear_gelu_backward: z required");
  return linear_backward_impl(dh, x, w, nullptr, z, dy_scratch, dx, dw, db, M, K, N, stream);
}
// Backward with the bf16 operand copy kept by mmskin_linear_forward_keep in x's place (x16 [M][mmskin_linear_x16_pitch]); y_relu /
// z_gelu as in the two entry points above (at most one).
int mmskin_linear_backward_keep(const float* dy, const void* x16, const float* w, const float* y_relu, const float* z_gelu, float* dy_scratch,
                                float* dx, float* dw, float* db, int M, int K, int N, void* stream) {
  ARG_CHECK(x16, "linear_backward_keep: x16 required");
  return linear_backward_impl(dy, nullptr, w, y_relu, z_gelu, dy_scratch, dx, dw, db, M, K, N, stream, x16);
}

int mmskin_layernorm_forward(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd,
                             int M, int N, float eps, int relu, void* stream) {
  ARG_CHECK(x && g && b && y && mean && rstd && M > 0 && N > 0, "layernorm_forward: bad argument");
  static const bool rows = [] { const char* v = getenv("MMSKIN_LN_ROWS"); return !v || atoi(v) != 0; }();
  if (rows && ln_rows_ok(N, relu)) {
#define CALL(NJ, LPR) hipLaunchKernelGGL((layernorm_fwd_rows_kernel<NJ, LPR>), dim3(ceil_div(M, 4 * (64 / LPR))), dim3(256), 0, ST(stream), x, g, b, y, (bf16_t*)nullptr, mean, rstd, M, N, eps)
    LN_ROWS_DISPATCH(N, CALL);
#undef CALL
  } else {
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, ST(stream), x, g, b, y, mean, rstd, M, N, eps, relu);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_layernorm_forward_mixed(const float* x, const float* g, const float* b, float* y_f32, void* y_bf16, int M, int N,
                                   float eps, void* stream) {
  ARG_CHECK(x && g && b && (y_f32 || y_bf16) && M > 0 && N > 0, "layernorm_forward_mixed: bad argument");
  ARG_CHECK(N % 4 == 0 && N <= 2048, "layernorm_forward_mixed: N=%d (needs N %% 4 == 0 and N <= 2048)", N);
#define CALL(NJ, LPR) hipLaunchKernelGGL((layernorm_fwd_rows_kernel<NJ, LPR>), dim3(ceil_div(M, 4 * (64 / LPR))), dim3(256), 0, ST(stream), x, g, b, y_f32, reinterpret_cast<bf16_t*>(y_bf16), (float*)nullptr, (float*)nullptr, M, N, eps)
  LN_ROWS_DISPATCH(N, CALL);
#undef CALL
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_layernorm_backward(const float* dy, const float* x, const float* g, const float* b, const float* mean,
                              const float* rstd, float* dx, float* dg, float* db, int M, int N, int relu, void* stream) {
  ARG_CHECK(dy && x && g && b && mean && rstd && M > 0 && N > 0, "layernorm_backward: bad argument");
  static const bool rows = [] { const char* v = getenv("MMSKIN_LN_ROWS"); return !v || atoi(v) != 0; }();
  if (rows && ln_rows_ok(N, relu)) {
    // <= 1024 workgroups; every wave keeps >= 8 rows so its column sums amortise the slot it writes
    const int rpw = N <= 128 ? 2 : 1;
    int G = ceil_div(M, 4 * rpw * 8);
    if (G > 1024) G = 1024;
    if (G < 1) G = 1;
    const int R = G;
    float* part = nullptr;
    if (dg || db) {
      part = head_scratch((size_t)R * 2 * N * sizeof(float));
      if (!part) { mmskin_set_error("layernorm_backward: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    }
#define CALL(NJ, LPR) hipLaunchKernelGGL((layernorm_bwd_rows_kernel<NJ, LPR>), dim3(G), dim3(256), 0, ST(stream), dy, x, g, mean, rstd, dx, part, M, N)
    LN_ROWS_DISPATCH(N, CALL);
#undef CALL
    HIP_CHECK_RET(hipGetLastError());
    if (part) return rows_sum(part, R, 2 * N, N, dg, db, ST(stream));
    return MMSKIN_OK;
  }
  if (dx) {
    hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, ST(stream), dy, x, g, b, mean, rstd, dx, M, N, relu);
    HIP_CHECK_RET(hipGetLastError());
  }
  if (dg || db) {
    int G = ceil_div(M, 64);
    if (G > 256) G = 256;
    float* part = head_scratch((size_t)G * 2 * N * sizeof(float));
    if (!part) { mmskin_set_error("layernorm_backward: scratch allocation failed"); return MMSKIN_ERR_HIP; }
    hipLaunchKernelGGL(layernorm_bwd_gb_kernel, dim3(ceil_div(N, 64), G), dim3(256), 0, ST(stream), dy, x, g, b, mean, rstd,
                       part, M, N, relu, ceil_div(M, G));
    hipLaunchKernelGGL(layernorm_gb_reduce_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, ST(stream), part, dg, db, G, N);
    HIP_CHECK_RET(hipGetLastError());
  }
  return MMSKIN_OK;
}

#define EW_LAUNCH(kern, n, ...)                                                                  \
  do {                                                                                           \
    if ((n) > 0) {                                                                               \
      hipLaunchKernelGGL(kern, dim3(grid1d(n)), dim3(256), 0, ST(stream), __VA_ARGS__);          \
      HIP_CHECK_RET(hipGetLastError());                                                          \
    }                                                                                            \
    return MMSKIN_OK;                                                                            \
  } while (0)

int mmskin_sigmoid_gate_forward(const float* z, const float* v, float* out, int64_t n, void* stream) {
  EW_LAUNCH(sigmoid_gate_fwd_kernel, n, z, v, out, n);
}
int mmskin_sigmoid_gate_backward(const float* dout, const float* z, const float* v, float* dz, float* dv, int64_t n, void* stream) {
  EW_LAUNCH(sigmoid_gate_bwd_kernel, n, dout, z, v, dz, dv, n);
}
int mmskin_gated_mix_forward(const float* z, const float* a, const float* q, float* out, int64_t n, void* stream) {
  EW_LAUNCH(gated_mix_fwd_kernel, n, z, a, q, out, n);
}
int mmskin_gated_mix_backward(const float* dout, const float* z, const float* a, const float* q, float* dz, float* da,
                              float* dq, int64_t n, void* stream) {
  EW_LAUNCH(gated_mix_bwd_kernel, n, dout, z, a, q, dz, da, dq, n);
}
int mmskin_metablock_gate_forward(const float* V, const float* t1, const float* t2, float* out, int64_t n, void* stream) {
  EW_LAUNCH(metablock_gate_fwd_kernel, n, V, t1, t2, out, n);
}
int mmskin_metablock_gate_backward(const float* dout, const float* V, const float* t1, const float* t2, float* dV,
                                   float* dt1, float* dt2, int64_t n, void* stream) {
  EW_LAUNCH(metablock_gate_bwd_kernel, n, dout, V, t1, t2, dV, dt1, dt2, n);
}
int mmskin_dropout_forward(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed, uint64_t offset,
                           void* stream) {
  EW_LAUNCH(dropout_fwd_kernel, n, x, y, mask, n, p, seed, offset);
}
int mmskin_attn_dropout_forward(const float* x, float* y, uint8_t* mask, int64_t n, int L, float p, uint64_t seed, uint64_t offset,
                                void* stream) {
  ARG_CHECK(L > 0 && n % L == 0, "attn_dropout_forward: n=%lld is not rows x L=%d", (long long)n, L);
  EW_LAUNCH(attn_dropout_fwd_kernel, n, x, y, mask, n, L, p, seed, offset);
}
// ---- Adam (torch.optim.Adam semantics: L2 weight decay added to the gradient, bias-corrected moments) over ONE flat fp32 range:
// the optimizer step of the reference's training loop (train_pad_20.py:54,113) as a single pass over the backbone's parameter arena
// instead of torch's multi-tensor launches (6 x 49 us for ResNet-50).  16-byte accesses; n % 4 handled by a scalar tail.
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                        int64_t n, float step_size, float b1c, float b2, float b2c, float eps, float wd, float inv_bc2_sqrt) {   // b1c = 1 - beta1, b2c = 1 - beta2 (formed in double on the host, like torch's scalars)
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto upd = [&](float& pv, float gv, float& mv, float& vv) {
    gv += wd * pv;
    mv += b1c * (gv - mv);
    vv = b2 * vv + b2c * gv * gv;
    pv -= step_size * mv / (sqrtf(vv) * inv_bc2_sqrt + eps);
  };
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
    reinterpret_cast<float4*>(p)[i] = pv; reinterpret_cast<float4*>(m)[i] = mv; reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    upd(p[i], g[i], m[i], v[i]);
  }
}
int mmskin_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int64_t step, void* stream) {
  ARG_CHECK(step >= 1 && n >= 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: step=%lld / 16-byte alignment", (long long)step);
  if (n == 0) return MMSKIN_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), p, g, m, v, n, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)(1.0 / sqrt(bc2)));
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_dropout_backward(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p, void* stream) {
  EW_LAUNCH(dropout_bwd_kernel, n, dy, mask, dx, n, p);
}
int mmskin_concat2_forward(const float* a, const float* b, float* out, int M, int Na, int Nb, void* stream) {
  EW_LAUNCH(concat2_fwd_kernel, (int64_t)M * (Na + Nb), a, b, out, M, Na, Nb);
}
int mmskin_concat2_backward(const float* dout, float* da, float* db, int M, int Na, int Nb, void* stream) {
  EW_LAUNCH(concat2_bwd_kernel, (int64_t)M * (Na + Nb), dout, da, db, M, Na, Nb);
}

int mmskin_attention_forward(const float* q, const float* k, const float* v, float* o, float* p, int B, int H, int L,
                             int Dh, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
  ARG_CHECK(q && k && v && o && p && B > 0 && H > 0 && L > 0 && Dh > 0, "attention_forward: bad argument");
  ARG_CHECK(drop_p >= 0.f && drop_p < 1.f, "attention_forward: dropout %f", drop_p);
  ARG_CHECK((size_t)L * L * 4 <= 64 * 1024, "attention_forward: L=%d too long for the small-attention kernel", L);
  hipLaunchKernelGGL(attention_fwd_kernel, dim3(B * H), dim3(256), (size_t)L * L * 4, ST(stream), q, k, v, o, p, L, Dh, drop_p, seed, offset);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_attention_backward(const float* dO, const float* q, const float* k, const float* v, const float* p,
                              float* dq, float* dk, float* dv, int B, int H, int L, int Dh, float drop_p,
                              uint64_t seed, uint64_t offset, void* stream) {
  ARG_CHECK(dO && q && k && v && p && dq && dk && dv, "attention_backward: null argument");
  ARG_CHECK((size_t)L * L * 4 <= 64 * 1024, "attention_backward: L=%d too long", L);
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * H), dim3(256), (size_t)L * L * 4, ST(stream), dO, q, k, v, p, dq, dk, dv, L, Dh, drop_p, seed, offset);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// One wave per (batch, head): L <= 64, Dh % 32 == 0; q / k / v (and their gradients) share one set of (batch, head, token) element
// strides, o (and dO) another; lse [B*H][L] carries the row log-sum-exp from forward to backward instead of the probabilities.
static int attn_rows_args(AttnRowsArgs& a, int B, int H, int L, int Dh, const int64_t* qkv_strides, const int64_t* o_strides, float scale,
                          float drop_p, uint64_t seed, uint64_t offset, const char* what) {
  ARG_CHECK(B > 0 && H > 0 && L > 0 && L <= 64 && (Dh == 32 || Dh == 64), "%s: needs L <= 64 and Dh 32 or 64 (L=%d Dh=%d)", what, L, Dh);
  ARG_CHECK(drop_p >= 0.f && drop_p < 1.f, "%s: dropout %f", what, (double)drop_p);
  for (int i = 0; i < 3; ++i)
    ARG_CHECK(qkv_strides[i] % 4 == 0 && o_strides[i] % 4 == 0, "%s: strides must keep rows 16-byte aligned", what);
  a.qs_b = qkv_strides[0]; a.qs_h = qkv_strides[1]; a.qs_l = qkv_strides[2];
  a.os_b = o_strides[0]; a.os_h = o_strides[1]; a.os_l = o_strides[2];
  a.nheads = B * H; a.H = H; a.L = L; a.Dh = Dh; a.scale = scale; a.drop_p = drop_p; a.seed = seed; a.offset = offset;
  a.win_ws = 0; a.win_nwy = 0; a.win_nwx = 0;
  return MMSKIN_OK;
}
static int attn_rows_fwd_launch(const float* q, const float* k, const float* v, float* o, float* lse, const AttnRowsArgs& a, void* stream) {
  const size_t lds = (size_t)4 * a.L * 64 * sizeof(float);
  if (a.Dh == 32) hipLaunchKernelGGL(attention_rows_fwd_kernel<32>, dim3((a.nheads + 3) / 4), dim3(256), lds, ST(stream), q, k, v, o, lse, a);
  else hipLaunchKernelGGL(attention_rows_fwd_kernel<64>, dim3((a.nheads + 3) / 4), dim3(256), lds, ST(stream), q, k, v, o, lse, a);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
static int attn_rows_bwd_launch(const float* dO, const float* q, const float* k, const float* v, const float* o, const float* lse,
                                float* dq, float* dk, float* dv, const AttnRowsArgs& a, void* stream) {
  const size_t lds = (size_t)4 * 2 * a.L * (a.L + 1) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {   // up to 4 waves x 2 tiles x 64 x 65 floats
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_rows_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 64 * 65 * 4));
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_rows_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 64 * 65 * 4));
    attr_done = true;
  }
  if (a.Dh == 32) hipLaunchKernelGGL(attention_rows_bwd_kernel<32>, dim3((a.nheads + 3) / 4), dim3(256), lds, ST(stream), q, k, v, o, dO, lse, dq, dk, dv, a);
  else hipLaunchKernelGGL(attention_rows_bwd_kernel<64>, dim3((a.nheads + 3) / 4), dim3(256), lds, ST(stream), q, k, v, o, dO, lse, dq, dk, dv, a);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
// Window attention on an image-major token grid [B][nwy*ws][nwx*ws] (DaViT SpatialBlock, timm davit.py:window_partition /
// window_reverse around WindowAttention): window (image, wy, wx) attends over its ws*ws tokens where they sit -- no partition /
// reverse copies.  token_stride / head_stride: element strides of q / k / v (and dq / dk / dv) and of o (and dO); lse is
// [B*nwy*nwx][H][ws*ws] (private to the forward / backward pair).
static int attn_window_args(AttnRowsArgs& a, int B, int nwy, int nwx, int ws, int H, int Dh, int64_t q_tok, int64_t q_head, int64_t o_tok,
                            int64_t o_head, float scale, float drop_p, uint64_t seed, uint64_t offset, const char* what) {
  ARG_CHECK(B > 0 && nwy > 0 && nwx > 0 && ws > 0 && ws * ws <= 64, "%s: needs ws*ws <= 64 (ws=%d)", what, ws);
  ARG_CHECK((int64_t)B * nwy * nwx * H < (int64_t)1 << 30, "%s: too many windows", what);
  const int64_t qs[3] = {0, q_head, q_tok}, os[3] = {0, o_head, o_tok};
  int rc = attn_rows_args(a, B * nwy * nwx, H, ws * ws, Dh, qs, os, scale, drop_p, seed, offset, what);
  if (rc) return rc;
  a.win_ws = ws; a.win_nwy = nwy; a.win_nwx = nwx;
  return MMSKIN_OK;
}
int mmskin_window_attention_forward(const float* q, const float* k, const float* v, float* o, float* lse, int B, int nwy, int nwx, int ws,
                                    int H, int Dh, int64_t q_tok, int64_t q_head, int64_t o_tok, int64_t o_head, float scale,
                                    float drop_p, uint64_t seed, uint64_t offset, void* stream) {
  ARG_CHECK(q && k && v && o, "window_attention_forward: null argument");
  ARG_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0, "window_attention_forward: 16-byte aligned tensors required");
  AttnRowsArgs a;
  int rc = attn_window_args(a, B, nwy, nwx, ws, H, Dh, q_tok, q_head, o_tok, o_head, scale, drop_p, seed, offset, "window_attention_forward");
  if (rc) return rc;
  return attn_rows_fwd_launch(q, k, v, o, lse, a, stream);
}
int mmskin_window_attention_backward(const float* dO, const float* q, const float* k, const float* v, const float* o, const float* lse,
                                     float* dq, float* dk, float* dv, int B, int nwy, int nwx, int ws, int H, int Dh, int64_t q_tok,
                                     int64_t q_head, int64_t o_tok, int64_t o_head, float scale, float drop_p, uint64_t seed,
                                     uint64_t offset, void* stream) {
  ARG_CHECK(dO && q && k && v && o && lse && dq && dk && dv, "window_attention_backward: null argument");
  ARG_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dO | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
            "window_attention_backward: 16-byte aligned tensors required");
  AttnRowsArgs a;
  int rc = attn_window_args(a, B, nwy, nwx, ws, H, Dh, q_tok, q_head, o_tok, o_head, scale, drop_p, seed, offset, "window_attention_backward");
  if (rc) return rc;
  return attn_rows_bwd_launch(dO, q, k, v, o, lse, dq, dk, dv, a, stream);
}
int mmskin_attention_rows_forward(const float* q, const float* k, const float* v, float* o, float* lse, int B, int H, int L, int Dh,
                                  const int64_t* qkv_strides, const int64_t* o_strides, float scale, float drop_p, uint64_t seed,
                                  uint64_t offset, void* stream) {
  ARG_CHECK(q && k && v && o && qkv_strides && o_strides, "attention_rows_forward: null argument");
  ARG_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0, "attention_rows_forward: 16-byte aligned tensors required");
  AttnRowsArgs a;
  int rc = attn_rows_args(a, B, H, L, Dh, qkv_strides, o_strides, scale, drop_p, seed, offset, "attention_rows_forward");
  if (rc) return rc;
  return attn_rows_fwd_launch(q, k, v, o, lse, a, stream);
}
int mmskin_attention_rows_backward(const float* dO, const float* q, const float* k, const float* v, const float* o, const float* lse,
                                   float* dq, float* dk, float* dv, int B, int H, int L, int Dh, const int64_t* qkv_strides,
                                   const int64_t* o_strides, float scale, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
  ARG_CHECK(dO && q && k && v && o && lse && dq && dk && dv && qkv_strides && o_strides, "attention_rows_backward: null argument");
  ARG_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dO | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0,
            "attention_rows_backward: 16-byte aligned tensors required");
  AttnRowsArgs a;
  int rc = attn_rows_args(a, B, H, L, Dh, qkv_strides, o_strides, scale, drop_p, seed, offset, "attention_rows_backward");
  if (rc) return rc;
  return attn_rows_bwd_launch(dO, q, k, v, o, lse, dq, dk, dv, a, stream);
}

int mmskin_mdnet_fuse_forward(const float* feat, const float* z, const float* t1, const float* t2, float* pooled,
                              int64_t NC, int HW, void* stream) {
  ARG_CHECK(feat && z && t1 && t2 && pooled && NC > 0 && HW > 0, "mdnet_fuse_forward: bad argument");
  hipLaunchKernelGGL(mdnet_fuse_fwd_kernel, dim3((unsigned)((NC + 3) / 4)), dim3(256), 0, ST(stream), feat, z, t1, t2, pooled, NC, HW);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_mdnet_fuse_backward(const float* dpooled, const float* feat, const float* z, const float* t1, const float* t2,
                               float* dfeat, float* dz, float* dt1, float* dt2, int64_t NC, int HW, void* stream) {
  ARG_CHECK(dpooled && feat && z && t1 && t2 && dz && dt1 && dt2 && NC > 0 && HW > 0, "mdnet_fuse_backward: bad argument");
  hipLaunchKernelGGL(mdnet_fuse_bwd_kernel, dim3((unsigned)((NC + 3) / 4)), dim3(256), 0, ST(stream), dpooled, feat, z, t1, t2,
                     dfeat, dz, dt1, dt2, NC, HW);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

int mmskin_bmm(const float* a, const float* b, float* c, int batch, int M, int N, int K, int64_t sam, int64_t sak, int64_t sab,
               int64_t sbn, int64_t sbk, int64_t sbb, int64_t ldc, int64_t scb, void* stream) {
  ARG_CHECK(a && b && c && batch > 0 && M > 0 && N > 0 && K > 0, "bmm: bad argument");
  if (batch == 1) return gemm_f32(a, b, c, nullptr, M, N, K, sam, sak, sbn, sbk, ldc, 0, ST(stream));
  ARG_CHECK(batch <= 65535, "bmm: batch %d exceeds the grid limit", batch);
  return gemm_f32(a, b, c, nullptr, M, N, K, sam, sak, sbn, sbk, ldc, 0, ST(stream), batch, sab, sbb, scb);
}
int mmskin_softmax_forward(const float* x, const float* mask_add, const float* bias, float* y, int64_t rows, int L,
                           int64_t rows_per_batch, float scale, int causal, void* stream) {
  ARG_CHECK(x && y && rows > 0 && L > 0 && rows_per_batch > 0, "softmax_forward: bad argument");
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), x, mask_add, bias, y, rows, L,
                     rows_per_batch, scale, causal);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_softmax_backward(const float* dy, const float* y, float* dx, int64_t rows, int L, float scale, void* stream) {
  ARG_CHECK(dy && y && dx && rows > 0 && L > 0, "softmax_backward: bad argument");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ST(stream), dy, y, dx, rows, L, scale);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_colsum(const float* x, float* out, int M, int N, void* stream) {
  ARG_CHECK(x && out && M > 0 && N > 0, "colsum: bad argument");
  return colsum(x, out, M, N, ST(stream));
}
int mmskin_scale_add_forward(const float* x, const float* b, const float* gamma, float* y, int64_t n, int C, void* stream) {
  ARG_CHECK(x && b && gamma && y && n > 0 && C > 0, "scale_add_forward: bad argument");
  EW_LAUNCH(scale_add_fwd_kernel, n, x, b, gamma, y, n, C);
}
/* out = dy * v, v per channel (per_channel = 1: v[C]) or elementwise (v[n]) */
int mmskin_scale_mul(const float* dy, const float* v, float* out, int64_t n, int C, int per_channel, void* stream) {
  ARG_CHECK(dy && v && out && n > 0 && C > 0, "scale_mul: bad argument");
  EW_LAUNCH(scale_mul_kernel, n, dy, v, out, n, C, per_channel);
}
int mmskin_token_mean_forward(const float* x, float* out, int B, int L, int E, int start, void* stream) {
  ARG_CHECK(x && out && B > 0 && L > start && E > 0, "token_mean_forward: bad argument");
  EW_LAUNCH(token_mean_fwd_kernel, (int64_t)B * E, x, out, B, L, E, start);
}
int mmskin_token_mean_backward(const float* dout, float* dx, int B, int L, int E, int start, void* stream) {
  ARG_CHECK(dout && dx && B > 0 && L > start && E > 0, "token_mean_backward: bad argument");
  EW_LAUNCH(token_mean_bwd_kernel, (int64_t)B * L * E, dout, dx, B, L, E, start);
}
int mmskin_set_linear_dtype(int dtype) {
  ARG_CHECK(dtype == 0 || dtype == 1, "set_linear_dtype: dtype %d (MMSKIN_F32 = 0, MMSKIN_BF16 = 1)", dtype);
  g_linear_dtype = dtype;
  return MMSKIN_OK;
}
int mmskin_get_linear_dtype(void) { return linear_bf16() ? 1 : 0; }
int mmskin_add(const float* a, const float* b, float* y, int64_t n, int64_t nb, void* stream) {
  ARG_CHECK(a && b && y && n > 0 && nb > 0 && n % nb == 0, "add: bad argument");
  EW_LAUNCH(add_kernel, n, a, b, y, n, nb);
}
int mmskin_gelu_forward(const float* x, float* y, int64_t n, void* stream) { EW_LAUNCH(gelu_fwd_kernel, n, x, y, n); }
int mmskin_gelu_tanh_forward(const float* x, float* y, int64_t n, void* stream) { EW_LAUNCH(gelu_tanh_fwd_kernel, n, x, y, n); }
int mmskin_gelu_tanh_backward(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  EW_LAUNCH(gelu_tanh_bwd_kernel, n, dy, x, dx, n);
}
int mmskin_gelu_backward(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  EW_LAUNCH(gelu_bwd_kernel, n, dy, x, dx, n);
}

int mmskin_embedding_forward(const float* table, const int64_t* ids, float* out, int B, int ncols, int card, int E,
                             void* stream) {
  EW_LAUNCH(embedding_fwd_kernel, (int64_t)B * ncols * E, table, ids, out, B, ncols, card, E);
}
int mmskin_embedding_backward(const float* dout, const int64_t* ids, float* dtable, int B, int ncols, int card, int E,
                              void* stream) {
  EW_LAUNCH(embedding_bwd_kernel, (int64_t)ncols * card * E, dout, ids, dtable, B, ncols, card, E);
}

int mmskin_direct_conv2d_forward(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W,
                                 int Cout, int kh, int kw, int stride, int pad, int relu, void* stream) {
  int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
  EW_LAUNCH(direct_conv_fwd_kernel, (int64_t)N * Cout * OH * OW, x, w, b, y, N, Cin, H, W, Cout, kh, kw, stride, pad, OH, OW, relu);
}
int mmskin_direct_conv2d_backward(const float* dy, const float* x, const float* y_relu, float* dw, float* db, int N,
                                  int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, void* stream) {
  ARG_CHECK(dy && x && dw, "direct_conv2d_backward: null argument");
  int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
  int blocks = Cout * Cin * kh * kw + (db ? Cout : 0);
  hipLaunchKernelGGL(direct_conv_bwd_kernel, dim3(blocks), dim3(256), 0, ST(stream), dy, x, y_relu, dw, db, N, Cin, H, W,
                     Cout, kh, kw, stride, pad, OH, OW);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_pool_gap_forward(const float* x, float* y, int32_t* idx, int N, int C, int H, int W, int k, void* stream) {
  ARG_CHECK(x && y && idx && k > 0 && H >= k && W >= k, "pool_gap_forward: bad argument");
  hipLaunchKernelGGL(pool_gap_fwd_kernel, dim3(N * C), dim3(64), 0, ST(stream), x, y, idx, C, H, W, k, H / k, W / k);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}
int mmskin_pool_gap_backward(const float* dy, const int32_t* idx, float* dx, int N, int C, int H, int W, int k,
                             void* stream) {
  ARG_CHECK(dy && idx && dx, "pool_gap_backward: null argument");
  HIP_CHECK_RET(hipMemsetAsync(dx, 0, (size_t)N * C * H * W * sizeof(float), ST(stream)));
  int64_t NC = (int64_t)N * C;
  int PHW = (H / k) * (W / k);
  EW_LAUNCH(pool_gap_bwd_kernel, NC * PHW, dy, idx, dx, NC, H * W, PHW);
}

}  // extern "C"

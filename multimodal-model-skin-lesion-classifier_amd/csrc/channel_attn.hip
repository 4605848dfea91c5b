// Channel ("transposed") attention of DaViT's ChannelBlock for gfx950, forward and backward, on the TOKEN-MAJOR packed qkv a fused qkv
// Linear writes -- no permute / contiguous copies on either side.
//
// Replaces timm davit.py ChannelAttention.forward (reached through the reference's generic timm branch,
// loadImageModelClassifier.py:117-131, `timm.create_model("davit_*")`):
//     q, k, v [B, G, N, 32]  (G groups of 32 channels, N tokens);  q *= scale  (dynamic_scale: N^-0.5)
//     A = softmax(q^T k, dim=-1)          [B, G, 32, 32]   channels attend over channels, the reduction runs over the tokens
//     x = (A v^T)^T                       [B, G, N, 32]    x[n][i] = sum_j A[i][j] v[n][j]
// which the Python path ran as two transposing copies of the whole qkv / output tensors around the generic attention kernel.
//
// One workgroup per (batch, group):
//   phase 1  S = q^T k: v_mfma_f32_32x32x2_f32 straight from global memory -- a lane's A / B operand is ONE float of two consecutive
//            token rows (lane = 32 * (token parity) + channel), i.e. two coalesced 128-byte rows per operand and MFMA; the four waves
//            take every fourth token pair and meet in LDS.  fp32 throughout (the 32 x 32 result decides a softmax).
//   phase 2  softmax rows (32 threads), A kept in LDS transposed; the forward also stores A [B*G][32][32] for the backward.
//   phase 3  one thread per token: x[n][:] = A v[n][:] with the matrix read as LDS broadcasts (every lane the same address).
// Backward: dA = dO^T v (phase 1 with dO in q's place), dS = A . (dA - rowsum(dA . A)) * scale, then per token
//   dv[n] = A^T dO[n],  dq[n] = dS k[n],  dk[n] = dS^T q[n]   -- three 32 x 32 matrix-vector products against LDS-resident matrices.
// More than CHAN_CHUNK tokens (DaViT stages 1 - 2: 3 136 / 784 tokens but only 192 / 384 (batch, group) pairs): the token range is cut
// into chunks of CHAN_CHUNK -- chan_outer_kernel writes one partial 32 x 32 product per (batch, group, chunk), and the apply kernels,
// one workgroup per chunk, add the partials (every chunk repeats the 32-row softmax) and handle their own tokens.  One workgroup per
// pair walked 392 dependent MFMA steps per wave behind 4-byte loads: 200 us forward / 250 us backward per launch on average.
// Deterministic (fixed summation order), no atomics.
#include "../../include/mmskin.h"
#include <stdlib.h>

#include "common.h"

#define ST(s) ((hipStream_t)(s))

namespace {

typedef float f32x16_t __attribute__((ext_vector_type(16)));

struct ChanArgs {
  int64_t q_tok, q_b;   // element strides of q / k / v (and dq / dk / dv): token, batch; a group's 32 channels are contiguous at g * 32
  int64_t o_tok, o_b;   // of x (and dO)
  int G, N;
  float scale;
  int nchunk;           // > 1: partial products in `part`, blockIdx.y = chunk
};
#define CHAN_CHUNK 256

// sum_n a[n][i] * b[n][j] over the workgroup's tokens -> red[0][i][j] (all threads return after the barrier)
__device__ __forceinline__ void tokens_outer_32x32(const float* __restrict__ a, int64_t a_tok, const float* __restrict__ b, int64_t b_tok,
                                                   int N, float (*red)[32][33]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int base = 0; base < N; base += 64) {                  // 64 tokens per trip: 8 token pairs per wave, their 16 loads in flight together
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = base + 8 * u + wave * 2 + lh;
      const bool ok = n < N;
      av[u] = ok ? a[(int64_t)n * a_tok + li] : 0.f;
      bv[u] = ok ? b[(int64_t)n * b_tok + li] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
  // D[i][j]: j = lane % 32, i = 8 * (r / 4) + 4 * (lane / 32) + r % 4
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][8 * (r >> 2) + 4 * lh + (r & 3)][li] = acc[r];
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 256) {
    const int i = e >> 5, j = e & 31;
    red[0][i][j] = (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]);
  }
  __syncthreads();
}

// out[c] = sum_r M[r][c] * in[r]   (M in LDS, row pitch 32 floats: every lane reads the same 16 bytes -> broadcast).
// The matrix is loop-invariant for the caller's token loop: left alone, the compiler hoists all 256 b128 reads (1024 registers' worth)
// out of it and spills them to scratch (the first build: 200 us per launch in this function alone).  The compiler-level memory fences
// pin each row's reads to their place -- row r + 1 is in flight while row r is multiplied, nothing more.
__device__ __forceinline__ void matvec32(const float (*M)[32], const float (&in)[32], float (&out)[32]) {
#pragma unroll
  for (int c = 0; c < 32; ++c) out[c] = 0.f;
  float4 m[8], nx[8];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int c = 0; c < 8; ++c) m[c] = *reinterpret_cast<const float4*>(&M[0][4 * c]);
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    if (r + 1 < 32) {
#pragma unroll
      for (int c = 0; c < 8; ++c) nx[c] = *reinterpret_cast<const float4*>(&M[r + 1][4 * c]);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      out[4 * c] = fmaf(m[c].x, in[r], out[4 * c]); out[4 * c + 1] = fmaf(m[c].y, in[r], out[4 * c + 1]);
      out[4 * c + 2] = fmaf(m[c].z, in[r], out[4 * c + 2]); out[4 * c + 3] = fmaf(m[c].w, in[r], out[4 * c + 3]);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = nx[c];
  }
}
__device__ __forceinline__ void load_row32(const float* __restrict__ p, float (&v)[32]) {
#pragma unroll
  for (int c = 0; c < 32; c += 4) {
    const float4 t = *reinterpret_cast<const float4*>(p + c);
    v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
  }
}
__device__ __forceinline__ void store_row32(float* __restrict__ p, const float (&v)[32]) {
#pragma unroll
  for (int c = 0; c < 32; c += 4) *reinterpret_cast<float4*>(p + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
}

// part[(bg * nchunk + chunk)][32][32] = sum over the chunk's tokens of a[n][i] * b[n][j]
__global__ __launch_bounds__(256) void chan_outer_kernel(const float* __restrict__ a, int64_t a_tok, int64_t a_b, const float* __restrict__ b,
                                                         int64_t b_tok, int64_t b_b, float* __restrict__ part, int G, int N, int nchunk) {
  __shared__ float red[4][32][33];
  const int bg = blockIdx.x, bi = bg / G, g = bg - bi * G, c = blockIdx.y;
  const int n0 = c * CHAN_CHUNK, cnt = min(N - n0, CHAN_CHUNK);
  tokens_outer_32x32(a + bi * a_b + (int64_t)g * 32 + (int64_t)n0 * a_tok, a_tok, b + bi * b_b + (int64_t)g * 32 + (int64_t)n0 * b_tok, b_tok, cnt, red);
  float* out = part + ((int64_t)bg * nchunk + c) * 1024;
  for (int e = threadIdx.x; e < 1024; e += 256) out[e] = red[0][e >> 5][e & 31];
}
// red[0] = sum of the pair's partial products (chunk order)
__device__ __forceinline__ void sum_partials(const float* __restrict__ part, int bg, int nchunk, float (*red)[32][33]) {
  const float* src = part + (int64_t)bg * nchunk * 1024;
  for (int e = threadIdx.x; e < 1024; e += 256) {
    float t = 0.f;
    for (int c = 0; c < nchunk; ++c) t += src[(int64_t)c * 1024 + e];
    red[0][e >> 5][e & 31] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void channel_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, float* __restrict__ x,
                                                               float* __restrict__ attn, const float* __restrict__ part, const ChanArgs p) {
  __shared__ float red[4][32][33];
  __shared__ __attribute__((aligned(16))) float AT[32][32];      // AT[j][i] = A[i][j]
  const int bg = blockIdx.x, b = bg / p.G, g = bg - b * p.G;
  const int64_t qoff = b * p.q_b + (int64_t)g * 32, ooff = b * p.o_b + (int64_t)g * 32;
  const int n_begin = blockIdx.y * CHAN_CHUNK, n_end = p.nchunk > 1 ? min(p.N, n_begin + CHAN_CHUNK) : p.N;
  if (p.nchunk > 1) sum_partials(part, bg, p.nchunk, red);
  else tokens_outer_32x32(q + qoff, p.q_tok, k + qoff, p.q_tok, p.N, red);
  if (threadIdx.x < 32) {
    const int i = threadIdx.x;
    float mx = -INFINITY;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) mx = fmaxf(mx, red[0][i][j] * p.scale);
    float sum = 0.f;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) { const float e = expf(red[0][i][j] * p.scale - mx); red[1][i][j] = e; sum += e; }
    const float inv = 1.f / sum;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      const float a = red[1][i][j] * inv;
      AT[j][i] = a;
      if (attn && blockIdx.y == 0) attn[((int64_t)bg * 32 + i) * 32 + j] = a;
    }
  }
  __syncthreads();
  for (int n = n_begin + threadIdx.x; n < n_end; n += 256) {
    float vin[32], out[32];
    load_row32(v + qoff + (int64_t)n * p.q_tok, vin);
    matvec32(AT, vin, out);                                      // out[i] = sum_j AT[j][i] v[j]
    store_row32(x + ooff + (int64_t)n * p.o_tok, out);
  }
}

__global__ __launch_bounds__(256) void channel_attn_bwd_kernel(const float* __restrict__ dO, const float* __restrict__ q,
                                                               const float* __restrict__ k, const float* __restrict__ v,
                                                               const float* __restrict__ attn, float* __restrict__ dq,
                                                               float* __restrict__ dk, float* __restrict__ dv,
                                                               const float* __restrict__ part, const ChanArgs p) {
  __shared__ float red[4][32][33];
  __shared__ __attribute__((aligned(16))) float A[32][32], dS[32][32], dST[32][32];
  const int bg = blockIdx.x, b = bg / p.G, g = bg - b * p.G;
  const int64_t qoff = b * p.q_b + (int64_t)g * 32, ooff = b * p.o_b + (int64_t)g * 32;
  const int n_begin = blockIdx.y * CHAN_CHUNK, n_end = p.nchunk > 1 ? min(p.N, n_begin + CHAN_CHUNK) : p.N;
  if (p.nchunk > 1) sum_partials(part, bg, p.nchunk, red);
  else tokens_outer_32x32(dO + ooff, p.o_tok, v + qoff, p.q_tok, p.N, red);     // dA[i][j] = sum_n dO[n][i] v[n][j]
  if (threadIdx.x < 32) {
    const int i = threadIdx.x;
    const float* ar = attn + ((int64_t)bg * 32 + i) * 32;
    float dot = 0.f;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) { const float a = ar[j]; A[i][j] = a; dot = fmaf(red[0][i][j], a, dot); }
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      const float d = A[i][j] * (red[0][i][j] - dot) * p.scale;
      dS[i][j] = d; dST[j][i] = d;
    }
  }
  __syncthreads();
  for (int n = n_begin + threadIdx.x; n < n_end; n += 256) {
    float in[32], out[32];
    load_row32(dO + ooff + (int64_t)n * p.o_tok, in);
    matvec32(A, in, out);                                        // dv[j] = sum_i A[i][j] dO[i]
    store_row32(dv + qoff + (int64_t)n * p.q_tok, out);
    load_row32(k + qoff + (int64_t)n * p.q_tok, in);
    matvec32(dST, in, out);                                      // dq[i] = sum_j dS[i][j] k[j]
    store_row32(dq + qoff + (int64_t)n * p.q_tok, out);
    load_row32(q + qoff + (int64_t)n * p.q_tok, in);
    matvec32(dS, in, out);                                       // dk[j] = sum_i dS[i][j] q[i]
    store_row32(dk + qoff + (int64_t)n * p.q_tok, out);
  }
}

int chan_args(ChanArgs& a, int B, int G, int N, int Dh, int64_t q_tok, int64_t q_b, int64_t o_tok, int64_t o_b, float scale, const char* what) {
  ARG_CHECK(B > 0 && G > 0 && N > 0 && Dh == 32, "%s: needs 32 channels per group (B=%d G=%d N=%d Dh=%d)", what, B, G, N, Dh);
  ARG_CHECK((int64_t)B * G < (int64_t)1 << 30, "%s: too many (batch, group) pairs", what);
  ARG_CHECK(q_tok % 4 == 0 && q_b % 4 == 0 && o_tok % 4 == 0 && o_b % 4 == 0, "%s: strides must keep rows 16-byte aligned", what);
  a.q_tok = q_tok; a.q_b = q_b; a.o_tok = o_tok; a.o_b = o_b; a.G = G; a.N = N; a.scale = scale;
  a.nchunk = (N + CHAN_CHUNK - 1) / CHAN_CHUNK;
  return MMSKIN_OK;
}

}  // namespace

extern "C" {

int64_t mmskin_channel_attention_scratch_floats(int B, int G, int N) {
  const int64_t nchunk = (N + CHAN_CHUNK - 1) / CHAN_CHUNK;
  return nchunk > 1 ? (int64_t)B * G * nchunk * 1024 : 0;
}

int mmskin_channel_attention_forward(const float* q, const float* k, const float* v, float* x, float* attn, float* scratch, int B, int G,
                                     int N, int Dh, int64_t q_tok, int64_t q_b, int64_t o_tok, int64_t o_b, float scale, void* stream) {
  ARG_CHECK(q && k && v && x, "channel_attention_forward: null argument");
  ARG_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)x) & 15) == 0, "channel_attention_forward: 16-byte aligned tensors required");
  ChanArgs a;
  int rc = chan_args(a, B, G, N, Dh, q_tok, q_b, o_tok, o_b, scale, "channel_attention_forward");
  if (rc) return rc;
  ARG_CHECK(a.nchunk == 1 || scratch, "channel_attention_forward: scratch required for N > %d", CHAN_CHUNK);
  if (a.nchunk > 1) hipLaunchKernelGGL(chan_outer_kernel, dim3(B * G, a.nchunk), dim3(256), 0, ST(stream), q, q_tok, q_b, k, q_tok, q_b, scratch, G, N, a.nchunk);
  hipLaunchKernelGGL(channel_attn_fwd_kernel, dim3(B * G, a.nchunk), dim3(256), 0, ST(stream), q, k, v, x, attn, scratch, a);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

int mmskin_channel_attention_backward(const float* dO, const float* q, const float* k, const float* v, const float* attn, float* dq,
                                      float* dk, float* dv, float* scratch, int B, int G, int N, int Dh, int64_t q_tok, int64_t q_b,
                                      int64_t o_tok, int64_t o_b, float scale, void* stream) {
  ARG_CHECK(dO && q && k && v && attn && dq && dk && dv, "channel_attention_backward: null argument");
  ARG_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dO | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)attn) & 15) == 0,
            "channel_attention_backward: 16-byte aligned tensors required");
  ChanArgs a;
  int rc = chan_args(a, B, G, N, Dh, q_tok, q_b, o_tok, o_b, scale, "channel_attention_backward");
  if (rc) return rc;
  ARG_CHECK(a.nchunk == 1 || scratch, "channel_attention_backward: scratch required for N > %d", CHAN_CHUNK);
  if (a.nchunk > 1) hipLaunchKernelGGL(chan_outer_kernel, dim3(B * G, a.nchunk), dim3(256), 0, ST(stream), dO, o_tok, o_b, v, q_tok, q_b, scratch, G, N, a.nchunk);
  hipLaunchKernelGGL(channel_attn_bwd_kernel, dim3(B * G, a.nchunk), dim3(256), 0, ST(stream), dO, q, k, v, attn, dq, dk, dv, scratch, a);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

}  // extern "C"

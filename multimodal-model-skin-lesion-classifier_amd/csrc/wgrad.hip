// Weight-gradient GEMM for gfx950:  dW[cout][tap][c] = sum_m dY[m][cout] * gather(in)[m][tap][c]
//
// Replaces ATen/cuDNN conv_backward(weight) reached through the reference's torchvision backbone.
// The reduction runs over pixels m, which is the *strided* dimension of both NHWC operands, so the
// MFMA fragments (8 consecutive k per lane) need a transpose.  Tiles are staged in LDS exactly as
// they sit in memory ([m][channel], 16-byte chunks, XOR-swizzled by row) and bf16 fragments are read
// with ds_read_b64_tr_b16 (the CDNA4 transposing LDS read); the exact-f32 path reads one dword per
// lane, which needs no transpose.  The pixel range is split over workgroups; every split writes its
// own fp32 slab (deterministic, no atomics) and wgrad_reduce sums the slabs straight into the
// OIHW gradient tensor.
#include <stdlib.h>

#include "conv.h"
#include "ops.h"

template <typename T> struct WG;
template <> struct WG<bf16_t> { static constexpr int MS = 32; };
template <> struct WG<float> { static constexpr int MS = 16; };

// row swizzle (in 16-byte chunks) for a tile whose rows are PITCH bytes
template <typename T, int PITCH> __device__ __forceinline__ int wg_swz(int row) {
  if constexpr (sizeof(T) == 4) {
    return (row & 1) << 2;
  } else if constexpr (PITCH % 256 == 0) {
    return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
  } else {
    return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
  }
}

__device__ uint4 g_wzero_page[16];  // zeros: source for rows past the split / padding taps

__device__ __forceinline__ uint2 lds_read_tr16_b64(const unsigned char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}

// One staged block of MS pixel rows: acc[kidx frag][cout frag] += Y^T X  (fragments via transposing LDS reads)
template <typename T, int BO, int BKK, int MS>
__device__ __forceinline__ void wg_compute(const unsigned char* Xb, const unsigned char* Yb,
                                           f32x4_t (&acc)[BKK / 32][BO / 32], int wo, int wk, int l15, int g) {
  constexpr int PX = BO * (int)sizeof(T), PY = BKK * (int)sizeof(T);
  constexpr int FO = BO / 32, FK = BKK / 32;
  if constexpr (sizeof(T) == 2) {
      // transposing reads: lane (q,pp) of a 16-lane group addresses row q, columns 4pp..4pp+3 of a
      // 4x16 block; lane i receives column i of the 4 rows.  Two reads give k = 8g .. 8g+7.
      const int q = l15 >> 2, pp = l15 & 3;
#pragma unroll
      for (int ks = 0; ks < MS / 32; ++ks) {
        uint4 fx[FO], fy[FK];
        const int r0 = 32 * ks + 8 * g + q, r1 = r0 + 4;
#pragma unroll
        for (int j = 0; j < FO; ++j) {
          int colb = (wo * (BO / 2) + j * 16 + 4 * pp) * 2;  // byte offset in row
          uint2 lo = lds_read_tr16_b64(Xb + r0 * PX + (((colb >> 4) ^ wg_swz<T, PX>(r0)) << 4) + (colb & 15));
          uint2 hi = lds_read_tr16_b64(Xb + r1 * PX + (((colb >> 4) ^ wg_swz<T, PX>(r1)) << 4) + (colb & 15));
          fx[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int i = 0; i < FK; ++i) {
          int colb = (wk * (BKK / 2) + i * 16 + 4 * pp) * 2;
          uint2 lo = lds_read_tr16_b64(Yb + r0 * PY + (((colb >> 4) ^ wg_swz<T, PY>(r0)) << 4) + (colb & 15));
          uint2 hi = lds_read_tr16_b64(Yb + r1 * PY + (((colb >> 4) ^ wg_swz<T, PY>(r1)) << 4) + (colb & 15));
          fy[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int i = 0; i < FK; ++i)
#pragma unroll
          for (int j = 0; j < FO; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fy[i]),
                                                                __builtin_bit_cast(bf16x8_t, fx[j]), acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < MS / 4; ++ks) {
        const int row = 4 * ks + g;
        float fx[FO], fy[FK];
#pragma unroll
        for (int j = 0; j < FO; ++j) {
          int col = wo * (BO / 2) + j * 16 + l15;
          fx[j] = *reinterpret_cast<const float*>(Xb + row * PX + ((((col >> 2) ^ wg_swz<T, PX>(row)) << 4) | ((col & 3) << 2)));
        }
#pragma unroll
        for (int i = 0; i < FK; ++i) {
          int col = wk * (BKK / 2) + i * 16 + l15;
          fy[i] = *reinterpret_cast<const float*>(Yb + row * PY + ((((col >> 2) ^ wg_swz<T, PY>(row)) << 4) | ((col & 3) << 2)));
        }
#pragma unroll
        for (int i = 0; i < FK; ++i)
#pragma unroll
          for (int j = 0; j < FO; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[i], fx[j], acc[i][j], 0, 0, 0);
      }
    }
}

// MSF scales the rows staged per barrier: small output tiles stage more pixel rows per step so every
// barrier-to-barrier interval carries >= 16 MFMAs per wave and 16-32 KB of loads (a 64x64 tile with 32
// rows per step was latency-bound at 48 TF/s).
// PF = stages kept in flight in registers (1 or 2).
// S1: 1x1 / stride 1 / no padding -- the gathered-input row of pixel m is row m: no pixel stepping and no bounds tests in the stage loop
// (~5 VALU per chunk and stage instead of ~25; a compile-time variant: the same test as a run-time flag cost more than it saved)
template <typename T, int BO, int BKK, int MSF, int PF, bool S1 = false>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs p) {
  constexpr int EPC = DT<T>::EPC, MS = WG<T>::MS * MSF;
  constexpr int PX = BO * (int)sizeof(T), PY = BKK * (int)sizeof(T);
  constexpr int X_BYTES = MS * PX, Y_BYTES = MS * PY;
  constexpr int CPR_X = BO / EPC, CPR_Y = BKK / EPC;  // chunks per row
  constexpr int NX = MS * CPR_X / 256 > 0 ? MS * CPR_X / 256 : 1;
  constexpr int NY = MS * CPR_Y / 256 > 0 ? MS * CPR_Y / 256 : 1;
  static_assert(MS * CPR_X % 256 == 0 && MS * CPR_Y % 256 == 0, "tile/thread mismatch");
  constexpr int FO = BO / 32, FK = BKK / 32;  // 16-wide fragments per wave (2x2 waves)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (X_BYTES + Y_BYTES)];
  unsigned char* Xs = smem;
  unsigned char* Ys = smem + 2 * X_BYTES;

  const int tid = threadIdx.x;
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int kb = tile % p.nblk_k; tile /= p.nblk_k;
  const int ob = tile % p.nblk_o;
  const int split = tile / p.nblk_o;
  const int o0 = ob * BO, k0 = kb * BKK;
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);

  // ---- X loader (dY rows): chunk idx = tid + 256*i
  int x_row[NX], x_ch[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) { int idx = tid + 256 * i; x_row[i] = idx / CPR_X; x_ch[i] = idx - x_row[i] * CPR_X; }
  // ---- Y loader (gathered input rows)
  int y_row[NY], y_ch[NY], y_c[NY], y_oy[NY], y_ox[NY], y_img[NY], y_dy[NY], y_dx[NY];
  const int ohw = p.OH * p.OW;
#pragma unroll
  for (int i = 0; i < NY; ++i) {
    int idx = tid + 256 * i;
    y_row[i] = idx / CPR_Y; y_ch[i] = idx - y_row[i] * CPR_Y;
    int kidx = k0 + y_ch[i] * EPC;
    int tap = kidx / p.C;
    y_c[i] = kidx - tap * p.C;
    y_dy[i] = p.offy[tap]; y_dx[i] = p.offx[tap];
    int m = m_begin + y_row[i];
    int img = m / ohw, rem = m - img * ohw;
    y_img[i] = img; y_oy[i] = rem / p.OW; y_ox[i] = rem - y_oy[i] * p.OW;
  }
  const unsigned char* dy_b = reinterpret_cast<const unsigned char*>(p.dy);
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);

  // Named staging registers (arrays of uint4 ended up in scratch memory), TWO sets: the kernel was
  // load-latency bound (ablation: removing the global loads saved 38 %, removing every MFMA 9 % -- the Y
  // operand is forward activations coming cold from HBM), so stages s+1 AND s+2 are kept in flight while
  // stage s is multiplied; the loop is unrolled by two so each set has a fixed name.
  static_assert(NX <= 4 && NY <= 4 && NX != 3 && NY != 3, "staging registers are written out for 1, 2 or 4 chunks per thread");
  const u32x4_t z4 = {0, 0, 0, 0};
  u32x4_t rxA0 = z4, rxA1 = z4, rxA2 = z4, rxA3 = z4, ryA0 = z4, ryA1 = z4, ryA2 = z4, ryA3 = z4;
  u32x4_t rxB0 = z4, rxB1 = z4, rxB2 = z4, rxB3 = z4, ryB0 = z4, ryB1 = z4, ryB2 = z4, ryB3 = z4;
#ifdef MMSKIN_ABLATE   // `make ablate` (scripts/ only)
  const int abl = p.ablate;
#else
  constexpr int abl = 0;
#endif
  int m_stage = m_begin;  // first row of the stage the next LOAD_STAGE() fetches
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(g_wzero_page);
  const int OWr = p.OW, OHr = p.OH;
  const int inv_ow = (65536 + OWr - 1) / OWr, inv_oh = (65536 + OHr - 1) / OHr;

#define LOAD_X(i, R)                                                                                   \
  {                                                                                                    \
    const int m = m_stage + x_row[i];                                                                  \
    const unsigned char* src = dy_b + ((size_t)m * p.Cout + o0 + x_ch[i] * EPC) * sizeof(T);           \
    R = *reinterpret_cast<const u32x4_t*>(m < m_end ? src : zero_page);                                \
  }
#define LOAD_Y(i, R)                                                                                   \
  { if constexpr (S1) {                                                                                \
    const int m = m_stage + y_row[i];                                                                  \
    const unsigned char* src = in_b + ((int64_t)m * p.Cpitch + y_c[i]) * (int)sizeof(T);               \
    R = *reinterpret_cast<const u32x4_t*>(m < m_end ? src : zero_page);                                \
  } else {                                                                                             \
    const int m = m_stage + y_row[i];                                                                  \
    const int iy = y_oy[i] * p.Sy + y_dy[i], ix = y_ox[i] * p.Sx + y_dx[i];                            \
    const bool ok = m < m_end && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;       \
    const unsigned char* src =                                                                         \
        in_b + ((int64_t)((y_img[i] * p.IH + iy) * p.IW + ix) * p.Cpitch + y_c[i]) * (int)sizeof(T);   \
    R = *reinterpret_cast<const u32x4_t*>(ok ? src : zero_page);                                       \
    /* advance this row's pixel by MS for the next stage; the quotients come from 16-bit reciprocal     \
       multiplies (exact for these ranges): two hardware-less integer divisions per chunk per stage     \
       cost more VALU time than the stage's 32 MFMAs */                                                \
    int nx = y_ox[i] + MS;                                                                             \
    const int qy = (nx * inv_ow) >> 16;                                                                \
    nx -= qy * OWr;                                                                                    \
    const int ny = y_oy[i] + qy;                                                                       \
    const int qi = (ny * inv_oh) >> 16;                                                                \
    y_ox[i] = nx; y_oy[i] = ny - qi * OHr; y_img[i] += qi;                                             \
  } }
#define LOAD_STAGE(S)                                                                                  \
  do {                                                                                                 \
    if (!(abl & 1)) { LOAD_X(0, rx##S##0) if constexpr (NX > 1) LOAD_X(1 % NX, rx##S##1)               \
    if constexpr (NX > 2) { LOAD_X(2 % NX, rx##S##2) LOAD_X(3 % NX, rx##S##3) } }                      \
    if (!(abl & 2)) { LOAD_Y(0, ry##S##0) if constexpr (NY > 1) LOAD_Y(1 % NY, ry##S##1)               \
    if constexpr (NY > 2) { LOAD_Y(2 % NY, ry##S##2) LOAD_Y(3 % NY, ry##S##3) } }                      \
    m_stage += MS;                                                                                     \
  } while (0)
#define ST_X(buf, i, R) *reinterpret_cast<u32x4_t*>(Xs + (buf) * X_BYTES + x_row[i] * PX + ((x_ch[i] ^ wg_swz<T, PX>(x_row[i])) << 4)) = R;
#define ST_Y(buf, i, R) *reinterpret_cast<u32x4_t*>(Ys + (buf) * Y_BYTES + y_row[i] * PY + ((y_ch[i] ^ wg_swz<T, PY>(y_row[i])) << 4)) = R;
#define STORE_STAGE(buf, S)                                                                            \
  if (!(abl & 16)) do {                                                                                \
    ST_X(buf, 0, rx##S##0) if constexpr (NX > 1) ST_X(buf, 1 % NX, rx##S##1)                           \
    if constexpr (NX > 2) { ST_X(buf, 2 % NX, rx##S##2) ST_X(buf, 3 % NX, rx##S##3) }                  \
    ST_Y(buf, 0, ry##S##0) if constexpr (NY > 1) ST_Y(buf, 1 % NY, ry##S##1)                           \
    if constexpr (NY > 2) { ST_Y(buf, 2 % NY, ry##S##2) ST_Y(buf, 3 % NY, ry##S##3) }                  \
  } while (0)

  const int wid = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wo = wid >> 1, wk = wid & 1;  // wave position: cout half, k half
  f32x4_t acc[FK][FO];
#pragma unroll
  for (int i = 0; i < FK; ++i)
#pragma unroll
    for (int j = 0; j < FO; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nstage = (m_end - m_begin + MS - 1) / MS;
  static_assert(PF == 1 || PF == 2, "prefetch depth");
  if constexpr (PF == 2) {
    if (nstage > 0) { LOAD_STAGE(A); STORE_STAGE(0, A); }
    if (nstage > 1) LOAD_STAGE(A);          // stage 1 -> set A
    if (nstage > 2) LOAD_STAGE(B);          // stage 2 -> set B
    __syncthreads();
    for (int s = 0; s < nstage; s += 2) {
      // LDS[0] holds stage s, set A holds stage s+1, set B holds stage s+2 (in flight)
      if (!(abl & 4)) wg_compute<T, BO, BKK, MS>(Xs, Ys, acc, wo, wk, l15, g);
      if (s + 1 < nstage) STORE_STAGE(1, A);
      __syncthreads();
      if (s + 3 < nstage) LOAD_STAGE(A);    // stage s+3
      if (s + 1 >= nstage) break;
      // LDS[1] holds stage s+1, set B holds stage s+2, set A holds stage s+3 (in flight)
      if (!(abl & 4)) wg_compute<T, BO, BKK, MS>(Xs + X_BYTES, Ys + Y_BYTES, acc, wo, wk, l15, g);
      if (s + 2 < nstage) STORE_STAGE(0, B);
      __syncthreads();
      if (s + 4 < nstage) LOAD_STAGE(B);    // stage s+4
    }
  } else {
    if (nstage > 0) { LOAD_STAGE(A); STORE_STAGE(0, A); }
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
      const int cur = s & 1;
      if (s + 1 < nstage) LOAD_STAGE(A);    // in flight while this stage is multiplied
      if (!(abl & 4)) wg_compute<T, BO, BKK, MS>(Xs + cur * X_BYTES, Ys + cur * Y_BYTES, acc, wo, wk, l15, g);
      if (s + 1 < nstage) STORE_STAGE(cur ^ 1, A);
      __syncthreads();
    }
  }

  // D[i = k index][j = cout]: lane holds cout = l15, k = g*4 + reg  -> float4 along k in the slab
  float* slab = p.slab + (size_t)split * p.Cout * p.Ktot;
  if (abl & 8) return;
#pragma unroll
  for (int i = 0; i < FK; ++i)
#pragma unroll
    for (int j = 0; j < FO; ++j) {
      int cout = o0 + wo * (BO / 2) + j * 16 + l15;
      int kidx = k0 + wk * (BKK / 2) + i * 16 + g * 4;
      *reinterpret_cast<float4*>(slab + (size_t)cout * p.Ktot + kidx) =
          make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
}

// sum split slabs; write dw[(cout*Cv + c)*ntaps + tap]  (OIHW when tap = r*kw + s) for cout < Coutv, c < Cv:
// operands padded to the GEMM's 64-multiples (DenseNet) reduce straight into the unpadded parameter gradient
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit,
                                    int Cout, int C, int ntaps, int Coutv, int Cv) {
  const int Ktot = C * ntaps;
  const size_t total = (size_t)Cout * Ktot;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slab[k * total + i];
    int cout = (int)(i / Ktot), kk = (int)(i - (size_t)cout * Ktot);
    int tap = kk / C, c = kk - tap * C;
    if (cout < Coutv && c < Cv) dw[((size_t)cout * Cv + c) * ntaps + tap] = s;
  }
}

// float4 variants of the slab reduction (the slabs are ~64 MB per layer: 3 GB per ResNet-50 step).
// Stage 1: slab[nsplit][total] -> part[G][total]; a thread owns ONE float4 column and walks its group's rows with four
// independent accumulators -- no LDS, no barrier (the generic partial_reduce gave a thread 1-2 rows and a barrier).
__global__ __launch_bounds__(256) void wgrad_slab_group_kernel(const float4* __restrict__ slab, float4* __restrict__ part,
                                                               int nsplit, int total4, int per) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= total4) return;
  const int r0 = blockIdx.y * per, r1 = min(nsplit, r0 + per);
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  const float4* src = slab + (size_t)r0 * total4 + c;
  int r = r0;
#define ACC4(a, u) a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
  for (; r + 3 < r1; r += 4, src += (size_t)4 * total4) {
    const float4 u0 = src[0], u1 = src[total4], u2 = src[(size_t)2 * total4], u3 = src[(size_t)3 * total4];
    ACC4(a0, u0) ACC4(a1, u1) ACC4(a2, u2) ACC4(a3, u3)
  }
  for (; r < r1; ++r, src += total4) { const float4 u = src[0]; ACC4(a0, u) }
  part[(size_t)blockIdx.y * total4 + c] =
      make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
}
// Final stage: sum nsplit slabs, 4 consecutive k (= 4 consecutive channels of one tap, C % 4 == 0) per thread, scatter to OIHW
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float4* __restrict__ slab, float* __restrict__ dw, int nsplit,
                                                            int Cout, int C, int ntaps, int Coutv, int Cv) {
  const int Ktot = C * ntaps;
  const int total4 = Cout * (Ktot / 4);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  const float4* src = slab + i;
  int k = 0;
  for (; k + 1 < nsplit; k += 2, src += (size_t)2 * total4) { const float4 u0 = src[0], u1 = src[total4]; ACC4(a0, u0) ACC4(a1, u1) }
  if (k < nsplit) { const float4 u = src[0]; ACC4(a0, u) }
#undef ACC4
  const float v[4] = {a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w};
  const int e = i * 4;
  const int cout = e / Ktot, kk = e - cout * Ktot;
  const int tap = kk / C, c = kk - tap * C;
  if (cout >= Coutv) return;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c + j < Cv) dw[((size_t)cout * Cv + c + j) * ntaps + tap] = v[j];
}

// Many splits (> WG_DIRECT_SPLITS) in ONE launch: 64 float4 columns x 4 row lanes per block; a lane walks every 4th slab with four
// independent accumulators, the lanes meet in LDS, lane 0 scatters to OIHW.  Replaces the group pre-reduction + final reduction
// pair (two launches per layer on the weight-gradient stream, ~25 pairs per ResNet-50 step).  Deterministic (fixed order).
__global__ __launch_bounds__(256) void wgrad_reduce4_lanes_kernel(const float4* __restrict__ slab, float* __restrict__ dw, int nsplit,
                                                                  int Cout, int C, int ntaps, int Coutv, int Cv) {
  __shared__ float4 red[4][64];
  const int Ktot = C * ntaps;
  const int total4 = Cout * (Ktot / 4);
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cx;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
#define ACC4(a, u) a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
  if (i < total4) {
    const float4* src = slab + (size_t)ry * total4 + i;
    int k = ry;
    for (; k + 12 < nsplit; k += 16, src += (size_t)16 * total4) {
      const float4 u0 = src[0], u1 = src[(size_t)4 * total4], u2 = src[(size_t)8 * total4], u3 = src[(size_t)12 * total4];
      ACC4(a0, u0) ACC4(a1, u1) ACC4(a2, u2) ACC4(a3, u3)
    }
    for (; k < nsplit; k += 4, src += (size_t)4 * total4) { const float4 u = src[0]; ACC4(a0, u) }
  }
  red[ry][cx] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (ry != 0 || i >= total4) return;
  float4 t = red[0][cx];
  ACC4(t, red[1][cx]) ACC4(t, red[2][cx]) ACC4(t, red[3][cx])
#undef ACC4
  const float v[4] = {t.x, t.y, t.z, t.w};
  const int e = i * 4;
  const int cout = e / Ktot, kk = e - cout * Ktot;
  const int tap = kk / C, c = kk - tap * C;
  if (cout >= Coutv) return;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c + j < Cv) dw[((size_t)cout * Cv + c + j) * ntaps + tap] = v[j];
}

// ------------------------------------------------------------------------------------------ host
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static void wgrad_plan(int M, int Cout, int Ktot, int MS, int& BO, int& BKK, int& nsplit, int& mps, int ntaps = 1) {
  BO = (Cout % 128 == 0) ? 128 : 64;
  BKK = (Ktot % 128 == 0) ? 128 : 64;
  // 256-wide cout tiles halve the re-reads of the gathered-input operand (env knob for A/B timing)
  static const int bo256 = [] { const char* v = getenv("MMSKIN_WGRAD_BO256"); return v ? atoi(v) : 0; }();
  if (bo256 && Cout % 256 == 0 && BKK == 128) BO = 256;
  MS *= (BO == 256) ? 1 : ((BO + BKK == 128) ? 4 : 2);   // rows per stage (MSF of the kernel)
  int tiles = (Cout / BO) * (Ktot / BKK);
  // workgroups to aim for (never exceeded: one extra workgroup costs a whole extra round of 2 x 256 resident slots).
  // Per-layer sweep on MI355X, isolated and inside the training step (scripts/wgrad_sweep.sh): 1x1 layers and 3x3
  // layers with >= 64 output tiles are fastest with ONE round (<= 512), the other 3x3 layers with two (<= 1024).
  // An LDS-DMA staging variant of this kernel was measured too and was 10 % slower than the register-staged loop kept here.
  static const int target_all = env_int("MMSKIN_WGRAD_BLOCKS", 1024);
  static const int t1 = env_int("MMSKIN_WGRAD_B1", target_all < 512 ? target_all : 512), t3 = env_int("MMSKIN_WGRAD_B3", target_all),
                   t3big = env_int("MMSKIN_WGRAD_B3BIG", t1);
  static const int use_floor = env_int("MMSKIN_WGRAD_FLOOR", 1);
  const int target = ntaps == 1 ? t1 : (tiles >= 64 ? t3big : t3);
  int want = use_floor ? (target / tiles > 0 ? target / tiles : 1) : ceil_div(target, tiles);
  int max_split = M / (MS * 4) > 0 ? M / (MS * 4) : 1;
  nsplit = want < max_split ? want : max_split;
  if (nsplit < 1) nsplit = 1;
  mps = ceil_div(ceil_div(M, nsplit), MS) * MS;
  nsplit = ceil_div(M, mps);
}

#define WG_DIRECT_SPLITS 32   // more splits than this are pre-reduced to WG_GROUPS partial slabs first
#define WG_GROUPS 64        // upper bound; small outputs use more groups so the first stage still fills the chip

// sum the nsplit slabs [Cout][Ktot] (optionally through G group partials stored behind them) into the OIHW gradient
static int reduce_slabs(float* slab, int nsplit, int Cout, int Ktot, float* dw, int C_for_layout, int ntaps_for_layout,
                        int cout_valid, int cin_valid, hipStream_t st) {
  size_t total = (size_t)Cout * Ktot;
  const int total4 = (int)(total / 4);
  const float* src = slab;
  int nsrc = nsplit;
  static const int one_launch = env_int("MMSKIN_WGRAD_REDUCE_ONE", 1);
  if (one_launch && nsplit > WG_DIRECT_SPLITS && C_for_layout % 4 == 0) {
    const int coutv = cout_valid > 0 ? cout_valid : Cout, cv = cin_valid > 0 ? cin_valid : C_for_layout;
    hipLaunchKernelGGL(wgrad_reduce4_lanes_kernel, dim3(ceil_div(total4, 64)), dim3(256), 0, st, reinterpret_cast<const float4*>(slab),
                       dw, nsplit, Cout, C_for_layout, ntaps_for_layout, coutv, cv);
    HIP_CHECK_RET(hipGetLastError());
    return MMSKIN_OK;
  }
  if (nsplit > WG_DIRECT_SPLITS) {   // many small splits: wide first-stage reduction to G <= WG_GROUPS partial slabs
    float* slab2 = slab + (size_t)nsplit * total;
    // >= 256k threads where the output allows it, <= 32 rows per thread, at least 2 rows per group
    int G = ceil_div(262144, total4);
    const int g_lo = ceil_div(nsplit, 32), g_hi = nsplit / 2 < WG_GROUPS ? nsplit / 2 : WG_GROUPS;
    if (G < g_lo) G = g_lo;
    if (G > g_hi) G = g_hi;
    const int per = ceil_div(nsplit, G);
    G = ceil_div(nsplit, per);
    hipLaunchKernelGGL(wgrad_slab_group_kernel, dim3(ceil_div(total4, 256), G), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(slab), reinterpret_cast<float4*>(slab2), nsplit, total4, per);
    HIP_CHECK_RET(hipGetLastError());
    src = slab2;
    nsrc = G;
  }
  const int coutv = cout_valid > 0 ? cout_valid : Cout, cv = cin_valid > 0 ? cin_valid : C_for_layout;
  if (C_for_layout % 4 == 0) {
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(ceil_div(total4, 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(src),
                       dw, nsrc, Cout, C_for_layout, ntaps_for_layout, coutv, cv);
  } else {
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, src, dw, nsrc, Cout, C_for_layout,
                       ntaps_for_layout, coutv, cv);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

template <typename T>
static size_t slab_bytes(int M, int Cout, int Ktot, int ntaps) {
  int BO, BKK, ns, mps;
  wgrad_plan(M, Cout, Ktot, WG<T>::MS, BO, BKK, ns, mps, ntaps);
  if constexpr (sizeof(T) == 2) {   // the ring kernel (wgrad_ring.hip) splits differently
    WgradRingPlan rp;
    for (int simple = 0; simple < 2; ++simple)
      if (wgrad_ring_tile(M, Cout, Ktot, rp, simple != 0) && rp.nsplit > ns) ns = rp.nsplit;
  }
  return (size_t)(ns + (ns > WG_DIRECT_SPLITS ? WG_GROUPS : 0)) * Cout * Ktot * sizeof(float);
}
static size_t wgrad3_slab_bytes(const ConvShape& s);   // all-taps 3x3 path below
size_t conv_wgrad_slab_bytes(const ConvShape& s) {
  int M = s.N * s.OH() * s.OW(), K = s.kh * s.kw * s.Cin;
  size_t a = slab_bytes<float>(M, s.Cout, K, s.kh * s.kw), b = slab_bytes<bf16_t>(M, s.Cout, K, s.kh * s.kw);
  const size_t c = wgrad3_slab_bytes(s);
  if (c > b) b = c;
  return a > b ? a : b;
}
size_t stem_wgrad_slab_bytes(int N, int OH, int OW) {
  size_t a = slab_bytes<float>(N * OH * OW, 64, 256, 8), b = slab_bytes<bf16_t>(N * OH * OW, 64, 256, 8);   // 8 virtual taps
  return a > b ? a : b;
}

template <typename T, int BO, int BKK, int MSF>
static int launch_wg(WgradArgs& a, hipStream_t st) {
  a.nblk_o = a.Cout / BO;
  a.nblk_k = a.Ktot / BKK;
  int grid = a.nblk_o * a.nblk_k * a.nsplit;
  // two-deep prefetch measured no faster in isolation and slower inside the training step (224 VGPRs)
  static const int pf = [] { const char* v = getenv("MMSKIN_WGRAD_PF"); return v ? atoi(v) : 1; }();
  static const int s1 = [] { const char* v = getenv("MMSKIN_WGRAD_S1"); return v ? atoi(v) : 1; }();
  if (pf == 2) hipLaunchKernelGGL((wgrad_kernel<T, BO, BKK, MSF, 2>), dim3(grid), dim3(256), 0, st, a);
  else if (s1 && a.simple1x1 && sizeof(T) == 2 && BO >= 64 && BKK >= 64 && BO + BKK >= 192) hipLaunchKernelGGL((wgrad_kernel<T, BO, BKK, MSF, 1, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad_kernel<T, BO, BKK, MSF, 1>), dim3(grid), dim3(256), 0, st, a);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}


template <typename T>
static int run_wgrad(WgradArgs& a, float* dw, int C_for_layout, int ntaps_for_layout, hipStream_t st,
                     int cout_valid = 0, int cin_valid = 0) {
  ARG_CHECK(a.Cout % 64 == 0 && a.Ktot % 64 == 0, "wgrad: Cout=%d Ktot=%d must be multiples of 64", a.Cout, a.Ktot);
  ARG_CHECK(a.C % DT<T>::EPC == 0, "wgrad: C=%d", a.C);
  ARG_CHECK(a.OW <= 240 && a.OH <= 240, "wgrad: output %dx%d too large for the 16-bit reciprocal pixel stepping", a.OH, a.OW);
  if constexpr (sizeof(T) == 2) {
    WgradRingPlan rp;
    if (wgrad_ring_plan(a, rp)) {
      int rc = wgrad_ring_launch(a, rp, st);
      if (rc) return rc;
      return reduce_slabs(a.slab, a.nsplit, a.Cout, a.Ktot, dw, C_for_layout, ntaps_for_layout, cout_valid, cin_valid, st);
    }
  }
  int BO, BKK;
  wgrad_plan(a.M, a.Cout, a.Ktot, WG<T>::MS, BO, BKK, a.nsplit, a.m_per_split, a.ntaps);
  a.ablate = 0;
#ifdef MMSKIN_ABLATE
  { const char* v = getenv("MMSKIN_WGRAD_ABLATE"); a.ablate = v ? atoi(v) : 0; }
#endif
  int rc;
  if (BO == 256) rc = launch_wg<T, 256, 128, 1>(a, st);
  else if (BO == 128 && BKK == 128) rc = launch_wg<T, 128, 128, 2>(a, st);
  else if (BO == 128) rc = launch_wg<T, 128, 64, 2>(a, st);
  else if (BKK == 128) rc = launch_wg<T, 64, 128, 2>(a, st);
  else rc = launch_wg<T, 64, 64, 4>(a, st);
  if (rc) return rc;
  return reduce_slabs(a.slab, a.nsplit, a.Cout, a.Ktot, dw, C_for_layout, ntaps_for_layout, cout_valid, cin_valid, st);
}


// ------------------------------------------------------------------------------------------ 3x3 / stride 1 / pad 1, all taps per workgroup
// The tapped kernel above gives every (cout tile, tap x cin tile) its own workgroup, so a 3x3 layer streams dY and the
// gathered input through L2 -> LDS nine times (layer1.conv2: 1.85 GB per launch, 32 FLOP per LDS-staged byte).  Here a
// workgroup owns dW[64 cout][9 taps][64 cin]: a stage is R whole image rows (R*W <= 64 pixels); their dY rows and the
// (R+2) x (W+2) zero-padded INPUT WINDOW are staged once, and tap (ty, tx) of pixel (r, x) is window row
// (r+ty)*(W+2) + x+tx -- a constant row offset per tap, no masks, no gather.  LDS rows are 64 channels + 16 B of
// padding (pitch 144 B: conflict-free for the transposing reads without an XOR swizzle, so tap offsets stay additive).
// 72 MFMAs per wave and stage against ~30 KB of loads (140 FLOP/B).  bf16 only; partial slabs as above.
struct Wgrad3Args {
  const bf16_t* dy; const bf16_t* in; float* slab;
  int N, H, W, C, Cout;
  int R, spi;                       // image rows per stage, stages per image
  int total_stages, stages_per_split, nsplit;
  int nblk_o, nblk_c;
  int yrows;                        // (R+2)*(W+2)
};
#define W3_PITCH 144
#define W3_XROWS 64
#define W3_YROWS 176
#define W3_LDS_BYTES (2 * (W3_XROWS + W3_YROWS) * W3_PITCH)

__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(const Wgrad3Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  unsigned char* Xs = smem3;
  unsigned char* Ys = smem3 + 2 * W3_XROWS * W3_PITCH;
  constexpr int X_BYTES = W3_XROWS * W3_PITCH, Y_BYTES = W3_YROWS * W3_PITCH;
  const int tid = threadIdx.x;
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = tile % p.nblk_c; tile /= p.nblk_c;
  const int ob = tile % p.nblk_o;
  const int split = tile / p.nblk_o;
  const int o0 = ob * 64, c0 = cb * 64;
  const int W = p.W, H = p.H, W2 = W + 2, R = p.R;
  const int st_begin = split * p.stages_per_split;
  const int st_end = min(p.total_stages, st_begin + p.stages_per_split);
  int img = st_begin / p.spi, y0 = (st_begin - img * p.spi) * R;

  // ---- loaders.  dY: 64 pixel slots x 8 chunks = 2 per thread; window: up to 176 rows x 8 chunks = 6 per thread
  const int ch = tid & 7;
  int x_r[2], x_pix[2];
  bool x_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = (tid >> 3) + 32 * i;
    x_r[i] = slot / W;
    x_pix[i] = slot;                                   // = r*W + x: pixel offset inside the stage's rows
    x_ok[i] = slot < R * W;
  }
  int y_wr[6], y_pix[6];
  bool y_ok[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int j = (tid >> 3) + 32 * i;
    const int wr = j / W2, wc = j - wr * W2;
    y_wr[i] = wr - 1;                                  // image row relative to y0
    y_pix[i] = (wr - 1) * W + (wc - 1);
    y_ok[i] = j < p.yrows && wc >= 1 && wc <= W;
  }
  const u32x4_t z4 = {0, 0, 0, 0};
  u32x4_t rx0 = z4, rx1 = z4, ry0 = z4, ry1 = z4, ry2 = z4, ry3 = z4, ry4 = z4, ry5 = z4;
#define W3_LOADX(i, Rg)                                                                                         \
  {                                                                                                             \
    const bool ok = x_ok[i] && (y0 + x_r[i] < H);                                                               \
    const bf16_t* src = p.dy + ((size_t)((img * H + y0) * W + x_pix[i]) * p.Cout + o0 + ch * 8);                \
    Rg = z4;                                                                                                    \
    if (ok) Rg = *reinterpret_cast<const u32x4_t*>(src);                                                        \
  }
#define W3_LOADY(i, Rg)                                                                                         \
  {                                                                                                             \
    const bool ok = y_ok[i] && (unsigned)(y0 + y_wr[i]) < (unsigned)H;                                          \
    const bf16_t* src = p.in + ((int64_t)((img * H + y0) * W + y_pix[i]) * p.C + c0 + ch * 8);                  \
    Rg = z4;                                                                                                    \
    if (ok) Rg = *reinterpret_cast<const u32x4_t*>(src);                                                        \
  }
#define W3_LOAD_STAGE()                                                                                         \
  do {                                                                                                          \
    W3_LOADX(0, rx0) W3_LOADX(1, rx1)                                                                           \
    W3_LOADY(0, ry0) W3_LOADY(1, ry1) W3_LOADY(2, ry2) W3_LOADY(3, ry3) W3_LOADY(4, ry4) W3_LOADY(5, ry5)       \
  } while (0)
#define W3_STX(buf, i, Rg) *reinterpret_cast<u32x4_t*>(Xs + (buf) * X_BYTES + ((tid >> 3) + 32 * i) * W3_PITCH + ch * 16) = Rg;
#define W3_STY(buf, i, Rg)                                                                                      \
  if ((tid >> 3) + 32 * i < W3_YROWS) *reinterpret_cast<u32x4_t*>(Ys + (buf) * Y_BYTES + ((tid >> 3) + 32 * i) * W3_PITCH + ch * 16) = Rg;
#define W3_STORE_STAGE(buf)                                                                                     \
  do {                                                                                                          \
    W3_STX(buf, 0, rx0) W3_STX(buf, 1, rx1)                                                                     \
    W3_STY(buf, 0, ry0) W3_STY(buf, 1, ry1) W3_STY(buf, 2, ry2) W3_STY(buf, 3, ry3) W3_STY(buf, 4, ry4) W3_STY(buf, 5, ry5) \
  } while (0)
#define W3_ADVANCE()                                                                                            \
  do { y0 += R; if (y0 >= H) { y0 = 0; ++img; } } while (0)

  // ---- fragment addressing (transposing reads: lane (q, pp) addresses row q, columns 4pp..4pp+3 of a 4 x 16 block)
  const int wid = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int q = l15 >> 2, pp = l15 & 3;
  const int wo = wid >> 1, wk = wid & 1;               // wave: cout half, cin half
  int xa[2][2], ya[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 32 * ks + 8 * g + q + 4 * h;       // pixel slot
      xa[ks][h] = k * W3_PITCH + (wo * 32 + 4 * pp) * 2;
      const int r = k / W, x = k - r * W;
      const int brow = k < R * W ? r * W2 + x : 0;     // slots past the stage have dY = 0: any finite window row will do
      ya[ks][h] = brow * W3_PITCH + (wk * 32 + 4 * pp) * 2;
    }
  f32x4_t acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#define W3_COMPUTE(buf)                                                                                         \
  do {                                                                                                          \
    const unsigned char* Xb = Xs + (buf) * X_BYTES;                                                             \
    const unsigned char* Yb = Ys + (buf) * Y_BYTES;                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                          \
      uint4 fx[2];                                                                                              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                           \
        uint2 lo = lds_read_tr16_b64(Xb + xa[ks][0] + j * 32);                                                  \
        uint2 hi = lds_read_tr16_b64(Xb + xa[ks][1] + j * 32);                                                  \
        fx[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);                                                             \
      }                                                                                                         \
      _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                           \
        const int tb = ((t / 3) * W2 + (t % 3)) * W3_PITCH;                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                         \
          uint2 lo = lds_read_tr16_b64(Yb + ya[ks][0] + tb + i * 32);                                           \
          uint2 hi = lds_read_tr16_b64(Yb + ya[ks][1] + tb + i * 32);                                           \
          const uint4 fy = make_uint4(lo.x, lo.y, hi.x, hi.y);                                                  \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fy),            \
                                                                   __builtin_bit_cast(bf16x8_t, fx[j]), acc[t][i][j], 0, 0, 0); \
        }                                                                                                       \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

  if (st_begin < st_end) { W3_LOAD_STAGE(); W3_STORE_STAGE(0); }
  __syncthreads();
  for (int s = st_begin; s < st_end; ++s) {
    const int cur = (s - st_begin) & 1;
    if (s + 1 < st_end) { W3_ADVANCE(); W3_LOAD_STAGE(); }     // in flight while this stage is multiplied
    W3_COMPUTE(cur);
    if (s + 1 < st_end) W3_STORE_STAGE(cur ^ 1);
    __syncthreads();
  }

  // D[cin][cout]: lane holds cout = l15, cin = 4g + reg -> float4 along k in the slab [cout][tap*C + cin]
  const int Ktot = 9 * p.C;
  float* slab = p.slab + (size_t)split * p.Cout * Ktot;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cout = o0 + wo * 32 + j * 16 + l15;
        const int kidx = t * p.C + c0 + wk * 32 + i * 16 + g * 4;
        *reinterpret_cast<float4*>(slab + (size_t)cout * Ktot + kidx) =
            make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
      }
}

static bool wgrad3_plan(const ConvShape& s, Wgrad3Args& a) {
  static const int enabled = env_int("MMSKIN_WGRAD_3X3", 1);
  if (!enabled || s.kh != 3 || s.kw != 3 || s.stride != 1 || s.pad != 1) return false;
  if (s.Cin % 64 || s.Cout % 64 || s.W > 64 || s.W < 1) return false;
  int R = 64 / s.W;
  if (R > s.H) R = s.H;
  if ((R + 2) * (s.W + 2) > W3_YROWS) return false;
  a.N = s.N; a.H = s.H; a.W = s.W; a.C = s.Cin; a.Cout = s.Cout;
  a.R = R; a.spi = ceil_div(s.H, R);
  a.total_stages = s.N * a.spi;
  a.nblk_o = s.Cout / 64; a.nblk_c = s.Cin / 64;
  a.yrows = (R + 2) * (s.W + 2);
  const int tiles = a.nblk_o * a.nblk_c;
  static const int target = env_int("MMSKIN_WGRAD3_BLOCKS", 512);      // one round of 2 workgroups per CU
  int ns = target / tiles > 0 ? target / tiles : 1;
  // every split writes a whole 64 x 9 x 64 fp32 tile (147 KB) and the reduction reads it back: a split should multiply at least
  // ~16 stages before it does.  DenseNet's 32-channel conv2 at 14^2 / 7^2 had 4 / 1 stages per split -- 6.2 GB of slab writes and
  // most of 7.9 GB of reduction reads per step (profiles/r03_step_traffic_densenet169-metablock.txt); ResNet-50's 3x3 layers have 28 - 32.
  static const int min_stages = env_int("MMSKIN_WGRAD3_MIN_STAGES", 16);
  if (min_stages > 1 && ns > a.total_stages / min_stages) ns = a.total_stages / min_stages > 0 ? a.total_stages / min_stages : 1;
  if (ns > a.total_stages) ns = a.total_stages;
  a.stages_per_split = ceil_div(a.total_stages, ns);
  a.nsplit = ceil_div(a.total_stages, a.stages_per_split);
  return true;
}
static size_t wgrad3_slab_bytes(const ConvShape& s) {
  Wgrad3Args a = {};
  int ns = wgrad3_plan(s, a) ? a.nsplit : 0, nr = 0;
  if (wgrad3_ring_takes(s, &nr) && nr > ns) ns = nr;
  if (!ns) return 0;
  return (size_t)(ns + (ns > WG_DIRECT_SPLITS ? WG_GROUPS : 0)) * s.Cout * 9 * s.Cin * sizeof(float);
}
static int launch_wgrad3(const ConvShape& s, Wgrad3Args& a, const bf16_t* dout, const bf16_t* in, float* slab, float* dw,
                         hipStream_t st, int cout_valid, int cin_valid) {
  a.dy = dout; a.in = in; a.slab = slab;
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      W3_LDS_BYTES));
    attr_done = true;
  }
  hipLaunchKernelGGL(wgrad3x3_kernel, dim3(a.nblk_o * a.nblk_c * a.nsplit), dim3(256), W3_LDS_BYTES, st, a);
  HIP_CHECK_RET(hipGetLastError());
  return reduce_slabs(slab, a.nsplit, s.Cout, 9 * s.Cin, dw, s.Cin, 9, cout_valid, cin_valid, st);
}

template <typename T>
int launch_conv_wgrad(const ConvShape& s, const T* dout, const T* in, float* slab, float* dw_oihw,
                      hipStream_t st, int cout_valid, int cin_valid) {
  ARG_CHECK(s.kh * s.kw <= MMSKIN_MAX_TAPS, "wgrad: too many taps");
  if constexpr (sizeof(T) == 2) {
    int nsr = 0;
    if (wgrad3_ring_takes(s, &nsr)) {
      int rc = launch_wgrad3_ring(s, dout, in, slab, st, &nsr);
      if (rc) return rc;
      return reduce_slabs(slab, nsr, s.Cout, 9 * s.Cin, dw_oihw, s.Cin, 9, cout_valid, cin_valid, st);
    }
    Wgrad3Args a3 = {};
    if (wgrad3_plan(s, a3)) return launch_wgrad3(s, a3, dout, in, slab, dw_oihw, st, cout_valid, cin_valid);
  }
  WgradArgs a = {};
  a.dy = dout; a.in = in; a.slab = slab;
  a.N = s.N; a.IH = s.H; a.IW = s.W; a.C = s.Cin; a.Cpitch = s.Cin;
  a.OH = s.OH(); a.OW = s.OW();
  a.Cout = s.Cout; a.Ktot = s.kh * s.kw * s.Cin;
  a.Sy = s.stride; a.Sx = s.stride; a.ntaps = s.kh * s.kw;
  a.M = s.N * a.OH * a.OW;
  a.simple1x1 = (s.kh == 1 && s.kw == 1 && s.stride == 1 && s.pad == 0) ? 1 : 0;
  for (int r = 0; r < s.kh; ++r)
    for (int q = 0; q < s.kw; ++q) {
      a.offy[r * s.kw + q] = (int8_t)(r - s.pad);
      a.offx[r * s.kw + q] = (int8_t)(q - s.pad);
    }
  return run_wgrad<T>(a, dw_oihw, s.Cin, s.kh * s.kw, st, cout_valid, cin_valid);
}

template <typename T>
int launch_stem_conv_wgrad(int N, int OH, int OW, int Hp, int Wp, const T* dout, const T* img4,
                           float* slab, float* dwv, hipStream_t st) {
  WgradArgs a = {};
  a.dy = dout; a.in = img4; a.slab = slab;
  a.N = N; a.IH = Hp; a.IW = Wp / 2; a.C = 32; a.Cpitch = 8;
  a.OH = OH; a.OW = OW; a.Cout = 64; a.Ktot = 256;
  a.Sy = 2; a.Sx = 1; a.ntaps = 8;
  a.M = N * OH * OW;
  for (int r = 0; r < 8; ++r) { a.offy[r] = (int8_t)r; a.offx[r] = 0; }
  // layout trick: C=256, ntaps=1 makes the reducer write dwv[cout][256] unchanged
  return run_wgrad<T>(a, dwv, 256, 1, st);
}

size_t vgg_first_wgrad_slab_bytes(int N, int H, int W) {
  size_t a = slab_bytes<float>(N * H * W, 64, 128, 4), b = slab_bytes<bf16_t>(N * H * W, 64, 128, 4);   // 4 virtual taps
  return a > b ? a : b;
}
template <typename T>
int launch_vgg_first_conv_wgrad(int N, int H, int W, int Hp, int Wp, const T* dout, const T* img8, float* slab,
                                float* dwv, hipStream_t st, int stride) {
  const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
  WgradArgs a = {};
  a.dy = dout; a.in = img8; a.slab = slab;
  a.N = N; a.IH = Hp; a.IW = Wp; a.C = 32; a.Cpitch = 8;
  a.OH = OH; a.OW = OW; a.Cout = 64; a.Ktot = 128;
  a.Sy = stride; a.Sx = stride; a.ntaps = 4;
  a.M = N * OH * OW;
  for (int r = 0; r < 4; ++r) { a.offy[r] = (int8_t)(r < 3 ? r : 0); a.offx[r] = 0; }
  return run_wgrad<T>(a, dwv, 128, 1, st);
}

#define INST(T)                                                                                   \
  template int launch_vgg_first_conv_wgrad<T>(int, int, int, int, int, const T*, const T*, float*, float*, hipStream_t, int); \
  template int launch_conv_wgrad<T>(const ConvShape&, const T*, const T*, float*, float*, hipStream_t, int, int); \
  template int launch_stem_conv_wgrad<T>(int, int, int, int, int, const T*, const T*, float*, float*, hipStream_t);
INST(float)
INST(bf16_t)

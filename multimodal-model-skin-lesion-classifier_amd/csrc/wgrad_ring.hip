// Weight-gradient GEMM, ring form (round 4):  dW[cout][k] = sum_m dY[m][cout] * gather(in)[m][k]      (bf16 operands, fp32 accumulate)
//
// Replaces ATen/cuDNN conv_backward(weight) reached through the reference's torchvision backbone
// (loadImageModelClassifier.py:65-75 via loss.backward(), train_pad_20.py:112).
//
// The register-staged kernel of wgrad.hip keeps ONE 32-pixel stage in flight per workgroup behind a __syncthreads() and is
// load-latency bound by its own ablation (-38 % without the loads, -9 % without the MFMAs).  Here the operands arrive by
// `buffer_load_dwordx4 ... offen lds` into a ring of NIT iterations with NIT - 1 of them (48 - 72 KB per CU) in flight behind a
// COUNTED s_waitcnt vmcnt and a raw s_barrier; no staging VGPRs, no ds_write, no zero page (rows past the split end and taps
// outside the image get an out-of-range offset, for which the buffer unit writes zeros).
//
// One workgroup = 8 waves = G pixel groups of WO x WK waves; a wave owns a 64 x 64 piece of the dW tile (BO = 64 WO couts x
// BK = 64 WK k-indices).  An iteration is G stages of 32 pixels, one per group: the groups multiply DIFFERENT pixels of the SAME
// dW tile and meet in LDS before the slab write, so a CU filled by one workgroup writes one fp32 slab tile where two 4-wave
// workgroups wrote two.  Stage image in LDS = the rows as they sit in memory ([pixel][channel], 16-byte chunks), the bank swizzle
// applied to the SOURCE chunk (LDS-DMA writes lane-linear); fragments by ds_read_b64_tr_b16 exactly as in wgrad.hip.
#include <stdlib.h>

#include "conv.h"

typedef uint32_t srd_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// chunk swizzle of a tile row (same function as wg_swz<bf16_t, PITCH> in wgrad.hip: conflict-free transposing reads)
template <int PITCH> __device__ __forceinline__ int ring_swz(int row) {
  if constexpr (PITCH % 256 == 0) return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
  else return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
}

__device__ __forceinline__ uint2 ring_tr16(const unsigned char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}

#define RING_OOB 0xF0000000u   // > any num_records the launcher admits

// one LDS-DMA piece: 64 lanes x 16 B -> 1 KB at LDS byte address `dst` (wave-uniform); M0 write + one wait state + the load
#define RING_DMA(voff, srd, dst)                                                                                   \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"((uint32_t)(dst)),   \
               "v"((uint32_t)(voff)), "s"(srd) : "memory")

template <int N> __device__ __forceinline__ void ring_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// S1: 1x1 / stride 1 / no padding -- gathered-input row m IS pixel m (offsets advance by a constant; no pixel stepping)
// STAG: the two halves of the workgroup (waves 0-3 / 4-7: one wave of each on every SIMD) run half an iteration apart -- two barriers per
// iteration, one half issues its DMA pieces and reads its fragments while the other half multiplies.  With every wave behind ONE barrier per
// 16 MFMAs, both waves of a SIMD read together and then queue on the matrix pipe together: ~55 % of it on the MFMA-bound layers 3 - 4.
template <int WO, int WK, int NIT, bool S1, bool STAG = false>
__global__ __launch_bounds__(512) void wgrad_ring_kernel(const WgradArgs p) {
  constexpr int BO = 64 * WO, BK = 64 * WK, WPG = WO * WK, G = 8 / WPG;
  constexpr int PX = BO * 2, PY = BK * 2;                       // LDS row pitches (bytes)
  constexpr int XB = 32 * PX, YB = 32 * PY, STG = XB + YB;      // one 32-pixel stage: dY rows, then input rows
  constexpr int ITB = G * STG;                                  // one iteration (G stages)
  constexpr int CPRX = PX / 16, CPRY = PY / 16;
  constexpr int PCX = XB / 1024, PCY = YB / 1024;               // DMA pieces per stage
  constexpr int NXW = G * PCX / 8, NYW = G * PCY / 8, NPW = NXW + NYW;   // pieces per wave and iteration
  static_assert(WPG == 2 || WPG == 4 || WPG == 8, "2, 4 or 8 waves per pixel group");
  static_assert((G * PCX) % 8 == 0 && (G * PCY) % 8 == 0, "pieces must divide over 8 waves");
  static_assert(NIT >= 2 && (NIT - 2) * NPW < 64, "ring depth");
  static_assert(!STAG || NIT >= 3, "the late half waits one iteration deeper");
  static_assert((G - 1) * BO * BK * 4 <= NIT * ITB, "group reduction overlays the ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int kb = tile % p.nblk_k; tile /= p.nblk_k;
  const int ob = tile % p.nblk_o;
  const int split = tile / p.nblk_o;
  const bool ytile = ob >= p.nblk_o_main;                       // Gram tile: the "dY" operand is the input tensor itself
  const int o0 = ob * BO, k0 = kb * BK;
  const int xcol0 = ytile ? (ob - p.nblk_o_main) * BO : o0;     // first channel of the X rows this tile reads
  const int xpitch = ytile ? p.Cpitch : p.Cout;
  const int xcols = ytile ? p.gram_cols : p.Cout;
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int nit = (m_end - m_begin + G * 32 - 1) / (G * 32);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

  srd_t srdX, srdY;
  {
    const uint64_t bx = (uint64_t)(uintptr_t)p.dy, by = (uint64_t)(uintptr_t)p.in;
    srdX = srd_t{(uint32_t)bx, (uint32_t)(bx >> 32) & 0xffffu, (uint32_t)((uint64_t)p.M * p.Cout * 2), 0x00020000u};
    srdY = srd_t{(uint32_t)by, (uint32_t)(by >> 32) & 0xffffu, (uint32_t)((uint64_t)p.N * p.IH * p.IW * p.Cpitch * 2), 0x00020000u};
  }

  // ---- dY pieces of this wave: piece jx = wid + 8 i -> stage (group) jx / PCX, 1 KB block jx % PCX
  uint32_t x_off[NXW], x_dst[NXW];
  bool x_ok[NXW];
  const uint32_t x_lim = (uint32_t)m_end * (uint32_t)xpitch * 2u, x_step = (uint32_t)(G * 32) * (uint32_t)xpitch * 2u;
  const srd_t srdXs = ytile ? srdY : srdX;
#pragma unroll
  for (int i = 0; i < NXW; ++i) {
    const int jx = wid + 8 * i, gg = jx / PCX, blk = jx - gg * PCX;
    const int L = blk * 64 + lane, r = L / CPRX, c = L - r * CPRX;
    const int sc = c ^ ring_swz<PX>(r);
    x_off[i] = (uint32_t)((m_begin + gg * 32 + r) * xpitch + xcol0 + sc * 8) * 2u;
    x_ok[i] = xcol0 + sc * 8 < xcols;
    x_dst[i] = (uint32_t)(gg * STG + blk * 1024);
  }
  // ---- input pieces
  uint32_t y_off[NYW], y_dst[NYW];
  const uint32_t y_lim = (uint32_t)m_end * (uint32_t)p.Cpitch * 2u, y_step = (uint32_t)(G * 32) * (uint32_t)p.Cpitch * 2u;
  int y_m[NYW], y_img[NYW], y_oy[NYW], y_ox[NYW], y_c[NYW], y_dy[NYW], y_dx[NYW];
  const int OWr = p.OW, OHr = p.OH;
  const int inv_ow = (65536 + OWr - 1) / OWr, inv_oh = (65536 + OHr - 1) / OHr;
#pragma unroll
  for (int i = 0; i < NYW; ++i) {
    const int jy = wid + 8 * i, gg = jy / PCY, blk = jy - gg * PCY;
    const int L = blk * 64 + lane, r = L / CPRY, c = L - r * CPRY;
    const int sc = c ^ ring_swz<PY>(r);
    y_dst[i] = (uint32_t)(gg * STG + XB + blk * 1024);
    const int m = m_begin + gg * 32 + r;
    if constexpr (S1) {
      y_off[i] = (uint32_t)(m * p.Cpitch + k0 + sc * 8) * 2u;
    } else {
      const int kidx = k0 + sc * 8;
      const int tap = kidx / p.C;
      y_c[i] = kidx - tap * p.C;
      y_dy[i] = p.offy[tap]; y_dx[i] = p.offx[tap];
      const int ohw = OHr * OWr;
      const int img = m / ohw, rem = m - img * ohw;
      y_m[i] = m; y_img[i] = img; y_oy[i] = rem / OWr; y_ox[i] = rem - y_oy[i] * OWr;
    }
  }

  // all pieces of iteration `it` into ring buffer `buf` (LDS byte offset); always issued, all-zero past the split end, so the
  // outstanding count in front of every wait is the same
#define RING_ISSUE(bufoff)                                                                                          \
  do {                                                                                                              \
    const uint32_t b_ = lds0 + (bufoff);                                                                            \
    _Pragma("unroll") for (int i = 0; i < NXW; ++i) {                                                               \
      const uint32_t v_ = (x_ok[i] && x_off[i] < x_lim) ? x_off[i] : RING_OOB;                                      \
      RING_DMA(v_, srdXs, b_ + x_dst[i]);                                                                           \
      x_off[i] += x_step;                                                                                           \
    }                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NYW; ++i) {                                                               \
      if constexpr (S1) {                                                                                           \
        const uint32_t v_ = y_off[i] < y_lim ? y_off[i] : RING_OOB;                                                 \
        RING_DMA(v_, srdY, b_ + y_dst[i]);                                                                          \
        y_off[i] += y_step;                                                                                         \
      } else {                                                                                                      \
        const int iy = y_oy[i] * p.Sy + y_dy[i], ix = y_ox[i] * p.Sx + y_dx[i];                                     \
        const bool ok = y_m[i] < m_end && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;           \
        const uint32_t o_ = (uint32_t)(((y_img[i] * p.IH + iy) * p.IW + ix) * p.Cpitch + y_c[i]) * 2u;              \
        RING_DMA(ok ? o_ : RING_OOB, srdY, b_ + y_dst[i]);                                                          \
        /* next iteration's pixel: 16-bit reciprocal multiplies, exact for the ranges the launcher admits */        \
        y_m[i] += G * 32;                                                                                           \
        int nx = y_ox[i] + G * 32;                                                                                  \
        const int qy = (nx * inv_ow) >> 16;                                                                         \
        nx -= qy * OWr;                                                                                             \
        const int ny = y_oy[i] + qy;                                                                                \
        const int qi = (ny * inv_oh) >> 16;                                                                         \
        y_ox[i] = nx; y_oy[i] = ny - qi * OHr; y_img[i] += qi;                                                      \
      }                                                                                                             \
    }                                                                                                               \
  } while (0)

  // ---- fragment addressing: group gg, wave (wo, wk) inside it
  const int gg = wid / WPG, wpos = wid - gg * WPG;
  const int wo = wpos / WK, wk = wpos - wo * WK;
  const int l15 = lane & 15, g = lane >> 4, q = l15 >> 2, pp = l15 & 3;
  const int r0 = 8 * g + q, r1 = r0 + 4;
  uint32_t xa[2][4], ya[2][4];   // byte offsets inside a stage
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cx = (wo * 64 + j * 16 + 4 * pp) * 2, cy = (wk * 64 + j * 16 + 4 * pp) * 2;
    xa[0][j] = r0 * PX + (((cx >> 4) ^ ring_swz<PX>(r0)) << 4) + (cx & 15);
    xa[1][j] = r1 * PX + (((cx >> 4) ^ ring_swz<PX>(r1)) << 4) + (cx & 15);
    ya[0][j] = XB + r0 * PY + (((cy >> 4) ^ ring_swz<PY>(r0)) << 4) + (cy & 15);
    ya[1][j] = XB + r1 * PY + (((cy >> 4) ^ ring_swz<PY>(r1)) << 4) + (cy & 15);
  }
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // Gram tiles: column sums of the X rows through one MFMA per cout fragment against a fragment of ones (waves with wk == 0)
  f32x4_t acc1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc1[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = ytile && wk == 0 && p.colsum != nullptr;
  const uint4 ones8 = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);

  // ---- main loop.  RAW: buffer (it % NIT) was filled by pieces issued NIT - 1 iterations ago; every wave waits for its own
  // (counted vmcnt leaves the NIT - 2 younger iterations in flight), then the barrier makes everybody's visible.  WAR: the
  // buffer refilled after the barrier of iteration `it` was read in iteration it - 1, and every wave's reads had returned
  // (the MFMAs consumed them) before it reached this barrier.
#pragma unroll
  for (int s = 0; s < NIT - 1; ++s) RING_ISSUE(s * ITB);
  int buf_rd = 0, buf_wr = (NIT - 1) * ITB;
  // STAG timeline (B = barrier; the late half runs one barrier behind):
  //   early:  B  issue(it+NIT-1) read(it) lgkm0   B  mfma(it)                      B  issue ...
  //   late:      mfma(it-1)                        B  issue(it+NIT-1) read(it) lgkm0 B  mfma(it) ...
  // RAW: a half reads iteration j after its own first barrier of j; its own pieces were waited for in front of that barrier, the
  // OTHER half's in front of the other half's previous barrier -- which for the early half reading j is the late half's first barrier
  // of j - 1: the late half therefore waits one iteration deeper (NIT - 3 instead of NIT - 2 iterations left in flight).
  // WAR: reads complete (lgkmcnt(0)) before the second barrier, and a buffer is refilled two or more barriers after that.
  const bool late = STAG && wid >= 4;
  if (late) {   // the early half reads iteration 0 behind this barrier: the late half's pieces of it must have landed
    ring_wait_vm<(NIT - 2) * NPW>();
    asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
  }
  for (int it = 0; it < nit; ++it) {
    if constexpr (STAG) {
      if (late) ring_wait_vm<(NIT - 3) * NPW>(); else ring_wait_vm<(NIT - 2) * NPW>();
    } else {
      ring_wait_vm<(NIT - 2) * NPW>();
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    RING_ISSUE(buf_wr);
    const unsigned char* sb = smem + buf_rd + gg * STG;
    uint4 fx[4], fy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint2 lo = ring_tr16(sb + xa[0][j]), hi = ring_tr16(sb + xa[1][j]);
      fx[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint2 lo = ring_tr16(sb + ya[0][i]), hi = ring_tr16(sb + ya[1][i]);
      fy[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    if constexpr (STAG) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fy[i]), __builtin_bit_cast(bf16x8_t, fx[j]),
                                                            acc[i][j], 0, 0, 0);
    if (do_colsum) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ones8), __builtin_bit_cast(bf16x8_t, fx[j]), acc1[j], 0, 0, 0);
    }
    buf_rd = buf_rd + ITB == NIT * ITB ? 0 : buf_rd + ITB;
    buf_wr = buf_wr + ITB == NIT * ITB ? 0 : buf_wr + ITB;
  }
  if (STAG && !late) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }   // the early half's share of the late half's last barrier
#undef RING_ISSUE
  // the trailing (all-zero) pieces still land in the ring: drain them before the ring is reused
  ring_wait_vm<0>();
  __syncthreads();

  if (do_colsum && g == 0) {   // every k row of acc1 holds the same sums: row 0 = lanes 0-15, register 0; one partial per (split, group)
    const int wpad = (p.nblk_o - p.nblk_o_main) * BO;
    float* cs = p.colsum + (size_t)(split * G + gg) * wpad + xcol0 + wo * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j * 16 + l15] = acc1[j][0];
  }
  // ---- the G groups hold partial sums of the same tile: groups 1.. park theirs in LDS, group 0 adds
  if constexpr (G > 1) {
    float4* red = reinterpret_cast<float4*>(smem);
    if (gg > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          red[((gg - 1) * WPG + wpos) * 1024 + (i * 4 + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    __syncthreads();
    if (gg > 0) return;
#pragma unroll
    for (int g2 = 1; g2 < G; ++g2)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 t = red[((g2 - 1) * WPG + wpos) * 1024 + (i * 4 + j) * 64 + lane];
          acc[i][j][0] += t.x; acc[i][j][1] += t.y; acc[i][j][2] += t.z; acc[i][j][3] += t.w;
        }
  }
  // D[i = k index][j = cout]: lane holds cout = l15, k = 4 g + reg -> float4 along k in the slab
  float* slab = p.slab + (size_t)split * (p.nblk_o * BO) * p.Ktot;   // slab rows = Cout (+ the Gram tiles' rows)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cout = o0 + wo * 64 + j * 16 + l15;
      const int kidx = k0 + wk * 64 + i * 16 + g * 4;
      *reinterpret_cast<float4*>(slab + (size_t)cout * p.Ktot + kidx) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
}

// ------------------------------------------------------------------------------------------ host
static int ring_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

bool wgrad_ring_tile(int M, int Cout, int Ktot, WgradRingPlan& r, bool simple) {
  static const int enabled = ring_env("MMSKIN_WGRAD_RING", 1);
  if (!enabled || Cout % 64 || Ktot % 64) return false;
  // 1x1 / stride 1 layers: 128 x 128 tiles with two pixel groups wherever they fit -- half the slab bytes per workgroup for 4/3 of the
  // L2 -> LDS fill: l2.c3 56.5 -> 48.6 us, l3.c3 43.7 -> 40.3, l4.c3 41.8 -> 38.6 (profiles/r04_experiments.txt (4)); the gathering
  // layers (stride-2 1x1, tapped 3x3) lose with it (l4.c2a 97 -> 134 us) and keep the 256 x 128 tile.  MMSKIN_WGRAD_RING_G2=0: off
  static const int prefer_g2 = ring_env("MMSKIN_WGRAD_RING_G2", 1);
  if (prefer_g2 && simple && Cout % 128 == 0 && Ktot % 128 == 0) { r.wo = 2; r.wk = 2; }
  else if (Cout % 256 == 0 && Ktot % 128 == 0) { r.wo = 4; r.wk = 2; }
  else if (Cout % 128 == 0 && Ktot % 256 == 0) { r.wo = 2; r.wk = 4; }
  else if (Cout % 128 == 0 && Ktot % 128 == 0) { r.wo = 2; r.wk = 2; }
  else if (Cout % 256 == 0) { r.wo = 4; r.wk = 1; }
  else if (Ktot % 256 == 0) { r.wo = 1; r.wk = 4; }
  else if (Cout % 128 == 0) { r.wo = 2; r.wk = 1; }     // Ktot = 64 x odd (DenseNet's 1x1 bottlenecks): 128 x 64 tiles, four pixel groups
  else if (Ktot % 128 == 0) { r.wo = 1; r.wk = 2; }
  else return false;
  const int G = 8 / (r.wo * r.wk), step = G * 32;
  const int tiles = (Cout / (64 * r.wo)) * (Ktot / (64 * r.wk));
  static const int target = ring_env("MMSKIN_WGRAD_RING_BLOCKS", 256);   // one 8-wave workgroup per CU
  int ns = target / tiles > 0 ? target / tiles : 1;
  const int max_split = M / (step * 8) > 0 ? M / (step * 8) : 1;         // at least 8 iterations per workgroup
  if (ns > max_split) ns = max_split;
  r.mps = ceil_div(ceil_div(M, ns), step) * step;
  r.nsplit = ceil_div(M, r.mps);
  r.s1 = false;
  r.gram_tiles = 0;
  return true;
}

// ---- dY^T in, in^T in and colsum(in) in one launch (see conv.h)
// mode 0: all three; mode 1: dY^T in only (the Gram matrix was kept by the forward pass); mode 2: in^T in and colsum(in) only (forward:
// the statistics of an expanding 1x1 convolution's output follow from the Gram matrix of its input, abn.hip gram_stats)
bool wgrad_gram_plan(int M, int Cout, int Cin, WgradRingPlan& r, int mode) {
  if (mode == 2) {
    if (Cin % 64 || Cin > 256) return false;
    r.wo = 2; r.wk = Cin % 128 == 0 ? 2 : 1;
    Cout = 0;
  } else if (!wgrad_ring_tile(M, Cout, Cin, r, mode == 1)) return false;
  const int BO = 64 * r.wo, G = 8 / (r.wo * r.wk), step = G * 32;
  if (Cin % 8 || (uint64_t)M * Cout * 2 >= 0xE0000000ull || (uint64_t)M * Cin * 2 >= 0xE0000000ull) return false;
  r.gram_tiles = mode == 1 ? 0 : ceil_div(Cin, BO);
  const int tiles = (Cout / BO + r.gram_tiles) * (Cin / (64 * r.wk));
  static const int target = ring_env("MMSKIN_WGRAD_RING_BLOCKS", 256);
  int ns = target / tiles > 0 ? target / tiles : 1;
  const int max_split = M / (step * 8) > 0 ? M / (step * 8) : 1;
  if (ns > max_split) ns = max_split;
  r.mps = ceil_div(ceil_div(M, ns), step) * step;
  r.nsplit = ceil_div(M, r.mps);
  r.s1 = true;
  return true;
}
static size_t gram_colsum_floats(const WgradRingPlan& r) { return (size_t)r.nsplit * (8 / (r.wo * r.wk)) * r.gram_tiles * 64 * r.wo; }
size_t wgrad_gram_slab_bytes(int M, int Cout, int Cin, int mode) {
  WgradRingPlan r;
  if (!wgrad_gram_plan(M, Cout, Cin, r, mode)) return 0;
  const size_t rows = (size_t)(mode == 2 ? 0 : Cout) + (size_t)r.gram_tiles * 64 * r.wo;
  return ((size_t)r.nsplit * rows * Cin + gram_colsum_floats(r)) * sizeof(float);
}

// sum the split slabs [nsplit][rows][Cin] -> s_out [rows][Cin] (blocks < nmain) and the colsum partials [np][wpad] -> colsum_out [Cin]
// (the blocks behind them, 16 columns each).  The output is small (8 - 24 K float4) and the slab count large: 16 columns x 16 slab / row
// lanes per block with two / four independent loads in flight, the lanes meet in LDS (one thread per column walking every slab ran 32
// blocks for 95 us; the column sums as the tail of ONE block, 1024 partial rows over four lanes, 39 us).  Deterministic (fixed order).
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float4* __restrict__ slab, float4* __restrict__ s_out, int nsplit, int total4, int nmain,
                                                          const float* __restrict__ cpart, float* __restrict__ colsum_out, int np, int wpad, int Cin) {
  __shared__ float4 red[16][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  if ((int)blockIdx.x >= nmain) {   // column sums
    float* redf = reinterpret_cast<float*>(red);
    const int col = ((int)blockIdx.x - nmain) * 16 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < Cin) {
      int k = ry;
      for (; k + 48 < np; k += 64) {
        s0 += cpart[(size_t)k * wpad + col]; s1 += cpart[(size_t)(k + 16) * wpad + col];
        s2 += cpart[(size_t)(k + 32) * wpad + col]; s3 += cpart[(size_t)(k + 48) * wpad + col];
      }
      for (; k < np; k += 16) s0 += cpart[(size_t)k * wpad + col];
    }
    redf[ry * 16 + cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && col < Cin) {
      float t = redf[cx];
#pragma unroll
      for (int j = 1; j < 16; ++j) t += redf[j * 16 + cx];
      colsum_out[col] = t;
    }
    return;
  }
  const int i = blockIdx.x * 16 + cx;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (i < total4) {
    int k = ry;
    for (; k + 16 < nsplit; k += 32) {
      const float4 u = slab[(size_t)k * total4 + i], v = slab[(size_t)(k + 16) * total4 + i];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    if (k < nsplit) { const float4 u = slab[(size_t)k * total4 + i]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  red[ry][cx] = a;
  __syncthreads();
  if (ry == 0 && i < total4) {
    float4 t = red[0][cx];
#pragma unroll
    for (int j = 1; j < 16; ++j) { const float4 u = red[j][cx]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    s_out[i] = t;
  }
}

int launch_wgrad_gram(int N, int H, int W, int Cin, int Cout, const bf16_t* g, const bf16_t* in, float* slab, float* s_out, float* colsum_out,
                      hipStream_t st, int mode) {
  WgradRingPlan r;
  const int M = N * H * W;
  ARG_CHECK(wgrad_gram_plan(M, Cout, Cin, r, mode) && (Cin == 64 || Cin == 128 || Cin == 256), "wgrad_gram: shape Cout=%d Cin=%d not tiled by the ring kernel", Cout, Cin);
  if (mode == 2) Cout = 0;
  WgradArgs a = {};
  a.dy = g; a.in = in; a.slab = slab;
  a.N = N; a.IH = H; a.IW = W; a.C = Cin; a.Cpitch = Cin; a.OH = H; a.OW = W;
  a.Cout = Cout; a.Ktot = Cin; a.Sy = 1; a.Sx = 1; a.ntaps = 1; a.M = M; a.simple1x1 = 1;
  const size_t rows = (size_t)Cout + (size_t)r.gram_tiles * 64 * r.wo;
  a.gram_cols = Cin;
  a.colsum = slab + (size_t)r.nsplit * rows * Cin;
  int rc = wgrad_ring_launch(a, r, st);
  if (rc) return rc;
  const int total4 = (int)(rows * Cin / 4);
  const int G = 8 / (r.wo * r.wk);
  const int nmain = ceil_div(total4, 16), ncs = (r.gram_tiles && colsum_out) ? ceil_div(Cin, 16) : 0;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(nmain + ncs), dim3(256), 0, st, reinterpret_cast<const float4*>(slab),
                     reinterpret_cast<float4*>(s_out), r.nsplit, total4, nmain, a.colsum, colsum_out, r.nsplit * G, r.gram_tiles * 64 * r.wo, Cin);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

bool wgrad_ring_plan(const WgradArgs& a, WgradRingPlan& r) {
  if (a.C % 8 || a.Cpitch % 8) return false;
  // 32-bit byte offsets, out-of-range marker above every tensor
  if ((uint64_t)a.M * a.Cout * 2 >= 0xE0000000ull || (uint64_t)a.N * a.IH * a.IW * a.Cpitch * 2 >= 0xE0000000ull) return false;
  const bool s1 = a.simple1x1 && a.Cpitch == a.C;
  if (!wgrad_ring_tile(a.M, a.Cout, a.Ktot, r, s1)) return false;
  const int step = (8 / (r.wo * r.wk)) * 32;
  r.s1 = s1;
  if (!r.s1 && ((a.OW + step) * a.OW >= 65536 || (a.OH + step) * a.OH >= 65536)) return false;   // reciprocal pixel stepping
  return true;
}

template <int WO, int WK, int NIT, bool STAG = false>
static int ring_launch_t(const WgradArgs& a, bool s1, hipStream_t st) {
  constexpr int G = 8 / (WO * WK), LDS = NIT * G * 32 * (64 * WO + 64 * WK) * 2;
  static_assert(LDS <= 160 * 1024, "ring exceeds the LDS");
  const int grid = a.nblk_o * a.nblk_k * a.nsplit;
  static bool attr_done[2] = {false, false};
  if (s1) {
    if (!attr_done[1]) {
      HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_ring_kernel<WO, WK, NIT, true, STAG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      attr_done[1] = true;
    }
    hipLaunchKernelGGL((wgrad_ring_kernel<WO, WK, NIT, true, STAG>), dim3(grid), dim3(512), LDS, st, a);
  } else {
    if (!attr_done[0]) {
      HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_ring_kernel<WO, WK, NIT, false, STAG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
      attr_done[0] = true;
    }
    hipLaunchKernelGGL((wgrad_ring_kernel<WO, WK, NIT, false, STAG>), dim3(grid), dim3(512), LDS, st, a);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

static int g_ring_launches = 0;
int wgrad_ring_launch_count() { return g_ring_launches; }
extern "C" int64_t mmskin_wgrad_ring_launches(void) { return g_ring_launches; }

int wgrad_ring_launch(WgradArgs& a, const WgradRingPlan& r, hipStream_t st) {
  a.nsplit = r.nsplit; a.m_per_split = r.mps;
  a.nblk_o_main = a.Cout / (64 * r.wo);
  a.nblk_o = a.nblk_o_main + r.gram_tiles; a.nblk_k = a.Ktot / (64 * r.wk);
  if (!r.gram_tiles) { a.gram_cols = 0; a.colsum = nullptr; }
  ++g_ring_launches;
  // ring depth: G = 1 tiles move 24 KB per iteration (4 deep = 96 KB), G = 2 tiles 32 - 40 KB (3 deep = 96 - 120 KB)
  static const int deep = ring_env("MMSKIN_WGRAD_RING_DEEP", 1);
  // MMSKIN_WGRAD_RING_STAG (default 1): the MFMA-bound tiles (layers 2 - 4) with the two wave halves half an iteration apart
  static const int stag = ring_env("MMSKIN_WGRAD_RING_STAG", 1);
  if (stag && r.wo == 4 && r.wk == 2) return ring_launch_t<4, 2, 4, true>(a, r.s1, st);
  if (stag && r.wo == 2 && r.wk == 4) return ring_launch_t<2, 4, 4, true>(a, r.s1, st);
  if (stag && r.wo == 2 && r.wk == 2) return ring_launch_t<2, 2, 4, true>(a, r.s1, st);
  if (r.wo == 4 && r.wk == 2) return deep ? ring_launch_t<4, 2, 4>(a, r.s1, st) : ring_launch_t<4, 2, 3>(a, r.s1, st);
  if (r.wo == 2 && r.wk == 4) return deep ? ring_launch_t<2, 4, 4>(a, r.s1, st) : ring_launch_t<2, 4, 3>(a, r.s1, st);
  if (r.wo == 2 && r.wk == 2) return deep ? ring_launch_t<2, 2, 3>(a, r.s1, st) : ring_launch_t<2, 2, 2>(a, r.s1, st);
  if (r.wo == 4 && r.wk == 1) return deep ? ring_launch_t<4, 1, 3>(a, r.s1, st) : ring_launch_t<4, 1, 2>(a, r.s1, st);
  if (r.wo == 1 && r.wk == 4) return deep ? ring_launch_t<1, 4, 3>(a, r.s1, st) : ring_launch_t<1, 4, 2>(a, r.s1, st);
  if (r.wo == 2 && r.wk == 1) return ring_launch_t<2, 1, 2>(a, r.s1, st);     // 48 KB per iteration (four groups): two deep
  return ring_launch_t<1, 2, 2>(a, r.s1, st);
}

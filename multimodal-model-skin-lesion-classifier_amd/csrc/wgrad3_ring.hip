// All-taps 3x3 / stride 1 / pad 1 weight gradient, ring form (round 4; bf16 operands, fp32 accumulate).
//
// Replaces ATen/cuDNN conv_backward(weight) of the bottleneck conv2 layers (loadImageModelClassifier.py:65-75 through
// loss.backward(), train_pad_20.py:112).  Same decomposition as wgrad3x3_kernel (wgrad.hip): a workgroup owns
// dW[cout tile][9 taps][64 cin]; a stage is R whole image rows (R W <= 64 pixel slots); their dY rows and the zero-bordered
// (R + 2) x (W + 2) input window are staged ONCE and tap (ty, tx) of pixel (r, x) is window row (r + ty) W2 + (x + tx).  What
// changes is everything around the MFMAs:
//   * operands by `buffer_load_dwordx4 ... offen lds` into a ring of NIT iterations behind a counted vmcnt and a raw s_barrier
//     (the register-staged kernel had one stage in flight per workgroup behind __syncthreads(): 2.4x its own MFMA time);
//     border columns / rows and slots past the stage get an out-of-range offset -> the buffer unit writes the zeros;
//   * LDS-DMA writes lane-linear, so the 144-byte pitch is gone: rows are 128 B, the window pitch W2 is W + 2 rounded up to a
//     multiple of 16 rows, and the bank swizzle (chunk ^ f(row bits 1 and 3)) is applied to the SOURCE chunk.  ty W2 never touches
//     those bits, so a tap is still an immediate offset for ty and one of three precomputed addresses for tx;
//   * 8 waves: either one group on a 128-cout tile (the window is shared by twice the MFMAs) or, for 64-cout layers, two groups
//     that take alternate stages of the same 64-cout tile and meet in LDS before the slab write.
#include <stdlib.h>

#include "conv.h"

typedef uint32_t srd_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Wgrad3RingArgs {
  const bf16_t* dy; const bf16_t* in; float* slab;
  int N, H, W, C, Cout;
  int R, spi;                        // image rows per stage, stages per image
  int W2p, yrows;                    // window pitch (rows), window rows per stage = (R + 2) W2p (multiple of 8)
  int total_stages, stages_per_split, nsplit;
  int nblk_o, nblk_c;
};

__device__ __forceinline__ int w3_swz(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }   // 128-byte rows
// dY rows: 128-byte pitch as above; 256-byte pitch (128-cout tile): every row starts on bank 0, so all three low row bits take part
template <int PITCH> __device__ __forceinline__ int w3_swzx(int row) {
  if constexpr (PITCH == 256) return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
  else return w3_swz(row);
}
__device__ __forceinline__ uint2 w3_tr16(const unsigned char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}
#define W3_OOB 0xF0000000u
#define W3_DMA(voff, srd, dst)                                                                                     \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"((uint32_t)(dst)),   \
               "v"((uint32_t)(voff)), "s"(srd) : "memory")
template <int N> __device__ __forceinline__ void w3_wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int W3R_YMAX = 192;        // window rows per stage the LDS layout reserves (3 x 64: one 56-wide row with its halo)

// WOB = 64-cout blocks per tile (1 or 2), G = pixel groups (WOB G == 2), NYW = window pieces per wave and iteration, NIT = ring depth
template <int WOB, int G, int NYW, int NIT>
__global__ __launch_bounds__(512) void wgrad3_ring_kernel(const Wgrad3RingArgs p) {
  static_assert(WOB * G == 2, "8 waves = G groups x WOB cout blocks x 4 waves");
  constexpr int PX = 128 * WOB;                        // dY row pitch in LDS
  constexpr int XB = 64 * PX;                          // 64 pixel slots
  constexpr int STG = XB + W3R_YMAX * 128, ITB = G * STG;
  constexpr int CPRX = PX / 16;
  constexpr int PCX = XB / 1024;                       // dY pieces per stage
  constexpr int NXW = G * PCX / 8;                     // per wave and iteration
  constexpr int NPW = NXW + NYW;
  static_assert((NIT - 2) * NPW < 64, "ring depth");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3r[];   // NIT * ITB + 1 KB scratch for idle pieces

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int cb = tile % p.nblk_c; tile /= p.nblk_c;
  const int ob = tile % p.nblk_o;
  const int split = tile / p.nblk_o;
  const int o0 = ob * 64 * WOB, c0 = cb * 64;
  const int W = p.W, H = p.H, W2p = p.W2p, R = p.R, spi = p.spi;
  const int st_begin = split * p.stages_per_split;
  const int st_end = min(p.total_stages, st_begin + p.stages_per_split);
  const int nit = (st_end - st_begin + G - 1) / G;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem3r;
  const uint32_t scratch = lds0 + NIT * ITB;
  const int ypieces = p.yrows / 8;

  srd_t srdX, srdY;
  {
    const uint64_t bx = (uint64_t)(uintptr_t)p.dy, by = (uint64_t)(uintptr_t)p.in;
    srdX = srd_t{(uint32_t)bx, (uint32_t)(bx >> 32) & 0xffffu, (uint32_t)((uint64_t)p.N * H * W * p.Cout * 2), 0x00020000u};
    srdY = srd_t{(uint32_t)by, (uint32_t)(by >> 32) & 0xffffu, (uint32_t)((uint64_t)p.N * H * W * p.C * 2), 0x00020000u};
  }

  // ---- dY pieces: piece jx = wid + 8 i -> group jx / PCX, block jx % PCX; lane -> (slot, chunk)
  int x_g[NXW], x_r[NXW];
  uint32_t x_rel[NXW], x_dst[NXW];
  bool x_ok[NXW];
#pragma unroll
  for (int i = 0; i < NXW; ++i) {
    const int jx = wid + 8 * i, gg = jx / PCX, blk = jx - gg * PCX;
    const int L = blk * 64 + lane, slot = L / CPRX, c = L - slot * CPRX;
    const int sc = c ^ w3_swzx<PX>(slot);
    const int r = slot / W;
    x_g[i] = gg; x_r[i] = r;
    x_ok[i] = slot < R * W;
    x_rel[i] = (uint32_t)(slot * p.Cout + o0 + sc * 8) * 2u;        // relative to the stage's first pixel
    x_dst[i] = (uint32_t)(gg * STG + blk * 1024);
  }
  // ---- window pieces: piece jy = wid + 8 i -> group jy / ypieces, block jy % ypieces; lane -> (window row, chunk); idle past G * ypieces
  int y_g[NYW], y_wr[NYW];
  uint32_t y_dst[NYW];
  int y_rel[NYW];
  bool y_ok[NYW];
#pragma unroll
  for (int i = 0; i < NYW; ++i) {
    const int jy = wid + 8 * i, gg = jy / ypieces, blk = jy - gg * ypieces;
    const bool live = gg < G;
    const int j = blk * 8 + (lane >> 3), c = lane & 7;
    const int sc = c ^ w3_swz(j);
    const int wr = j / W2p, wc = j - wr * W2p;
    y_g[i] = live ? gg : 0; y_wr[i] = wr - 1;
    y_ok[i] = live && wc >= 1 && wc <= W;
    y_rel[i] = (((wr - 1) * W + (wc - 1)) * p.C + c0 + sc * 8) * 2;  // signed: the halo row above the stage is negative
    y_dst[i] = live ? (uint32_t)(gg * STG + XB + blk * 1024) : 0xffffffffu;
  }
  // per-group stage cursor: stage index, image, first image row
  int s_cur[G], s_img[G], s_six[G];
#pragma unroll
  for (int g2 = 0; g2 < G; ++g2) {
    s_cur[g2] = st_begin + g2;
    s_img[g2] = s_cur[g2] / spi;
    s_six[g2] = s_cur[g2] - s_img[g2] * spi;
  }
#define W3R_ISSUE(bufoff)                                                                                           \
  do {                                                                                                              \
    const uint32_t b_ = lds0 + (bufoff);                                                                            \
    int base_[G], y0_[G]; bool live_[G];                                                                            \
    _Pragma("unroll") for (int g2 = 0; g2 < G; ++g2) {                                                              \
      y0_[g2] = s_six[g2] * R;                                                                                      \
      base_[g2] = (s_img[g2] * H + y0_[g2]) * W;                                                                    \
      live_[g2] = s_cur[g2] < st_end;                                                                               \
    }                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NXW; ++i) {                                                               \
      const int g2 = G == 1 ? 0 : x_g[i];                                                                           \
      const int y0 = G == 1 ? y0_[0] : (g2 ? y0_[G - 1] : y0_[0]);                                                  \
      const int bs = G == 1 ? base_[0] : (g2 ? base_[G - 1] : base_[0]);                                            \
      const bool lv = G == 1 ? live_[0] : (g2 ? live_[G - 1] : live_[0]);                                           \
      const bool ok = lv && x_ok[i] && (y0 + x_r[i] < H);                                                           \
      const uint32_t v_ = ok ? x_rel[i] + (uint32_t)bs * (uint32_t)p.Cout * 2u : W3_OOB;                            \
      W3_DMA(v_, srdX, b_ + x_dst[i]);                                                                              \
    }                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NYW; ++i) {                                                               \
      const int g2 = G == 1 ? 0 : y_g[i];                                                                           \
      const int y0 = G == 1 ? y0_[0] : (g2 ? y0_[G - 1] : y0_[0]);                                                  \
      const int bs = G == 1 ? base_[0] : (g2 ? base_[G - 1] : base_[0]);                                            \
      const bool lv = G == 1 ? live_[0] : (g2 ? live_[G - 1] : live_[0]);                                           \
      const bool ok = lv && y_ok[i] && (unsigned)(y0 + y_wr[i]) < (unsigned)H;                                      \
      const uint32_t v_ = ok ? (uint32_t)(y_rel[i] + bs * p.C * 2) : W3_OOB;                                        \
      W3_DMA(v_, srdY, y_dst[i] == 0xffffffffu ? scratch : b_ + y_dst[i]);                                          \
    }                                                                                                               \
    _Pragma("unroll") for (int g2 = 0; g2 < G; ++g2) {                                                              \
      s_cur[g2] += G; s_six[g2] += G;                                                                               \
      while (s_six[g2] >= spi) { s_six[g2] -= spi; ++s_img[g2]; }                                                   \
    }                                                                                                               \
  } while (0)

  // ---- fragment addressing (transposing reads: lane (q, pp) addresses row q, columns 4pp..4pp+3 of a 4 x 16 block)
  const int gg = wid / (4 * WOB), wpos = wid - gg * 4 * WOB;
  const int wob = wpos >> 2, wo = (wpos >> 1) & 1, wk = wpos & 1;
  const int l15 = lane & 15, g = lane >> 4, q = l15 >> 2, pp = l15 & 3;
  uint32_t xa[2][2][2];       // [ks][h][j]
  uint32_t ya[2][2][3][2];    // [ks][h][tx][i]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 32 * ks + 8 * g + q + 4 * h;       // pixel slot
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cx = (wob * 64 + wo * 32 + j * 16 + 4 * pp) * 2;
        xa[ks][h][j] = k * PX + (((cx >> 4) ^ w3_swzx<PX>(k)) << 4) + (cx & 15);
      }
      const int r = k / W, x = k - r * W;
      const int brow = k < R * W ? r * W2p + x : 0;    // slots past the stage have dY = 0: any window row will do
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = brow + tx;
          const int cy = (wk * 32 + i * 16 + 4 * pp) * 2;
          ya[ks][h][tx][i] = XB + row * 128 + (((cy >> 4) ^ w3_swz(row)) << 4) + (cy & 15);
        }
    }
  f32x4_t acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int ty_step = W2p * 128;   // bytes per window image row

#pragma unroll
  for (int s = 0; s < NIT - 1; ++s) W3R_ISSUE(s * ITB);
  int buf_rd = 0, buf_wr = (NIT - 1) * ITB;
  for (int it = 0; it < nit; ++it) {
    w3_wait_vm<(NIT - 2) * NPW>();
    asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    W3R_ISSUE(buf_wr);
    const unsigned char* sb = smem3r + buf_rd + gg * STG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 fx[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint2 lo = w3_tr16(sb + xa[ks][0][j]), hi = w3_tr16(sb + xa[ks][1][j]);
        fx[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int tyo = (t / 3) * ty_step;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint2 lo = w3_tr16(sb + ya[ks][0][t % 3][i] + tyo), hi = w3_tr16(sb + ya[ks][1][t % 3][i] + tyo);
          const uint4 fy = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fy), __builtin_bit_cast(bf16x8_t, fx[j]),
                                                                   acc[t][i][j], 0, 0, 0);
        }
      }
    }
    buf_rd = buf_rd + ITB == NIT * ITB ? 0 : buf_rd + ITB;
    buf_wr = buf_wr + ITB == NIT * ITB ? 0 : buf_wr + ITB;
  }
#undef W3R_ISSUE
  w3_wait_vm<0>();
  __syncthreads();

  // ---- two groups on one tile: group 1 parks its sums in LDS, group 0 adds (the 147 KB tile goes through in two tap halves)
  if constexpr (G == 2) {
    float4* red = reinterpret_cast<float4*>(smem3r);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int t0 = half ? 5 : 0, t1 = half ? 9 : 5;
      if (gg == 1) {
#pragma unroll
        for (int t = t0; t < t1; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              red[((wpos * 5 + (t - t0)) * 4 + i * 2 + j) * 64 + lane] = make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
      }
      __syncthreads();
      if (gg == 0) {
#pragma unroll
        for (int t = t0; t < t1; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const float4 u = red[((wpos * 5 + (t - t0)) * 4 + i * 2 + j) * 64 + lane];
              acc[t][i][j][0] += u.x; acc[t][i][j][1] += u.y; acc[t][i][j][2] += u.z; acc[t][i][j][3] += u.w;
            }
      }
      __syncthreads();
    }
    if (gg == 1) return;
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
        const int cout = o0 + wob * 64 + wo * 32 + j * 16 + l15;
        const int kidx = t * p.C + c0 + wk * 32 + i * 16 + g * 4;
        *reinterpret_cast<float4*>(slab + (size_t)cout * Ktot + kidx) = make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
      }
}

// ------------------------------------------------------------------------------------------ host
static int w3r_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }

struct Wgrad3RingPlan { Wgrad3RingArgs a; int wob, g, nyw; };

static bool wgrad3_ring_plan_(const ConvShape& s, Wgrad3RingPlan& pl) {
  static const int enabled = w3r_env("MMSKIN_WGRAD3_RING", 1);
  if (!enabled || s.kh != 3 || s.kw != 3 || s.stride != 1 || s.pad != 1) return false;
  if (s.Cin % 64 || s.Cout % 64 || s.W > 62 || s.W < 1) return false;
  if ((uint64_t)s.N * s.H * s.W * s.Cout * 2 >= 0xE0000000ull || (uint64_t)s.N * s.H * s.W * s.Cin * 2 >= 0xE0000000ull) return false;
  Wgrad3RingArgs& a = pl.a;
  int R = 64 / s.W;
  if (R > s.H) R = s.H;
  const int W2p = (s.W + 2 + 15) / 16 * 16;
  if ((R + 2) * W2p > W3R_YMAX) return false;
  a.N = s.N; a.H = s.H; a.W = s.W; a.C = s.Cin; a.Cout = s.Cout;
  a.R = R; a.spi = ceil_div(s.H, R);
  a.W2p = W2p; a.yrows = (R + 2) * W2p;
  a.total_stages = s.N * a.spi;
  // MMSKIN_WGRAD3_WOB=1: two pixel groups on a 64-cout tile for every layer (half the slab bytes per workgroup, the window staged per group)
  static const int wob_env = w3r_env("MMSKIN_WGRAD3_WOB", 0);
  pl.wob = wob_env ? wob_env : ((s.Cout % 128 == 0) ? 2 : 1);
  if (s.Cout % 128) pl.wob = 1;
  pl.g = 2 / pl.wob;
  pl.nyw = ceil_div(pl.g * (a.yrows / 8), 8);
  a.nblk_o = s.Cout / (64 * pl.wob); a.nblk_c = s.Cin / 64;
  const int tiles = a.nblk_o * a.nblk_c;
  static const int target = w3r_env("MMSKIN_WGRAD3_RING_BLOCKS", 256);
  int ns = target / tiles > 0 ? target / tiles : 1;
  // a split writes a whole tile x 9 taps of fp32 (147 - 295 KB) and the reduction reads it back: at least ~16 stages of work per split
  static const int min_stages = w3r_env("MMSKIN_WGRAD3_MIN_STAGES", 16);
  if (min_stages > 1 && ns > a.total_stages / min_stages) ns = a.total_stages / min_stages > 0 ? a.total_stages / min_stages : 1;
  if (ns > a.total_stages) ns = a.total_stages;
  a.stages_per_split = ceil_div(ceil_div(a.total_stages, ns), pl.g) * pl.g;   // whole iterations
  a.nsplit = ceil_div(a.total_stages, a.stages_per_split);
  return true;
}

bool wgrad3_ring_takes(const ConvShape& s, int* nsplit) {
  Wgrad3RingPlan pl;
  if (!wgrad3_ring_plan_(s, pl)) return false;
  if (nsplit) *nsplit = pl.a.nsplit;
  return true;
}

static int64_t g_w3r_launches = 0;
extern "C" int64_t mmskin_wgrad3_ring_launches(void) { return g_w3r_launches; }

template <int WOB, int G, int NYW, int NIT>
static int w3r_launch_t(const Wgrad3RingArgs& a, hipStream_t st) {
  constexpr int LDS = NIT * G * (64 * 128 * WOB + W3R_YMAX * 128) + 1024;
  static_assert(LDS <= 160 * 1024, "ring exceeds the LDS");
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3_ring_kernel<WOB, G, NYW, NIT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad3_ring_kernel<WOB, G, NYW, NIT>), dim3(a.nblk_o * a.nblk_c * a.nsplit), dim3(512), LDS, st, a);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// launches the GEMM; the caller reduces the a.nsplit slabs [Cout][9 Cin]
int launch_wgrad3_ring(const ConvShape& s, const bf16_t* dout, const bf16_t* in, float* slab, hipStream_t st, int* nsplit) {
  Wgrad3RingPlan pl;
  ARG_CHECK(wgrad3_ring_plan_(s, pl), "wgrad3_ring: shape not taken");
  pl.a.dy = dout; pl.a.in = in; pl.a.slab = slab;
  *nsplit = pl.a.nsplit;
  ++g_w3r_launches;
  // stage = 40 KB (128-cout tile) / 32 KB (64-cout tile, two per iteration): three resp. two iterations in the ring
  if (pl.wob == 2) {
    switch (pl.nyw) {
      case 1: case 2: return w3r_launch_t<2, 1, 2, 3>(pl.a, st);
      default: return w3r_launch_t<2, 1, 3, 3>(pl.a, st);
    }
  }
  switch (pl.nyw) {
    case 1: case 2: case 3: return w3r_launch_t<1, 2, 3, 2>(pl.a, st);
    case 4: return w3r_launch_t<1, 2, 4, 2>(pl.a, st);
    case 5: return w3r_launch_t<1, 2, 5, 2>(pl.a, st);
    default: return w3r_launch_t<1, 2, 6, 2>(pl.a, st);
  }
}

// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (ResNet-50 layer1.*.conv2, forward and data gradient), bf16.
//
// Replaces the tapped gather-GEMM of conv_gemm.hip for this one shape (reference call site: torchvision Bottleneck.conv2 reached through
// loadImageModelClassifier.py:71-75).  There a 128 x 64 tile re-gathers its 128 input pixels nine times through L2 (24 KB of LDS fill
// per 1.05 MFLOP) and reads six LDS fragments per eight MFMAs: 104 - 107 us per launch against a 33 us HBM floor and a 24 us MFMA
// floor (profiles/r03_experiments.txt).  Here
//   * ONE 256-thread workgroup per image walks its rows top to bottom in tiles of R = 4 rows (32 * FH pixels = 2 * FH MFMA fragments);
//   * all nine taps' weights (64 x 9 x 64 bf16 = 72 KB) stay in LDS for the whole image;
//   * the input arrives ONCE: a ring of 10 zero-bordered image rows (pitch 144 B per pixel: 16 consecutive pixels read 16-byte chunks
//     from all 64 banks, and tap (oy, ox) of a pixel is a constant byte offset from its row address -- no per-tap address arithmetic,
//     no masks); the four rows of the next tile are fetched to registers while this tile computes and written after its last tap;
//   * wave (a, b) owns pixel fragments FH*a .. FH*a + FH-1 and output channels 32 b .. 32 b + 31: 2 + FH fragment reads per FH * 2
//     MFMAs and k half (0.64 reads per MFMA), fragments double-buffered across taps and interleaved with the MFMAs;
//   * the weight rows are permuted while staged so that a lane ends up with EIGHT consecutive output channels of a pixel: 16-byte
//     stores straight from the accumulators, BatchNorm partial sums (forward) or the consumer's BatchNorm-backward mask + sums (dgrad,
//     profile 3 of conv_gemm.hip) accumulated in registers over the whole image -> one statistics row per image.
#include <stdlib.h>

#include "conv.h"

namespace {

constexpr int PITCH = 144;   // bytes per window pixel
constexpr int RING = 10;     // window rows: 6 of the current tile + 4 of the next
constexpr int WBYTES = 9 * 64 * 128;

struct C3Args {
  const bf16_t* in;      // [N][H][W][64]
  const bf16_t* w;       // [64][9][64]: forward weights [cout][tap][cin], or the dgrad staging [cin][tap][cout]
  bf16_t* out;           // [N][H][W][64]
  float* stat_sum;       // optional: [N][stat_stride] partial sums (one row per image)
  float* stat_sq;
  int stat_stride;
  const bf16_t* ep_x;    // EPI 3: raw BatchNorm input of the consumer unit (same shape as out)
  const float* ep_scale;
  const float* ep_shift;
  int N, H, W;
  int ablate;            // -DMMSKIN_ABLATE builds only: 1 skip the row prefetch, 4 skip MFMAs, 8 skip the output stores, 32 skip fragment reads
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// FLIP = 0: forward taps (oy, ox) = (r - 1, q - 1); FLIP = 1: data gradient (1 - r, 1 - q); the weight tap is r * 3 + q either way.
template <int FH, int EPI, int FLIP>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_kernel(const C3Args p) {
  constexpr int W = 8 * FH, WP = W + 2, ROWB = WP * PITCH, NPIX = 32 * FH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Wl = smem;             // weights: [tap][64 permuted cout rows][128 B], XOR-swizzled chunks
  unsigned char* Wn = smem + WBYTES;    // window ring: [RING][WP][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
#ifdef MMSKIN_ABLATE
  const int abl = p.ablate;
#else
  constexpr int abl = 0;
#endif
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int a = wid >> 1, b = wid & 1;
  const int n = blockIdx.x, H = p.H;
  const unsigned char* in_img = reinterpret_cast<const unsigned char*>(p.in) + (size_t)n * H * W * 128;

  // ---- weights -> LDS.  LDS row (half b', fragment i, row rho) holds output channel 32 b' + 8 (rho >> 2) + 4 i + (rho & 3): lane (l15, g) of
  // the MFMA result then owns channels 8 g .. 8 g + 7 of its half (fragment 0: +0..3, fragment 1: +4..7).
  {   // 18 chunks per thread, all loads in flight before the first LDS store (a rolled load -> store loop waited for every load: 18 us)
    uint4 wv[18];
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      const int c = tid + 256 * q, row = c >> 3, ch = c & 7, tap = row >> 6, lrow = row & 63;
      const int bh = lrow >> 5, i = (lrow >> 4) & 1, rho = lrow & 15;
      const int cout = bh * 32 + 8 * (rho >> 2) + 4 * i + (rho & 3);
      wv[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.w) + ((size_t)(cout * 9 + tap) * 64) * 2 + ch * 16);
    }
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      const int c = tid + 256 * q, row = c >> 3, ch = c & 7, lrow = row & 63;
      *reinterpret_cast<uint4*>(Wl + row * 128 + ((ch ^ ((lrow >> 1) & 7)) << 4)) = wv[q];
    }
  }
  // ---- window ring: zero everything once (borders stay zero: loads only ever write pixels 1 .. W of a row), then rows 0 .. 4
  for (int c = tid; c < RING * ROWB / 16; c += 256) *reinterpret_cast<uint4*>(Wn + c * 16) = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  {   // rows 0 .. 4 (H >= 8): slot(row) = (row + 1) mod RING
    constexpr int NQ0 = (5 * W * 8 + 255) / 256;
    uint4 rv[NQ0];
#pragma unroll
    for (int q = 0; q < NQ0; ++q) {
      const int c = tid + 256 * q;
      rv[q] = c < 5 * W * 8 ? *reinterpret_cast<const uint4*>(in_img + (size_t)c * 16) : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int q = 0; q < NQ0; ++q) {
      const int c = tid + 256 * q, pp = c >> 3, ch = c & 7, r = pp / W, x = pp - r * W;
      if (c < 5 * W * 8) *reinterpret_cast<uint4*>(Wn + ((r + 1) * WP + x + 1) * PITCH + ch * 16) = rv[q];
    }
  }
  // ---- per-lane constants
  int yl[FH], xo[FH];      // row within the tile and byte offset of the pixel one to the LEFT (tap ox = -1) of this lane's pixel, per fragment
#pragma unroll
  for (int f = 0; f < FH; ++f) {
    const int pp = 16 * (FH * a + f) + l15;
    yl[f] = pp / W;
    xo[f] = (pp - yl[f] * W) * PITCH + g * 16;
  }
  const int sw = l15 >> 1;
  const unsigned char* wb[2] = {Wl + (b * 32 + l15) * 128 + ((g ^ sw) << 4), Wl + (b * 32 + l15) * 128 + (((4 + g) ^ sw) << 4)};
  // steady-state loader: chunk c = tid + 256 i of the next tile's four rows (contiguous in memory); its LDS position without the slot
  int ld_r[FH], ld_off[FH];
#pragma unroll
  for (int i = 0; i < FH; ++i) {
    const int c = tid + 256 * i, pp = c >> 3;
    ld_r[i] = pp / W;
    ld_off[i] = (pp - ld_r[i] * W + 1) * PITCH + (c & 7) * 16;
  }
  float ssum[8], ssq[8], esc[8], esh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; esc[e] = 0.f; esh[e] = 0.f; }
  const int ch0 = b * 32 + 8 * g;   // first of this lane's eight output channels
  if constexpr (EPI == 3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { esc[e] = p.ep_scale[ch0 + e]; esh[e] = p.ep_shift[ch0 + e]; }
  }
  __syncthreads();

  const int ntile = H / 4;
  for (int k = 0; k < ntile; ++k) {
    const int ybase = 4 * k;
    // ---- the next tile's rows: ybase + 5 .. ybase + 8 -> registers now, LDS after the last tap
    uint4 nx[FH];
    const bool have_next = k + 1 < ntile && !(abl & 1);
    if (have_next) {
      // only the LAST prefetch reaches past the image (its fourth row is row H): that row is fetched from the row above (in bounds) and
      // zeroed by a select -- no per-chunk branch around the loads
      const bool past = ybase + 8 >= H;
      const unsigned char* src = in_img + (size_t)(ybase + 5) * W * 128;
#pragma unroll
      for (int i = 0; i < FH; ++i) {
        const bool zero = past && ld_r[i] == 3;
        const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src + (size_t)(tid + 256 * i) * 16 - (zero ? W * 128 : 0)));
        nx[i] = zero ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(t[0], t[1], t[2], t[3]);
      }
    }
    // ---- row addresses of this tile: rows ybase - 1 + j, j = yl + oy + 1, live in slot (ybase + j) mod RING
    const int kk = ybase % RING;
    const unsigned char* ra[FH][3];
#pragma unroll
    for (int f = 0; f < FH; ++f)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        int s = kk + yl[f] + j;
        s = s >= RING ? s - RING : s;
        ra[f][j] = Wn + s * ROWB + xo[f];
      }
    f32x4_t acc[FH][2];
#pragma unroll
    for (int f = 0; f < FH; ++f) { acc[f][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[f][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    uint4 fa[2][FH][2], fb[2][2][2];
#ifdef MMSKIN_ABLATE
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int f = 0; f < FH; ++f) fa[q][f][ks] = make_uint4(0x3c003c00u + tid, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
        fb[q][0][ks] = make_uint4(0x3c003c00u, 0x3c003c00u + tid, 0x3c003c00u, 0x3c003c00u); fb[q][1][ks] = fb[q][0][ks];
      }
#endif
#define TAP_OY(t) (FLIP ? 1 - (t) / 3 : (t) / 3 - 1)
#define TAP_OX(t) (FLIP ? 1 - (t) % 3 : (t) % 3 - 1)
#define READ_TAP(buf, t)                                                                                                   \
  if (!(abl & 32)) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) fb[buf][i][ks] = *reinterpret_cast<const uint4*>(wb[ks] + (t) * 8192 + i * 2048); \
    _Pragma("unroll") for (int f = 0; f < FH; ++f)                                                                         \
      fa[buf][f][ks] = *reinterpret_cast<const uint4*>(ra[f][TAP_OY(t) + 1] + (TAP_OX(t) + 1) * PITCH + ks * 64);         \
  }
    READ_TAP(0, 0)
    // EPI 3: the consumer unit's raw BatchNorm input of this tile, fetched behind the LAST tap's fragment reads (the other fragment buffer is
    // dead by then: no extra registers) instead of in the epilogue, where every fragment waited a full memory round trip for its 16 bytes
    uint4 xpre[EPI == 3 ? FH : 1];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cur = t & 1;
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < 9) { READ_TAP(cur ^ 1, t + 1) }
      if constexpr (EPI == 3) {
        if (t == 8) {
#pragma unroll
          for (int f = 0; f < FH; ++f) {
            const size_t offx = (((size_t)n * H + ybase) * W + 16 * (FH * a + f) + l15) * 64 + ch0;
            xpre[f] = *reinterpret_cast<const uint4*>(p.ep_x + offx);
          }
        }
      }
      if (!(abl & 4))
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < FH; ++f)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[f][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[cur][i][ks]),
                                                                __builtin_bit_cast(bf16x8_t, fa[cur][f][ks]), acc[f][i], 0, 0, 0);
      if (t + 1 < 9) {   // one fragment read of the next tap behind each of the first MFMAs of this one
#pragma unroll
        for (int q = 0; q < 4 * FH; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (q < 2 * (FH + 2)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#undef READ_TAP
#undef TAP_OY
#undef TAP_OX
    // ---- the next tile's rows -> their ring slots (none of them is read by this tile)
    if (have_next) {
      const int k5 = (ybase + 6) % RING;   // slot of row ybase + 5
#pragma unroll
      for (int i = 0; i < FH; ++i) {
        int s = k5 + ld_r[i];
        s = s >= RING ? s - RING : s;
        *reinterpret_cast<uint4*>(Wn + s * ROWB + ld_off[i]) = nx[i];
      }
    }
    // ---- epilogue of this tile: 16 bytes (eight channels) per lane and fragment, straight from the accumulators
#pragma unroll
    for (int f = 0; f < FH; ++f) {
      const int pp = 16 * (FH * a + f) + l15;
      const size_t off = (((size_t)n * H + ybase) * W + pp) * 64 + ch0;   // elements
      float v[8] = {acc[f][0][0], acc[f][0][1], acc[f][0][2], acc[f][0][3], acc[f][1][0], acc[f][1][1], acc[f][1][2], acc[f][1][3]};
      float xv[8];
      if constexpr (EPI == 3) {
        const uint4 xr = xpre[f];
        const uint32_t xw[4] = {xr.x, xr.y, xr.z, xr.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { xv[2 * e] = __uint_as_float(xw[e] << 16); xv[2 * e + 1] = __uint_as_float(xw[e] & 0xffff0000u); }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (xv[e] * esc[e] + esh[e] > 0.f) ? v[e] : 0.f;
      }
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16_bits(v[2 * e]) | (f32_to_bf16_bits(v[2 * e + 1]) << 16);
      if (!(abl & 8)) *reinterpret_cast<uint4*>(p.out + off) = make_uint4(o[0], o[1], o[2], o[3]);
      if (p.stat_sum) {   // statistics on the value as stored (rounded to bf16), like conv_gemm.hip
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r0 = __uint_as_float(o[e] << 16), r1 = __uint_as_float(o[e] & 0xffff0000u);
          ssum[2 * e] += r0; ssum[2 * e + 1] += r1;
          if constexpr (EPI == 3) { ssq[2 * e] += r0 * xv[2 * e]; ssq[2 * e + 1] += r1 * xv[2 * e + 1]; }
          else { ssq[2 * e] += r0 * r0; ssq[2 * e + 1] += r1 * r1; }
        }
      }
    }
    // the new rows are in place and every wave is done reading this tile's window; the output stores stay in flight (__syncthreads()
    // would wait for them: vmcnt(0), 1.2 us per tile)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // ---- one statistics row per image: over the 16 pixel lanes (shuffles), then over the two pixel halves (LDS)
  if (p.stat_sum) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int sh = 1; sh < 16; sh <<= 1) { ssum[e] += __shfl_xor(ssum[e], sh, 64); ssq[e] += __shfl_xor(ssq[e], sh, 64); }
    float* red = reinterpret_cast<float*>(Wn);   // [2 halves a][2 stats][64 channels]
    if (l15 == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[(a * 2 + 0) * 64 + ch0 + e] = ssum[e]; red[(a * 2 + 1) * 64 + ch0 + e] = ssq[e]; }
    }
    __syncthreads();
    if (tid < 64) {
      p.stat_sum[(size_t)n * p.stat_stride + tid] = red[tid] + red[128 + tid];
      p.stat_sq[(size_t)n * p.stat_stride + tid] = red[64 + tid] + red[192 + tid];
    }
  }
}

template <int FH, int EPI, int FLIP>
int launch_one(const C3Args& a, hipStream_t st) {
  constexpr int lds = WBYTES + RING * (8 * FH + 2) * PITCH;
  static bool attr_done = false;
  auto kern = conv3x3_c64_kernel<FH, EPI, FLIP>;
  if (!attr_done) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
#ifdef MMSKIN_ABLATE
  { const char* e = getenv("MMSKIN_C3_ABLATE"); const_cast<C3Args&>(a).ablate = e ? atoi(e) : 0; }
#endif
  hipLaunchKernelGGL(kern, dim3(a.N), dim3(256), lds, st, a);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

}  // namespace

// MMSKIN_CONV3X3_C64=0 sends these launches back to the tapped kernel (A/B knob).  The shape: 64 -> 64 channels, 3x3 / stride 1 / pad 1,
// 56 x 56 (FH = 7) rows of whole 4-row tiles, at least 32 images (one workgroup per image must cover a useful part of the chip).
bool conv3x3_c64_takes(const ConvShape& s, bool bf16) {
  static const bool on = [] { const char* v = getenv("MMSKIN_CONV3X3_C64"); return !v || atoi(v) != 0; }();
  static const int min_n = [] { const char* v = getenv("MMSKIN_CONV3X3_C64_MIN_N"); return v ? atoi(v) : 32; }();
  return on && bf16 && s.Cin == 64 && s.Cout == 64 && s.kh == 3 && s.kw == 3 && s.stride == 1 && s.pad == 1 && s.W == 56 && s.H % 4 == 0 &&
         s.H >= 8 && s.N >= min_n;
}

static int64_t g_c64_launches = 0;
extern "C" int64_t mmskin_conv3x3_c64_launches(void) { return g_c64_launches; }

// forward: w_staged [cout][tap][cin]; stats (optional): one row per image, *stat_rows_out = N
int launch_conv3x3_c64_fwd(const ConvShape& s, const bf16_t* in, const bf16_t* w_staged, bf16_t* out, float* stat_sum, float* stat_sq,
                           int stat_stride, hipStream_t st) {
  C3Args a = {};
  a.in = in; a.w = w_staged; a.out = out; a.stat_sum = stat_sum; a.stat_sq = stat_sq; a.stat_stride = stat_stride;
  a.N = s.N; a.H = s.H; a.W = s.W;
  ++g_c64_launches;
  return launch_one<7, 0, 0>(a, st);
}

// data gradient: wt_staged [cin][tap][cout]; fuse (optional): conv_gemm.hip's profile 3 -- mask from x * scale + shift, partial sums
// [N][2][64] (sum dz, sum dz * x), fuse->rows_written = N
int launch_conv3x3_c64_dgrad(const ConvShape& s, const bf16_t* dout, const bf16_t* wt_staged, bf16_t* din, DgradFuse* fuse, hipStream_t st) {
  C3Args a = {};
  a.in = dout; a.w = wt_staged; a.out = din;
  a.N = s.N; a.H = s.H; a.W = s.W;
  ++g_c64_launches;
  if (fuse) {
    a.ep_x = reinterpret_cast<const bf16_t*>(fuse->x); a.ep_scale = fuse->scale; a.ep_shift = fuse->shift;
    a.stat_sum = fuse->partial; a.stat_sq = fuse->partial + 64; a.stat_stride = 128;
    fuse->rows_written = s.N;
    return launch_one<7, 3, 1>(a, st);
  }
  return launch_one<7, 0, 1>(a, st);
}

// ResNet / DenseNet stem: 7x7 / stride 2 / pad 3 convolution of the 3-channel image to 64 channels, bf16, as a DIRECT convolution from an
// LDS-resident input window (torchvision resnet.py conv1 reached through loadImageModelClassifier.py:65-75; forward of train_pad_20.py:102).
//
// The gather-GEMM form (conv_gemm.hip launch_stem_conv_fwd: 8 virtual taps of 32 elements) fetched every output pixel's 8 x 64 B of input
// through the texture path again -- 64 KB per 128-pixel tile for 16 KB of distinct bytes -- with four dependent K-tiles per workgroup:
// 286 us for 3.2 M pixels (210 TF/s on the padded K = 256).  Here a workgroup (4 waves) takes FOUR output rows of one image:
//   * the 13 padded input rows they touch are one contiguous 24 KB block of the NHWC4 image (stem_pack): copied to LDS once, 16 B per lane;
//   * the staged weights wv[64][8][32] (rows 0..6 used: K = 7 x 32 = 224, the eighth all-zero row is skipped) sit beside them, swizzled;
//   * a wave owns one output row: 7 pixel fragments x 4 channel fragments of v_mfma_f32_16x16x32_bf16.  For kernel row r the A operand of
//     pixel ox is the 64 B at input row 2 oy + r, padded column 2 ox -- eight pixels x (3 + 1 zero) channels, of which the eighth carries a zero
//     weight -- so a lane's fragment is ONE aligned ds_read_b128 at (ox + k-chunk) x 16 B: no im2col, no address arithmetic per tap;
//   * epilogue: the wave's 16 x 64 fragment through 2.3 KB of LDS to full 128-byte lines, BatchNorm partial sums of the values as stored
//     (one statistics row per workgroup, deterministic).
// LDS 61 KB, <= 200 VGPRs: two workgroups per CU, one loads while the other multiplies.
#include "conv.h"

namespace {
constexpr int ST_ROWS = 4;                       // output rows per workgroup (= waves)
constexpr int ST_IN_ROWS = 2 * ST_ROWS + 5;      // padded input rows they touch
constexpr int ST_MF = 7;                         // 16-pixel fragments per output row (OW = 112)
constexpr int ST_W_BYTES = 7 * 64 * 64;          // [r][cout][64 B]
constexpr int ST_IN_LOADS = 7;                   // 16-byte input chunks per thread: 13 rows x Wp x 8 B <= 7 x 256 x 16 B (Wp <= 274)
constexpr int ST_SP = 144;                       // staging pitch per pixel: 128 B of channels + 16 (a 128 B pitch put every pixel's write on the same banks)
constexpr int ST_STAGE = 16 * ST_SP;             // per wave
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// The staging buffer is wave-private and the LDS serves one wave's instructions in order: a compiler barrier is all the write -> read and
// read -> overwrite hand-offs need (a wavefront-scope fence builtin made hipcc drain the global stores of every fragment: 15 us per workgroup)
#define ST_LDS_ORDER() asm volatile("" ::: "memory")
__device__ __forceinline__ int st_wswz(int cout) { return (cout >> 1) & 3; }   // 16-byte chunk XOR of a weight row

__global__ __launch_bounds__(256, 2) void stem7x7_kernel(const bf16_t* __restrict__ img4, const bf16_t* __restrict__ wv, bf16_t* __restrict__ out,
                                                         float* __restrict__ stat_sum, float* __restrict__ stat_sq, int OH, int OW, int Hp, int Wp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int row_bytes = Wp * 8;
  unsigned char* s_in = smem;                                   // [13][Wp * 8]
  unsigned char* s_w = smem + ST_IN_ROWS * row_bytes;           // [7][64][64] swizzled
  unsigned char* s_st = s_w + ST_W_BYTES;                       // [4 waves][16 px][144 B]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int tiles = OH / ST_ROWS;
  const int n = blockIdx.x / tiles, oy0 = (blockIdx.x - n * tiles) * ST_ROWS;

  {   // input window: rows 2 oy0 .. 2 oy0 + 12 of image n, contiguous.  Every load of the thread is issued before the first LDS store
      // (a rolled load -> store loop ran six dependent memory round trips per workgroup)
    const uint4* src = reinterpret_cast<const uint4*>(img4 + ((size_t)n * Hp + 2 * oy0) * Wp * 4);
    const int nchunk = ST_IN_ROWS * row_bytes / 16;
    const uint4* wsrc = reinterpret_cast<const uint4*>(wv);
    uint4 vi[ST_IN_LOADS], vw[7];
#pragma unroll
    for (int k = 0; k < ST_IN_LOADS; ++k) {
      const int i = tid + 256 * k;
      vi[k] = i < nchunk ? src[i] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {   // weights: chunk i = (cout, r, j) of 64 x 7 x 4
      const int i = tid + 256 * k, j = i & 3, r = (i >> 2) % 7, cout = i / 28;
      vw[k] = wsrc[(cout * 8 + r) * 4 + j];
    }
#pragma unroll
    for (int k = 0; k < ST_IN_LOADS; ++k) {
      const int i = tid + 256 * k;
      if (i < nchunk) reinterpret_cast<uint4*>(s_in)[i] = vi[k];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {   // -> s_w[(r * 64 + cout) * 64 + (j ^ swz(cout)) * 16]
      const int i = tid + 256 * k, j = i & 3, r = (i >> 2) % 7, cout = i / 28;
      *reinterpret_cast<uint4*>(s_w + (r * 64 + cout) * 64 + ((j ^ st_wswz(cout)) << 4)) = vw[k];
    }
  }
  __syncthreads();

  f32x4_t acc[ST_MF][4];
#pragma unroll
  for (int m = 0; m < ST_MF; ++m)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[m][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const unsigned char* a_base = s_in + (2 * wid) * row_bytes + (l15 + g) * 16;
  const unsigned char* b_base = s_w + l15 * 64;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    uint4 fb[4], fa[ST_MF];
#pragma unroll
    for (int f = 0; f < 4; ++f) fb[f] = *reinterpret_cast<const uint4*>(b_base + (r * 64 + f * 16) * 64 + ((g ^ st_wswz(f * 16 + l15)) << 4));
#pragma unroll
    for (int m = 0; m < ST_MF; ++m) fa[m] = *reinterpret_cast<const uint4*>(a_base + r * row_bytes + m * 256);
#pragma unroll
    for (int m = 0; m < ST_MF; ++m)
#pragma unroll
      for (int f = 0; f < 4; ++f)
        acc[m][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[f]), __builtin_bit_cast(bf16x8_t, fa[m]), acc[m][f], 0, 0, 0);
  }

  // ---- epilogue.  acc[m][f]: pixel 16 m + l15, channels 16 f + 4 g .. + 3.  Wave-private staging: 16 pixels x 128 B.
  unsigned char* st = s_st + wid * ST_STAGE;
  const int oy = oy0 + wid;
  bf16_t* orow = out + (((size_t)n * OH + oy) * OW) * 64;
  float ssum[8], ssq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
  const int ch = lane & 7, pr = lane >> 3;   // read-back: 16-byte channel chunk, pixel row of a pass
#pragma unroll
  for (int m = 0; m < ST_MF; ++m) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint32_t lo = f32_to_bf16_bits(acc[m][f][0]) | (f32_to_bf16_bits(acc[m][f][1]) << 16);
      const uint32_t hi = f32_to_bf16_bits(acc[m][f][2]) | (f32_to_bf16_bits(acc[m][f][3]) << 16);
      *reinterpret_cast<uint2*>(st + l15 * ST_SP + (f * 16 + 4 * g) * 2) = make_uint2(lo, hi);
    }
    ST_LDS_ORDER();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int px = ps * 8 + pr;
      const uint4 v = *reinterpret_cast<const uint4*>(st + px * ST_SP + ch * 16);
      *reinterpret_cast<uint4*>(orow + (size_t)(m * 16 + px) * 64 + ch * 8) = v;
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = bf16_bits_to_f32(w4[e] & 0xffffu), b = bf16_bits_to_f32(w4[e] >> 16);
        ssum[2 * e] += a; ssq[2 * e] += a * a; ssum[2 * e + 1] += b; ssq[2 * e + 1] += b * b;
      }
    }
    ST_LDS_ORDER();
  }
  if (stat_sum) {
    // lanes with equal (lane & 7) hold the same eight channels: fold lane bits 3..5, then the four waves through LDS (the input window is dead)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int d = 8; d < 64; d <<= 1) { ssum[e] += __shfl_xor(ssum[e], d); ssq[e] += __shfl_xor(ssq[e], d); }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(s_in);   // [4][2][64]
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[(wid * 2) * 64 + lane * 8 + e] = ssum[e]; red[(wid * 2 + 1) * 64 + lane * 8 + e] = ssq[e]; }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const float t = (red[(0 * 2 + which) * 64 + c] + red[(1 * 2 + which) * 64 + c]) + (red[(2 * 2 + which) * 64 + c] + red[(3 * 2 + which) * 64 + c]);
      (which ? stat_sq : stat_sum)[(size_t)blockIdx.x * 64 + c] = t;
    }
  }
}
}   // namespace

static int g_stem7_launches = 0;
extern "C" int64_t mmskin_stem7x7_launches(void) { return g_stem7_launches; }

// MMSKIN_STEM7X7=0: the gather-GEMM form for every shape (A/B, tests)
bool stem7x7_takes(int OH, int OW, int Hp, int Wp) {
  static const int on = [] { const char* v = getenv("MMSKIN_STEM7X7"); return v ? atoi(v) : 1; }();
  return on && OW == 16 * ST_MF && OH % ST_ROWS == 0 && Hp >= 2 * OH + 5 && Wp >= 2 * OW + 8 && Wp % 2 == 0 && ST_IN_ROWS * Wp * 8 <= ST_IN_LOADS * 256 * 16 &&
         ST_IN_ROWS * Wp * 8 + ST_W_BYTES + ST_ROWS * ST_STAGE <= 80 * 1024;
}
int stem7x7_stat_rows(int N, int OH) { return N * (OH / ST_ROWS); }
int launch_stem7x7_fwd(int N, int OH, int OW, int Hp, int Wp, const bf16_t* img4, const bf16_t* wv, bf16_t* out, float* stat_sum, float* stat_sq,
                       hipStream_t st) {
  ARG_CHECK(stem7x7_takes(OH, OW, Hp, Wp), "stem7x7: shape OH=%d OW=%d Hp=%d Wp=%d not taken", OH, OW, Hp, Wp);
  const int lds = ST_IN_ROWS * Wp * 8 + ST_W_BYTES + ST_ROWS * ST_STAGE;
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(stem7x7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_done = true;
  }
  ++g_stem7_launches;
  hipLaunchKernelGGL(stem7x7_kernel, dim3(N * (OH / ST_ROWS)), dim3(256), lds, st, img4, wv, out, stat_sum, stat_sq, OH, OW, Hp, Wp);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

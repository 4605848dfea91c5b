// Pipelined tapped gather-GEMM (forward conv + dgrad) for gfx950 -- the production variant.
//
// Same math and argument block as conv_gemm.hip (kept as the simple reference variant), restructured
// around what the PMC counters showed (profiles/r01_b_*): with one barrier-synchronised K-tile in
// flight per workgroup the kernel was latency-bound (L2 hit 91 %, MFMA busy 13 %, waves parked 55 %).
//
//   * 512-thread workgroup = 8 waves = 2 per SIMD, ONE workgroup per CU.
//   * NS-stage LDS ring (3 x (BM+BN) x 128 B) filled by LDS-DMA (global_load_lds_dwordx4).  The DMA is
//     issued from inline asm: hipcc otherwise treats it as an LDS store that may alias every ds_read
//     and drains vmcnt(0) in front of the MFMA loop.  Two K-tiles stay in flight across the single
//     raw s_barrier per K-tile; completion is tracked with a counted s_waitcnt vmcnt(N).
//   * bank swizzle on the per-lane SOURCE chunk (the DMA writes LDS linearly), same XOR on ds_read.
//   * epilogue identical to conv_gemm.hip: LDS-staged packed rows, full-line stores, BN partial sums.
#include "conv.h"

template <typename T> struct MmaP;
template <> struct MmaP<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct MmaP<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

#define PIPE_TAP_BYTES 256
#define PIPE_NS 3
#define PIPE_THREADS 512

__device__ uint4 g_pipe_zero_page[16];

template <int BM, int BN, typename T> constexpr int conv_pipe_lds_bytes() {
  constexpr int ring = PIPE_NS * (BM + BN) * 128;
  constexpr int cpitch = BN * (int)sizeof(T) + 16;
  constexpr int rows_per_pass = PIPE_THREADS / (BN / DT<T>::EPC);
  constexpr int cs = BM * cpitch + 2 * rows_per_pass * BN * 4;
  return PIPE_TAP_BYTES + (ring > cs ? ring : cs);
}

// one LDS-DMA: 64 lanes x 16 B land at lds_off + lane*16 (M0 = wave-uniform LDS byte offset)
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(gsrc) : "memory");
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(PIPE_THREADS, 2) void conv_pipe_kernel(const ConvGemmArgs p) {
  constexpr int EPC = DT<T>::EPC, BK = DT<T>::BK;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, FM = WM / 16, FN = WN / 16;
  constexpr int AP = BM / 64, BP = BN / 64;      // 64 rows per loader pass (512 threads x 16 B)
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int CPITCH = BN * (int)sizeof(T) + 16;
  constexpr int LOADS = AP + BP;                 // LDS-DMA instructions per thread per K-tile
  static_assert(WAVES_M * WAVES_N == 8, "8 waves");
  static_assert(LOADS >= 2 && LOADS <= 6, "vmcnt immediates below cover 2..6 loads per tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* s_tap = reinterpret_cast<int*>(smem);
  unsigned char* ring = smem + PIPE_TAP_BYTES;

  const int tid = threadIdx.x;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mblk = tile / p.nblk_n, nblk = tile - mblk * p.nblk_n;
  int ci = 0;
  for (int i = 1; i < p.ncls; ++i)
    if (mblk >= p.cls[i].mblk_start) ci = i;
  const int a_dim = p.cls[ci].a_dim, b_dim = p.cls[ci].b_dim;
  const int ntaps = p.cls[ci].ntaps, rows = p.cls[ci].rows;
  const int m0 = (mblk - p.cls[ci].mblk_start) * BM;
  const int n0 = nblk * BN;
  const int C = p.C, IH = p.IH, IW = p.IW;

  if (tid < ntaps) {
    int oy = p.cls[ci].offy[tid], ox = p.cls[ci].offx[tid], wt = p.cls[ci].wtap[tid];
    s_tap[tid] = (oy * IW + ox) * p.Cpitch;
    s_tap[16 + tid] = (oy & 0xffff) | (ox << 16);
    s_tap[32 + tid] = wt * C;
  }
  __syncthreads();

  // ---- loader state: this lane fills LDS position jc of rows lr + 64*i with source chunk sc
  const int lr = tid >> 3, jc = tid & 7;
  const int sc = jc ^ ((lr >> 1) & 7);
  int a_base[AP], a_iy[AP], a_ix[AP];
  const int ab = a_dim * b_dim;
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    int m = m0 + lr + 64 * i;
    if (m < rows) {
      int img = m / ab, rem = m - img * ab;
      int a = rem / b_dim, b = rem - a * b_dim;
      a_iy[i] = a * p.Sy;
      a_ix[i] = b * p.Sx;
      a_base[i] = ((img * IH + a_iy[i]) * IW + a_ix[i]) * p.Cpitch;
    } else {
      a_iy[i] = -(1 << 20);
      a_ix[i] = -(1 << 20);
      a_base[i] = 0;
    }
  }
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);
  const unsigned char* w_b = reinterpret_cast<const unsigned char*>(p.w) + (size_t)(n0 + lr) * p.wrow * sizeof(T);
  const size_t w_pass = (size_t)64 * p.wrow * sizeof(T);
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(g_pipe_zero_page);
  int tap = (sc * EPC) / C, c = sc * EPC - tap * C;
  const int nk = ntaps * C / BK;
  const int wid_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int abl = p.ablate;
  // LDS byte offset of the ring (dynamic LDS starts at the workgroup's LDS base)
  const uint32_t ring_off = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)ring);
  const uint32_t wave_off = ring_off + wid_u * 8 * 128;

#define ISSUE_TILE(stage)                                                                     \
  do {                                                                                        \
    const int toff = s_tap[tap], tyx = s_tap[16 + tap], wk = s_tap[32 + tap] + c;             \
    const int oy = (int)(short)(tyx & 0xffff), ox = tyx >> 16;                                \
    const uint32_t sbase = wave_off + (stage) * STAGE;                                        \
    if (!(abl & 1)) _Pragma("unroll") for (int i = 0; i < AP; ++i) {                          \
      const int iy = a_iy[i] + oy, ix = a_ix[i] + ox;                                         \
      const bool ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;             \
      const unsigned char* src = in_b + (int64_t)(a_base[i] + toff + c) * (int)sizeof(T);     \
      glds16(ok ? src : zero_page, sbase + i * 64 * 128);                                     \
    }                                                                                         \
    if (!(abl & 2)) _Pragma("unroll") for (int i = 0; i < BP; ++i)                            \
      glds16(w_b + i * w_pass + (size_t)wk * sizeof(T), sbase + A_BYTES + i * 64 * 128);      \
    c += BK;                                                                                  \
    const int wrap = (c >= C ? 1 : 0) + (c >= 2 * C ? 1 : 0);                                 \
    c -= wrap * C;                                                                            \
    tap += wrap;                                                                              \
  } while (0)

  const int wid = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wm = wid / WAVES_N, wn = wid - wm * WAVES_N;
  f32x4_t acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // prologue: two tiles in flight
  if (nk > 0) ISSUE_TILE(0);
  if (nk > 1) ISSUE_TILE(1);
  const unsigned char* Ab0 = ring + (wm * WM + l15) * 128;
  const unsigned char* Bb0 = ring + A_BYTES + (wn * WN + l15) * 128;
  const int sw = l15 >> 1;
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // my DMA of tile kt has landed once at most tile kt+1's loads are still outstanding
    if (kt + 1 < nk) {
      if constexpr (LOADS == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if constexpr (LOADS == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if constexpr (LOADS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (LOADS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // every wave's part of tile kt has landed AND every wave has finished reading tile kt-1
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) {
      int nxt = stage + 2;
      if (nxt >= PIPE_NS) nxt -= PIPE_NS;
      ISSUE_TILE(nxt);   // overwrites the stage tile kt-1 was read from
    }
    const unsigned char* Ab = Ab0 + stage * STAGE;
    const unsigned char* Bb = Bb0 + stage * STAGE;
    if (!(abl & 4))
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int off = ((4 * s + g) ^ sw) << 4;
      uint4 fa[FM], fb[FN];
#pragma unroll
      for (int j = 0; j < FM; ++j) fa[j] = *reinterpret_cast<const uint4*>(Ab + j * 16 * 128 + off);
#pragma unroll
      for (int i = 0; i < FN; ++i) fb[i] = *reinterpret_cast<const uint4*>(Bb + i * 16 * 128 + off);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) MmaP<T>::run(fb[i], fa[j], acc[i][j]);
    }
    if (++stage == PIPE_NS) stage = 0;
  }
#undef ISSUE_TILE
  __syncthreads();   // all ds_reads of the last tile retired before the ring is reused as C staging

  // ---- epilogue: acc -> LDS [pixel][channel] (packed) -> coalesced 16-byte stores
  unsigned char* Cs = smem + PIPE_TAP_BYTES;
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      int prow = wm * WM + j * 16 + l15;
      int ccol = wn * WN + i * 16 + g * 4;
      unsigned char* dst = Cs + prow * CPITCH + ccol * (int)sizeof(T);
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      } else {
        uint32_t lo = f32_to_bf16_bits(acc[i][j][0]) | (f32_to_bf16_bits(acc[i][j][1]) << 16);
        uint32_t hi = f32_to_bf16_bits(acc[i][j][2]) | (f32_to_bf16_bits(acc[i][j][3]) << 16);
        *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
      }
    }
  __syncthreads();

  constexpr int CH_PER_ROW = BN / EPC;
  constexpr int ROWS_PER_PASS = PIPE_THREADS / CH_PER_ROW;
  const int cj = tid % CH_PER_ROW, r0 = tid / CH_PER_ROW;
  float ssum[EPC], ssq[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
  unsigned char* out_b = reinterpret_cast<unsigned char*>(p.out);
  const unsigned char* add_b = reinterpret_cast<const unsigned char*>(p.addend);
  const bool simple_rows = (p.OS == 1 && p.ncls == 1);
  const int ph = p.cls[ci].ph, pw = p.cls[ci].pw;
  for (int row = r0; row < BM; row += ROWS_PER_PASS) {
    int m = m0 + row;
    if (m >= rows) break;
    int orow = m;
    if (!simple_rows) {
      int img = m / ab, rem = m - img * ab;
      int a = rem / b_dim, b = rem - a * b_dim;
      orow = (img * p.OHf + a * p.OS + ph) * p.OWf + b * p.OS + pw;
    }
    Chunk<T> v;
    v.load(Cs + row * CPITCH + cj * 16);
    size_t goff = ((size_t)orow * p.Cout + n0 + cj * EPC) * sizeof(T);
    if (add_b) {
      Chunk<T> ad;
      ad.load(add_b + goff);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] += ad.v[e];
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) { ssum[e] += v.v[e]; ssq[e] += v.v[e] * v.v[e]; }
    if (!(abl & 8)) v.store(out_b + goff);
  }
  if (p.stat_sum) {
    float* red = reinterpret_cast<float*>(Cs + BM * CPITCH);  // [2][ROWS_PER_PASS][BN]
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      red[r0 * BN + cj * EPC + e] = ssum[e];
      red[(ROWS_PER_PASS + r0) * BN + cj * EPC + e] = ssq[e];
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int r = 0; r < ROWS_PER_PASS; ++r) { s += red[r * BN + tid]; q += red[(ROWS_PER_PASS + r) * BN + tid]; }
      p.stat_sum[(size_t)mblk * p.Cout + n0 + tid] = s;
      p.stat_sq[(size_t)mblk * p.Cout + n0 + tid] = q;
    }
  }
}

// ------------------------------------------------------------------------------------------ host
template <typename T, int BM, int BN, int WMv, int WNv>
static int launch_pipe_cfg(const ConvGemmArgs& a, hipStream_t st) {
  constexpr int lds = conv_pipe_lds_bytes<BM, BN, T>();
  static bool attr_done = false;
  auto kern = conv_pipe_kernel<T, BM, BN, WMv, WNv>;
  if (!attr_done) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  int grid = a.total_mblk * a.nblk_n;
  if (grid == 0) return MMSKIN_OK;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(PIPE_THREADS), lds, st, a);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

// Row-block height of the pipelined variant is fixed at 128 so stat slabs / class tables built by
// conv_gemm.hip's host code (CONV_BM = 128) stay valid.
template <typename T>
int launch_conv_pipe(ConvGemmArgs& a, hipStream_t st) {
  if (a.Cout % 128 == 0) {
    a.nblk_n = a.Cout / 128;
    return launch_pipe_cfg<T, 128, 128, 2, 4>(a, st);
  }
  a.nblk_n = a.Cout / 64;
  return launch_pipe_cfg<T, 128, 64, 4, 2>(a, st);
}
template int launch_conv_pipe<float>(ConvGemmArgs&, hipStream_t);
template int launch_conv_pipe<bf16_t>(ConvGemmArgs&, hipStream_t);

// Tapped gather-GEMM for gfx950: forward convolution and data-gradient (dgrad) as one kernel.
//
//   out[pix][n] = sum_{tap,c} in[pix + tap][c] * w[n][tap][c]        (+ addend[pix][n])
//
// Replaces the ATen/cuDNN conv2d forward + conv_backward(input) that the reference reaches through
// torchvision's ResNet (multimodalIntraInterModal.py:167 -> loadImageModelClassifier.py:65-75).
//
// Design (MI355X): 256-thread workgroup (4 waves, one per SIMD), BM x BN output tile, K-tiles of
// 128 bytes per row (64 bf16 / 32 f32).  Both operands are K-contiguous, so a lane's 16-byte chunk
// is directly an MFMA operand fragment (v_mfma_f32_16x16x32_bf16, or 4x v_mfma_f32_16x16x4_f32 in
// the exact-f32 parity mode).  Register-staged double buffering: global loads of tile k+1 are in
// flight while tile k is read from LDS and multiplied; one barrier per K-tile.  LDS rows are
// XOR-swizzled (chunk ^= (row>>1)&7) so ds_read_b128 fragment reads are bank-conflict free.
// Weights are the MFMA "A" operand and pixels the "B" operand, so each lane ends up holding four
// consecutive output channels of one pixel -> packed LDS-staged epilogue with full-line stores.
// Optional epilogue: per-row-block BatchNorm partial sums (deterministic slab, no atomics), and an
// addend tensor (residual / accumulation).
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "conv.h"

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// LDS header: tap table (3 x 16 ints, 64 reserved) | output-row table of strided / multi-class launches (BM ints) | source-row table (3 x BM ints)
template <int BM> constexpr int tap_lds_bytes() { return (64 + 4 * BM) * 4; }

// 256 bytes of zeros in global memory: the source of every out-of-image (padding) tap.
__device__ uint4 g_zero_page[16];

// NST == 8 selects the 8-phase pipelined main loop (PIPE): two 64 KB stages of [A: 2 wave rows x 128 tile rows][B: 256 rows] x 128 B
// plus one int4 per K-tile (source offset, tap index, weight offset) behind the tap / row tables.
constexpr int KT_LDS_BYTES = 2048;   // <= 128 K-tiles
template <int BM, int BN, typename T, int NST, int NT = 256> constexpr int conv_gemm_lds_bytes() {
  constexpr int ab = NST == 8 ? 2 * (256 + BN) * 128 : NST * (BM + BN) * 128;
  constexpr int cpitch = BN * (int)sizeof(T) + 16;
  constexpr int rows_per_pass = NT / (BN / DT<T>::EPC);
  constexpr int cs = BM * cpitch;                       // C staging; the stat reduction buffer overlays it
  constexpr int red = 3 * rows_per_pass * (BN + 8) * 4;
  constexpr int m1 = ab > cs ? ab : cs;
  return tap_lds_bytes<BM>() + (NST == 8 ? KT_LDS_BYTES : 0) + (m1 > red ? m1 : red);
}

// EPI = 0: plain epilogue (forward conv: store + BN partial sums); EPI = 1: addend and/or the fused
// BatchNorm-backward mask + sums (dgrad launches).  Separate instantiations keep the forward lean.
// NST = LDS stages: 2 = double-buffered K loop (2 workgroups per CU); 1 = single buffer, ~35 KB of LDS so
// 3-4 workgroups share a CU -- for short-K, output-heavy layers where the epilogue dominates and only
// inter-workgroup overlap can hide it.
// WAVES_M x WAVES_N = 4 waves (256 threads, 128-row tiles, 2 - 4 workgroups per CU) or 8 waves (512 threads, the 256 x 256 tile: ONE
// workgroup per CU with two waves per SIMD, half the L2 -> LDS fill bytes per FLOP of the 128 x 128 tile).
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int NST>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (EPI == 6 && WAVES_M * WAVES_N == 4) ? 4 : 2) void conv_gemm_kernel(const ConvGemmArgs p) {
  constexpr int EPC = DT<T>::EPC, BK = DT<T>::BK;
  constexpr int NT = 64 * WAVES_M * WAVES_N;     // threads
  constexpr int PR = NT / 8;                     // tile rows one loader pass covers (8 chunks of 16 B per 128-byte row)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, FM = WM / 16, FN = WN / 16;
  constexpr int AP = BM / PR, BP = BN / PR;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int CPITCH = BN * (int)sizeof(T) + 16;
  constexpr bool PIPE = NST == 8;
  constexpr int TAP_LDS_BYTES = tap_lds_bytes<BM>() + (PIPE ? KT_LDS_BYTES : 0);
  static_assert(WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8, "4 or 8 waves");
  static_assert(PIPE || (BM <= NT && BN <= NT && BM % PR == 0 && BN % PR == 0), "tile / thread mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* s_tap = reinterpret_cast<int*>(smem);
  unsigned char* As = smem + TAP_LDS_BYTES;
  unsigned char* Bs = As + NST * A_BYTES;

  const int tid = threadIdx.x;
#ifdef MMSKIN_ABLATE
  if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8] = clock64();
#endif
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  int mblk = tile / p.nblk_n, nblk = tile - mblk * p.nblk_n;
  if (p.group_m > 1) {   // grouped order (see ConvGemmArgs::group_m); wave-uniform
    const int gsz = p.group_m * p.nblk_n, g = tile / gsz, t = tile - g * gsz;
    const int mg = g * p.group_m, gm = min(p.group_m, p.total_mblk - mg);
    nblk = t / gm; mblk = mg + (t - nblk * gm);
  }
  int ci = 0;
  for (int i = 1; i < p.ncls; ++i)
    if (mblk >= p.cls[i].mblk_start) ci = i;
  const int a_dim = p.cls[ci].a_dim, b_dim = p.cls[ci].b_dim;
  const int ntaps = p.cls[ci].ntaps, rows = p.cls[ci].rows;
  // PIPE tiles may use fewer valid rows than they compute (bm_step <= BM): 196-row tiles (one 14 x 14 image) put the layer-3 / 4
  // convolutions of a 256-image batch on exactly 256 / 512 / 1024 workgroups instead of 196 / 392 / 784
  const int bm_step = PIPE ? p.bm_step : BM;
  const int m0 = (mblk - p.cls[ci].mblk_start) * bm_step;
  const int n0 = nblk * BN;
  const int C = p.C, IH = p.IH, IW = p.IW;

  if (tid < ntaps) {
    int oy = p.cls[ci].offy[tid], ox = p.cls[ci].offx[tid], wt = p.cls[ci].wtap[tid];
    s_tap[tid] = (oy * IW + ox) * p.Cpitch;
    s_tap[16 + tid] = (oy & 0xffff) | (ox << 16);
    s_tap[32 + tid] = wt * C;
  }
  // source pixel of every tile row: (image, y, x) costs two integer divisions -- one thread per ROW does them, not every
  // thread for each of its rows (eight threads share a row; the divisions were most of a short-K workgroup's prologue)
  const int ab = a_dim * b_dim;
  int* s_src = s_tap + 64 + BM;   // [3][BM]: base offset, y, x
  int pipe_min_toff = 0;   // PIPE: most negative tap offset of this class (elements)
  const bool simple_src = !PIPE && p.simple_src != 0;   // 1x1 / stride 1: no source-row table (in-kernel stamps: the prologue was a fifth of a short-K workgroup's life)
  if constexpr (PIPE) {
    // tap offsets as three packed words each (scalar loads issued together; a rolled loop over the taps waited for two dependent
    // scalar loads per tap and was a quarter of the 3.7 us prologue of a 3x3 tile)
    static_assert(MMSKIN_MAX_TAPS == 12 && offsetof(TapClass, offy) % 4 == 0 && offsetof(TapClass, offx) % 4 == 0 && offsetof(TapClass, wtap) % 4 == 0, "packed tap words");
    const uint32_t* oyw = reinterpret_cast<const uint32_t*>(p.cls[ci].offy);
    const uint32_t* oxw = reinterpret_cast<const uint32_t*>(p.cls[ci].offx);
    const uint32_t wy[3] = {oyw[0], oyw[1], oyw[2]}, wx[3] = {oxw[0], oxw[1], oxw[2]};
#define TAP_Y(t) ((int)(int8_t)(wy[(t) >> 2] >> (8 * ((t) & 3))))
#define TAP_X(t) ((int)(int8_t)(wx[(t) >> 2] >> (8 * ((t) & 3))))
    // [0][BM] source offset of the row's pixel (elements), [1][BM] bit t = tap t of this row lies inside the image
    if (tid < BM) {
      const int m = m0 + tid;
      int base = 0;
      uint32_t mask = 0;
      if (tid < bm_step && m < rows) {
        const int img = m / ab, rem = m - img * ab;
        const int a = rem / b_dim, b = rem - a * b_dim;
        const int iy = a * p.Sy, ix = b * p.Sx;
        base = ((img * IH + iy) * IW + ix) * p.Cpitch;
#pragma unroll
        for (int t = 0; t < MMSKIN_MAX_TAPS; ++t) {
          const int y = iy + TAP_Y(t), x = ix + TAP_X(t);
          mask |= (t < ntaps && (unsigned)y < (unsigned)IH && (unsigned)x < (unsigned)IW) ? (1u << t) : 0u;
        }
      }
      s_src[tid] = base; s_src[BM + tid] = (int)mask;
    }
    // one int4 per K-tile (C % 64 == 0: a K-tile never straddles two taps), in BYTES: gather offset above the most negative tap
    // offset (the buffer descriptor's base carries that one, so the scalar offset of the DMA is never negative), tap bit, weight-row offset
#pragma unroll
    for (int t = 0; t < MMSKIN_MAX_TAPS; ++t) {
      const int toff = (TAP_Y(t) * IW + TAP_X(t)) * p.Cpitch;
      pipe_min_toff = (t < ntaps && toff < pipe_min_toff) ? toff : pipe_min_toff;
    }
#undef TAP_Y
#undef TAP_X
    int4* s_kt = reinterpret_cast<int4*>(smem + tap_lds_bytes<BM>());
    const int cpt = C / BK, nkt = ntaps * cpt;
    for (int kt = tid; kt < nkt; kt += NT) {
      const int t = kt / cpt, c0 = (kt - t * cpt) * BK;
      const int oy = p.cls[ci].offy[t], ox = p.cls[ci].offx[t], wt = p.cls[ci].wtap[t];
      s_kt[kt] = make_int4(((oy * IW + ox) * p.Cpitch - pipe_min_toff + c0) * (int)sizeof(T), 1 << t, (wt * C + c0) * (int)sizeof(T), 0);
    }
  } else if (!simple_src && tid < BM) {
    const int m = m0 + tid;
    int base = 0, iy = -(1 << 20), ix = -(1 << 20);
    if (m < rows) {
      const int img = m / ab, rem = m - img * ab;
      const int a = rem / b_dim, b = rem - a * b_dim;
      iy = a * p.Sy;
      ix = b * p.Sx;
      base = ((img * IH + iy) * IW + ix) * p.Cpitch;
    }
    s_src[tid] = base; s_src[BM + tid] = iy; s_src[2 * BM + tid] = ix;
  }
  __syncthreads();

  // ---- loader state: this thread moves chunk column jc of rows lr + 32*i
  const int lr = tid >> 3, jc = tid & 7;
  int a_base[AP], a_iy[AP], a_ix[AP];
#pragma unroll
  for (int i = 0; i < (PIPE ? 0 : AP); ++i) {
    const int r = lr + PR * i;
    if (simple_src) {
      const int m = m0 + r;
      const bool in = m < rows;
      a_base[i] = in ? m * p.Cpitch : 0; a_iy[i] = in ? 0 : -(1 << 20); a_ix[i] = 0;
    } else {
      a_base[i] = s_src[r]; a_iy[i] = s_src[BM + r]; a_ix[i] = s_src[2 * BM + r];
    }
  }
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(p.in);
  const unsigned char* in2_b = reinterpret_cast<const unsigned char*>(p.in2);
  const bool two_src = !PIPE && p.in2 != nullptr;
  const unsigned char* w_b = reinterpret_cast<const unsigned char*>(p.w) +
                             (size_t)(n0 + lr) * p.wrow * sizeof(T);
  const size_t w_pass = (size_t)PR * p.wrow * sizeof(T);

  // LDS-DMA staging (global_load_lds_dwordx4): one wave-instruction lands 64 x 16 B = 8 tile rows
  // contiguously in LDS (wave-uniform base + lane*16), so the bank swizzle is applied to the per-lane
  // SOURCE chunk instead: LDS position jc of row r receives source chunk jc ^ ((r>>1)&7), and the
  // fragment reads below apply the same XOR.  No staging VGPRs, no ds_write.
  const int sc = jc ^ ((lr >> 1) & 7);            // source chunk this lane fetches (same for every pass)
  int tap = (sc * EPC) / C, c = sc * EPC - tap * C;
  const int nk = ntaps * C / BK;
  const unsigned char* zero_page = reinterpret_cast<const unsigned char*>(g_zero_page);
  const int wid_u = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id, provably uniform
#ifdef MMSKIN_ABLATE   // `make ablate` (scripts/ only): the production library has no work-skipping switch
  const int abl = p.ablate;
#else
  constexpr int abl = 0;
#endif
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifdef MMSKIN_ABLATE
#define STAMP(i) do { if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + (i)] = clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// The DMA is issued from inline asm: through the builtin, hipcc models it as an LDS store that may alias
// every ds_read and drains s_waitcnt vmcnt(0) in front of the MFMA loop, which serialised load and
// compute (ablation in profiles/r01_c_conv_ablation.txt: 35 us loads + 35 us MFMA = 61 us, no overlap).
// M0 = wave-uniform LDS byte offset; one wait state between the M0 write and the DMA.
#define GLDS16(gsrc, ldst)                                                                                \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"((uint32_t)(uintptr_t)(lds_ptr_t)(ldst)), \
               "v"((const void*)(gsrc)) : "memory")
#define LOAD_TILE(buf)                                                                       \
  do {                                                                                       \
    const int toff = s_tap[tap], tyx = s_tap[16 + tap], wk = s_tap[32 + tap] + c;            \
    const int oy = (int)(short)(tyx & 0xffff), ox = tyx >> 16;                               \
    if (!(abl & 1)) _Pragma("unroll") for (int i = 0; i < AP; ++i) {                         \
      const int iy = a_iy[i] + oy, ix = a_ix[i] + ox;                                        \
      const bool ok = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;            \
      const unsigned char* src = in_b + (int64_t)(a_base[i] + toff + c) * (int)sizeof(T);    \
      if (two_src && c >= p.K1) src = in2_b + (int64_t)((a_base[i] >> p.pitch2_shift) + c - p.K1) * (int)sizeof(T); \
      GLDS16(ok ? src : zero_page, As + (buf) * A_BYTES + (PR * i + wid_u * 8) * 128);       \
    }                                                                                        \
    if (!(abl & 2)) _Pragma("unroll") for (int i = 0; i < BP; ++i)                           \
      GLDS16(w_b + i * w_pass + (size_t)wk * sizeof(T), Bs + (buf) * B_BYTES + (PR * i + wid_u * 8) * 128); \
    c += BK;                                                                                 \
    const int wrap = (c >= C ? 1 : 0) + (c >= 2 * C ? 1 : 0);                                \
    c -= wrap * C;                                                                           \
    tap += wrap;                                                                             \
  } while (0)

  const int wid = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int wm = wid / WAVES_N, wn = wid - wm * WAVES_N;
  f32x4_t acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const unsigned char* Ab0 = As + (wm * WM + l15) * 128;
  const unsigned char* Bb0 = Bs + (wn * WN + l15) * 128;
  const int sw = l15 >> 1;  // (row>>1)&7 for row = 16*f + l15
#define COMPUTE_TILE(cur)                                                                      \
  do {                                                                                         \
    const unsigned char* Ab = Ab0 + (cur) * A_BYTES;                                           \
    const unsigned char* Bb = Bb0 + (cur) * B_BYTES;                                           \
    if (!(abl & 4))                                                                            \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                          \
        const int off = ((4 * s + g) ^ sw) << 4;                                               \
        uint4 fa[FM], fb[FN];                                                                  \
        _Pragma("unroll") for (int j = 0; j < FM; ++j) fa[j] = *reinterpret_cast<const uint4*>(Ab + j * 16 * 128 + off); \
        _Pragma("unroll") for (int i = 0; i < FN; ++i) fb[i] = *reinterpret_cast<const uint4*>(Bb + i * 16 * 128 + off); \
        _Pragma("unroll") for (int i = 0; i < FN; ++i)                                         \
          _Pragma("unroll") for (int j = 0; j < FM; ++j) Mma<T>::run(fb[i], fa[j], acc[i][j]); \
      }                                                                                        \
  } while (0)

  if constexpr (PIPE) {
    // ---- pipelined main loop (MI355X: LDS-DMA ring with COUNTED vmcnt, raw s_barrier, two waves per SIMD).
    // A K-tile (64 deep) is four phases of 16 MFMAs per wave (one quadrant of the wave's 16*FM x 64 output tile); its operands are four
    // 16 KB units in one of two stages (K-tile parity):
    //   A0 = fragment rows 0-3 of both wave rows     B0 = channels 0-31 of every wave column
    //   A1 = fragment rows 4..FM-1                   B1 = channels 32-63
    // Unit sequence q = 4t + {A0, B0, B1, A1}.  Every wave runs the same stream, software-pipelined one phase deep:
    //   phase g:  MFMA cluster of quadrant g  ||  ds_read of unit g + 2's fragments (used by cluster g + 1 / g + 2)
    //             ||  LDS-DMA issue of unit g + 6  ->  s_waitcnt vmcnt(6)  ->  s_barrier
    // The fragment reads and the DMA sit BETWEEN the wave's own MFMAs (sched_group_barrier / sched_barrier pin that): an MFMA
    // occupies the issue port for 8 of its 16 cycles and two waves share the pipe, so the ~25 non-MFMA instructions of a phase
    // issue in the gaps instead of in front of the cluster (measured before this form: load-side issue time and MFMA time ADDED,
    // 790 cycles per phase for 512 cycles of MFMA; profiles/r03_experiments.txt).
    //   RAW: unit q is read in phase q - 2; the barrier in front of that phase follows every wave's wait of phase q - 3, which
    //        leaves only units > (q - 3) + 3 in flight.
    //   WAR: unit q + 8 (same region) is issued in phase q + 2, four phases after the reads, which were consumed (lgkmcnt) three
    //        barriers earlier.
    // Operands arrive through `buffer_load_dwordx4 ... offen lds`: the per-lane part of the address is a loop-invariant 32-bit
    // VGPR offset, the K-tile's offset is the instruction's SCALAR offset, and a row whose tap lies outside the image (or past
    // the tile's valid rows) gets an out-of-range offset, for which the buffer unit writes zeros: no zero page, no 64-bit address
    // arithmetic, three VALU instructions per gathered row and none for the weights.
    static_assert(sizeof(T) == 2 && WAVES_M == 2 && WAVES_N == 4 && BN == 256 && FM >= 5 && FM <= 8, "PIPE tile");
    constexpr int STAGE = (256 + BN) * 128;
    constexpr uint32_t OOB = 0xF0000000u;   // >= any num_records the launcher admits
    typedef uint32_t srd_t __attribute__((ext_vector_type(4)));
    const int4* s_kt = reinterpret_cast<const int4*>(smem + tap_lds_bytes<BM>());
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)As;
    const int NQ = 4 * nk;
    const int lane_ = tid & 63;
    // gather rows of this thread: unit u (A0 / A1), wave row i -> tile row i * WM + u * 64 + lr
    uint32_t pa_off[2][2], pa_mask[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int local = u * 64 + lr;
        const bool okr = local < WM;
        const int trow = okr ? i * WM + local : 0;
        pa_off[u][i] = (uint32_t)(s_src[trow] + sc * EPC) * (uint32_t)sizeof(T);
        pa_mask[u][i] = okr ? (uint32_t)s_src[BM + trow] : 0u;
      }
    const int rho0 = (wid_u >> 2) * 64 + (wid_u & 3) * 8;                 // first B row of this wave's 8-row DMA piece (unit 0, pass 0)
    const uint32_t pb_off = (uint32_t)((rho0 + (lane_ >> 3)) * p.wrow + sc * EPC) * (uint32_t)sizeof(T);
    const uint32_t w32 = (uint32_t)(32 * p.wrow) * (uint32_t)sizeof(T);
    const uint32_t dstA = lds0 + wid_u * 8 * 128, dstB = lds0 + 256 * 128 + rho0 * 128;
    srd_t srdA, srdB;
    {
      const uint64_t ba = (uint64_t)(uintptr_t)in_b + (int64_t)pipe_min_toff * (int)sizeof(T);
      const uint64_t bb = (uint64_t)(uintptr_t)p.w + (uint64_t)n0 * p.wrow * sizeof(T);
      // the range check covers voffset + soffset from the (biased) base
      srdA = srd_t{(uint32_t)ba, (uint32_t)(ba >> 32) & 0xffffu, (uint32_t)p.in_bytes - (uint32_t)(pipe_min_toff * (int)sizeof(T)), 0x00020000u};
      srdB = srd_t{(uint32_t)bb, (uint32_t)(bb >> 32) & 0xffffu, 0xE0000000u, 0x00020000u};
    }
    uint32_t s_zero;   // opaque scalar zero: scalar offsets handed to the DMA are SALU results (a VALU-written SGPR needs 5 wait states before a VMEM reads it)
    asm volatile("s_mov_b32 %0, 0" : "=s"(s_zero));
// two DMA pieces (1 KB each, 8 tile rows) of one unit; M0 = LDS byte address of the piece (one wait state after the M0 write)
#define BLDS2(v0, v1, srd, so0, so1, d0, d1)                                                               \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, %5 offen lds\n\t"               \
               "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %6 offen lds"                   \
               ::"s"((uint32_t)(d0)), "s"((uint32_t)(d1)), "v"((uint32_t)(v0)), "v"((uint32_t)(v1)), "s"(srd), \
               "s"((uint32_t)(so0)), "s"((uint32_t)(so1)) : "memory")
#define ISSUE_A(u, ka, kbit, sel)                                                             \
  if (!(abl & 1)) {                                                                           \
    const uint32_t v0_ = (pa_mask[u][0] & (kbit)) ? pa_off[u][0] : OOB;                       \
    const uint32_t v1_ = (pa_mask[u][1] & (kbit)) ? pa_off[u][1] : OOB;                       \
    const uint32_t so_ = (ka) + s_zero;                                                       \
    BLDS2(v0_, v1_, srdA, so_, so_, dstA + (sel) + ((u) * 64) * 128, dstA + (sel) + (128 + (u) * 64) * 128); \
  }
#define ISSUE_B(u, kw, sel)                                                                   \
  if (!(abl & 2)) {                                                                           \
    BLDS2(pb_off, pb_off, srdB, (kw) + s_zero + (u) * w32, (kw) + (4 + (u)) * w32, dstB + (sel) + ((u) * 32) * 128, dstB + (sel) + (128 + (u) * 32) * 128); \
  }
#define WAIT_VM(rem)                                                            \
  do {                                                                          \
    const int r_ = (rem);                                                       \
    if (abl & 64) break;                                                        \
    if (r_ >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");               \
    else if (r_ == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          \
    else if (r_ == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
  } while (0)
#define BAR() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define LDSQ(ptr) (*reinterpret_cast<const uint4*>(ptr))
    STAMP(1);
    const int off0 = ((g) ^ sw) << 4, off1 = ((4 + g) ^ sw) << 4;
    const unsigned char* Ard = As + (wm * 128 + l15) * 128;              // A region: [wave row][128 rows]
    const unsigned char* Brd = As + 256 * 128 + (wn * 64 + l15) * 128;   // B region
    uint4 fa0[4][2], fa1[4][2], fbx[2][2], fby[2][2];   // A0 / A1 fragments; the two B halves swap roles every K-tile
    {   // prologue: both stages = units 0..7 = K-tiles 0 and 1
      const int4 k0 = s_kt[0];
      const uint32_t a0_ = RFL(k0.x), b0_ = RFL(k0.y), w0_ = RFL(k0.z);
      ISSUE_A(0, a0_, b0_, 0) ISSUE_B(0, w0_, 0) ISSUE_B(1, w0_, 0) ISSUE_A(1, a0_, b0_, 0)
      if (nk > 1) {
        const int4 k1p = s_kt[1];
        const uint32_t a1_ = RFL(k1p.x), b1_ = RFL(k1p.y), w1_ = RFL(k1p.z);
        ISSUE_A(0, a1_, b1_, STAGE) ISSUE_B(0, w1_, STAGE) ISSUE_B(1, w1_, STAGE) ISSUE_A(1, a1_, b1_, STAGE)
      }
      STAMP(2);
      WAIT_VM(NQ - 4);   // K-tile 0 has landed (phases 0 and 1 read its B1 / A1 before the loop's first barrier)
      BAR();
      STAMP(3);
#pragma unroll
      for (int j = 0; j < 4; ++j) { fa0[j][0] = LDSQ(Ard + off0 + j * 2048); fa0[j][1] = LDSQ(Ard + off1 + j * 2048); }
#pragma unroll
      for (int i = 0; i < 2; ++i) { fbx[i][0] = LDSQ(Brd + off0 + i * 2048); fbx[i][1] = LDSQ(Brd + off1 + i * 2048); }
#ifdef MMSKIN_ABLATE
      if (abl & 32) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { fa0[j][s] = make_uint4(0x3c003c00u + tid, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); fa1[j][s] = fa0[j][s]; }
#pragma unroll
          for (int i = 0; i < 2; ++i) { fbx[i][s] = make_uint4(0x3c003c00u, 0x3c003c00u + tid, 0x3c003c00u, 0x3c003c00u); fby[i][s] = fbx[i][s]; }
        }
      }
#endif
    }
    uint32_t ka2, kb2, kw2;   // K-tile t + 2: gather offset, tap bit, weight offset (scalars)
    { const int4 k2i = s_kt[nk > 2 ? 2 : 0]; ka2 = RFL(k2i.x); kb2 = RFL(k2i.y); kw2 = RFL(k2i.z); }
// the MFMAs of one k half of a quadrant; NRD fragment reads are spread between the first ones (INTERLEAVE), the unit's DMA follows
#define MMA_HALF(FA, FB, IB, NJ, JB, S)                                                  \
  if (!(abl & 4)) {                                                                      \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                        \
      _Pragma("unroll") for (int j = 0; j < (NJ); ++j) Mma<T>::run(FB[i][S], FA[j][S], acc[(IB) + i][(JB) + j]); \
  }
#define INTERLEAVE(NRD, NMF)                                                             \
  _Pragma("unroll") for (int q_ = 0; q_ < (NMF); ++q_) {                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
    if (q_ < (NRD)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                   \
  }
// Every second phase ends with the rendezvous: my DMA pieces up to unit g + 4 have landed (4 younger units stay in flight), my
// fragment reads have returned (so the regions they came from may be refilled after the barrier), then the barrier.
#define PHASE_SYNC(STEADY, REM)                                                          \
  if (STEADY) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else { WAIT_VM(REM); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } \
  BAR();
// One K-tile = four phases; phase g = 4t + ph issues unit g + 8 (the same region two K-tiles later) and reads unit g + 2:
//   ph0  mfma (A0, B0)   reads B1(t)   -> FBY      issues A0(t+2)
//   ph1  mfma (A0, B1)   reads A1(t)   -> fa1      issues B0(t+2)      | sync
//   ph2  mfma (A1, B1)   reads A0(t+1) -> fa0      issues B1(t+2)
//   ph3  mfma (A1, B0)   reads B0(t+1) -> FBY      issues A1(t+2)      | sync   (FBY is dead after ph2: the next K-tile swaps the roles)
// RAW: the reads of phases g + 1, g + 2 (units g + 3, g + 4) follow the sync of odd phase g, whose wait leaves only units > g + 4
// in flight.  WAR: unit g + 8 overwrites unit g, read in phase g - 2 and returned before the sync that ends phase g - 2 or g - 1.
// STEADY = 1: K-tile t + 2 exists (all four units are issued, four stay in flight).
#define PIPE_KTILE(STEADY, FBX, FBY)                                                                                    \
  {                                                                                                                     \
    const int sel = (t & 1) * STAGE, seln = STAGE - sel;                                                                \
    const int g4 = 4 * t;                                                                                               \
    const int4 k3v = s_kt[(STEADY) || t + 3 < nk ? t + 3 : nk - 1];                                                     \
    const unsigned char* A0p = Ard + sel + off0; const unsigned char* A1p = Ard + sel + off1;                           \
    const unsigned char* B0p = Brd + sel + off0; const unsigned char* B1p = Brd + sel + off1;                           \
    const unsigned char* A0n = Ard + seln + off0; const unsigned char* A1n = Ard + seln + off1;                         \
    const unsigned char* B0n = Brd + seln + off0; const unsigned char* B1n = Brd + seln + off1;                         \
    const bool more = (STEADY) || t + 1 < nk, more2 = (STEADY) || t + 2 < nk;                                           \
    /* ---- phase 0 */                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (!(abl & 32)) { _Pragma("unroll") for (int i = 0; i < 2; ++i) { FBY[i][0] = LDSQ(B0p + (2 + i) * 2048); FBY[i][1] = LDSQ(B1p + (2 + i) * 2048); } } \
    MMA_HALF(fa0, FBX, 0, 4, 0, 0)                                                                                      \
    INTERLEAVE(4, 8)                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (more2) { ISSUE_A(0, ka2, kb2, sel) }                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    MMA_HALF(fa0, FBX, 0, 4, 0, 1)                                                                                      \
    /* ---- phase 1 */                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    const uint32_t ka3 = RFL(k3v.x), kb3 = RFL(k3v.y), kw3 = RFL(k3v.z);                                                \
    if (!(abl & 32)) { _Pragma("unroll") for (int j = 0; j < FM - 4; ++j) { fa1[j][0] = LDSQ(A0p + (4 + j) * 2048); fa1[j][1] = LDSQ(A1p + (4 + j) * 2048); } } \
    MMA_HALF(fa0, FBY, 2, 4, 0, 0)                                                                                      \
    INTERLEAVE(2 * (FM - 4), 8)                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (more2) { ISSUE_B(0, kw2, sel) }                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    MMA_HALF(fa0, FBY, 2, 4, 0, 1)                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    PHASE_SYNC(STEADY, NQ - 6 - g4)                                                                                     \
    /* ---- phase 2 */                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (!(abl & 32) && more) { _Pragma("unroll") for (int j = 0; j < 4; ++j) { fa0[j][0] = LDSQ(A0n + j * 2048); fa0[j][1] = LDSQ(A1n + j * 2048); } } \
    MMA_HALF(fa1, FBY, 2, FM - 4, 4, 0)                                                                                 \
    INTERLEAVE(8, 2 * (FM - 4))                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (more2) { ISSUE_B(1, kw2, sel) }                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    MMA_HALF(fa1, FBY, 2, FM - 4, 4, 1)                                                                                 \
    /* ---- phase 3 */                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (!(abl & 32) && more) { _Pragma("unroll") for (int i = 0; i < 2; ++i) { FBY[i][0] = LDSQ(B0n + i * 2048); FBY[i][1] = LDSQ(B1n + i * 2048); } } \
    MMA_HALF(fa1, FBX, 0, FM - 4, 4, 0)                                                                                 \
    INTERLEAVE(4, 2 * (FM - 4))                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (more2) { ISSUE_A(1, ka2, kb2, sel) }                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    MMA_HALF(fa1, FBX, 0, FM - 4, 4, 1)                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    PHASE_SYNC(STEADY, NQ - 8 - g4)                                                                                     \
    ka2 = ka3; kb2 = kb3; kw2 = kw3;                                                                                    \
  }
    int t = 0;
    for (; t + 3 < nk; t += 2) { PIPE_KTILE(1, fbx, fby) ++t; PIPE_KTILE(1, fby, fbx) --t; }
    for (; t < nk; ++t) { PIPE_KTILE(0, fbx, fby) if (++t >= nk) break; PIPE_KTILE(0, fby, fbx) }
#undef PHASE_SYNC
    STAMP(4);
#undef PIPE_KTILE
#undef INTERLEAVE
#undef MMA_HALF
#undef LDSQ
#undef RFL
#undef BAR
#undef WAIT_VM
#undef ISSUE_B
#undef ISSUE_A
#undef BLDS2
  } else if constexpr (NST == 2) {
    if (nk > 0) LOAD_TILE(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my part of tile 0 has landed ...
    __syncthreads();                                   // ... and so has everybody else's
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) LOAD_TILE(cur ^ 1);   // DMA of the next tile runs under this tile's MFMAs
      COMPUTE_TILE(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed (it flew under the MFMAs above)
      __syncthreads();
    }
  } else {
    STAMP(1);
    for (int kt = 0; kt < nk; ++kt) {
      LOAD_TILE(0);
      if (kt == 0) STAMP(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt == 0) STAMP(3);
      COMPUTE_TILE(0);
      __syncthreads();   // every wave is done with the buffer before it is refilled / reused as C staging
    }
    STAMP(4);
  }
#undef COMPUTE_TILE
#undef LOAD_TILE
#undef GLDS16

  if (abl & 16) return;   // timing experiments only
  // ---- epilogue: acc -> LDS [pixel][channel] (packed) -> coalesced 16-byte stores
  // Strided dgrad classes scatter their rows over the full-resolution output: one thread per tile row works out its
  // destination row (two integer divisions) once, instead of every thread doing it for each of its 8 rows.
  const bool simple_rows = (p.OS == 1 && p.ncls == 1);
  int* s_orow = s_tap + 64;
  if (!simple_rows && tid < BM) {
    const int m = m0 + tid;
    const int mm = m < rows ? m : m0;
    const int img = mm / ab, rem = mm - img * ab;
    const int a = rem / b_dim, b = rem - a * b_dim;
    s_orow[tid] = (img * p.OHf + a * p.OS + p.cls[ci].ph) * p.OWf + b * p.OS + p.cls[ci].pw;
  } else if (p.addend_sub && tid < BM) {   // row of the compact (stride-2 sampled) addend, or -1: the output pixel is not one of its samples
    const int m = m0 + tid;
    const int mm = m < rows ? m : m0;
    const int hw = p.OHf * p.OWf, img = mm / hw, rem = mm - img * hw;
    const int y = rem / p.OWf, x = rem - y * p.OWf;
    s_orow[tid] = ((y | x) & 1) ? -1 : (img * ((p.OHf + 1) >> 1) + (y >> 1)) * ((p.OWf + 1) >> 1) + (x >> 1);
  }
  unsigned char* Cs = smem + TAP_LDS_BYTES;
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
  STAMP(5);

  constexpr int CH_PER_ROW = BN / EPC;
  constexpr int ROWS_PER_PASS = NT / CH_PER_ROW;
  const int cj = tid % CH_PER_ROW, r0 = tid / CH_PER_ROW;
  float ssum[EPC], ssq[EPC], ssb[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; ssb[e] = 0.f; }
  unsigned char* out_b = reinterpret_cast<unsigned char*>(p.out);
  // Epilogue profiles (compile-time null operands cost no registers; 168 VGPRs for the full set = 3 workgroups per CU):
  //   1 everything | 2 light: addend / bias / ReLU (folded-BN inference, accumulating dgrads; 120 VGPRs = 4 per CU)
  //   3 x (+ mask from x*scale+shift) and the BatchNorm-backward sums          (dgrads into a conv -> BN -> ReLU unit)
  //   4 addend + 1-bit ReLU mask + x and the sums                               (dgrads into a residual block output)
  //   5 = 4 + x2 and its sums                                                   (... whose block has a downsample BatchNorm too)
  //   6 = 2 + layer scale, dropout and an fp32 residual stream                  (Linear layers of the frozen transformer encoders)
  //   7 addend + 1-bit ReLU mask, sums of the masked values only (dgrads into a residual block output whose BatchNorm-backward
  //     x sums come from the weight-gradient GEMM instead: the algebraic path's second phase, abn.hip)
  constexpr bool HAS_ADD = EPI == 1 || EPI == 2 || EPI == 4 || EPI == 5 || EPI == 7, HAS_BR = EPI == 1 || EPI == 2 || EPI == 6, HAS_MY = EPI == 1;
  constexpr bool HAS_TR = EPI == 6;
  constexpr bool HAS_BIAS = HAS_BR || EPI == 3;   // profile 3 + bias: the algebraic BatchNorm-backward dgrad (constant row of the folded coefficients)
  constexpr bool HAS_MB = EPI == 1 || EPI == 4 || EPI == 5 || EPI == 7, HAS_X = EPI == 1 || (EPI >= 3 && EPI != 7), HAS_X2 = EPI == 1 || EPI == 5;
  constexpr bool HAS_MO = EPI == 2;   // light profile: optional mask byte out and per-channel multiplier (two-pass BatchNorm forward)
  const unsigned char* add_b = HAS_ADD ? reinterpret_cast<const unsigned char*>(p.addend) : nullptr;
  const unsigned char* my_b = HAS_MY ? reinterpret_cast<const unsigned char*>(p.ep_mask_y) : nullptr;
  const uint8_t* mb_b = HAS_MB ? p.ep_mask_bits : nullptr;
  const unsigned char* ex_b = HAS_X ? reinterpret_cast<const unsigned char*>(p.ep_x) : nullptr;
  const unsigned char* ex2_b = HAS_X2 ? reinterpret_cast<const unsigned char*>(p.ep_x2) : nullptr;
  const bool mask_from_x = HAS_X && p.ep_scale != nullptr;
  const bool has_bias = HAS_BIAS && p.ep_bias != nullptr;
  const bool do_relu = HAS_BR && p.ep_relu == 1;
  const bool do_gelu = HAS_BR && p.ep_relu == 2;          // exact (erf) GELU: the MLP of the transformer blocks
  float* out_f32 = HAS_BR ? p.out_f32 : nullptr;         // Linear layers at the fp32 op boundary: widen while storing
  const bool has_mul = HAS_MO && p.ep_mul != nullptr;
  float emul[EPC];
  if (has_mul) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) emul[e] = p.ep_mul[n0 + cj * EPC + e];
  }
  float ebias[EPC];
  if (has_bias) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) ebias[e] = p.ep_bias[n0 + cj * EPC + e];
  }
  const float* res_f32 = HAS_TR ? p.res_f32 : nullptr;
  const bool has_gamma = HAS_TR && p.ep_gamma != nullptr;
  const float drop_p = HAS_TR ? p.ep_drop_p : 0.f;
  const float drop_scale = drop_p < 1.f ? 1.f / (1.f - drop_p) : 0.f;
  float egam[EPC];
  if (has_gamma) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) egam[e] = p.ep_gamma[n0 + cj * EPC + e];
  }
  float esc[EPC], esh[EPC];
  if (mask_from_x) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) { esc[e] = p.ep_scale[n0 + cj * EPC + e]; esh[e] = p.ep_shift[n0 + cj * EPC + e]; }
  }
  // Rows are handled in groups of EG: all global reads of a group (addend, mask source, BN inputs) are
  // issued before any of them is consumed, so EG*3 16-byte loads are in flight per lane instead of one
  // dependent load->store chain per row (the fused epilogue was running at ~3 TB/s that way).
  constexpr int NR = BM / ROWS_PER_PASS;
#ifndef MMSKIN_EPI_ROWS4
#define MMSKIN_EPI_ROWS4 4
#endif
#ifndef MMSKIN_EPI_ROWS7
#define MMSKIN_EPI_ROWS7 4
#endif
  constexpr int EG_MAX = EPI == 6 ? 2 : (EPI == 4 ? MMSKIN_EPI_ROWS4 : (EPI == 7 ? MMSKIN_EPI_ROWS7 : 4));   // profile 6 holds 32 B of fp32 residual per row in flight: two rows keep it at the main loop's register count
  // the pipelined kernel has ONE workgroup per CU and nothing else to overlap its epilogue with, while its 112 - 128 accumulator
  // registers are dead by now: half of a thread's 14 / 16 rows in flight at a time (7 / 8 x up to 3 16-byte loads) instead of 2 / 4
  constexpr int EG = PIPE ? NR / 2 : (NR < EG_MAX ? NR : (NR % EG_MAX == 0 ? EG_MAX : 2));
  static_assert(NR % EG == 0, "row groups");
  const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_zero_page);
  for (int grp = 0; grp < NR; grp += EG) {
    u32x4_t q_ad[EG], q_my[EG], q_x[EG], q_x2[EG];
    float4 q_res[EG][EPC / 4];
    uint32_t q_mb[EG];
    size_t goffs[EG];
    bool valid[EG];
#pragma unroll
    for (int k = 0; k < EG; ++k) {
      const int row = r0 + (grp + k) * ROWS_PER_PASS;
      const int m = m0 + row;
      valid[k] = m < rows && (!PIPE || row < bm_step);
      const int orow = simple_rows ? (valid[k] ? m : m0) : s_orow[row];
      goffs[k] = ((size_t)orow * p.Cout + n0 + cj * EPC) * sizeof(T);
      if constexpr (HAS_ADD) {
        if (p.addend_sub) {
          const int ar = s_orow[row];
          q_ad[k] = *reinterpret_cast<const u32x4_t*>(add_b && valid[k] && ar >= 0 ? add_b + ((size_t)ar * p.Cout + n0 + cj * EPC) * sizeof(T) : zp);
        } else q_ad[k] = *reinterpret_cast<const u32x4_t*>(add_b && valid[k] ? add_b + goffs[k] : zp);
      }
      if constexpr (HAS_MY) q_my[k] = *reinterpret_cast<const u32x4_t*>(my_b && valid[k] ? my_b + goffs[k] : zp);
      if constexpr (HAS_MB) q_mb[k] = (mb_b && valid[k]) ? (uint32_t)mb_b[goffs[k] >> 4] : 0u;
      if constexpr (HAS_X) {
        // ep_x may be a channel prefix of a wider (concatenated) tensor: its rows are ep_x_pitch elements apart
        const size_t xoff = p.ep_x_pitch ? ((size_t)orow * p.ep_x_pitch + n0 + cj * EPC) * sizeof(T) : goffs[k];
        q_x[k] = *reinterpret_cast<const u32x4_t*>(ex_b && valid[k] ? ex_b + xoff : zp);
      }
      if constexpr (HAS_X2) q_x2[k] = *reinterpret_cast<const u32x4_t*>(ex2_b && valid[k] ? ex2_b + goffs[k] : zp);
      if constexpr (HAS_TR) {
        if (res_f32 && valid[k]) {
#pragma unroll
          for (int e = 0; e < EPC / 4; ++e) q_res[k][e] = *reinterpret_cast<const float4*>(res_f32 + goffs[k] / sizeof(T) + 4 * e);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < EG; ++k) {
      if (!valid[k]) continue;
      const int row = r0 + (grp + k) * ROWS_PER_PASS;
      Chunk<T> v;
      v.load(Cs + row * CPITCH + cj * 16);
      if constexpr (HAS_MO) {
        if (has_mul) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v.v[e] *= emul[e];
        }
      }
      if (add_b) {
        Chunk<T> ad;
        ad.from_raw(q_ad[k]);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] += ad.v[e];
      }
      if (has_bias) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] += ebias[e];
      }
      if (do_relu) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] = fmaxf(v.v[e], 0.f);
      }
      if (do_gelu) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] = gelu_erf_fast(v.v[e]);
      }
      if constexpr (HAS_TR) {
        if (drop_p > 0.f) {
          // keep bits first (one rolled loop: eight interleaved 64-bit hash chains cost a wave per SIMD in registers), then applied
          const uint64_t gi = p.ep_offset + goffs[k] / sizeof(T);
          uint32_t keep = 0;
#pragma unroll 1
          for (int e = 0; e < EPC; ++e) {
            const uint64_t h = mix64(p.ep_seed_mix ^ (gi + (uint64_t)e));
            keep |= ((float)(h >> 40) * (1.0f / 16777216.0f) >= drop_p ? 1u : 0u) << e;
          }
#pragma unroll
          for (int e = 0; e < EPC; ++e) v.v[e] = ((keep >> e) & 1u) ? v.v[e] * drop_scale : 0.f;
        }
        if (has_gamma) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v.v[e] *= egam[e];
        }
        if (res_f32) {
#pragma unroll
          for (int e = 0; e < EPC / 4; ++e) {
            v.v[4 * e] += q_res[k][e].x; v.v[4 * e + 1] += q_res[k][e].y; v.v[4 * e + 2] += q_res[k][e].z; v.v[4 * e + 3] += q_res[k][e].w;
          }
        }
      }
      if (my_b) {
        Chunk<T> my;
        my.from_raw(q_my[k]);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] = my.v[e] > 0.f ? v.v[e] : 0.f;
      }
      if (mb_b) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) v.v[e] = ((q_mb[k] >> e) & 1u) ? v.v[e] : 0.f;
      }
      if (ex_b) {
        Chunk<T> xv;
        xv.from_raw(q_x[k]);
        if (mask_from_x) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v.v[e] = (xv.v[e] * esc[e] + esh[e] > 0.f) ? v.v[e] : 0.f;
        }
        // statistics are taken on the value as stored (rounded to T), like the stand-alone reduce kernel
        Chunk<T> vr = v;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) vr.v[e] = bf16_bits_to_f32(f32_to_bf16_bits(v.v[e]));
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) { ssum[e] += vr.v[e]; ssq[e] += vr.v[e] * xv.v[e]; }
        if (ex2_b) {
          Chunk<T> x2v;
          x2v.from_raw(q_x2[k]);
#pragma unroll
          for (int e = 0; e < EPC; ++e) ssb[e] += vr.v[e] * x2v.v[e];
        }
      } else {
        if constexpr (EPI == 7 && sizeof(T) == 2) {   // sums of the masked gradient as stored (rounded), like the profiles that also read x
#pragma unroll
          for (int e = 0; e < EPC; ++e) ssum[e] += bf16_bits_to_f32(f32_to_bf16_bits(v.v[e]));
        } else {
#pragma unroll
          for (int e = 0; e < EPC; ++e) { ssum[e] += v.v[e]; ssq[e] += v.v[e] * v.v[e]; }
        }
      }
      if constexpr (HAS_MO) {
        if (p.ep_mask_out) {
          uint32_t bits = 0;
#pragma unroll
          for (int e = 0; e < EPC; ++e) bits |= (from_f32<T>(v.v[e]) != 0 && v.v[e] > 0.f ? 1u : 0u) << e;   // bit = (stored value > 0), as bn_apply writes it
          p.ep_mask_out[goffs[k] >> 4] = (uint8_t)bits;
        }
      }
      if (out_f32) {
        float* dst = out_f32 + goffs[k] / sizeof(T);
#pragma unroll
        for (int e = 0; e < EPC; e += 4) *reinterpret_cast<float4*>(dst + e) = make_float4(v.v[e], v.v[e + 1], v.v[e + 2], v.v[e + 3]);
      } else if (!(abl & 8) && out_b) v.store(out_b + goffs[k]);   // out == nullptr: statistics-only pass
    }
  }
  STAMP(6);
  if (p.stat_sum) {
    __syncthreads();                               // all rows of Cs consumed: the reduction buffer overlays it
    // [3][ROWS_PER_PASS][RPITCH], column e * CH_PER_ROW + cj for channel cj * EPC + e: a wave's lanes (cj fastest, then 2 - 8 row groups
    // RPITCH = BN + 8 floats apart) spread over all 32 banks.  With the channel-major column cj * EPC + e the 16 lanes of a row group hit
    // FOUR banks (16-way conflicts on 32 scalar stores per thread): 4 096 LDS cycles per tile, +14 us on the layer-3 conv3 forward.
    constexpr int RPITCH = BN + 8;
    float* red = reinterpret_cast<float*>(Cs);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      red[r0 * RPITCH + e * CH_PER_ROW + cj] = ssum[e];
      red[(ROWS_PER_PASS + r0) * RPITCH + e * CH_PER_ROW + cj] = ssq[e];
      if (p.stat_b_sq) red[(2 * ROWS_PER_PASS + r0) * RPITCH + e * CH_PER_ROW + cj] = ssb[e];
    }
    __syncthreads();
    if (tid < BN) {
      float s = 0.f, q = 0.f, b3 = 0.f;
#pragma unroll
      for (int r = 0; r < ROWS_PER_PASS; ++r) {
        s += red[r * RPITCH + tid];
        q += red[(ROWS_PER_PASS + r) * RPITCH + tid];
        if (p.stat_b_sq) b3 += red[(2 * ROWS_PER_PASS + r) * RPITCH + tid];
      }
      const int chan = (tid % CH_PER_ROW) * EPC + tid / CH_PER_ROW;   // column tid holds this channel of the tile
      const size_t o = (size_t)mblk * p.stat_stride + n0 + chan;
      p.stat_sum[o] = s;
      p.stat_sq[o] = q;
      if (p.stat_b_sq) { p.stat_b_sum[o] = s; p.stat_b_sq[o] = b3; }
    }
  }
  STAMP(7);
#undef STAMP
}

// ------------------------------------------------------------------------------------------ host
#ifdef MMSKIN_ABLATE   // timing-experiment library only
static unsigned long long* g_conv_stamps = nullptr;
extern "C" void mmskin_debug_set_conv_stamps(void* device_ptr) { g_conv_stamps = reinterpret_cast<unsigned long long*>(device_ptr); }
#endif
template <typename T, int BM, int BN, int WMv, int WNv, int EPI, int NST>
static int launch_cfg(const ConvGemmArgs& a, hipStream_t st) {
  constexpr int NT = 64 * WMv * WNv;
  constexpr int lds = conv_gemm_lds_bytes<BM, BN, T, NST, NT>();
  static bool attr_done = false;
  auto kern = conv_gemm_kernel<T, BM, BN, WMv, WNv, EPI, NST>;
  if (!attr_done) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done = true;
  }
  int grid = a.total_mblk * a.nblk_n;
  if (grid == 0) return MMSKIN_OK;
  // Linear GEMMs ({M, 1, 1, K, N} shapes) with a weight matrix larger than an XCD's L2: with column blocks fastest every row block
  // streamed the whole weight through L2 again (BEiT fc1: 1.2 GB of L2 fills per launch for 60 MB of operands,
  // profiles/r03_step_traffic_beitv2-large-bert-rgatt.txt); groups of 8 row blocks keep their A rows resident instead.
  static const int gm_env = [] { const char* v = getenv("MMSKIN_GEMM_GROUP_M"); return v ? atoi(v) : 8; }();
  ConvGemmArgs b = a;
  b.group_m = 0;
  if (gm_env > 1 && a.ncls == 1 && a.IH == 1 && a.IW == 1 && a.nblk_n >= 4 && (size_t)a.Cout * a.wrow * sizeof(T) > ((size_t)2 << 20) &&
      a.total_mblk >= 2 * gm_env)
    b.group_m = gm_env;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, b);
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

#define CONV_BM 128

static void finish_classes(ConvGemmArgs& a, int step = CONV_BM) {
  int blk = 0;
  for (int i = 0; i < a.ncls; ++i) {
    a.cls[i].rows = a.N * a.cls[i].a_dim * a.cls[i].b_dim;
    a.cls[i].mblk_start = blk;
    blk += ceil_div(a.cls[i].rows, step);
  }
  a.total_mblk = blk;
  a.bm_step = step;
}

// ---- selection of the 8-phase pipelined kernel (NST == 8).  MMSKIN_CONV_PIPE=0 switches it off, MMSKIN_CONV_PIPE_MINK (elements of
// the reduction) is the shortest K it takes, MMSKIN_CONV_PIPE_FORCE=1 takes every eligible launch (tests), MMSKIN_CONV_PIPE_TILE=
// 256 | 224 | 196 pins the tile rows instead of the tile-count model below.
static int64_t g_pipe_launches = 0;
extern "C" int64_t mmskin_conv_pipe_launches(void) { return g_pipe_launches; }
struct PipeChoice { int bm, step; };
// Selection of the pipelined kernel (NST == 8).  One 512-thread workgroup per CU: it pays where the reduction is deep enough to
// amortise a ~8 us prologue + epilogue that nothing overlaps (row-weighted K >= MMSKIN_CONV_PIPE_MINK, default 1024) and the launch
// has enough tiles to cover most of the chip (>= MMSKIN_CONV_PIPE_MINTILES, default 192); measured per ResNet-50 layer in
// profiles/r03_experiments.txt.  MMSKIN_CONV_PIPE=0 switches it off, MMSKIN_CONV_PIPE_FORCE=1 takes every eligible launch (tests),
// MMSKIN_CONV_PIPE_TILE = 256 | 224 | 196 pins the tile rows instead of the tile-count model below.
static bool pipe_choose(const ConvGemmArgs& a, bool tr, int prof, bool heavy, PipeChoice* out) {
  static const int on = [] { const char* v = getenv("MMSKIN_CONV_PIPE"); return v ? atoi(v) : 1; }();
  static const int force = [] { const char* v = getenv("MMSKIN_CONV_PIPE_FORCE"); return v ? atoi(v) : 0; }();
  static const int mink = [] { const char* v = getenv("MMSKIN_CONV_PIPE_MINK"); return v ? atoi(v) : 1024; }();
  static const int mintiles = [] { const char* v = getenv("MMSKIN_CONV_PIPE_MINTILES"); return v ? atoi(v) : 192; }();
  static const int pin = [] { const char* v = getenv("MMSKIN_CONV_PIPE_TILE"); return v ? atoi(v) : 0; }();
  if (!on || !a.pipe_ok || a.in2 || a.addend_sub || prof == 7 || a.ep_mask_out || a.ep_mul || !a.out || a.in_bytes == 0 || a.in_bytes > 0xE0000000ull || (heavy && prof == 1) || a.Cout % 256 != 0 ||
      a.C % 64 != 0 || a.ncls < 1)
    return false;
  if (tr && (heavy || a.addend || a.stat_sum || !a.out_f32)) return false;   // the transformer-residual epilogue's own contract (checked below)
  double ksum = 0, rsum = 0;
  for (int i = 0; i < a.ncls; ++i) {
    const int nk = a.cls[i].ntaps * a.C / 64;
    if (nk < 1 || nk > KT_LDS_BYTES / 16) return false;
    ksum += (double)a.cls[i].rows * nk * 64; rsum += a.cls[i].rows;
  }
  if (rsum <= 0) return false;
  // the launch takes ceil(tiles / 256) rounds of a tile whose MFMA time goes with its COMPUTED rows
  const int cand[3][2] = {{256, 256}, {224, 224}, {224, 196}};
  double best = 1e30;
  long best_tiles = 0;
  for (int c = 0; c < 3; ++c) {
    if (pin && cand[c][1] != pin) continue;
    long tiles = 0;
    for (int i = 0; i < a.ncls; ++i) tiles += (long)ceil_div(a.cls[i].rows, cand[c][1]) * (a.Cout / 256);
    const double cost = (double)((tiles + 255) / 256) * cand[c][0];
    if (cost < best) { best = cost; best_tiles = tiles; out->bm = cand[c][0]; out->step = cand[c][1]; }
  }
  if (best >= 1e30) return false;
  // Linear GEMMs ({M, 1, 1, K, N}: no gather, one tap) already pay at K = 768 (BERT-base qkv / fc1 / output.dense: config 5 50.7 -> 50.3 ms);
  // the ResNet convolutions do not below 1024 (profiles/r04_experiments.txt (0))
  const double mink_eff = (a.IH == 1 && a.IW == 1 && a.ncls == 1 && mink == 1024) ? 768 : mink;
  return force || (ksum / rsum >= mink_eff && best_tiles >= mintiles);
}

template <typename T>
static int dispatch_conv_gemm(ConvGemmArgs& a, hipStream_t st) {
  ARG_CHECK(a.Cout % 64 == 0, "conv_gemm: Cout=%d must be a multiple of 64", a.Cout);
  ARG_CHECK((a.C % DT<T>::EPC) == 0, "conv_gemm: C=%d not a multiple of %d", a.C, DT<T>::EPC);
  for (int i = 0; i < a.ncls; ++i)
    ARG_CHECK((a.cls[i].ntaps * a.C) % DT<T>::BK == 0, "conv_gemm: K=%d not a multiple of %d",
              a.cls[i].ntaps * a.C, DT<T>::BK);
  a.ablate = 0;
#ifdef MMSKIN_ABLATE
  { const char* abl = getenv("MMSKIN_CONV_ABLATE"); a.ablate = abl ? atoi(abl) : 0; }
  a.stamps = g_conv_stamps;
#endif
  const bool tr = a.ep_gamma || a.res_f32 || a.ep_drop_p > 0.f;   // transformer-residual epilogue
  const bool epi = a.addend || a.ep_mask_y || a.ep_mask_bits || a.ep_x || a.ep_bias || a.ep_relu || a.out_f32 || a.ep_mul || a.ep_mask_out || tr;
  // Single-buffer variant (3-4 workgroups per CU) whenever the launch has enough workgroups to use the
  // extra residency: measured on the ResNet-50 shape mix (scripts/conv_mix.py) it wins for every layer
  // with more than ~2.5 workgroups per CU and loses for the 392-workgroup layer-4 launches.
  static const int nst1_min_blocks = [] { const char* v = getenv("MMSKIN_CONV_NST1_MINBLOCKS"); return v ? atoi(v) : 640; }();
  static const int nst1_min_blocks_epi = [] { const char* v = getenv("MMSKIN_CONV_NST1_MINBLOCKS_EPI"); return v ? atoi(v) : 800; }();   // fused-epilogue launches: 168 VGPRs = 3 workgroups/CU = 768 single-buffer slots; a 784-workgroup launch would spill into a second round
  const int bn_sel = (a.Cout % 128 == 0) ? 128 : 64;
  const bool heavy = a.ep_mask_y || a.ep_mask_bits || a.ep_x;   // needs a fused BatchNorm-backward epilogue
  // 256 x 256 tile, 8 waves (one workgroup per CU, half the L2 -> LDS bytes per FLOP): plain GEMM-like launches (one tap class, no
  // statistics, light epilogue) with a deep reduction whose grid still covers the chip -- fc2 / output.dense of the transformer
  // encoders (K = 3072 / 4096: 1.03 - 1.05 PFLOP/s against 0.88 - 0.92 on the 128 x 128 tile; at K <= 1024 the per-tile prologue and
  // the 2-wave-per-SIMD epilogue cost more than the fill saves: profiles/r02_experiments.txt (9)).  MMSKIN_GEMM_BIG_MINK=0: off.
  static const int big_min_k = [] { const char* v = getenv("MMSKIN_GEMM_BIG_MINK"); return v ? atoi(v) : 2048; }();
  static const bool profiles_on = [] { const char* v = getenv("MMSKIN_CONV_EPI_PROFILES"); return !v || atoi(v) != 0; }();
  int prof = 1;
  if (profiles_on && heavy && !a.ep_mask_y && !a.ep_relu) {
    if (a.ep_x && !a.addend && !a.ep_mask_bits && !a.ep_x2) prof = 3;   // (+ optional bias)
    else if (!a.ep_bias && a.ep_x && a.ep_mask_bits && !a.ep_scale) prof = a.ep_x2 ? 5 : 4;
    else if (!a.ep_bias && !a.ep_x && !a.ep_x2 && a.ep_mask_bits && !a.ep_scale) prof = 7;
  }
  if constexpr (sizeof(T) == 2) {
    PipeChoice pc;
    if (pipe_choose(a, tr, prof, heavy, &pc)) {
      finish_classes(a, pc.step);
      a.nblk_n = a.Cout / 256;
      ++g_pipe_launches;
      const int e = tr ? 6 : (!epi ? 0 : (!heavy ? 2 : prof));
#define GOP(BMv) (e == 0 ? launch_cfg<T, BMv, 256, 2, 4, 0, 8>(a, st) : e == 2 ? launch_cfg<T, BMv, 256, 2, 4, 2, 8>(a, st) : \
                  e == 6 ? launch_cfg<T, BMv, 256, 2, 4, 6, 8>(a, st) : \
                  e == 3 ? launch_cfg<T, BMv, 256, 2, 4, 3, 8>(a, st) : e == 4 ? launch_cfg<T, BMv, 256, 2, 4, 4, 8>(a, st) : launch_cfg<T, BMv, 256, 2, 4, 5, 8>(a, st))
      return pc.bm == 256 ? GOP(256) : GOP(224);
#undef GOP
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (big_min_k > 0 && a.out && !a.ep_mask_out && !a.ep_mul && a.wrow >= big_min_k && a.ncls == 1 && !heavy && !a.stat_sum && !a.addend && a.Cout % 256 == 0 && a.ep_relu != 2 &&
        a.cls[0].mblk_start == 0 && a.cls[0].ntaps == 1 && a.Sy == 1 && a.Sx == 1 && a.OS == 1) {   // plain GEMMs only (what the tests cover)
      const int mb = ceil_div(a.cls[0].rows, 256), nb = a.Cout / 256;
      if (mb * nb >= 224) {
        a.total_mblk = mb; a.nblk_n = nb;
        return !epi ? launch_cfg<T, 256, 256, 2, 4, 0, 2>(a, st) : (tr ? launch_cfg<T, 256, 256, 2, 4, 6, 2>(a, st) : launch_cfg<T, 256, 256, 2, 4, 2, 2>(a, st));
      }
    }
  }
  // only the everything-profile is a 168-VGPR kernel (768 single-buffer slots); the others have the forward kernel's residency
  const bool one = a.total_mblk * (a.Cout / bn_sel) > ((heavy && prof == 1) ? nst1_min_blocks_epi : nst1_min_blocks);
#define GO(BNv, E) (one ? launch_cfg<T, CONV_BM, BNv, 2, 2, E, 1>(a, st) : launch_cfg<T, CONV_BM, BNv, 2, 2, E, 2>(a, st))
  if (tr) {
    ARG_CHECK(!heavy && !a.addend && !a.stat_sum && a.out_f32 && a.Cout % 128 == 0,
              "conv_gemm: the transformer-residual epilogue needs an fp32 output, Cout %% 128 == 0 and no other fused operand");
    if constexpr (sizeof(T) == 2) { a.nblk_n = a.Cout / 128; return GO(128, 6); }
    ARG_CHECK(false, "conv_gemm: the transformer-residual epilogue is a bf16-operand feature");
  }
  if (a.Cout % 128 == 0) {
    a.nblk_n = a.Cout / 128;
    return !epi ? GO(128, 0) : (!heavy ? GO(128, 2) : (prof == 3 ? GO(128, 3) : (prof == 4 ? GO(128, 4) : (prof == 5 ? GO(128, 5) : (prof == 7 ? GO(128, 7) : GO(128, 1))))));
  }
  a.nblk_n = a.Cout / 64;
  return !epi ? GO(64, 0) : (!heavy ? GO(64, 2) : (prof == 3 ? GO(64, 3) : (prof == 4 ? GO(64, 4) : (prof == 5 ? GO(64, 5) : (prof == 7 ? GO(64, 7) : GO(64, 1))))));
#undef GO
}


int conv_fwd_stat_rows(const ConvShape& s) { return ceil_div(s.N * s.OH() * s.OW(), CONV_BM); }
int stem_conv_stat_rows(int N, int OH, int OW) { return ceil_div(N * OH * OW, CONV_BM); }

template <typename T>
int launch_conv_fwd(const ConvShape& s, const T* in, const T* w_staged, T* out, float* stat_sum,
                    float* stat_sq, hipStream_t st, const FwdFuse* fuse, int* stat_rows_out) {
  ARG_CHECK(s.kh * s.kw <= MMSKIN_MAX_TAPS, "conv_fwd: %dx%d kernel has too many taps", s.kh, s.kw);
  if constexpr (sizeof(T) == 2) {
    if (!fuse && (!stat_sum || stat_rows_out) && conv3x3_c64_takes(s, true)) {
      if (stat_rows_out) *stat_rows_out = s.N;
      return launch_conv3x3_c64_fwd(s, in, w_staged, out, stat_sum, stat_sq, s.Cout, st);
    }
  }
  ConvGemmArgs a = {};
  a.in = in; a.w = w_staged; a.out = out; a.addend = nullptr;
  if (fuse) {
    a.ep_bias = fuse->bias; a.addend = fuse->addend; a.ep_relu = fuse->gelu ? 2 : (fuse->relu ? 1 : 0); a.out_f32 = fuse->out_f32;
    a.ep_gamma = fuse->gamma; a.res_f32 = fuse->res_f32; a.ep_drop_p = fuse->drop_p; a.ep_seed_mix = mix64(fuse->seed); a.ep_offset = fuse->offset;
    a.ep_mask_out = fuse->mask_out; a.ep_mul = fuse->mul;
  }
  a.stat_sum = stat_sum; a.stat_sq = stat_sq; a.stat_stride = s.Cout;
  a.N = s.N; a.IH = s.H; a.IW = s.W; a.C = s.Cin; a.Cpitch = s.Cin;
  a.in_bytes = (uint64_t)s.N * s.H * s.W * s.Cin * sizeof(T);
  a.Cout = s.Cout; a.wrow = s.kh * s.kw * s.Cin;
  a.Sy = s.stride; a.Sx = s.stride; a.OS = 1;
  a.OHf = s.OH(); a.OWf = s.OW();
  a.ncls = 1;
  TapClass& c = a.cls[0];
  c.a_dim = s.OH(); c.b_dim = s.OW(); c.ph = 0; c.pw = 0; c.ntaps = s.kh * s.kw;
  for (int r = 0; r < s.kh; ++r)
    for (int q = 0; q < s.kw; ++q) {
      int t = r * s.kw + q;
      c.offy[t] = (int8_t)(r - s.pad); c.offx[t] = (int8_t)(q - s.pad); c.wtap[t] = (int8_t)t;
    }
  finish_classes(a);
  a.simple_src = (s.kh == 1 && s.kw == 1 && s.stride == 1 && s.pad == 0) ? 1 : 0;
  a.pipe_ok = (!stat_sum || stat_rows_out) ? 1 : 0;
  const int rc = dispatch_conv_gemm<T>(a, st);
  if (stat_rows_out) *stat_rows_out = a.total_mblk;
  return rc;
}

int conv_dgrad_partial_rows(const ConvShape& s) {
  // every parity class rounds its row count up to a whole row block
  return ceil_div(s.N * s.H * s.W, CONV_BM) + s.stride * s.stride;
}

// Tap-less parity classes of a strided dgrad (1x1 / stride 2: three of four input pixels receive no gradient) are plain
// zero fills.  As GEMM workgroups they cost a full prologue + row-index epilogue each (layer2's downsample dgrad: 9 408 of
// 12 544 workgroups, 104 us of launch overhead with every load, MFMA and store ablated); here 16 B per lane, row-contiguous.
__global__ __launch_bounds__(256) void parity_zero_fill_kernel(uint4* __restrict__ out, int H, int W, int chunks_per_pixel, int S,
                                                               unsigned zero_mask) {
  const int y = blockIdx.y, n = blockIdx.z;
  const int t = blockIdx.x * 256 + threadIdx.x;          // over W * chunks_per_pixel
  if (t >= W * chunks_per_pixel) return;
  const int x = t / chunks_per_pixel;
  if (!((zero_mask >> ((y % S) * S + (x % S))) & 1u)) return;
  out[((size_t)(n * H + y) * W) * chunks_per_pixel + t] = make_uint4(0u, 0u, 0u, 0u);
}

template <typename T>
int launch_conv_dgrad(const ConvShape& s, const T* dout, const T* wt_staged, T* din, const T* addend,
                      hipStream_t st, DgradFuse* fuse) {
  ARG_CHECK(s.kh * s.kw <= MMSKIN_MAX_TAPS, "conv_dgrad: too many taps");
  ARG_CHECK(s.stride == 1 || s.stride == 2, "conv_dgrad: stride %d unsupported", s.stride);
  if constexpr (sizeof(T) == 2) {
    const bool prof3 = fuse && fuse->x && fuse->scale && fuse->shift && !fuse->mask_y && !fuse->mask_bits && !fuse->x2 && fuse->x_pitch == 0 && fuse->partial;
    if (!addend && (!fuse || prof3) && conv3x3_c64_takes(s, true))
      return launch_conv3x3_c64_dgrad(s, dout, wt_staged, din, fuse, st);
  }
  ConvGemmArgs a = {};
  a.in = dout; a.w = wt_staged; a.out = din; a.addend = addend;
  if (fuse) {
    ARG_CHECK(!(addend == din && addend != nullptr && s.stride != 1 && s.kh == 1),
              "conv_dgrad: epilogue fusion needs a launch that covers every output pixel");
    a.ep_mask_y = fuse->mask_y; a.ep_mask_bits = fuse->mask_bits; a.ep_x = fuse->x; a.ep_x_pitch = fuse->x_pitch; a.ep_scale = fuse->scale; a.ep_shift = fuse->shift;
    a.ep_x2 = fuse->x2;
    if (fuse->addend_s2) {
      ARG_CHECK(addend && addend != din && s.kh == 1 && s.kw == 1 && s.stride == 1 && s.pad == 0, "conv_dgrad: a compact stride-2 addend needs a 1x1 / stride 1 launch and its own buffer");
      a.addend_sub = 2;
    }
    a.stat_stride = 2 * s.Cin;
    a.stat_sum = fuse->partial; a.stat_sq = fuse->partial + s.Cin;
    if (fuse->x2) { a.stat_b_sum = fuse->partial_b; a.stat_b_sq = fuse->partial_b + s.Cin; }
  }
  a.N = s.N; a.IH = s.OH(); a.IW = s.OW(); a.C = s.Cout; a.Cpitch = s.Cout;
  a.in_bytes = (uint64_t)s.N * s.OH() * s.OW() * s.Cout * sizeof(T);
  a.Cout = s.Cin; a.wrow = s.kh * s.kw * s.Cout;
  if (fuse && fuse->in2) {
    int sh = 0;
    while ((fuse->k2 << sh) < s.Cout) ++sh;
    ARG_CHECK(s.kh == 1 && s.kw == 1 && s.stride == 1 && s.pad == 0 && fuse->k2 > 0 && (fuse->k2 << sh) == s.Cout && fuse->k2 % DT<T>::BK == 0 &&
              s.Cout % DT<T>::BK == 0, "conv_dgrad: the second operand needs a 1x1 / stride 1 layer and k2 * 2^j == Cout (k2=%d Cout=%d)", fuse->k2, s.Cout);
    a.in2 = fuse->in2; a.K1 = s.Cout; a.pitch2_shift = sh;
    a.C = s.Cout + fuse->k2; a.wrow = a.C;
  }
  if (fuse && fuse->bias) a.ep_bias = fuse->bias;
  a.Sy = 1; a.Sx = 1; a.OS = s.stride;
  a.OHf = s.H; a.OWf = s.W;
  a.ncls = 0;
  unsigned zero_mask = 0;   // parity classes (ph * stride + pw) without taps that only need zeros
  for (int ph = 0; ph < s.stride; ++ph)
    for (int pw = 0; pw < s.stride; ++pw) {
      TapClass c = {};
      c.a_dim = (s.H - ph + s.stride - 1) / s.stride;
      c.b_dim = (s.W - pw + s.stride - 1) / s.stride;
      c.ph = ph; c.pw = pw; c.ntaps = 0;
      for (int r = 0; r < s.kh; ++r) {
        if ((ph + s.pad - r) % s.stride != 0) continue;
        for (int q = 0; q < s.kw; ++q) {
          if ((pw + s.pad - q) % s.stride != 0) continue;
          int t = c.ntaps++;
          c.offy[t] = (int8_t)((ph + s.pad - r) / s.stride);
          c.offx[t] = (int8_t)((pw + s.pad - q) / s.stride);
          c.wtap[t] = (int8_t)(r * s.kw + q);
        }
      }
      if (c.a_dim <= 0 || c.b_dim <= 0) continue;
      // a tap-less class contributes zeros: skip it when accumulating in place
      if (c.ntaps == 0 && addend == din && addend != nullptr) continue;
      static const bool zf = [] { const char* v = getenv("MMSKIN_DGRAD_ZEROFILL"); return !v || atoi(v) != 0; }();
      if (zf && c.ntaps == 0 && addend == nullptr && !fuse) { zero_mask |= 1u << (ph * s.stride + pw); continue; }   // plain zero fill
      a.cls[a.ncls++] = c;
    }
  if (zero_mask) {
    const int cpp = s.Cin * (int)sizeof(T) / 16;
    ARG_CHECK(s.H <= 65535 && s.N <= 65535 && (s.Cin * sizeof(T)) % 16 == 0, "conv_dgrad: zero fill grid");
    hipLaunchKernelGGL(parity_zero_fill_kernel, dim3(ceil_div(s.W * cpp, 256), s.H, s.N), dim3(256), 0, st,
                       reinterpret_cast<uint4*>(din), s.H, s.W, cpp, s.stride, zero_mask);
    HIP_CHECK_RET(hipGetLastError());
  }
  // Parity classes from the heaviest down (3x3 / stride 2 / pad 1: 4, 2, 2, 1 taps; 1x1 / stride 2: 1, 0, 0, 0): the long-K
  // workgroups start first and the short ones fill the tail of the launch instead of the other way round.
  for (int i = 1; i < a.ncls; ++i)
    for (int j = i; j > 0 && a.cls[j].ntaps > a.cls[j - 1].ntaps; --j) { TapClass t = a.cls[j]; a.cls[j] = a.cls[j - 1]; a.cls[j - 1] = t; }
  finish_classes(a);
  a.simple_src = (s.kh == 1 && s.kw == 1 && s.stride == 1 && s.pad == 0 && a.ncls == 1) ? 1 : 0;
  a.pipe_ok = 1;
  const int rc = dispatch_conv_gemm<T>(a, st);
  if (fuse) fuse->rows_written = a.total_mblk;
  return rc;
}

template <typename T>
int launch_stem_conv_fwd(int N, int OH, int OW, int Hp, int Wp, const T* img4, const T* wv, T* out,
                         float* stat_sum, float* stat_sq, hipStream_t st, int* stat_rows_out) {
  if constexpr (sizeof(T) == 2) {
    if (stem7x7_takes(OH, OW, Hp, Wp)) {   // direct convolution from an LDS-resident window (stem7x7.hip)
      if (stat_rows_out) *stat_rows_out = stem7x7_stat_rows(N, OH);
      return launch_stem7x7_fwd(N, OH, OW, Hp, Wp, img4, wv, out, stat_sum, stat_sq, st);
    }
  }
  if (stat_rows_out) *stat_rows_out = stem_conv_stat_rows(N, OH, OW);
  // virtual conv: macro pixel = 2 real pixels x 4 channels = 8 elements; one tap per kernel row r,
  // each reading 32 contiguous elements (8 real pixels x 4 ch) starting at macro pixel wo.
  ConvGemmArgs a = {};
  a.in = img4; a.w = wv; a.out = out; a.addend = nullptr;
  a.stat_sum = stat_sum; a.stat_sq = stat_sq; a.stat_stride = 64;
  a.N = N; a.IH = Hp; a.IW = Wp / 2; a.C = 32; a.Cpitch = 8;
  a.Cout = 64; a.wrow = 8 * 32;
  a.Sy = 2; a.Sx = 1; a.OS = 1;
  a.OHf = OH; a.OWf = OW;
  a.ncls = 1;
  TapClass& c = a.cls[0];
  c.a_dim = OH; c.b_dim = OW; c.ph = 0; c.pw = 0; c.ntaps = 8;
  for (int r = 0; r < 8; ++r) { c.offy[r] = (int8_t)r; c.offx[r] = 0; c.wtap[r] = (int8_t)r; }
  finish_classes(a);
  ARG_CHECK(2 * (OH - 1) + 7 < Hp && (OW - 1) + 3 < Wp / 2, "stem conv: padded image too small");
  return dispatch_conv_gemm<T>(a, st);
}

template <typename T>
int launch_vgg_first_conv_fwd(int N, int H, int W, int Hp, int Wp, const T* img8, const T* wv, T* out,
                              const FwdFuse* fuse, hipStream_t st, int stride, float* stat_sum, float* stat_sq) {
  // H, W: input image size; output = (H + 2 - 3) / stride + 1
  ARG_CHECK(Hp >= H + 2 && Wp >= W + 4 && (stride == 1 || stride == 2), "first 3x3 conv: padded image too small / stride");
  const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
  ConvGemmArgs a = {};
  a.in = img8; a.w = wv; a.out = out;
  a.stat_sum = stat_sum; a.stat_sq = stat_sq; a.stat_stride = 64;
  if (fuse) { a.ep_bias = fuse->bias; a.addend = fuse->addend; a.ep_relu = fuse->relu ? 1 : 0; }
  a.N = N; a.IH = Hp; a.IW = Wp; a.C = 32; a.Cpitch = 8;
  a.Cout = 64; a.wrow = 4 * 32;
  a.Sy = stride; a.Sx = stride; a.OS = 1;
  a.OHf = OH; a.OWf = OW;
  a.ncls = 1;
  TapClass& c = a.cls[0];
  c.a_dim = OH; c.b_dim = OW; c.ph = 0; c.pw = 0; c.ntaps = 4;
  for (int r = 0; r < 4; ++r) { c.offy[r] = (int8_t)(r < 3 ? r : 0); c.offx[r] = 0; c.wtap[r] = (int8_t)r; }
  finish_classes(a);
  return dispatch_conv_gemm<T>(a, st);
}

#define INST(T)                                                                                      \
  template int launch_vgg_first_conv_fwd<T>(int, int, int, int, int, const T*, const T*, T*, const FwdFuse*, hipStream_t, int, float*, float*); \
  template int launch_conv_fwd<T>(const ConvShape&, const T*, const T*, T*, float*, float*, hipStream_t, const FwdFuse*, int*); \
  template int launch_conv_dgrad<T>(const ConvShape&, const T*, const T*, T*, const T*, hipStream_t, DgradFuse*);      \
  template int launch_stem_conv_fwd<T>(int, int, int, int, int, const T*, const T*, T*, float*, float*, hipStream_t, int*);
INST(float)
INST(bf16_t)

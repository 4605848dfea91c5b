// Fused softmax attention for gfx950:  O = dropout(softmax(Q K^T * scale + bias + key_mask)) V   in ONE kernel.
//
// Replaces, in bf16-operand mode, the chain the transformer encoders of the reference reach through timm / transformers
// (loadImageModelClassifier.py:117-121 `timm.create_model`, :170-181 `AutoModel.from_pretrained`; consumed at
// multimodalIntraInterModal.py:167,180-183): QK^T GEMM -> row softmax (+ relative-position bias / key padding mask /
// causal mask) -> dropout -> PV GEMM, which as separate fp32 launches materialised the [B, H, L, L] scores three times
// (47 % of a BEiT-large + BERT step, profiles/r02_*).  Here the scores never leave the chip:
//
//   * one 256-thread workgroup = 64 query rows of one (batch, head); wave w owns rows 16w..16w+15
//   * K / V are walked in 64-key tiles: fp32 global -> registers (the next tile's loads are in flight while this tile is
//     multiplied) -> bf16 -> LDS (row pitch D*2 + 16 B: conflict-free for both the b128 row reads of K and the
//     transposing reads of V)
//   * S = Q K^T on v_mfma_f32_16x16x32_bf16 (Q fragments live in registers for the whole kernel, pre-multiplied by
//     `scale`); online softmax in fp32 registers (running max / sum per query row, 16-lane xor-shuffle reductions);
//     P -> bf16 -> a wave-private LDS tile (the accumulator layout holds a key per lane, the A operand needs a query per
//     lane) -> O += P V on the same MFMA, V fragments by ds_read_b64_tr_b16
//   * dropout uses the library's counter-based generator on the element index of the [B, H, L, L] probability tensor, so
//     a fused launch drops exactly the elements the unfused path would
//   * q / k / v / o are addressed through element strides: the [B, L, 3, H, Dh] output of a fused qkv Linear is read in
//     place and O can be written token-major, without permute copies
// fp32 or bf16 tensors at the boundary (io_dtype; fp32 inputs are rounded to bf16 on arrival), fp32 accumulation and softmax; LSE is
// written for a backward pass.
#include "../../include/mmskin.h"
#include <stdlib.h>

#include "common.h"

namespace {

struct FlashArgs {
  const void* q; const void* k; const void* v;   // IO = float or bf16_t (template parameter of the kernel)
  void* o; float* lse;
  const float* mask_add;   // [B][L] additive key mask or null
  const float* bias;       // [H][L][L] additive score bias or null
  int B, H, L;
  int64_t q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, o_sl;   // element strides (last dim contiguous)
  float scale, drop_p;
  int causal;
  uint64_t seed, offset;
  int xcd;      // 1: XCD-aware tile order (MMSKIN_FLASH_XCD)
  int ablate;   // -DMMSKIN_ABLATE builds only (scripts/flash_ablate.py): bit0 K / V global loads after tile 0, bit1 softmax VALU work, bit2 MFMAs, bit3 K / V LDS stores
};

__device__ __forceinline__ uint2 lds_tr16_b64(const unsigned char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) { return f32_to_bf16_bits(a) | (f32_to_bf16_bits(b) << 16); }

// 8 consecutive elements of one row as a packed bf16x8 MFMA fragment / LDS chunk
__device__ __forceinline__ uint4 load8_bf16(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
  return make_uint4(pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(c.x, c.y), pack_bf16(c.z, c.w));
}
__device__ __forceinline__ uint4 load8_bf16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

// NB = LDS buffers for the K / V tiles (1: two barriers per key tile; 2: one); IO = tensor dtype; DROP: attention-probability dropout compiled in
// (as a run-time test the generator's code kept 24 - 28 more VGPRs live in the launches that never drop: 3 -> 2 waves per SIMD at Dh = 64)
template <int D, int NB, typename IO, bool DROP>
__global__ __launch_bounds__(256) void flash_fwd_kernel(const FlashArgs p) {
  constexpr int BQ = 64, BK = 64;
  constexpr int PITCH = D * 2 + 16;          // bytes per K / V row in LDS
  constexpr int PPITCH = BK * 2 + 16;        // bytes per P row
  constexpr int KS = D / 32;                 // k-steps of the QK^T product
  constexpr int DN = D / 16;                 // 16-wide output column tiles
  constexpr int CPT = BK * (D / 8) / 256;    // 8-element chunks per thread per tile (K and V each)
  static_assert(D == 32 || D == 64, "head dim");
  constexpr int KV_BYTES = 2 * BK * PITCH;   // one K tile + one V tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[NB * KV_BYTES + 4 * 16 * PPITCH];
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  unsigned char* Ps = smem + NB * KV_BYTES + wid * 16 * PPITCH;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (own L2 each), so the query tiles of one (batch, head) -- which
  // all stream the same K / V -- landed on up to 8 XCDs and K / V were fetched once per XCD (BERT shape: 0.9 GB of L2 fills per launch for
  // 0.3 GB of q, k, v; profiles/r03_step_traffic_beitv2-large-bert-rgatt.txt).  xcd_remap gives an XCD a contiguous run of tiles.
  const int lin0 = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  const int lin = p.xcd ? xcd_remap(lin0, (int)(gridDim.x * gridDim.y)) : lin0;
  const int bh = lin / (int)gridDim.x, b = bh / p.H, h = bh - b * p.H;
  const int q0 = (lin - bh * (int)gridDim.x) * BQ;
  const int L = p.L;
#ifdef MMSKIN_ABLATE
  const int abl = p.ablate;
#else
  constexpr int abl = 0;
#endif

  // ---- Q fragments: lane (row l15 of this wave's 16, d = 32 ks + 8 g .. +7) as bf16; `scale` is applied to S in fp32
  uint4 qf[KS];
  {
    const int qi = q0 + wid * 16 + l15;
    const IO* qp = reinterpret_cast<const IO*>(p.q) + b * p.q_sb + h * p.q_sh + (int64_t)(qi < L ? qi : L - 1) * p.q_sl;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = load8_bf16(qp + 32 * ks + 8 * g);
  }
  // rows this lane owns in the accumulator layout: query 4 g + r of the wave's 16
  float m_run[4], l_run[4];
  f32x4_t oacc[DN];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_run[r] = -3.4028235e38f; l_run[r] = 0.f; }
#pragma unroll
  for (int i = 0; i < DN; ++i) oacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const IO* kbase = reinterpret_cast<const IO*>(p.k) + b * p.k_sb + h * p.k_sh;
  const IO* vbase = reinterpret_cast<const IO*>(p.v) + b * p.v_sb + h * p.v_sh;
  const int nt_all = (L + BK - 1) / BK;
  // causal: key tiles beyond this workgroup's last query row contribute nothing
  const int nt = p.causal ? min(nt_all, (min(q0 + BQ, L) + BK - 1) / BK) : nt_all;

  uint4 kr[CPT], vr[CPT];       // the next tile, already bf16 (fp32 tensors are converted on arrival)
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + 256 * i, key = c / (D / 8), dc = c - key * (D / 8);
      const int kj = t * BK + key;
      if (kj < L) {
        kr[i] = load8_bf16(kbase + (int64_t)kj * p.k_sl + dc * 8);
        vr[i] = load8_bf16(vbase + (int64_t)kj * p.v_sl + dc * 8);
      } else {
        kr[i] = make_uint4(0u, 0u, 0u, 0u);
        vr[i] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* Kd = smem + buf * KV_BYTES;
    unsigned char* Vd = Kd + BK * PITCH;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + 256 * i, key = c / (D / 8), dc = c - key * (D / 8);
      *reinterpret_cast<uint4*>(Kd + key * PITCH + dc * 16) = kr[i];
      *reinterpret_cast<uint4*>(Vd + key * PITCH + dc * 16) = vr[i];
    }
  };

  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const AttnDropKey dk = attn_drop_key(p.seed, p.offset, p.drop_p);
  const int Lh = (L + 1) >> 1;
  const int q_lane0 = q0 + wid * 16 + 4 * g;      // first of this lane's 4 query rows
  uint64_t drb[4] = {0, 0, 0, 0};
  if constexpr (DROP) {
#pragma unroll
    for (int r = 0; r < 4; ++r) drb[r] = attn_row_base((uint64_t)bh * L + (uint64_t)(q_lane0 + r), Lh);
  }
  const int tq = l15 >> 2, tp = l15 & 3;          // transposing-read roles inside a 16-lane group

  // ONE barrier per key tile: tile t+1 is fetched into registers while tile t is multiplied and lands in the OTHER LDS buffer
  // (last read in iteration t-1, which every wave has left through the barrier that ended it).
  if (nt > 0) { load_tile(0); store_tile(0); }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const unsigned char* Ks = smem + (NB == 2 ? (t & 1) : 0) * KV_BYTES;
    const unsigned char* Vs = Ks + BK * PITCH;
    // Score addends of THIS tile first: vmcnt retires in issue order, so a wait for these (younger) loads would otherwise also
    // drain the next tile's K / V prefetch issued below and serialise it with the softmax.
    float madd[4], badd[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int kj = t * BK + 16 * n + l15;
      madd[n] = (p.mask_add && kj < L) ? p.mask_add[(int64_t)b * L + kj] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q_lane0 + r;
        badd[n][r] = (p.bias && kj < L && qi < L) ? p.bias[((int64_t)h * L + qi) * L + kj] : 0.f;
      }
    }
    if (t + 1 < nt && !(abl & 1)) load_tile(t + 1);   // in flight under this tile's MFMAs and softmax

    // ---- S = (scale Q) K^T : 4 key groups of 16
    f32x4_t s[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      s[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (16 * n + l15) * PITCH + (32 * ks + 8 * g) * 2);
        if (!(abl & 4)) s[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, qf[ks]), __builtin_bit_cast(bf16x8_t, kf), s[n], 0, 0, 0);
      }
    }
    if (abl & 2) {   // timing experiment: no softmax, P = S
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<uint16_t*>(Ps + (4 * g + r) * PPITCH + (16 * n + l15) * 2) = (uint16_t)f32_to_bf16_bits(s[n][r]);
    } else {
    // ---- bias / masks; running max
    float mx[4] = {-3.4028235e38f, -3.4028235e38f, -3.4028235e38f, -3.4028235e38f};   // -FLT_MAX, not -1e30: a row whose keys all carry the finite HF mask (finfo.min) must come out as the uniform average of V, as torch's softmax gives it
    bool ok[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int kj = t * BK + 16 * n + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q_lane0 + r;
        const bool valid = kj < L && qi < L && !(p.causal && kj > qi);
        const float x = s[n][r] * p.scale + madd[n] + badd[n][r];
        s[n][r] = x;
        ok[n][r] = valid;
        if (valid) mx[r] = fmaxf(mx[r], x);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int sh = 1; sh < 16; sh <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], sh, 64));
    }
    float alpha[4], rs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m_new = fmaxf(m_run[r], mx[r]);
      alpha[r] = __expf(m_run[r] - m_new);
      m_run[r] = m_new;
      rs[r] = 0.f;
    }
    // ---- P = exp(S - m); row sums on the un-dropped probabilities; dropout; bf16 -> LDS [query][key]
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int kj = t * BK + 16 * n + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = ok[n][r] ? __expf(s[n][r] - m_run[r]) : 0.f;
        rs[r] += pv;
        if constexpr (DROP) {
          if (ok[n][r]) pv = attn_keep(dk, drb[r], kj) ? pv * inv_keep : 0.f;
        }
        *reinterpret_cast<uint16_t*>(Ps + (4 * g + r) * PPITCH + (16 * n + l15) * 2) = (uint16_t)f32_to_bf16_bits(pv);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int sh = 1; sh < 16; sh <<= 1) rs[r] += __shfl_xor(rs[r], sh, 64);
      l_run[r] = l_run[r] * alpha[r] + rs[r];
    }
#pragma unroll
    for (int i = 0; i < DN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[i][r] *= alpha[r];
    }
    // The P tile is wave-private and DS instructions of one wave execute in order: the reads below see the writes above
    // without a workgroup barrier (the compiler keeps the order: same LDS array).
    // ---- O += P V : A = P [16 q x 32 keys] (row reads), B = V [32 keys x 16 d] (transposing reads of the [key][d] tile)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 pf = *reinterpret_cast<const uint4*>(Ps + l15 * PPITCH + (32 * ks + 8 * g) * 2);
      const int r0 = 32 * ks + 8 * g + tq, r1 = r0 + 4;
#pragma unroll
      for (int dn = 0; dn < DN; ++dn) {
        const int colb = (16 * dn + 4 * tp) * 2;
        const uint2 lo = lds_tr16_b64(Vs + r0 * PITCH + colb);
        const uint2 hi = lds_tr16_b64(Vs + r1 * PITCH + colb);
        const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (!(abl & 4)) oacc[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pf), __builtin_bit_cast(bf16x8_t, vf), oacc[dn], 0, 0, 0);
      }
    }
    if constexpr (NB == 2) {
      if (t + 1 < nt && !(abl & 8)) store_tile((t + 1) & 1);
      __syncthreads();
    } else {
      __syncthreads();                       // every wave is done with this tile's K / V
      if (t + 1 < nt) store_tile(0);
      __syncthreads();
    }
  }

  // ---- O / l, LSE
  IO* obase = reinterpret_cast<IO*>(p.o) + b * p.o_sb + h * p.o_sh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q_lane0 + r;
    if (qi >= L) continue;
    const float inv = l_run[r] > 0.f ? 1.f / l_run[r] : 0.f;
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) obase[(int64_t)qi * p.o_sl + 16 * dn + l15] = from_f32<IO>(oacc[dn][r] * inv);
    if (p.lse && l15 == 0) p.lse[(int64_t)bh * L + qi] = m_run[r] + __logf(fmaxf(l_run[r], 1e-38f));
  }
}

// ---- round 4: 32 query rows per wave, no P round trip through LDS (Dh = 64)
// The kernel above gives a wave 16 query rows: every wave re-reads the whole 64-key K and V tiles for 16 MFMAs, P goes accumulator -> bf16 ->
// LDS -> operand, and every row statistic is a 4-step xor-shuffle (ablation, profiles/r02_experiments.txt (14): 307 of 822 us remain with
// loads, softmax, MFMAs and LDS stores all removed).  Here:
//   * S^T = K Q^T on v_mfma_f32_32x32x16_bf16 with the operands SWAPPED (A = K rows, B = Q^T): the accumulator holds a QUERY per lane column
//     and 16 of a 32-key tile's scores in its registers (the other 16 in lane + 32) -- max / sum over keys are in-register loops plus ONE
//     xor-32 shuffle, bias / mask / dropout index one (query, key) per register
//   * O^T += V^T P^T takes the exponentiated accumulator ITSELF as the B operand (registers 8s..8s+7 -> bf16 are the fragment of k-step s; the
//     matching k order of the V^T fragment is keys 16s + 4h + {0..3} and 16s + 8 + 4h + {0..3}: two transposing reads) -- no LDS write,
//     no barrier, no second layout
//   * a workgroup is 128 query rows (4 waves x 32): the K / V tiles are fetched and staged once per 128 queries instead of per 64, and
//     each fragment read feeds twice the MFMA work
// Same staging (registers -> bf16 -> LDS, pitch D*2 + 16), same dropout generator, masks, strides and LSE as the kernel above.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <typename IO, bool DROP>
__global__ __launch_bounds__(256) void flash_fwd2_kernel(const FlashArgs p) {
  constexpr int D = 64, BQ = 128, BK = 64;
  constexpr int PITCH = D * 2 + 16;
  constexpr int CPT = BK * (D / 8) / 256;    // 8-element chunks per thread per tile (K and V each) = 2
  constexpr int KV_BYTES = 2 * BK * PITCH;
  constexpr int MAXL = 1024;                 // key-mask row kept in LDS (log2 domain); longer sequences take the kernel above
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KV_BYTES + MAXL * 4];
  float* mask_s = reinterpret_cast<float*>(smem + 2 * KV_BYTES);
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int lin0 = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  const int lin = p.xcd ? xcd_remap(lin0, (int)(gridDim.x * gridDim.y)) : lin0;
  const int bh = lin / (int)gridDim.x, b = bh / p.H, h = bh - b * p.H;
  const int q0 = (lin - bh * (int)gridDim.x) * BQ;
  const int L = p.L;
  const int qi = q0 + wid * 32 + r;          // this lane's query (both lane halves hold the same one)
  const bool q_ok = qi < L;
  constexpr float LOG2E = 1.4426950408889634f;
  const float sc2 = p.scale * LOG2E;         // scores live in the log2 domain: exp2 needs no multiply per element

  // the key mask of this batch row, once, in the log2 domain; keys past L get the most negative finite value's stand-in (never used: invalid)
  for (int k = tid; k < ((L + BK - 1) / BK) * BK; k += 256) mask_s[k] = (p.mask_add && k < L) ? fmaxf(p.mask_add[(int64_t)b * L + k] * LOG2E, -3.4028235e38f) : 0.f;   // HF's finfo.min stays finite: a fully masked row is the uniform average, as torch gives it

  // Q^T as the B operand: lane (r, hh) holds Q[query r][d = 16 ks + 8 hh .. +7]
  uint4 qf[4];
  {
    const IO* qp = reinterpret_cast<const IO*>(p.q) + b * p.q_sb + h * p.q_sh + (int64_t)(q_ok ? qi : L - 1) * p.q_sl;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = load8_bf16(qp + 16 * ks + 8 * hh);
  }
  float m_run = -3.4028235e38f, l_run = 0.f;     // running max in the log2 domain
  f32x16_t oacc[2];      // O^T: rows d = 32 dt + (i & 3) + 8 (i >> 2) + 4 hh in register i, column = query r
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;

  const IO* kbase = reinterpret_cast<const IO*>(p.k) + b * p.k_sb + h * p.k_sh;
  const IO* vbase = reinterpret_cast<const IO*>(p.v) + b * p.v_sb + h * p.v_sh;
  const int nt_all = (L + BK - 1) / BK;
  const int nt = p.causal ? min(nt_all, (min(q0 + BQ, L) + BK - 1) / BK) : nt_all;

  uint4 kr[CPT], vr[CPT];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + 256 * i, key = c >> 3, dc = c & 7;
      const int kj = t * BK + key;
      if (kj < L) {
        kr[i] = load8_bf16(kbase + (int64_t)kj * p.k_sl + dc * 8);
        vr[i] = load8_bf16(vbase + (int64_t)kj * p.v_sl + dc * 8);
      } else {
        kr[i] = make_uint4(0u, 0u, 0u, 0u);
        vr[i] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* Kd = smem + buf * KV_BYTES;
    unsigned char* Vd = Kd + BK * PITCH;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + 256 * i, key = c >> 3, dc = c & 7;
      *reinterpret_cast<uint4*>(Kd + key * PITCH + dc * 16) = kr[i];
      *reinterpret_cast<uint4*>(Vd + key * PITCH + dc * 16) = vr[i];
    }
  };

  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const AttnDropKey dk = attn_drop_key(p.seed, p.offset, p.drop_p);
  const int Lh = (L + 1) >> 1;
  const uint64_t drow = attn_row_base((uint64_t)bh * L + (uint64_t)qi, Lh);
  const int l15 = lane & 15, tq = l15 >> 2, tp = l15 & 3, G = lane >> 4;   // transposing-read roles: group G = 16 d columns 16 (G & 1).., lane half G >> 1

  if (nt > 0) { load_tile(0); store_tile(0); }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const unsigned char* Ks = smem + (t & 1) * KV_BYTES;
    const unsigned char* Vs = Ks + BK * PITCH;
    if (t + 1 < nt) load_tile(t + 1);

    // ---- S^T = K Q^T: two 32-key tiles, 4 k-steps of 16 each
    f32x16_t s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kt][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (32 * kt + r) * PITCH + (16 * ks + 8 * hh) * 2);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[ks]), s[kt], 0, 0, 0);
      }
    }
    // ---- scale + key mask (LDS, four consecutive keys per read), running max: in-lane over 32 scores + one shuffle across the lane halves.
    // A tile that lies wholly inside the sequence of a non-causal launch needs no per-element validity (lanes whose query is past L
    // compute a row nobody stores).
    const bool edge = p.causal || (t + 1) * BK > L;
    float mx = -3.4028235e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 mk = *reinterpret_cast<const float4*>(mask_s + t * BK + 32 * kt + 8 * gq + 4 * hh);
        const float mv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = 4 * gq + e;
          float x = s[kt][i] * sc2 + mv[e];
          if (edge) {
            const int kj = t * BK + 32 * kt + 8 * gq + 4 * hh + e;
            if (!(kj < L && !(p.causal && kj > qi))) x = -INFINITY;       // exp2 -> 0; the max ignores it
          }
          s[kt][i] = x;
          mx = fmaxf(mx, x);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    mx = fmaxf(mx, -3.4028235e38f);                                         // a row with no valid key so far: finite, so that exp2(-inf - m) = 0 below
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float rs = 0.f;
    uint4 pf[2][2];     // P^T fragments: [key tile][k-step]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float pv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float e = __builtin_amdgcn_exp2f(s[kt][i] - m_run);
        rs += e;
        pv[i] = e;
      }
      if constexpr (DROP) {   // a lane's registers are quads of consecutive keys (first key a multiple of 4): two pair hashes per quad
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int kj0 = t * BK + 32 * kt + 8 * gq + 4 * hh;
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const uint32_t hsh = attn_pair_hash(dk, drow, kj0 + 2 * pr);
            pv[4 * gq + 2 * pr] = attn_keep_bits(hsh, 0, dk.thr) ? pv[4 * gq + 2 * pr] * inv_keep : 0.f;
            pv[4 * gq + 2 * pr + 1] = attn_keep_bits(hsh, 1, dk.thr) ? pv[4 * gq + 2 * pr + 1] * inv_keep : 0.f;
          }
        }
      }
#pragma unroll
      for (int sk = 0; sk < 2; ++sk)
        pf[kt][sk] = make_uint4(pack_bf16(pv[8 * sk], pv[8 * sk + 1]), pack_bf16(pv[8 * sk + 2], pv[8 * sk + 3]),
                                pack_bf16(pv[8 * sk + 4], pv[8 * sk + 5]), pack_bf16(pv[8 * sk + 6], pv[8 * sk + 7]));
    }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
    // ---- O^T += V^T P^T: A = V^T fragment (d = 32 dt + lane & 31; keys 16 sk + 4 hh + {0..3}, 16 sk + 8 + 4 hh + {0..3} of the 32-key tile)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const int k0 = 32 * kt + 16 * sk + 4 * (G >> 1);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int colb = (32 * dt + 16 * (G & 1) + 4 * tp) * 2;
          const uint2 lo = lds_tr16_b64(Vs + (k0 + tq) * PITCH + colb);
          const uint2 hi = lds_tr16_b64(Vs + (k0 + 8 + tq) * PITCH + colb);
          const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf), __builtin_bit_cast(bf16x8_t, pf[kt][sk]), oacc[dt], 0, 0, 0);
        }
      }
    if (t + 1 < nt) store_tile((t + 1) & 1);
    __syncthreads();
  }

  // ---- O / l, LSE (natural log): lane holds O[qi][32 dt + 8 (i >> 2) + 4 hh + (i & 3)]: four consecutive d per register quad
  if (!q_ok) return;
  IO* orow = reinterpret_cast<IO*>(p.o) + b * p.o_sb + h * p.o_sh + (int64_t)qi * p.o_sl;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      IO* dst = orow + 32 * dt + 8 * gq + 4 * hh;
      const float a0 = oacc[dt][4 * gq] * inv, a1 = oacc[dt][4 * gq + 1] * inv, a2 = oacc[dt][4 * gq + 2] * inv, a3 = oacc[dt][4 * gq + 3] * inv;
      if constexpr (sizeof(IO) == 4) *reinterpret_cast<float4*>(dst) = make_float4(a0, a1, a2, a3);
      else *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16(a0, a1), pack_bf16(a2, a3));
    }
  if (p.lse && hh == 0) p.lse[(int64_t)bh * L + qi] = m_run * 0.6931471805599453f + __logf(fmaxf(l_run, 1e-38f));
}

}  // namespace

template <typename IO>
static int flash_launch(FlashArgs& a, int Dh, hipStream_t st) {
  static const int v2 = [] { const char* e = getenv("MMSKIN_FLASH_V2"); return e ? atoi(e) : 1; }();
  if (v2 && Dh == 64 && !a.bias && a.L <= 1024) {   // 128 query rows per workgroup, in-register softmax (flash_fwd2_kernel); a score bias [H][L][L]
                                                     // is read per (query, key): with a query per lane its rows are L floats apart -- the kernel above reads it along keys
    if (a.drop_p > 0.f) hipLaunchKernelGGL((flash_fwd2_kernel<IO, true>), dim3(ceil_div(a.L, 128), a.B * a.H), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((flash_fwd2_kernel<IO, false>), dim3(ceil_div(a.L, 128), a.B * a.H), dim3(256), 0, st, a);
    HIP_CHECK_RET(hipGetLastError());
    return MMSKIN_OK;
  }
  const dim3 grid(ceil_div(a.L, 64), a.B * a.H);
  static const int nb = [] { const char* e = getenv("MMSKIN_FLASH_BUFFERS"); return e ? atoi(e) : 2; }();
#define FA_GO(DV, NBV) do { if (a.drop_p > 0.f) hipLaunchKernelGGL((flash_fwd_kernel<DV, NBV, IO, true>), grid, dim3(256), 0, st, a); \
                             else hipLaunchKernelGGL((flash_fwd_kernel<DV, NBV, IO, false>), grid, dim3(256), 0, st, a); } while (0)
  if (Dh == 32) { if (nb == 1) FA_GO(32, 1); else FA_GO(32, 2); }
  else { if (nb == 1) FA_GO(64, 1); else FA_GO(64, 2); }
#undef FA_GO
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

extern "C" {

int mmskin_flash_attention_forward(const void* q, const void* k, const void* v, const float* mask_add, const float* bias,
                                   void* o, float* lse, int B, int H, int L, int Dh, const int64_t* strides12, int io_dtype,
                                   float scale, int causal, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
  ARG_CHECK(q && k && v && o && strides12, "flash_attention_forward: null argument");
  ARG_CHECK(B > 0 && H > 0 && L > 0 && (Dh == 32 || Dh == 64), "flash_attention_forward: B=%d H=%d L=%d Dh=%d (Dh must be 32 or 64)", B, H, L, Dh);
  ARG_CHECK(io_dtype == MMSKIN_F32 || io_dtype == MMSKIN_BF16, "flash_attention_forward: io_dtype %d", io_dtype);
  ARG_CHECK(drop_p >= 0.f && drop_p < 1.f, "flash_attention_forward: dropout %f", drop_p);
  ARG_CHECK((int64_t)B * H <= 65535, "flash_attention_forward: B*H = %lld exceeds the grid", (long long)B * H);
  const int per16 = io_dtype == MMSKIN_BF16 ? 8 : 4;    // elements per 16-byte load
  for (int i = 0; i < 9; ++i) ARG_CHECK(strides12[i] % per16 == 0, "flash_attention_forward: stride %d = %lld is not a multiple of %d elements (16-byte loads)", i, (long long)strides12[i], per16);
  ARG_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "flash_attention_forward: q / k / v must be 16-byte aligned");
  FlashArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.mask_add = mask_add; a.bias = bias;
  a.B = B; a.H = H; a.L = L;
  a.q_sb = strides12[0]; a.q_sh = strides12[1]; a.q_sl = strides12[2];
  a.k_sb = strides12[3]; a.k_sh = strides12[4]; a.k_sl = strides12[5];
  a.v_sb = strides12[6]; a.v_sh = strides12[7]; a.v_sl = strides12[8];
  a.o_sb = strides12[9]; a.o_sh = strides12[10]; a.o_sl = strides12[11];
  a.scale = scale; a.drop_p = drop_p; a.causal = causal; a.seed = seed; a.offset = offset;
  a.ablate = 0;
  { static const int xcd = [] { const char* e = getenv("MMSKIN_FLASH_XCD"); return e ? atoi(e) : 1; }(); a.xcd = xcd; }
#ifdef MMSKIN_ABLATE
  { const char* e = getenv("MMSKIN_FLASH_ABLATE"); a.ablate = e ? atoi(e) : 0; }
#endif
  return io_dtype == MMSKIN_BF16 ? flash_launch<bf16_t>(a, Dh, (hipStream_t)stream) : flash_launch<float>(a, Dh, (hipStream_t)stream);
}

}  // extern "C"

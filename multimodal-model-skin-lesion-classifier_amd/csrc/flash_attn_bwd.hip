// Backward of the fused softmax attention (flash_attn.hip) for gfx950: dQ, dK, dV (and dS for the score-bias gradient) recomputed from
// Q, K, V and the forward's row log-sum-exp -- no [B, H, L, L] probability tensor is kept between forward and backward.
//
// Replaces, for TRAINABLE transformer blocks in bf16-operand mode, autograd's backward through the attention of the reference's timm /
// transformers encoders (loadImageModelClassifier.py:117-131 `timm.create_model(...)` with `unfrozen_weights` / `partial`; :170-181
// `AutoModel.from_pretrained`), which until now left the fused path for the unfused fp32 chain (QK^T GEMM -> softmax with the
// probabilities saved -> PV GEMM and their five backward GEMMs).
//
//   P = exp(scale Q K^T + bias + mask - lse)          dP = dO V^T (. keep / (1 - p) under dropout)
//   dS = P . (dP - delta),  delta_i = sum_d dO_id O_id
//   dV = (P . keep / (1 - p))^T dO      dK = scale dS^T Q      dQ = scale dS K      d(bias) = sum_batch dS
//
// Two kernels, both the forward kernel's tile machinery (64-row tiles, bf16 operands, fp32 accumulation, 16x16x32 MFMA, LDS row
// pitch D * 2 + 16 B that serves b128 row reads and transposing reads alike, one barrier per streamed tile):
//   * flash_bwd_dq_kernel : one workgroup = 64 QUERY rows, K / V tiles streamed; S and dP in the accumulator layout (a key per lane,
//     four queries per lane), dS -> bf16 -> a wave-private LDS tile -> A operand of dQ += dS K (K by transposing reads);
//   * flash_bwd_dkv_kernel: one workgroup = 64 KEY rows, Q / dO tiles streamed; S^T and dP^T (a query per lane, four keys per lane),
//     (P . keep)^T and dS^T through wave-private tiles -> dV += P^T dO, dK += dS^T Q.
// Each recomputes S (2 x the forward's QK^T); neither needs a reduction across workgroups, so the result is deterministic.
// The dropout mask is regenerated from the forward's counter (element index of the [B, H, L, L] tensor).  q, k, v, o, dO, dq, dk, dv are
// fp32 with element strides (last dim contiguous); bias [H, L, L] and its transpose biasT [H, key, query] (coalesced in the second
// kernel); ds_out (optional) [B, H, L, L] fp32 receives dS for the caller's sum over the batch.
#include "../../include/mmskin.h"
#include <stdlib.h>

#include "common.h"

namespace {

struct FlashBwdArgs {
  const float* q; const float* k; const float* v; const float* o; const float* dout;
  const float* lse;        // [B * H][L]
  float* delta;            // [B * H][L] scratch: sum_d dO O
  const float* mask_add;   // [B][L] or null
  const float* bias;       // [H][L][L] or null
  const float* biasT;      // [H][L key][L query] or null (required with bias)
  float* dq; float* dk; float* dv;
  float* ds_out;           // [B][H][L][L] or null
  int B, H, L;
  int64_t q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, o_sl;      // o strides serve o and dout
  int64_t g_sb, g_sh, g_sl;                                                              // dq / dk / dv
  float scale, drop_p;
  int causal;
  uint64_t seed, offset;
};

__device__ __forceinline__ uint2 fb_tr16_b64(const unsigned char* p) {
  s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ uint32_t fb_pack(float a, float b) { return f32_to_bf16_bits(a) | (f32_to_bf16_bits(b) << 16); }
__device__ __forceinline__ uint4 fb_load8(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
  return make_uint4(fb_pack(a.x, a.y), fb_pack(a.z, a.w), fb_pack(c.x, c.y), fb_pack(c.z, c.w));
}
__device__ __forceinline__ f32x4_t fb_mma(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// delta[bh][i] = sum_d dO[i][d] * O[i][d]; one thread per row
template <int D>
__global__ __launch_bounds__(256) void flash_delta_kernel(const FlashBwdArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x, bh = blockIdx.y;
  if (i >= p.L) return;
  const int b = bh / p.H, h = bh - b * p.H;
  const float* o = p.o + b * p.o_sb + h * p.o_sh + (int64_t)i * p.o_sl;
  const float* g = p.dout + b * p.o_sb + h * p.o_sh + (int64_t)i * p.o_sl;
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const float4 a = *reinterpret_cast<const float4*>(o + d), c = *reinterpret_cast<const float4*>(g + d);
    s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
  p.delta[(int64_t)bh * p.L + i] = s;
}

// Shared tile streamer: two [64][D] fp32 tensors (rows t*64 .. of `xa`, `xb`) -> bf16 -> LDS [row][PITCH], double-buffered.
template <int D>
struct Tiles {
  static constexpr int PITCH = D * 2 + 16, CPT = 64 * (D / 8) / 256, BYTES = 2 * 64 * PITCH;
};

template <int D>
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(const FlashBwdArgs p) {
  constexpr int BQ = 64, BK = 64, PITCH = Tiles<D>::PITCH, PPITCH = BK * 2 + 16, KS = D / 32, DN = D / 16, CPT = Tiles<D>::CPT;
  constexpr int KV_BYTES = Tiles<D>::BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * KV_BYTES + 4 * 16 * PPITCH];
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  unsigned char* Ps = smem + 2 * KV_BYTES + wid * 16 * PPITCH;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * BQ, L = p.L;

  uint4 qf[KS], dof[KS];
  {
    const int qi = q0 + wid * 16 + l15, qc = qi < L ? qi : L - 1;
    const float* qp = p.q + b * p.q_sb + h * p.q_sh + (int64_t)qc * p.q_sl;
    const float* gp = p.dout + b * p.o_sb + h * p.o_sh + (int64_t)qc * p.o_sl;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { qf[ks] = fb_load8(qp + 32 * ks + 8 * g); dof[ks] = fb_load8(gp + 32 * ks + 8 * g); }
  }
  const int q_lane0 = q0 + wid * 16 + 4 * g;
  float lse_r[4], dl_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q_lane0 + r, qc = qi < L ? qi : L - 1;
    lse_r[r] = p.lse[(int64_t)bh * L + qc]; dl_r[r] = p.delta[(int64_t)bh * L + qc];
  }
  f32x4_t acc[DN];
#pragma unroll
  for (int i = 0; i < DN; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const float* kbase = p.k + b * p.k_sb + h * p.k_sh;
  const float* vbase = p.v + b * p.v_sb + h * p.v_sh;
  const int nt_all = (L + BK - 1) / BK;
  const int nt = p.causal ? min(nt_all, (min(q0 + BQ, L) + BK - 1) / BK) : nt_all;
  uint4 kr[CPT], vr[CPT];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + 256 * i, key = c / (D / 8), dc = c - key * (D / 8), kj = t * BK + key;
      if (kj < L) { kr[i] = fb_load8(kbase + (int64_t)kj * p.k_sl + dc * 8); vr[i] = fb_load8(vbase + (int64_t)kj * p.v_sl + dc * 8); }
      else { kr[i] = make_uint4(0u, 0u, 0u, 0u); vr[i] = make_uint4(0u, 0u, 0u, 0u); }
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
  const int tq = l15 >> 2, tp = l15 & 3;

  if (nt > 0) { load_tile(0); store_tile(0); }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const unsigned char* Ks = smem + (t & 1) * KV_BYTES;
    const unsigned char* Vs = Ks + BK * PITCH;
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
    if (t + 1 < nt) load_tile(t + 1);
    // ---- S = Q K^T, dP = dO V^T (a key per lane, queries 4 g + r)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (16 * n + l15) * PITCH + (32 * ks + 8 * g) * 2);
        const uint4 vf = *reinterpret_cast<const uint4*>(Vs + (16 * n + l15) * PITCH + (32 * ks + 8 * g) * 2);
        s = fb_mma(qf[ks], kf, s);
        dp = fb_mma(dof[ks], vf, dp);
      }
      const int kj = t * BK + 16 * n + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = q_lane0 + r;
        const bool valid = kj < L && qi < L && !(p.causal && kj > qi);
        const float x = s[r] * p.scale + madd[n] + badd[n][r];
        const float pr = valid ? __expf(x - lse_r[r]) : 0.f;
        float zs = 1.f;
        if (p.drop_p > 0.f && valid) zs = attn_keep(dk, attn_row_base((uint64_t)bh * L + (uint64_t)qi, Lh), kj) ? inv_keep : 0.f;
        const float ds = pr * (dp[r] * zs - dl_r[r]);
        if (p.ds_out && valid) p.ds_out[(((int64_t)bh * L + qi) * L) + kj] = ds;
        *reinterpret_cast<uint16_t*>(Ps + (4 * g + r) * PPITCH + (16 * n + l15) * 2) = (uint16_t)f32_to_bf16_bits(ds * p.scale);
      }
    }
    // ---- dQ += dS K : A = dS [16 q x 32 keys] (row reads of the wave-private tile), B = K [32 keys x 16 d] (transposing reads)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 pf = *reinterpret_cast<const uint4*>(Ps + l15 * PPITCH + (32 * ks + 8 * g) * 2);
      const int r0 = 32 * ks + 8 * g + tq, r1 = r0 + 4;
#pragma unroll
      for (int dn = 0; dn < DN; ++dn) {
        const int colb = (16 * dn + 4 * tp) * 2;
        const uint2 lo = fb_tr16_b64(Ks + r0 * PITCH + colb), hi = fb_tr16_b64(Ks + r1 * PITCH + colb);
        acc[dn] = fb_mma(pf, make_uint4(lo.x, lo.y, hi.x, hi.y), acc[dn]);
      }
    }
    if (t + 1 < nt) store_tile((t + 1) & 1);
    __syncthreads();
  }
  float* dqb = p.dq + b * p.g_sb + h * p.g_sh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q_lane0 + r;
    if (qi >= L) continue;
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) dqb[(int64_t)qi * p.g_sl + 16 * dn + l15] = acc[dn][r];
  }
}

template <int D>
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(const FlashBwdArgs p) {
  constexpr int BQ = 64, BK = 64, PITCH = Tiles<D>::PITCH, PPITCH = BQ * 2 + 16, KS = D / 32, DN = D / 16, CPT = Tiles<D>::CPT;
  constexpr int QG_BYTES = Tiles<D>::BYTES;   // one Q tile + one dO tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * QG_BYTES + 4 * 2 * 16 * PPITCH];
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  unsigned char* Pt = smem + 2 * QG_BYTES + wid * 2 * 16 * PPITCH;   // (P . keep)^T tile, then the dS^T tile behind it
  unsigned char* St = Pt + 16 * PPITCH;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int k0 = blockIdx.x * BK, L = p.L;

  uint4 kf[KS], vf[KS];     // this wave's 16 keys as A-operand fragments (key l15, d = 32 ks + 8 g ..)
  {
    const int kj = k0 + wid * 16 + l15, kc = kj < L ? kj : L - 1;
    const float* kp = p.k + b * p.k_sb + h * p.k_sh + (int64_t)kc * p.k_sl;
    const float* vp = p.v + b * p.v_sb + h * p.v_sh + (int64_t)kc * p.v_sl;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { kf[ks] = fb_load8(kp + 32 * ks + 8 * g); vf[ks] = fb_load8(vp + 32 * ks + 8 * g); }
  }
  const int k_lane0 = k0 + wid * 16 + 4 * g;     // first of this lane's 4 key rows (accumulator layout)
  float madd_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kj = k_lane0 + r;
    madd_r[r] = (p.mask_add && kj < L) ? p.mask_add[(int64_t)b * L + kj] : 0.f;
  }
  f32x4_t dka[DN], dva[DN];
#pragma unroll
  for (int i = 0; i < DN; ++i) { dka[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dva[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  const float* qbase = p.q + b * p.q_sb + h * p.q_sh;
  const float* gbase = p.dout + b * p.o_sb + h * p.o_sh;
  const int nt = (L + BQ - 1) / BQ;
  const int t_first = p.causal ? k0 / BQ : 0;     // causal: query tiles before this key tile see none of its keys
  uint4 qr[CPT], gr[CPT];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + 256 * i, row = c / (D / 8), dc = c - row * (D / 8), qi = t * BQ + row;
      if (qi < L) { qr[i] = fb_load8(qbase + (int64_t)qi * p.q_sl + dc * 8); gr[i] = fb_load8(gbase + (int64_t)qi * p.o_sl + dc * 8); }
      else { qr[i] = make_uint4(0u, 0u, 0u, 0u); gr[i] = make_uint4(0u, 0u, 0u, 0u); }
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* Qd = smem + buf * QG_BYTES;
    unsigned char* Gd = Qd + BQ * PITCH;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = tid + 256 * i, row = c / (D / 8), dc = c - row * (D / 8);
      *reinterpret_cast<uint4*>(Qd + row * PITCH + dc * 16) = qr[i];
      *reinterpret_cast<uint4*>(Gd + row * PITCH + dc * 16) = gr[i];
    }
  };
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const AttnDropKey dk = attn_drop_key(p.seed, p.offset, p.drop_p);
  const int Lh = (L + 1) >> 1;
  const int tq = l15 >> 2, tp = l15 & 3;

  if (t_first < nt) { load_tile(t_first); store_tile(t_first & 1); }
  __syncthreads();
  for (int t = t_first; t < nt; ++t) {
    const unsigned char* Qs = smem + (t & 1) * QG_BYTES;
    const unsigned char* Gs = Qs + BQ * PITCH;
    // per-column (query) constants and the bias of this tile first (in front of the prefetch: vmcnt retires in issue order)
    float lse_c[4], dl_c[4], badd[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int qi = t * BQ + 16 * n + l15, qc = qi < L ? qi : L - 1;
      lse_c[n] = p.lse[(int64_t)bh * L + qc]; dl_c[n] = p.delta[(int64_t)bh * L + qc];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = k_lane0 + r;
        badd[n][r] = (p.biasT && kj < L && qi < L) ? p.biasT[((int64_t)h * L + kj) * L + qi] : 0.f;
      }
    }
    if (t + 1 < nt) load_tile(t + 1);
    // ---- S^T = K Q^T, dP^T = V dO^T (a query per lane, keys 4 g + r)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 qrow = *reinterpret_cast<const uint4*>(Qs + (16 * n + l15) * PITCH + (32 * ks + 8 * g) * 2);
        const uint4 grow = *reinterpret_cast<const uint4*>(Gs + (16 * n + l15) * PITCH + (32 * ks + 8 * g) * 2);
        s = fb_mma(kf[ks], qrow, s);
        dp = fb_mma(vf[ks], grow, dp);
      }
      const int qi = t * BQ + 16 * n + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kj = k_lane0 + r;
        const bool valid = kj < L && qi < L && !(p.causal && kj > qi);
        const float x = s[r] * p.scale + madd_r[r] + badd[n][r];
        const float pr = valid ? __expf(x - lse_c[n]) : 0.f;
        float zs = 1.f;
        if (p.drop_p > 0.f && valid) zs = attn_keep(dk, attn_row_base((uint64_t)bh * L + (uint64_t)qi, Lh), kj) ? inv_keep : 0.f;
        const float ds = pr * (dp[r] * zs - dl_c[n]);
        *reinterpret_cast<uint16_t*>(Pt + (4 * g + r) * PPITCH + (16 * n + l15) * 2) = (uint16_t)f32_to_bf16_bits(pr * zs);
        *reinterpret_cast<uint16_t*>(St + (4 * g + r) * PPITCH + (16 * n + l15) * 2) = (uint16_t)f32_to_bf16_bits(ds * p.scale);
      }
    }
    // ---- dV += (P . keep)^T dO,  dK += dS^T Q : A = the wave-private [16 keys x 64 queries] tiles (row reads), B = dO / Q (transposing reads)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint4 pf = *reinterpret_cast<const uint4*>(Pt + l15 * PPITCH + (32 * ks + 8 * g) * 2);
      const uint4 sf = *reinterpret_cast<const uint4*>(St + l15 * PPITCH + (32 * ks + 8 * g) * 2);
      const int r0 = 32 * ks + 8 * g + tq, r1 = r0 + 4;
#pragma unroll
      for (int dn = 0; dn < DN; ++dn) {
        const int colb = (16 * dn + 4 * tp) * 2;
        const uint2 glo = fb_tr16_b64(Gs + r0 * PITCH + colb), ghi = fb_tr16_b64(Gs + r1 * PITCH + colb);
        dva[dn] = fb_mma(pf, make_uint4(glo.x, glo.y, ghi.x, ghi.y), dva[dn]);
        const uint2 qlo = fb_tr16_b64(Qs + r0 * PITCH + colb), qhi = fb_tr16_b64(Qs + r1 * PITCH + colb);
        dka[dn] = fb_mma(sf, make_uint4(qlo.x, qlo.y, qhi.x, qhi.y), dka[dn]);
      }
    }
    if (t + 1 < nt) store_tile((t + 1) & 1);
    __syncthreads();
  }
  float* dkb = p.dk + b * p.g_sb + h * p.g_sh;
  float* dvb = p.dv + b * p.g_sb + h * p.g_sh;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int kj = k_lane0 + r;
    if (kj >= L) continue;
#pragma unroll
    for (int dn = 0; dn < DN; ++dn) {
      dkb[(int64_t)kj * p.g_sl + 16 * dn + l15] = dka[dn][r];
      dvb[(int64_t)kj * p.g_sl + 16 * dn + l15] = dva[dn][r];
    }
  }
}

}  // namespace

extern "C" {

/* strides15: element strides (batch, head, token) of q, k, v, o (= dout), then of dq / dk / dv (shared); last dims contiguous */
int mmskin_flash_attention_backward(const float* q, const float* k, const float* v, const float* o, const float* dout, const float* lse,
                                    const float* mask_add, const float* bias, const float* biasT, float* delta, float* dq, float* dk,
                                    float* dv, float* ds_out, int B, int H, int L, int Dh, const int64_t* strides15, float scale,
                                    int causal, float drop_p, uint64_t seed, uint64_t offset, void* stream) {
  ARG_CHECK(q && k && v && o && dout && lse && delta && dq && dk && dv && strides15, "flash_attention_backward: null argument");
  ARG_CHECK(B > 0 && H > 0 && L > 0 && (Dh == 32 || Dh == 64), "flash_attention_backward: B=%d H=%d L=%d Dh=%d (Dh must be 32 or 64)", B, H, L, Dh);
  ARG_CHECK(drop_p >= 0.f && drop_p < 1.f, "flash_attention_backward: dropout %f", drop_p);
  ARG_CHECK((int64_t)B * H <= 65535, "flash_attention_backward: B*H = %lld exceeds the grid", (long long)B * H);
  ARG_CHECK(!bias || biasT, "flash_attention_backward: a score bias needs its [H][key][query] transpose too");
  for (int i = 0; i < 15; ++i) ARG_CHECK(strides15[i] % 4 == 0, "flash_attention_backward: stride %d = %lld is not a multiple of 4 elements (16-byte accesses)", i, (long long)strides15[i]);
  ARG_CHECK((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dout) & 15) == 0, "flash_attention_backward: tensors must be 16-byte aligned");
  FlashBwdArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o; a.dout = dout; a.lse = lse; a.delta = delta; a.mask_add = mask_add; a.bias = bias; a.biasT = biasT;
  a.dq = dq; a.dk = dk; a.dv = dv; a.ds_out = ds_out; a.B = B; a.H = H; a.L = L;
  a.q_sb = strides15[0]; a.q_sh = strides15[1]; a.q_sl = strides15[2];
  a.k_sb = strides15[3]; a.k_sh = strides15[4]; a.k_sl = strides15[5];
  a.v_sb = strides15[6]; a.v_sh = strides15[7]; a.v_sl = strides15[8];
  a.o_sb = strides15[9]; a.o_sh = strides15[10]; a.o_sl = strides15[11];
  a.g_sb = strides15[12]; a.g_sh = strides15[13]; a.g_sl = strides15[14];
  a.scale = scale; a.drop_p = drop_p; a.causal = causal; a.seed = seed; a.offset = offset;
  hipStream_t st = (hipStream_t)stream;
  const dim3 tiles(ceil_div(L, 64), B * H), rows(ceil_div(L, 256), B * H);
  if (ds_out) HIP_CHECK_RET(hipMemsetAsync(ds_out, 0, (size_t)B * H * L * L * sizeof(float), st));   // masked / causal entries stay zero
  if (Dh == 32) {
    hipLaunchKernelGGL(flash_delta_kernel<32>, rows, dim3(256), 0, st, a);
    hipLaunchKernelGGL(flash_bwd_dq_kernel<32>, tiles, dim3(256), 0, st, a);
    hipLaunchKernelGGL(flash_bwd_dkv_kernel<32>, tiles, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(flash_delta_kernel<64>, rows, dim3(256), 0, st, a);
    hipLaunchKernelGGL(flash_bwd_dq_kernel<64>, tiles, dim3(256), 0, st, a);
    hipLaunchKernelGGL(flash_bwd_dkv_kernel<64>, tiles, dim3(256), 0, st, a);
  }
  HIP_CHECK_RET(hipGetLastError());
  return MMSKIN_OK;
}

}  // extern "C"

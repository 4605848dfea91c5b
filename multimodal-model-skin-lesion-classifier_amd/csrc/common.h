// Shared device/host helpers for the mmskin gfx950 kernels.
// Activations inside the library are NHWC; T is float (exact f32 MFMA path, parity mode) or
// bf16_t (bf16 MFMA with f32 accumulate, throughput mode).  16 bytes = one "chunk" = the unit
// every loader moves per lane (8 bf16 or 4 f32); a K-tile row is always 128 bytes = 8 chunks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint16_t bf16_t;  // storage type for bf16 activations / staged weights
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

#define MMSKIN_OK 0
#define MMSKIN_ERR_ARG 1
#define MMSKIN_ERR_HIP 2
#define MMSKIN_ERR_UNSUPPORTED 3

void mmskin_set_error(const char* fmt, ...);

#define HIP_CHECK_RET(expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      mmskin_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return MMSKIN_ERR_HIP;                                                                 \
    }                                                                                        \
  } while (0)

#define ARG_CHECK(cond, ...)          \
  do {                                \
    if (!(cond)) {                    \
      mmskin_set_error(__VA_ARGS__);  \
      return MMSKIN_ERR_ARG;          \
    }                                 \
  } while (0)

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int EPC = 4;   // elements per 16-byte chunk
  static constexpr int BK = 32;   // elements per 128-byte K-tile row
  static constexpr int ID = 0;
};
template <> struct DT<bf16_t> {
  static constexpr int EPC = 8;
  static constexpr int BK = 64;
  static constexpr int ID = 1;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return (uint32_t)__builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return bf16_bits_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)f32_to_bf16_bits(v); }

// One 16-byte chunk viewed as EPC floats.
// splitmix64 finaliser: the counter-based generator of every dropout in the library (element i of a call = mix64(mix64(seed) ^ (offset + i)))
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// Attention-PROBABILITY dropout (nn.MultiheadAttention(dropout=p) of the TabTransformer, attention_probs_dropout_prob of BERT / GPT-2,
// attn_drop of timm blocks): keep / drop of probability (row, key) of one call, the same in every kernel that applies or regenerates it
// (fused forward, its backward, the unfused softmax-attention kernels).  Counter-based like the dropout op, but cheaper: the 64-bit
// splitmix finaliser per ELEMENT was 275 of the 586 us of the fused forward on the BERT shape (32 hashes per lane and key tile, two
// 64-bit multiplies each); here ONE 32-bit hash (three 32-bit multiplies) serves the key PAIR (key >> 1) of a row, 16 bits per element.
// row = flat (batch * head, query) index, Lh = (L + 1) / 2, thr = round(p * 65536) (drop rate exact to 2^-16; the survivors are scaled
// by 1 / (1 - p) as before).  k0 / k1 = the two halves of mix64(mix64(seed) ^ offset), computed once per kernel.
struct AttnDropKey { uint32_t k0, k1, thr; };
__host__ __device__ __forceinline__ AttnDropKey attn_drop_key(uint64_t seed, uint64_t offset, float drop_p) {
  const uint64_t k = mix64(mix64(seed) ^ offset);
  AttnDropKey a;
  a.k0 = (uint32_t)k; a.k1 = (uint32_t)(k >> 32);
  a.thr = (uint32_t)(drop_p * 65536.0f + 0.5f);
  return a;
}
// rbase = row * Lh (one 64-bit multiply per ROW, hoisted out of the key loops by the callers that walk a row)
__device__ __forceinline__ uint64_t attn_row_base(uint64_t row, int Lh) { return row * (uint64_t)Lh; }
__device__ __forceinline__ uint32_t attn_pair_hash(const AttnDropKey& a, uint64_t rbase, int key) {
  const uint64_t pair = rbase + (uint64_t)(key >> 1);
  uint32_t h = ((uint32_t)pair ^ a.k0) * 0x9E3779B1u;
  h ^= h >> 15;
  h = (h ^ (uint32_t)(pair >> 32) ^ a.k1) * 0x85EBCA77u;
  h ^= h >> 13;
  h *= 0xC2B2AE3Du;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool attn_keep_bits(uint32_t pair_hash, int key, uint32_t thr) { return ((pair_hash >> (16 * (key & 1))) & 0xffffu) >= thr; }
__device__ __forceinline__ bool attn_keep(const AttnDropKey& a, uint64_t rbase, int key) {
  return attn_keep_bits(attn_pair_hash(a, rbase, key), key, a.thr);
}

// x * Phi(x) with Phi(-|x|) = erfc(|x| / sqrt 2) / 2 from Abramowitz & Stegun 7.1.26 (|error of erf| <= 1.5e-7, i.e. <= 0.75e-7 |x|
// on the result: below bf16 AND below the fp32 rounding of an O(1) activation): 1 v_rcp + 1 v_exp + 9 multiply-adds, against the
// ~40-instruction erff() whose cost in a GEMM epilogue was a quarter of the launch (fc1 of BEiT-large: 377 us -> see r02_experiments (10)).
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float tail = 0.5f * poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);   // Phi(-|x|)
  return x * (x >= 0.f ? 1.f - tail : tail);
}

template <typename T> struct Chunk;
template <> struct Chunk<float> {
  float v[4];
  __device__ __forceinline__ void from_raw(const u32x4_t& t) {
    v[0] = __uint_as_float(t[0]); v[1] = __uint_as_float(t[1]); v[2] = __uint_as_float(t[2]); v[3] = __uint_as_float(t[3]);
  }
  __device__ __forceinline__ void load(const void* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(void* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  // streamed-once variants (nontemporal cache policy): keep L2 / Infinity Cache for data that is re-read
  __device__ __forceinline__ void load_nt(const void* p) {
    u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    from_raw(t);
  }
  __device__ __forceinline__ void store_nt(void* p) const {
    u32x4_t t = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
  }
};
template <> struct Chunk<bf16_t> {
  float v[8];
  __device__ __forceinline__ void from_raw(const u32x4_t& t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(t[i] << 16);
      v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void load(const void* p) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void store(void* p) const {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = f32_to_bf16_bits(v[2 * i]) | (f32_to_bf16_bits(v[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __device__ __forceinline__ void load_nt(const void* p) {
    u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    from_raw(t);
  }
  __device__ __forceinline__ void store_nt(void* p) const {
    u32x4_t t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = f32_to_bf16_bits(v[2 * i]) | (f32_to_bf16_bits(v[2 * i + 1]) << 16);
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
  }
};

// XCD-aware tile order: blocks are dealt round-robin over the 8 XCDs, so give each XCD a
// contiguous run of tile ids (bijective for any grid size).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// common.hpp — shared device/host helpers for libmst_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mst_hip.h"

namespace mst {

// ---------------------------------------------------------------- host side
void set_error(const char* fmt, ...);

#define MST_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      ::mst::set_error(__VA_ARGS__);             \
      return MST_ERR_INVALID;                    \
    }                                            \
  } while (0)

#define MST_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      ::mst::set_error("%s: %s", what, hipGetErrorString(e__));                  \
      return MST_ERR_LAUNCH;                                                     \
    }                                                                            \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t roundup(int64_t a, int64_t b) { return cdiv(a, b) * b; }

// -------------------------------------------------------------- device side
constexpr int WAVE = 64;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(8))) short i16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// 16-bit activation element traits
template <typename T> struct Act;
template <> struct Act<__bf16> {
  typedef bf16x8 vec8;
  typedef bf16x4 vec4;
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Act<_Float16> {
  typedef f16x8 vec8;
  typedef f16x4 vec4;
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// 16-byte vector of 8 act elements, viewed as raw bits
union Pack8 {
  u32x4 u;
  uint16_t h[8];
};

template <typename T> __device__ __forceinline__ float bits_to_f32(uint16_t b) {
  T t;
  __builtin_memcpy(&t, &b, 2);
  return (float)t;
}
template <typename T> __device__ __forceinline__ uint16_t f32_to_bits(float f) {
  T t = (T)f;
  uint16_t b;
  __builtin_memcpy(&b, &t, 2);
  return b;
}

// 8 uint8 values (piano-roll frame bytes) -> 8 activation elements, exact for 0..255 in both 16-bit types
template <typename T> __device__ __forceinline__ u32x4 expand_u8x8(u32x2 raw) {
  Pack8 p;
#pragma unroll
  for (int e = 0; e < 8; ++e) p.h[e] = f32_to_bits<T>((float)((raw[e >> 2] >> (8 * (e & 3))) & 0xFFu));
  return p.u;
}

// Cross-lane reductions over aligned groups of LANES lanes; every lane ends with the group's value. Levels 1..8 are DPP
// operands of the add itself (quad_perm, then row_half_mirror / row_mirror, which pair up the already-uniform quads / octets),
// level 16 is one ds_swizzle, level 32 one ds_bpermute. (As __shfl_xor butterflies every level was a ds_bpermute_b32 with
// its address arithmetic and LDS-crossbar latency: the LayerNorm epilogue of ffn_ln_kernel spent 7.5 of its 9.5 us in
// ten dependent ones per row.)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_xor16(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // swizzle(SWAP, 16)
}
template <int LANES> __device__ __forceinline__ float group_sum(float v) {
  static_assert(LANES == 4 || LANES == 8 || LANES == 16 || LANES == 32 || LANES == 64, "group size");
  v += dpp_mov<0xB1>(v);                              // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);                              // quad_perm [2,3,0,1]
  if constexpr (LANES >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
  if constexpr (LANES >= 16) v += dpp_mov<0x140>(v);  // row_mirror
  if constexpr (LANES >= 32) v += lane_xor16(v);
  if constexpr (LANES >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
template <int LANES> __device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  if constexpr (LANES >= 8) v = fmaxf(v, dpp_mov<0x141>(v));
  if constexpr (LANES >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
  if constexpr (LANES >= 32) v = fmaxf(v, lane_xor16(v));
  if constexpr (LANES >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }

// Work index of launch-order id `bid` when the n items are to sit in XCD-contiguous eighths: workgroup ids are dealt
// round-robin to the 8 XCDs, so XCD x receives ids x, x + 8, ... and takes items [x n / 8, (x + 1) n / 8). EVERY kernel
// that walks the rows of the activations uses this order (the GEMMs through gemm_mainloop's tile order): rows
// [x M / 8, (x + 1) M / 8) are produced and consumed by XCD x, so a launch finds its input rows in the L2 the previous
// launch left them in instead of fetching them across the fabric.
__device__ __forceinline__ int64_t xcd_chunk(int64_t bid, int64_t n) {
  const int64_t q = n / 8, r = n % 8, x = bid % 8, y = bid / 8;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
}

// logical → physical row remap (see mst_gemm_args)
__device__ __forceinline__ int64_t remap_row(int64_t m, int64_t rpg, int64_t stride, int64_t off) {
  return rpg > 0 ? (m / rpg) * stride + off + (m % rpg) : m;
}

// Counter-based RNG. dropout_hash (one 64-bit mix per draw) feeds the Gaussian eps; the dropout masks use the cheaper
// 32-bit mix below. Forward and backward regenerate a mask from (seed, site, index); nothing is stored.
__host__ __device__ __forceinline__ uint32_t dropout_hash(uint64_t seed, uint32_t site, uint64_t idx) {
  uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(site + 1)) ^ (idx * 0xD1B54A32D192ED03ull);
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return (uint32_t)x;
}
// Dropout keep decisions: TWO per 32-bit word (16 bits each), i.e. two words per group of four elements — element idx
// uses field (idx & 1) of word idx >> 1; keep iff field >= thr with thr = floor(p * 65536); the inverted-dropout scale
// uses the exact keep probability (65536 - thr) / 65536 (p = 0.2 -> 1.25000), so the mask is unbiased. A word is a
// multiplicative scramble of the index, the launch key, and one multiply-xorshift round (7 VALU per two elements; the
// former 64-bit two-round mix cost ~30 per four and was a quarter of the FFN1 epilogue).
// The per-launch key folds the 64-bit step seed and the site, so (seed, site, index) still identifies an element.
__host__ __device__ __forceinline__ uint32_t dropout_thr(float p) { return (uint32_t)(p * 65536.0f); }
__host__ __device__ __forceinline__ float dropout_inv_keep(float p) {
  return p > 0.f ? 65536.0f / (65536.0f - (float)dropout_thr(p)) : 1.f;
}
__host__ __device__ __forceinline__ uint32_t dropout_key(uint64_t seed, uint32_t site) {
  const uint64_t k = (seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(site + 1))) * 0xD6E8FEB86659FD93ull;
  return (uint32_t)(k >> 32) ^ (uint32_t)k;
}
__host__ __device__ __forceinline__ uint32_t dropout_word(uint32_t key, uint64_t idx2) {
  uint32_t x = ((uint32_t)idx2 * 0x9E3779B1u) ^ key ^ ((uint32_t)(idx2 >> 32) * 0x85EBCA6Bu);
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15;
  return x;
}
// the same word for an index below 2^32 (the high-half term of dropout_word vanishes)
__host__ __device__ __forceinline__ uint32_t dropout_word32(uint32_t key, uint32_t idx2) {
  uint32_t x = (idx2 * 0x9E3779B1u) ^ key;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15;
  return x;
}
__host__ __device__ __forceinline__ uint32_t dropout_keep4k32(uint32_t key, uint32_t idx4, uint32_t thr) {
  const uint32_t x0 = dropout_word32(key, 2 * idx4), x1 = dropout_word32(key, 2 * idx4 + 1);
  return (uint32_t)((x0 & 0xFFFFu) >= thr) | ((uint32_t)((x0 >> 16) >= thr) << 1) | ((uint32_t)((x1 & 0xFFFFu) >= thr) << 2) |
         ((uint32_t)((x1 >> 16) >= thr) << 3);
}
// bit e of the result = keep decision of element 4*idx4 + e
__host__ __device__ __forceinline__ uint32_t dropout_keep4k(uint32_t key, uint64_t idx4, uint32_t thr) {
  const uint32_t x0 = dropout_word(key, 2 * idx4), x1 = dropout_word(key, 2 * idx4 + 1);
  return (uint32_t)((x0 & 0xFFFFu) >= thr) | ((uint32_t)((x0 >> 16) >= thr) << 1) | ((uint32_t)((x1 & 0xFFFFu) >= thr) << 2) |
         ((uint32_t)((x1 >> 16) >= thr) << 3);
}
// The four decisions of dropout_keep4k applied in place: t[e] <- keep(4 idx4 + e) ? t[e] * inv_keep : 0. Same words, same fields; what
// it saves is packing the four comparisons into a nibble and testing its bits again (~15 of the ~58 vector instructions a group of four
// values costs in an epilogue — the FFN1 chunk epilogue is VALU-issue-bound: SQ_ACTIVE_INST_VALU is 42 % of that launch's SIMD cycles).
__host__ __device__ __forceinline__ void dropout_apply4(uint32_t key, uint64_t idx4, uint32_t thr, float inv_keep, float (&t)[4]) {
  const uint32_t x0 = dropout_word(key, 2 * idx4), x1 = dropout_word(key, 2 * idx4 + 1);
  t[0] = (x0 & 0xFFFFu) >= thr ? t[0] * inv_keep : 0.f;
  t[1] = (x0 >> 16) >= thr ? t[1] * inv_keep : 0.f;
  t[2] = (x1 & 0xFFFFu) >= thr ? t[2] * inv_keep : 0.f;
  t[3] = (x1 >> 16) >= thr ? t[3] * inv_keep : 0.f;
}
__host__ __device__ __forceinline__ void dropout_apply4_32(uint32_t key, uint32_t idx4, uint32_t thr, float inv_keep, float (&t)[4]) {
  const uint32_t x0 = dropout_word32(key, 2 * idx4), x1 = dropout_word32(key, 2 * idx4 + 1);
  t[0] = (x0 & 0xFFFFu) >= thr ? t[0] * inv_keep : 0.f;
  t[1] = (x0 >> 16) >= thr ? t[1] * inv_keep : 0.f;
  t[2] = (x1 & 0xFFFFu) >= thr ? t[2] * inv_keep : 0.f;
  t[3] = (x1 >> 16) >= thr ? t[3] * inv_keep : 0.f;
}
// ... with the index's multiplication done by the caller: premul = (2 idx4) * 0x9E3779B1 mod 2^32 (indices below 2^32). A thread that
// walks a tile visits indices base + constant, so premul is ONE multiplication plus additions of constants where dropout_word32 spends
// a quarter-rate v_mul_lo_u32 per word.
constexpr uint32_t DROPOUT_MUL = 0x9E3779B1u;
__host__ __device__ __forceinline__ void dropout_apply4_pre(uint32_t key, uint32_t premul, uint32_t thr, float inv_keep, float (&t)[4]) {
  uint32_t x0 = premul ^ key, x1 = (premul + DROPOUT_MUL) ^ key;
  x0 ^= x0 >> 16; x0 *= 0x7FEB352Du; x0 ^= x0 >> 15;
  x1 ^= x1 >> 16; x1 *= 0x7FEB352Du; x1 ^= x1 >> 15;
  t[0] = (x0 & 0xFFFFu) >= thr ? t[0] * inv_keep : 0.f;
  t[1] = (x0 >> 16) >= thr ? t[1] * inv_keep : 0.f;
  t[2] = (x1 & 0xFFFFu) >= thr ? t[2] * inv_keep : 0.f;
  t[3] = (x1 >> 16) >= thr ? t[3] * inv_keep : 0.f;
}
// ... and the same decisions as multipliers: k[e] = keep ? inv_keep : 0
__host__ __device__ __forceinline__ void dropout_scale4(uint32_t key, uint64_t idx4, uint32_t thr, float inv_keep, float (&k)[4]) {
  const uint32_t x0 = dropout_word(key, 2 * idx4), x1 = dropout_word(key, 2 * idx4 + 1);
  k[0] = (x0 & 0xFFFFu) >= thr ? inv_keep : 0.f;
  k[1] = (x0 >> 16) >= thr ? inv_keep : 0.f;
  k[2] = (x1 & 0xFFFFu) >= thr ? inv_keep : 0.f;
  k[3] = (x1 >> 16) >= thr ? inv_keep : 0.f;
}
__host__ __device__ __forceinline__ uint32_t dropout_keep4(uint64_t seed, uint32_t site, uint64_t idx4, float p) {
  return dropout_keep4k(dropout_key(seed, site), idx4, dropout_thr(p));
}
__host__ __device__ __forceinline__ bool dropout_keep(uint64_t seed, uint32_t site, uint64_t idx, float p) {
  return (dropout_keep4(seed, site, idx >> 2, p) >> (idx & 3)) & 1u;
}

template <typename F> static inline int dispatch_act(int dtype, F&& f) {
  if (dtype == MST_BF16) return f((__bf16)0);
  if (dtype == MST_F16) return f((_Float16)0);
  set_error("unsupported activation dtype %d (want MST_BF16 or MST_F16)", dtype);
  return MST_ERR_UNSUPPORTED;
}

}  // namespace mst

// attention.hip — MultiHeadDotAttention with the reference's KEY-ROW softmax, forward and backward.
//
// Reference arithmetic (VarAutoEncoder/transformer.py:85-126), per (batch b, head h):
//   L[k,q] = K[k]·Q[q] / sqrt(dh) + (keymask[k] ? 0 : -1e9)        :96-99,106-126
//   P[k,:] = softmax over q (the LAST axis of [B,H,T_K,T_Q])        :100
//   O[q]   = sum_k P[k,q] V[k]                                      :102 (transpose_a)
// Quirks reproduced on purpose: normalisation runs over queries, not keys; the padding mask adds
// the same -1e9 to a whole softmax row, so in fp32 a padded key row collapses to a (near-)uniform
// 1/S row instead of being excluded. The kernels add -1e9 in fp32 to the fp32 MFMA accumulator
// exactly as the reference does, in 16-bit modes too (no -inf, no NaN).
//
// Nothing S x S ever reaches HBM. Four kernels, all on v_mfma_f32_32x32x16 with the softmax axis
// chosen per kernel so reductions stay in-lane:
//   fwd_stats (key-owner, key on the lane)   : lse[k] = logsumexp_q L[k,q]        (online, in-lane)
//   fwd_out   (query-owner, query on the lane): O[q] += P^T V, P tile reused straight from the
//                                               accumulator registers as the next MFMA's A operand
//   bwd_kv    (key-owner)  : dV[k] = sum_q P dO[q];  delta[k] = sum_q P dP;  dK[k] = s * sum_q dL Q[q]
//   bwd_q     (query-owner): dQ[q] = s * sum_k dL[k,q] K[k],  dL = P * (dP - delta[k]),  dP = dO[q]·V[k]
// Operands that a product needs "transposed" (V, dO, Q, K as the B operand of an X^T·B product) are
// staged row-major in LDS and read with ds_read_tr16_b64.
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"

namespace mst {

// rows staged in LDS per step: the whole key (or query) range of a 256-long sequence in ONE stage, so a workgroup
// pays one exposed global-load latency and two barriers instead of four of each (the kernels are latency-bound:
// 60 % of wave cycles were waits with 64-row stages)
#ifndef MST_ATT_STAGE
#define MST_ATT_STAGE 64
#endif
template <int DH> struct Stage { static constexpr int ROWS = MST_ATT_STAGE; };
// LDS row stride of the staged tiles: +8 elements (16 bytes). With rows of exactly DH*2 = 32/64/128 bytes the
// 16-byte row-fragment reads of 16 different rows fall on 4 bank groups (4-way conflict, measured 57 % of LDS cycles
// in attn_bwd_kv); 80-byte rows put them on 16 distinct ones and leave the transposed reads at most 2-way.
template <int DH> struct LdsLd { static constexpr int V = DH + 8; };
constexpr int ATT_WG_ROWS = 128;  // owner rows per workgroup (4 waves x 32)
constexpr float MASK_VALUE = -1e9f;
constexpr float NEG_BIG = -3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;

// Softmax probabilities are evaluated as  P[k,q] = exp2(x * sk2[k] + ck2[k])  with per-key constants
//   unmasked key: sk2 = scale*log2(e),  ck2 = -(rowmax + log(rowsum)) * log2(e)
//   padded key  : sk2 = 0,              ck2 = -log(rowsum) * log2(e)
//   key >= S    : sk2 = 0,              ck2 = -inf                       (P = 0, no per-element guard)
// i.e. one FMA and one v_exp_f32 per element. For a padded key this is exact only while |x*scale| < 32, where
// fl(x*scale - 1e9) == -1e9 == rowmax and the logit cancels (transformer.py:111-125). Beyond that the reference's
// fp32 sum lands on another multiple of 64 and the row stops being uniform, so whenever a tile holds a padded key
// the kernels take the EXACT path instead (exact_prob: the reference's operation order in fp32); the fast form is
// used for tiles without padding (all of them in a full-length batch).
__device__ __forceinline__ void key_consts(bool in_range, bool valid_key, float rmax, float logl, float scale, float& sk2,
                                           float& ck2) {
  if (!in_range) { sk2 = 0.f; ck2 = -INFINITY; }
  else if (!valid_key) { sk2 = 0.f; ck2 = -logl * LOG2E; }
  else { sk2 = scale * LOG2E; ck2 = -(rmax + logl) * LOG2E; }
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// (the tiles of a bit set, lowest first: scalar loop control)
#define MST_FOR_TILES(kt, bits) for (uint64_t mst_m_ = (bits); mst_m_; mst_m_ &= mst_m_ - 1) if (const int kt = __builtin_ctzll(mst_m_); true)
__device__ __forceinline__ uint64_t tile_bits(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
// Which 32-key tiles of a sequence hold a PADDED key (bit t: keys 32 t .. 32 t + 31), from the per-key constants in LDS: a padded
// key has sk2 = 0 with a finite ck2 (a key beyond the sequence: ck2 = -inf, and its P = 0 needs no exact arithmetic). The
// query-owner phases take the exact form for those tiles only; a padded batch used to pay it for every tile of every sequence
// that holds a padded key at all (attention forward + backward at S 256, lengths uniform in [S/2, S]: +29 % over a full batch).
// Scalar (wave-uniform) result; sequences of up to 64 tiles (the resident kernels' LDS bounds them far below that).
__device__ __forceinline__ uint64_t padded_tile_mask(const float* sSk, const float* sCk, int n_tiles, int lane) {
  uint32_t lo = 0u, hi = 0u;
  for (int t = 0; t < n_tiles; ++t) {
    const int k = t * 32 + (lane & 31);
    const bool p = sSk[k] == 0.f && sCk[k] > -INFINITY;
    if (__any(p)) { if (t < 32) lo |= 1u << t; else hi |= 1u << (t - 32); }
  }
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
}
// the reference's arithmetic, step by step: t = fl(x*scale + madd); p = exp((t - rowmax) - log(rowsum))
__device__ __forceinline__ float exact_prob(float x, float scale, float madd, float rmax, float logl) {
  return fast_exp2(((fmaf(x, scale, madd) - rmax) - logl) * LOG2E);
}

__device__ __forceinline__ i16x4 att_tr_read(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(uintptr_t)p);
}

// Staging of rows [row0, row0+ATT_STAGE) x DH of a [S, ld] matrix is split in two so that it can be software-pipelined:
// stage_load issues the global loads into registers (zero for rows >= S), stage_store writes them to the LDS tile
// [ATT_STAGE][DH+8]. Every kernel below loads stage i+1 right after storing stage i, i.e. BEFORE it computes on
// stage i, so the L2 / Infinity-Cache latency of the next tile runs under the current tile's MFMA + exp work
// (measured before: 55-67 % of the wave cycles of these kernels were waits on exactly that latency, five exposed
// round trips per workgroup).
template <typename T, int DH> struct StageRegs {
  static constexpr int CPR = DH / 8;
  static constexpr int N = (Stage<DH>::ROWS * CPR + 255) / 256;
  u32x4 v[N];
};
template <typename T, int DH>
__device__ __forceinline__ void stage_load(StageRegs<T, DH>& r, const T* __restrict__ g, int64_t ld, int64_t row0, int64_t S, int tid) {
  constexpr int CPR = DH / 8;
#pragma unroll
  for (int i = 0; i < StageRegs<T, DH>::N; ++i) {
    const int c = tid + i * 256, row = c / CPR, ch = c % CPR;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (c < Stage<DH>::ROWS * CPR && row0 + row < S) v = *reinterpret_cast<const u32x4*>(g + (row0 + row) * ld + ch * 8);
    r.v[i] = v;
  }
}
template <typename T, int DH>
__device__ __forceinline__ void stage_store(T* lds, const StageRegs<T, DH>& r, int tid) {
  constexpr int CPR = DH / 8;
#pragma unroll
  for (int i = 0; i < StageRegs<T, DH>::N; ++i) {
    const int c = tid + i * 256, row = c / CPR, ch = c % CPR;
    if (c < Stage<DH>::ROWS * CPR) *reinterpret_cast<u32x4*>(lds + row * LdsLd<DH>::V + ch * 8) = r.v[i];
  }
}

// per-key scalars of one stage (thread t < ATT_STAGE holds key k0 + t)
struct KeyRegs { float rmax, logl, delta; bool in, valid; };
__device__ __forceinline__ void key_load(KeyRegs& kr, const uint8_t* __restrict__ keymask, const float* __restrict__ lse,
                                         const float* __restrict__ delta, int64_t plane, int64_t b, int64_t bh, int64_t S,
                                         int64_t k, bool active) {
  kr.in = active && k < S;
  kr.valid = kr.in && keymask[b * S + k];
  kr.rmax = kr.in ? lse[bh * S + k] : 0.f;
  kr.logl = kr.in ? lse[plane + bh * S + k] : 0.f;
  kr.delta = (kr.in && delta) ? delta[bh * S + k] : 0.f;
}

// Workgroup -> (row tile, batch*head). Linear workgroup ids are dealt round-robin to the 8 XCDs; when the batch is a
// multiple of 8 the ids are remapped so that every tile of every head of one batch element runs on the same XCD:
// the K / V rows that the tiles of a head share, and the other half of each 128-byte line (the neighbouring head),
// are then re-read from that XCD's L2 instead of crossing the fabric again.
__device__ __forceinline__ void attn_wg_coords(int64_t B, int64_t H, int& tile, int64_t& bh) {
  const uint32_t gx = gridDim.x;  // (32-bit quotients: grid.x * grid.y < 2^31 workgroups)
  if (B % 8 == 0) {
    const uint32_t lin = blockIdx.x + gx * blockIdx.y, xcd = lin % 8u, j = lin / 8u, G = (uint32_t)H * gx;
    const uint32_t jq = j / G, r = j - jq * G, rq = r / gx;
    bh = (int64_t)(xcd + 8u * jq) * H + rq;
    tile = (int)(r - rq * gx);
  } else {
    tile = blockIdx.x;
    bh = blockIdx.y;
  }
}

// row fragment (A or B operand, k-contiguous) of rows [r0, r0+32) from an LDS tile with DH columns
template <typename T, int DH>
__device__ __forceinline__ typename Act<T>::vec8 lds_row_frag(const T* tile, int r0, int s, int lane) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(tile + (r0 + (lane & 31)) * LdsLd<DH>::V + 16 * s + 8 * (lane >> 5));
  return __builtin_bit_cast(typename Act<T>::vec8, v);
}

// row fragment straight from global (owner rows, loaded once per wave); zero beyond S
template <typename T>
__device__ __forceinline__ typename Act<T>::vec8 glb_row_frag(const T* __restrict__ g, int64_t ld, int64_t row, int64_t S, int s,
                                                              int lane) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (row < S) v = *reinterpret_cast<const u32x4*>(g + row * ld + 16 * s + 8 * (lane >> 5));
  return __builtin_bit_cast(typename Act<T>::vec8, v);
}

// B operand of an X^T·B product where X is a 32x32 accumulator tile used as the A operand:
// element j of lane-half h must be row 16*s2 + 8*(j>>2) + 4*h + (j&3) of the staged tile, column d0 + (lane&31).
template <typename T, int DH>
__device__ __forceinline__ typename Act<T>::vec8 lds_tr_frag(const T* tile, int r0, int s2, int d0, int lane) {
  const int h = lane >> 5, i = lane & 15, q = i >> 2, p = i & 3;
  const int c0 = (DH >= 32) ? d0 + 16 * ((lane >> 4) & 1) : 0;
  const int rlo = r0 + 16 * s2 + 4 * h + q;
  const i16x4 lo = att_tr_read(tile + rlo * LdsLd<DH>::V + c0 + 4 * p);
  const i16x4 hi = att_tr_read(tile + (rlo + 8) * LdsLd<DH>::V + c0 + 4 * p);
  const i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(typename Act<T>::vec8, v);
}

template <typename T>
__device__ __forceinline__ typename Act<T>::vec8 acc_to_frag(const f32x16& x, int s2) {
  typename Act<T>::vec8 a;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (T)x[8 * s2 + j];
  return a;
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int DH>
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// Output tiles are accumulated "owner row on the lane": the accumulator of a 32-wide block of d holds rows d (registers)
// x owner rows (lane & 31) — the X-as-B-operand form of the accumulator-as-operand products below — so a lane holds
// elements d = 32 blk + 8 g + 4 (lane >> 5) + e (g, e = 0..3) of ITS row in registers 4 g + e: four 8-byte row pieces
// per block instead of sixteen 2-byte stores scattered over 16 rows.
template <typename T, int DH>
__device__ __forceinline__ void owner_store(T* __restrict__ row /* this lane's output row; nullptr: nothing to store */,
                                            const f32x16 (&acc)[(DH + 31) / 32], int lane) {
  if (!row) return;
  const int h4 = 4 * (lane >> 5);
#pragma unroll
  for (int d = 0; d < (DH + 31) / 32; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (32 * d + 8 * g < DH) {
        u32x2 o;
        o[0] = (uint32_t)f32_to_bits<T>(acc[d][4 * g]) | ((uint32_t)f32_to_bits<T>(acc[d][4 * g + 1]) << 16);
        o[1] = (uint32_t)f32_to_bits<T>(acc[d][4 * g + 2]) | ((uint32_t)f32_to_bits<T>(acc[d][4 * g + 3]) << 16);
        *reinterpret_cast<u32x2*>(row + 32 * d + 8 * g + h4) = o;
      }
}

// delta[k] = sum_q P[k,q] dP[k,q] with dP[k,q] = dO[q].V[k]  =  V[k] . (sum_q P[k,q] dO[q])  =  V[k] . dV[k]
// (the key-row analogue of flash attention's rowsum(dO o O)): one dot product per key from the finished dV accumulator
// instead of two MFMAs and sixteen FMAs per 32 x 32 tile of the dV sweep. acc is in owner_store's layout (key on the
// lane); vf is the key's V row as an MFMA fragment (lane half h' holds d = 16 u + 8 h' + j), so each lane half gets the
// four elements per 16 it lacks from its partner lane.
template <typename T, int DH>
__device__ __forceinline__ float delta_from_dv(const f32x16 (&acc)[(DH + 31) / 32], const typename Act<T>::vec8 (&vf)[DH / 16], int lane) {
  const bool hi_half = (lane >> 5) != 0;
  float dsum = 0.f;
#pragma unroll
  for (int u = 0; u < DH / 16; ++u) {
    const u32x4 w = __builtin_bit_cast(u32x4, vf[u]);
    const uint32_t s0 = hi_half ? w[0] : w[2], s1 = hi_half ? w[1] : w[3];
    const uint32_t r0 = (uint32_t)__shfl_xor((int)s0, 32, 64), r1 = (uint32_t)__shfl_xor((int)s1, 32, 64);
    const uint32_t e4[4] = {hi_half ? r0 : w[0], hi_half ? r1 : w[1], hi_half ? w[2] : r0, hi_half ? w[3] : r1};
    const int blk = u / 2, g = 2 * (u % 2);  // registers 4 g .. 4 g + 7 of block blk: d = 16 u + {4 h + e, 8 + 4 h + e}
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint16_t bits = (uint16_t)((e4[e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
      dsum = fmaf(acc[blk][4 * g + e], bits_to_f32<T>(bits), dsum);
    }
  }
  return dsum + __shfl_xor(dsum, 32, 64);
}

struct AttnArgs {
  int64_t B, S, H;
  const void* qkv; int64_t ld_qkv, k_off, q_off, v_off;
  const uint8_t* keymask;
  float* lse;
  void* out; int64_t ld_out;
  const void* dout; int64_t ld_dout;
  void* dqkv; int64_t ld_dqkv;
  float* delta;
  float scale;
  int64_t q_limit;  // forward: only queries [0, q_limit) are produced (the top encoder layer needs query 0 alone)
  // fused K | Q | V projection (attn_fwd_res_kernel<.., QKV = true>): qkv = x W^T + bias is COMPUTED here (and written to
  // `qkv` for the backward pass) instead of read. x: [B*S, ld_x] layer input; w: [3 Dm, ld_w] 16-bit weights whose row c is
  // output column c of the qkv layout; bias: fp32 [3 Dm]
  const void* x; int64_t ld_x; const void* w; int64_t ld_w; const float* bias; int64_t Dm;
  // resident forward with TWO staged tiles instead of three (sequences whose Q | K | V do not fit together: Q for the statistics phase,
  // then K and V over it for the output phase; the owners' own fragments come from global memory)
  int restage;
};

// Online softmax statistics of one 32-query x 32-key tile, key on the lane.
// stats_tile_exact keeps the reference's operation order, t = fl(x*scale + madd) then exp(t - max): needed when the
// wave holds a padded key (madd = -1e9 swallows the logit, see key_consts). stats_tile_fast is for waves without one
// (madd = 0 for every lane): it tracks the maximum of the raw products and works in the exp2 domain,
//   m2 = max(x) * c,  l += exp2(x * c - m2),  c = scale * log2(e)
// i.e. max, fma, v_exp, add per element instead of fma, select, max, sub, mul, v_exp, select, add.
// GUARD: the tile may hold query rows beyond the sequence (the last tile of a ragged sequence); a whole tile needs neither the
// sixteen 64-bit row comparisons nor the selects around the exponentials (207 -> 110 issue slots; same values).
template <typename T, int DH, bool GUARD>
__device__ __forceinline__ void stats_tile_exact(const T* sQ, int r0, int64_t q_base, int64_t S, float scale, float madd,
                                                 const typename Act<T>::vec8 (&kf)[DH / 16], float& m, float& l, int lane) {
  f32x16 x = zero16<DH>();
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) x = Act<T>::mfma32(lds_row_frag<T, DH>(sQ, r0, s, lane), kf[s], x);
  float t[16], tmax = NEG_BIG;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const bool valid = !GUARD || q_base + acc_row(r, lane) < S;
    t[r] = valid ? fmaf(x[r], scale, madd) : NEG_BIG;
    tmax = fmaxf(tmax, t[r]);
  }
  const float m_new = fmaxf(m, tmax);
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += (!GUARD || t[r] > NEG_BIG) ? __expf(t[r] - m_new) : 0.f;
  l = l * __expf(m - m_new) + sum;
  m = m_new;
}
template <typename T, int DH, bool GUARD>
__device__ __forceinline__ void stats_tile_fast(const T* sQ, int r0, int64_t q_base, int64_t S, float c,
                                                const typename Act<T>::vec8 (&kf)[DH / 16], float& m2, float& l, int lane) {
  f32x16 x = zero16<DH>();
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) x = Act<T>::mfma32(lds_row_frag<T, DH>(sQ, r0, s, lane), kf[s], x);
  if (GUARD) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (q_base + acc_row(r, lane) < S) ? x[r] : -INFINITY;
  }
  float tmax = x[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, x[r]);
  const float m_new = fmaxf(m2, tmax * c);
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += fast_exp2(fmaf(x[r], c, -m_new));
  l = l * fast_exp2(m2 - m_new) + sum;
  m2 = m_new;
}
// THIN tile: only the tile's first partner row exists (the lone row of a 32 n + 1 sequence, lone_row_shape): accumulator
// element 0 of the lower lane half; one exponential instead of sixteen.
template <typename T, int DH>
__device__ __forceinline__ void stats_tile_thin(const T* sQ, int r0, float c, const typename Act<T>::vec8 (&kf)[DH / 16], float& m2, float& l,
                                                int lane) {
  f32x16 x = zero16<DH>();
#pragma unroll
  for (int s = 0; s < DH / 16; ++s) x = Act<T>::mfma32(lds_row_frag<T, DH>(sQ, r0, s, lane), kf[s], x);
  const float x0 = (lane < 32) ? x[0] : -INFINITY;
  const float m_new = fmaxf(m2, x0 * c);
  l = l * fast_exp2(m2 - m_new) + fast_exp2(fmaf(x0, c, -m_new));
  m2 = m_new;
}
// Full statistics of the 32 keys on this wave's lanes over the query tiles [0, n_tiles) staged at sQ (tile t at rows
// 32 t, global query index q0 + 32 t): updates the running (max, sum) pair, natural-log domain on entry and exit.
constexpr float LN2 = 0.6931471805599453f;
template <typename T, int DH>
__device__ __forceinline__ void stats_sweep(const T* sQ, int n_tiles, int64_t q0, int64_t S, float scale, float madd, bool exact_w,
                                            const typename Act<T>::vec8 (&kf)[DH / 16], float& m, float& l, int lane, bool thin_tail = false) {
  if (exact_w) {
    for (int t = 0; t < n_tiles; ++t) {
      if (q0 + t * 32 + 32 <= S) stats_tile_exact<T, DH, false>(sQ, t * 32, q0 + t * 32, S, scale, madd, kf, m, l, lane);
      else stats_tile_exact<T, DH, true>(sQ, t * 32, q0 + t * 32, S, scale, madd, kf, m, l, lane);
    }
    return;
  }
  const float c = scale * LOG2E;
  float m2 = (m > NEG_BIG) ? m * LOG2E : NEG_BIG;
  for (int t = 0; t < n_tiles; ++t) {
    if (q0 + t * 32 + 32 <= S) stats_tile_fast<T, DH, false>(sQ, t * 32, q0 + t * 32, S, c, kf, m2, l, lane);
    else if (thin_tail) stats_tile_thin<T, DH>(sQ, t * 32, c, kf, m2, l, lane);
    else stats_tile_fast<T, DH, true>(sQ, t * 32, q0 + t * 32, S, c, kf, m2, l, lane);
  }
  m = (m2 > NEG_BIG) ? m2 * LN2 : NEG_BIG;
}

// ------------------------------------------------------------------------------------ fwd_stats
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_fwd_stats_kernel(AttnArgs a) {
  constexpr int ATT_STAGE = Stage<DH>::ROWS;
  constexpr int KS = DH / 16;
  __shared__ __attribute__((aligned(16))) T sQ[ATT_STAGE * LdsLd<DH>::V];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile; int64_t bh;
  attn_wg_coords(a.B, a.H, tile, bh);
  const int64_t b = (int64_t)((uint32_t)bh / (uint32_t)a.H), hd = bh - b * a.H;  // (B * H <= 65535: a 32-bit division, not the ~200-instruction 64-bit one, at the head of the kernel)
  const int64_t S = a.S;
  const T* base = reinterpret_cast<const T*>(a.qkv) + b * S * a.ld_qkv + hd * DH;
  const T* Kg = base + a.k_off;
  const T* Qg = base + a.q_off;
  const int64_t k_lane = (int64_t)tile * ATT_WG_ROWS + wave * 32 + (lane & 31);
  StageRegs<T, DH> rq;
  stage_load<T, DH>(rq, Qg, a.ld_qkv, 0, S, tid);

  typename Act<T>::vec8 kf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) kf[s] = glb_row_frag<T>(Kg, a.ld_qkv, k_lane, S, s, lane);
  const bool key_ok = k_lane < S && a.keymask[b * S + k_lane];
  const float madd = key_ok ? 0.f : MASK_VALUE;
  const bool exact_w = __any(!key_ok);  // a padded (or out-of-range) key on this wave: reference operation order

  float m = NEG_BIG, l = 0.f;
  for (int64_t q0 = 0; q0 < S; q0 += ATT_STAGE) {
    __syncthreads();
    stage_store<T, DH>(sQ, rq, tid);
    if (q0 + ATT_STAGE < S) stage_load<T, DH>(rq, Qg, a.ld_qkv, q0 + ATT_STAGE, S, tid);
    __syncthreads();
    const int64_t left = S - q0;
    stats_sweep<T, DH>(sQ, (int)((left < ATT_STAGE ? left : ATT_STAGE) + 31) / 32, q0, S, a.scale, madd, exact_w, kf, m, l, lane);
  }
  // the two lane halves hold disjoint query subsets of the same key
  const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64);
  const float M = fmaxf(m, m2);
  const float L = l * __expf(m - M) + l2 * __expf(m2 - M);
  // Stored as TWO planes (row max, log of the row sum): for a padded key row the max is -1e9 and a
  // single fp32 logsumexp would round log(S) away, turning the uniform 1/S row into ones.
  if (lane < 32 && k_lane < S) {
    a.lse[bh * S + k_lane] = M;
    a.lse[a.B * a.H * S + bh * S + k_lane] = __logf(L);
  }
}

// One 32-key x 32-query tile of the query-owner kernels. Straight-line on purpose: EXACT is a template parameter (the
// callers branch once per tile), so the compiler is free to issue the per-key constant reads ahead of the exps.
// (THIN: only the tile's first key exists — accumulator element 0; the other fifteen probabilities are zero)
template <typename T, int DH, bool EXACT, bool THIN = false>
__device__ __forceinline__ void fwd_out_tile(const T* sK, const T* sV, const float* sSk, const float* sCk, const float* sMadd,
                                             const float* sMax, const float* sLogl, int blk, float scale,
                                             const typename Act<T>::vec8 (&qf)[DH / 16], f32x16 (&o)[(DH + 31) / 32], int lane) {
  constexpr int KS = DH / 16, DB = (DH + 31) / 32;
  typename Act<T>::vec8 kfr[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) kfr[s] = lds_row_frag<T, DH>(sK, blk * 32, s, lane);
  typename Act<T>::vec8 vfr[2][DB];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int d = 0; d < DB; ++d) vfr[s2][d] = lds_tr_frag<T, DH>(sV, blk * 32, s2, d * 32, lane);
  f32x16 x = zero16<DH>();
#pragma unroll
  for (int s = 0; s < KS; ++s) x = Act<T>::mfma32(kfr[s], qf[s], x);
  if constexpr (THIN) {
    const int kr = blk * 32 + 4 * (lane >> 5);  // (upper lane half: a key beyond the sequence, its constants give p = 0)
    const float p0 = EXACT ? exact_prob(x[0], scale, sMadd[kr], sMax[kr], sLogl[kr]) : fast_exp2(fmaf(x[0], sSk[kr], sCk[kr]));
    x = zero16<DH>();
    x[0] = p0;
    const typename Act<T>::vec8 pf = acc_to_frag<T>(x, 0);  // (keys 8.. of the tile: all zero, no second product)
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = Act<T>::mfma32(vfr[0][d], pf, o[d]);
    return;
  }
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {  // accumulator rows 4g..4g+3 are 4 consecutive keys: one 16-byte read per constant
    const int kr = blk * 32 + 8 * g4 + 4 * (lane >> 5);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>((EXACT ? sMadd : sSk) + kr);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>((EXACT ? sMax : sCk) + kr);
    f32x4 c2 = c1;
    if (EXACT) c2 = *reinterpret_cast<const f32x4*>(sLogl + kr);  // +inf for keys >= S: p = 0
#pragma unroll
    for (int e = 0; e < 4; ++e)
      x[4 * g4 + e] = EXACT ? exact_prob(x[4 * g4 + e], scale, c0[e], c1[e], c2[e]) : fast_exp2(fmaf(x[4 * g4 + e], c0[e], c1[e]));
  }
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const typename Act<T>::vec8 pf = acc_to_frag<T>(x, s2);
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = Act<T>::mfma32(vfr[s2][d], pf, o[d]);  // O^T += V^T P: rows d, query on the lane
  }
}

// dQ tile. With sCk holding ck2 + log2(scale) (so the exponential IS P * scale) and the dP accumulator started at
// -delta[k] (sNd, one value per accumulator row), dL = P (dP - delta) scale is one multiply per element:
//   x = exp2(S sk2 + ck2') * (dO.V - delta);  acc^T += K^T-fragment x   (rows d, query on the lane)
// LIGHT: the owned queries' dO rows are zero (dP = 0): no V fragments, no dP MFMAs.
template <typename T, int DH, bool EXACT, bool LIGHT = false, bool THIN = false>
__device__ __forceinline__ void bwd_q_tile(const T* sK, const T* sV, const float* sSk, const float* sCk, const float* sMadd,
                                           const float* sMax, const float* sLogl, const float* sNd, int blk, float scale,
                                           const typename Act<T>::vec8 (&qf)[DH / 16], const typename Act<T>::vec8 (&dof)[DH / 16],
                                           f32x16 (&acc)[(DH + 31) / 32], int lane) {
  constexpr int KS = DH / 16, DB = (DH + 31) / 32;
  typename Act<T>::vec8 kfr[KS], vfr[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    kfr[s] = lds_row_frag<T, DH>(sK, blk * 32, s, lane);
    if (!LIGHT) vfr[s] = lds_row_frag<T, DH>(sV, blk * 32, s, lane);
  }
  typename Act<T>::vec8 ktr[2][DB];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int d = 0; d < DB; ++d) ktr[s2][d] = lds_tr_frag<T, DH>(sK, blk * 32, s2, d * 32, lane);
  f32x16 x = zero16<DH>(), dp;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {  // accumulator rows 4g..4g+3 are 4 consecutive keys: one 16-byte read
    const f32x4 nd = *reinterpret_cast<const f32x4*>(sNd + blk * 32 + 8 * g4 + 4 * (lane >> 5));
#pragma unroll
    for (int e = 0; e < 4; ++e) dp[4 * g4 + e] = nd[e];
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    x = Act<T>::mfma32(kfr[s], qf[s], x);
    if (!LIGHT) dp = Act<T>::mfma32(vfr[s], dof[s], dp);
  }
  if constexpr (THIN) {  // only the tile's first key exists (accumulator element 0)
    const int kr = blk * 32 + 4 * (lane >> 5);
    const float pr = EXACT ? exact_prob(x[0], scale, sMadd[kr], sMax[kr], sLogl[kr]) * scale : fast_exp2(fmaf(x[0], sSk[kr], sCk[kr]));
    const float d0 = pr * dp[0];
    x = zero16<DH>();
    x[0] = d0;
    const typename Act<T>::vec8 pf = acc_to_frag<T>(x, 0);
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[d] = Act<T>::mfma32(ktr[0][d], pf, acc[d]);
    return;
  }
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int kr = blk * 32 + 8 * g4 + 4 * (lane >> 5);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>((EXACT ? sMadd : sSk) + kr);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>((EXACT ? sMax : sCk) + kr);
    f32x4 c2 = c1;
    if (EXACT) c2 = *reinterpret_cast<const f32x4*>(sLogl + kr);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pr = EXACT ? exact_prob(x[4 * g4 + e], scale, c0[e], c1[e], c2[e]) * scale : fast_exp2(fmaf(x[4 * g4 + e], c0[e], c1[e]));
      x[4 * g4 + e] = pr * dp[4 * g4 + e];  // (LIGHT: dp is still -delta)
    }
  }
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const typename Act<T>::vec8 pf = acc_to_frag<T>(x, s2);
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[d] = Act<T>::mfma32(ktr[s2][d], pf, acc[d]);
  }
}

// One 32-query x 32-key tile of the key-owner backward, key on the lane throughout.
//   PASS 0: dV^T += dO^T-fragment P                     (delta follows from the finished dV: delta_from_dv)
//   PASS 1: dK^T += Q^T-fragment (P scale (dP - delta)): ck2s = ck2 + log2(scale) makes the exponential P * scale, and the
//           dP accumulator starts at -delta (the key's own value in every register), so dL is one multiply per element.
// sQ / sdO are the staged query-side operands (rows >= S are zero, so query rows beyond the sequence contribute nothing
// and need no guard).
// LIGHT (PASS 1 only): the tile's dO rows are all zero, so dP = 0 and dL = -P * delta * s — no dO fragments, no dP MFMAs;
// bit-identical to the full tile on zero dO rows (its accumulator stays at -delta).
template <typename T, int DH, int PASS, bool EXACT, bool LIGHT = false, bool THIN = false>
__device__ __forceinline__ void bwd_kv_tile(const T* sQ, const T* sdO, int qt, float scale, const typename Act<T>::vec8 (&kf)[DH / 16],
                                            const typename Act<T>::vec8 (&vf)[DH / 16], float sk2, float ck2x /* PASS 1: ck2s */, float madd,
                                            float rmax, float logl, float neg_delta, f32x16 (&acc)[(DH + 31) / 32], int lane) {
  constexpr int KS = DH / 16, DB = (DH + 31) / 32;
  static_assert(!LIGHT || PASS == 1, "a light tile contributes nothing to pass 0");
  constexpr bool NEED_DP = PASS == 1 && !LIGHT;
  typename Act<T>::vec8 qfr[KS], dofr[KS], trf[2][DB];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    qfr[s] = lds_row_frag<T, DH>(sQ, qt * 32, s, lane);
    if (NEED_DP) dofr[s] = lds_row_frag<T, DH>(sdO, qt * 32, s, lane);
  }
  const T* tr_src = (PASS == 0) ? sdO : sQ;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int d = 0; d < DB; ++d) trf[s2][d] = lds_tr_frag<T, DH>(tr_src, qt * 32, s2, d * 32, lane);
  f32x16 x = zero16<DH>(), dp;
#pragma unroll
  for (int r = 0; r < 16; ++r) dp[r] = neg_delta;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    x = Act<T>::mfma32(qfr[s], kf[s], x);
    if (NEED_DP) dp = Act<T>::mfma32(dofr[s], vf[s], dp);
  }
  if constexpr (THIN) {  // only the tile's first query exists (accumulator element 0; the upper lane half's row is a zero row of sQ / sdO)
    float p0 = EXACT ? exact_prob(x[0], scale, madd, rmax, logl) : fast_exp2(fmaf(x[0], sk2, ck2x));
    if (PASS == 1) p0 = (EXACT ? p0 * scale : p0) * dp[0];
    x = zero16<DH>();
    x[0] = p0;
    const typename Act<T>::vec8 pf = acc_to_frag<T>(x, 0);
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[d] = Act<T>::mfma32(trf[0][d], pf, acc[d]);
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (PASS == 0) x[r] = EXACT ? exact_prob(x[r], scale, madd, rmax, logl) : fast_exp2(fmaf(x[r], sk2, ck2x));
    else x[r] = (EXACT ? exact_prob(x[r], scale, madd, rmax, logl) * scale : fast_exp2(fmaf(x[r], sk2, ck2x))) * dp[r];
  }
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const typename Act<T>::vec8 pf = acc_to_frag<T>(x, s2);
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[d] = Act<T>::mfma32(trf[s2][d], pf, acc[d]);
  }
}

// ------------------------------------------------------------------------------------ fwd_out
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_fwd_out_kernel(AttnArgs a) {
  constexpr int ATT_STAGE = Stage<DH>::ROWS;
  constexpr int KS = DH / 16, DB = (DH + 31) / 32;
  __shared__ __attribute__((aligned(16))) T sK[ATT_STAGE * LdsLd<DH>::V];
  __shared__ __attribute__((aligned(16))) T sV[ATT_STAGE * LdsLd<DH>::V];
  __shared__ __attribute__((aligned(16))) float sSk[ATT_STAGE], sCk[ATT_STAGE], sMadd[ATT_STAGE], sMax[ATT_STAGE], sLogl[ATT_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile; int64_t bh;
  attn_wg_coords(a.B, a.H, tile, bh);
  const int64_t b = (int64_t)((uint32_t)bh / (uint32_t)a.H), hd = bh - b * a.H;  // (B * H <= 65535: a 32-bit division, not the ~200-instruction 64-bit one, at the head of the kernel)
  const int64_t S = a.S;
  const int64_t plane = a.B * a.H * S;
  const T* base = reinterpret_cast<const T*>(a.qkv) + b * S * a.ld_qkv + hd * DH;
  const T* Kg = base + a.k_off;
  const T* Qg = base + a.q_off;
  const T* Vg = base + a.v_off;
  const int64_t q_wave0 = (int64_t)tile * ATT_WG_ROWS + wave * 32;
  const int64_t q_lane = q_wave0 + (lane & 31);
  StageRegs<T, DH> rk, rv;
  KeyRegs kr;
  stage_load<T, DH>(rk, Kg, a.ld_qkv, 0, S, tid);
  stage_load<T, DH>(rv, Vg, a.ld_qkv, 0, S, tid);
  key_load(kr, a.keymask, a.lse, nullptr, plane, b, bh, S, tid, tid < ATT_STAGE);

  typename Act<T>::vec8 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) qf[s] = glb_row_frag<T>(Qg, a.ld_qkv, q_lane, S, s, lane);

  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) o[d] = zero16<DH>();
  const bool wave_active = q_wave0 < a.q_limit;  // inactive waves still stage tiles and meet the barriers

  for (int64_t k0 = 0; k0 < S; k0 += ATT_STAGE) {
    __syncthreads();
    stage_store<T, DH>(sK, rk, tid);
    stage_store<T, DH>(sV, rv, tid);
    int padded = 0;
    if (tid < ATT_STAGE) {
      key_consts(kr.in, kr.valid, kr.rmax, kr.logl, a.scale, sSk[tid], sCk[tid]);
      sMadd[tid] = kr.valid ? 0.f : MASK_VALUE; sMax[tid] = kr.rmax; sLogl[tid] = kr.in ? kr.logl : INFINITY;
      padded = kr.in && !kr.valid;
    }
    if (k0 + ATT_STAGE < S) {
      stage_load<T, DH>(rk, Kg, a.ld_qkv, k0 + ATT_STAGE, S, tid);
      stage_load<T, DH>(rv, Vg, a.ld_qkv, k0 + ATT_STAGE, S, tid);
      key_load(kr, a.keymask, a.lse, nullptr, plane, b, bh, S, k0 + ATT_STAGE + tid, tid < ATT_STAGE);
    }
    const bool exact = __syncthreads_or(padded);  // wave-uniform: does this stage hold a padded key?
    if (!wave_active) continue;
#pragma unroll
    for (int blk = 0; blk < ATT_STAGE / 32; ++blk) {
      if (k0 + blk * 32 >= S) break;
      if (exact) fwd_out_tile<T, DH, true>(sK, sV, sSk, sCk, sMadd, sMax, sLogl, blk, a.scale, qf, o, lane);
      else fwd_out_tile<T, DH, false>(sK, sV, sSk, sCk, sMadd, sMax, sLogl, blk, a.scale, qf, o, lane);
    }
  }
  T* og = reinterpret_cast<T*>(a.out) + b * S * a.ld_out + hd * DH;
  owner_store<T, DH>((q_lane < S && q_lane < a.q_limit) ? og + q_lane * a.ld_out : nullptr, o, lane);
}

// ------------------------------------------------------------------------------------ bwd_kv
template <typename T, int DH, bool SPARSE = false>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnArgs a) {
  constexpr int ATT_STAGE = Stage<DH>::ROWS;
  constexpr int KS = DH / 16, DB = (DH + 31) / 32;
  __shared__ __attribute__((aligned(16))) T sQ[ATT_STAGE * LdsLd<DH>::V];
  __shared__ __attribute__((aligned(16))) T sdO[ATT_STAGE * LdsLd<DH>::V];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile; int64_t bh;
  attn_wg_coords(a.B, a.H, tile, bh);
  const int64_t b = (int64_t)((uint32_t)bh / (uint32_t)a.H), hd = bh - b * a.H;  // (B * H <= 65535: a 32-bit division, not the ~200-instruction 64-bit one, at the head of the kernel)
  const int64_t S = a.S;
  const T* base = reinterpret_cast<const T*>(a.qkv) + b * S * a.ld_qkv + hd * DH;
  const T* Kg = base + a.k_off;
  const T* Qg = base + a.q_off;
  const T* Vg = base + a.v_off;
  const T* dOg = reinterpret_cast<const T*>(a.dout) + b * S * a.ld_dout + hd * DH;
  const int64_t k_wave0 = (int64_t)tile * ATT_WG_ROWS + wave * 32;
  const int64_t k_lane = k_wave0 + (lane & 31);
  StageRegs<T, DH> rq, rdo;
  stage_load<T, DH>(rq, Qg, a.ld_qkv, 0, S, tid);
  stage_load<T, DH>(rdo, dOg, a.ld_dout, 0, S, tid);

  typename Act<T>::vec8 kf[KS], vf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    kf[s] = glb_row_frag<T>(Kg, a.ld_qkv, k_lane, S, s, lane);
    vf[s] = glb_row_frag<T>(Vg, a.ld_qkv, k_lane, S, s, lane);
  }
  float sk2, ck2, madd, rmax, logl;
  bool exact;
  {
    const bool in = k_lane < S;
    const bool vk = in && a.keymask[b * S + k_lane];
    rmax = in ? a.lse[bh * S + k_lane] : 0.f;
    logl = in ? a.lse[a.B * a.H * S + bh * S + k_lane] : INFINITY;
    madd = vk ? 0.f : MASK_VALUE;
    key_consts(in, vk, rmax, in ? logl : 0.f, a.scale, sk2, ck2);
    exact = __any(in && !vk);  // wave-uniform: one of this wave's 32 keys is padded
  }
  const float ck2s = ck2 + __log2f(a.scale);  // pass 1: the exponential is P * scale

  f32x16 acc[DB];
  float neg_delta = 0.f;
  T* const drow = (k_lane < S) ? reinterpret_cast<T*>(a.dqkv) + (b * S + k_lane) * a.ld_dqkv + hd * DH : nullptr;
  // Sparse mode (0 < q_limit <= 32: dO is zero from row q_limit on — the top encoder layer, as in attn_bwd_res_kernel): dV and delta
  // need the first query tile alone (pass 0 is one stage, one tile), every other tile of dK is the LIGHT form (dP = 0). The dense
  // kernel spent the same 139 us on this layer as on the one below it.
  constexpr bool sparse = SPARSE;  // host: 0 < q_limit <= 32 (a separate instantiation: as run-time branches the dense kernel paid 18 us for them)
  // pass 0: dV and delta; pass 1: dK
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[d] = zero16<DH>();
    const int64_t Sp = (sparse && pass == 0 && S > ATT_STAGE) ? ATT_STAGE : S;  // query rows this pass sweeps
    for (int64_t q0 = 0; q0 < Sp; q0 += ATT_STAGE) {
      __syncthreads();
      stage_store<T, DH>(sQ, rq, tid);
      stage_store<T, DH>(sdO, rdo, tid);
      {  // next stage (wrapping to the first one for the second pass)
        const int64_t qn = (q0 + ATT_STAGE < Sp) ? q0 + ATT_STAGE : 0;
        if (qn != 0 || pass == 0) {
          stage_load<T, DH>(rq, Qg, a.ld_qkv, qn, S, tid);
          stage_load<T, DH>(rdo, dOg, a.ld_dout, qn, S, tid);
        }
      }
      __syncthreads();
#pragma unroll
      for (int blk = 0; blk < ATT_STAGE / 32; ++blk) {
        if (q0 + blk * 32 >= S) break;
        // (query rows >= S need no guard: their staged Q and dO rows are zero)
        if (pass == 0) {
          if (sparse && (q0 > 0 || blk > 0)) break;  // (their dO rows are zero: nothing for dV)
          if (exact) bwd_kv_tile<T, DH, 0, true>(sQ, sdO, blk, a.scale, kf, vf, sk2, ck2, madd, rmax, logl, neg_delta, acc, lane);
          else bwd_kv_tile<T, DH, 0, false>(sQ, sdO, blk, a.scale, kf, vf, sk2, ck2, madd, rmax, logl, neg_delta, acc, lane);
        } else if (sparse && (q0 > 0 || blk > 0)) {
          if (exact) bwd_kv_tile<T, DH, 1, true, true>(sQ, sdO, blk, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
          else bwd_kv_tile<T, DH, 1, false, true>(sQ, sdO, blk, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
        } else {
          if (exact) bwd_kv_tile<T, DH, 1, true>(sQ, sdO, blk, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
          else bwd_kv_tile<T, DH, 1, false>(sQ, sdO, blk, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
        }
      }
    }
    if (pass == 0) {
      const float delta = delta_from_dv<T, DH>(acc, vf, lane);
      if (lane < 32 && k_lane < S) a.delta[bh * S + k_lane] = delta;
      neg_delta = -delta;
    }
    owner_store<T, DH>(drow ? drow + (pass == 0 ? a.v_off : a.k_off) : nullptr, acc, lane);
  }
}

// ------------------------------------------------------------------------------------ bwd_q
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnArgs a) {
  constexpr int ATT_STAGE = Stage<DH>::ROWS;
  constexpr int KS = DH / 16, DB = (DH + 31) / 32;
  __shared__ __attribute__((aligned(16))) T sK[ATT_STAGE * LdsLd<DH>::V];
  __shared__ __attribute__((aligned(16))) T sV[ATT_STAGE * LdsLd<DH>::V];
  __shared__ __attribute__((aligned(16))) float sSk[ATT_STAGE], sCk[ATT_STAGE], sNd[ATT_STAGE], sMadd[ATT_STAGE], sMax[ATT_STAGE], sLogl[ATT_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tile; int64_t bh;
  attn_wg_coords(a.B, a.H, tile, bh);
  const int64_t b = (int64_t)((uint32_t)bh / (uint32_t)a.H), hd = bh - b * a.H;  // (B * H <= 65535: a 32-bit division, not the ~200-instruction 64-bit one, at the head of the kernel)
  const int64_t S = a.S;
  const int64_t plane = a.B * a.H * S;
  const T* base = reinterpret_cast<const T*>(a.qkv) + b * S * a.ld_qkv + hd * DH;
  const T* Kg = base + a.k_off;
  const T* Qg = base + a.q_off;
  const T* Vg = base + a.v_off;
  const T* dOg = reinterpret_cast<const T*>(a.dout) + b * S * a.ld_dout + hd * DH;
  const int64_t q_wave0 = (int64_t)tile * ATT_WG_ROWS + wave * 32;
  const int64_t q_lane = q_wave0 + (lane & 31);
  const float log2_scale = __log2f(a.scale);
  StageRegs<T, DH> rk, rv;
  KeyRegs kr;
  stage_load<T, DH>(rk, Kg, a.ld_qkv, 0, S, tid);
  stage_load<T, DH>(rv, Vg, a.ld_qkv, 0, S, tid);
  key_load(kr, a.keymask, a.lse, a.delta, plane, b, bh, S, tid, tid < ATT_STAGE);

  typename Act<T>::vec8 qf[KS], dof[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    qf[s] = glb_row_frag<T>(Qg, a.ld_qkv, q_lane, S, s, lane);
    dof[s] = glb_row_frag<T>(dOg, a.ld_dout, q_lane, S, s, lane);
  }
  f32x16 acc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) acc[d] = zero16<DH>();

  for (int64_t k0 = 0; k0 < S; k0 += ATT_STAGE) {
    __syncthreads();
    stage_store<T, DH>(sK, rk, tid);
    stage_store<T, DH>(sV, rv, tid);
    int padded = 0;
    if (tid < ATT_STAGE) {
      float sk2, ck2;
      key_consts(kr.in, kr.valid, kr.rmax, kr.logl, a.scale, sk2, ck2);
      sSk[tid] = sk2; sCk[tid] = ck2 + log2_scale;  // the exponential is P * scale (bwd_q_tile)
      sMadd[tid] = kr.valid ? 0.f : MASK_VALUE; sMax[tid] = kr.rmax; sLogl[tid] = kr.in ? kr.logl : INFINITY;
      sNd[tid] = -kr.delta;
      padded = kr.in && !kr.valid;
    }
    if (k0 + ATT_STAGE < S) {
      stage_load<T, DH>(rk, Kg, a.ld_qkv, k0 + ATT_STAGE, S, tid);
      stage_load<T, DH>(rv, Vg, a.ld_qkv, k0 + ATT_STAGE, S, tid);
      key_load(kr, a.keymask, a.lse, a.delta, plane, b, bh, S, k0 + ATT_STAGE + tid, tid < ATT_STAGE);
    }
    const bool exact = __syncthreads_or(padded);
#pragma unroll
    for (int blk = 0; blk < ATT_STAGE / 32; ++blk) {
      if (k0 + blk * 32 >= S) break;
      if (exact) bwd_q_tile<T, DH, true>(sK, sV, sSk, sCk, sMadd, sMax, sLogl, sNd, blk, a.scale, qf, dof, acc, lane);
      else bwd_q_tile<T, DH, false>(sK, sV, sSk, sCk, sMadd, sMax, sLogl, sNd, blk, a.scale, qf, dof, acc, lane);
    }
  }
  T* dst = reinterpret_cast<T*>(a.dqkv) + b * S * a.ld_dqkv + hd * DH + a.q_off;
  owner_store<T, DH>(q_lane < S ? dst + q_lane * a.ld_dqkv : nullptr, acc, lane);
}

// ------------------------------------------------------------------------------------ resident kernels
// One workgroup per (batch, head) with the whole sequence resident in LDS: the two kernels of each direction become
// two PHASES of one launch (every launch in the captured step costs ~4.7 us before its first wave does useful work,
// and the streaming kernels above re-stage the other operand once per 128 owner rows). Rows are owned in blocks of
// 32 by the waves of the workgroup round-robin (block ob -> wave ob % NW), first as keys (phase A: lane = key, the
// softmax axis in-lane), then as queries (phase B: lane = query, contraction over keys from the accumulators). The
// tile arithmetic is the streaming kernels' own (same functions, same operation order), so results are identical.
// Used when the staged operands fit in LDS (choose_resident below); longer sequences take the streaming kernels.
template <typename T, int DH>
__device__ __forceinline__ void stage_all(T* lds, const T* __restrict__ g, int64_t ld, int64_t S, int SP, int tid, int nthr) {
  constexpr int CPR = DH / 8;
#pragma unroll 4
  for (int c = tid; c < SP * CPR; c += nthr) {
    const int row = c / CPR, ch = c % CPR;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < S) v = *reinterpret_cast<const u32x4*>(g + (int64_t)row * ld + ch * 8);
    *reinterpret_cast<u32x4*>(lds + row * LdsLd<DH>::V + ch * 8) = v;
  }
}

// Two tensors at once: every global load of both is issued before the first LDS store (stage_all twice is two exposed
// round trips: the second tensor's loads wait behind the first one's stores)
template <typename T, int DH>
__device__ __forceinline__ void stage_pair(T* ldsA, const T* __restrict__ gA, int64_t ldA, int64_t rowsA, T* ldsB,
                                           const T* __restrict__ gB, int64_t ldB, int64_t rowsB, int SP, int tid, int nthr) {
  constexpr int CPR = DH / 8, BATCH = 4;
  for (int c0 = tid; c0 < SP * CPR; c0 += nthr * BATCH) {
    u32x4 va[BATCH], vb[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int c = c0 + i * nthr, row = c / CPR, ch = c % CPR;
      va[i] = u32x4{0u, 0u, 0u, 0u}; vb[i] = va[i];
      if (c < SP * CPR && row < rowsA) va[i] = *reinterpret_cast<const u32x4*>(gA + (int64_t)row * ldA + ch * 8);
      if (c < SP * CPR && row < rowsB) vb[i] = *reinterpret_cast<const u32x4*>(gB + (int64_t)row * ldB + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int c = c0 + i * nthr, row = c / CPR, ch = c % CPR;
      if (c < SP * CPR) {
        *reinterpret_cast<u32x4*>(ldsA + row * LdsLd<DH>::V + ch * 8) = va[i];
        *reinterpret_cast<u32x4*>(ldsB + row * LdsLd<DH>::V + ch * 8) = vb[i];
      }
    }
  }
}

// ---- the LONE ROW of a sequence of 32 n + 1 rows (the decoder of configs[1]: 257 = the state row + 256 positions).
// As a ninth 32-row owner block it costs a whole wave's sweep over all nine partner tiles for one useful lane — 81 tile
// visits per sweep instead of 64, and nine waves per workgroup leave room for one workgroup per CU instead of two. As a
// PARTNER the row is no problem (the ninth partner tile is an ordinary partial tile). As an OWNER it is handled here with
// the roles turned over: the partner rows go on the lanes, each lane forms its row's dot products with the lone row
// in-lane (rows are read straight from the staged LDS tiles), and what the MFMA form gets by summing down the register
// axis becomes one cross-lane sum per output element at the end of the sweep — ~300 VALU operations per wave instead of
// ~1200, and the owner blocks left are whole: eight waves, two workgroups per CU, one resident round.
template <typename T, int DH>
__device__ __forceinline__ void lds_row_load(const T* row, float (&v)[DH]) {
#pragma unroll
  for (int c = 0; c < DH / 8; ++c) {
    Pack8 p; p.u = *reinterpret_cast<const u32x4*>(row + 8 * c);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[8 * c + e] = bits_to_f32<T>(p.h[e]);
  }
}
template <typename T, int DH>
__device__ __forceinline__ float lds_row_dot(const T* row, const float (&w)[DH]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < DH / 8; ++c) {
    Pack8 p; p.u = *reinterpret_cast<const u32x4*>(row + 8 * c);
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(bits_to_f32<T>(p.h[e]), w[8 * c + e], s);
  }
  return s;
}
template <typename T, int DH>
__device__ __forceinline__ void lds_row_axpy(float alpha, const T* row, float (&acc)[DH]) {
#pragma unroll
  for (int c = 0; c < DH / 8; ++c) {
    Pack8 p; p.u = *reinterpret_cast<const u32x4*>(row + 8 * c);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[8 * c + e] = fmaf(alpha, bits_to_f32<T>(p.h[e]), acc[8 * c + e]);
  }
}
constexpr int LONE_RED = 16 * 32;  // floats of cross-wave reduction scratch behind the per-key arrays
__host__ __device__ inline bool lone_row_shape(int64_t S, int dh) { return dh <= 16 && S > 32 && (S & 31) == 1; }  // (head size 32: the three fp32 rows spill the dense kernel)

// waves per SIMD the head-size-16 resident kernels are compiled for (5 = 96 VGPRs, two 9-wave workgroups per CU, measured no faster forward and spills backward: the kernels are VALU-bound, not occupancy-bound)
#ifndef MST_ATT16_WAVES_FWD
#define MST_ATT16_WAVES_FWD 4
#endif
#ifndef MST_ATT16_WAVES_BWD
#define MST_ATT16_WAVES_BWD 4
#endif
// Waves per workgroup the resident kernels are compiled for. Head size 64 carries twice the fragments and accumulators per
// wave: under 1024-thread bounds (128 registers per lane) its kernels spilled 46-162 registers; eight waves (256 registers,
// two per SIMD) hold everything, and three 64-wide tiles of a sequence leave room for one workgroup per CU anyway.
template <int DH> constexpr int RES_MAX_WAVES = DH == 64 ? 8 : 16;
// (the plain forward kernel fits 118 registers at head size 64 and keeps sixteen waves — a 257- or 512-row sequence wants more
// than eight owners —; its chunked and fused-projection forms are the ones that spilled)
template <int DH, bool HEAVY> constexpr int FWD_MAX_WAVES = HEAVY ? RES_MAX_WAVES<DH> : 16;

// (batch*head) of a 1-D grid; batch elements are dealt to the XCDs so that the heads of one element share an L2
#ifndef MST_XCD_ROWS
#define MST_XCD_ROWS 1
#endif
__device__ __forceinline__ int64_t res_wg_bh(int64_t B, int64_t H) {
  const int64_t lin = blockIdx.x;
  // (batch, head) pairs in XCD-contiguous eighths: the samples whose K | Q | V rows the projection GEMM's tiles left in
  // this XCD's L2 (common.hpp xcd_chunk; the heads of a sample stay together)
  if (MST_XCD_ROWS) return xcd_chunk(lin, B * H);
  if (B % 8 != 0) return lin;
  const int64_t xcd = lin % 8, j = lin / 8;
  return (xcd + 8 * (j / H)) * H + j % H;
}

// ---- the K | Q | V projection of ONE (batch, head) inside the attention launch (head size 32). The Dense layers' GEMM
// (transformer.py:88-93: three Dense(D -> D) on the same input, run as one [3 D, D] product) has exactly this workgroup's
// operands as one of its tiles: rows = the sample's S positions, columns = the head's 32 columns of K, of Q and of V. As a
// launch of its own it cost 17.6 us and wrote 25 MB that the attention launch read straight back; here every wave forms the
// three 32 x 32 tiles of its 32 rows (x fragments straight from global, the head's 96 weight rows staged through LDS in
// 32-deep slices shared by the waves), adds the bias, rounds once to the activation type and leaves the SAME bits in the
// staged LDS tiles and in `qkv` (the backward pass recomputes the probabilities from what is stored).
// Accumulators are [feature, row-on-lane] (weights = A operand, x = B operand): owner_store's layout.
// diagnostic build only (-DMST_ATT_STAMPS): realtime stamps (100 MHz) of the forward kernel's phases, per workgroup
#ifdef MST_ATT_STAMPS
__device__ uint64_t g_att_stamps[1024 * 8];
#define ATT_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_att_stamps[8 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ uint64_t g_att_loop[4 * 64];  // workgroups 0, 100, 300, 500: s_memtime stamps inside the projection loop
#define LOOP_STAMP(k) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == 300 || blockIdx.x == 500) && (k) < 64) \
    g_att_loop[(blockIdx.x == 0 ? 0 : blockIdx.x == 100 ? 1 : blockIdx.x == 300 ? 2 : 3) * 64 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ATT_STAMP(k) do { } while (0)
#define LOOP_STAMP(k) do { } while (0)
#endif
constexpr int QKV_KC = 64;                         // contraction slice staged per step
constexpr int QKV_LDS = QKV_KC + 8;                // row stride (elements) of the staged slices: 144-byte rows, conflict-free fragments
constexpr int QKV_WROWS = 3 * 32;                  // the head's weight rows: 32 of K, of Q and of V
// The slices are staged in the LDS the attention phases use afterwards for the Q / K / V tiles themselves (free until the
// projection's epilogue writes them): x slice [SP][72] over the Q and K tiles, weight slice [96][72] over the V tile. Both are
// fetched with whole-line loads (8 lanes x 16 bytes per row) through registers one slice ahead. A first form read every wave's x
// fragments straight from global — 32 rows x 32 bytes per instruction, i.e. 32 cache lines touched per load — and its 8-slice
// loop took 10.9 us per workgroup for 0.6 us of MFMAs per wave, bound by the CU's address / tag pipeline.
template <typename T>
__device__ __forceinline__ void qkv_prologue(const AttnArgs& a, T* sQ, T* sK, T* sV, int64_t b, int64_t hd, int NB) {
  constexpr int DH = 32, LD = LdsLd<DH>::V;
  static_assert(2 * LD >= QKV_LDS && QKV_WROWS * QKV_LDS <= 256 * LD, "the staged slices must fit the tiles they borrow");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int64_t S = a.S;
  const int SP = NB * 32;
  T* const sX = sQ;   // [SP][72]: spans the Q and K tiles (contiguous)
  T* const sW = sV;   // [96][72]
  const T* xg = reinterpret_cast<const T*>(a.x) + b * S * a.ld_x;
  const T* wg = reinterpret_cast<const T*>(a.w);
  const int nch = (int)(a.Dm / QKV_KC);
  const int64_t sec_off[3] = {a.k_off, a.q_off, a.v_off};
  const bool active = wave < NB;
  const int64_t row = (int64_t)wave * 32 + (lane & 31);
  // accumulators start at the bias: feature (8 g + 4 h + e) of the section on the register axis (owner_store's layout)
  const int h4 = 4 * (lane >> 5);
  f32x16 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + sec_off[c] + hd * DH + 8 * g + h4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[c][4 * g + e] = bv[e];
    }
  // staging pieces of 16 bytes: x piece p = (row p / 8, chunk p % 8) for p < SP * 8, four per thread at most (512 threads, SP <= 256);
  // weight piece p < 768 likewise, two per thread. Rows beyond the sequence read the last row (their products are dropped below);
  // every load is unconditional (a conditional load costs a vmcnt(0) drain at the join): surplus pieces re-read piece 0.
  constexpr int XP = 4, WP = 2;
  const T* xsrc[XP]; T* xdst[XP]; bool xok[XP];
  const T* wsrc[WP]; T* wdst[WP]; bool wok[WP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int p = tid + i * nthr;
    xok[i] = p < SP * 8;
    const int r = xok[i] ? p >> 3 : 0, ch = xok[i] ? p & 7 : 0;
    xsrc[i] = xg + (int64_t)(r < S ? r : S - 1) * a.ld_x + ch * 8;
    xdst[i] = sX + r * QKV_LDS + ch * 8;
  }
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int p = tid + i * nthr;
    wok[i] = p < QKV_WROWS * 8;
    const int r = wok[i] ? p >> 3 : 0, ch = wok[i] ? p & 7 : 0;
    wsrc[i] = wg + (sec_off[r >> 5] + hd * DH + (r & 31)) * a.ld_w + ch * 8;
    wdst[i] = sW + r * QKV_LDS + ch * 8;
  }
  u32x4 xr[XP], wr[WP];
  auto load_slice = [&](int kc) {
    const int k = (kc < nch ? kc : nch - 1) * QKV_KC;
#pragma unroll
    for (int i = 0; i < XP; ++i) xr[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + k);
#pragma unroll
    for (int i = 0; i < WP; ++i) wr[i] = *reinterpret_cast<const u32x4*>(wsrc[i] + k);
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int i = 0; i < XP; ++i)
      if (xok[i]) *reinterpret_cast<u32x4*>(xdst[i]) = xr[i];
#pragma unroll
    for (int i = 0; i < WP; ++i)
      if (wok[i]) *reinterpret_cast<u32x4*>(wdst[i]) = wr[i];
  };
  LOOP_STAMP(0);
  load_slice(0);
  store_slice();
  __syncthreads();
  LOOP_STAMP(1);
  const T* const fx = sX + (wave * 32 + (lane & 31)) * QKV_LDS + 8 * (lane >> 5);
  const T* const fw = sW + (lane & 31) * QKV_LDS + 8 * (lane >> 5);
  // (Measured and not kept: the next slice's loads interleaved between the MFMAs — the loop is bound by LDS fragment traffic,
  // 16 KB per wave and slice of which 12 are the weight fragments every wave re-reads, next to 44 KB per slice through the CU's
  // 64-byte-per-clock load path: 9.3 -> 9.5 us either way for 3.4 us of MFMAs per SIMD.)
  for (int kc = 0; kc < nch; ++kc) {
    load_slice(kc + 1);  // (past the end: the last slice again)
    LOOP_STAMP(2 + 5 * kc);
#pragma unroll
    for (int s4 = 0; s4 < QKV_KC / 16; ++s4) {
      const typename Act<T>::vec8 xv = __builtin_bit_cast(typename Act<T>::vec8, *reinterpret_cast<const u32x4*>(fx + 16 * s4));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(fw + c * 32 * QKV_LDS + 16 * s4);
        acc[c] = Act<T>::mfma32(__builtin_bit_cast(typename Act<T>::vec8, v), xv, acc[c]);
      }
    }
    LOOP_STAMP(3 + 5 * kc);
    __syncthreads();  // every wave has read this slice
    LOOP_STAMP(4 + 5 * kc);
    store_slice();
    LOOP_STAMP(5 + 5 * kc);
    __syncthreads();
    LOOP_STAMP(6 + 5 * kc);
  }
  ATT_STAMP(1);
  if (!active) return;
  // epilogue: one rounding, the same bits to the LDS tiles and to HBM. Nothing between the stores: a first form loaded each bias
  // vector right before its use, and since stores count in vmcnt every one of those loads waited for the previous store's
  // acknowledgement (12 dependent round trips, 6.9 us per workgroup; 3.4 with the bias in the accumulators from the start).
  const bool in = row < S;
  T* const tiles[3] = {sK, sQ, sV};
  T* qrow = reinterpret_cast<T*>(const_cast<void*>(a.qkv)) + (b * S + row) * a.ld_qkv + hd * DH;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2 o = {0u, 0u};
      if (in) {
        o[0] = (uint32_t)f32_to_bits<T>(acc[c][4 * g]) | ((uint32_t)f32_to_bits<T>(acc[c][4 * g + 1]) << 16);
        o[1] = (uint32_t)f32_to_bits<T>(acc[c][4 * g + 2]) | ((uint32_t)f32_to_bits<T>(acc[c][4 * g + 3]) << 16);
        *reinterpret_cast<u32x2*>(qrow + sec_off[c] + 8 * g + h4) = o;
      }
      *reinterpret_cast<u32x2*>(tiles[c] + row * LD + 8 * g + h4) = o;  // (rows beyond the sequence: zeros, as stage_all leaves them)
    }
  }
}

// CHUNKED (a.restage = C >= 2, sequences whose K and V do not fit LDS even without Q — configs[4]'s encoder, S 1024 at head size 32):
// the output phase stages K and V in C chunks of SP / C keys over the Q tile and every wave carries the accumulators of its (at most
// two) owned query blocks across the chunks. A separate instantiation: the two accumulator sets are registers the other forms do not pay.
template <typename T, int DH, bool QKV = false, bool CHUNKED = false>
__global__ __launch_bounds__((FWD_MAX_WAVES<DH, QKV || CHUNKED> * 64)) __attribute__((amdgpu_waves_per_eu((DH == 16 ? MST_ATT16_WAVES_FWD : (FWD_MAX_WAVES<DH, QKV || CHUNKED> == 8 ? 2 : 4))))) void attn_fwd_res_kernel(AttnArgs a) {
  constexpr int KS = DH / 16, DB = (DH + 31) / 32, LD = LdsLd<DH>::V;
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
  const int64_t S = a.S;
  const int NB = (int)((S + 31) / 32), SP = NB * 32;
  const bool restage = !QKV && a.restage;  // two tiles: [Q, then K][V]; CHUNKED: one tile: [Q, then a chunk of K | the same chunk of V]
  const int CR = CHUNKED ? SP / a.restage : SP;  // keys staged at a time in the output phase
  T* sQ = reinterpret_cast<T*>(att_smem);
  T* sK = restage ? sQ : sQ + SP * LD;
  T* sV = sK + CR * LD;
  float* sSk = reinterpret_cast<float*>(CHUNKED ? sQ + SP * LD : sV + SP * LD);
  float* sCk = sSk + SP; float* sMadd = sCk + SP; float* sMax = sMadd + SP; float* sLogl = sMax + SP;
  float* sRed = sLogl + SP;  // [LONE_RED]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, NW = nthr >> 6;
  const int64_t bh = res_wg_bh(a.B, a.H), b = (int64_t)((uint32_t)bh / (uint32_t)a.H), hd = bh - b * a.H;  // (32-bit division: B * H <= 65535)
  const int64_t plane = a.B * a.H * S;
  const bool lone = lone_row_shape(S, DH) && a.q_limit >= S;  // the last row is handled apart (above)
  const int NBo = lone ? NB - 1 : NB;                            // owner blocks swept with MFMA tiles
  const T* base = reinterpret_cast<const T*>(a.qkv) + b * S * a.ld_qkv + hd * DH;
  ATT_STAMP(0);
  if constexpr (QKV) {
    static_assert(DH == 32, "the fused projection is built for head size 32");
    qkv_prologue<T>(a, sQ, sK, sV, b, hd, NB);
  } else {
    stage_all<T, DH>(sQ, base + a.q_off, a.ld_qkv, S, SP, tid, nthr);
    if (!restage) {
      stage_all<T, DH>(sK, base + a.k_off, a.ld_qkv, S, SP, tid, nthr);
      stage_all<T, DH>(sV, base + a.v_off, a.ld_qkv, S, SP, tid, nthr);
    }
  }
  __syncthreads();  // (a workgroup barrier waits for LDS traffic only: the projection's qkv stores drain under the statistics phase)
  ATT_STAMP(2);

  // ---- phase A: softmax statistics of every key row (the arithmetic of attn_fwd_stats_kernel)
  int padded = 0;
  for (int ob = wave; ob < NBo; ob += NW) {
    const int64_t k_lane = ob * 32 + (lane & 31);
    typename Act<T>::vec8 kf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) kf[s] = restage ? glb_row_frag<T>(base + a.k_off, a.ld_qkv, k_lane, S, s, lane) : lds_row_frag<T, DH>(sK, ob * 32, s, lane);
    const bool in = k_lane < S;
    const bool vk = in && a.keymask[b * S + k_lane];
    const float madd = vk ? 0.f : MASK_VALUE;
    float m = NEG_BIG, l = 0.f;
    stats_sweep<T, DH>(sQ, NB, 0, S, a.scale, madd, __any(!vk), kf, m, l, lane, lone);
    const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64);
    const float M = fmaxf(m, m2);
    const float logl = __logf(l * __expf(m - M) + l2 * __expf(m2 - M));
    if (lane < 32) {
      if (in) {
        a.lse[bh * S + k_lane] = M;
        a.lse[plane + bh * S + k_lane] = logl;
      }
      key_consts(in, vk, in ? M : 0.f, in ? logl : 0.f, a.scale, sSk[k_lane], sCk[k_lane]);
      sMadd[k_lane] = madd; sMax[k_lane] = in ? M : 0.f; sLogl[k_lane] = in ? logl : INFINITY;
    }
    padded |= (in && !vk);
  }
  if constexpr (DH <= 16) {
    if (lone) {  // statistics of the lone key: queries on the lanes, the reference's operation order (stats_tile_exact)
      const int64_t e = S - 1;
      const bool vk = a.keymask[b * S + e];
      const float madd = vk ? 0.f : MASK_VALUE;
      float ke[DH];
      if (restage) {
#pragma unroll
        for (int f = 0; f < DH; ++f) ke[f] = to_f32(base[a.k_off + e * a.ld_qkv + f]);
      } else {
        lds_row_load<T, DH>(sK + e * LD, ke);
      }
      float m = NEG_BIG, l = 0.f;
      for (int q = tid; q < S; q += nthr) {
        const float t = fmaf(lds_row_dot<T, DH>(sQ + q * LD, ke), a.scale, madd);
        const float m_new = fmaxf(m, t);
        l = l * __expf(m - m_new) + __expf(t - m_new);
        m = m_new;
      }
      const float mw = wave_max(m);
      const float lw = wave_sum(l * __expf(m - mw));  // (a lane without a query: l = 0)
      if (lane == 0) { sRed[wave] = mw; sRed[16 + wave] = lw; }
      __syncthreads();
      if (tid == 0) {
        float M = NEG_BIG, L = 0.f;
        for (int w = 0; w < NW; ++w) M = fmaxf(M, sRed[w]);
        for (int w = 0; w < NW; ++w) L += sRed[16 + w] * __expf(sRed[w] - M);
        const float logl = __logf(L);
        a.lse[bh * S + e] = M;
        a.lse[plane + bh * S + e] = logl;
        key_consts(true, vk, M, logl, a.scale, sSk[e], sCk[e]);
        sMadd[e] = madd; sMax[e] = M; sLogl[e] = logl;
        for (int k = (int)S; k < SP; ++k) { key_consts(false, false, 0.f, 0.f, a.scale, sSk[k], sCk[k]); sMadd[k] = MASK_VALUE; sMax[k] = 0.f; sLogl[k] = INFINITY; }
      }
      padded |= !vk;
    }
  }
  const bool exact = __syncthreads_or(padded);  // does this sequence hold a padded key?
  const uint64_t pad_tiles = exact ? padded_tile_mask(sSk, sCk, NB, lane) : 0ull;  // ... and which of its key tiles do
  ATT_STAMP(3);
  if constexpr (CHUNKED) {
    // ---- phase B in C chunks of CR keys; this wave's owned query blocks are ob0 = wave and ob1 = wave + NW (host: NB <= 2 NW, no lone row)
    constexpr int OBM = 2;
    typename Act<T>::vec8 qf[OBM][KS];
    f32x16 o[OBM][DB];
    bool act[OBM];
#pragma unroll
    for (int i = 0; i < OBM; ++i) {
      const int ob = wave + i * NW;
      act[i] = ob < NBo && (int64_t)ob * 32 < a.q_limit;
#pragma unroll
      for (int s = 0; s < KS; ++s) qf[i][s] = glb_row_frag<T>(base + a.q_off, a.ld_qkv, act[i] ? (int64_t)ob * 32 + (lane & 31) : S, S, s, lane);
#pragma unroll
      for (int d = 0; d < DB; ++d) o[i][d] = zero16<DH>();
    }
    const int tiles_c = CR / 32;
    for (int c = 0; c < a.restage; ++c) {
      if (c > 0) __syncthreads();  // every wave is done with the previous chunk (c = 0: with the staged Q, the barrier above)
      const int64_t k0 = (int64_t)c * CR;
      stage_pair<T, DH>(sK, base + a.k_off + k0 * a.ld_qkv, a.ld_qkv, S - k0, sV, base + a.v_off + k0 * a.ld_qkv, a.ld_qkv, S - k0, CR, tid, nthr);
      __syncthreads();
      const int nt = (NBo - c * tiles_c) < tiles_c ? (NBo - c * tiles_c) : tiles_c;  // (the last chunk may hold fewer tiles)
#pragma unroll
      for (int i = 0; i < OBM; ++i) {
        if (!act[i]) continue;
        if (exact) {
          // (two loops, not a branch per tile: with both forms in one loop body the register allocator spilled 44 registers)
          const uint64_t padc = (pad_tiles >> (c * tiles_c)) & tile_bits(nt);
          MST_FOR_TILES(kt, ~padc & tile_bits(nt)) fwd_out_tile<T, DH, false>(sK, sV, sSk + k0, sCk + k0, sMadd + k0, sMax + k0, sLogl + k0, kt, a.scale, qf[i], o[i], lane);
          MST_FOR_TILES(kt, padc) fwd_out_tile<T, DH, true>(sK, sV, sSk + k0, sCk + k0, sMadd + k0, sMax + k0, sLogl + k0, kt, a.scale, qf[i], o[i], lane);
        } else {
          for (int kt = 0; kt < nt; ++kt) fwd_out_tile<T, DH, false>(sK, sV, sSk + k0, sCk + k0, sMadd + k0, sMax + k0, sLogl + k0, kt, a.scale, qf[i], o[i], lane);
        }
      }
    }
    T* og = reinterpret_cast<T*>(a.out) + b * S * a.ld_out + hd * DH;
#pragma unroll
    for (int i = 0; i < OBM; ++i) {
      const int64_t q_lane = (int64_t)(wave + i * NW) * 32 + (lane & 31);
      owner_store<T, DH>((act[i] && q_lane < S && q_lane < a.q_limit) ? og + q_lane * a.ld_out : nullptr, o[i], lane);
    }
    return;
  }
  if (restage) {  // (every wave is done with the staged Q)
    stage_pair<T, DH>(sK, base + a.k_off, a.ld_qkv, S, sV, base + a.v_off, a.ld_qkv, S, SP, tid, nthr);
    __syncthreads();
  }

  // ---- phase B: O = P^T V for the owned queries (attn_fwd_out_kernel's tiles)
  for (int ob = wave; ob < NBo; ob += NW) {
    if ((int64_t)ob * 32 >= a.q_limit) break;
    typename Act<T>::vec8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
      qf[s] = restage ? glb_row_frag<T>(base + a.q_off, a.ld_qkv, (int64_t)ob * 32 + (lane & 31), S, s, lane) : lds_row_frag<T, DH>(sQ, ob * 32, s, lane);
    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = zero16<DH>();
    if (exact) {
      MST_FOR_TILES(kt, ~pad_tiles & tile_bits(NBo)) fwd_out_tile<T, DH, false>(sK, sV, sSk, sCk, sMadd, sMax, sLogl, kt, a.scale, qf, o, lane);
      MST_FOR_TILES(kt, pad_tiles & tile_bits(NBo)) fwd_out_tile<T, DH, true>(sK, sV, sSk, sCk, sMadd, sMax, sLogl, kt, a.scale, qf, o, lane);
    } else {
      for (int kt = 0; kt < NBo; ++kt) fwd_out_tile<T, DH, false>(sK, sV, sSk, sCk, sMadd, sMax, sLogl, kt, a.scale, qf, o, lane);
    }
    if constexpr (DH <= 16) {
      if (lone) {  // the ninth key tile holds the lone key only
        if (exact) fwd_out_tile<T, DH, true, true>(sK, sV, sSk, sCk, sMadd, sMax, sLogl, NB - 1, a.scale, qf, o, lane);
        else fwd_out_tile<T, DH, false, true>(sK, sV, sSk, sCk, sMadd, sMax, sLogl, NB - 1, a.scale, qf, o, lane);
      }
    }
    T* og = reinterpret_cast<T*>(a.out) + b * S * a.ld_out + hd * DH;
    const int64_t q_lane = ob * 32 + (lane & 31);
    owner_store<T, DH>((q_lane < S && q_lane < a.q_limit) ? og + q_lane * a.ld_out : nullptr, o, lane);
  }
  ATT_STAMP(4);
  if constexpr (DH <= 16) {
    if (lone) {  // O[e] = sum_k P[k,e] V[k]: keys on the lanes (P stays fp32 here; the MFMA form rounds it to the activation type)
      const int64_t e = S - 1;
      float qe[DH], acc[DH];
      if (restage) {
#pragma unroll
        for (int f = 0; f < DH; ++f) qe[f] = to_f32(base[a.q_off + e * a.ld_qkv + f]);
      } else {
        lds_row_load<T, DH>(sQ + e * LD, qe);
      }
#pragma unroll
      for (int f = 0; f < DH; ++f) acc[f] = 0.f;
      for (int k = tid; k < S; k += nthr) {
        const float x = lds_row_dot<T, DH>(sK + k * LD, qe);
        const float p = exact ? exact_prob(x, a.scale, sMadd[k], sMax[k], sLogl[k]) : fast_exp2(fmaf(x, sSk[k], sCk[k]));
        lds_row_axpy<T, DH>(p, sV + k * LD, acc);
      }
#pragma unroll
      for (int f = 0; f < DH; ++f) acc[f] = wave_sum(acc[f]);
      if (lane == 0) {
#pragma unroll
        for (int f = 0; f < DH; ++f) sRed[wave * DH + f] = acc[f];
      }
      __syncthreads();
      if (tid < DH) {
        float o_e = 0.f;
        for (int w = 0; w < NW; ++w) o_e += sRed[w * DH + tid];
        T* og = reinterpret_cast<T*>(a.out) + b * S * a.ld_out + hd * DH;
        og[e * a.ld_out + tid] = (T)o_e;
      }
    }
  }
}

template <typename T, int DH, bool SPARSE>
__global__ __launch_bounds__(RES_MAX_WAVES<DH> * 64) __attribute__((amdgpu_waves_per_eu(DH == 16 ? MST_ATT16_WAVES_BWD : (DH == 64 ? 2 : 4)))) void attn_bwd_res_kernel(AttnArgs a) {
  constexpr int KS = DH / 16, DB = (DH + 31) / 32, LD = LdsLd<DH>::V;
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
  const int64_t S = a.S;
  const int NB = (int)((S + 31) / 32), SP = NB * 32;
  T* bufA = reinterpret_cast<T*>(att_smem);  // phase A: Q   phase B: K
  T* bufB = bufA + SP * LD;                  // phase A: dO  phase B: V
  float* sSk = reinterpret_cast<float*>(bufB + SP * LD);
  float* sCk = sSk + SP; float* sMadd = sCk + SP; float* sMax = sMadd + SP; float* sLogl = sMax + SP; float* sNd = sLogl + SP;
  float* sRed = sNd + SP;  // [LONE_RED]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, NW = nthr >> 6;
  const int64_t bh = res_wg_bh(a.B, a.H), b = (int64_t)((uint32_t)bh / (uint32_t)a.H), hd = bh - b * a.H;  // (32-bit division: B * H <= 65535)
  const int64_t plane = a.B * a.H * S;
  const bool lone = !SPARSE && lone_row_shape(S, DH);  // the last row is handled apart (see lone_row_shape)
  const int NBo = lone ? NB - 1 : NB;                    // owner blocks swept with MFMA tiles
  const float log2_scale = __log2f(a.scale);
  const T* base = reinterpret_cast<const T*>(a.qkv) + b * S * a.ld_qkv + hd * DH;
  const T* Kg = base + a.k_off;
  const T* Qg = base + a.q_off;
  const T* Vg = base + a.v_off;
  const T* dOg = reinterpret_cast<const T*>(a.dout) + b * S * a.ld_dout + hd * DH;
  T* dbase = reinterpret_cast<T*>(a.dqkv) + b * S * a.ld_dqkv + hd * DH;
  // Sparse mode (0 < q_limit <= 32: dO is zero from row q_limit on — the top encoder layer, whose output is read at
  // position 0 only): dP vanishes outside query block 0, so dV and delta need that block alone and every other
  // tile of dK / dQ is the LIGHT form. Same arithmetic as the dense path on the zero rows, about 45 % of its work.
  constexpr bool sparse = SPARSE;  // host: 0 < q_limit <= 32 (a separate instantiation keeps the dense kernel's registers)
  const int64_t do_rows = sparse ? a.q_limit : S;  // rows of dO that are read; the staged tile is zero beyond them
  // the first owned key block's K / V fragments are requested before the staging loads: one exposed round trip, not two
  typename Act<T>::vec8 kf[KS], vf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    kf[s] = glb_row_frag<T>(Kg, a.ld_qkv, wave * 32 + (lane & 31), S, s, lane);
    vf[s] = glb_row_frag<T>(Vg, a.ld_qkv, wave * 32 + (lane & 31), S, s, lane);
  }
  ATT_STAMP(0);
  stage_pair<T, DH>(bufA, Qg, a.ld_qkv, S, bufB, dOg, a.ld_dout, do_rows, SP, tid, nthr);
  __syncthreads();
  ATT_STAMP(1);
  const int nq0 = sparse ? 1 : NBo;  // (whole) query tiles that contribute to pass 0

  // ---- phase A: dV, delta, dK for the owned keys (attn_bwd_kv_kernel's two passes over the query tiles)
  int padded = 0;
  for (int ob = wave; ob < NBo; ob += NW) {
    const int64_t k_lane = ob * 32 + (lane & 31);
    if (ob != wave) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        kf[s] = glb_row_frag<T>(Kg, a.ld_qkv, k_lane, S, s, lane);
        vf[s] = glb_row_frag<T>(Vg, a.ld_qkv, k_lane, S, s, lane);
      }
    }
    const bool in = k_lane < S;
    const bool vk = in && a.keymask[b * S + k_lane];
    const float rmax = in ? a.lse[bh * S + k_lane] : 0.f;
    const float logl = in ? a.lse[plane + bh * S + k_lane] : INFINITY;
    const float madd = vk ? 0.f : MASK_VALUE;
    float sk2, ck2;
    key_consts(in, vk, rmax, in ? logl : 0.f, a.scale, sk2, ck2);
    const float ck2s = ck2 + log2_scale;  // pass 1 and phase B: the exponential is P * scale
    const bool exact_w = __any(in && !vk);  // wave-uniform: one of this block's 32 keys is padded
    padded |= (in && !vk);
    f32x16 acc[DB];
    float neg_delta = 0.f;
    T* const drow = in ? dbase + k_lane * a.ld_dqkv : nullptr;
    // four straight-line tile loops (pass x exact) instead of branches inside one: the merged form shuffled the
    // probability tile through 15 v_mov per tile to reconcile the two passes' register assignments
    // ---- pass 0: dV, then delta = V . dV
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[d] = zero16<DH>();
    if (exact_w) for (int qt = 0; qt < nq0; ++qt) bwd_kv_tile<T, DH, 0, true>(bufA, bufB, qt, a.scale, kf, vf, sk2, ck2, madd, rmax, logl, neg_delta, acc, lane);
    else for (int qt = 0; qt < nq0; ++qt) bwd_kv_tile<T, DH, 0, false>(bufA, bufB, qt, a.scale, kf, vf, sk2, ck2, madd, rmax, logl, neg_delta, acc, lane);
    if constexpr (DH <= 16 && !SPARSE) {
      if (lone) {  // the ninth query tile holds the lone query only
        if (exact_w) bwd_kv_tile<T, DH, 0, true, false, true>(bufA, bufB, NB - 1, a.scale, kf, vf, sk2, ck2, madd, rmax, logl, neg_delta, acc, lane);
        else bwd_kv_tile<T, DH, 0, false, false, true>(bufA, bufB, NB - 1, a.scale, kf, vf, sk2, ck2, madd, rmax, logl, neg_delta, acc, lane);
      }
    }
    {
      const float delta = delta_from_dv<T, DH>(acc, vf, lane);
      if (lane < 32 && in) a.delta[bh * S + k_lane] = delta;
      neg_delta = -delta;
    }
    owner_store<T, DH>(drow ? drow + a.v_off : nullptr, acc, lane);
    ATT_STAMP(2);
    // ---- pass 1: dK
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[d] = zero16<DH>();
    if (exact_w) for (int qt = 0; qt < nq0; ++qt) bwd_kv_tile<T, DH, 1, true>(bufA, bufB, qt, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
    else for (int qt = 0; qt < nq0; ++qt) bwd_kv_tile<T, DH, 1, false>(bufA, bufB, qt, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
    if (sparse) {
      if (exact_w) for (int qt = nq0; qt < NB; ++qt) bwd_kv_tile<T, DH, 1, true, true>(bufA, bufB, qt, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
      else for (int qt = nq0; qt < NB; ++qt) bwd_kv_tile<T, DH, 1, false, true>(bufA, bufB, qt, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
    }
    if constexpr (DH <= 16 && !SPARSE) {
      if (lone) {
        if (exact_w) bwd_kv_tile<T, DH, 1, true, false, true>(bufA, bufB, NB - 1, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
        else bwd_kv_tile<T, DH, 1, false, false, true>(bufA, bufB, NB - 1, a.scale, kf, vf, sk2, ck2s, madd, rmax, logl, neg_delta, acc, lane);
      }
    }
    owner_store<T, DH>(drow ? drow + a.k_off : nullptr, acc, lane);
    if (lane < 32) {
      sSk[k_lane] = sk2; sCk[k_lane] = ck2s; sMadd[k_lane] = madd; sMax[k_lane] = rmax; sLogl[k_lane] = logl;
      sNd[k_lane] = in ? neg_delta : 0.f;
    }
  }
  if constexpr (DH <= 16 && !SPARSE) {
    if (lone) {  // the lone key: queries on the lanes; dV, delta = V . dV, then dK (two sweeps, as the MFMA form)
      const int64_t e = S - 1;
      float ke[DH], ve[DH], acc[DH];
#pragma unroll
      for (int f = 0; f < DH; ++f) { ke[f] = to_f32(Kg[e * a.ld_qkv + f]); ve[f] = to_f32(Vg[e * a.ld_qkv + f]); }
      const bool vk = a.keymask[b * S + e];
      const float rmax = a.lse[bh * S + e], logl = a.lse[plane + bh * S + e], madd = vk ? 0.f : MASK_VALUE;
      float sk2, ck2;
      key_consts(true, vk, rmax, logl, a.scale, sk2, ck2);
      const float ck2s = ck2 + log2_scale;
      padded |= !vk;
#pragma unroll
      for (int f = 0; f < DH; ++f) acc[f] = 0.f;
      for (int q = tid; q < S; q += nthr) {
        const float x = lds_row_dot<T, DH>(bufA + q * LD, ke);
        const float pr = vk ? fast_exp2(fmaf(x, sk2, ck2)) : exact_prob(x, a.scale, madd, rmax, logl);
        lds_row_axpy<T, DH>(pr, bufB + q * LD, acc);
      }
#pragma unroll
      for (int f = 0; f < DH; ++f) acc[f] = wave_sum(acc[f]);
      if (lane == 0) {
#pragma unroll
        for (int f = 0; f < DH; ++f) sRed[wave * DH + f] = acc[f];
      }
      __syncthreads();
      if (tid < DH) {
        float dv = 0.f;
        for (int w = 0; w < NW; ++w) dv += sRed[w * DH + tid];
        sRed[LONE_RED / 2 + tid] = dv;
        dbase[e * a.ld_dqkv + a.v_off + tid] = (T)dv;
      }
      __syncthreads();
      float delta = 0.f;
#pragma unroll
      for (int f = 0; f < DH; ++f) delta = fmaf(ve[f], sRed[LONE_RED / 2 + f], delta);
#pragma unroll
      for (int f = 0; f < DH; ++f) acc[f] = 0.f;
      for (int q = tid; q < S; q += nthr) {
        const float x = lds_row_dot<T, DH>(bufA + q * LD, ke);
        const float ps = vk ? fast_exp2(fmaf(x, sk2, ck2s)) : exact_prob(x, a.scale, madd, rmax, logl) * a.scale;
        const float dl = ps * (lds_row_dot<T, DH>(bufB + q * LD, ve) - delta);
        lds_row_axpy<T, DH>(dl, bufA + q * LD, acc);
      }
#pragma unroll
      for (int f = 0; f < DH; ++f) acc[f] = wave_sum(acc[f]);
      __syncthreads();  // (the dV partials have been read)
      if (lane == 0) {
#pragma unroll
        for (int f = 0; f < DH; ++f) sRed[wave * DH + f] = acc[f];
      }
      __syncthreads();
      if (tid < DH) {
        float dk = 0.f;
        for (int w = 0; w < NW; ++w) dk += sRed[w * DH + tid];
        dbase[e * a.ld_dqkv + a.k_off + tid] = (T)dk;
      }
      if (tid == 0) {
        a.delta[bh * S + e] = delta;
        sSk[e] = sk2; sCk[e] = ck2s; sMadd[e] = madd; sMax[e] = rmax; sLogl[e] = logl; sNd[e] = -delta;
        for (int k = (int)S; k < SP; ++k) {
          float s0, c0;
          key_consts(false, false, 0.f, 0.f, a.scale, s0, c0);
          sSk[k] = s0; sCk[k] = c0 + log2_scale; sMadd[k] = MASK_VALUE; sMax[k] = 0.f; sLogl[k] = INFINITY; sNd[k] = 0.f;
        }
      }
    }
  }
  ATT_STAMP(3);
  // phase B's first Q / dO fragments: requested before the barrier and the K / V staging
  typename Act<T>::vec8 qf[KS], dof[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    qf[s] = glb_row_frag<T>(Qg, a.ld_qkv, wave * 32 + (lane & 31), S, s, lane);
    dof[s] = glb_row_frag<T>(dOg, a.ld_dout, wave * 32 + (lane & 31), do_rows, s, lane);
  }
  __syncthreads();  // every wave is done with the staged Q and dO
  stage_pair<T, DH>(bufA, Kg, a.ld_qkv, S, bufB, Vg, a.ld_qkv, S, SP, tid, nthr);
  const bool exact = __syncthreads_or(padded);
  const uint64_t pad_tiles = exact ? padded_tile_mask(sSk, sCk, NB, lane) : 0ull;  // (the barrier above: every key's constants are in LDS)
  ATT_STAMP(4);

  // ---- phase B: dQ for the owned queries (attn_bwd_q_kernel's tiles)
  for (int ob = wave; ob < NBo; ob += NW) {
    const int64_t q_lane = ob * 32 + (lane & 31);
    if (ob != wave) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        qf[s] = glb_row_frag<T>(Qg, a.ld_qkv, q_lane, S, s, lane);
        dof[s] = glb_row_frag<T>(dOg, a.ld_dout, q_lane, do_rows, s, lane);
      }
    }
    f32x16 acc[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[d] = zero16<DH>();
    if (sparse && ob > 0) {  // this block's dO rows are zero
      if (exact) {
        MST_FOR_TILES(kt, ~pad_tiles & tile_bits(NB)) bwd_q_tile<T, DH, false, true>(bufA, bufB, sSk, sCk, sMadd, sMax, sLogl, sNd, kt, a.scale, qf, dof, acc, lane);
        MST_FOR_TILES(kt, pad_tiles & tile_bits(NB)) bwd_q_tile<T, DH, true, true>(bufA, bufB, sSk, sCk, sMadd, sMax, sLogl, sNd, kt, a.scale, qf, dof, acc, lane);
      } else for (int kt = 0; kt < NB; ++kt) bwd_q_tile<T, DH, false, true>(bufA, bufB, sSk, sCk, sMadd, sMax, sLogl, sNd, kt, a.scale, qf, dof, acc, lane);
    } else if (exact) {
      MST_FOR_TILES(kt, ~pad_tiles & tile_bits(NBo)) bwd_q_tile<T, DH, false>(bufA, bufB, sSk, sCk, sMadd, sMax, sLogl, sNd, kt, a.scale, qf, dof, acc, lane);
      MST_FOR_TILES(kt, pad_tiles & tile_bits(NBo)) bwd_q_tile<T, DH, true>(bufA, bufB, sSk, sCk, sMadd, sMax, sLogl, sNd, kt, a.scale, qf, dof, acc, lane);
    } else {
      for (int kt = 0; kt < NBo; ++kt) bwd_q_tile<T, DH, false>(bufA, bufB, sSk, sCk, sMadd, sMax, sLogl, sNd, kt, a.scale, qf, dof, acc, lane);
    }
    if constexpr (DH <= 16 && !SPARSE) {
      if (lone) {  // the ninth key tile holds the lone key only
        if (exact) bwd_q_tile<T, DH, true, false, true>(bufA, bufB, sSk, sCk, sMadd, sMax, sLogl, sNd, NB - 1, a.scale, qf, dof, acc, lane);
        else bwd_q_tile<T, DH, false, false, true>(bufA, bufB, sSk, sCk, sMadd, sMax, sLogl, sNd, NB - 1, a.scale, qf, dof, acc, lane);
      }
    }
    owner_store<T, DH>(q_lane < S ? dbase + a.q_off + q_lane * a.ld_dqkv : nullptr, acc, lane);
  }
  ATT_STAMP(5);
  if constexpr (DH <= 16 && !SPARSE) {
    if (lone) {  // dQ of the lone query: keys on the lanes
      const int64_t e = S - 1;
      float qe[DH], doe[DH], acc[DH];
#pragma unroll
      for (int f = 0; f < DH; ++f) { qe[f] = to_f32(Qg[e * a.ld_qkv + f]); doe[f] = to_f32(dOg[e * a.ld_dout + f]); acc[f] = 0.f; }
      for (int k = tid; k < S; k += nthr) {
        const float x = lds_row_dot<T, DH>(bufA + k * LD, qe);
        const float ps = exact ? exact_prob(x, a.scale, sMadd[k], sMax[k], sLogl[k]) * a.scale : fast_exp2(fmaf(x, sSk[k], sCk[k]));
        const float dl = ps * (lds_row_dot<T, DH>(bufB + k * LD, doe) + sNd[k]);
        lds_row_axpy<T, DH>(dl, bufA + k * LD, acc);
      }
#pragma unroll
      for (int f = 0; f < DH; ++f) acc[f] = wave_sum(acc[f]);
      if (lane == 0) {
#pragma unroll
        for (int f = 0; f < DH; ++f) sRed[wave * DH + f] = acc[f];
      }
      __syncthreads();
      if (tid < DH) {
        float dq = 0.f;
        for (int w = 0; w < NW; ++w) dq += sRed[w * DH + tid];
        dbase[e * a.ld_dqkv + a.q_off + tid] = (T)dq;
      }
    }
  }
}

// ---- dQ with the keys staged in CHUNKS (sequences too long for the resident backward kernel: configs[4]'s encoder, S 1024).
// The streaming attn_bwd_q_kernel re-stages 64 keys at a time for four waves and spends 860 cycles per tile visit — twice the
// resident kernels — on per-stage bookkeeping; here one workgroup owns a (batch, head), stages K | V in chunks of CR keys for all
// of its waves at once, fills the per-key constants once, and every wave carries the dQ accumulators of its (up to OBM) query
// blocks across the chunks. Same tile arithmetic (bwd_q_tile). 8 waves: 256 registers each hold four blocks' accumulators and fragments.
template <typename T, int DH, int OBM>
__global__ __launch_bounds__(512) void attn_bwd_q_chunk_kernel(AttnArgs a, int n_chunks) {
  constexpr int KS = DH / 16, DB = (DH + 31) / 32, LD = LdsLd<DH>::V;
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
  const int64_t S = a.S;
  const int NB = (int)((S + 31) / 32), SP = NB * 32, CR = SP / n_chunks;
  T* sK = reinterpret_cast<T*>(att_smem);
  T* sV = sK + CR * LD;
  float* sSk = reinterpret_cast<float*>(sV + CR * LD);
  float* sCk = sSk + SP; float* sMadd = sCk + SP; float* sMax = sMadd + SP; float* sLogl = sMax + SP; float* sNd = sLogl + SP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, NW = nthr >> 6;
  const int64_t bh = res_wg_bh(a.B, a.H), b = (int64_t)((uint32_t)bh / (uint32_t)a.H), hd = bh - b * a.H;
  const int64_t plane = a.B * a.H * S;
  const float log2_scale = __log2f(a.scale);
  const T* base = reinterpret_cast<const T*>(a.qkv) + b * S * a.ld_qkv + hd * DH;
  const T* dOg = reinterpret_cast<const T*>(a.dout) + b * S * a.ld_dout + hd * DH;
  const bool sparse = a.q_limit > 0 && a.q_limit <= 32;    // dO is zero from row q_limit on (the top encoder layer)
  const int64_t do_rows = sparse ? a.q_limit : S;
  // the owned blocks' Q / dO fragments: requested first, they are not needed before the first chunk is staged
  typename Act<T>::vec8 qf[OBM][KS], dof[OBM][KS];
  f32x16 acc[OBM][DB];
  bool act[OBM];
#pragma unroll
  for (int i = 0; i < OBM; ++i) {
    const int ob = wave + i * NW;
    act[i] = ob < NB;
    const int64_t q_lane = act[i] ? (int64_t)ob * 32 + (lane & 31) : S;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      qf[i][s] = glb_row_frag<T>(base + a.q_off, a.ld_qkv, q_lane, S, s, lane);
      dof[i][s] = glb_row_frag<T>(dOg, a.ld_dout, q_lane, do_rows, s, lane);
    }
#pragma unroll
    for (int d = 0; d < DB; ++d) acc[i][d] = zero16<DH>();
  }
  // per-key constants of the whole sequence (attn_bwd_q_kernel computes them per stage)
  int padded = 0;
  for (int k = tid; k < SP; k += nthr) {
    const bool in = k < S;
    const bool vk = in && a.keymask[b * S + k];
    const float rmax = in ? a.lse[bh * S + k] : 0.f, logl = in ? a.lse[plane + bh * S + k] : 0.f;
    float sk2, ck2;
    key_consts(in, vk, rmax, logl, a.scale, sk2, ck2);
    sSk[k] = sk2; sCk[k] = ck2 + log2_scale;  // the exponential is P * scale (bwd_q_tile)
    sMadd[k] = vk ? 0.f : MASK_VALUE; sMax[k] = rmax; sLogl[k] = in ? logl : INFINITY;
    sNd[k] = in ? -a.delta[bh * S + k] : 0.f;
    padded |= (in && !vk);
  }
  const bool exact = __syncthreads_or(padded);
  const uint64_t pad_tiles = exact ? padded_tile_mask(sSk, sCk, NB, lane) : 0ull;
  const int tiles_c = CR / 32;
  for (int c = 0; c < n_chunks; ++c) {
    if (c > 0) __syncthreads();  // every wave is done with the previous chunk
    const int64_t k0 = (int64_t)c * CR;
    if (k0 >= S) break;  // (uniform)
    stage_pair<T, DH>(sK, base + a.k_off + k0 * a.ld_qkv, a.ld_qkv, S - k0, sV, base + a.v_off + k0 * a.ld_qkv, a.ld_qkv, S - k0, CR, tid, nthr);
    __syncthreads();
    const int nt = (NB - c * tiles_c) < tiles_c ? (NB - c * tiles_c) : tiles_c;
#pragma unroll
    for (int i = 0; i < OBM; ++i) {
      if (!act[i]) continue;
      const bool light = sparse && (wave + i * NW) > 0;  // this block's dO rows are zero
      if (light) {
        if (exact) {
          const uint64_t padc = (pad_tiles >> (c * tiles_c)) & tile_bits(nt);
          MST_FOR_TILES(kt, ~padc & tile_bits(nt)) bwd_q_tile<T, DH, false, true>(sK, sV, sSk + k0, sCk + k0, sMadd + k0, sMax + k0, sLogl + k0, sNd + k0, kt, a.scale, qf[i], dof[i], acc[i], lane);
          MST_FOR_TILES(kt, padc) bwd_q_tile<T, DH, true, true>(sK, sV, sSk + k0, sCk + k0, sMadd + k0, sMax + k0, sLogl + k0, sNd + k0, kt, a.scale, qf[i], dof[i], acc[i], lane);
        } else for (int kt = 0; kt < nt; ++kt) bwd_q_tile<T, DH, false, true>(sK, sV, sSk + k0, sCk + k0, sMadd + k0, sMax + k0, sLogl + k0, sNd + k0, kt, a.scale, qf[i], dof[i], acc[i], lane);
      } else if (exact) {
        const uint64_t padc = (pad_tiles >> (c * tiles_c)) & tile_bits(nt);
        MST_FOR_TILES(kt, ~padc & tile_bits(nt)) bwd_q_tile<T, DH, false>(sK, sV, sSk + k0, sCk + k0, sMadd + k0, sMax + k0, sLogl + k0, sNd + k0, kt, a.scale, qf[i], dof[i], acc[i], lane);
        MST_FOR_TILES(kt, padc) bwd_q_tile<T, DH, true>(sK, sV, sSk + k0, sCk + k0, sMadd + k0, sMax + k0, sLogl + k0, sNd + k0, kt, a.scale, qf[i], dof[i], acc[i], lane);
      } else {
        for (int kt = 0; kt < nt; ++kt) bwd_q_tile<T, DH, false>(sK, sV, sSk + k0, sCk + k0, sMadd + k0, sMax + k0, sLogl + k0, sNd + k0, kt, a.scale, qf[i], dof[i], acc[i], lane);
      }
    }
  }
  T* dst = reinterpret_cast<T*>(a.dqkv) + b * S * a.ld_dqkv + hd * DH + a.q_off;
#pragma unroll
  for (int i = 0; i < OBM; ++i) {
    const int64_t q_lane = (int64_t)(wave + i * NW) * 32 + (lane & 31);
    owner_store<T, DH>((act[i] && q_lane < S) ? dst + q_lane * a.ld_dqkv : nullptr, acc[i], lane);
  }
}

// Waves per workgroup for the resident kernels, or 0 when the sequence does not fit: maximise (resident waves per CU)
// x (balance of the 32-row owner blocks over the waves); 128 VGPRs per lane (launch bounds 1024) allow 16 waves per CU,
// the 96 of the head-size-16 instantiations 20 (two 9-wave workgroups of a 257-row sequence: the decoder of configs[1]).
static int choose_resident(int64_t S, int64_t n_wg, size_t lds_bytes, int waves_cu = 16, bool lone = false, int max_nw = 16) {
  const char* force = getenv("MST_ATTN_PATH");  // "stream" / "resident": pin the path (tests cover both)
  if (force && force[0] == 's') return 0;
  const size_t LDS_CU = 160 * 1024;
  if (lds_bytes > LDS_CU - 1024) return 0;
  const int NB = (int)cdiv(S, 32) - (lone ? 1 : 0);  // owner blocks
  const int by_grid = (int)cdiv(n_wg, 256);
  int best = 0;
  double best_score = 0.0;
  for (int nw = 1; nw <= max_nw && nw <= NB; ++nw) {
    int wgs = waves_cu / nw;
    if ((size_t)wgs * lds_bytes > LDS_CU) wgs = (int)(LDS_CU / lds_bytes);
    if (wgs > by_grid) wgs = by_grid;
    if (wgs < 1) continue;
    const double eff = (double)NB / (double)(nw * cdiv(NB, nw));
    const double score = (double)(wgs * nw) * eff;
    if (score >= best_score) { best_score = score; best = nw; }
  }
  return best;
}
template <int DH> static size_t res_lds_fwd(int64_t S, int tiles = 3) { const size_t SP = (size_t)cdiv(S, 32) * 32; return (size_t)tiles * SP * LdsLd<DH>::V * 2 + 5 * SP * 4 + LONE_RED * 4; }
template <int DH> static size_t res_lds_bwd(int64_t S) { const size_t SP = (size_t)cdiv(S, 32) * 32; return 2 * SP * LdsLd<DH>::V * 2 + 6 * SP * 4 + LONE_RED * 4; }

static int attn_check(int64_t B, int64_t S, int64_t H, int64_t dh, int64_t ld, int64_t k_off, int64_t q_off, int64_t v_off) {
  MST_CHECK_ARG(B > 0 && S > 0 && H > 0, "attention: B,S,H must be positive");
  MST_CHECK_ARG(dh == 16 || dh == 32 || dh == 64, "attention: head size must be 16, 32 or 64 (got %lld)", (long long)dh);
  MST_CHECK_ARG(ld % 8 == 0 && k_off % 8 == 0 && q_off % 8 == 0 && v_off % 8 == 0, "attention: ld and offsets must be multiples of 8");
  MST_CHECK_ARG(B * H <= 65535, "attention: B*H too large for grid.y");
  return MST_OK;
}

// the fused-projection launch: 1 when it ran, 0 when this shape does not take it (the caller then runs the projection GEMM and the
// plain launch), negative on error
template <typename T>
static int launch_fwd_qkv(const AttnArgs& a, hipStream_t s) {
  constexpr int DH = 32;
  const char* off = getenv("MST_ATTN_QKV");
  if (off && off[0] == '0') return 0;
  const size_t lds = res_lds_fwd<DH>(a.S);  // (the projection's slices borrow the Q / K / V tiles)
  const int NB = (int)cdiv(a.S, 32);
  const int nw = choose_resident(a.S, a.B * a.H, lds, 16, false);
  // every owner block needs its own wave (one pass over the weight slices); the staging pattern is laid out for 512 threads
  // (four x pieces and two weight pieces per thread; the weight slice must fit the V tile: NB >= 6)
  if (nw < NB || nw < 6 || a.Dm % QKV_KC != 0 || a.Dm != a.H * DH) return 0;
  static size_t attr_lds = 64 * 1024;
  if (lds > attr_lds) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_res_kernel<T, DH, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("attn_fwd_res_kernel (fused projection): LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
    attr_lds = lds;
  }
  hipLaunchKernelGGL((attn_fwd_res_kernel<T, DH, true>), dim3((unsigned)(a.B * a.H)), dim3(nw * 64), lds, s, a);
  MST_CHECK_LAUNCH("attn_fwd_res_kernel (fused projection)");
  return 1;
}

template <typename T, int DH>
static int launch_fwd(const AttnArgs& a_in, hipStream_t s) {
  AttnArgs a = a_in;
  size_t lds = res_lds_fwd<DH>(a.S);
  const bool lone_f = lone_row_shape(a.S, DH) && a.q_limit >= a.S;
  const int waves_cu = DH == 16 ? 4 * MST_ATT16_WAVES_FWD : 16;
  int nw = choose_resident(a.S, a.B * a.H, lds, waves_cu, lone_f);
  if (!nw) {  // Q | K | V do not fit together: two tiles, K and V staged over Q between the phases (configs[4]'s decoder: S 1025, dh 16)
    static const bool off = getenv("MST_ATTN_RESTAGE") && getenv("MST_ATTN_RESTAGE")[0] == '0';
    const size_t lds2 = res_lds_fwd<DH>(a.S, 2);
    const int nw2 = off ? 0 : choose_resident(a.S, a.B * a.H, lds2, waves_cu, lone_f);
    if (nw2) { nw = nw2; lds = lds2; a.restage = 1; }
    else if (!off && !lone_f && DH >= 32) {
      // ... nor K | V alone: one tile, the output phase in two chunks of keys (configs[4]'s encoder: S 1024, head size 32)
      const int NB = (int)cdiv(a.S, 32);
      const size_t lds1 = res_lds_fwd<DH>(a.S, 1);
      constexpr int MAXW = FWD_MAX_WAVES<DH, true>;  // (the chunked instantiation's launch bounds)
      const int nw1 = NB % 2 == 0 ? choose_resident(a.S, a.B * a.H, lds1, DH == 16 ? waves_cu : MAXW, false, MAXW) : 0;
      if (nw1 && NB <= 2 * nw1) {
        a.restage = 2;
        static size_t attr_lds1 = 64 * 1024;
        if (lds1 > attr_lds1) {
          const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_res_kernel<T, DH, false, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
          if (e != hipSuccess) { set_error("attn_fwd_res_kernel (chunked): LDS opt-in of %zu bytes: %s", lds1, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
          attr_lds1 = lds1;
        }
        hipLaunchKernelGGL((attn_fwd_res_kernel<T, DH, false, true>), dim3((unsigned)(a.B * a.H)), dim3(nw1 * 64), lds1, s, a);
        MST_CHECK_LAUNCH("attn_fwd_res_kernel (chunked)");
        return MST_OK;
      }
    }
  }
  if (nw) {
    static size_t attr_lds = 64 * 1024;  // dynamic LDS above 64 KB has to be opted into
    if (lds > attr_lds) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_res_kernel<T, DH>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("attn_fwd_res_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
      attr_lds = lds;
    }
    hipLaunchKernelGGL((attn_fwd_res_kernel<T, DH>), dim3((unsigned)(a.B * a.H)), dim3(nw * 64), lds, s, a);
    MST_CHECK_LAUNCH("attn_fwd_res_kernel");
    return MST_OK;
  }
  dim3 grid((unsigned)cdiv(a.S, ATT_WG_ROWS), (unsigned)(a.B * a.H));
  hipLaunchKernelGGL((attn_fwd_stats_kernel<T, DH>), grid, dim3(256), 0, s, a);
  MST_CHECK_LAUNCH("attn_fwd_stats_kernel");
  dim3 grid_o((unsigned)cdiv(a.q_limit, ATT_WG_ROWS), (unsigned)(a.B * a.H));
  hipLaunchKernelGGL((attn_fwd_out_kernel<T, DH>), grid_o, dim3(256), 0, s, a);
  MST_CHECK_LAUNCH("attn_fwd_out_kernel");
  return MST_OK;
}
template <typename T, int DH>
static int launch_bwd(const AttnArgs& a, hipStream_t s) {
  const size_t lds = res_lds_bwd<DH>(a.S);
  const bool sparse_shape = a.q_limit > 0 && a.q_limit <= 32;
  if (const int nw = choose_resident(a.S, a.B * a.H, lds, DH == 16 ? 4 * MST_ATT16_WAVES_BWD : RES_MAX_WAVES<DH>, lone_row_shape(a.S, DH) && !sparse_shape, RES_MAX_WAVES<DH>)) {
    const bool sparse = sparse_shape;
    static size_t attr_lds[2] = {64 * 1024, 64 * 1024};  // dynamic LDS above 64 KB has to be opted into, per kernel
    if (lds > attr_lds[sparse]) {
      const void* fn = sparse ? reinterpret_cast<const void*>(&attn_bwd_res_kernel<T, DH, true>)
                              : reinterpret_cast<const void*>(&attn_bwd_res_kernel<T, DH, false>);
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("attn_bwd_res_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
      attr_lds[sparse] = lds;
    }
    if (sparse) hipLaunchKernelGGL((attn_bwd_res_kernel<T, DH, true>), dim3((unsigned)(a.B * a.H)), dim3(nw * 64), lds, s, a);
    else hipLaunchKernelGGL((attn_bwd_res_kernel<T, DH, false>), dim3((unsigned)(a.B * a.H)), dim3(nw * 64), lds, s, a);
    MST_CHECK_LAUNCH("attn_bwd_res_kernel");
    return MST_OK;
  }
  dim3 grid((unsigned)cdiv(a.S, ATT_WG_ROWS), (unsigned)(a.B * a.H));
  if (sparse_shape) hipLaunchKernelGGL((attn_bwd_kv_kernel<T, DH, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_bwd_kv_kernel<T, DH, false>), grid, dim3(256), 0, s, a);
  MST_CHECK_LAUNCH("attn_bwd_kv_kernel");
  {  // dQ: one workgroup per (batch, head) with the keys staged in chunks where the constants and two chunk tiles fit LDS
    static const bool off = getenv("MST_ATTN_QCHUNK") && getenv("MST_ATTN_QCHUNK")[0] == '0';
    constexpr int OBM = 4, NWQ = 8;
    const int NB = (int)cdiv(a.S, 32);
    if (!off && DH == 32 && NB <= OBM * NWQ) {  // (head size 32: at 64 four blocks' accumulators and fragments do not fit 256 registers)
      for (int nc = 1; nc <= 4; nc *= 2) {
        if (NB % nc != 0) break;
        const size_t lds = (size_t)2 * (NB * 32 / nc) * LdsLd<DH>::V * 2 + (size_t)6 * NB * 32 * 4;
        if (lds > 150 * 1024) continue;
        static size_t attr_lds = 64 * 1024;
        if (lds > attr_lds) {
          const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_chunk_kernel<T, DH, OBM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          if (e != hipSuccess) { set_error("attn_bwd_q_chunk_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
          attr_lds = lds;
        }
        hipLaunchKernelGGL((attn_bwd_q_chunk_kernel<T, DH, OBM>), dim3((unsigned)(a.B * a.H)), dim3(NWQ * 64), lds, s, a, nc);
        MST_CHECK_LAUNCH("attn_bwd_q_chunk_kernel");
        return MST_OK;
      }
    }
  }
  hipLaunchKernelGGL((attn_bwd_q_kernel<T, DH>), grid, dim3(256), 0, s, a);
  MST_CHECK_LAUNCH("attn_bwd_q_kernel");
  return MST_OK;
}


// ------------------------------------------------------------------------------------ incremental decode
// One new query row per sample against the rows cached so far (inference: model.py:259-272, transformer.py:70-77,242-249).
// The cache holds the layer's K | Q | V projections of every position fed so far ([B, t_max, ld] with the training layout
// k_off / q_off / v_off); the new row (position n_keys - 1) is already in it. Per (sample, head):
//   mode 0, the reference's arithmetic: logits[k, q] = K[k].Q[q] / sqrt(dh) + mask[k], softmax over the QUERY axis — which
//           holds the single new query, so every P[k, 0] is exp(0) / 1 = 1 and the output is the plain sum of the cached
//           value rows (computed as such: the logit cancels exactly, a -1e9 mask included)
//   mode 1, softmax over the cached KEYS (conventional incremental attention), offered beside it.
// A wave per (sample, head): lanes stride over the keys, each accumulating its keys' value rows, then a cross-lane sum.
template <typename T, int DH>
__global__ __launch_bounds__(64) void attn_decode_kernel(int64_t H, int64_t n_keys, int64_t t_max, const T* __restrict__ cache,
                                                          int64_t ld, int64_t k_off, int64_t q_off, int64_t v_off, int mode,
                                                          float scale, T* __restrict__ out, int64_t ld_out) {
  const int64_t b = blockIdx.x / H, hd = blockIdx.x % H;
  const int lane = threadIdx.x;
  const T* base = cache + b * t_max * ld + hd * DH;
  float q[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) q[d] = to_f32(base[(n_keys - 1) * ld + q_off + d]);
  float m = -INFINITY;
  if (mode == 1) {
    for (int64_t k = lane; k < n_keys; k += 64) {
      float l = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) l = fmaf(to_f32(base[k * ld + k_off + d]), q[d], l);
      m = fmaxf(m, l * scale);
    }
    m = wave_max(m);
  }
  float acc[DH], z = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) acc[d] = 0.f;
  for (int64_t k = lane; k < n_keys; k += 64) {
    float p = 1.f;
    if (mode == 1) {
      float l = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) l = fmaf(to_f32(base[k * ld + k_off + d]), q[d], l);
      p = __expf(l * scale - m);
    }
    z += p;
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = fmaf(p, to_f32(base[k * ld + v_off + d]), acc[d]);
  }
  z = wave_sum(z);
  const float inv = (mode == 1) ? 1.f / z : 1.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    const float v = wave_sum(acc[d]);
    if (lane == 0) out[b * ld_out + hd * DH + d] = from_f32<T>(v * inv);
  }
}

}  // namespace mst

using namespace mst;

extern "C" int mst_attn_keysoftmax_fwd(int dtype, int64_t B, int64_t S, int64_t H, int64_t dh, const void* qkv,
                                       int64_t ld_qkv, int64_t k_off, int64_t q_off, int64_t v_off,
                                       const uint8_t* keymask, float* lse, void* out, int64_t ld_out,
                                       int64_t q_limit, mst_stream_t stream) {
  int rc = attn_check(B, S, H, dh, ld_qkv, k_off, q_off, v_off);
  if (rc) return rc;
  MST_CHECK_ARG(qkv && keymask && lse && out, "mst_attn_keysoftmax_fwd: null pointer");
  MST_CHECK_ARG(ld_out >= H * dh, "mst_attn_keysoftmax_fwd: ld_out < H*dh");
  AttnArgs a = {};
  a.B = B; a.S = S; a.H = H; a.qkv = qkv; a.ld_qkv = ld_qkv; a.k_off = k_off; a.q_off = q_off; a.v_off = v_off;
  a.keymask = keymask; a.lse = lse; a.out = out; a.ld_out = ld_out;
  a.q_limit = (q_limit > 0 && q_limit < S) ? q_limit : S;
  a.scale = 1.f / sqrtf((float)dh);
  hipStream_t s = (hipStream_t)stream;
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    if (dh == 16) return launch_fwd<T, 16>(a, s);
    if (dh == 32) return launch_fwd<T, 32>(a, s);
    return launch_fwd<T, 64>(a, s);
  });
}

#ifdef MST_ATT_STAMPS
extern "C" int mst_debug_att_loop(uint64_t* host_out) {  // diagnostic builds only
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mst::g_att_loop), sizeof(uint64_t) * 256) == hipSuccess ? 0 : -1;
}
extern "C" int mst_debug_att_stamps(uint64_t* host_out) {  // diagnostic builds only: 1024 x 8 realtime stamps
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mst::g_att_stamps), sizeof(uint64_t) * 8192) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int mst_attn_qkv_fwd(int dtype, int64_t B, int64_t S, int64_t H, int64_t dh, const void* x, int64_t ld_x, const void* w,
                                int64_t ld_w, const float* bias, void* qkv, int64_t ld_qkv, int64_t k_off, int64_t q_off, int64_t v_off,
                                const uint8_t* keymask, float* lse, void* out, int64_t ld_out, int64_t q_limit, mst_stream_t stream) {
  int rc = attn_check(B, S, H, dh, ld_qkv, k_off, q_off, v_off);
  if (rc) return rc;
  const int64_t Dm = H * dh;
  MST_CHECK_ARG(x && w && bias && qkv && keymask && lse && out, "mst_attn_qkv_fwd: null pointer");
  MST_CHECK_ARG(ld_x % 8 == 0 && ld_x >= Dm && ld_w % 8 == 0 && ld_w >= Dm && ld_qkv >= 3 * Dm && ld_out >= Dm,
                "mst_attn_qkv_fwd: leading dimensions must be multiples of 8 and cover the model width");
  MST_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)bias % 16 == 0) && ((uintptr_t)qkv % 16 == 0),
                "mst_attn_qkv_fwd: operands must be 16-byte aligned");
  MST_CHECK_ARG(k_off + Dm <= 3 * Dm && q_off + Dm <= 3 * Dm && v_off + Dm <= 3 * Dm, "mst_attn_qkv_fwd: section offsets beyond the 3 D weight rows");
  hipStream_t s = (hipStream_t)stream;
  if (dh == 32) {
    AttnArgs a = {};
    a.B = B; a.S = S; a.H = H; a.qkv = qkv; a.ld_qkv = ld_qkv; a.k_off = k_off; a.q_off = q_off; a.v_off = v_off;
    a.keymask = keymask; a.lse = lse; a.out = out; a.ld_out = ld_out;
    a.q_limit = (q_limit > 0 && q_limit < S) ? q_limit : S;
    a.scale = 1.f / sqrtf((float)dh);
    a.x = x; a.ld_x = ld_x; a.w = w; a.ld_w = ld_w; a.bias = bias; a.Dm = Dm;
    rc = dispatch_act(dtype, [&](auto tag) -> int { return launch_fwd_qkv<decltype(tag)>(a, s); });
    if (rc != 0) return rc < 0 ? rc : MST_OK;
  }
  // shapes the fused form does not take: the projection as the GEMM it is, then the plain attention launch
  mst_gemm_args g = {};
  g.dtype = dtype; g.M = B * S; g.N = 3 * Dm; g.K = Dm;
  g.A = x; g.lda = ld_x; g.B = w; g.ldb = ld_w; g.C = qkv; g.ldc = ld_qkv; g.bias = bias; g.alpha = 1.f;
  rc = mst_gemm_nt(&g, stream);
  if (rc) return rc;
  return mst_attn_keysoftmax_fwd(dtype, B, S, H, dh, qkv, ld_qkv, k_off, q_off, v_off, keymask, lse, out, ld_out, q_limit, stream);
}

extern "C" int mst_attn_keysoftmax_bwd(int dtype, int64_t B, int64_t S, int64_t H, int64_t dh, const void* qkv,
                                       int64_t ld_qkv, int64_t k_off, int64_t q_off, int64_t v_off,
                                       const uint8_t* keymask, const float* lse, const void* dout, int64_t ld_dout,
                                       void* dqkv, int64_t ld_dqkv, float* delta, int64_t q_limit, mst_stream_t stream) {
  int rc = attn_check(B, S, H, dh, ld_qkv, k_off, q_off, v_off);
  if (rc) return rc;
  MST_CHECK_ARG(qkv && keymask && lse && dout && dqkv && delta, "mst_attn_keysoftmax_bwd: null pointer");
  MST_CHECK_ARG(ld_dout % 8 == 0 && ld_dout >= H * dh && ld_dqkv % 8 == 0, "mst_attn_keysoftmax_bwd: bad leading dims");
  AttnArgs a = {};
  a.B = B; a.S = S; a.H = H; a.qkv = qkv; a.ld_qkv = ld_qkv; a.k_off = k_off; a.q_off = q_off; a.v_off = v_off;
  a.keymask = keymask; a.lse = const_cast<float*>(lse); a.dout = dout; a.ld_dout = ld_dout; a.dqkv = dqkv;
  a.ld_dqkv = ld_dqkv; a.delta = delta;
  a.q_limit = (q_limit > 0 && q_limit < S) ? q_limit : 0;  // 0: dense dO
  a.scale = 1.f / sqrtf((float)dh);
  hipStream_t s = (hipStream_t)stream;
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    if (dh == 16) return launch_bwd<T, 16>(a, s);
    if (dh == 32) return launch_bwd<T, 32>(a, s);
    return launch_bwd<T, 64>(a, s);
  });
}

extern "C" int mst_attn_decode(int dtype, int64_t B, int64_t H, int64_t dh, int64_t n_keys, int64_t t_max, const void* cache,
                               int64_t ld, int64_t k_off, int64_t q_off, int64_t v_off, int mode, void* out, int64_t ld_out,
                               mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && H > 0 && n_keys > 0 && n_keys <= t_max && cache && out, "mst_attn_decode: bad argument");
  MST_CHECK_ARG(dh == 16 || dh == 32 || dh == 64, "mst_attn_decode: head size must be 16, 32 or 64 (got %lld)", (long long)dh);
  MST_CHECK_ARG(mode == 0 || mode == 1, "mst_attn_decode: mode must be 0 (softmax over the query axis, the reference) or 1 (over the keys)");
  MST_CHECK_ARG(ld_out >= H * dh, "mst_attn_decode: ld_out < H*dh");
  const float scale = 1.f / sqrtf((float)dh);
  hipStream_t s = (hipStream_t)stream;
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    const dim3 grid((unsigned)(B * H));
    if (dh == 16) hipLaunchKernelGGL((attn_decode_kernel<T, 16>), grid, dim3(64), 0, s, H, n_keys, t_max, (const T*)cache, ld, k_off, q_off, v_off, mode, scale, (T*)out, ld_out);
    else if (dh == 32) hipLaunchKernelGGL((attn_decode_kernel<T, 32>), grid, dim3(64), 0, s, H, n_keys, t_max, (const T*)cache, ld, k_off, q_off, v_off, mode, scale, (T*)out, ld_out);
    else hipLaunchKernelGGL((attn_decode_kernel<T, 64>), grid, dim3(64), 0, s, H, n_keys, t_max, (const T*)cache, ld, k_off, q_off, v_off, mode, scale, (T*)out, ld_out);
    MST_CHECK_LAUNCH("attn_decode_kernel");
    return MST_OK;
  });
}

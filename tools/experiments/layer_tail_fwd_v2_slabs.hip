// layer_tail.hip — the row-wise part of a Transformer encoder layer (everything after the attention mix) on a SMALL,
// strided set of rows, one launch:
//     h1 = x_in + dropout(att W_proj^T + b)      x1 = LayerNorm1(h1)
//     a  = dropout(relu(x1 W_ff1^T + b))         h2 = x1 + dropout(a W_ff2^T + b)      x2 = LayerNorm2(h2)
// (VarAutoEncoder/transformer.py:150-159). The model reads the top encoder layer at position 0 only
// (VarAutoEncoder/model.py:97), so that layer's tail runs on B rows (64 at configs[1]) of the B*T: as separate GEMM /
// LayerNorm launches that was 5 kernels of 5-15 us each (40 us) whose arithmetic is microseconds. Here a workgroup owns 16
// rows (one MFMA row block) for the whole chain: activations stay in LDS, the sixteen waves split every layer's output
// columns, and the weights — all 1.2 MB of them, the bound of the kernel: every workgroup has to read each weight once —
// stream through LDS in 256-row x 64-column slabs (one full 128-byte line per row, fetched with perfectly coalesced
// 16-byte loads: a CU pulls ~127 GB/s from L2 that way, measured, against 57 GB/s for 64-byte pieces read straight into
// MFMA operand order), four slabs in flight in registers across stage boundaries (weights depend on nothing).
// Every intermediate the backward pass needs (h1, x1, a, h2, x2, the statistics) is written where the unfused sequence
// wrote it, with the same rounding points and the same dropout counters (physical output row x width + column).
#include <type_traits>
#include "common.hpp"

namespace mst {

constexpr int LT_ROWS = 16;     // rows per workgroup (one 16x16x32 MFMA row block)
constexpr int LT_WAVES = 16;
constexpr int LT_SN = 256;      // weight rows of a slab: one 16-row tile per wave
constexpr int LT_SK = 64;       // k elements of a slab: one 128-byte line per row
constexpr int LT_SLD = LT_SK + 8;  // LDS row stride of a slab (elements)
constexpr int LT_RING = 4;      // slabs in flight in registers
constexpr int LT_PAD = 8;       // LDS row padding of the activation buffers (elements)
constexpr int LT_THREADS = LT_WAVES * 64;

// The weight stream of the whole chain: stage st has its matrix W ([N, K] row-major, row stride ldw), walked as groups of
// 256 rows x slabs of 64 k. The LOAD position runs LT_RING slabs ahead of the COMPUTE position, across stage boundaries.
// Everything that moves per slab is wave-uniform (a base pointer and three counters: scalar registers); the per-thread
// part of an address is a 32-bit offset that changes only when the stream enters the next matrix. (A first version
// recomputed 64-bit per-thread addresses and stage selects per slab: ~100 vector instructions x 16 waves x 36 slabs cost
// more than the 1.2 MB of loads.)
template <typename T>
struct SlabPos {
  const T* base;          // uniform: row 0 of the current group, column of the current slab
  uint32_t off0, off1;    // per thread (elements): (tid / 8) * ld + (tid % 8) * 8, and 128 rows further
  int ks_left, g_left;    // slabs left in this group (including the current one), groups left in this matrix
  int st;                 // matrix of the load position
  int64_t ld;
  int nks;
  int parity;             // LDS buffer of the next slab to be computed
};

template <typename T>
__device__ __forceinline__ void slab_enter(const mst_layer_tail_args& pl, SlabPos<T>& p, int st, int tid) {
  // uniform branches on st, the fields read straight from the kernel arguments (a local copy of them was placed in scratch
  // memory and indexed); readfirstlane pins the stream state to scalar registers
  const void* w = pl.w_proj; int64_t ld = pl.ld_wproj; int64_t n = pl.D, k = pl.D;
  if (st == 1) { w = pl.w_ff1; ld = pl.ld_wff1; n = pl.F; k = pl.D; }
  if (st == 2) { w = pl.w_ff2; ld = pl.ld_wff2; n = pl.D; k = pl.F; }
  p.st = st; p.base = reinterpret_cast<const T*>(w); p.ld = ld; p.nks = __builtin_amdgcn_readfirstlane((int)(k / LT_SK));
  p.ks_left = p.nks; p.g_left = __builtin_amdgcn_readfirstlane((int)(n / LT_SN));
  p.off0 = (uint32_t)(tid >> 3) * (uint32_t)ld + (uint32_t)(tid & 7) * 8u;
  p.off1 = p.off0 + 128u * (uint32_t)ld;
}

// Branch-free loads (a conditional load makes the compiler wait for ALL outstanding loads at the join): every matrix has
// whole 256-row groups (host check), and past the end of the plan the last slab is simply requested again.
template <typename T>
__device__ __forceinline__ void slab_load_next(const mst_layer_tail_args& pl, SlabPos<T>& p, u32x4 (&r)[2], int tid) {
  typedef const __attribute__((address_space(1))) u32x4* gptr_t;
  r[0] = *(gptr_t)(uintptr_t)(p.base + p.off0);
  r[1] = *(gptr_t)(uintptr_t)(p.base + p.off1);
  const bool at_end = p.st == 2 && p.g_left == 1 && p.ks_left == 1;
  if (!at_end) {
    p.base += LT_SK;
    p.ks_left = __builtin_amdgcn_readfirstlane(p.ks_left - 1);
    if (p.ks_left == 0) {
      p.base += (int64_t)LT_SN * p.ld - (int64_t)p.nks * LT_SK;
      p.ks_left = p.nks;
      p.g_left = __builtin_amdgcn_readfirstlane(p.g_left - 1);
      if (p.g_left == 0) slab_enter<T>(pl, p, __builtin_amdgcn_readfirstlane(p.st + 1), tid);
    }
  }
}
template <typename T>
__device__ __forceinline__ void slab_store(const u32x4 (&r)[2], T* buf, int tid) {
#pragma unroll
  for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4*>(buf + ((tid >> 3) + h * 128) * LT_SLD + (tid & 7) * 8) = r[h];
}
// first LT_RING slabs requested, slab 0 in LDS (ends with a barrier)
template <typename T>
__device__ __forceinline__ void slab_start(const mst_layer_tail_args& pl, SlabPos<T>& p, u32x4 (&ring)[LT_RING][2], T* sW, int tid) {
  slab_enter<T>(pl, p, 0, tid);
  p.parity = 0;
  slab_load_next<T>(pl, p, ring[0], tid);
  slab_load_next<T>(pl, p, ring[1], tid);
  slab_load_next<T>(pl, p, ring[2], tid);
  slab_load_next<T>(pl, p, ring[3], tid);
  slab_store<T>(ring[0], sW, tid);
  __syncthreads();
}
// One stage (n rows of W, k columns; its slab count is a multiple of LT_RING, so slab s of the stage sits in ring slot
// s % LT_RING): C[16, n] = A[16, k] (LDS, row stride lda) x W^T; epi(tile, acc) gets lane (frow, fq)'s columns
// tile*16 + fq*4 .. +3 of row frow when a 16-column tile is complete. Every wave must call it (barriers).
template <typename T, typename Epi>
__device__ __forceinline__ void slab_run_stage(const mst_layer_tail_args& pl, SlabPos<T>& p, u32x4 (&ring)[LT_RING][2], int n, int k, T* sW,
                                               const T* sA, int lda, int tid, Epi&& epi) {
  typedef typename Act<T>::vec8 vec8;
  const int lane = tid & 63, wave = tid >> 6, frow = lane & 15, fq = lane >> 4;
  const int nks = k / LT_SK;
  const int slabs = (n / LT_SN) * nks;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int ks = 0, g = 0;
  const T* ap = sA + frow * lda + fq * 8;
  // one slab: `slot` is a compile-time index (the ring must stay in registers; a run-time slot put it on the stack)
  auto step = [&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    T* cur = sW + p.parity * (LT_SN * LT_SLD);
    T* nxt = sW + (p.parity ^ 1) * (LT_SN * LT_SLD);
    slab_store<T>(ring[(slot + 1) % LT_RING], nxt, tid);      // slab s + 1 -> the other LDS buffer (its readers passed the last barrier)
    slab_load_next<T>(pl, p, ring[slot], tid);                // slab s + LT_RING (this slot's slab s is in LDS already)
    const int tile = g * (LT_SN / 16) + wave;  // (n % 256 == 0: every wave has a tile in every group)
    {
      const T* wp = cur + (wave * 16 + frow) * LT_SLD + fq * 8;
#pragma unroll
      for (int kk = 0; kk < LT_SK / 32; ++kk) {
        const u32x4 wv = *reinterpret_cast<const u32x4*>(wp + kk * 32);
        const u32x4 av = *reinterpret_cast<const u32x4*>(ap + ks * LT_SK + kk * 32);
        acc = Act<T>::mfma16(__builtin_bit_cast(vec8, wv), __builtin_bit_cast(vec8, av), acc);
      }
    }
    if (++ks == nks) {
      epi(tile, acc);
      acc = f32x4{0.f, 0.f, 0.f, 0.f};
      ks = 0;
      ++g;
    }
    p.parity ^= 1;
    __syncthreads();
  };
  static_assert(LT_RING == 4, "the slab loop below is written out for four ring slots");
  for (int s0 = 0; s0 < slabs; s0 += LT_RING) {
    step(std::integral_constant<int, 0>());
    step(std::integral_constant<int, 1>());
    step(std::integral_constant<int, 2>());
    step(std::integral_constant<int, 3>());
  }
}

template <typename T>
__device__ __forceinline__ void unpack4(u32x2 r, float v[4]) {
  v[0] = bits_to_f32<T>((uint16_t)(r[0] & 0xffff));
  v[1] = bits_to_f32<T>((uint16_t)(r[0] >> 16));
  v[2] = bits_to_f32<T>((uint16_t)(r[1] & 0xffff));
  v[3] = bits_to_f32<T>((uint16_t)(r[1] >> 16));
}
// rounds v to the activation type in place and returns the packed bits
template <typename T>
__device__ __forceinline__ u32x2 round_pack4(float v[4]) {
  uint16_t b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { b[e] = f32_to_bits<T>(v[e]); v[e] = bits_to_f32<T>(b[e]); }
  return u32x2{(uint32_t)b[0] | ((uint32_t)b[1] << 16), (uint32_t)b[2] | ((uint32_t)b[3] << 16)};
}

// sum over the columns of a row: the four fq lanes of a row inside the wave, then the sixteen waves through LDS.
// red: [LT_WAVES][LT_ROWS]; contains a barrier — every wave must call it.
__device__ __forceinline__ float row_total(float s, float* red, int wave, int lane) {
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (lane < 16) red[wave * LT_ROWS + lane] = s;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < LT_WAVES; ++w) t += red[w * LT_ROWS + (lane & 15)];
  return t;
}

constexpr int LT_TPW = 2;  // column tiles of a D-wide row per wave: D <= 16 * 16 * LT_TPW = 512

// LayerNorm of the rows whose (rounded) elements the waves hold in v[t][4] (tile wave + 16 t): y to global and to the
// LDS operand buffer, statistics to mean / rstd. Two-pass statistics like layernorm_fwd_kernel. Contains barriers.
template <typename T>
__device__ __forceinline__ void tail_layernorm(const float (&v)[LT_TPW][4], int n_tiles, int D, const float* sG, const float* sB,
                                               float eps, float* red0, float* red1, T* y_row, T* sY, int ldy_lds, bool live,
                                               float* mean_out, float* rstd_out, int wave, int lane) {
  const int fq = lane >> 4;
  const float inv_d = 1.f / (float)D;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < LT_TPW; ++t)
    if (wave + t * LT_WAVES < n_tiles) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
  const float mean = row_total(s, red0, wave, lane) * inv_d;
  float ss = 0.f;
#pragma unroll
  for (int t = 0; t < LT_TPW; ++t)
    if (wave + t * LT_WAVES < n_tiles) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[t][e] - mean; ss += d * d; }
    }
  const float rstd = 1.f / sqrtf(row_total(ss, red1, wave, lane) * inv_d + eps);
#pragma unroll
  for (int t = 0; t < LT_TPW; ++t) {
    const int tile = wave + t * LT_WAVES;
    if (tile < n_tiles) {
      const int n = tile * 16 + fq * 4;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[t][e] - mean) * rstd * sG[n + e] + sB[n + e];
      const u32x2 ob = round_pack4<T>(o);
      if (live) *reinterpret_cast<u32x2*>(y_row + n) = ob;
      *reinterpret_cast<u32x2*>(sY + (lane & 15) * ldy_lds + n) = ob;
    }
  }
  if (wave == 0 && lane < 16 && live) { *mean_out = mean; *rstd_out = rstd; }
}

template <typename T>
// (waves_per_eu 4: one 16-wave workgroup per CU is all LDS allows — without the hint the compiler aimed at 5 waves per
// SIMD, i.e. 96 registers, and spilled the loop state)
__global__ __launch_bounds__(LT_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void layer_tail_fwd_kernel(mst_layer_tail_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lt_smem[];
  const int D = (int)a.D, F = (int)a.F;
  const int ldx = D + LT_PAD, ldh = F + LT_PAD;
  T* sX = reinterpret_cast<T*>(lt_smem);             // [16][D+8]: attention rows, then x1
  T* sH = sX + LT_ROWS * ldx;                        // [16][F+8]: the FFN's hidden activation
  float* sP = reinterpret_cast<float*>(sH + LT_ROWS * ldh);  // b_proj, gamma1, beta1, b_ff2, gamma2, beta2 [D each], b_ff1 [F]
  float* red0 = sP + 6 * D + F;                      // [16 waves][16 rows]
  float* red1 = red0 + LT_WAVES * LT_ROWS;
  T* sW = reinterpret_cast<T*>(red1 + LT_WAVES * LT_ROWS);  // [2][256][64+8]: the weight slabs
  SlabPos<T> pos;
  u32x4 ring[LT_RING][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 15, fq = lane >> 4;
  const int64_t m = (int64_t)blockIdx.x * LT_ROWS + frow;
  const bool live = m < a.M;
  const int64_t pm = (live ? m : 0) * a.row_stride;  // physical row of every activation buffer and of the statistics
  const float p = a.dropout_p;
  const bool has_drop = p > 0.f;
  const uint64_t seed = a.dropout_seed ^ ((has_drop && a.dropout_seed_ptr) ? a.dropout_seed_ptr[0] : 0ull);
  const uint32_t dthr = dropout_thr(p);
  const float inv_keep = dropout_inv_keep(p);
  const int tiles_d = D / 16, tiles_f = F / 16;

  // ---- stage 0: the attention rows and the small parameter vectors into LDS
  {
    const int cpr = D / 8;
    const T* att = reinterpret_cast<const T*>(a.att);
    for (int c = tid; c < LT_ROWS * cpr; c += LT_WAVES * 64) {
      const int row = c / cpr, ch = c % cpr;
      const int64_t mr = (int64_t)blockIdx.x * LT_ROWS + row;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (mr < a.M) v = *reinterpret_cast<const u32x4*>(att + mr * a.row_stride * a.ld_att + ch * 8);
      *reinterpret_cast<u32x4*>(sX + row * ldx + ch * 8) = v;
    }
    for (int c = tid; c < D; c += LT_WAVES * 64) {
      sP[c] = a.b_proj[c]; sP[D + c] = a.gamma1[c]; sP[2 * D + c] = a.beta1[c];
      sP[3 * D + c] = a.b_ff2[c]; sP[4 * D + c] = a.gamma2[c]; sP[5 * D + c] = a.beta2[c];
    }
    for (int c = tid; c < F; c += LT_WAVES * 64) sP[6 * D + c] = a.b_ff1[c];
  }
  // the residual rows of stage 1, requested before the first barrier
  u32x2 xres[LT_TPW];
#pragma unroll
  for (int t = 0; t < LT_TPW; ++t) {
    xres[t] = u32x2{0u, 0u};
    const int tile = wave + t * LT_WAVES;
    if (tile < tiles_d && live)
      xres[t] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(a.x_in) + pm * a.ld_x + tile * 16 + fq * 4);
  }
  slab_start<T>(a, pos, ring, sW, tid);  // (ends with the barrier that publishes stage 0's LDS writes)

  // ---- stage 1: h1 = x_in + dropout(att W_proj^T + b), x1 = LayerNorm1(h1)
  float v[LT_TPW][4];
#pragma unroll
  for (int t = 0; t < LT_TPW; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[t][e] = 0.f;
  {
    const uint32_t dkey = dropout_key(seed, a.site0);
    T* h1 = reinterpret_cast<T*>(a.h1) + pm * a.ld_h1;
    slab_run_stage<T>(a, pos, ring, D, D, sW, sX, ldx, tid, [&](int tile, const f32x4& acc) {
      const int n = tile * 16 + fq * 4;
      float t4[4], r4[4];
      const int ti = (tile - wave) / LT_WAVES;
      unpack4<T>(ti == 0 ? xres[0] : xres[1], r4);
      uint32_t keep = 0xFu;
      if (has_drop) keep = dropout_keep4k(dkey, (uint64_t)(pm * D + n) >> 2, dthr);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = acc[e] + sP[n + e];
        if (has_drop) u = ((keep >> e) & 1u) ? u * inv_keep : 0.f;
        t4[e] = u + r4[e];
      }
      const u32x2 hb = round_pack4<T>(t4);
      if (live) *reinterpret_cast<u32x2*>(h1 + n) = hb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ti == 0) v[0][e] = t4[e]; else v[1][e] = t4[e];
      }
    });
  }
  __syncthreads();  // every wave is done reading the attention rows: sX becomes x1
  tail_layernorm<T>(v, tiles_d, D, sP + D, sP + 2 * D, a.eps, red0, red1, reinterpret_cast<T*>(a.x1) + pm * a.ld_x1, sX, ldx, live,
                    a.mean1 + pm, a.rstd1 + pm, wave, lane);
  __syncthreads();

  // ---- stage 2: a = dropout(relu(x1 W_ff1^T + b))
  {
    const uint32_t dkey = dropout_key(seed, a.site0 + 1);
    T* arow = reinterpret_cast<T*>(a.a) + pm * a.ld_a;
    slab_run_stage<T>(a, pos, ring, F, D, sW, sX, ldx, tid, [&](int tile, const f32x4& acc) {
      const int n = tile * 16 + fq * 4;
      float t4[4];
      uint32_t keep = 0xFu;
      if (has_drop) keep = dropout_keep4k(dkey, (uint64_t)(pm * F + n) >> 2, dthr);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = fmaxf(acc[e] + sP[6 * D + n + e], 0.f);
        if (has_drop) u = ((keep >> e) & 1u) ? u * inv_keep : 0.f;
        t4[e] = u;
      }
      const u32x2 ab = round_pack4<T>(t4);
      if (live) *reinterpret_cast<u32x2*>(arow + n) = ab;
      *reinterpret_cast<u32x2*>(sH + frow * ldh + n) = ab;
    });
  }
  __syncthreads();

  // ---- stage 3: h2 = x1 + dropout(a W_ff2^T + b), x2 = LayerNorm2(h2)
  {
    const uint32_t dkey = dropout_key(seed, a.site0 + 2);
    T* h2 = reinterpret_cast<T*>(a.h2) + pm * a.ld_h2;
    slab_run_stage<T>(a, pos, ring, D, F, sW, sH, ldh, tid, [&](int tile, const f32x4& acc) {
      const int n = tile * 16 + fq * 4;
      float t4[4], r4[4];
      const int ti = (tile - wave) / LT_WAVES;
      unpack4<T>(*reinterpret_cast<const u32x2*>(sX + frow * ldx + n), r4);  // x1, as stored
      uint32_t keep = 0xFu;
      if (has_drop) keep = dropout_keep4k(dkey, (uint64_t)(pm * D + n) >> 2, dthr);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = acc[e] + sP[3 * D + n + e];
        if (has_drop) u = ((keep >> e) & 1u) ? u * inv_keep : 0.f;
        t4[e] = u + r4[e];
      }
      const u32x2 hb = round_pack4<T>(t4);
      if (live) *reinterpret_cast<u32x2*>(h2 + n) = hb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ti == 0) v[0][e] = t4[e]; else v[1][e] = t4[e];
      }
    });
  }
  __syncthreads();  // red0 / red1 are reused; sH is dead and receives x2 (unused)
  tail_layernorm<T>(v, tiles_d, D, sP + 4 * D, sP + 5 * D, a.eps, red0, red1, reinterpret_cast<T*>(a.x2) + pm * a.ld_x2, sH, ldh, live,
                    a.mean2 + pm, a.rstd2 + pm, wave, lane);
}

static size_t lt_fwd_lds(int64_t D, int64_t F) {
  return (size_t)LT_ROWS * (D + LT_PAD) * 2 + (size_t)LT_ROWS * (F + LT_PAD) * 2 + (size_t)(6 * D + F) * 4 + (size_t)2 * LT_WAVES * LT_ROWS * 4 +
         (size_t)2 * LT_SN * LT_SLD * 2;
}

}  // namespace mst

using namespace mst;

static int lt_check_shapes(const char* who, int64_t M, int64_t D, int64_t F) {
  MST_CHECK_ARG(M > 0, "%s: M must be positive", who);
  // (every stage's slab count K/64 * ceil(N/256) has to be a multiple of the register ring: K % 256 == 0)
  MST_CHECK_ARG(D % 256 == 0 && D >= 256 && D <= 16 * LT_WAVES * LT_TPW, "%s: D must be 256 or 512 (got %lld)", who, (long long)D);
  MST_CHECK_ARG(F % 256 == 0 && F >= 256 && F <= 4096, "%s: F must be a multiple of 256, at most 4096 (got %lld)", who, (long long)F);
  return MST_OK;
}

extern "C" int mst_layer_tail_fwd(const mst_layer_tail_args* args, mst_stream_t stream) {
  MST_CHECK_ARG(args != nullptr, "mst_layer_tail_fwd: null args");
  const mst_layer_tail_args& a = *args;
  int rc = lt_check_shapes("mst_layer_tail_fwd", a.M, a.D, a.F);
  if (rc) return rc;
  MST_CHECK_ARG(a.row_stride >= 1, "mst_layer_tail_fwd: row_stride must be >= 1");
  MST_CHECK_ARG(a.att && a.x_in && a.w_proj && a.w_ff1 && a.w_ff2 && a.b_proj && a.b_ff1 && a.b_ff2 && a.gamma1 && a.beta1 &&
                a.gamma2 && a.beta2 && a.h1 && a.x1 && a.a && a.h2 && a.x2 && a.mean1 && a.rstd1 && a.mean2 && a.rstd2,
                "mst_layer_tail_fwd: null pointer");
  MST_CHECK_ARG(a.ld_att % 8 == 0 && a.ld_att >= a.D && (uintptr_t)a.att % 16 == 0, "mst_layer_tail_fwd: bad att layout");
  MST_CHECK_ARG(a.ld_x % 4 == 0 && a.ld_h1 % 4 == 0 && a.ld_x1 % 4 == 0 && a.ld_a % 4 == 0 && a.ld_h2 % 4 == 0 && a.ld_x2 % 4 == 0 &&
                a.ld_x >= a.D && a.ld_h1 >= a.D && a.ld_x1 >= a.D && a.ld_a >= a.F && a.ld_h2 >= a.D && a.ld_x2 >= a.D,
                "mst_layer_tail_fwd: leading dimensions must be multiples of 4 and cover the row");
  MST_CHECK_ARG(((uintptr_t)a.x_in | (uintptr_t)a.h1 | (uintptr_t)a.x1 | (uintptr_t)a.a | (uintptr_t)a.h2 | (uintptr_t)a.x2) % 8 == 0,
                "mst_layer_tail_fwd: activation buffers must be 8-byte aligned");
  MST_CHECK_ARG(a.ld_wproj % 8 == 0 && a.ld_wff1 % 8 == 0 && a.ld_wff2 % 8 == 0 && a.ld_wproj >= a.D && a.ld_wff1 >= a.D && a.ld_wff2 >= a.F &&
                ((uintptr_t)a.w_proj | (uintptr_t)a.w_ff1 | (uintptr_t)a.w_ff2) % 16 == 0, "mst_layer_tail_fwd: bad weight layout");
  MST_CHECK_ARG(a.dropout_p >= 0.f && a.dropout_p < 1.f, "mst_layer_tail_fwd: dropout_p must be in [0,1)");
  const size_t lds = lt_fwd_lds(a.D, a.F);
  MST_CHECK_ARG(lds <= 160 * 1024, "mst_layer_tail_fwd: D, F too large for LDS");
  const unsigned grid = (unsigned)cdiv(a.M, LT_ROWS);
  return dispatch_act(a.dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    if (lds > 64 * 1024) {
      static size_t opted = 64 * 1024;
      if (lds > opted) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_tail_fwd_kernel<T>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("layer_tail_fwd_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
        opted = lds;
      }
    }
    hipLaunchKernelGGL((layer_tail_fwd_kernel<T>), dim3(grid), dim3(LT_WAVES * 64), lds, (hipStream_t)stream, a);
    MST_CHECK_LAUNCH("layer_tail_fwd_kernel");
    return MST_OK;
  });
}

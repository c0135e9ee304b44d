// layer_tail.hip — the row-wise part of a Transformer encoder layer (everything after the attention mix) on a SMALL,
// strided set of rows, one launch per direction:
//     h1 = x_in + dropout(att W_proj^T + b)      x1 = LayerNorm1(h1)
//     a  = dropout(relu(x1 W_ff1^T + b))         h2 = x1 + dropout(a W_ff2^T + b)      x2 = LayerNorm2(h2)
// (VarAutoEncoder/transformer.py:150-159) and its backward. The model reads the top encoder layer at position 0 only
// (VarAutoEncoder/model.py:97), so that layer's tail runs on B rows (64 at configs[1]) of the B*T: as separate GEMM /
// LayerNorm launches that was 5 + 5 kernels of 5-15 us each whose arithmetic is microseconds — 80 us of the 930 us step
// went to launch floors and to 4-workgroup GEMMs. Here a workgroup owns 16 rows (one MFMA row block) for the whole
// chain: activations stay in LDS, the sixteen waves split every layer's output columns, and the weights stream from
// L2 straight into the MFMA B operands through a four-deep register ring (each workgroup reads every weight once:
// ~1.2 MB, the bound of the kernel). Every intermediate the other kernels need (h1, x1, a, h2, x2, the statistics;
// backward: the operands of the weight-gradient launch) is written where the unfused sequence wrote it, with the same
// rounding points and the same dropout counters (physical output row x width + column), so either path can be used.
#include "common.hpp"

namespace mst {

constexpr int LT_ROWS = 16;     // rows per workgroup (one 16x16x32 MFMA row block)
constexpr int LT_WAVES = 16;
constexpr int LT_CH = 128;      // k elements per ring slot (4 MFMA k-steps)
constexpr int LT_KS = LT_CH / 32;
constexpr int LT_RING = 4;
constexpr int LT_PAD = 8;       // LDS row padding (elements)

// One wave's share of C[16, N] = A[16, K] (LDS, row stride lda) x W[N, K]^T (global, row stride ldw): the 16-column
// tiles tile0, tile0 + 16, ... ; epi(tile, acc) receives lane (frow, fq)'s four consecutive columns tile*16 + fq*4 ..
// of row frow. The weight fragments of the next three 128-k slots (across tile boundaries) are always in flight.
template <typename T, typename Epi>
__device__ __forceinline__ void wave_tiles(const T* __restrict__ W, int64_t ldw, int K, int n_tiles, const T* sA, int lda,
                                           int wave, int lane, Epi&& epi) {
  typedef typename Act<T>::vec8 vec8;
  const int frow = lane & 15, fq = lane >> 4;
  const int nch = K / LT_CH;
  const int my_tiles = wave < n_tiles ? (n_tiles - wave + LT_WAVES - 1) / LT_WAVES : 0;
  const int items = my_tiles * nch;
  if (items == 0) return;
  u32x4 ring[LT_RING][LT_KS];
  // Global loads are issued with lane l on row l / 4, 16-byte piece l % 4: four adjacent lanes cover 64 contiguous bytes
  // (one quarter-wave = 4 rows). In MFMA operand order (lane = row + 16 * piece) the four lanes the memory pipeline
  // handles together sit on four different rows: 64 tag lookups per instruction, measured 42 us for the whole chain.
  // A ds_bpermute per dword puts the fragment into operand order afterwards.
  const int src_lane4 = (frow * 4 + fq) * 4;  // byte address of the lane that loaded (row frow, piece fq)
  const T* lp = W + (int64_t)(wave * 16 + (lane >> 2)) * ldw + (lane & 3) * 8;
  int l_c = 0, l_it = 0;
  auto load_next = [&](u32x4 (&r)[LT_KS]) {
#pragma unroll
    for (int s = 0; s < LT_KS; ++s) r[s] = *reinterpret_cast<const u32x4*>(lp + s * 32);
    ++l_it;
    if (++l_c == nch) { l_c = 0; lp += (int64_t)LT_WAVES * 16 * ldw - (int64_t)(nch - 1) * LT_CH; }
    else lp += LT_CH;
  };
#pragma unroll
  for (int s = 0; s < LT_RING - 1; ++s)
    if (s < items) load_next(ring[s]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int c = 0, tile = wave;
  const T* ap = sA + frow * lda + fq * 8;
  for (int it0 = 0; it0 < items; it0 += LT_RING) {
#pragma unroll
    for (int s = 0; s < LT_RING; ++s) {
      if (it0 + s < items) {
        if (l_it < items) load_next(ring[(s + LT_RING - 1) % LT_RING]);
#pragma unroll
        for (int k = 0; k < LT_KS; ++k) {
          const u32x4 av = *reinterpret_cast<const u32x4*>(ap + c * LT_CH + k * 32);
          u32x4 wv;
#pragma unroll
          for (int d = 0; d < 4; ++d) wv[d] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)ring[s][k][d]);
          acc = Act<T>::mfma16(__builtin_bit_cast(vec8, wv), __builtin_bit_cast(vec8, av), acc);
        }
        if (++c == nch) {
          epi(tile, acc);
          acc = f32x4{0.f, 0.f, 0.f, 0.f};
          c = 0;
          tile += LT_WAVES;
        }
      }
    }
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
__global__ __launch_bounds__(LT_WAVES * 64) void layer_tail_fwd_kernel(mst_layer_tail_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lt_smem[];
  const int D = (int)a.D, F = (int)a.F;
  const int ldx = D + LT_PAD, ldh = F + LT_PAD;
  T* sX = reinterpret_cast<T*>(lt_smem);             // [16][D+8]: attention rows, then x1
  T* sH = sX + LT_ROWS * ldx;                        // [16][F+8]: the FFN's hidden activation
  float* sP = reinterpret_cast<float*>(sH + LT_ROWS * ldh);  // b_proj, gamma1, beta1, b_ff2, gamma2, beta2 [D each], b_ff1 [F]
  float* red0 = sP + 6 * D + F;                      // [16 waves][16 rows]
  float* red1 = red0 + LT_WAVES * LT_ROWS;
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
  __syncthreads();

  // ---- stage 1: h1 = x_in + dropout(att W_proj^T + b), x1 = LayerNorm1(h1)
  float v[LT_TPW][4];
#pragma unroll
  for (int t = 0; t < LT_TPW; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[t][e] = 0.f;
  {
    const uint32_t dkey = dropout_key(seed, a.site0);
    T* h1 = reinterpret_cast<T*>(a.h1) + pm * a.ld_h1;
    wave_tiles<T>(reinterpret_cast<const T*>(a.w_proj), a.ld_wproj, D, tiles_d, sX, ldx, wave, lane, [&](int tile, const f32x4& acc) {
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
    wave_tiles<T>(reinterpret_cast<const T*>(a.w_ff1), a.ld_wff1, D, tiles_f, sX, ldx, wave, lane, [&](int tile, const f32x4& acc) {
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
    wave_tiles<T>(reinterpret_cast<const T*>(a.w_ff2), a.ld_wff2, F, tiles_d, sH, ldh, wave, lane, [&](int tile, const f32x4& acc) {
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
  return (size_t)LT_ROWS * (D + LT_PAD) * 2 + (size_t)LT_ROWS * (F + LT_PAD) * 2 + (size_t)(6 * D + F) * 4 + (size_t)2 * LT_WAVES * LT_ROWS * 4;
}

}  // namespace mst

using namespace mst;

static int lt_check_shapes(const char* who, int64_t M, int64_t D, int64_t F) {
  MST_CHECK_ARG(M > 0, "%s: M must be positive", who);
  MST_CHECK_ARG(D % LT_CH == 0 && D >= LT_CH && D <= 16 * LT_WAVES * LT_TPW, "%s: D must be a multiple of %d in [%d, %d] (got %lld)", who,
                LT_CH, LT_CH, 16 * LT_WAVES * LT_TPW, (long long)D);
  MST_CHECK_ARG(F % LT_CH == 0 && F >= LT_CH && F <= 4096, "%s: F must be a multiple of %d, at most 4096 (got %lld)", who, LT_CH, (long long)F);
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

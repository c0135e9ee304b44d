// layernorm.hip — LayerNorm forward / backward over the last axis, one wave per row.
//
// Replaces gluon.nn.LayerNorm (eps 1e-5, biased variance, gamma/beta) at
// VarAutoEncoder/transformer.py:142,147,175,180 as applied at :155,158,197,200. The residual
// add that precedes every LayerNorm in the reference is fused into the producing GEMM's
// epilogue (gemm_nt.hip), so x here is the pre-norm sum.
//
// HBM-bound streaming kernels: 8-byte vector loads (4 x 16-bit per lane → a 256-wide row is one
// fully coalesced 512-byte wave access), all statistics in fp32 registers, two-pass variance.
// Backward also produces the dropout-masked copy of dx that the producing GEMM's dgrad/wgrad
// consume when dropout is enabled (mask regenerated from the counter RNG, never stored).
#include <type_traits>
#include "common.hpp"

namespace mst {

constexpr int LN_MAXV = 4;  // 4 elements * 64 lanes * LN_MAXV = D up to 1024 (kernel template parameter NV <= LN_MAXV)

template <typename T>
__device__ __forceinline__ void load4(const T* p, float v[4]) {
  u32x2 r = *reinterpret_cast<const u32x2*>(p);
  v[0] = bits_to_f32<T>((uint16_t)(r[0] & 0xffff));
  v[1] = bits_to_f32<T>((uint16_t)(r[0] >> 16));
  v[2] = bits_to_f32<T>((uint16_t)(r[1] & 0xffff));
  v[3] = bits_to_f32<T>((uint16_t)(r[1] >> 16));
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float v[4]) {
  u32x2 o;
  o[0] = (uint32_t)f32_to_bits<T>(v[0]) | ((uint32_t)f32_to_bits<T>(v[1]) << 16);
  o[1] = (uint32_t)f32_to_bits<T>(v[2]) | ((uint32_t)f32_to_bits<T>(v[3]) << 16);
  *reinterpret_cast<u32x2*>(p) = o;
}

// NV = 64-lane vector groups per row (D <= 256 NV): 1 for D <= 256, 2 for <= 512, 4 for <= 1024. R = rows a wave works
// on at once: all R rows' loads are issued before the first reduction, so a wave pays one memory round trip per R
// rows instead of one per row (the kernels are pure streaming with 2-4 rows per wave at configs[1]).
template <typename T, int NV, int R>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(int64_t M, int D, const T* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            T* __restrict__ y, int64_t ldy,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int64_t row_id_stride) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int nvec = D / 4;
  const float inv_d = 1.f / (float)D;
  for (int64_t m0 = wave_global; m0 < M; m0 += nwaves * R) {
    float v[R][NV][4];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t m = m0 + r * nwaves;
      s[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[r][i][e] = 0.f;
        if (m < M && c < nvec) {
          load4<T>(x + m * ldx + c * 4, v[r][i]);
          s[r] += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
        }
      }
    }
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) * inv_d;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (lane + i * 64 < nvec) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[r][i][e] - mean[r]; ss += d * d; }
        }
      }
      s[r] = ss;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) rstd[r] = 1.f / sqrtf(wave_sum(s[r]) * inv_d + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nvec) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c * 4);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int64_t m = m0 + r * nwaves;
          if (m < M) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[r][i][e] - mean[r]) * rstd[r] * g[e] + b[e];
            store4<T>(y + m * ldy + c * 4, o);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t m = m0 + r * nwaves;
      if (lane == 0 && m < M) { mean_out[m * row_id_stride] = mean[r]; rstd_out[m * row_id_stride] = rstd[r]; }
    }
  }
}

// mask_mode: 0 = dx only; 1 = dx and dxm = dx * keep/(1-p); 2 = dx <- dx * (1 + keep/(1-p))
template <typename T, int NV, int R>
__global__ __launch_bounds__(1024) void layernorm_bwd_kernel(int64_t M, int D, const T* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in,
                                                            const T* __restrict__ dy, int64_t ldy,
                                                            T* __restrict__ dx, int64_t ld_dx,
                                                            T* __restrict__ dxm, int64_t ld_dxm,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int mask_mode, float p, uint64_t seed_in, uint32_t site,
                                                            const uint64_t* __restrict__ seed_ptr, int64_t row_id_stride,
                                                            float* __restrict__ partials) {
  // blockDim = 64 * NW waves (NW chosen by the host so that the [2][NW][D] reduction buffer fits 32 KiB):
  // many waves per workgroup hide the row-after-row load latency, few workgroups keep the same-address
  // atomics of the parameter gradients rare
  extern __shared__ float red_dyn[];
  const int NW = blockDim.x >> 6;
  float* red0 = red_dyn;            // [NW][D]
  float* red1 = red_dyn + NW * D;   // [NW][D]
  const uint64_t seed = seed_in ^ ((p > 0.f && seed_ptr) ? seed_ptr[0] : 0ull);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * NW + wave;
  const int64_t nwaves = (int64_t)gridDim.x * NW;
  const int nvec = D / 4;
  const float inv_keep = dropout_inv_keep(p), inv_d = 1.f / (float)D;
  float dg[NV][4], db[NV][4], gm[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
    if (c < nvec) g4 = *reinterpret_cast<const f32x4*>(gamma + c * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; gm[i][e] = g4[e]; }
  }

  for (int64_t m0 = wave_global; m0 < M; m0 += nwaves * R) {
    // rows may be a strided subset of the forward's rows (e.g. position 0 of every sample): statistics and the
    // dropout counter are indexed by the forward's row id
    float xh[R][NV][4], g[R][NV][4], mean[R], rstd[R], s1[R], s2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t m = m0 + r * nwaves;
      const bool live = m < M;
      mean[r] = live ? mean_in[m * row_id_stride] : 0.f;
      rstd[r] = live ? rstd_in[m * row_id_stride] : 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
#pragma unroll
        for (int e = 0; e < 4; ++e) { xh[r][i][e] = 0.f; g[r][i][e] = 0.f; }
        if (live && c < nvec) {
          load4<T>(x + m * ldx + c * 4, xh[r][i]);   // raw x for now
          load4<T>(dy + m * ldy + c * 4, g[r][i]);   // raw dy for now
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      s1[r] = 0.f; s2[r] = 0.f;
      const bool live = m0 + r * nwaves < M;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (live && lane + i * 64 < nvec) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xhv = (xh[r][i][e] - mean[r]) * rstd[r], dv = g[r][i][e], gv = dv * gm[i][e];
            xh[r][i][e] = xhv;
            g[r][i][e] = gv;
            s1[r] += gv;
            s2[r] += gv * xhv;
            dg[i][e] += dv * xhv;
            db[i][e] += dv;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { s1[r] = wave_sum(s1[r]) * inv_d; s2[r] = wave_sum(s2[r]) * inv_d; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t m = m0 + r * nwaves;
      if (m >= M) continue;
      const int64_t rid = m * row_id_stride;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
          float o[4], om[4];
          float k4[4] = {1.f, 1.f, 1.f, 1.f};  // D % 4 == 0, so (m*D + c*4) is the first element of one 4-decision word
          if (mask_mode != 0 && p > 0.f) dropout_scale4(dropout_key(seed, site), (uint64_t)(rid * D + c * 4) >> 2, dropout_thr(p), inv_keep, k4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = rstd[r] * (g[r][i][e] - s1[r] - xh[r][i][e] * s2[r]);
            if (mask_mode != 0) {
              const float k = k4[e];
              if (mask_mode == 1) om[e] = o[e] * k; else o[e] = o[e] * (1.f + k);
            }
          }
          store4<T>(dx + m * ld_dx + c * 4, o);
          if (mask_mode == 1) store4<T>(dxm + m * ld_dxm + c * 4, om);
        }
      }
    }
  }
  // cross-wave reduction of the parameter gradients, then one atomic per column per workgroup
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red0[wave * D + c * 4 + e] = dg[i][e];
        red1[wave * D + c * 4 + e] = db[i][e];
      }
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < NW; ++w) { a += red0[w * D + d]; b += red1[w * D + d]; }
    if (partials) {  // summed later by partial_sums_kernel: no same-address atomic chains
      partials[(int64_t)blockIdx.x * 2 * D + d] = a;
      partials[(int64_t)blockIdx.x * 2 * D + D + d] = b;
    } else {
      atomicAdd(dgamma + d, a);
      atomicAdd(dbeta + d, b);
    }
  }
}

// NV by row width, R by how many rows each wave gets
template <typename F>
static int ln_dispatch_nv(int64_t D, F&& f) {
  if (D <= 256) return f(std::integral_constant<int, 1>());
  if (D <= 512) return f(std::integral_constant<int, 2>());
  return f(std::integral_constant<int, 4>());
}

}  // namespace mst

using namespace mst;

// Workgroups (and waves per workgroup) of the backward launch. One workgroup per CU at most: every workgroup ends with
// one column-sum row (an atomic per column on the SAME 2*D addresses, or one row of `partials`).
static unsigned ln_bwd_grid(int64_t M, int64_t D, int& nw) {
  nw = (int)(32768 / (8 * D));  // reduction buffer 2 * nw * D floats <= 32 KiB
  if (nw > 16) nw = 16;
  if (nw < 1) nw = 1;
  // ~4 rows per wave when there are enough rows to fill the chip, one row per wave for small M (the top encoder
  // layer's B rows used to run on ONE workgroup: 11 us)
  int64_t wgs = cdiv(M, 4 * nw);
  if (wgs < 64) wgs = cdiv(M, nw) < 64 ? cdiv(M, nw) : 64;
  return (unsigned)(wgs < 256 ? wgs : 256);
}

extern "C" int64_t mst_layernorm_bwd_parts(int64_t M, int64_t D) {
  if (M <= 0 || D <= 0) return 0;
  int nw;
  return (int64_t)ln_bwd_grid(M, D, nw);
}

static int ln_check(int64_t M, int64_t D, int64_t ldx, int64_t ldy) {
  MST_CHECK_ARG(M > 0 && D > 0, "layernorm: M and D must be positive");
  MST_CHECK_ARG(D % 4 == 0 && D <= 4 * 64 * LN_MAXV, "layernorm: D must be a multiple of 4 and <= %d (got %lld)",
                4 * 64 * LN_MAXV, (long long)D);
  MST_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= D && ldy >= D, "layernorm: leading dims must be multiples of 4 and >= D");
  return MST_OK;
}

extern "C" int mst_layernorm_fwd(int dtype, int64_t M, int64_t D, const void* x, int64_t ldx, const float* gamma,
                                 const float* beta, float eps, void* y, int64_t ldy, float* mean, float* rstd,
                                 int64_t row_id_stride, mst_stream_t stream) {
  int rc = ln_check(M, D, ldx, ldy);
  if (rc) return rc;
  MST_CHECK_ARG(x && gamma && beta && y && mean && rstd, "mst_layernorm_fwd: null pointer");
  const unsigned grid = (unsigned)(cdiv(M, 4) < 2048 ? cdiv(M, 4) : 2048);
  const bool multi = M > (int64_t)grid * 4;  // more than one row per wave: work on two at a time
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    return ln_dispatch_nv(D, [&](auto nv) -> int {
      constexpr int NV = decltype(nv)::value;
      if (multi)
        hipLaunchKernelGGL((layernorm_fwd_kernel<T, NV, 2>), dim3(grid), dim3(256), 0, (hipStream_t)stream, M, (int)D,
                           (const T*)x, ldx, gamma, beta, eps, (T*)y, ldy, mean, rstd, row_id_stride > 0 ? row_id_stride : 1);
      else
        hipLaunchKernelGGL((layernorm_fwd_kernel<T, NV, 1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, M, (int)D,
                           (const T*)x, ldx, gamma, beta, eps, (T*)y, ldy, mean, rstd, row_id_stride > 0 ? row_id_stride : 1);
      MST_CHECK_LAUNCH("layernorm_fwd_kernel");
      return MST_OK;
    });
  });
}

extern "C" int mst_layernorm_bwd(int dtype, int64_t M, int64_t D, const void* x, int64_t ldx, const float* gamma,
                                 const float* mean, const float* rstd, const void* dy, int64_t ldy, void* dx,
                                 int64_t ld_dx, void* dx_masked, int64_t ld_dxm, float* dgamma, float* dbeta,
                                 int mask_mode, float dropout_p, uint64_t dropout_seed, uint32_t dropout_site,
                                 const uint64_t* dropout_seed_ptr, int64_t row_id_stride, float* partials,
                                 mst_stream_t stream) {
  int rc = ln_check(M, D, ldx, ldy);
  if (rc) return rc;
  MST_CHECK_ARG(x && gamma && mean && rstd && dy && dx && (partials || (dgamma && dbeta)), "mst_layernorm_bwd: null pointer");
  MST_CHECK_ARG(ld_dx % 4 == 0 && ld_dx >= D, "mst_layernorm_bwd: bad ld_dx");
  MST_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "mst_layernorm_bwd: mask_mode must be 0,1,2");
  MST_CHECK_ARG(mask_mode != 1 || (dx_masked && ld_dxm % 4 == 0 && ld_dxm >= D), "mst_layernorm_bwd: mask_mode 1 needs dx_masked");
  MST_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "mst_layernorm_bwd: dropout_p must be in [0,1)");
  MST_CHECK_ARG(!partials || (uintptr_t)partials % 16 == 0, "mst_layernorm_bwd: partials must be 16-byte aligned");
  int nw;
  const unsigned grid = ln_bwd_grid(M, D, nw);
  const size_t lds = (size_t)2 * nw * D * sizeof(float);
  const int64_t rows_per_wave = cdiv(M, (int64_t)grid * nw);
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    return ln_dispatch_nv(D, [&](auto nv) -> int {
      constexpr int NV = decltype(nv)::value;
      constexpr int RB = NV == 1 ? 2 : 1;  // rows in flight per wave, bounded by the 128 registers of a 1024-thread block
      if (rows_per_wave > 1 && RB > 1)
        hipLaunchKernelGGL((layernorm_bwd_kernel<T, NV, RB>), dim3(grid), dim3(64 * nw), lds, (hipStream_t)stream, M, (int)D,
                           (const T*)x, ldx, gamma, mean, rstd, (const T*)dy, ldy, (T*)dx, ld_dx, (T*)dx_masked, ld_dxm,
                           dgamma, dbeta, mask_mode, dropout_p, dropout_seed, dropout_site, dropout_seed_ptr,
                           row_id_stride > 0 ? row_id_stride : 1, partials);
      else
        hipLaunchKernelGGL((layernorm_bwd_kernel<T, NV, 1>), dim3(grid), dim3(64 * nw), lds, (hipStream_t)stream, M, (int)D,
                           (const T*)x, ldx, gamma, mean, rstd, (const T*)dy, ldy, (T*)dx, ld_dx, (T*)dx_masked, ld_dxm,
                           dgamma, dbeta, mask_mode, dropout_p, dropout_seed, dropout_site, dropout_seed_ptr,
                           row_id_stride > 0 ? row_id_stride : 1, partials);
      MST_CHECK_LAUNCH("layernorm_bwd_kernel");
      return MST_OK;
    });
  });
}

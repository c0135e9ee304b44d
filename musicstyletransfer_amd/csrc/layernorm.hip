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
#include "common.hpp"

namespace mst {

constexpr int LN_MAXV = 4;  // 4 elements * 64 lanes * LN_MAXV = D up to 1024

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

template <typename T>
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
  for (int64_t m = wave_global; m < M; m += nwaves) {
    float v[LN_MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int c = lane + i * 64;
      if (c < nvec) {
        load4<T>(x + m * ldx + c * 4, v[i]);
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int c = lane + i * 64;
      if (c < nvec) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { float d = v[i][e] - mean; ss += d * d; }
      }
    }
    const float var = wave_sum(ss) / (float)D;
    const float rstd = 1.f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int c = lane + i * 64;
      if (c < nvec) {
        f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4);
        f32x4 b = *reinterpret_cast<const f32x4*>(beta + c * 4);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        store4<T>(y + m * ldy + c * 4, o);
      }
    }
    if (lane == 0) { mean_out[m * row_id_stride] = mean; rstd_out[m * row_id_stride] = rstd; }
  }
}

// mask_mode: 0 = dx only; 1 = dx and dxm = dx * keep/(1-p); 2 = dx <- dx * (1 + keep/(1-p))
template <typename T>
__global__ __launch_bounds__(1024) void layernorm_bwd_kernel(int64_t M, int D, const T* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in,
                                                            const T* __restrict__ dy, int64_t ldy,
                                                            T* __restrict__ dx, int64_t ld_dx,
                                                            T* __restrict__ dxm, int64_t ld_dxm,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int mask_mode, float p, uint64_t seed_in, uint32_t site,
                                                            const uint64_t* __restrict__ seed_ptr, int64_t row_id_stride) {
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
  const float inv_keep = dropout_inv_keep(p);
  float dg[LN_MAXV][4], db[LN_MAXV][4];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; }

  for (int64_t m = wave_global; m < M; m += nwaves) {
    // rows may be a strided subset of the forward's rows (e.g. position 0 of every sample): statistics and the
    // dropout counter are indexed by the forward's row id
    const int64_t rid = m * row_id_stride;
    const float mean = mean_in[rid], rstd = rstd_in[rid];
    float xh[LN_MAXV][4], g[LN_MAXV][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int c = lane + i * 64;
      if (c < nvec) {
        float xv[4], dv[4];
        load4<T>(x + m * ldx + c * 4, xv);
        load4<T>(dy + m * ldy + c * 4, dv);
        f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[i][e] = (xv[e] - mean) * rstd;
          g[i][e] = dv[e] * gm[e];
          s1 += g[i][e];
          s2 += g[i][e] * xh[i][e];
          dg[i][e] += dv[e] * xh[i][e];
          db[i][e] += dv[e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int c = lane + i * 64;
      if (c < nvec) {
        float o[4], om[4];
        uint32_t keep4 = 0xFu;  // D % 4 == 0, so (m*D + c*4) is the first element of one 4-decision word
        if (mask_mode != 0 && p > 0.f) keep4 = dropout_keep4(seed, site, (uint64_t)(rid * D + c * 4) >> 2, p);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = rstd * (g[i][e] - s1 - xh[i][e] * s2);
          if (mask_mode != 0) {
            float k = (p > 0.f) ? (((keep4 >> e) & 1u) ? inv_keep : 0.f) : 1.f;
            if (mask_mode == 1) om[e] = o[e] * k; else o[e] = o[e] * (1.f + k);
          }
        }
        store4<T>(dx + m * ld_dx + c * 4, o);
        if (mask_mode == 1) store4<T>(dxm + m * ld_dxm + c * 4, om);
      }
    }
  }
  // cross-wave reduction of the parameter gradients, then one atomic per column per workgroup
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int c = lane + i * 64;
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
    atomicAdd(dgamma + d, a);
    atomicAdd(dbeta + d, b);
  }
}

}  // namespace mst

using namespace mst;

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
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    hipLaunchKernelGGL((layernorm_fwd_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, M, (int)D,
                       (const T*)x, ldx, gamma, beta, eps, (T*)y, ldy, mean, rstd, row_id_stride > 0 ? row_id_stride : 1);
    MST_CHECK_LAUNCH("layernorm_fwd_kernel");
    return MST_OK;
  });
}

extern "C" int mst_layernorm_bwd(int dtype, int64_t M, int64_t D, const void* x, int64_t ldx, const float* gamma,
                                 const float* mean, const float* rstd, const void* dy, int64_t ldy, void* dx,
                                 int64_t ld_dx, void* dx_masked, int64_t ld_dxm, float* dgamma, float* dbeta,
                                 int mask_mode, float dropout_p, uint64_t dropout_seed, uint32_t dropout_site,
                                 const uint64_t* dropout_seed_ptr, int64_t row_id_stride, mst_stream_t stream) {
  int rc = ln_check(M, D, ldx, ldy);
  if (rc) return rc;
  MST_CHECK_ARG(x && gamma && mean && rstd && dy && dx && dgamma && dbeta, "mst_layernorm_bwd: null pointer");
  MST_CHECK_ARG(ld_dx % 4 == 0 && ld_dx >= D, "mst_layernorm_bwd: bad ld_dx");
  MST_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "mst_layernorm_bwd: mask_mode must be 0,1,2");
  MST_CHECK_ARG(mask_mode != 1 || (dx_masked && ld_dxm % 4 == 0 && ld_dxm >= D), "mst_layernorm_bwd: mask_mode 1 needs dx_masked");
  MST_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "mst_layernorm_bwd: dropout_p must be in [0,1)");
  // one workgroup per CU: every workgroup ends with one atomic per column on the SAME 2*D addresses, so the
  // count of workgroups (not rows) sets the contention
  int nw = (int)(32768 / (8 * D));  // reduction buffer 2 * nw * D floats <= 32 KiB
  if (nw > 16) nw = 16;
  if (nw < 1) nw = 1;
  const unsigned grid = (unsigned)(cdiv(M, 4 * nw) < 256 ? cdiv(M, 4 * nw) : 256);
  const size_t lds = (size_t)2 * nw * D * sizeof(float);
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    hipLaunchKernelGGL((layernorm_bwd_kernel<T>), dim3(grid), dim3(64 * nw), lds, (hipStream_t)stream, M, (int)D,
                       (const T*)x, ldx, gamma, mean, rstd, (const T*)dy, ldy, (T*)dx, ld_dx, (T*)dx_masked, ld_dxm,
                       dgamma, dbeta, mask_mode, dropout_p, dropout_seed, dropout_site, dropout_seed_ptr,
                       row_id_stride > 0 ? row_id_stride : 1);
    MST_CHECK_LAUNCH("layernorm_bwd_kernel");
    return MST_OK;
  });
}

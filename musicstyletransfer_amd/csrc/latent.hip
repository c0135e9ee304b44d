// latent.hip — the VAE bottleneck between the two Transformers, fp32 math on B (batch) rows.
//
// Forward (one workgroup per sample):
//   h0 = enc_out[b,0,:]                                        model.py:97
//   [mu | sigma] = h0 · Wl^T + bl                              model.py:100-103 (sigma is a raw linear output)
//   z = mu + eps * sigma                                       model.py:292   (eps injected by the caller)
//   kl[b] = 0.5 * sum(sigma^2 + mu^2 - 1 - log(sigma^2))       loss.py:8-12   (no epsilon inside the log)
//   dec_in[b,0,:] = alpha_d * (z · Wh^T + bh + cls_d[c_b]) + pos_d[0]   model.py:229-232,244; transformer.py:237
// Backward: per-sample vectors in one kernel, then the parameter gradients as batch reductions
// (no atomics: each output element is owned by one thread that loops over the batch) — a second launch (mst_latent_bwd), or,
// in the training step, two mst_outer_job that ride on the weight-gradient reduction pass (mst_latent_bwd_vec + outer_jobs.hpp).
//
// These are B x {De, 2Z, Dd} problems (64 x 256 x 128): far too small for MFMA tiles to matter;
// they are kept in fp32 because the KL term's log(sigma^2) is the most precision-sensitive
// quantity in the ELBO.
#include <math.h>
#include "common.hpp"
#include "outer_jobs.hpp"
#include "latent_fwd.hpp"

namespace mst {

template <typename T, bool PRE>
__global__ __launch_bounds__(LAT_THREADS) void latent_fwd_kernel(LatentFwdArgs a) {
  extern __shared__ float sm[];
  latent_fwd_wg<T, PRE>(a, (int64_t)blockIdx.x, sm);
}

// sum over j = j0, j0 + step, ... < n of v[j] * W[j, col] (v in LDS), UNR weight loads in flight at a time and the next batch
// requested before the current one is consumed; the FMA order of the plain loop. (One load per FMA made the generic path a
// chain of dependent cold round trips: latent_bwd_vec 56 us at configs[2], 128 rows per thread.)
template <int UNR>
__device__ __forceinline__ float strided_col_dot(const float* v, const float* __restrict__ W, int64_t ldw, int col, int j0, int step, int n) {
  float acc = 0.f;
  if (j0 >= n) return acc;
  const int cnt = (n - j0 + step - 1) / step;  // terms of this thread
  const float* wp = W + (int64_t)j0 * ldw + col;
  const int64_t wstep = (int64_t)step * ldw;
  auto load = [&](float (&w)[UNR], int t0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) w[u] = wp[(int64_t)(t0 + u < cnt ? t0 + u : cnt - 1) * wstep];
  };
  auto fold = [&](const float (&w)[UNR], int t0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (t0 + u < cnt) acc = fmaf(v[j0 + (t0 + u) * step], w[u], acc);
  };
  float wa[UNR], wb[UNR];
  load(wa, 0);
  for (int t0 = 0;;) {
    load(wb, t0 + UNR < cnt ? t0 + UNR : t0);
    fold(wa, t0);
    t0 += UNR;
    if (t0 >= cnt) break;
    load(wa, t0 + UNR < cnt ? t0 + UNR : t0);
    fold(wb, t0);
    t0 += UNR;
    if (t0 >= cnt) break;
  }
  return acc;
}

// per-sample backward vectors: t = alpha_d * g0, dz, dlat = [dmu | dsigma], dh0.
// Columns of the weight matrices are contracted (consecutive threads read consecutive columns, coalesced); each
// output is split over NP row parts that are combined through LDS, so a thread's dependent FMA chain is
// rows / NP long instead of rows.
constexpr int LAT_PRE_H = 8, LAT_PRE_L = 32;
__host__ __device__ inline bool latent_bwd_pre_shape(int64_t De, int64_t Z, int64_t Dd) {
  if (Z > LAT_THREADS || De > LAT_THREADS) return false;
  const int64_t np_h = LAT_THREADS / Z, np_l = LAT_THREADS / De;
  return (Dd + np_h - 1) / np_h <= LAT_PRE_H && (2 * Z + np_l - 1) / np_l <= LAT_PRE_L;
}
// the same for FOUR consecutive columns (one 16-byte load per row): four times fewer load instructions and four times more row parts
// per output (latent_bwd_vec's general path at configs[2]: 512 rows x 256 columns — 32 rows per thread in four batches instead of 128 in eight)
template <int UNR>
__device__ __forceinline__ f32x4 strided_col_dot4(const float* v, const float* __restrict__ W, int64_t ldw, int col, int j0, int step, int n) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (j0 >= n) return acc;
  const int cnt = (n - j0 + step - 1) / step;
  const float* wp = W + (int64_t)j0 * ldw + col;
  const int64_t wstep = (int64_t)step * ldw;
  auto load = [&](f32x4 (&w)[UNR], int t0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) w[u] = *reinterpret_cast<const f32x4*>(wp + (int64_t)(t0 + u < cnt ? t0 + u : cnt - 1) * wstep);
  };
  auto fold = [&](const f32x4 (&w)[UNR], int t0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      if (t0 + u < cnt) {
        const float x = v[j0 + (t0 + u) * step];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(x, w[u][e], acc[e]);
      }
  };
  f32x4 wa[UNR], wb[UNR];
  load(wa, 0);
  for (int t0 = 0;;) {
    load(wb, t0 + UNR < cnt ? t0 + UNR : t0);
    fold(wa, t0);
    t0 += UNR;
    if (t0 >= cnt) break;
    load(wa, t0 + UNR < cnt ? t0 + UNR : t0);
    fold(wb, t0);
    t0 += UNR;
    if (t0 >= cnt) break;
  }
  return acc;
}

// out[j] = sum_k x[k] * W[j, k] for 16-bit rows of W with n_in = 64 * VEC (VEC even: 4-byte pieces), U outputs per wave and pass with
// all U * VEC / 2 loads of a lane in flight before the first is consumed
template <typename T, int U, int VEC, typename F>
__device__ __forceinline__ void wave_dots_row(const float* x, const T* __restrict__ W, int64_t ldw, int n_out, int wave, int n_waves, int lane,
                                              F&& emit) {
  static_assert(VEC % 2 == 0, "a lane's piece of a row is whole 4-byte words");
  float xs[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) xs[e] = x[lane * VEC + e];
  for (int j0 = wave * U; j0 < n_out; j0 += n_waves * U) {
    uint32_t w[U][VEC / 2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t* row = reinterpret_cast<const uint32_t*>(W + (int64_t)(j0 + u < n_out ? j0 + u : n_out - 1) * ldw + lane * VEC);
#pragma unroll
      for (int e = 0; e < VEC / 2; ++e) w[u][e] = row[e];
    }
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[u] = 0.f;
#pragma unroll
      for (int e = 0; e < VEC / 2; ++e) {
        acc[u] = fmaf(xs[2 * e], bits_to_f32<T>((uint16_t)(w[u][e] & 0xffffu)), acc[u]);
        acc[u] = fmaf(xs[2 * e + 1], bits_to_f32<T>((uint16_t)(w[u][e] >> 16)), acc[u]);
      }
    }
    reduce_emit<U>(acc, j0, n_out, lane, emit);
  }
}

struct LatentDx0 {  // (optional) what latent_bwd_vec_kernel computes d(dec_in[b, 0, :]) from instead of reading it
  const void* dq; int64_t dq_stride; const void* Wq; int64_t ld_wq; int nq; const void* resid; int64_t resid_stride;
};

template <typename T, bool PRE>
__global__ __launch_bounds__(LAT_THREADS) void latent_bwd_vec_kernel(int De, int Z, int Dd, const float* __restrict__ Wl,
                                                                     const float* __restrict__ eps,
                                                                     const float* __restrict__ Wh,
                                                                     const float* __restrict__ mu,
                                                                     const float* __restrict__ sigma,
                                                                     const T* __restrict__ d_dec_in, int64_t dec_stride,
                                                                     float alpha_d, float kl_weight, float gscale,
                                                                     float enc_scale, float* __restrict__ tvec,
                                                                     float* __restrict__ dlat, T* __restrict__ d_enc_out,
                                                                     int64_t denc_stride, const int32_t* __restrict__ classes,
                                                                     float* __restrict__ dcls, int64_t ld_cls, LatentDx0 x0) {
  extern __shared__ float sm[];
  float* t = sm;             // [Dd]
  float* dl = sm + Dd;       // [2Z]
  float* part = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(dl + 2 * Z) + 15) & ~(uintptr_t)15);  // [LAT_THREADS] partial sums (general form: float4 each)
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  // Fast path (one round per product, few rows per thread): the weight elements a thread will contract, and mu / sigma
  // / eps, are loaded before the first barrier — their addresses depend on nothing computed here, and in the step they
  // are cold lines (three dependent round trips otherwise).
  constexpr int PRE_H = LAT_PRE_H, PRE_L = LAT_PRE_L;
  const int zc = Z < LAT_THREADS ? Z : LAT_THREADS, np_h = LAT_THREADS / zc > 0 ? LAT_THREADS / zc : 1;
  const int dc = De < LAT_THREADS ? De : LAT_THREADS, np_l = LAT_THREADS / dc > 0 ? LAT_THREADS / dc : 1;
  constexpr bool pre = PRE;  // host: latent_bwd_pre_shape
  // (the computed form of d(dec_in) runs BEFORE the weight preloads below: with both live the kernel spilled 15 registers, and a
  // dispatch that needs scratch memory costs ~15 us more on this stack)
  if (x0.dq) {
    // d(dec_in[b, 0, :]) is not read but COMPUTED: dq[b, 0, :] (the gradient of the decoder's first K | Q | V projection at position 0,
    // nq values) against Wt [Dd, nq] (the TRANSPOSED 16-bit weight: row j holds what output j contracts, contiguous) plus the
    // residual branch's row, rounded to the activation type as the GEMM launch this replaces rounds it (mst_latent_bwd_vec_proj:
    // that launch's other B T rows ride on the backward tail). A wave per 8 outputs, a lane owns nq / 64 consecutive inputs
    // (12-byte pieces of a row at the decoder's width 128), every weight load of the pass in flight before the first is used.
    const T* __restrict__ dq = reinterpret_cast<const T*>(x0.dq) + b * x0.dq_stride;
    for (int k = tid; k < x0.nq; k += LAT_THREADS) part[k] = to_f32(dq[k]);  // (`part` is free until the dz product below)
    // (the residual row waits in t[]: a load per OUTPUT inside emit was a chain of dependent round trips on lane 0)
    const T* __restrict__ rs = x0.resid ? reinterpret_cast<const T*>(x0.resid) + b * x0.resid_stride : nullptr;
    for (int j = tid; j < Dd; j += LAT_THREADS) t[j] = rs ? to_f32(rs[j]) : 0.f;
    __syncthreads();
    // (emit only fills t[]: tvec and the class table's atomics go out below as whole-wave instructions — as 128 single-lane
    // atomics per workgroup on the same eight cache lines they serialised in the L2 for 16 us)
    auto emit = [&](int j, float a) { t[j] = alpha_d * to_f32(from_f32<T>(a + t[j])); };
    const T* Wt = reinterpret_cast<const T*>(x0.Wq);
    const int wave = tid >> 6, lane = tid & 63;
    if (x0.nq == 6 * 64) wave_dots_row<T, 8, 6>(part, Wt, x0.ld_wq, Dd, wave, LAT_THREADS / 64, lane, emit);
    else wave_dots_row<T, 8, 12>(part, Wt, x0.ld_wq, Dd, wave, LAT_THREADS / 64, lane, emit);  // (host: nq is 384 or 768)
    __syncthreads();
    for (int j = tid; j < Dd; j += LAT_THREADS) {
      const float v = t[j];
      tvec[b * Dd + j] = v;
      if (dcls) atomicAdd(dcls + (int64_t)classes[b] * ld_cls + j, v);
    }
  }
  float wh[PRE_H], wl[PRE_L], m_r = 0.f, s_r = 0.f, e_r = 0.f;
  if constexpr (pre) {
    const int i = tid % zc, pt = tid / zc;
#pragma unroll
    for (int k = 0; k < PRE_H; ++k) {
      const int j = pt + k * np_h;
      wh[k] = (pt < np_h && j < Dd) ? Wh[(int64_t)j * Z + i] : 0.f;
    }
    const int d = tid % dc, pl = tid / dc;
#pragma unroll
    for (int k = 0; k < PRE_L; ++k) {
      const int j = pl + k * np_l;
      wl[k] = (pl < np_l && j < 2 * Z) ? Wl[(int64_t)j * De + d] : 0.f;
    }
    if (tid < Z) { m_r = mu[b * Z + tid]; s_r = sigma[b * Z + tid]; e_r = eps[b * Z + tid]; }
  }
  if (!x0.dq)
  for (int j = tid; j < Dd; j += LAT_THREADS) {
    const float v = alpha_d * to_f32(d_dec_in[b * dec_stride + j]);
    t[j] = v;
    tvec[b * Dd + j] = v;
    if (dcls) atomicAdd(dcls + (int64_t)classes[b] * ld_cls + j, v);  // (mst_latent_bwd_vec: the class table's gradient)
  }
  __syncthreads();
  // dz[i] = sum_j t[j] * Wh[j,i]: thread (i, part) sums every np-th j
  {
    const int np = np_h;  // parts per output (16 at Z = 64)
    for (int i0 = 0; i0 < Z; i0 += LAT_THREADS) {             // one round unless Z > 1024
      const int i = i0 + tid % zc, pt = tid / zc;
      float acc = 0.f;
      if constexpr (pre) {
#pragma unroll
        for (int k = 0; k < PRE_H; ++k)
          if (pt + k * np < Dd) acc = fmaf(t[pt + k * np], wh[k], acc);
      } else if (i < Z && pt < np) {
        acc = strided_col_dot<16>(t, Wh, Z, i, pt, np, Dd);
      }
      part[tid] = acc;
      __syncthreads();
      if (tid < Z - i0 && tid < LAT_THREADS) {
        float a = 0.f;
        for (int p2 = 0; p2 < np && p2 * zc + tid < LAT_THREADS; ++p2) a += part[p2 * zc + tid];
        const int ii = i0 + tid;
        const float m = pre ? m_r : mu[b * Z + ii], s2 = pre ? s_r : sigma[b * Z + ii], ee = pre ? e_r : eps[b * Z + ii];
        // gscale: loss scale of everything upstream of here (the encoder); enc_scale = gscale / (loss scale the
        // incoming decoder-side gradient carries). Both are 1 unless fp16 loss scaling is on.
        const float dm = kl_weight * gscale * m + enc_scale * a;
        const float ds = kl_weight * gscale * (s2 - 1.f / s2) + enc_scale * ee * a;
        dl[ii] = dm;
        dl[Z + ii] = ds;
        dlat[b * 2 * Z + ii] = dm;
        dlat[b * 2 * Z + Z + ii] = ds;
      }
      __syncthreads();
    }
  }
  // dh0[d] = sum_j dlat[j] * Wl[j,d], same split
  if (!pre && De % 4 == 0 && De / 4 <= LAT_THREADS && ((uintptr_t)Wl & 15) == 0) {
    // general path, four columns per thread: `part` holds LAT_THREADS float4 here (the host sizes it for the general form)
    const int dc4 = De / 4, np4 = LAT_THREADS / dc4;
    const int c4 = tid % dc4, pt = tid / dc4;
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    if (pt < np4) a4 = strided_col_dot4<8>(dl, Wl, De, 4 * c4, pt, np4, 2 * Z);
    reinterpret_cast<f32x4*>(part)[tid] = a4;
    __syncthreads();
    for (int d = tid; d < De; d += LAT_THREADS) {
      float a = 0.f;
      for (int p2 = 0; p2 < np4; ++p2) a += part[(p2 * dc4 + d / 4) * 4 + (d & 3)];
      d_enc_out[b * denc_stride + d] = from_f32<T>(a);
    }
  } else {
    const int np = np_l;  // 4 at De = 256
    for (int d0 = 0; d0 < De; d0 += LAT_THREADS) {
      const int d = d0 + tid % dc, pt = tid / dc;
      float acc = 0.f;
      if constexpr (pre) {
#pragma unroll
        for (int k = 0; k < PRE_L; ++k)
          if (pt + k * np < 2 * Z) acc = fmaf(dl[pt + k * np], wl[k], acc);
      } else if (d < De && pt < np) {
        acc = strided_col_dot<16>(dl, Wl, De, d, pt, np, 2 * Z);
      }
      part[tid] = acc;
      __syncthreads();
      if (tid < dc && d0 + tid < De) {
        float a = 0.f;
        for (int p2 = 0; p2 < np; ++p2) a += part[p2 * dc + tid];
        d_enc_out[b * denc_stride + d0 + tid] = from_f32<T>(a);
      }
      __syncthreads();
    }
  }
}

// (batch_outer: outer_jobs.hpp)
// ONE launch for the three parameter-gradient pieces of the latent block (they were three ~5-14 us launches):
// blocks [0, n_wl): dWl[2Z, De] += dlat^T h0, dbl; blocks [n_wl, n_wl + n_wh): dWh[Dd, Z] += t^T z, dbh;
// the rest: dcls_d[class_b, :] += t[b, :]
template <typename T>
__global__ __launch_bounds__(256) void latent_param_grads_kernel(int64_t B, int De, int Z, int Dd, const float* __restrict__ dlat,
                                                                 const T* __restrict__ enc_out, int64_t enc_stride,
                                                                 const float* __restrict__ tvec, const float* __restrict__ z,
                                                                 const int32_t* __restrict__ classes, float* __restrict__ dWl,
                                                                 float* __restrict__ dbl, float* __restrict__ dWh,
                                                                 float* __restrict__ dbh, float* __restrict__ dcls, int64_t ld_cls,
                                                                 int n_wl, int n_wh) {
  __shared__ float red[8][64];
  const int blk = blockIdx.x;
  if (blk < n_wl) {
    batch_outer<T>(blk, threadIdx.x, B, 2 * Z, De, dlat, enc_out, enc_stride, dWl, dbl, red);
  } else if (blk < n_wl + n_wh) {
    batch_outer<float>(blk - n_wl, threadIdx.x, B, Dd, Z, tvec, z, Z, dWh, dbh, red);
  } else {
    const int64_t idx = (int64_t)(blk - n_wl - n_wh) * 256 + threadIdx.x;
    if (idx < B * Dd) {
      const uint32_t bq = (uint32_t)idx / (uint32_t)Dd, dr = (uint32_t)idx - bq * (uint32_t)Dd;
      atomicAdd(dcls + (int64_t)classes[bq] * ld_cls + dr, tvec[idx]);
    }
  }
}

}  // namespace mst

using namespace mst;

static int latent_fwd_impl(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const void* enc_out,
                           int64_t enc_sample_stride, const float* Wl, const float* bl, const float* eps,
                           const float* Wh, const float* bh, const int32_t* classes, const float* cls_d,
                           int64_t ld_cls, const float* pos_d, float alpha_d, float* mu, float* sigma, float* z,
                           float* kl, void* dec_in, int64_t dec_sample_stride, const void* Wq, int64_t ld_wq, const float* bq, void* qkv0,
                           int64_t qkv_sample_stride, int64_t nq, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && De > 0 && Z > 0 && Dd > 0, "mst_latent_fwd: sizes must be positive");
  MST_CHECK_ARG(enc_out && Wl && bl && eps && Wh && bh && classes && cls_d && pos_d && mu && sigma && z && kl && dec_in,
                "mst_latent_fwd: null pointer");
  MST_CHECK_ARG(!Wq || (qkv0 && nq > 0 && (Dd == 64 || Dd == 128 || Dd == 256) && ld_wq >= Dd && ((uintptr_t)Wq % 8) == 0 && ld_wq % 4 == 0),
                "mst_latent_fwd_proj: the row-0 projection takes a decoder width of 64, 128 or 256 and an 8-byte aligned weight");
  const size_t lds = sizeof(float) * (De + 3 * Z + (Wq ? Dd + nq : 0));
  MST_CHECK_ARG(lds <= 60000, "mst_latent_fwd: De + 3Z too large for one workgroup");
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    LatentFwdArgs la = {};
    la.De = (int)De; la.Z = (int)Z; la.Dd = (int)Dd; la.enc_out = enc_out; la.enc_stride = enc_sample_stride;
    la.Wl = Wl; la.bl = bl; la.eps = eps; la.Wh = Wh; la.bh = bh; la.classes = classes; la.cls_d = cls_d; la.ld_cls = ld_cls;
    la.pos_d = pos_d; la.alpha_d = alpha_d; la.mu = mu; la.sigma = sigma; la.z = z; la.kl = kl; la.dec_in = dec_in;
    la.dec_stride = dec_sample_stride;
    la.Wq = Wq; la.ld_wq = ld_wq; la.bq = bq; la.qkv0 = qkv0; la.qkv_stride = qkv_sample_stride; la.nq = (int)nq;
    if (latent_fwd_pre_shape(De, Z, Dd)) hipLaunchKernelGGL((latent_fwd_kernel<T, true>), dim3((unsigned)B), dim3(LAT_THREADS), lds, (hipStream_t)stream, la);
    else hipLaunchKernelGGL((latent_fwd_kernel<T, false>), dim3((unsigned)B), dim3(LAT_THREADS), lds, (hipStream_t)stream, la);
    MST_CHECK_LAUNCH("latent_fwd_kernel");
    return MST_OK;
  });
}
extern "C" int mst_latent_fwd(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const void* enc_out,
                              int64_t enc_sample_stride, const float* Wl, const float* bl, const float* eps,
                              const float* Wh, const float* bh, const int32_t* classes, const float* cls_d,
                              int64_t ld_cls, const float* pos_d, float alpha_d, float* mu, float* sigma, float* z,
                              float* kl, void* dec_in, int64_t dec_sample_stride, mst_stream_t stream) {
  return latent_fwd_impl(dtype, B, De, Z, Dd, enc_out, enc_sample_stride, Wl, bl, eps, Wh, bh, classes, cls_d, ld_cls, pos_d, alpha_d, mu, sigma, z,
                         kl, dec_in, dec_sample_stride, nullptr, 0, nullptr, nullptr, 0, 0, stream);
}
extern "C" int mst_latent_fwd_proj(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const void* enc_out,
                                   int64_t enc_sample_stride, const float* Wl, const float* bl, const float* eps,
                                   const float* Wh, const float* bh, const int32_t* classes, const float* cls_d,
                                   int64_t ld_cls, const float* pos_d, float alpha_d, float* mu, float* sigma, float* z,
                                   float* kl, void* dec_in, int64_t dec_sample_stride, const void* Wq, int64_t ld_wq, const float* bq,
                                   void* qkv0, int64_t qkv_sample_stride, int64_t nq, mst_stream_t stream) {
  MST_CHECK_ARG(Wq != nullptr, "mst_latent_fwd_proj: null projection weight");
  return latent_fwd_impl(dtype, B, De, Z, Dd, enc_out, enc_sample_stride, Wl, bl, eps, Wh, bh, classes, cls_d, ld_cls, pos_d, alpha_d, mu, sigma, z,
                         kl, dec_in, dec_sample_stride, Wq, ld_wq, bq, qkv0, qkv_sample_stride, nq, stream);
}

extern "C" int mst_latent_bwd(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const void* enc_out,
                              int64_t enc_sample_stride, const float* Wl, const float* eps, const float* Wh,
                              const int32_t* classes, const float* mu, const float* sigma, const float* z,
                              const void* d_dec_in, int64_t dec_sample_stride, float alpha_d, float kl_weight,
                              float gscale, float enc_scale, float* dWl, float* dbl, float* dWh, float* dbh, float* dcls_d,
                              int64_t ld_cls, void* d_enc_out, int64_t denc_sample_stride, float* scratch,
                              mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && De > 0 && Z > 0 && Dd > 0, "mst_latent_bwd: sizes must be positive");
  MST_CHECK_ARG(enc_out && Wl && eps && Wh && classes && mu && sigma && z && d_dec_in && dWl && dbl && dWh && dbh &&
                    dcls_d && d_enc_out && scratch,
                "mst_latent_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  float* tvec = scratch;            // [B, Dd]
  float* dlat = scratch + B * Dd;   // [B, 2Z]
  MST_CHECK_ARG(Z <= LAT_THREADS, "mst_latent_bwd: latent size above %d", LAT_THREADS);
  MST_CHECK_ARG(2 * Z * De < (1ll << 31) && Dd * Z < (1ll << 31) && B * Dd < (1ll << 31), "mst_latent_bwd: sizes above 2^31 elements");
  const size_t lds = sizeof(float) * (Dd + 2 * Z + (latent_bwd_pre_shape(De, Z, Dd) ? 1 : 4) * LAT_THREADS + 4);  // (+ 4: the float4 view's alignment)
  const int n_wl = (int)cdiv(2 * Z * De, 64), n_wh = (int)cdiv(Dd * Z, 64), n_cls = (int)cdiv(B * Dd, 256);
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    if (latent_bwd_pre_shape(De, Z, Dd))
      hipLaunchKernelGGL((latent_bwd_vec_kernel<T, true>), dim3((unsigned)B), dim3(LAT_THREADS), lds, s, (int)De, (int)Z, (int)Dd, Wl,
                         eps, Wh, mu, sigma, (const T*)d_dec_in, dec_sample_stride, alpha_d, kl_weight, gscale, enc_scale, tvec,
                         dlat, (T*)d_enc_out, denc_sample_stride, (const int32_t*)nullptr, (float*)nullptr, (int64_t)0, LatentDx0{});
    else
      hipLaunchKernelGGL((latent_bwd_vec_kernel<T, false>), dim3((unsigned)B), dim3(LAT_THREADS), lds, s, (int)De, (int)Z, (int)Dd, Wl,
                         eps, Wh, mu, sigma, (const T*)d_dec_in, dec_sample_stride, alpha_d, kl_weight, gscale, enc_scale, tvec,
                         dlat, (T*)d_enc_out, denc_sample_stride, (const int32_t*)nullptr, (float*)nullptr, (int64_t)0, LatentDx0{});
    MST_CHECK_LAUNCH("latent_bwd_vec_kernel");
    hipLaunchKernelGGL((latent_param_grads_kernel<T>), dim3((unsigned)(n_wl + n_wh + n_cls)), dim3(256), 0, s, B, (int)De, (int)Z,
                       (int)Dd, dlat, (const T*)enc_out, enc_sample_stride, tvec, z, classes, dWl, dbl, dWh, dbh, dcls_d, ld_cls,
                       n_wl, n_wh);
    MST_CHECK_LAUNCH("latent_param_grads_kernel");
    return MST_OK;
  });
}

static int latent_bwd_vec_impl(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const float* Wl, const float* eps,
                               const float* Wh, const int32_t* classes, const float* mu, const float* sigma,
                               const void* d_dec_in, int64_t dec_sample_stride, float alpha_d, float kl_weight, float gscale,
                               float enc_scale, float* dcls_d, int64_t ld_cls, void* d_enc_out, int64_t denc_sample_stride,
                               float* scratch, const LatentDx0& x0, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && De > 0 && Z > 0 && Dd > 0, "mst_latent_bwd_vec: sizes must be positive");
  MST_CHECK_ARG(Wl && eps && Wh && classes && mu && sigma && (d_dec_in || x0.dq) && dcls_d && d_enc_out && scratch, "mst_latent_bwd_vec: null pointer");
  MST_CHECK_ARG(Z <= LAT_THREADS, "mst_latent_bwd_vec: latent size above %d", LAT_THREADS);
  MST_CHECK_ARG(!x0.dq || (x0.Wq && (x0.nq == 384 || x0.nq == 768) && x0.ld_wq >= x0.nq && x0.ld_wq % 2 == 0 && ((uintptr_t)x0.Wq % 4) == 0),
                "mst_latent_bwd_vec_proj: the projection must have 384 or 768 outputs (decoder width 128 or 256) and a 4-byte aligned transposed weight");
  const size_t lds = sizeof(float) * (Dd + 2 * Z + (latent_bwd_pre_shape(De, Z, Dd) ? 1 : 4) * LAT_THREADS + 4);  // (+ 4: the float4 view's alignment)
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    if (latent_bwd_pre_shape(De, Z, Dd))
      hipLaunchKernelGGL((latent_bwd_vec_kernel<T, true>), dim3((unsigned)B), dim3(LAT_THREADS), lds, (hipStream_t)stream, (int)De, (int)Z,
                         (int)Dd, Wl, eps, Wh, mu, sigma, (const T*)d_dec_in, dec_sample_stride, alpha_d, kl_weight, gscale, enc_scale,
                         scratch, scratch + B * Dd, (T*)d_enc_out, denc_sample_stride, classes, dcls_d, ld_cls, x0);
    else
      hipLaunchKernelGGL((latent_bwd_vec_kernel<T, false>), dim3((unsigned)B), dim3(LAT_THREADS), lds, (hipStream_t)stream, (int)De, (int)Z,
                         (int)Dd, Wl, eps, Wh, mu, sigma, (const T*)d_dec_in, dec_sample_stride, alpha_d, kl_weight, gscale, enc_scale,
                         scratch, scratch + B * Dd, (T*)d_enc_out, denc_sample_stride, classes, dcls_d, ld_cls, x0);
    MST_CHECK_LAUNCH("latent_bwd_vec_kernel");
    return MST_OK;
  });
}
extern "C" int mst_latent_bwd_vec(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const float* Wl, const float* eps,
                                  const float* Wh, const int32_t* classes, const float* mu, const float* sigma,
                                  const void* d_dec_in, int64_t dec_sample_stride, float alpha_d, float kl_weight, float gscale,
                                  float enc_scale, float* dcls_d, int64_t ld_cls, void* d_enc_out, int64_t denc_sample_stride,
                                  float* scratch, mst_stream_t stream) {
  MST_CHECK_ARG(d_dec_in != nullptr, "mst_latent_bwd_vec: null pointer");
  return latent_bwd_vec_impl(dtype, B, De, Z, Dd, Wl, eps, Wh, classes, mu, sigma, d_dec_in, dec_sample_stride, alpha_d, kl_weight, gscale, enc_scale,
                             dcls_d, ld_cls, d_enc_out, denc_sample_stride, scratch, LatentDx0{}, stream);
}
extern "C" int mst_latent_bwd_vec_proj(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const float* Wl, const float* eps,
                                       const float* Wh, const int32_t* classes, const float* mu, const float* sigma,
                                       const void* dq0, int64_t dq_sample_stride, const void* Wq, int64_t ld_wq, int64_t nq,
                                       const void* resid0, int64_t resid_sample_stride, float alpha_d, float kl_weight, float gscale,
                                       float enc_scale, float* dcls_d, int64_t ld_cls, void* d_enc_out, int64_t denc_sample_stride,
                                       float* scratch, mst_stream_t stream) {
  MST_CHECK_ARG(dq0 != nullptr && Wq != nullptr, "mst_latent_bwd_vec_proj: null pointer");
  LatentDx0 x0 = {dq0, dq_sample_stride, Wq, ld_wq, (int)nq, resid0, resid_sample_stride};
  return latent_bwd_vec_impl(dtype, B, De, Z, Dd, Wl, eps, Wh, classes, mu, sigma, nullptr, 0, alpha_d, kl_weight, gscale, enc_scale, dcls_d, ld_cls,
                             d_enc_out, denc_sample_stride, scratch, x0, stream);
}

namespace mst {
__global__ __launch_bounds__(256) void outer_jobs_kernel(OuterBatch b) {
  __shared__ float red[8][64];
  outer_jobs_wg(b, (int)blockIdx.x, red);
}
}  // namespace mst

extern "C" int mst_outer_jobs(const mst_outer_job* jobs, int n, mst_stream_t stream) {
  OuterBatch b;
  int rc = pack_outer_jobs(jobs, n, b);
  if (rc) return rc;
  if (b.wg_prefix[b.n] == 0) return MST_OK;
  hipLaunchKernelGGL(outer_jobs_kernel, dim3((unsigned)b.wg_prefix[b.n]), dim3(256), 0, (hipStream_t)stream, b);
  MST_CHECK_LAUNCH("outer_jobs_kernel");
  return MST_OK;
}

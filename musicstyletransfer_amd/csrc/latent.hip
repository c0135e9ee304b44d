// latent.hip — the VAE bottleneck between the two Transformers, fp32 math on B (batch) rows.
//
// Forward (one workgroup per sample):
//   h0 = enc_out[b,0,:]                                        model.py:97
//   [mu | sigma] = h0 · Wl^T + bl                              model.py:100-103 (sigma is a raw linear output)
//   z = mu + eps * sigma                                       model.py:292   (eps injected by the caller)
//   kl[b] = 0.5 * sum(sigma^2 + mu^2 - 1 - log(sigma^2))       loss.py:8-12   (no epsilon inside the log)
//   dec_in[b,0,:] = alpha_d * (z · Wh^T + bh + cls_d[c_b]) + pos_d[0]   model.py:229-232,244; transformer.py:237
// Backward: per-sample vectors in one kernel, then the parameter gradients as batch reductions
// (no atomics: each output element is owned by one thread that loops over the batch).
//
// These are B x {De, 2Z, Dd} problems (64 x 256 x 128): far too small for MFMA tiles to matter;
// they are kept in fp32 because the KL term's log(sigma^2) is the most precision-sensitive
// quantity in the ELBO.
#include <math.h>
#include "common.hpp"

namespace mst {

constexpr int LAT_THREADS = 1024;  // 16 waves: these kernels are B workgroups of dependent dot products (latency-bound)

template <typename T>
__global__ __launch_bounds__(LAT_THREADS) void latent_fwd_kernel(int De, int Z, int Dd, const T* __restrict__ enc_out,
                                                         int64_t enc_stride, const float* __restrict__ Wl,
                                                         const float* __restrict__ bl, const float* __restrict__ eps,
                                                         const float* __restrict__ Wh, const float* __restrict__ bh,
                                                         const int32_t* __restrict__ classes,
                                                         const float* __restrict__ cls_d, int64_t ld_cls,
                                                         const float* __restrict__ pos_d, float alpha_d,
                                                         float* __restrict__ mu, float* __restrict__ sigma,
                                                         float* __restrict__ z, float* __restrict__ kl,
                                                         T* __restrict__ dec_in, int64_t dec_stride) {
  extern __shared__ float sm[];
  float* h0 = sm;            // [De]
  float* lat = sm + De;      // [2Z]
  float* zs = lat + 2 * Z;   // [Z]
  __shared__ float klred[LAT_THREADS / 64];
  constexpr int NW = LAT_THREADS / 64;
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int d = tid; d < De; d += LAT_THREADS) h0[d] = to_f32(enc_out[b * enc_stride + d]);
  __syncthreads();
  // one wave per output, lanes across the contraction (coalesced weight rows)
  for (int j = wave; j < 2 * Z; j += NW) {
    float acc = 0.f;
    for (int d = lane; d < De; d += 64) acc += h0[d] * Wl[(int64_t)j * De + d];
    acc = wave_sum(acc);
    if (lane == 0) lat[j] = acc + bl[j];
  }
  __syncthreads();
  float klacc = 0.f;
  for (int i = tid; i < Z; i += LAT_THREADS) {
    const float m = lat[i], s = lat[Z + i];
    const float zz = m + eps[b * Z + i] * s;
    mu[b * Z + i] = m;
    sigma[b * Z + i] = s;
    z[b * Z + i] = zz;
    zs[i] = zz;
    const float s2 = s * s;
    klacc += 0.5f * (s2 + m * m - 1.f - logf(s2));
  }
  klacc = wave_sum(klacc);
  if (lane == 0) klred[wave] = klacc;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < NW; ++w) t += klred[w];
    kl[b] = t;
  }
  const int c = classes[b];
  for (int j = wave; j < Dd; j += NW) {
    float acc = 0.f;
    for (int i = lane; i < Z; i += 64) acc += zs[i] * Wh[(int64_t)j * Z + i];
    acc = wave_sum(acc);
    if (lane == 0) {
      const float v = alpha_d * (acc + bh[j] + cls_d[(int64_t)c * ld_cls + j]) + pos_d[j];
      dec_in[b * dec_stride + j] = from_f32<T>(v);
    }
  }
}

// per-sample backward vectors: t = alpha_d * g0, dz, dlat = [dmu | dsigma], dh0
template <typename T>
__global__ __launch_bounds__(256) void latent_bwd_vec_kernel(int De, int Z, int Dd, const float* __restrict__ Wl,
                                                             const float* __restrict__ eps,
                                                             const float* __restrict__ Wh,
                                                             const float* __restrict__ mu,
                                                             const float* __restrict__ sigma,
                                                             const T* __restrict__ d_dec_in, int64_t dec_stride,
                                                             float alpha_d, float kl_weight, float gscale,
                                                             float enc_scale, float* __restrict__ tvec, float* __restrict__ dlat,
                                                             T* __restrict__ d_enc_out, int64_t denc_stride) {
  extern __shared__ float sm[];
  float* t = sm;            // [Dd]
  float* dl = sm + Dd;      // [2Z]
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  for (int j = tid; j < Dd; j += 256) {
    const float v = alpha_d * to_f32(d_dec_in[b * dec_stride + j]);
    t[j] = v;
    tvec[b * Dd + j] = v;
  }
  __syncthreads();
  // dz[i] = sum_j t[j] * Wh[j,i]: thread (i, part) sums a quarter of j (consecutive threads read consecutive
  // Wh columns), the parts are combined through LDS
  float* part = dl + 2 * Z;  // [4][max(Z, De)] scratch
  const int nparts = 4;
  for (int w = tid; w < Z * nparts; w += 256) {
    const int i = w % Z, pt = w / Z;
    float acc = 0.f;
    for (int j = pt; j < Dd; j += nparts) acc += t[j] * Wh[(int64_t)j * Z + i];
    part[pt * Z + i] = acc;
  }
  __syncthreads();
  for (int i = tid; i < Z; i += 256) {
    const float acc = (part[i] + part[Z + i]) + (part[2 * Z + i] + part[3 * Z + i]);
    const float m = mu[b * Z + i], s = sigma[b * Z + i];
    // gscale: loss scale of everything upstream of here (the encoder); enc_scale = gscale / (loss scale the
    // incoming decoder-side gradient carries). Both are 1 unless fp16 loss scaling is on.
    const float dm = kl_weight * gscale * m + enc_scale * acc;
    const float ds = kl_weight * gscale * (s - 1.f / s) + enc_scale * eps[b * Z + i] * acc;
    dl[i] = dm;
    dl[Z + i] = ds;
    dlat[b * 2 * Z + i] = dm;
    dlat[b * 2 * Z + Z + i] = ds;
  }
  __syncthreads();
  // dh0[d] = sum_j dlat[j] * Wl[j,d]
  for (int d = tid; d < De; d += 256) {
    float acc = 0.f;
    for (int j = 0; j < 2 * Z; ++j) acc += dl[j] * Wl[(int64_t)j * De + d];
    d_enc_out[b * denc_stride + d] = from_f32<T>(acc);
  }
}

// parameter gradients: out[j, i] += sum_b L[b, j] * R[b, i]; obias[j] += sum_b L[b, j]
template <typename RT>
__global__ __launch_bounds__(256) void batch_outer_kernel(int64_t B, int J, int I, const float* __restrict__ L,
                                                          const RT* __restrict__ R, int64_t r_stride,
                                                          float* __restrict__ out, float* __restrict__ obias) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < (int64_t)J * I) {
    const int j = (int)(idx / I), i = (int)(idx % I);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // independent chains: the loop is load-latency bound
    int64_t b = 0;
    for (; b + 4 <= B; b += 4) {
      a0 += L[(b + 0) * J + j] * to_f32(R[(b + 0) * r_stride + i]);
      a1 += L[(b + 1) * J + j] * to_f32(R[(b + 1) * r_stride + i]);
      a2 += L[(b + 2) * J + j] * to_f32(R[(b + 2) * r_stride + i]);
      a3 += L[(b + 3) * J + j] * to_f32(R[(b + 3) * r_stride + i]);
    }
    for (; b < B; ++b) a0 += L[b * J + j] * to_f32(R[b * r_stride + i]);
    out[idx] += (a0 + a1) + (a2 + a3);
  }
  if (obias && idx < J) {
    float acc = 0.f;
    for (int64_t b = 0; b < B; ++b) acc += L[b * J + idx];
    obias[idx] += acc;
  }
}

__global__ __launch_bounds__(256) void class_scatter_kernel(int64_t B, int J, const float* __restrict__ L,
                                                            const int32_t* __restrict__ classes,
                                                            float* __restrict__ dcls, int64_t ld) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < B * J) {
    const int64_t b = idx / J;
    const int j = (int)(idx % J);
    atomicAdd(dcls + (int64_t)classes[b] * ld + j, L[idx]);
  }
}

}  // namespace mst

using namespace mst;

extern "C" int mst_latent_fwd(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const void* enc_out,
                              int64_t enc_sample_stride, const float* Wl, const float* bl, const float* eps,
                              const float* Wh, const float* bh, const int32_t* classes, const float* cls_d,
                              int64_t ld_cls, const float* pos_d, float alpha_d, float* mu, float* sigma, float* z,
                              float* kl, void* dec_in, int64_t dec_sample_stride, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && De > 0 && Z > 0 && Dd > 0, "mst_latent_fwd: sizes must be positive");
  MST_CHECK_ARG(enc_out && Wl && bl && eps && Wh && bh && classes && cls_d && pos_d && mu && sigma && z && kl && dec_in,
                "mst_latent_fwd: null pointer");
  const size_t lds = sizeof(float) * (De + 3 * Z);
  MST_CHECK_ARG(lds <= 60000, "mst_latent_fwd: De + 3Z too large for one workgroup");
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    hipLaunchKernelGGL((latent_fwd_kernel<T>), dim3((unsigned)B), dim3(LAT_THREADS), lds, (hipStream_t)stream, (int)De, (int)Z,
                       (int)Dd, (const T*)enc_out, enc_sample_stride, Wl, bl, eps, Wh, bh, classes, cls_d, ld_cls, pos_d,
                       alpha_d, mu, sigma, z, kl, (T*)dec_in, dec_sample_stride);
    MST_CHECK_LAUNCH("latent_fwd_kernel");
    return MST_OK;
  });
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
  const size_t lds = sizeof(float) * (Dd + 2 * Z + 4 * Z);
  int rc = dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    hipLaunchKernelGGL((latent_bwd_vec_kernel<T>), dim3((unsigned)B), dim3(256), lds, s, (int)De, (int)Z, (int)Dd, Wl, eps,
                       Wh, mu, sigma, (const T*)d_dec_in, dec_sample_stride, alpha_d, kl_weight, gscale, enc_scale, tvec, dlat,
                       (T*)d_enc_out, denc_sample_stride);
    MST_CHECK_LAUNCH("latent_bwd_vec_kernel");
    // dWl[2Z, De] += dlat^T · h0 ; dbl += sum_b dlat
    hipLaunchKernelGGL((batch_outer_kernel<T>), dim3((unsigned)cdiv(2 * Z * De, 256)), dim3(256), 0, s, B, (int)(2 * Z),
                       (int)De, dlat, (const T*)enc_out, enc_sample_stride, dWl, dbl);
    MST_CHECK_LAUNCH("batch_outer_kernel(Wl)");
    return MST_OK;
  });
  if (rc) return rc;
  // dWh[Dd, Z] += t^T · z ; dbh += sum_b t
  hipLaunchKernelGGL((batch_outer_kernel<float>), dim3((unsigned)cdiv(Dd * Z, 256)), dim3(256), 0, s, B, (int)Dd, (int)Z, tvec,
                     z, Z, dWh, dbh);
  MST_CHECK_LAUNCH("batch_outer_kernel(Wh)");
  hipLaunchKernelGGL(class_scatter_kernel, dim3((unsigned)cdiv(B * Dd, 256)), dim3(256), 0, s, B, (int)Dd, tvec, classes,
                     dcls_d, ld_cls);
  MST_CHECK_LAUNCH("class_scatter_kernel");
  return MST_OK;
}

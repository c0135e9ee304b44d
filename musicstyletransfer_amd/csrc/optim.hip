// optim.hip — multi-tensor Adam over one flat fp32 bucket + 16-bit shadow maintenance.
//
// Replaces gluon.Trainer.step(batch_size) → mxnet.optimizer.Adam → adam_update, called once per
// parameter tensor in the reference (58 launches; trainer.py:94-101,177), by ONE launch over the
// flat parameter / gradient / moment buffers (the same flat gradient buffer RCCL all-reduces):
//   g   = clip(grad * rescale + wd * w, ±clip)            (clip < 0: no clipping)
//   m   = b1*m + (1-b1)*g ;  v = b2*v + (1-b2)*g*g
//   w  -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
// HBM-bound: 16 B/param read (w,g,m,v) + 12 B/param written (w,m,v) + 2 B/param 16-bit shadow.
// The step counter lives on the device and is advanced by a 1-thread kernel in the same stream,
// so a captured graph replays with the right bias correction.
#include <math.h>
#include "common.hpp"
#include "loss_combine.hpp"
#include "shadows.hpp"

namespace mst {

// state[0] = step count t (int32), state[1] = bits of lr_t (float)
__global__ void adam_tick_kernel(int32_t* state, double lr, double beta1, double beta2) {
  const int t = state[0] + 1;
  state[0] = t;
  const double c1 = 1.0 - pow(beta1, (double)t);
  const double c2 = 1.0 - pow(beta2, (double)t);
  reinterpret_cast<float*>(state)[1] = (float)(lr * sqrt(c2) / c1);
}

// transposed shadows kept current by the optimizer itself (mst_adam_flat_emb): up to two matrices, flat offsets relative to the
// launch's own `w`
struct AdamEmb {
  int n;
  int64_t lo[2], hi[2], dst[2];
  int32_t rows[2], cols[2];
  void* wt16;
};
template <typename T>
__device__ __forceinline__ void adam_emb_store(const AdamEmb& e, int64_t i, float wnew) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (j < e.n && i >= e.lo[j] && i < e.hi[j]) {
      const uint32_t k = (uint32_t)(i - e.lo[j]);
      const uint32_t r = k / (uint32_t)e.cols[j], c = k - r * (uint32_t)e.cols[j];
      const int64_t ld_t = ((int64_t)e.rows[j] + 7) / 8 * 8;
      reinterpret_cast<T*>(e.wt16)[e.dst[j] + (int64_t)c * ld_t + r] = from_f32<T>(wnew);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void adam_flat_kernel(int64_t n, float* __restrict__ w, const float* __restrict__ grad,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        T* __restrict__ w16, const int32_t* __restrict__ state,
                                                        float beta1, float beta2, float eps, float wd, float rescale,
                                                        float clip, mst_step_metrics mt, int32_t* state_rw, AdamEmb emb) {
  __shared__ float red[2][4];
  bool incomplete;
  if (step_is_bad(mt, incomplete)) {
    // the step's position-0 tail did not finish (mst_step_metrics): no update, no metric sums, the step count taken back —
    // by the launch that carries the step's bookkeeping (a second range of the same step only skips)
    if (blockIdx.x == 0 && threadIdx.x == 0 && mt.recon) {
      step_mark_bad(mt, incomplete);
      state_rw[0] -= 1;
    }
    return;
  }
  if (step_loss_nonfinite(mt)) {  // (uniform over the launch: every workgroup reads the same losses)
    if (blockIdx.x == 0 && threadIdx.x == 0 && mt.recon) {
      __hip_atomic_fetch_add(mt.status + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      state_rw[0] -= 1;
    }
    return;
  }
  if (blockIdx.x == 0 && mt.recon) loss_combine_wg(mt.B, mt.recon, mt.kl, mt.kl_weight, mt.total, mt.metric, red);  // (uniform branch)
  const float lr_t = reinterpret_cast<const float*>(state)[1];
  const int64_t nvec = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    f32x4 wv = reinterpret_cast<const f32x4*>(w)[i];
    f32x4 gv = reinterpret_cast<const f32x4*>(grad)[i];
    f32x4 mv = reinterpret_cast<const f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<const f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g = gv[e] * rescale + wd * wv[e];
      if (clip >= 0.f) g = fminf(fmaxf(g, -clip), clip);
      mv[e] = beta1 * mv[e] + (1.f - beta1) * g;
      vv[e] = beta2 * vv[e] + (1.f - beta2) * g * g;
      wv[e] = wv[e] - lr_t * mv[e] / (sqrtf(vv[e]) + eps);
    }
    reinterpret_cast<f32x4*>(w)[i] = wv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (w16) {
      u32x2 o;
      o[0] = (uint32_t)f32_to_bits<T>(wv[0]) | ((uint32_t)f32_to_bits<T>(wv[1]) << 16);
      o[1] = (uint32_t)f32_to_bits<T>(wv[2]) | ((uint32_t)f32_to_bits<T>(wv[3]) << 16);
      reinterpret_cast<u32x2*>(w16)[i] = o;
    }
    if (emb.n > 0 && ((4 * i + 3 >= emb.lo[0] && 4 * i < emb.hi[0]) || (emb.n > 1 && 4 * i + 3 >= emb.lo[1] && 4 * i < emb.hi[1]))) {
#pragma unroll
      for (int e = 0; e < 4; ++e) adam_emb_store<T>(emb, 4 * i + e, wv[e]);
    }
  }
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (int64_t i = nvec * 4 + threadIdx.x; i < n; i += 256) {
      float g = grad[i] * rescale + wd * w[i];
      if (clip >= 0.f) g = fminf(fmaxf(g, -clip), clip);
      const float mm = beta1 * m[i] + (1.f - beta1) * g;
      const float vv = beta2 * v[i] + (1.f - beta2) * g * g;
      const float ww = w[i] - lr_t * mm / (sqrtf(vv) + eps);
      m[i] = mm; v[i] = vv; w[i] = ww;
      if (w16) w16[i] = from_f32<T>(ww);
      adam_emb_store<T>(emb, i, ww);
    }
  }
}

// transposed 16-bit shadows: for matrix i, src fp32 [rows, cols] at w + desc[4i], dst [cols, ld_t] at
// wt16 + desc[4i+1] with ld_t = roundup8(rows); pad columns rows..ld_t are zeroed.
template <typename T>
__global__ __launch_bounds__(256) void transpose_shadows_kernel(const float* __restrict__ w, T* __restrict__ wt16,
                                                                const int64_t* __restrict__ desc,
                                                                const int64_t* __restrict__ tile_prefix, int n_mat) {
  __shared__ float tile[32][33];
  shadow_tile_wg<T>(w, wt16, desc, tile_prefix, n_mat, (int64_t)blockIdx.x, tile);
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(int64_t n, const float* __restrict__ src, T* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = from_f32<T>(src[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(int64_t n, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = from_f32<T>(to_f32(a[i]) + to_f32(b[i]));
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(int64_t n, float p, uint64_t seed, uint32_t site, uint8_t* keep) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    keep[i] = dropout_keep(seed, site, (uint64_t)i, p) ? 1 : 0;
}

}  // namespace mst

using namespace mst;

static unsigned grid_for(int64_t n, int per_thread) {
  int64_t g = cdiv(n, 256 * (int64_t)per_thread);
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (unsigned)g;
}

static int adam_flat_impl(int dtype, int64_t n, float* w, const float* grad, float* m, float* v, void* w16, double lr,
                          double beta1, double beta2, float eps, float wd, float rescale, float clip,
                          int32_t* step_state, int advance_step, const mst_step_metrics* metrics, const AdamEmb& emb, mst_stream_t stream) {
  MST_CHECK_ARG(n > 0 && w && grad && m && v && step_state, "mst_adam_flat: bad argument");
  mst_step_metrics mt = {};
  if (metrics) {
    MST_CHECK_ARG(metrics->recon == nullptr || (metrics->B > 0 && metrics->kl), "mst_adam_flat: metrics need B, recon and kl");
    MST_CHECK_ARG(metrics->status || (!metrics->expect_ptr0 && !metrics->expect_ptr1), "mst_adam_flat: expectations need the status words");
    mt = *metrics;
  }
  MST_CHECK_ARG(((uintptr_t)w % 16 == 0) && ((uintptr_t)grad % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
                "mst_adam_flat: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (advance_step) {  // a second launch over another range of the same step passes 0
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, step_state, lr, beta1, beta2);
    MST_CHECK_LAUNCH("adam_tick_kernel");
  }
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    hipLaunchKernelGGL((adam_flat_kernel<T>), dim3(grid_for(n, 4)), dim3(256), 0, s, n, w, grad, m, v, (T*)w16, step_state,
                       (float)beta1, (float)beta2, eps, wd, rescale, clip, mt, step_state, emb);
    MST_CHECK_LAUNCH("adam_flat_kernel");
    return MST_OK;
  });
}

extern "C" int mst_adam_flat(int dtype, int64_t n, float* w, const float* grad, float* m, float* v, void* w16, double lr,
                             double beta1, double beta2, float eps, float wd, float rescale, float clip,
                             int32_t* step_state, int advance_step, const mst_step_metrics* metrics, mst_stream_t stream) {
  AdamEmb none = {};
  return adam_flat_impl(dtype, n, w, grad, m, v, w16, lr, beta1, beta2, eps, wd, rescale, clip, step_state, advance_step, metrics, none, stream);
}

extern "C" int mst_adam_flat_emb(int dtype, int64_t n, float* w, const float* grad, float* m, float* v, void* w16, double lr, double beta1,
                                 double beta2, float eps, float wd, float rescale, float clip, int32_t* step_state,
                                 const mst_step_metrics* metrics, int64_t base, const int64_t* emb, int64_t n_emb, void* wt16,
                                 mst_stream_t stream) {
  MST_CHECK_ARG(n_emb >= 0 && n_emb <= 2 && (n_emb == 0 || (emb && wt16)) && base >= 0, "mst_adam_flat_emb: up to two matrices, with their table and wt16");
  AdamEmb e = {};
  for (int j = 0; j < (int)n_emb; ++j) {
    const int64_t so = emb[4 * j], dst = emb[4 * j + 1], rows = emb[4 * j + 2], cols = emb[4 * j + 3];
    MST_CHECK_ARG(rows > 0 && cols > 0 && rows * cols < (1ll << 31) && dst >= 0, "mst_adam_flat_emb: bad matrix %d", j);
    // (a matrix outside this launch's range [base, base + n) simply never matches)
    e.lo[e.n] = so - base; e.hi[e.n] = so - base + rows * cols; e.dst[e.n] = dst; e.rows[e.n] = (int32_t)rows; e.cols[e.n] = (int32_t)cols;
    ++e.n;
  }
  e.wt16 = wt16;
  return adam_flat_impl(dtype, n, w, grad, m, v, w16, lr, beta1, beta2, eps, wd, rescale, clip, step_state, 0, metrics, e, stream);
}

extern "C" int mst_transpose_shadows(int dtype, const float* w, void* wt16, const int64_t* desc,
                                     const int64_t* tile_prefix, int64_t n_mat, int64_t total_tiles,
                                     mst_stream_t stream) {
  MST_CHECK_ARG(w && wt16 && desc && tile_prefix && n_mat > 0 && total_tiles > 0, "mst_transpose_shadows: bad argument");
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    hipLaunchKernelGGL((transpose_shadows_kernel<T>), dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, w, (T*)wt16,
                       desc, tile_prefix, (int)n_mat);
    MST_CHECK_LAUNCH("transpose_shadows_kernel");
    return MST_OK;
  });
}

// per-tensor gradient norms for the reference's periodic gradient log (trainer.py:257-270): out[i] = sum of squares of
// x[offsets[i] .. offsets[i+1]) — one workgroup per tensor of the flat bucket, one launch for all of them
__global__ __launch_bounds__(256) void segment_sumsq_kernel(const float* __restrict__ x, const int64_t* __restrict__ offsets,
                                                            float* __restrict__ out) {
  __shared__ float red[4];
  const int64_t lo = offsets[2 * blockIdx.x], hi = offsets[2 * blockIdx.x + 1];
  float acc = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) acc += x[i] * x[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int mst_segment_sumsq(const float* x, const int64_t* ranges, int64_t n_segments, float* out, mst_stream_t stream) {
  MST_CHECK_ARG(x && ranges && out && n_segments > 0 && n_segments <= 65535, "mst_segment_sumsq: bad argument");
  hipLaunchKernelGGL(segment_sumsq_kernel, dim3((unsigned)n_segments), dim3(256), 0, (hipStream_t)stream, x, ranges, out);
  MST_CHECK_LAUNCH("segment_sumsq_kernel");
  return MST_OK;
}

extern "C" int mst_cast_f32_to_act(int dtype, int64_t n, const float* src, void* dst, mst_stream_t stream) {
  MST_CHECK_ARG(n > 0 && src && dst, "mst_cast_f32_to_act: bad argument");
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    hipLaunchKernelGGL((cast_kernel<T>), dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, n, src, (T*)dst);
    MST_CHECK_LAUNCH("cast_kernel");
    return MST_OK;
  });
}

extern "C" int mst_add_act(int dtype, int64_t n, const void* a, const void* b, void* y, mst_stream_t stream) {
  MST_CHECK_ARG(n > 0 && a && b && y, "mst_add_act: bad argument");
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, n, (const T*)a, (const T*)b, (T*)y);
    MST_CHECK_LAUNCH("add_kernel");
    return MST_OK;
  });
}

extern "C" int mst_dropout_mask(int64_t n, float p, uint64_t seed, uint32_t site, uint8_t* keep, mst_stream_t stream) {
  MST_CHECK_ARG(n > 0 && keep && p >= 0.f && p < 1.f, "mst_dropout_mask: bad argument");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, n, p, seed, site, keep);
  MST_CHECK_LAUNCH("dropout_mask_kernel");
  return MST_OK;
}

// losses.hip — fused loss heads, wavefront reductions, fp32 math on 16-bit logits.
//
//   mst_softmax_ce    : softmax over the vocabulary + SoftmaxCrossEntropy
//                       (VarAutoEncoder/model.py:256, loss.py:15-23)
//   mst_sigmoid_bce   : sigmoid + BinaryCrossEntropy with label smoothing and the reference's
//                       negative-label down-weighting w_b * bce^2 (loss.py:27-80)
//   mst_reparam_kl_*  : z = mu + eps*sigma and VariationalKLLoss (model.py:292, loss.py:4-12)
//   mst_loss_combine  : loss = recon + kl_weight * kl and the running metric sums
//                       (trainer.py:107-120,172,181-186)
//
// All are HBM-bound: one read of logits (+labels), optional writes of probabilities and of the
// logit gradient in the same pass, per-sample sums reduced in-wave then with one atomic per wave.
#include <math.h>
#include <type_traits>
#include "common.hpp"
#include "loss_combine.hpp"
#include "bce_math.hpp"

namespace mst {

// ------------------------------------------------------------------ softmax + CE
template <typename T>
__global__ __launch_bounds__(256) void softmax_ce_kernel(int64_t M, int64_t T_len, int V, const T* __restrict__ logits,
                                                         int64_t ld, const int32_t* __restrict__ labels,
                                                         float* __restrict__ loss, float* __restrict__ probs,
                                                         int64_t ldp, T* __restrict__ dlogits, int64_t ldd,
                                                         float gscale, float* __restrict__ tok_parts, int top_k) {
  __shared__ float tred[4][4];
  float t_nll = 0.f, t_acc = 0.f, t_topk = 0.f, t_n = 0.f;  // lane 0 of every wave: token metrics of the wave's rows
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float inv_T = 1.f / (float)T_len;
  for (int64_t m = wave_global; m < M; m += nwaves) {
    const T* row = logits + m * ld;
    const int label = labels[m];
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, to_f32(row[v]));
    mx = wave_max(mx);
    float se = 0.f;
    for (int v = lane; v < V; v += 64) se += __expf(to_f32(row[v]) - mx);
    se = wave_sum(se);
    const float lse = mx + __logf(se);
    const float inv = 1.f / se;
    const float maskv = (label != 0) ? 1.f : 0.f;
    const float xl = to_f32(row[label]);
    if (lane == 0) {
      const float lp = xl - lse;  // log p[label]
      atomicAdd(loss + m / T_len, -lp * maskv * inv_T);
    }
    if (tok_parts && label != 0) {
      // masked token metrics (trainer.py:107-113; metrics.py): rank of the label = number of entries that beat it
      // (ties go to the lower index, as argmax does); perplexity term -log(max(p, 1e-10)) like mx.metric.Perplexity
      float beat = 0.f;
      for (int v = lane; v < V; v += 64) {
        const float xv = to_f32(row[v]);
        beat += (xv > xl || (xv == xl && v < label)) ? 1.f : 0.f;
      }
      beat = wave_sum(beat);
      if (lane == 0) {
        t_nll += -__logf(fmaxf(__expf(xl - lse), 1e-10f));
        t_acc += beat == 0.f ? 1.f : 0.f;
        t_topk += beat < (float)top_k ? 1.f : 0.f;
        t_n += 1.f;
      }
    }
    if (probs || dlogits) {
      const float gs = maskv * inv_T * gscale;
      const int vend = dlogits ? (int)((V + 3) / 4 * 4 < ldd ? (V + 3) / 4 * 4 : ldd) : V;
      for (int v = lane; v < vend; v += 64) {
        float p = 0.f;
        if (v < V) p = __expf(to_f32(row[v]) - mx) * inv;
        if (probs && v < V) probs[m * ldp + v] = p;
        if (dlogits) dlogits[m * ldd + v] = from_f32<T>(v < V ? (p - (v == label ? 1.f : 0.f)) * gs : 0.f);
      }
    }
  }
  if (tok_parts) {  // one row of partial sums per workgroup, owned by it: plain read-modify-write, no atomics
    if (lane == 0) { float* r = tred[threadIdx.x >> 6]; r[0] = t_nll; r[1] = t_acc; r[2] = t_topk; r[3] = t_n; }
    __syncthreads();
    if (threadIdx.x < 4) {
      const int k = threadIdx.x;
      tok_parts[(int64_t)blockIdx.x * 4 + k] += tred[0][k] + tred[1][k] + tred[2][k] + tred[3][k];
    }
  }
}

// SoftmaxCrossEntropy on PROBABILITIES, the reference's call form (loss.py:16-23: log(pred), pick, mask, mean over the
// non-batch axes = / padded T): one element per row is read.
template <typename TP>
__global__ __launch_bounds__(256) void ce_probs_kernel(int64_t B, int64_t T_len, const TP* __restrict__ probs, int64_t ldp,
                                                       const int32_t* __restrict__ labels, float* __restrict__ loss) {
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  float acc = 0.f;
  for (int64_t t = threadIdx.x; t < T_len; t += 256) {
    const int label = labels[b * T_len + t];
    if (label != 0) acc += -__logf((float)probs[(b * T_len + t) * ldp + label]);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[b] = (red[0] + red[1] + red[2] + red[3]) / (float)T_len;
}

// ------------------------------------------------------------------ sigmoid + BCE
// A few workgroups of 1024 threads per sample (grid.x of them): per-sample sums are block reductions followed by ONE
// atomic per workgroup — same-address atomics are serialised at ~0.3 us each, the former wave-level adds (32 to 128
// per sample) were half the kernel's time. Every workgroup counts the sample's positives itself (32 KB of labels)
// instead of a counting launch + atomics in front.
constexpr int BCE_THREADS = 1024;
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();  // red may still be read from a previous use
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < BCE_THREADS / 64; ++w) t += red[w];
  return t;
}

template <typename T>
__global__ __launch_bounds__(BCE_THREADS) void sigmoid_bce_kernel(int64_t rows_per_sample, int P, const T* __restrict__ logits,
                                                                  int64_t ld, const uint8_t* __restrict__ labels, float ls,
                                                                  int downweight, int32_t* __restrict__ npos,
                                                                  float* __restrict__ loss, T* __restrict__ probs, int64_t ldp,
                                                                  T* __restrict__ dlogits, int64_t ldd, float gscale) {
  __shared__ float red[BCE_THREADS / 64];
  const int64_t b = blockIdx.y;
  const int tid = threadIdx.x;
  const int64_t per_sample = rows_per_sample * P;
  const float inv_n = 1.f / ((float)rows_per_sample * (float)P);
  float w = 0.f;
  if (downweight) {  // labels are {0,1} bytes
    const uint8_t* base = labels + b * per_sample;
    int cnt = 0;
    if ((reinterpret_cast<uintptr_t>(base) & 7) == 0) {
      for (int64_t i = (int64_t)tid * 8; i < per_sample; i += (int64_t)BCE_THREADS * 8) {
        if (i + 8 <= per_sample) cnt += __popcll(*reinterpret_cast<const uint64_t*>(base + i) & 0x0101010101010101ull);
        else for (int64_t j = i; j < per_sample; ++j) cnt += (base[j] == 1);
      }
    } else {
      for (int64_t i = tid; i < per_sample; i += BCE_THREADS) cnt += (base[i] == 1);
    }
    const float np = block_sum_1024((float)cnt, red);
    if (tid == 0 && npos && blockIdx.x == 0) npos[b] = (int)np;
    const float nn = (float)rows_per_sample * (float)P - np;
    w = np / (nn + 1e-12f);
  }
  // a thread handles VEC consecutive pitches of one frame per iteration: 8 (16-byte logit / probability / gradient accesses, one
  // 8-byte label load) where the rows allow, else 4
  const float s1 = (1.f - ls) + 0.5f * ls, s0 = 0.5f * ls;
  float acc = 0.f;
  auto sweep = [&](auto dwc, auto vecc) {
    constexpr bool DW = decltype(dwc)::value;
    constexpr int VEC = decltype(vecc)::value;
    const int vec_per_row = (P + VEC - 1) / VEC;
    const int64_t nvec = rows_per_sample * vec_per_row;
    for (int64_t i = (int64_t)blockIdx.x * BCE_THREADS + tid; i < nvec; i += (int64_t)gridDim.x * BCE_THREADS) {
      // (32-bit quotient / remainder: nvec < 2^31 is checked on the host; the 64-bit forms were ~200 instructions per thread
      // in a kernel whose threads handle one vector each)
      const uint32_t iq = (uint32_t)i / (uint32_t)vec_per_row;
      const int64_t r = b * rows_per_sample + iq;
      const int c0 = (int)((uint32_t)i - iq * (uint32_t)vec_per_row) * VEC;
      float x[VEC];
      uint64_t lab = 0;  // the label bytes in one load when the row is aligned
      if constexpr (VEC == 8) {
        Pack8 raw;
        raw.u = *reinterpret_cast<const u32x4*>(logits + r * ld + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = bits_to_f32<T>(raw.h[e]);
        lab = *reinterpret_cast<const uint64_t*>(labels + r * P + c0);
      } else {
        u32x2 raw = *reinterpret_cast<const u32x2*>(logits + r * ld + c0);
        x[0] = bits_to_f32<T>((uint16_t)(raw[0] & 0xffff)); x[1] = bits_to_f32<T>((uint16_t)(raw[0] >> 16));
        x[2] = bits_to_f32<T>((uint16_t)(raw[1] & 0xffff)); x[3] = bits_to_f32<T>((uint16_t)(raw[1] >> 16));
        if (P % 4 == 0) {
          lab = *reinterpret_cast<const uint32_t*>(labels + r * P + c0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < P) lab |= (uint64_t)labels[r * P + c0 + e] << (8 * e);
        }
      }
      bool in_dom = true;  // (pad columns hold zero logits)
#pragma unroll
      for (int e = 0; e < VEC; ++e) in_dom = in_dom && ((VEC != 8 && c0 + e >= P) || bce_fast_domain(x[e]));
      const bool fast = __all(in_dom);  // wave-uniform: three transcendental instructions per element instead of six (bce_math.hpp)
      Pack8 pv, gv;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int c = c0 + e;
        pv.h[e] = 0; gv.h[e] = 0;
        if (VEC == 8 || c < P) {  // (VEC 8: P is a multiple of 8, no partial vector)
          const float y = (float)((lab >> (8 * e)) & 0xFFull);
          float p, bce, dbce;
          bce_fast<DW>(x[e], y, s1, s0, w, p, bce, dbce);
          if (!fast) {  // a saturated logit somewhere in this wave: ITS element takes the reference's operation order (an element's
                        // result depends on its own logit only, so the fused and the two-launch forms agree bit for bit)
            float p2, b2, d2;
            bce_exact<DW>(x[e], y, ls, w, p2, b2, d2);
            if (!bce_fast_domain(x[e])) { p = p2; bce = b2; dbce = d2; }
          }
          acc += bce;
          pv.h[e] = f32_to_bits<T>(p);
          gv.h[e] = f32_to_bits<T>(dbce * inv_n * gscale);
        }
      }
      if constexpr (VEC == 8) {
        if (probs) *reinterpret_cast<u32x4*>(probs + r * ldp + c0) = pv.u;
        if (dlogits) *reinterpret_cast<u32x4*>(dlogits + r * ldd + c0) = gv.u;
      } else {
        if (probs) *reinterpret_cast<u32x2*>(probs + r * ldp + c0) = u32x2{pv.u[0], pv.u[1]};
        if (dlogits) *reinterpret_cast<u32x2*>(dlogits + r * ldd + c0) = u32x2{gv.u[0], gv.u[1]};
      }
    }
  };
  const bool wide = P % 8 == 0 && ld % 8 == 0 && (!probs || ldp % 8 == 0) && (!dlogits || ldd % 8 == 0) &&
                    (((uintptr_t)logits | (uintptr_t)probs | (uintptr_t)dlogits) & 15) == 0 && ((uintptr_t)labels & 7) == 0;
  typedef std::integral_constant<int, 8> V8;
  typedef std::integral_constant<int, 4> V4;
  if (downweight) { if (wide) sweep(std::true_type(), V8()); else sweep(std::true_type(), V4()); }
  else { if (wide) sweep(std::false_type(), V8()); else sweep(std::false_type(), V4()); }
  const float total = block_sum_1024(acc, red);
  if (tid == 0) atomicAdd(loss + b, total * inv_n);
}

// BinaryCrossEntropy(from_sigmoid=True) (loss.py:40-56): `pred` already holds probabilities. Forward only; one
// workgroup per sample, deterministic (no atomics).
template <typename TP>
__global__ __launch_bounds__(BCE_THREADS) void bce_probs_kernel(int64_t n_per_sample, const TP* __restrict__ probs,
                                                                const uint8_t* __restrict__ labels, float ls, int downweight,
                                                                float* __restrict__ loss) {
  __shared__ float red[BCE_THREADS / 64];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const TP* p0 = probs + b * n_per_sample;
  const uint8_t* y0 = labels + b * n_per_sample;
  float w = 0.f;
  if (downweight) {
    float cnt = 0.f;
    for (int64_t i = tid; i < n_per_sample; i += BCE_THREADS) cnt += (y0[i] == 1) ? 1.f : 0.f;
    const float np = block_sum_1024(cnt, red);
    w = np / (((float)n_per_sample - np) + 1e-12f);
  }
  float acc = 0.f;
  for (int64_t i = tid; i < n_per_sample; i += BCE_THREADS) {
    const float p = (float)p0[i], y = (float)y0[i];
    const float s = (1.f - ls) * y + 0.5f * ls;
    float bce = -(s * __logf(1e-12f + p) + (1.f - s) * __logf(1e-12f + (1.f - p)));
    if (downweight && y == 0.f) bce = w * bce * bce;
    acc += bce;
  }
  const float total = block_sum_1024(acc, red);
  if (tid == 0) loss[b] = total / (float)n_per_sample;
}

// ------------------------------------------------------------------ reparameterisation + KL
__global__ __launch_bounds__(64) void reparam_kl_fwd_kernel(int Z, const float* __restrict__ mu,
                                                            const float* __restrict__ sigma,
                                                            const float* __restrict__ eps, float* __restrict__ z,
                                                            float* __restrict__ kl) {
  const int64_t b = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < Z; i += 64) {
    const float m = mu[b * Z + i], s = sigma[b * Z + i];
    z[b * Z + i] = m + eps[b * Z + i] * s;
    const float s2 = s * s;
    acc += 0.5f * (s2 + m * m - 1.f - logf(s2));
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) kl[b] = acc;
}

__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(int64_t n, const float* __restrict__ mu,
                                                             const float* __restrict__ sigma,
                                                             const float* __restrict__ eps,
                                                             const float* __restrict__ dz, float kl_weight,
                                                             float* __restrict__ dmu, float* __restrict__ dsigma) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float g = dz ? dz[i] : 0.f;
    const float s = sigma[i];
    dmu[i] = kl_weight * mu[i] + g;
    dsigma[i] = kl_weight * (s - 1.f / s) + eps[i] * g;
  }
}

__global__ __launch_bounds__(256) void loss_combine_kernel(int64_t B, const float* __restrict__ recon,
                                                           const float* __restrict__ kl, float kl_weight,
                                                           float* __restrict__ total, float* __restrict__ metric) {
  __shared__ float red[2][4];
  loss_combine_wg(B, recon, kl, kl_weight, total, metric, red);
}

}  // namespace mst

using namespace mst;

extern "C" int mst_softmax_ce(int dtype, int64_t B, int64_t T, int64_t V, const void* logits, int64_t ld,
                              const int32_t* labels, float* loss, float* probs, int64_t ldp, void* dlogits,
                              int64_t ldd, float gscale, int pre_zeroed, float* tok_parts, int top_k, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && T > 0 && V > 0, "mst_softmax_ce: B,T,V must be positive");
  MST_CHECK_ARG(logits && labels && loss, "mst_softmax_ce: null pointer");
  MST_CHECK_ARG(ld >= V && (!probs || ldp >= V) && (!dlogits || ldd >= V), "mst_softmax_ce: leading dim < V");
  hipStream_t s = (hipStream_t)stream;
  if (!pre_zeroed) {
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float) * B, s);
    if (e != hipSuccess) { set_error("mst_softmax_ce: memset: %s", hipGetErrorString(e)); return MST_ERR_LAUNCH; }
  }
  const int64_t M = B * T;
  const unsigned grid = (unsigned)(cdiv(M, 4) < MST_CE_MAX_WORKGROUPS ? cdiv(M, 4) : MST_CE_MAX_WORKGROUPS);
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) TT;
    hipLaunchKernelGGL((softmax_ce_kernel<TT>), dim3(grid), dim3(256), 0, s, M, T, (int)V, (const TT*)logits, ld, labels,
                       loss, probs, ldp, (TT*)dlogits, ldd, gscale, tok_parts, top_k);
    MST_CHECK_LAUNCH("softmax_ce_kernel");
    return MST_OK;
  });
}

template <typename F> static int dispatch_prob(int dtype, F&& f) {
  if (dtype == MST_F32) return f(0.f);
  return dispatch_act(dtype, f);
}

extern "C" int mst_ce_from_probs(int dtype, int64_t B, int64_t T, int64_t V, const void* probs, int64_t ldp,
                                 const int32_t* labels, float* loss, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && T > 0 && V > 0 && probs && labels && loss && ldp >= V, "mst_ce_from_probs: bad argument");
  return dispatch_prob(dtype, [&](auto tag) -> int {
    typedef decltype(tag) TP;
    hipLaunchKernelGGL((ce_probs_kernel<TP>), dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, B, T, (const TP*)probs, ldp,
                       labels, loss);
    MST_CHECK_LAUNCH("ce_probs_kernel");
    return MST_OK;
  });
}

extern "C" int mst_bce_from_probs(int dtype, int64_t B, int64_t n_per_sample, const void* probs, const uint8_t* labels,
                                  float label_smoothing, int downweight, float* loss, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && n_per_sample > 0 && probs && labels && loss, "mst_bce_from_probs: bad argument");
  return dispatch_prob(dtype, [&](auto tag) -> int {
    typedef decltype(tag) TP;
    hipLaunchKernelGGL((bce_probs_kernel<TP>), dim3((unsigned)B), dim3(BCE_THREADS), 0, (hipStream_t)stream, n_per_sample,
                       (const TP*)probs, labels, label_smoothing, downweight, loss);
    MST_CHECK_LAUNCH("bce_probs_kernel");
    return MST_OK;
  });
}

extern "C" int mst_sigmoid_bce(int dtype, int64_t B, int64_t T, int64_t P, const void* logits, int64_t ld,
                               const uint8_t* labels, float label_smoothing, int downweight, int32_t* npos,
                               float* loss, void* probs, int64_t ldp, void* dlogits, int64_t ldd, float gscale,
                               int pre_zeroed, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && T > 0 && P > 0, "mst_sigmoid_bce: B,T,P must be positive");
  MST_CHECK_ARG(logits && labels && loss, "mst_sigmoid_bce: null pointer");
  MST_CHECK_ARG(ld % 4 == 0 && ld >= P, "mst_sigmoid_bce: ld must be a multiple of 4 and >= P");
  MST_CHECK_ARG(!probs || (ldp % 4 == 0 && ldp >= P), "mst_sigmoid_bce: bad ldp");
  MST_CHECK_ARG(!dlogits || (ldd % 4 == 0 && ldd >= P), "mst_sigmoid_bce: bad ldd");
  MST_CHECK_ARG(!downweight || npos, "mst_sigmoid_bce: down-weighting needs the npos scratch");
  MST_CHECK_ARG(B <= 65535, "mst_sigmoid_bce: B too large for grid.y");
  MST_CHECK_ARG(T * ((P + 3) / 4) < (1ll << 31), "mst_sigmoid_bce: T * P / 4 must stay below 2^31");
  hipStream_t s = (hipStream_t)stream;
  if (!pre_zeroed) {
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float) * B, s);
    if (e != hipSuccess) { set_error("mst_sigmoid_bce: memset: %s", hipGetErrorString(e)); return MST_ERR_LAUNCH; }
  }
  // enough workgroups to put ~2 on every CU, at least one 4-pitch vector per thread
  const int64_t nvec = T * ((P + 3) / 4);
  int64_t gx = cdiv(512, B);
  if (gx > cdiv(nvec, BCE_THREADS)) gx = cdiv(nvec, BCE_THREADS);
  if (gx < 1) gx = 1;
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) TT;
    hipLaunchKernelGGL((sigmoid_bce_kernel<TT>), dim3((unsigned)gx, (unsigned)B), dim3(BCE_THREADS), 0, s, T, (int)P, (const TT*)logits,
                       ld, labels, label_smoothing, downweight, npos, loss, (TT*)probs, ldp, (TT*)dlogits, ldd, gscale);
    MST_CHECK_LAUNCH("sigmoid_bce_kernel");
    return MST_OK;
  });
}

extern "C" int mst_reparam_kl_fwd(int64_t B, int64_t Z, const float* mu, const float* sigma, const float* eps,
                                  float* z, float* kl, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && Z > 0 && mu && sigma && eps && z && kl, "mst_reparam_kl_fwd: bad argument");
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, (int)Z, mu, sigma, eps, z, kl);
  MST_CHECK_LAUNCH("reparam_kl_fwd_kernel");
  return MST_OK;
}

extern "C" int mst_reparam_kl_bwd(int64_t B, int64_t Z, const float* mu, const float* sigma, const float* eps,
                                  const float* dz, float kl_weight, float* dmu, float* dsigma, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && Z > 0 && mu && sigma && eps && dmu && dsigma, "mst_reparam_kl_bwd: bad argument");
  const int64_t n = B * Z;
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, mu, sigma,
                     eps, dz, kl_weight, dmu, dsigma);
  MST_CHECK_LAUNCH("reparam_kl_bwd_kernel");
  return MST_OK;
}

__global__ __launch_bounds__(256) void loss_combine_guarded_kernel(mst_step_metrics mt) {
  __shared__ float red[2][4];
  bool incomplete;
  if (step_is_bad(mt, incomplete)) {
    if (threadIdx.x == 0) step_mark_bad(mt, incomplete);
    return;
  }
  loss_combine_wg(mt.B, mt.recon, mt.kl, mt.kl_weight, mt.total, mt.metric, red);
}

extern "C" int mst_loss_combine_v(const mst_step_metrics* metrics, mst_stream_t stream) {
  MST_CHECK_ARG(metrics && metrics->B > 0 && metrics->recon && metrics->kl, "mst_loss_combine_v: bad argument");
  MST_CHECK_ARG(metrics->status || (!metrics->expect_ptr0 && !metrics->expect_ptr1), "mst_loss_combine_v: expectations need the status words");
  hipLaunchKernelGGL(loss_combine_guarded_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *metrics);
  MST_CHECK_LAUNCH("loss_combine_guarded_kernel");
  return MST_OK;
}

extern "C" int mst_loss_combine(int64_t B, const float* recon, const float* kl, float kl_weight, float* total,
                                float* metric_acc, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && recon && kl, "mst_loss_combine: bad argument");
  hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B, recon, kl, kl_weight, total, metric_acc);
  MST_CHECK_LAUNCH("loss_combine_kernel");
  return MST_OK;
}

// latent_fwd.hpp — the forward half of the VAE bottleneck as a workgroup-level function, so that the launch can carry other
// workgroups beside it (gemm_nt.hip: mst_latent_fwd_qkv — the decoder's first K | Q | V projection rides on this launch).
// See latent.hip for the arithmetic and its reference lines.
#pragma once
#include <math.h>
#include <type_traits>
#include "common.hpp"

namespace mst {

constexpr int LAT_THREADS = 1024;  // 16 waves: these kernels are B workgroups of dependent dot products (latency-bound)
constexpr int OPW = 8;             // outputs a wave works on at once
constexpr int PRE_C = 4;           // 64-lane chunks of a contraction whose weights the forward keeps in registers (De <= 256)

// The U cross-lane sums of a pass, then ONE emit from lanes 0 .. U-1 (lane u: output j0 + u). Reducing and emitting output by output
// made every reduction wait for the previous emit's memory operation (the swizzle / permute levels and LDS writes share a counter):
// ~300 cycles x U in a dependent chain — 3 us for the 24 outputs per wave of the latent block's K | Q | V projection.
template <int U, typename F>
__device__ __forceinline__ void reduce_emit(const float (&acc)[U], int j0, int n_out, int lane, F&& emit) {
  static_assert(U <= 64, "one lane per output of the pass");
  float mine = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float v = wave_sum(acc[u]);
    if (lane == u) mine = v;
  }
  if (lane < U && j0 + lane < n_out) emit(j0 + lane, mine);
}

// out[j] = sum_d x[d] * W[j, d] for j < n_out: wave w takes outputs [w*U, w*U+U), then strides by n_waves*U;
// `emit(j, value)` runs on lane j - j0 of the wave that owns the pass
template <int U, typename F>
__device__ __forceinline__ void wave_dots(const float* x, int n_in, const float* __restrict__ W, int n_out, int wave, int n_waves,
                                          int lane, F&& emit) {
  for (int j0 = wave * U; j0 < n_out; j0 += n_waves * U) {
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.f;
    for (int d = lane; d < n_in; d += 64) {
      const float xv = x[d];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (j0 + u < n_out) acc[u] = fmaf(xv, W[(int64_t)(j0 + u) * n_in + d], acc[u]);
    }
    reduce_emit<U>(acc, j0, n_out, lane, emit);
  }
}

// wave_dots for contractions of at most 64 * CH elements with every weight load of a pass of U outputs issued up front and the
// NEXT pass's loads in flight while the current one is reduced (two register sets): the weights are cold lines after every
// optimizer step, and one dependent round trip per 64 elements of every output made latent_fwd 58 us at configs[2]
// (2Z = 512 outputs of 256: four passes of four). Same FMA order per output as wave_dots. Loads are unconditional
// (indices clamped, surplus products multiplied by zero): a conditional load costs a vmcnt(0) drain at the join.
template <int U, int CH, typename F>
__device__ __forceinline__ void wave_dots_pre(const float* x, int n_in, const float* __restrict__ W, int n_out, int wave, int n_waves,
                                              int lane, F&& emit) {
  const int step = n_waves * U;
  float xs[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) xs[k] = (lane + 64 * k < n_in) ? x[lane + 64 * k] : 0.f;
  auto load = [&](float (&w)[U][CH], int j0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u < n_out ? j0 + u : n_out - 1;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int d = lane + 64 * k < n_in ? lane + 64 * k : n_in - 1;
        w[u][k] = W[(int64_t)j * n_in + d];
      }
    }
  };
  auto reduce = [&](const float (&w)[U][CH], int j0) {
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[u] = 0.f;
#pragma unroll
      for (int k = 0; k < CH; ++k)
        if (lane + 64 * k < n_in) acc[u] = fmaf(xs[k], w[u][k], acc[u]);
    }
    reduce_emit<U>(acc, j0, n_out, lane, emit);
  };
  int j0 = wave * U;
  if (j0 >= n_out) return;
  float wa[U][CH], wb[U][CH];
  load(wa, j0);
  for (;;) {
    const int j1 = j0 + step;
    load(wb, j1 < n_out ? j1 : j0);  // (past the end: the same rows again, unused)
    reduce(wa, j0);
    if (j1 >= n_out) break;
    const int j2 = j1 + step;
    load(wa, j2 < n_out ? j2 : j1);
    reduce(wb, j1);
    if (j2 >= n_out) break;
    j0 = j2;
  }
}

// PRE: the small-shape form (both products in one pass of OPW outputs per wave, weights preloaded: latent_fwd_pre_shape); the
// general form is a separate instantiation so that its two register sets per product do not cost the small one its registers
// (in one kernel the 1024-thread launch bounds made the compiler spill 16 / 94 registers of the forward / backward fast paths).
__host__ __device__ inline bool latent_fwd_pre_shape(int64_t De, int64_t Z, int64_t Dd) {
  return 2 * Z <= (LAT_THREADS / 64) * OPW && De <= 64 * PRE_C && Dd <= (LAT_THREADS / 64) * OPW && Z <= 64;
}
struct LatentFwdArgs {
  int De, Z, Dd;
  const void* enc_out; int64_t enc_stride;
  const float *Wl, *bl, *eps, *Wh, *bh;
  const int32_t* classes;
  const float* cls_d; int64_t ld_cls;
  const float* pos_d; float alpha_d;
  float *mu, *sigma, *z, *kl;
  void* dec_in; int64_t dec_stride;
  // optional: the decoder's first K | Q | V projection of THIS row (position 0 of the sample, which no other workgroup of the
  // step has until now): qkv0[b * qkv_stride + j] = dec_in[b, 0, :] . Wq[j, :] + bq[j], j < nq. Wq: the 16-bit shadow, row-major
  const void* Wq; int64_t ld_wq; const float* bq; void* qkv0; int64_t qkv_stride; int nq;
};

// out[j] = sum_d x[d] * W[j, d] with 16-bit weights (the forward shadows), n_in = 64 * VEC: a lane owns VEC consecutive elements of
// the contraction (one 2 * VEC-byte load per output row), a wave U outputs per pass with all U loads in flight — at the decoder's
// width (n_in 128, 384 outputs over 16 waves) ONE pass, i.e. one memory round trip in the latent block's dependent chain (the
// first form, 8 outputs per pass and two 2-byte loads per output, added 12 us to the launch)
template <typename T, int U, int VEC, typename F>
__device__ __forceinline__ void wave_dots_t(const float* x, const T* __restrict__ W, int64_t ldw, int n_out, int wave, int n_waves, int lane,
                                            F&& emit) {
  typedef typename std::conditional<VEC == 4, u32x2, typename std::conditional<VEC == 2, uint32_t, uint16_t>::type>::type raw_t;
  float xs[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) xs[e] = x[lane * VEC + e];
  for (int j0 = wave * U; j0 < n_out; j0 += n_waves * U) {
    raw_t w[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      w[u] = *reinterpret_cast<const raw_t*>(W + (int64_t)(j0 + u < n_out ? j0 + u : n_out - 1) * ldw + lane * VEC);
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint16_t h[VEC];
      __builtin_memcpy(h, &w[u], sizeof(raw_t));
      acc[u] = 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[u] = fmaf(xs[e], bits_to_f32<T>(h[e]), acc[u]);
    }
    reduce_emit<U>(acc, j0, n_out, lane, emit);
  }
}

// LDS: De + 3 Z floats (+ Dd with the projection), at `sm`
template <typename T, bool PRE>
__device__ __forceinline__ void latent_fwd_wg(const LatentFwdArgs& A, int64_t b, float* sm) {
  const int De = A.De, Z = A.Z, Dd = A.Dd;
  const T* __restrict__ enc_out = reinterpret_cast<const T*>(A.enc_out);
  const int64_t enc_stride = A.enc_stride, ld_cls = A.ld_cls, dec_stride = A.dec_stride;
  const float* __restrict__ Wl = A.Wl; const float* __restrict__ bl = A.bl; const float* __restrict__ eps = A.eps;
  const float* __restrict__ Wh = A.Wh; const float* __restrict__ bh = A.bh; const int32_t* __restrict__ classes = A.classes;
  const float* __restrict__ cls_d = A.cls_d; const float* __restrict__ pos_d = A.pos_d; const float alpha_d = A.alpha_d;
  float* __restrict__ mu = A.mu; float* __restrict__ sigma = A.sigma; float* __restrict__ z = A.z; float* __restrict__ kl = A.kl;
  T* __restrict__ dec_in = reinterpret_cast<T*>(A.dec_in);
  float* h0 = sm;            // [De]
  float* lat = sm + De;      // [2Z]
  float* zs = lat + 2 * Z;   // [Z]
  __shared__ float klred[LAT_THREADS / 64];
  constexpr int NW = LAT_THREADS / 64;
  float* x0s = zs + Z;       // [Dd] (projection only): the decoder input row as stored
  const bool proj = A.Wq != nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (projection) its bias: a cold line after every optimizer step — requested now, it waits in LDS behind x0s (one load per
  // OUTPUT inside the dot products' emit was a chain of 24 dependent round trips per wave: +6.6 us on the launch)
  float* bqs = x0s + Dd;      // [nq]
  if (proj) for (int j = tid; j < A.nq; j += LAT_THREADS) bqs[j] = A.bq ? A.bq[j] : 0.f;
  // Every load whose ADDRESS does not depend on a result is issued before the first barrier: in the step all of these
  // lines are cold (the weights were rewritten by the optimizer), and the three phases used to pay five dependent
  // memory round trips (20 us for a few hundred kFLOP). Fast path: one pass of OPW outputs per wave in both products.
  constexpr bool pre1 = PRE, pre2 = PRE;
  const int c = classes[b];
  float w1[OPW][PRE_C], w2[OPW], b1 = 0.f, bh2 = 0.f, cls2 = 0.f, pos2 = 0.f, eps_r = 0.f;
  const int j1 = wave * OPW;  // this wave's outputs in both products
  if constexpr (pre1) {
#pragma unroll
    for (int u = 0; u < OPW; ++u)
#pragma unroll
      for (int k = 0; k < PRE_C; ++k) {
        const int d = lane + k * 64;
        w1[u][k] = (j1 + u < 2 * Z && d < De) ? Wl[(int64_t)(j1 + u) * De + d] : 0.f;
      }
    if (lane < OPW && j1 + lane < 2 * Z) b1 = bl[j1 + lane];
  }
  if constexpr (pre2) {
#pragma unroll
    for (int u = 0; u < OPW; ++u) w2[u] = (j1 + u < Dd && lane < Z) ? Wh[(int64_t)(j1 + u) * Z + lane] : 0.f;
    if (lane < OPW && j1 + lane < Dd) {
      bh2 = bh[j1 + lane];
      cls2 = cls_d[(int64_t)c * ld_cls + j1 + lane];
      pos2 = pos_d[j1 + lane];
    }
  }
  if (tid < Z) eps_r = eps[b * Z + tid];
  for (int d = tid; d < De; d += LAT_THREADS) h0[d] = to_f32(enc_out[b * enc_stride + d]);
  __syncthreads();
  // one wave per output, lanes across the contraction (coalesced weight rows); OPW outputs at a time so that their
  // weight loads are all in flight together (one output at a time was eight dependent L2 round trips per wave)
  if constexpr (pre1) {
    float acc[OPW];
#pragma unroll
    for (int u = 0; u < OPW; ++u) {
      acc[u] = 0.f;
#pragma unroll
      for (int k = 0; k < PRE_C; ++k)
        if (lane + k * 64 < De) acc[u] = fmaf(h0[lane + k * 64], w1[u][k], acc[u]);  // same order as wave_dots
    }
    {  // (all reductions first, ONE write from lanes 0 .. OPW-1: reduce_emit's note)
      float mine = 0.f;
#pragma unroll
      for (int u = 0; u < OPW; ++u) {
        const float v = wave_sum(acc[u]);
        if (lane == u) mine = v;
      }
      if (lane < OPW && j1 + lane < 2 * Z) lat[j1 + lane] = mine + b1;
    }
  } else {
    if (De <= 64 * PRE_C) wave_dots_pre<OPW, PRE_C>(h0, De, Wl, 2 * Z, wave, NW, lane, [&](int j, float acc) { lat[j] = acc + bl[j]; });
    else wave_dots<OPW>(h0, De, Wl, 2 * Z, wave, NW, lane, [&](int j, float acc) { lat[j] = acc + bl[j]; });
  }
  __syncthreads();
  float klacc = 0.f;
  for (int i = tid; i < Z; i += LAT_THREADS) {
    const float m = lat[i], s = lat[Z + i];
    const float zz = m + (i == tid ? eps_r : eps[b * Z + i]) * s;
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
  if constexpr (pre2) {
    const float zv = lane < Z ? zs[lane] : 0.f;
    float mine = 0.f;
#pragma unroll
    for (int u = 0; u < OPW; ++u) {
      const float v = wave_sum(zv * w2[u]);  // (fmaf(zv, w, 0) of wave_dots)
      if (lane == u) mine = v;
    }
    if (lane < OPW && j1 + lane < Dd) {
      const T o = from_f32<T>(alpha_d * (mine + bh2 + cls2) + pos2);
      dec_in[b * dec_stride + j1 + lane] = o;
      if (proj) x0s[j1 + lane] = to_f32(o);
    }
  } else {
    auto emit2 = [&](int j, float acc) {
      const T o = from_f32<T>(alpha_d * (acc + bh[j] + cls_d[(int64_t)c * ld_cls + j]) + pos_d[j]);
      dec_in[b * dec_stride + j] = o;
      if (proj) x0s[j] = to_f32(o);
    };
    if (Z <= 64 * PRE_C) wave_dots_pre<OPW, PRE_C>(zs, Z, Wh, Dd, wave, NW, lane, emit2);
    else wave_dots<OPW>(zs, Z, Wh, Dd, wave, NW, lane, emit2);
  }
  if (proj) {  // (host: Dd <= 256)
    __syncthreads();
    T* __restrict__ q0 = reinterpret_cast<T*>(A.qkv0) + b * A.qkv_stride;
    auto emitq = [&](int j, float acc) { bqs[j] += acc; };  // (stored below as whole rows, not as 2-byte single-lane stores)
    const T* Wq = reinterpret_cast<const T*>(A.Wq);
    constexpr int UQ = 24;  // (3 Dd / 16 waves at the decoder's width 128)
    if (Dd == 64) wave_dots_t<T, UQ, 1>(x0s, Wq, A.ld_wq, A.nq, wave, NW, lane, emitq);
    else if (Dd == 128) wave_dots_t<T, UQ, 2>(x0s, Wq, A.ld_wq, A.nq, wave, NW, lane, emitq);
    else wave_dots_t<T, UQ, 4>(x0s, Wq, A.ld_wq, A.nq, wave, NW, lane, emitq);  // (host: Dd is 64, 128 or 256)
    __syncthreads();
    for (int j = tid; j < A.nq; j += LAT_THREADS) q0[j] = from_f32<T>(bqs[j]);
  }
}

}  // namespace mst

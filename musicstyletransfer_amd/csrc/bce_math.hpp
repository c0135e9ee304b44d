// bce_math.hpp — one element of sigmoid + BinaryCrossEntropy (loss.py:27-80) and its logit gradient, shared by
// sigmoid_bce_kernel (losses.hip) and the output-layer GEMM's epilogue (gemm_nt.hip: gemm_bce_tile), so the two produce
// the same bits.
//
// bce_exact is the reference's operation order in fp32:
//     p = 1 / (1 + e^-x);  bce = -(s log(1e-12 + p) + (1 - s) log(1e-12 + (1 - p)))                     loss.py:40-48
//     d bce / dx = -(s / (1e-12 + p) - (1 - s) / (1e-12 + (1 - p))) p (1 - p)
// — six transcendental instructions per element; at configs[2] (33.5 M elements) that arithmetic, not HBM, was the loss
// launch (92 us). bce_fast needs three (exp, rcp, log): with q = 1 / (1 + e^-|x|) the LARGER of (p, 1 - p) — so q >= 1/2,
// 1e-12 + q == q in fp32 and log q is as accurate as the reference's —, the smaller is e^-|x| q, its log is log q - |x|,
// and the gradient collapses to p - s (the 1e-12 terms cancel to below one ulp). It agrees with bce_exact to fp32 rounding
// while the 1e-12 epsilons are invisible (smaller probability >= 1e-7: x >= -16) AND while the reference's own
// fl(1 - p) still resolves the smaller probability (x <= 9: beyond that p is within a few ulp of 1 and the reference's
// log(1 - p) is off by 1e-4 and more — which parity keeps); outside [-16, 9] callers take bce_exact. The choice is per
// ELEMENT (bce_fast_domain), so the fused and the unfused launch agree bit for bit whatever their element-to-lane maps;
// callers evaluate bce_exact only in waves that hold an out-of-domain element (a wave-uniform vote), then select.
#pragma once
#include "common.hpp"

namespace mst {

constexpr float BCE_FAST_LO = -16.f, BCE_FAST_HI = 9.f;

__device__ __forceinline__ bool bce_fast_domain(float x) { return x >= BCE_FAST_LO && x <= BCE_FAST_HI; }

// y in {0, 1}; s = (1 - ls) y + ls / 2 (loss.py:34-36); dw: the (w bce) bce form where y == 0 (loss.py:52-54)
template <bool DW>
__device__ __forceinline__ void bce_exact(float x, float y, float ls, float w, float& p, float& bce, float& dbce) {
  p = __frcp_rn(1.f + __expf(-x));  // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE divide
  const float omp = 1.f - p;
  const float s = (1.f - ls) * y + 0.5f * ls;
  const float lp = __logf(1e-12f + p), lq = __logf(1e-12f + omp);
  bce = -(s * lp + (1.f - s) * lq);
  // d bce / d logit = d bce/dp * p(1-p)
  dbce = -(s * __frcp_rn(1e-12f + p) - (1.f - s) * __frcp_rn(1e-12f + omp)) * p * omp;
  if (DW && y == 0.f) {
    dbce = 2.f * w * bce * dbce;
    bce = w * bce * bce;
  }
}

template <bool DW>
__device__ __forceinline__ void bce_fast(float x, float y, float s1 /* s for y = 1 */, float s0 /* s for y = 0 */, float w, float& p,
                                         float& bce, float& dbce) {
  // raw v_exp_f32 / v_rcp_f32 / v_log_f32 (1 ulp): every argument is a normal number inside the fast domain, and hipcc expands
  // __expf / __logf / __frcp_rn into denormal-safe scaling and a ~10-instruction IEEE division
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax);  // e^-|x| in (1e-7, 1]
  const float q = __builtin_amdgcn_rcpf(1.f + t);                     // the larger probability, in [1/2, 1)
  const float lbig = 0.6931471805599453f * __builtin_amdgcn_logf(q), lsmall = lbig - ax;
  const bool pos = x >= 0.f;
  p = pos ? q : t * q;
  const float lp = pos ? lbig : lsmall, lq = pos ? lsmall : lbig;
  const float s = (y != 0.f) ? s1 : s0;
  bce = -(s * lp + (1.f - s) * lq);
  dbce = p - s;
  if (DW && y == 0.f) {
    dbce = 2.f * w * bce * dbce;
    bce = w * bce * bce;
  }
}

}  // namespace mst

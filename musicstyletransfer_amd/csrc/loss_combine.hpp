// loss_combine.hpp — loss = recon + kl_weight * kl per sample and the running metric sums (trainer.py:107-120,172,181-186),
// the body of mst_loss_combine; mst_adam_flat runs it on its first workgroup when handed an mst_step_metrics (one launch
// less per training step: every launch of the captured step costs ~4.7 us however small).
#pragma once
#include "common.hpp"

namespace mst {

// one 256-thread workgroup; red: 8 floats of LDS
__device__ __forceinline__ void loss_combine_wg(int64_t B, const float* __restrict__ recon, const float* __restrict__ kl,
                                                float kl_weight, float* __restrict__ total, float* __restrict__ metric,
                                                float (*red)[4]) {
  float skl = 0.f, stot = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    const float t = recon[b] + kl_weight * kl[b];
    if (total) total[b] = t;
    skl += kl[b];
    stot += t;
  }
  skl = wave_sum(skl);
  stot = wave_sum(stot);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = skl; red[1][threadIdx.x >> 6] = stot; }
  __syncthreads();
  if (threadIdx.x == 0 && metric) {
    metric[0] += red[0][0] + red[0][1] + red[0][2] + red[0][3];
    metric[1] += red[1][0] + red[1][1] + red[1][2] + red[1][3];
    metric[2] += (float)B;
  }
}

// the step guard of mst_step_metrics: true when this step must not count (uniform over the launch: every workgroup reads the
// same words, and the only writer — step_mark_bad — runs when the answer is already `true` for everybody)
__device__ __forceinline__ bool step_is_bad(const mst_step_metrics& mt, bool& incomplete) {
  incomplete = false;
  if (!mt.status) return false;
  if (mt.expect_ptr0 && __hip_atomic_load(mt.expect_ptr0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != mt.expect_val0) incomplete = true;
  if (mt.expect_ptr1 && __hip_atomic_load(mt.expect_ptr1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != mt.expect_val1) incomplete = true;
  return incomplete || __hip_atomic_load(mt.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
// the non-finite guard of mst_step_metrics: true (for every thread of every workgroup alike) when one of the step's per-sample
// losses is not finite. Called by all threads of a 256-thread workgroup (workgroup barrier inside).
__device__ __forceinline__ bool step_loss_nonfinite(const mst_step_metrics& mt) {
  if (!mt.fin_recon || !mt.status) return false;
  int ok = 1;
  for (int64_t b = threadIdx.x; b < mt.fin_B; b += blockDim.x) {
    const float t = mt.fin_recon[b] + (mt.fin_kl ? mt.fin_kl[b] : 0.f);
    ok &= (fabsf(t) <= 3.0e38f) ? 1 : 0;  // (false for inf and for NaN)
  }
  return !__syncthreads_and(ok);
}
// one thread, once per step
__device__ __forceinline__ void step_mark_bad(const mst_step_metrics& mt, bool incomplete) {
  if (incomplete) __hip_atomic_fetch_or(mt.status, MST_STEP_INCOMPLETE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(mt.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace mst

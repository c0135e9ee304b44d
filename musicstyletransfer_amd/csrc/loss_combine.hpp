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

}  // namespace mst

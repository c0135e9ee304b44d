// outer_jobs.hpp — deferred batch outer products (mst_outer_job): out[j, i] += sum_b L[b, j] * R[b, i] and obias[j] += sum_b L[b, j],
// the parameter gradients of a layer that sees ONE row per sample (the latent block: latent_proj and latent2hid, model.py:97-103,
// 229-232). Nothing downstream but the optimizer reads them, so they ride as extra workgroups on the weight-gradient reduction
// pass (gemm_wgrad.hip) — or on one launch of their own (latent.hip) when there is no reduction pass.
// A 256-thread workgroup owns 64 outputs; its four waves take a quarter of the batch each (8 rows in flight per thread: the strided
// rows of R are cold lines and the loop is one memory round trip per group — one thread per output walking the whole batch was
// 11 us) and the quarters are added in order through LDS (deterministic).
#pragma once
#include "common.hpp"

namespace mst {

constexpr int OJ_MAXJ = 2;
struct OuterBatch {
  int n;
  int wg_prefix[OJ_MAXJ + 1];  // workgroups of job j are [wg_prefix[j], wg_prefix[j+1])
  mst_outer_job j[OJ_MAXJ];
};

template <typename RT>
__device__ __forceinline__ void batch_outer(int64_t blk, int tid, int64_t B, int J, int I, const float* __restrict__ L,
                                            const RT* __restrict__ R, int64_t r_stride, float* __restrict__ out,
                                            float* __restrict__ obias, float (*red)[64]) {
  const int o = tid & 63, part = tid >> 6;
  const int64_t idx = blk * 64 + o;
  const int64_t per = (B + 3) / 4, b0 = part * per, b1 = b0 + per < B ? b0 + per : B;
  float acc = 0.f, accb = 0.f;
  if (idx < (int64_t)J * I) {
    const int j = (int)((uint32_t)idx / (uint32_t)I), i = (int)((uint32_t)idx - (uint32_t)j * (uint32_t)I);  // (J * I < 2^31: host check)
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    int64_t b = b0;
    for (; b + 8 <= b1; b += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = fmaf(L[(b + u) * J + j], to_f32(R[(b + u) * r_stride + i]), a[u]);
    }
    for (; b < b1; ++b) a[0] = fmaf(L[b * J + j], to_f32(R[b * r_stride + i]), a[0]);
    acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  if (obias && idx < J) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int64_t b = b0;
    for (; b + 4 <= b1; b += 4) {
      a0 += L[(b + 0) * J + idx]; a1 += L[(b + 1) * J + idx]; a2 += L[(b + 2) * J + idx]; a3 += L[(b + 3) * J + idx];
    }
    for (; b < b1; ++b) a0 += L[b * J + idx];
    accb = (a0 + a1) + (a2 + a3);
  }
  red[part][o] = acc;
  red[4 + part][o] = accb;
  __syncthreads();
  if (part == 0) {
    if (idx < (int64_t)J * I) out[idx] += (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
    if (obias && idx < J) obias[idx] += (red[4][o] + red[5][o]) + (red[6][o] + red[7][o]);
  }
}

// red: 8 x 64 floats of LDS; wg: workgroup index within the batch (callers guarantee wg < wg_prefix[n]); 256 threads
__device__ __forceinline__ void outer_jobs_wg(const OuterBatch& b, int wg, float (*red)[64]) {
  const int ji = (b.n > 1 && wg >= b.wg_prefix[1]) ? 1 : 0;
  const mst_outer_job& q = b.j[ji];
  const int64_t blk = wg - b.wg_prefix[ji];
  if (q.r_dtype == MST_F32) batch_outer<float>(blk, threadIdx.x, q.B, (int)q.J, (int)q.I, q.L, (const float*)q.R, q.r_stride, q.out, q.obias, red);
  else if (q.r_dtype == MST_BF16) batch_outer<__bf16>(blk, threadIdx.x, q.B, (int)q.J, (int)q.I, q.L, (const __bf16*)q.R, q.r_stride, q.out, q.obias, red);
  else batch_outer<_Float16>(blk, threadIdx.x, q.B, (int)q.J, (int)q.I, q.L, (const _Float16*)q.R, q.r_stride, q.out, q.obias, red);
}

// host: validate and pack the jobs; returns MST_OK or a status with mst_last_error() set
static inline int pack_outer_jobs(const mst_outer_job* jobs, int n, OuterBatch& b) {
  MST_CHECK_ARG(n >= 0 && n <= OJ_MAXJ && (n == 0 || jobs != nullptr), "mst_outer_jobs: 0..%d jobs per launch (got %d)", OJ_MAXJ, n);
  b.n = n;
  b.wg_prefix[0] = 0;
  for (int i = 0; i < OJ_MAXJ; ++i) {
    if (i < n) {
      const mst_outer_job& q = jobs[i];
      MST_CHECK_ARG(q.L && q.R && q.out && q.B > 0 && q.J > 0 && q.I > 0, "mst_outer_jobs: job %d: null pointer or empty", i);
      MST_CHECK_ARG(q.J * q.I < (1ll << 31), "mst_outer_jobs: job %d: more than 2^31 outputs", i);
      MST_CHECK_ARG(q.r_dtype == MST_F32 || q.r_dtype == MST_BF16 || q.r_dtype == MST_F16, "mst_outer_jobs: job %d: bad r_dtype", i);
      MST_CHECK_ARG(q.r_stride >= q.I, "mst_outer_jobs: job %d: r_stride < I", i);
      b.j[i] = q;
      b.wg_prefix[i + 1] = b.wg_prefix[i] + (int)cdiv(q.J * q.I, 64);
    } else {
      b.j[i] = mst_outer_job{};
      b.wg_prefix[i + 1] = b.wg_prefix[i];
    }
  }
  return MST_OK;
}

}  // namespace mst

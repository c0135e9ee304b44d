// partial_sums.hpp — deferred column sums (mst_partial_sums): dst[c] += scale * sum_p src[p*stride + c], parts in index
// order. One 256-thread workgroup per 64 columns of a job: 16 part-groups x 16 float4 lanes; every thread adds its parts in
// index order and the 16 groups are combined in group order, so the summation tree is fixed (deterministic).
// Launched on its own (util.hip) or as extra workgroups of the weight-gradient reduction pass (gemm_wgrad.hip).
#pragma once
#include "common.hpp"

namespace mst {

constexpr int PS_MAXJ = 20;  // (the reduction pass of gemm_wgrad.hip passes this and the outer-product jobs next to its own 2.8 KB of arguments)
struct PartialSumBatch {
  int n;
  int wg_prefix[PS_MAXJ + 1];  // workgroups of job j are [wg_prefix[j], wg_prefix[j+1])
  mst_partial_sum j[PS_MAXJ];
};

// red: 16 x 16 float4 of LDS; wg: workgroup index within the batch (callers guarantee wg < wg_prefix[n]); 256 threads
__device__ __forceinline__ void partial_sums_wg(const PartialSumBatch& b, int wg, f32x4 (*red)[16]) {
  int ji = 0;
#pragma unroll
  for (int i = 1; i < PS_MAXJ; ++i)
    if (i < b.n && wg >= b.wg_prefix[i]) ji = i;
  const mst_partial_sum& job = b.j[ji];
  const int c4 = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const int64_t col = (int64_t)(wg - b.wg_prefix[ji]) * 64 + c4 * 4;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  if (col < job.len) {
    const float* src = job.src + col;
#pragma unroll 4
    for (int64_t p = pg; p < job.n_parts; p += 16) sum += *reinterpret_cast<const f32x4*>(src + p * job.stride);
  }
  red[pg][c4] = sum;
  __syncthreads();
  if (pg == 0 && col < job.len) {
    f32x4 t = red[0][c4];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += red[g][c4];
    f32x4* d = reinterpret_cast<f32x4*>(job.dst + col);
    *d = *d + t * job.scale;
  }
}

// host: validate and pack the jobs; returns MST_OK or a status with mst_last_error() set
static inline int pack_partial_sums(const mst_partial_sum* jobs, int n, PartialSumBatch& b) {
  MST_CHECK_ARG(jobs != nullptr && n > 0 && n <= PS_MAXJ, "mst_partial_sums: 1..%d jobs per launch (got %d)", PS_MAXJ, n);
  b.n = n;
  b.wg_prefix[0] = 0;
  for (int i = 0; i < n; ++i) {
    const mst_partial_sum& j = jobs[i];
    MST_CHECK_ARG(j.src && j.dst && j.n_parts > 0 && j.len > 0, "mst_partial_sums: job %d: null pointer or empty", i);
    MST_CHECK_ARG(j.len % 4 == 0 && j.stride % 4 == 0 && j.stride >= j.len,
                  "mst_partial_sums: job %d: len and stride must be multiples of 4, stride >= len", i);
    MST_CHECK_ARG((uintptr_t)j.src % 16 == 0 && (uintptr_t)j.dst % 16 == 0, "mst_partial_sums: job %d: src and dst must be 16-byte aligned", i);
    b.j[i] = j;
    b.wg_prefix[i + 1] = b.wg_prefix[i] + (int)cdiv(j.len, 64);
  }
  for (int i = n; i < PS_MAXJ; ++i) {
    b.wg_prefix[i + 1] = b.wg_prefix[n];
    b.j[i] = mst_partial_sum{nullptr, 0, 0, 0, nullptr, 0.f};
  }
  return MST_OK;
}

}  // namespace mst

// util.hip — small stream-ordered utilities that keep the whole training step graph-replayable:
//   mst_zero        : hipMemsetAsync wrapper (gradient bucket, metric sums)
//   mst_rng_advance : per-step device-resident RNG seed (dropout masks and eps change on every replay
//                     of a captured graph because kernels read the seed from device memory)
//   mst_randn       : eps ~ N(0,1) for the reparameterisation (replaces mx.nd.random_normal,
//                     VarAutoEncoder/model.py:292); Box-Muller over the counter hash of common.hpp
#include <math.h>
#include "common.hpp"
#include "partial_sums.hpp"
#include "step_begin.hpp"

namespace mst {

// state[0] = seed used by this step's kernels, state[1] = step counter, state[2] = base seed
// (state[3] = arrival counter of mst_step_begin, zero between launches)
__global__ void rng_advance_kernel(uint64_t* state) {
  const uint64_t step = state[1] + 1;
  state[1] = step;
  uint64_t x = state[2] ^ (step * 0x9E3779B97F4A7C15ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  state[0] = x;
}

__global__ __launch_bounds__(256) void randn_kernel(int64_t n, float* __restrict__ out, uint64_t seed,
                                                    const uint64_t* __restrict__ seed_ptr, uint32_t site) {
  const uint64_t s = seed ^ (seed_ptr ? seed_ptr[0] : 0ull);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n + 1) / 2; i += (int64_t)gridDim.x * 256) {
    const uint32_t a = dropout_hash(s, site, (uint64_t)(2 * i));
    const uint32_t b = dropout_hash(s, site, (uint64_t)(2 * i + 1));
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);           // [0, 1)
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    out[2 * i] = r * cs;
    if (2 * i + 1 < n) out[2 * i + 1] = r * sn;
  }
}

__global__ __launch_bounds__(SB_THREADS) void step_begin_kernel(StepBegin q) { step_begin_wg<SB_THREADS>(q, (int)blockIdx.x, (int)gridDim.x); }

__global__ __launch_bounds__(256) void partial_sums_kernel(PartialSumBatch b) {
  __shared__ f32x4 red[16][16];
  partial_sums_wg(b, (int)blockIdx.x, red);
}

}  // namespace mst

using namespace mst;

extern "C" int mst_step_begin_v(const mst_step_begin_args* args, mst_stream_t stream) {
  MST_CHECK_ARG(args != nullptr, "mst_step_begin: null args");
  StepBegin q;
  int64_t grid = 0;
  int rc = pack_step_begin(*args, q, &grid);
  if (rc) return rc;
  hipLaunchKernelGGL(step_begin_kernel, dim3((unsigned)grid), dim3(SB_THREADS), 0, (hipStream_t)stream, q);
  MST_CHECK_LAUNCH("step_begin_kernel");
  // (as a launch of its own the bookkeeping hosts nothing: a requested shadow refresh is the separate launch it always was)
  if (args->sh_w)
    return mst_transpose_shadows(args->sh_dtype, args->sh_w, args->sh_wt16, args->sh_desc, args->sh_prefix, args->sh_n_mat, args->sh_tiles, stream);
  return MST_OK;
}

extern "C" int mst_step_begin(uint64_t* rng_state, int32_t* adam_state, double lr, double beta1, double beta2, float* eps_out,
                              int64_t n_eps, uint32_t eps_site, int64_t eps_index0, const int32_t* lens, int64_t B, uint8_t* mask_e, int64_t Se,
                              int32_t add_e, uint8_t* mask_d, int64_t Sd, int32_t add_d, void* zero_a, int64_t zero_a_bytes,
                              void* zero_b, int64_t zero_b_bytes, mst_stream_t stream) {
  const mst_step_begin_args a = {rng_state, adam_state, lr, beta1, beta2, eps_out, n_eps, eps_site, eps_index0, lens, B, mask_e, Se,
                                 add_e, mask_d, Sd, add_d, zero_a, zero_a_bytes, zero_b, zero_b_bytes, 0, nullptr, nullptr, nullptr, nullptr, 0, 0};
  return mst_step_begin_v(&a, stream);
}

extern "C" int mst_partial_sums(const mst_partial_sum* jobs, int n, mst_stream_t stream) {
  PartialSumBatch b;
  int rc = pack_partial_sums(jobs, n, b);
  if (rc) return rc;
  hipLaunchKernelGGL(partial_sums_kernel, dim3((unsigned)b.wg_prefix[n]), dim3(256), 0, (hipStream_t)stream, b);
  MST_CHECK_LAUNCH("partial_sums_kernel");
  return MST_OK;
}

extern "C" int mst_zero(void* ptr, int64_t bytes, mst_stream_t stream) {
  MST_CHECK_ARG(ptr != nullptr && bytes > 0, "mst_zero: bad argument");
  hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream);
  if (e != hipSuccess) {
    set_error("mst_zero: %s", hipGetErrorString(e));
    return MST_ERR_LAUNCH;
  }
  return MST_OK;
}

extern "C" int mst_rng_advance(uint64_t* state, mst_stream_t stream) {
  MST_CHECK_ARG(state != nullptr, "mst_rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  MST_CHECK_LAUNCH("rng_advance_kernel");
  return MST_OK;
}

extern "C" int mst_randn(int64_t n, float* out, uint64_t seed, const uint64_t* seed_ptr, uint32_t site,
                         mst_stream_t stream) {
  MST_CHECK_ARG(n > 0 && out != nullptr, "mst_randn: bad argument");
  int64_t g = cdiv((n + 1) / 2, 256);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, n, out, seed, seed_ptr, site);
  MST_CHECK_LAUNCH("randn_kernel");
  return MST_OK;
}

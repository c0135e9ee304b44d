// util.hip — small stream-ordered utilities that keep the whole training step graph-replayable:
//   mst_zero        : hipMemsetAsync wrapper (gradient bucket, metric sums)
//   mst_rng_advance : per-step device-resident RNG seed (dropout masks and eps change on every replay
//                     of a captured graph because kernels read the seed from device memory)
//   mst_randn       : eps ~ N(0,1) for the reparameterisation (replaces mx.nd.random_normal,
//                     VarAutoEncoder/model.py:292); Box-Muller over the counter hash of common.hpp
#include <math.h>
#include "common.hpp"
#include "partial_sums.hpp"

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

// One launch at the top of every step (each kernel in the captured graph costs ~4.7 us however small):
// advances the RNG state, advances Adam's step counter and bias-corrected learning rate, draws eps, and writes the
// two padding masks from the sequence lengths (SequenceMask, model.py:246-247; the encoder's for the piano-roll ends).
constexpr int SB_THREADS = 1024;
__device__ __forceinline__ uint64_t step_seed(uint64_t base, uint64_t step) {
  uint64_t x = base ^ (step * 0x9E3779B97F4A7C15ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

// Grid of several workgroups (the single-workgroup form took 29 us of the step). Every workgroup derives the new
// seed itself from (base seed, step counter + 1); the state is written back by the workgroup that ARRIVES LAST at
// rng_state[3], i.e. after every other workgroup has read the old counter.
__global__ __launch_bounds__(SB_THREADS) void step_begin_kernel(uint64_t* rng_state, int32_t* adam_state, double lr, double beta1,
                                                         double beta2, float* eps_out, int64_t n_eps, uint32_t eps_site, int64_t eps_index0,
                                                         const int32_t* lens, int64_t B, uint8_t* mask_e, int64_t Se,
                                                         int32_t add_e, uint8_t* mask_d, int64_t Sd, int32_t add_d,
                                                         u32x4* zero_a, int64_t n16_a, u32x4* zero_b, int64_t n16_b, int n_state) {
  // workgroups [0, n_state) do the bookkeeping (and take part in the arrival count); the rest of the grid only helps
  // clearing the two buffers — the 7.5 MB gradient bucket is most of this launch's bytes and has no business waiting on
  // 64 workgroups' worth of store bandwidth
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  {
    const int64_t zid = (int64_t)blockIdx.x * SB_THREADS + threadIdx.x, zsz = (int64_t)gridDim.x * SB_THREADS;
    for (int64_t i = zid; i < n16_a; i += zsz) zero_a[i] = z4;
    for (int64_t i = zid; i < n16_b; i += zsz) zero_b[i] = z4;
  }
  if ((int)blockIdx.x >= n_state) return;
  const int64_t gid = (int64_t)blockIdx.x * SB_THREADS + threadIdx.x, gsz = (int64_t)n_state * SB_THREADS;
  uint64_t step = 0, s = 0;
  if (rng_state) {
    step = rng_state[1] + 1;
    s = step_seed(rng_state[2], step);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && adam_state) {  // nobody else in this launch touches the Adam state
    const int t = adam_state[0] + 1;
    adam_state[0] = t;
    const double c1 = 1.0 - pow(beta1, (double)t), c2 = 1.0 - pow(beta2, (double)t);  // double, like the reference
    reinterpret_cast<float*>(adam_state)[1] = (float)(lr * sqrt(c2) / c1);
  }
  if (eps_out) {
    for (int64_t i = gid; i < (n_eps + 1) / 2; i += gsz) {
      const uint32_t a = dropout_hash(s, eps_site, (uint64_t)(eps_index0 + 2 * i));
      const uint32_t b = dropout_hash(s, eps_site, (uint64_t)(eps_index0 + 2 * i + 1));
      const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
      const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
      const float r = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      eps_out[2 * i] = r * cs;
      if (2 * i + 1 < n_eps) eps_out[2 * i + 1] = r * sn;
    }
  }
  // (32-bit quotients: B * S < 2^31 is checked on the host; each thread handles about one element of each mask, and the
  // 64-bit division it used to start with was several hundred instructions)
  if (mask_e)
    for (int64_t i = gid; i < B * Se; i += gsz) {
      const uint32_t q = (uint32_t)i / (uint32_t)Se, r = (uint32_t)i - q * (uint32_t)Se;
      mask_e[i] = ((int64_t)r < (int64_t)lens[q] + add_e) ? 1 : 0;
    }
  if (mask_d)
    for (int64_t i = gid; i < B * Sd; i += gsz) {
      const uint32_t q = (uint32_t)i / (uint32_t)Sd, r = (uint32_t)i - q * (uint32_t)Sd;
      mask_d[i] = ((int64_t)r < (int64_t)lens[q] + add_d) ? 1 : 0;
    }
  if (rng_state) {
    __syncthreads();  // every thread of this workgroup has read the old counter
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned long long arrived = atomicAdd(reinterpret_cast<unsigned long long*>(rng_state + 3), 1ull);
      if (arrived == (unsigned long long)n_state - 1) {
        rng_state[3] = 0;
        rng_state[1] = step;
        rng_state[0] = s;
      }
    }
  }
}

__global__ __launch_bounds__(256) void partial_sums_kernel(PartialSumBatch b) {
  __shared__ f32x4 red[16][16];
  partial_sums_wg(b, (int)blockIdx.x, red);
}

}  // namespace mst

using namespace mst;

extern "C" int mst_step_begin(uint64_t* rng_state, int32_t* adam_state, double lr, double beta1, double beta2, float* eps_out,
                              int64_t n_eps, uint32_t eps_site, int64_t eps_index0, const int32_t* lens, int64_t B, uint8_t* mask_e, int64_t Se,
                              int32_t add_e, uint8_t* mask_d, int64_t Sd, int32_t add_d, void* zero_a, int64_t zero_a_bytes,
                              void* zero_b, int64_t zero_b_bytes, mst_stream_t stream) {
  MST_CHECK_ARG((!zero_a || ((uintptr_t)zero_a % 16 == 0 && zero_a_bytes % 16 == 0)) &&
                    (!zero_b || ((uintptr_t)zero_b % 16 == 0 && zero_b_bytes % 16 == 0)),
                "mst_step_begin: zero buffers must be 16-byte aligned with sizes that are multiples of 16");
  MST_CHECK_ARG(!eps_out || (rng_state && n_eps > 0), "mst_step_begin: eps needs the rng state");
  MST_CHECK_ARG(eps_index0 >= 0 && eps_index0 % 2 == 0, "mst_step_begin: eps_index0 must be even (Box-Muller pairs)");
  MST_CHECK_ARG((!mask_e && !mask_d) || (lens && B > 0), "mst_step_begin: masks need the lengths");
  MST_CHECK_ARG((!mask_e || (Se > 0 && B * Se < (1ll << 31))) && (!mask_d || (Sd > 0 && B * Sd < (1ll << 31))), "mst_step_begin: B * S must stay below 2^31");
  int64_t work = n_eps / 2;
  if (mask_e && B * Se > work) work = B * Se;
  if (mask_d && B * Sd > work) work = B * Sd;
  const int64_t n16_a = zero_a ? zero_a_bytes / 16 : 0, n16_b = zero_b ? zero_b_bytes / 16 : 0;
  if (n16_a > work) work = n16_a;
  if (n16_b > work) work = n16_b;
  // few, fat workgroups: every workgroup ends with one atomic on the SAME arrival counter, and same-address atomics
  // are serialised at ~40 ns each (512 workgroups measured 20 us for this launch)
  int64_t n_state = cdiv(work > 0 ? work : 1, SB_THREADS * 4);
  if (n_state > 64) n_state = 64;
  // (+ workgroups that only clear: 16 KiB of the zero lists each, one per CU at most)
  int64_t grid = cdiv(n16_a + n16_b, SB_THREADS);
  if (grid > 256) grid = 256;
  if (grid < n_state) grid = n_state;
  hipLaunchKernelGGL(step_begin_kernel, dim3((unsigned)grid), dim3(SB_THREADS), 0, (hipStream_t)stream, rng_state, adam_state, lr, beta1, beta2,
                     eps_out, n_eps, eps_site, eps_index0, lens, B, mask_e, Se, add_e, mask_d, Sd, add_d, (u32x4*)zero_a, n16_a, (u32x4*)zero_b,
                     n16_b, (int)n_state);
  MST_CHECK_LAUNCH("step_begin_kernel");
  return MST_OK;
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

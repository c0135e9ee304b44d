// step_begin.hpp — the bookkeeping at the top of every training step (mst_step_begin): advances the RNG state, advances Adam's step
// counter and bias-corrected learning rate, draws eps, writes the two padding masks from the sequence lengths (SequenceMask,
// model.py:246-247; the encoder's for the piano-roll ends) and clears the loss sums and the gradient bucket. A launch of its own
// (util.hip) or extra workgroups of the step's first GEMM launch (mst_gemm_nt_pair_begin, gemm_nt.hip): every kernel in the captured
// graph costs ~4.7 us however small, and nothing in the embedding GEMMs depends on this.
#pragma once
#include <math.h>
#include "common.hpp"

namespace mst {

constexpr int SB_THREADS = 1024;  // threads of the stand-alone launch
__device__ __forceinline__ uint64_t step_seed(uint64_t base, uint64_t step) {
  uint64_t x = base ^ (step * 0x9E3779B97F4A7C15ull);
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

struct StepBegin {  // mst_step_begin_args with the zero lists in 16-byte units and the number of bookkeeping workgroups
  uint64_t* rng_state; int32_t* adam_state; double lr, beta1, beta2;
  float* eps_out; int64_t n_eps; uint32_t eps_site; int64_t eps_index0;
  const int32_t* lens; int64_t B; uint8_t* mask_e; int64_t Se; int32_t add_e; uint8_t* mask_d; int64_t Sd; int32_t add_d;
  u32x4* zero_a; int64_t n16_a; u32x4* zero_b; int64_t n16_b; int n_state;
  // optional transposed-shadow refresh hosted by the same launch (mst_step_begin_args.sh_*; sh_w == nullptr: none)
  const float* sh_w; void* sh_wt16; const int64_t* sh_desc; const int64_t* sh_prefix; int sh_n_mat; int64_t sh_tiles;
};

// Workgroup `wg` of `nwg` (NT threads each). Every workgroup derives the new seed itself from (base seed, step counter + 1); the
// state is written back by the workgroup that ARRIVES LAST at rng_state[3], i.e. after every other workgroup has read the old
// counter. Workgroups [0, n_state) do the bookkeeping (and take part in the arrival count); the rest only help clearing the two
// buffers — the gradient bucket is most of the bytes and has no business waiting on 64 workgroups' worth of store bandwidth.
template <int NT>
__device__ __forceinline__ void step_begin_wg(const StepBegin& q, int wg, int nwg) {
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  {
    const int64_t zid = (int64_t)wg * NT + threadIdx.x, zsz = (int64_t)nwg * NT;
    for (int64_t i = zid; i < q.n16_a; i += zsz) q.zero_a[i] = z4;
    for (int64_t i = zid; i < q.n16_b; i += zsz) q.zero_b[i] = z4;
  }
  if (wg >= q.n_state) return;
  const int64_t gid = (int64_t)wg * NT + threadIdx.x, gsz = (int64_t)q.n_state * NT;
  uint64_t step = 0, s = 0;
  if (q.rng_state) {
    step = q.rng_state[1] + 1;
    s = step_seed(q.rng_state[2], step);
  }
  if (wg == 0 && threadIdx.x == 0 && q.adam_state) {  // nobody else in this launch touches the Adam state
    const int t = q.adam_state[0] + 1;
    q.adam_state[0] = t;
    const double c1 = 1.0 - pow(q.beta1, (double)t), c2 = 1.0 - pow(q.beta2, (double)t);  // double, like the reference
    reinterpret_cast<float*>(q.adam_state)[1] = (float)(q.lr * sqrt(c2) / c1);
  }
  if (q.eps_out) {
    for (int64_t i = gid; i < (q.n_eps + 1) / 2; i += gsz) {
      const uint32_t a = dropout_hash(s, q.eps_site, (uint64_t)(q.eps_index0 + 2 * i));
      const uint32_t b = dropout_hash(s, q.eps_site, (uint64_t)(q.eps_index0 + 2 * i + 1));
      const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
      const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
      const float r = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      q.eps_out[2 * i] = r * cs;
      if (2 * i + 1 < q.n_eps) q.eps_out[2 * i + 1] = r * sn;
    }
  }
  // (32-bit quotients: B * S < 2^31 is checked on the host; each thread handles about one element of each mask, and the
  // 64-bit division it used to start with was several hundred instructions)
  if (q.mask_e)
    for (int64_t i = gid; i < q.B * q.Se; i += gsz) {
      const uint32_t b = (uint32_t)i / (uint32_t)q.Se, r = (uint32_t)i - b * (uint32_t)q.Se;
      q.mask_e[i] = ((int64_t)r < (int64_t)q.lens[b] + q.add_e) ? 1 : 0;
    }
  if (q.mask_d)
    for (int64_t i = gid; i < q.B * q.Sd; i += gsz) {
      const uint32_t b = (uint32_t)i / (uint32_t)q.Sd, r = (uint32_t)i - b * (uint32_t)q.Sd;
      q.mask_d[i] = ((int64_t)r < (int64_t)q.lens[b] + q.add_d) ? 1 : 0;
    }
  if (q.rng_state) {
    __syncthreads();  // every thread of this workgroup has read the old counter
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned long long arrived = atomicAdd(reinterpret_cast<unsigned long long*>(q.rng_state + 3), 1ull);
      if (arrived == (unsigned long long)q.n_state - 1) {
        q.rng_state[3] = 0;
        q.rng_state[1] = step;
        q.rng_state[0] = s;
      }
    }
  }
}

// host: validate mst_step_begin_args and pack them; *grid = workgroups of a stand-alone launch (callers hosting the work in another
// launch use min(*grid, what fits)). Returns MST_OK or a status with mst_last_error() set.
static inline int pack_step_begin(const mst_step_begin_args& a, StepBegin& q, int64_t* grid) {
  MST_CHECK_ARG((!a.zero_a || ((uintptr_t)a.zero_a % 16 == 0 && a.zero_a_bytes % 16 == 0)) &&
                    (!a.zero_b || ((uintptr_t)a.zero_b % 16 == 0 && a.zero_b_bytes % 16 == 0)),
                "mst_step_begin: zero buffers must be 16-byte aligned with sizes that are multiples of 16");
  MST_CHECK_ARG(!a.eps_out || (a.rng_state && a.n_eps > 0), "mst_step_begin: eps needs the rng state");
  MST_CHECK_ARG(a.eps_index0 >= 0 && a.eps_index0 % 2 == 0, "mst_step_begin: eps_index0 must be even (Box-Muller pairs)");
  MST_CHECK_ARG((!a.mask_e && !a.mask_d) || (a.lens && a.B > 0), "mst_step_begin: masks need the lengths");
  MST_CHECK_ARG((!a.mask_e || (a.Se > 0 && a.B * a.Se < (1ll << 31))) && (!a.mask_d || (a.Sd > 0 && a.B * a.Sd < (1ll << 31))),
                "mst_step_begin: B * S must stay below 2^31");
  int64_t work = a.n_eps / 2;
  if (a.mask_e && a.B * a.Se > work) work = a.B * a.Se;
  if (a.mask_d && a.B * a.Sd > work) work = a.B * a.Sd;
  const int64_t n16_a = a.zero_a ? a.zero_a_bytes / 16 : 0, n16_b = a.zero_b ? a.zero_b_bytes / 16 : 0;
  if (n16_a > work) work = n16_a;
  if (n16_b > work) work = n16_b;
  // few, fat workgroups: every workgroup ends with one atomic on the SAME arrival counter, and same-address atomics
  // are serialised at ~40 ns each (512 workgroups measured 20 us for this launch)
  int64_t n_state = cdiv(work > 0 ? work : 1, SB_THREADS * 4);
  if (n_state > 64) n_state = 64;
  // (+ workgroups that only clear: 16 KiB of the zero lists each, one per CU at most)
  int64_t g = cdiv(n16_a + n16_b, SB_THREADS);
  if (g > 256) g = 256;
  if (g < n_state) g = n_state;
  *grid = g;
  q = StepBegin{a.rng_state, a.adam_state, a.lr, a.beta1, a.beta2, a.eps_out, a.n_eps, a.eps_site, a.eps_index0, a.lens, a.B,
                a.mask_e, a.Se, a.add_e, a.mask_d, a.Sd, a.add_d, (u32x4*)a.zero_a, n16_a, (u32x4*)a.zero_b, n16_b, (int)n_state,
                nullptr, nullptr, nullptr, nullptr, 0, 0};
  if (a.sh_w) {
    MST_CHECK_ARG(a.sh_wt16 && a.sh_desc && a.sh_prefix && a.sh_n_mat > 0 && a.sh_tiles > 0 && a.sh_tiles < (1ll << 30),
                  "mst_step_begin: the shadow refresh needs w, wt16, the matrix table and its tile prefix sums");
    q.sh_w = a.sh_w; q.sh_wt16 = a.sh_wt16; q.sh_desc = a.sh_desc; q.sh_prefix = a.sh_prefix; q.sh_n_mat = (int)a.sh_n_mat; q.sh_tiles = a.sh_tiles;
  }
  return MST_OK;
}

}  // namespace mst

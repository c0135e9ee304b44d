// row_tail.hip — the position-0 tail of the top encoder layer in ONE launch.
//
// The model reads the encoder at position 0 only (VarAutoEncoder/model.py:97), so after the attention mix the top layer's
//     h1 = x_in + dropout(att W_proj^T + b)        x1 = LN1(h1)                         (transformer.py:154-155)
//     a  = dropout(relu(x1 W1^T + b1))             h2 = x1 + dropout(a W2^T + b2)       (transformer.py:42-46,157)
//     x2 = LN2(h2)                                                                        (transformer.py:158)
// run on B rows (one per sample): microseconds of arithmetic that used to be five launches of ~9 us each, every one a
// dependent launch boundary plus a cold fetch of its weights. Here the chain is one launch of G = D / 16 workgroups that
// each own a slice of every GEMM's OUTPUT columns (so every workgroup streams 1/G of every weight matrix, once) and meet
// at three grid-wide barriers where the chain needs whole rows: after W_proj (LayerNorm 1 needs the row), after FFN1 (FFN2
// contracts over the whole hidden row), after FFN2 (LayerNorm 2). Rows travel between the stages through HBM/L2 (they
// are the tensors the backward pass reads anyway); the LayerNorms are computed where they are consumed. 64 x 256 x
// (256 + 1024 + 1024) MACs spread over 16 CUs is ~1 us of MFMA; the launch is three barrier latencies plus four L2 round
// trips. Arithmetic, rounding points and dropout counters are those of mst_gemm_nt / mst_layernorm_fwd on the same rows
// (counter = physical row * N + column), so the backward pass regenerates the same masks.
//
// Inter-workgroup visibility WITHOUT fences (a release is a write-back of the XCD's L2, an acquire an invalidate of the CU's
// L1: the fenced form of this kernel spent ~8 us per barrier, 38 us in all — the five launches took 43): every byte
// that crosses a barrier (h1, a, h2) is stored write-through (`sc1`: relaxed agent-scope atomic stores of 8 bytes) and
// loaded with `sc1` loads (relaxed agent-scope atomic loads), which bypass the non-coherent levels; every storing wave
// drains its stores (s_waitcnt vmcnt(0)) before the workgroup barrier, then ONE lane adds to the counter and polls it with
// relaxed agent loads (bounded), and the workgroup barriers again before anyone loads. Weights, att and the residual come
// from earlier launches and are read with plain loads. The G workgroups are co-resident by construction (G <= 16 << 256).
#include <math.h>
#include "common.hpp"
#include "gemm_tile.hpp"
#include "shadows.hpp"

namespace mst {

// Status bits left in the caller's sticky device word (mst_row_tail_*_args.status, optional) when a launch could not do its work:
//   1 / 2   a grid barrier of the forward / backward kernel gave up waiting (fewer than G workgroups of the launch were placed
//           on the claimed XCD — CU masks, another partition mode, a co-tenant holding the CUs — so the rows behind it are stale)
// A launch in which NOBODY got a role (sync words not zeroed) cannot flag itself; it leaves its barrier counter short, which is
// what the step guard of the optimizer launch checks (mst_step_metrics.expect_ptr: MST_STEP_INCOMPLETE).
// The optimizer launch of the same step reads the word and leaves parameters, moments and the step count untouched when it is
// set (mst_step_metrics.status), and the host switches to the five-launch form: a failed tail costs skipped batches, never a
// silently wrong update.
// (MST_TAIL_SPIN_FWD / _BWD of include/mst_hip.h)
#ifndef MST_TAIL_SPIN_TICKS
#define MST_TAIL_SPIN_TICKS 20000000ull /* 0.2 s of the 100 MHz s_memrealtime clock; a healthy barrier waits microseconds */
#endif

__device__ __forceinline__ void grid_sync(uint32_t* ctr, uint32_t target, uint32_t* status, uint32_t spin_bit) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    // (agent-scope counter operations even though the participants share an XCD: an atomic add without scope bits + sc0 polling
    // loads was tried and is NOT reliable — most launches sat out the spin bound)
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0u) {  // (a healthy barrier waits microseconds and never gets here)
        // a run that is ALREADY flagged (this launch's earlier barrier, or a step before it that the host has not looked at
        // yet) does not sit out the bound again at every barrier of every following step: its result is discarded anyway
        if (status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > MST_TAIL_SPIN_TICKS) {
          // never in a correct launch; a bound instead of a hung GPU — and a flag instead of a silently wrong step
          if (status) __hip_atomic_fetch_or(status, spin_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  }
  __syncthreads();
}

// write-through 8-byte store / L2-served loads of the bytes that cross a grid barrier (relaxed, agent scope -> sc1)
__device__ __forceinline__ void store8_sc1(void* p, u32x2 v) {
  __hip_atomic_store(reinterpret_cast<uint64_t*>(p), (uint64_t)v[0] | ((uint64_t)v[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32x2 load8_sc1(const void* p) {
  const uint64_t v = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u32x2{(uint32_t)v, (uint32_t)(v >> 32)};
}
__device__ __forceinline__ uint32_t load4_sc1(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same loads WITHOUT the wait hipcc puts behind every atomic load (sixteen rows read one after the other were sixteen
// exposed L2 round trips: 4.8 us of stage 2): issue a batch with *_nowait, then sc1_wait_all on the destinations —
// the values may be used only behind that wait (its "+v" operands make every use depend on it).
__device__ __forceinline__ void load8_sc1_nowait(u32x2& dst, const void* p) {
  asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void load4_sc1_nowait(uint32_t& dst, const void* p) {
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
template <typename V>
__device__ __forceinline__ void sc1_wait_all(V (&r)[16]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                 "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
               :
               : "memory");
}

template <typename T>
__device__ __forceinline__ typename Act<T>::vec8 frag16(const T* p, bool ok) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (ok) v = *reinterpret_cast<const u32x4*>(p);
  return __builtin_bit_cast(typename Act<T>::vec8, v);
}

// wave_sum of N independent values at once: the six cross-lane steps of all of them interleaved. One after the other
// (16 rows x 2 sums x 6 dependent permutes, alone on its SIMD) the LayerNorm of 16 rows took 8.7 us.
template <int N>
__device__ __forceinline__ void wave_sum_batch(float (&v)[N]) {
  // level by level over all N values (DPP adds, one swizzle, one permute each: common.hpp group_sum), so that the N chains overlap
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_mov<0xB1>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x4E>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x141>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x140>(v[i]);
  float t[N];
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = lane_xor16(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += t[i];
#pragma unroll
  for (int i = 0; i < N; ++i) t[i] = __shfl_xor(v[i], 32, 64);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += t[i];
}

template <typename T>
__device__ __forceinline__ typename Act<T>::vec8 frag16_sc1(const T* p, bool ok) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (ok) { const u32x2 lo = load8_sc1(p), hi = load8_sc1(p + 4); v = u32x4{lo[0], lo[1], hi[0], hi[1]}; }
  return __builtin_bit_cast(typename Act<T>::vec8, v);
}

// diagnostic build only (-DMST_TAIL_STAMPS): workgroup 0 leaves s_memrealtime stamps (100 MHz) in a device array of its own, read
// back with mst_debug_tail_stamps (they used to go behind the caller's three sync words, i.e. out of bounds for any caller but
// the stamp tools)
#ifdef MST_TAIL_STAMPS
__device__ uint32_t g_tail_stamps[32];
#define TAIL_STAMP(i) do { if (g == 0 && threadIdx.x == 0 && (i) < 32) g_tail_stamps[(i)] = (uint32_t)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TAIL_STAMP(i) do { } while (0)
#endif

// ONE XCD. An agent-scope (sc1) access has to be served behind the L2s — each XCD has its own, and they are not coherent
// with each other — so with the G workgroups dealt round-robin over the eight XCDs every row that crossed a barrier
// travelled over the fabric: 8 us for the 128 KB of stage 3, 3-5 us per barrier waiting for write-through acknowledgements
// from memory. The kernels therefore launch 12 x more workgroups than they need and let them JOIN: the first arrival
// claims its XCD (s_getreg XCC_ID) with a compare-and-swap, workgroups on that XCD take the roles 0..G-1 in arrival order,
// everybody else leaves at once. All participants then share one L2, and the bytes that cross a barrier are written through
// the CU's L1 to that L2 (sc0 stores, complete — s_waitcnt vmcnt(0) — before the workgroup arrives at the barrier) and read
// behind the barrier with loads that find them there. What makes those loads safe is NOT a cache-bypass bit (a polling loop
// of sc0 loads was tried for the barrier counter and kept reading its first value: the L1 serves repeats) but that every such
// line is read by a CU for the FIRST time since the L1 invalidate at kernel start — each stage's operand rows are new to the
// workgroup that loads them, and the L1 does not allocate on the write-through of a partial line (the 32-byte column slices a
// workgroup itself stored: parity tests cover exactly that) — so the load misses and is served by the shared L2. Correct
// for ANY dispatch order (participants share an XCD by construction; a launch that put fewer than G workgroups on the
// claimed XCD would leave the bounded barrier spin and fail the step's parity, not hang). The barrier counter itself stays
// agent-scope.
// sync: [0] barrier counter, [1] claimed XCD + 1, [2] roles handed out (all zero at launch).
#ifndef MST_TAIL_OVERSUBSCRIBE
#define MST_TAIL_OVERSUBSCRIBE 12  /* 8 is exact under round-robin dispatch; measured 8 / 10 / 12 / 16: 0.7245 / 0.7234 / 0.7256 / 0.7256 ms per step */
#endif
constexpr int TAIL_OVERSUBSCRIBE = MST_TAIL_OVERSUBSCRIBE;
__device__ __forceinline__ int tail_join(uint32_t* sync, int G) {
  __shared__ int role;
  if (threadIdx.x == 0) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & 0xFu) + 1u;
    uint32_t seen = 0u;
    __hip_atomic_compare_exchange_strong(sync + 1, &seen, xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t claimed = seen == 0u ? xcc : seen;
    int r = -1;  // -1: another XCD (free to do riding work), -2: the claimed XCD without a role (leaves: its L2 is the chain's)
    if (claimed == xcc) {
      const uint32_t k = __hip_atomic_fetch_add(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      r = k < (uint32_t)G ? (int)k : -2;
    }
    role = r;
  }
  __syncthreads();
  return role;
}

// RIDERS. A tail launch keeps ONE XCD busy for ~26 us of dependent round trips while seven idle. Work that nothing in the chain
// waits for can use them: the workgroups that join on another XCD, instead of leaving, take 128 x 128 tiles of ONE GEMM
// (mst_gemm_args `g`: the decoder's K | Q | V projection of rows 1..T behind the forward tail, the input gradient of that
// projection behind the backward tail — both were launches of their own in the step's dependent chain, 12 and 10 us) from a
// work queue (`counter`: a zeroed device word in a cache line of its OWN — next to the barrier words, the riders' first 224 ticket
// atomics queued in front of the chain's barrier arrivals: +3 us on the launch) until it is empty. 16 waves as a 2 x 8 grid of 64 x 16 wave tiles, the
// tile code of gemm_nt.hip (gemm_tile.hpp); the tails' own participants never touch it. A tile's result does not depend on who
// computes it or when, so the launch stays deterministic; and the chain's own participants pass by the queue when they are done
// (normally empty by then), so every tile is computed by the end of the launch even if NO workgroup landed on another XCD.
// BM = 256 (4 x 4 waves of 64 x 32, the epilogue staged one 64-row block at a time): for GEMMs with more 128 x 128 tiles than riders —
// a tile is a chain of dependent latencies (~9 us whatever its size) and a 16-wave workgroup has a CU to itself, so ONE round of
// bigger tiles ends long before two rounds of small ones (the decoder projection: 192 tiles on 224 riders instead of 384).
// ... and BEHIND the GEMM's tiles in the same queue (forward tail only): the step's transposed 16-bit shadow refresh (shadows.hpp; the
// matrices only the backward pass reads — nothing in this launch reads them), sixteen 32 x 32 tiles per ticket, four per 256 threads with
// their loads requested together (68 KB of the rider's LDS; at four tiles per ticket the refresh was two rounds of round trips and the
// chain's participants met an unfinished queue). On the
// step's first launch they were 4.9 us of its 24; here they run on compute units that have nothing else to do until the chain ends.
struct TailShadow {
  const float* w; void* wt16; const int64_t* desc; const int64_t* prefix; int n_mat; int n_groups; int64_t tiles;
};
// (a function of its own, not inlined: inside the tail kernel its index arithmetic raised the kernel's register pressure and the
// allocator spilled 49 registers instead of 18 — some of them in the chain's stages: forward tail 34 -> 42 us)
template <typename T>
__device__ __attribute__((noinline)) void tail_shadow_ticket(const TailShadow& sh, int grp, unsigned char* smem) {
  const int quarter = threadIdx.x >> 8;
  shadow_tile_quad<T>(sh.w, reinterpret_cast<T*>(sh.wt16), sh.desc, sh.prefix, sh.n_mat, (int64_t)grp * 16 + 4 * quarter, sh.tiles,
                      reinterpret_cast<float(*)[32][33]>(smem) + 4 * quarter, threadIdx.x & 255);
}
template <typename T, int BM>
__device__ __forceinline__ void tail_ride_bm(const mst_gemm_args& g, int n_tiles, uint32_t* counter, unsigned char* smem, const TailShadow& sh) {
  constexpr int BN = 128, WGM = BM == 256 ? 4 : 2, WGN = 16 / WGM, PATH = BM == 256 ? 4 : 1;
  __shared__ int s_tile;
  for (;;) {
    if (threadIdx.x == 0) s_tile = (int)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int tile = s_tile;
    if (tile >= n_tiles) {
      const int grp = tile - n_tiles;
      if (grp >= sh.n_groups) return;
      tail_shadow_ticket<T>(sh, grp, smem);
      __syncthreads();  // (the tile buffers and s_tile are rewritten)
      continue;
    }
    f32x4 acc[(BN / WGN) / 16][(BM / WGM) / 16];
    int64_t m0, n0;
    float bias_pre[8];
    gemm_bias_preload<BM, BN>(g, bias_pre, tile);
    gemm_mainloop<T, BM, BN, WGM, WGN, 64, true, false>(g, smem, acc, m0, n0, tile);
    gemm_epilogue<T, BM, BN, WGM, WGN, false, true, PATH, false>(g, smem, acc, m0, n0, bias_pre);
    __syncthreads();  // (the epilogue's staging tile is the next tile's first K stage; s_tile is rewritten)
  }
}
__device__ __forceinline__ void store8_l2(void* p, u32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store4_l2(void* p, uint32_t v) {
  asm volatile("global_store_dword %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void load16_l2_nowait(u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void load8_l2_nowait(u32x2& dst, const void* p) {
  asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void load4_l2_nowait(uint32_t& dst, const void* p) {
  asm volatile("global_load_dword %0, %1, off sc0" : "=v"(dst) : "v"(p) : "memory");
}
template <typename V, int N>
__device__ __forceinline__ void l2_wait(V (&r)[N]) {  // the *_nowait destinations may be used only behind this
  static_assert(N == 1 || N == 2 || N == 4 || N == 8, "batch size");
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]) : : "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]) : : "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : : "memory");
}

// The chain's participants at the end of their kernel: every tile of the rider must have been HANDED OUT by then (whoever took one
// finishes it before the launch ends). The queue counter is looked at with a load requested at the START of the chain's last stage
// (`seen`, thread 0; sc1: the riders' atomics live behind the L2s) — normally it already shows an empty queue and the participants
// leave without another round trip (an atomic here cost the launch ~2 us); if it does not — or if no workgroup ever landed on
// another XCD, so that nobody rode — they take tiles themselves until the queue is empty.
template <typename T, int BM>
__device__ __forceinline__ void tail_drain(const mst_gemm_args& g, int n_tiles, uint32_t* counter, uint32_t& seen, unsigned char* smem,
                                           const TailShadow& sh) {
  __shared__ int s_drain;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(seen) : : "memory");
    s_drain = seen < (uint32_t)(n_tiles + sh.n_groups);
  }
  __syncthreads();
  if (s_drain) tail_ride_bm<T, BM>(g, n_tiles, counter, smem, sh);
}

// LayerNorm of one row held as 4 elements per lane (D = 256) or 2 (D = 128): layernorm_fwd_kernel's arithmetic
template <typename T, int D>
__device__ __forceinline__ void ln_row(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                       int lane, float (&y)[D / 64], float& mean, float& rstd) {
  constexpr int E = D / 64;
  float v[E];
  float s = 0.f;
  if constexpr (E == 4) {
    u32x2 r[1];
    load8_l2_nowait(r[0], x + lane * 4);
    l2_wait(r);
    v[0] = bits_to_f32<T>((uint16_t)(r[0][0] & 0xffff)); v[1] = bits_to_f32<T>((uint16_t)(r[0][0] >> 16));
    v[2] = bits_to_f32<T>((uint16_t)(r[0][1] & 0xffff)); v[3] = bits_to_f32<T>((uint16_t)(r[0][1] >> 16));
  } else {
    uint32_t r[1];
    load4_l2_nowait(r[0], x + lane * 2);
    l2_wait(r);
    v[0] = bits_to_f32<T>((uint16_t)(r[0] & 0xffff)); v[1] = bits_to_f32<T>((uint16_t)(r[0] >> 16));
  }
#pragma unroll
  for (int e = 0; e < E; ++e) s += v[e];
  mean = wave_sum(s) * (1.f / (float)D);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) { const float d = v[e] - mean; ss += d * d; }
  rstd = 1.f / sqrtf(wave_sum(ss) * (1.f / (float)D) + eps);
#pragma unroll
  for (int e = 0; e < E; ++e) y[e] = (v[e] - mean) * rstd * gamma[lane * E + e] + beta[lane * E + e];
}

// Sixteen waves per workgroup (four per SIMD): every stage is a chain of dependent L2 round trips, and with one wave per
// SIMD nothing ran under them (four waves: forward 30 us, backward 43 us against 39 for its five launches). Wave (mi, wq):
// row block mi = wave & 3 (rows 16 mi .. 16 mi + 15 in the MFMA stages), quarter wq = wave >> 2 — the 64-column slices are
// split by 16-column block, the 16-column slices by quarter of K (partial accumulators meet in LDS); the LayerNorms take
// four rows per wave.
constexpr int TAIL_WAVES = 16;
template <typename V>
__device__ __forceinline__ void sc1_wait4(V (&r)[4]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "memory");
}
// partial 16-column accumulators of the four K quarters -> their sum on the waves of quarter 0 (sP: [4][64][16] floats)
__device__ __forceinline__ f32x4 quarter_sum(float* sP, const f32x4& part, int wq, int m, int lq) {
  *reinterpret_cast<f32x4*>(sP + (wq * 64 + m) * 16 + 4 * lq) = part;
  __syncthreads();
  f32x4 acc = part;
  if (wq == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(sP + (w * 64 + m) * 16 + 4 * lq);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += o[e];
    }
  }
  return acc;
}

// RIDE: 0, or the rider tile's rows (128 / 256) — one rider form per kernel: both in one made the register allocator spill
template <typename T, int D, int RIDE = 0>
__global__ __launch_bounds__(1024) void row_tail_fwd_kernel(mst_row_tail_args q, mst_gemm_args ride, int ride_tiles, uint32_t* ride_queue,
                                                            TailShadow ride_sh) {
  constexpr int G = D / 16, F = 4 * D, LDX = D + 8, E = D / 64;
  constexpr int KQ1 = D / 32 / 4, KQ2 = F / 32 / 4;  // k-steps per K quarter of the two 16-column stages
  __shared__ __attribute__((aligned(16))) T sX1[64 * LDX];  // LayerNorm-1 output of every row (FFN1's operand, FFN2's residual)
  __shared__ __attribute__((aligned(16))) float sP[4 * 64 * 16];
  // every bias / gamma / beta of the chain, requested at once at the start: they are cold lines after an optimizer step, and
  // each would be one more dependent round trip inside its stage ([bp | g1 | be1 | b2 | g2 | be2 | b1])
  __shared__ __attribute__((aligned(16))) float sPar[6 * D + F];
  typedef typename Act<T>::vec8 vec8;
  typedef typename std::conditional<E == 4, u32x2, uint32_t>::type raw_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mi = wave & 3, wq = wave >> 2;
  const int g = tail_join(q.sync, G);
  if (g < 0) {
    if constexpr (RIDE != 0) {
      extern __shared__ __attribute__((aligned(16))) unsigned char ride_smem[];
      if (g == -1) tail_ride_bm<T, RIDE>(ride, ride_tiles, ride_queue, ride_smem, ride_sh);
    }
    return;
  }
  for (int i = tid; i < D; i += TAIL_WAVES * 64) {
    sPar[i] = q.bp[i]; sPar[D + i] = q.g1[i]; sPar[2 * D + i] = q.be1[i];
    sPar[3 * D + i] = q.b2[i]; sPar[4 * D + i] = q.g2[i]; sPar[5 * D + i] = q.be2[i];
  }
  for (int i = tid; i < F; i += TAIL_WAVES * 64) sPar[6 * D + i] = q.b1[i];
  const float* const s_bp = sPar, * const s_g1 = sPar + D, * const s_be1 = sPar + 2 * D, * const s_b2 = sPar + 3 * D,
             * const s_g2 = sPar + 4 * D, * const s_be2 = sPar + 5 * D, * const s_b1 = sPar + 6 * D;
  const int li = lane & 15, lq = lane >> 4;
  const int B = (int)q.B;
  const int m = mi * 16 + li;  // the row this lane's accumulator column belongs to
  const bool m_ok = m < B;
  const int mc = m_ok ? m : 0;  // (row of the unconditional loads)
  const int64_t pm = (int64_t)m * q.phys_stride;  // physical row in the [B * S, N] tensors: the dropout counter's row
  const float p = q.dropout_p;
  const bool drop = p > 0.f;
  const uint64_t dseed = q.dropout_seed ^ ((drop && q.dropout_seed_ptr) ? q.dropout_seed_ptr[0] : 0ull);
  const uint32_t thr = dropout_thr(p);
  const float inv_keep = dropout_inv_keep(p);
  const T* att = reinterpret_cast<const T*>(q.att);
  const T* xin = reinterpret_cast<const T*>(q.resid);
  T* h1 = reinterpret_cast<T*>(q.h1); T* x1 = reinterpret_cast<T*>(q.x1); T* a = reinterpret_cast<T*>(q.a);
  T* h2 = reinterpret_cast<T*>(q.h2); T* x2 = reinterpret_cast<T*>(q.x2);

  auto finish4 = [&](const f32x4& acc, const float* bias, int n, uint32_t site, int64_t N, bool relu, const T* res /* row ptr or null */,
                     T* dst /* row ptr */) {
    // one lane's four consecutive output columns n..n+3 of row m: bias, ReLU, dropout, residual, rounding — gemm_epilogue's order
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + n);
    uint32_t keep = 0xFu;
    if (drop) keep = dropout_keep4k(dropout_key(dseed, site), (uint64_t)(pm * N + n) >> 2, thr);
    float r4[4] = {0.f, 0.f, 0.f, 0.f};
    if (res) {
      const u32x2 rv = *reinterpret_cast<const u32x2*>(res + n);
      r4[0] = bits_to_f32<T>((uint16_t)(rv[0] & 0xffff)); r4[1] = bits_to_f32<T>((uint16_t)(rv[0] >> 16));
      r4[2] = bits_to_f32<T>((uint16_t)(rv[1] & 0xffff)); r4[3] = bits_to_f32<T>((uint16_t)(rv[1] >> 16));
    }
    uint16_t hb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = acc[e] + b4[e];
      if (relu) t = fmaxf(t, 0.f);
      if (drop) t = ((keep >> e) & 1u) ? t * inv_keep : 0.f;
      hb[e] = f32_to_bits<T>(t + r4[e]);
    }
    store8_l2(dst + n, u32x2{(uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16)});
  };

  TAIL_STAMP(0);
  // ---------------- stage 1: h1[:, 16 g .. 16 g + 15] = x_in + dropout(att W_proj^T + b)     (K quarter wq per wave)
  {
    const int n0 = g * 16;
    const T* Wp = reinterpret_cast<const T*>(q.Wp) + (int64_t)(n0 + li) * q.ldwp + 8 * lq + 32 * KQ1 * wq;
    const T* Ar = att + (int64_t)m * q.rs_att + 8 * lq + 32 * KQ1 * wq;
    vec8 wf[KQ1], xf[KQ1];
#pragma unroll
    for (int ks = 0; ks < KQ1; ++ks) { wf[ks] = frag16<T>(Wp + 32 * ks, true); xf[ks] = frag16<T>(Ar + 32 * ks, m_ok); }
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KQ1; ++ks) part = Act<T>::mfma16(wf[ks], xf[ks], part);
    const f32x4 acc = quarter_sum(sP, part, wq, m, lq);
    if (wq == 0 && m_ok) finish4(acc, s_bp, n0 + 4 * lq, q.site0, D, false, xin + (int64_t)m * q.rs_res, h1 + (int64_t)m * q.rs_d);
  }
  // (memory-level parallelism is the whole game here: every stage is a few dependent L2 / fabric round trips, so the
  // next stage's weight fragments are requested BEFORE the barrier they do not depend on, and rows are loaded in batches)
  const int n1 = g * 64 + 16 * wq;  // this wave's 16-column block of the FFN1 slice
  const T* W1p = reinterpret_cast<const T*>(q.W1) + (int64_t)(n1 + li) * q.ldw1 + 8 * lq;
  vec8 w1f[D / 32];
#pragma unroll
  for (int ks = 0; ks < D / 32; ++ks) w1f[ks] = frag16<T>(W1p + 32 * ks, true);
  TAIL_STAMP(1);
  grid_sync(q.sync, G, q.status, MST_TAIL_SPIN_FWD);
  TAIL_STAMP(2);

  // ---------------- stage 2: x1 = LN1(h1) for every row (each workgroup, into LDS; the rows' owner also to HBM), then
  //                  a[:, 64 g .. 64 g + 63] = dropout(relu(x1 W1^T + b1))
  {
    constexpr int RPW = 4;  // rows per wave: rows wave, wave + 16, ...
    float v[RPW][E];
    // all rows requested at once, unconditionally (a row past the batch re-reads the last one and is zeroed below)
    raw_t raw[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + TAIL_WAVES * i, rc = r < B ? r : B - 1;
      if constexpr (E == 4) load8_l2_nowait(raw[i], h1 + (int64_t)rc * q.rs_d + lane * 4);
      else load4_l2_nowait(raw[i], h1 + (int64_t)rc * q.rs_d + lane * 2);
    }
    l2_wait(raw);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const bool ok = wave + TAIL_WAVES * i < B;
      if constexpr (E == 4) {
        const u32x2 t = raw[i];
        v[i][0] = ok ? bits_to_f32<T>((uint16_t)(t[0] & 0xffff)) : 0.f; v[i][1] = ok ? bits_to_f32<T>((uint16_t)(t[0] >> 16)) : 0.f;
        v[i][2] = ok ? bits_to_f32<T>((uint16_t)(t[1] & 0xffff)) : 0.f; v[i][3] = ok ? bits_to_f32<T>((uint16_t)(t[1] >> 16)) : 0.f;
      } else {
        const uint32_t t = raw[i];
        v[i][0] = ok ? bits_to_f32<T>((uint16_t)(t & 0xffff)) : 0.f; v[i][1] = ok ? bits_to_f32<T>((uint16_t)(t >> 16)) : 0.f;
      }
    }
    float gm[E], bt[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { gm[e] = s_g1[lane * E + e]; bt[e] = s_be1[lane * E + e]; }
#ifdef MST_TAIL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TAIL_STAMP(8);
#endif
    float mean[RPW], rstd[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      mean[i] = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) mean[i] += v[i][e];
    }
    wave_sum_batch<RPW>(mean);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      mean[i] *= (1.f / (float)D);
      rstd[i] = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) { const float d = v[i][e] - mean[i]; rstd[i] += d * d; }
    }
    wave_sum_batch<RPW>(rstd);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + TAIL_WAVES * i;
      rstd[i] = 1.f / sqrtf(rstd[i] * (1.f / (float)D) + q.eps);
      uint16_t yb[E];
#pragma unroll
      for (int e = 0; e < E; ++e) yb[e] = (r < B) ? f32_to_bits<T>((v[i][e] - mean[i]) * rstd[i] * gm[e] + bt[e]) : (uint16_t)0;  // rows past the batch: zeros
      if constexpr (E == 4) {
        const u32x2 o = {(uint32_t)yb[0] | ((uint32_t)yb[1] << 16), (uint32_t)yb[2] | ((uint32_t)yb[3] << 16)};
        *reinterpret_cast<u32x2*>(sX1 + r * LDX + lane * 4) = o;
        if (r < B && r % G == g) *reinterpret_cast<u32x2*>(x1 + (int64_t)r * q.rs_d + lane * 4) = o;
      } else {
        const uint32_t o = (uint32_t)yb[0] | ((uint32_t)yb[1] << 16);
        *reinterpret_cast<uint32_t*>(sX1 + r * LDX + lane * 2) = o;
        if (r < B && r % G == g) *reinterpret_cast<uint32_t*>(x1 + (int64_t)r * q.rs_d + lane * 2) = o;
      }
      if (r < B && r % G == g && lane == 0) { q.mean1[(int64_t)r * q.stat_stride] = mean[i]; q.rstd1[(int64_t)r * q.stat_stride] = rstd[i]; }
    }
  }
  TAIL_STAMP(9);
  __syncthreads();
  TAIL_STAMP(10);
  const int n2 = g * 16;
  const T* W2p = reinterpret_cast<const T*>(q.W2) + (int64_t)(n2 + li) * q.ldw2 + 8 * lq + 32 * KQ2 * wq;
  vec8 w2f[KQ2];  // this wave's quarter of the FFN2 weights: requested before the barrier
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
      const vec8 xf = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4*>(sX1 + m * LDX + 32 * ks + 8 * lq));
      acc = Act<T>::mfma16(w1f[ks], xf, acc);
    }
    TAIL_STAMP(11);
#pragma unroll
    for (int ks = 0; ks < KQ2; ++ks) w2f[ks] = frag16<T>(W2p + 32 * ks, true);
    if (m_ok) finish4(acc, s_b1, n1 + 4 * lq, q.site0 + 1, F, true, nullptr, a + (int64_t)m * q.rs_a);
  }
  TAIL_STAMP(3);
  grid_sync(q.sync, 2 * G, q.status, MST_TAIL_SPIN_FWD);
  TAIL_STAMP(4);

  // ---------------- stage 3: h2[:, 16 g ..] = x1 + dropout(a W2^T + b2)        (K = 4 D: the whole hidden row, a quarter per wave)
  {
    const T* Ar = a + (int64_t)mc * q.rs_a + 8 * lq + 32 * KQ2 * wq;
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
    u32x4 xr[KQ2];  // the hidden row's fragments: every load in flight at once
#pragma unroll
    for (int ks = 0; ks < KQ2; ++ks) load16_l2_nowait(xr[ks], Ar + ks * 32);
    l2_wait(xr);
#pragma unroll
    for (int ks = 0; ks < KQ2; ++ks) part = Act<T>::mfma16(w2f[ks], __builtin_bit_cast(vec8, m_ok ? xr[ks] : u32x4{0u, 0u, 0u, 0u}), part);
    const f32x4 acc = quarter_sum(sP, part, wq, m, lq);
    if (wq == 0 && m_ok) finish4(acc, s_b2, n2 + 4 * lq, q.site0 + 2, D, false, sX1 + m * LDX, h2 + (int64_t)m * q.rs_d);
  }
  TAIL_STAMP(5);
  grid_sync(q.sync, 3 * G, q.status, MST_TAIL_SPIN_FWD);
  TAIL_STAMP(6);
  uint32_t ride_seen = 0u;  // (riders) the tile queue's counter as of now: tail_drain
  if constexpr (RIDE != 0) { if (tid == 0) load4_sc1_nowait(ride_seen, ride_queue); }

  // ---------------- stage 4: x2 = LN2(h2), rows dealt to the workgroups
  for (int r = g + G * wave; r < B; r += TAIL_WAVES * G) {
    float y[E], mean, rstd;
    ln_row<T, D>(h2 + (int64_t)r * q.rs_d, s_g2, s_be2, q.eps, lane, y, mean, rstd);
    uint16_t yb[E];
#pragma unroll
    for (int e = 0; e < E; ++e) yb[e] = f32_to_bits<T>(y[e]);
    if constexpr (E == 4) *reinterpret_cast<u32x2*>(x2 + (int64_t)r * q.rs_d + lane * 4) =
        u32x2{(uint32_t)yb[0] | ((uint32_t)yb[1] << 16), (uint32_t)yb[2] | ((uint32_t)yb[3] << 16)};
    else *reinterpret_cast<uint32_t*>(x2 + (int64_t)r * q.rs_d + lane * 2) = (uint32_t)yb[0] | ((uint32_t)yb[1] << 16);
    if (lane == 0) { q.mean2[(int64_t)r * q.stat_stride] = mean; q.rstd2[(int64_t)r * q.stat_stride] = rstd; }
  }
  TAIL_STAMP(7);
  if constexpr (RIDE != 0) {  // the chain is done: whatever the riders left in the queue (normally nothing)
    extern __shared__ __attribute__((aligned(16))) unsigned char ride_smem[];
    __syncthreads();
    tail_drain<T, RIDE>(ride, ride_tiles, ride_queue, ride_seen, ride_smem, ride_sh);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same rows on the way back (mst_row_tail_bwd): LayerNorm-2 backward -> FFN2 dgrad (ReLU gate) -> FFN1 dgrad (+ the
// residual branch) -> LayerNorm-1 backward -> W_proj dgrad, five launches of the step (two LayerNorm backward launches on B
// rows and three GEMMs with M = B) as one, with the forward kernel's decomposition: D / 16 workgroups, each owning 64 hidden
// columns of the FFN2 dgrad and 16 output columns of the FFN1 and W_proj dgrads; both LayerNorm backward passes are
// recomputed by every workgroup on all rows (their results are the next GEMM's operand, kept in LDS), so only two results
// cross a grid barrier: d(pre-activation) and the FFN1 dgrad's output. The workgroup's FFN2-dgrad weights wait in LDS.
template <typename T, int D, int RIDE = 0>
__global__ __launch_bounds__(1024) void row_tail_bwd_kernel(mst_row_tail_bwd_args q, mst_gemm_args ride, int ride_tiles, uint32_t* ride_queue) {
  constexpr int G = D / 16, F = 4 * D, LDX = D + 8, E = D / 64, RPW = 4;
  constexpr int KQ1 = D / 32 / 4, KQ2 = F / 32 / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char tail_smem[];
  T* sA = reinterpret_cast<T*>(tail_smem);                     // [64][LDX] masked LayerNorm-backward rows: the GEMM operand
  T* sW = sA + 64 * LDX;                                        // [64][LDX] W2t rows 64 g .. 64 g + 63
  float* sRed = reinterpret_cast<float*>(sW + 64 * LDX);        // [16 waves][2][D]
  float* sP = sRed + TAIL_WAVES * 2 * D;                        // [4][64][16]
  float* sGam = sP + 4 * 64 * 16;                               // [g2 | g1]: cold parameter lines, requested at the start
  typedef typename Act<T>::vec8 vec8;
  typedef typename std::conditional<E == 4, u32x2, uint32_t>::type raw_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mi = wave & 3, wq = wave >> 2;
  const int g = tail_join(q.sync, G);
  if (g < 0) {
    if constexpr (RIDE != 0) {
      if (g == -1) tail_ride_bm<T, RIDE>(ride, ride_tiles, ride_queue, tail_smem, TailShadow{});
    }
    return;
  }
  const int li = lane & 15, lq = lane >> 4;
  const int B = (int)q.B;
  const int m = mi * 16 + li;
  const bool m_ok = m < B;
  const int mc = m_ok ? m : 0;
  const float p = q.dropout_p;
  const bool drop = p > 0.f;
  const uint64_t dseed = q.dropout_seed ^ ((drop && q.dropout_seed_ptr) ? q.dropout_seed_ptr[0] : 0ull);
  const uint32_t thr = dropout_thr(p);
  const float inv_keep = dropout_inv_keep(p), inv_d = 1.f / (float)D;
  const T* dy = reinterpret_cast<const T*>(q.dy);
  const T* h2 = reinterpret_cast<const T*>(q.h2); const T* h1 = reinterpret_cast<const T*>(q.h1);
  const T* a = reinterpret_cast<const T*>(q.a);
  T* dh = reinterpret_cast<T*>(q.dh); T* dhm = reinterpret_cast<T*>(q.dhm); T* dx1 = reinterpret_cast<T*>(q.dx1);
  T* dh1m = reinterpret_cast<T*>(q.dh1m); T* dpre = reinterpret_cast<T*>(q.dpre); T* dh1 = reinterpret_cast<T*>(q.dh1);
  T* datt = reinterpret_cast<T*>(q.datt);
  const int n1 = g * 64 + 16 * wq, n2 = g * 16;

  auto unpack = [&](raw_t t, float (&v)[E]) {
    if constexpr (E == 4) {
      v[0] = bits_to_f32<T>((uint16_t)(t[0] & 0xffff)); v[1] = bits_to_f32<T>((uint16_t)(t[0] >> 16));
      v[2] = bits_to_f32<T>((uint16_t)(t[1] & 0xffff)); v[3] = bits_to_f32<T>((uint16_t)(t[1] >> 16));
    } else {
      v[0] = bits_to_f32<T>((uint16_t)(t & 0xffff)); v[1] = bits_to_f32<T>((uint16_t)(t >> 16));
    }
  };
  auto pack = [&](const float (&v)[E]) -> raw_t {
    if constexpr (E == 4) return u32x2{(uint32_t)f32_to_bits<T>(v[0]) | ((uint32_t)f32_to_bits<T>(v[1]) << 16),
                                       (uint32_t)f32_to_bits<T>(v[2]) | ((uint32_t)f32_to_bits<T>(v[3]) << 16)};
    else return (uint32_t)f32_to_bits<T>(v[0]) | ((uint32_t)f32_to_bits<T>(v[1]) << 16);
  };
  auto store_row = [&](T* dst, raw_t v, bool through) {  // `through`: another workgroup reads it behind a grid barrier
    if constexpr (E == 4) { if (through) store8_l2(dst, v); else *reinterpret_cast<u32x2*>(dst) = v; }
    else { if (through) store4_l2(dst, v); else *reinterpret_cast<uint32_t*>(dst) = v; }
  };

  // LayerNorm backward of every row (rows wave, wave + 16, ... of this wave; columns E * lane ..): layernorm_bwd_kernel's
  // arithmetic. raw_dy / raw_x: the rows' dy and pre-norm x; the masked result goes to sA, the rows this workgroup owns
  // (r % G == g) to dx_out / dxm_out; dgamma / dbeta are added by workgroup 0.
  auto ln_bwd_rows = [&](raw_t (&raw_dy)[RPW], raw_t (&raw_x)[RPW], const float* mean_in, const float* rstd_in, const float* gamma,
                         uint32_t site, T* dx_out, int64_t rs_dx, bool dx_through, T* dxm_out, int64_t rs_dxm, float* dgamma, float* dbeta) {
    float gm[E], dg[E], db[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { gm[e] = gamma[lane * E + e]; dg[e] = 0.f; db[e] = 0.f; }
    float xh[RPW][E], gv[RPW][E], s1[RPW], s2[RPW], rs[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + TAIL_WAVES * i, rc = r < B ? r : B - 1;
      const bool ok = r < B;
      const float mean = mean_in[(int64_t)rc * q.stat_stride];
      rs[i] = rstd_in[(int64_t)rc * q.stat_stride];
      float xv[E], dv[E];
      unpack(raw_x[i], xv); unpack(raw_dy[i], dv);
      s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float x_hat = ok ? (xv[e] - mean) * rs[i] : 0.f, d = ok ? dv[e] : 0.f, gg = d * gm[e];
        xh[i][e] = x_hat; gv[i][e] = gg;
        s1[i] += gg; s2[i] += gg * x_hat;
        dg[e] += d * x_hat; db[e] += d;
      }
    }
    wave_sum_batch<RPW>(s1);
    wave_sum_batch<RPW>(s2);
    const uint32_t key = dropout_key(dseed, site);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + TAIL_WAVES * i;
      const bool ok = r < B;
      const float m1 = s1[i] * inv_d, m2 = s2[i] * inv_d;
      uint32_t keep = 0xFu;
      if (drop) keep = dropout_keep4k(key, (uint64_t)((int64_t)r * q.phys_stride * D + lane * E) >> 2, thr) >> ((lane * E) & 3);
      float o[E], om[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        o[e] = rs[i] * (gv[i][e] - m1 - xh[i][e] * m2);
        om[e] = drop ? (((keep >> e) & 1u) ? o[e] * inv_keep : 0.f) : o[e];
        if (!ok) { o[e] = 0.f; om[e] = 0.f; }
      }
      const raw_t ob = pack(o), mb = pack(om);
      *reinterpret_cast<raw_t*>(sA + r * LDX + lane * E) = mb;
      if (ok && r % G == g) {
        store_row(dx_out + (int64_t)r * rs_dx + lane * E, ob, dx_through);
        store_row(dxm_out + (int64_t)r * rs_dxm + lane * E, mb, false);
      }
    }
    if (g == 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) { sRed[(wave * 2 + 0) * D + lane * E + e] = dg[e]; sRed[(wave * 2 + 1) * D + lane * E + e] = db[e]; }
    }
    __syncthreads();  // (also: sA is complete)
    if (g == 0) {
      for (int c = tid; c < 2 * D; c += TAIL_WAVES * 64) {
        const int which = c / D, col = c % D;
        float t = 0.f;
        for (int w = 0; w < TAIL_WAVES; ++w) t += sRed[(w * 2 + which) * D + col];
        atomicAdd((which ? dbeta : dgamma) + col, t);
      }
    }
  };

  TAIL_STAMP(0);
  // ---------------- this workgroup's FFN2-dgrad weights -> LDS; stage 1's rows requested
  {
    const T* W2t = reinterpret_cast<const T*>(q.W2t);
    constexpr int CPR = D / 8, WCH = 64 * CPR / (TAIL_WAVES * 64);
    u32x4 wv[WCH];
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int c = tid + i * TAIL_WAVES * 64, row = c / CPR, ch = c % CPR;
      wv[i] = *reinterpret_cast<const u32x4*>(W2t + (int64_t)(g * 64 + row) * q.ldw2t + ch * 8);
    }
    raw_t rdy[RPW], rx[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + TAIL_WAVES * i, rc = r < B ? r : B - 1;
      rdy[i] = *reinterpret_cast<const raw_t*>(dy + (int64_t)rc * q.rs_dy + lane * E);
      rx[i] = *reinterpret_cast<const raw_t*>(h2 + (int64_t)rc * q.rs_d + lane * E);
    }
    float gpre[2] = {0.f, 0.f};
    if (tid < D) { gpre[0] = q.g2[tid]; gpre[1] = q.g1[tid]; }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int c = tid + i * TAIL_WAVES * 64, row = c / CPR, ch = c % CPR;
      *reinterpret_cast<u32x4*>(sW + row * LDX + ch * 8) = wv[i];
    }
    if (tid < D) { sGam[tid] = gpre[0]; sGam[D + tid] = gpre[1]; }
    __syncthreads();  // (the gammas; every load of this stage is already in flight)
    // ---------------- stage 1: LayerNorm-2 backward of every row
    ln_bwd_rows(rdy, rx, q.mean2, q.rstd2, sGam, q.site0 + 2, dh, q.rs_c, true, dhm, q.rs_c, q.dg2, q.db2);
  }
  TAIL_STAMP(1);
  // ---------------- stage 2: d(pre)[:, 64 g ..] = ((dhm W2t^T) / (1 - p)) gated by a > 0     (16-column block wq per wave)
  const T* W1t = reinterpret_cast<const T*>(q.W1t) + (int64_t)(n2 + li) * q.ldw1t + 8 * lq + 32 * KQ2 * wq;
  vec8 w1f[KQ2];  // this wave's quarter of the FFN1-dgrad weights of stage 3: requested before the barrier
  {
    const u32x2 gate8 = m_ok ? *reinterpret_cast<const u32x2*>(a + (int64_t)m * q.rs_a + n1 + 4 * lq) : u32x2{0u, 0u};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks) {
      const vec8 xf = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4*>(sA + m * LDX + 32 * ks + 8 * lq));
      const vec8 wf = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4*>(sW + (16 * wq + li) * LDX + 32 * ks + 8 * lq));
      acc = Act<T>::mfma16(wf, xf, acc);
    }
#pragma unroll
    for (int ks = 0; ks < KQ2; ++ks) w1f[ks] = frag16<T>(W1t + 32 * ks, true);
    if (m_ok) {
      const uint16_t gb[4] = {(uint16_t)(gate8[0] & 0xffff), (uint16_t)(gate8[0] >> 16), (uint16_t)(gate8[1] & 0xffff), (uint16_t)(gate8[1] >> 16)};
      uint16_t hb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) hb[e] = f32_to_bits<T>((bits_to_f32<T>(gb[e]) > 0.f) ? acc[e] * inv_keep : 0.f);
      store8_l2(dpre + (int64_t)m * q.rs_dpre + n1 + 4 * lq, u32x2{(uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16)});
    }
  }
  TAIL_STAMP(2);
  grid_sync(q.sync, G, q.status, MST_TAIL_SPIN_BWD);
  TAIL_STAMP(3);

  // ---------------- stage 3: dx1[:, 16 g ..] = d(pre) W1t^T + dh        (K = 4 D, a quarter per wave)
  const T* Wpt = reinterpret_cast<const T*>(q.Wpt) + (int64_t)(n2 + li) * q.ldwpt + 8 * lq + 32 * KQ1 * wq;
  vec8 wpf[KQ1];
  {
    const T* Ar = dpre + (int64_t)mc * q.rs_dpre + 8 * lq + 32 * KQ2 * wq;
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
    u32x4 xr[KQ2];
#pragma unroll
    for (int ks = 0; ks < KQ2; ++ks) load16_l2_nowait(xr[ks], Ar + ks * 32);
    u32x2 rvv[1];
    load8_l2_nowait(rvv[0], dh + (int64_t)mc * q.rs_c + n2 + 4 * lq);
    l2_wait(xr);
    l2_wait(rvv);
    const u32x2 rv = rvv[0];
#pragma unroll
    for (int ks = 0; ks < KQ1; ++ks) wpf[ks] = frag16<T>(Wpt + 32 * ks, true);  // stage 5's weights
#pragma unroll
    for (int ks = 0; ks < KQ2; ++ks) part = Act<T>::mfma16(w1f[ks], __builtin_bit_cast(vec8, m_ok ? xr[ks] : u32x4{0u, 0u, 0u, 0u}), part);
    const f32x4 acc = quarter_sum(sP, part, wq, m, lq);
    if (wq == 0 && m_ok) {
      const float r4[4] = {bits_to_f32<T>((uint16_t)(rv[0] & 0xffff)), bits_to_f32<T>((uint16_t)(rv[0] >> 16)),
                           bits_to_f32<T>((uint16_t)(rv[1] & 0xffff)), bits_to_f32<T>((uint16_t)(rv[1] >> 16))};
      uint16_t hb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) hb[e] = f32_to_bits<T>(acc[e] + r4[e]);
      store8_l2(dx1 + (int64_t)m * q.rs_c + n2 + 4 * lq, u32x2{(uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16)});
    }
  }
  TAIL_STAMP(4);
  grid_sync(q.sync, 2 * G, q.status, MST_TAIL_SPIN_BWD);
  TAIL_STAMP(5);
  uint32_t ride_seen = 0u;  // (riders) the tile queue's counter as of now: tail_drain
  if constexpr (RIDE != 0) { if (tid == 0) load4_sc1_nowait(ride_seen, ride_queue); }

  // ---------------- stage 4: LayerNorm-1 backward of every row (operand of stage 5 in LDS)
  {
    raw_t rdy[RPW], rx[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wave + TAIL_WAVES * i, rc = r < B ? r : B - 1;
      if constexpr (E == 4) load8_l2_nowait(rdy[i], dx1 + (int64_t)rc * q.rs_c + lane * 4);
      else load4_l2_nowait(rdy[i], dx1 + (int64_t)rc * q.rs_c + lane * 2);
      rx[i] = *reinterpret_cast<const raw_t*>(h1 + (int64_t)rc * q.rs_d + lane * E);
    }
    l2_wait(rdy);
    ln_bwd_rows(rdy, rx, q.mean1, q.rstd1, sGam + D, q.site0, dh1, q.rs_dh1, false, dh1m, q.rs_c, q.dg1, q.db1);
  }
  TAIL_STAMP(6);
  // ---------------- stage 5: datt[:, 16 g ..] = dh1m Wpt^T        (K quarter per wave)
  {
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KQ1; ++ks) {
      const vec8 xf = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4*>(sA + m * LDX + 32 * (KQ1 * wq + ks) + 8 * lq));
      part = Act<T>::mfma16(wpf[ks], xf, part);
    }
    const f32x4 acc = quarter_sum(sP, part, wq, m, lq);
    if (wq == 0 && m_ok) {
      uint16_t hb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) hb[e] = f32_to_bits<T>(acc[e]);
      *reinterpret_cast<u32x2*>(datt + (int64_t)m * q.rs_datt + n2 + 4 * lq) =
          u32x2{(uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16)};
    }
  }
  TAIL_STAMP(7);
  if constexpr (RIDE != 0) {  // the chain is done: whatever the riders left in the queue (normally nothing)
    __syncthreads();
    tail_drain<T, RIDE>(ride, ride_tiles, ride_queue, ride_seen, tail_smem, TailShadow{});
  }
}

}  // namespace mst

using namespace mst;

#ifdef MST_TAIL_STAMPS
extern "C" int mst_debug_tail_stamps(uint32_t* host_out) {  // diagnostic builds only: 32 realtime stamps of the last tail launch
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mst::g_tail_stamps), sizeof(uint32_t) * 32) == hipSuccess ? 0 : -1;
}
#endif

// a rider GEMM: what tail_ride's tile code takes — whole 128 x 128 tiles, 64-deep K stages, the fast epilogue (16-bit C, bias and / or
// residual, row remaps whose groups are whole tiles)
static int check_rider(const char* who, const mst_gemm_args& g, int dtype) {
  MST_CHECK_ARG(g.dtype == dtype && g.M > 0 && g.N > 0 && g.K > 0 && g.A && g.B && g.C, "%s: rider: bad GEMM", who);
  MST_CHECK_ARG(g.M % 128 == 0 && g.N % 128 == 0 && g.K % 64 == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.ldc % 8 == 0 && g.M * (g.N / 128) < (1ll << 30),
                "%s: rider: M and N must be multiples of 128, K of 64 (got %lld, %lld, %lld)", who, (long long)g.M, (long long)g.N, (long long)g.K);
  MST_CHECK_ARG(!g.c_f32 && !g.a_u8 && !g.gate && !g.rowadd && !g.grpadd && g.act == MST_ACT_NONE && g.dropout_p == 0.f && !g.self_resid,
                "%s: rider: bias, alpha, a residual and row remaps only", who);
  MST_CHECK_ARG((g.a_rows_per_group <= 0 || g.a_rows_per_group % 128 == 0) && (g.c_rows_per_group <= 0 || g.c_rows_per_group % 128 == 0),
                "%s: rider: row-remap groups must be whole 128-row tiles", who);
  const int64_t phys_rows = g.c_rows_per_group > 0 ? (g.M / g.c_rows_per_group) * g.c_group_stride + g.c_group_offset : g.M;
  MST_CHECK_ARG((uint64_t)phys_rows * (uint64_t)g.N < (1ull << 32) && (!g.resid || (g.ldr % 8 == 0 && (uintptr_t)g.resid % 16 == 0)) &&
                ((uintptr_t)g.A | (uintptr_t)g.B | (uintptr_t)g.C) % 16 == 0 && (!g.bias || (uintptr_t)g.bias % 16 == 0),
                "%s: rider: operands must be 16-byte aligned and the output below 2^32 elements", who);
  return MST_OK;
}
// Grid of a launch with riders: G x over workgroups dealt round-robin over the 8 XCDs, 7/8 of them riders. As few as give every
// tile a rider of its own (+ one per XCD to spare), at most 16 x G: the launch pays ~0.6 us per 16 extra 1024-thread workgroups
// (measured: a one-tile rider costs the forward tail +0.6 / +1.3 / +3.5 us at over = 9 / 12 / 16).
static unsigned ride_grid(int64_t G, int tiles) {
  int64_t over = (((int64_t)tiles * 8 + 6) / 7 + G - 1) / G + 1;
  if (over < 9) over = 9;
  if (over > 16) over = 16;
  return (unsigned)(G * over);
}
// (tile rows, dynamic LDS) of a rider: 256-row tiles when the 128-row ones would need a second round of the ~7 x 32 riding workgroups
static void ride_shape(const mst_gemm_args& g, int& tiles_signed, size_t& lds) {
  const int64_t t128 = (g.M / 128) * (g.N / 128);
  const bool big = t128 > 7 * 32 - 16 && g.M % 256 == 0 && (g.a_rows_per_group <= 0 || g.a_rows_per_group % 256 == 0) &&
                   (g.c_rows_per_group <= 0 || g.c_rows_per_group % 256 == 0);
  if (big) { tiles_signed = -(int)((g.M / 256) * (g.N / 128)); lds = (size_t)2 * (256 + 128) * 64 * 2; }  // (the split epilogue stages 64 rows: 34 KB)
  else { tiles_signed = (int)t128; lds = (size_t)128 * (128 + 4) * 4; }  // the epilogue's fp32 staging tile (> the 64 KB of the two K stages)
  // (negative: 256-row tiles)
}

static int row_tail_fwd_impl(const mst_row_tail_args* args, const mst_gemm_args* rider, uint32_t* queue, mst_stream_t stream,
                             const TailShadow& shadow = TailShadow{}) {
  MST_CHECK_ARG(args != nullptr, "mst_row_tail_fwd: null args");
  const mst_row_tail_args& q = *args;
  MST_CHECK_ARG(q.B > 0 && q.B <= 64, "mst_row_tail_fwd: 1..64 rows (got %lld)", (long long)q.B);
  MST_CHECK_ARG(q.D == 128 || q.D == 256, "mst_row_tail_fwd: width must be 128 or 256 (got %lld)", (long long)q.D);
  MST_CHECK_ARG(q.att && q.resid && q.Wp && q.bp && q.g1 && q.be1 && q.W1 && q.b1 && q.W2 && q.b2 && q.g2 && q.be2 && q.h1 && q.x1 && q.a &&
                q.h2 && q.x2 && q.mean1 && q.rstd1 && q.mean2 && q.rstd2 && q.sync, "mst_row_tail_fwd: null pointer");
  MST_CHECK_ARG(q.rs_att % 8 == 0 && q.rs_res % 8 == 0 && q.rs_d % 8 == 0 && q.rs_a % 8 == 0 && q.ldwp % 8 == 0 && q.ldw1 % 8 == 0 &&
                q.ldw2 % 8 == 0, "mst_row_tail_fwd: strides must be multiples of 8 elements");
  MST_CHECK_ARG((((uintptr_t)q.att | (uintptr_t)q.resid | (uintptr_t)q.Wp | (uintptr_t)q.W1 | (uintptr_t)q.W2 | (uintptr_t)q.h1 | (uintptr_t)q.x1 |
                  (uintptr_t)q.a | (uintptr_t)q.h2 | (uintptr_t)q.x2 | (uintptr_t)q.bp | (uintptr_t)q.b1 | (uintptr_t)q.b2) % 16) == 0,
                "mst_row_tail_fwd: operands must be 16-byte aligned");
  MST_CHECK_ARG(q.dropout_p >= 0.f && q.dropout_p < 1.f && q.phys_stride > 0 && q.stat_stride > 0, "mst_row_tail_fwd: bad dropout / strides");
  if (rider) {
    const int rc = check_rider("mst_row_tail_fwd_ride", *rider, q.dtype);
    if (rc) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  const mst_gemm_args none = {};
  return dispatch_act(q.dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    if (rider) {
      int tiles;
      size_t rlds;
      ride_shape(*rider, tiles, rlds);
      if (shadow.n_groups > 0 && rlds < (size_t)16 * 32 * 33 * 4) rlds = (size_t)16 * 32 * 33 * 4;  // sixteen shadow tiles per ticket
      const int wi = (q.D == 256 ? 0 : 1) + (tiles < 0 ? 2 : 0);
      typedef void (*kern_t)(mst_row_tail_args, mst_gemm_args, int, uint32_t*, TailShadow);
      const kern_t fns[4] = {&row_tail_fwd_kernel<T, 256, 128>, &row_tail_fwd_kernel<T, 128, 128>, &row_tail_fwd_kernel<T, 256, 256>, &row_tail_fwd_kernel<T, 128, 256>};
      static size_t opted[4] = {0, 0, 0, 0};
      if (rlds > opted[wi]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[wi]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlds);
        if (e != hipSuccess) { set_error("row_tail_fwd_kernel (riders): LDS opt-in of %zu bytes: %s", rlds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
        opted[wi] = rlds;
      }
      const unsigned grid = ride_grid(q.D / 16, tiles < 0 ? -tiles : tiles);
      hipLaunchKernelGGL(fns[wi], dim3(grid), dim3(1024), rlds, s, q, *rider, tiles < 0 ? -tiles : tiles, queue, shadow);
    } else if (q.D == 256) hipLaunchKernelGGL((row_tail_fwd_kernel<T, 256>), dim3(16 * TAIL_OVERSUBSCRIBE), dim3(1024), 0, s, q, none, 0, (uint32_t*)nullptr, TailShadow{});
    else hipLaunchKernelGGL((row_tail_fwd_kernel<T, 128>), dim3(8 * TAIL_OVERSUBSCRIBE), dim3(1024), 0, s, q, none, 0, (uint32_t*)nullptr, TailShadow{});
    MST_CHECK_LAUNCH("row_tail_fwd_kernel");
    return MST_OK;
  });
}
extern "C" int mst_row_tail_fwd(const mst_row_tail_args* args, mst_stream_t stream) { return row_tail_fwd_impl(args, nullptr, nullptr, stream); }
extern "C" int mst_row_tail_fwd_ride(const mst_row_tail_args* args, const mst_gemm_args* rider, uint32_t* queue, mst_stream_t stream) {
  MST_CHECK_ARG(rider != nullptr && queue != nullptr, "mst_row_tail_fwd_ride: null rider / queue word");
  return row_tail_fwd_impl(args, rider, queue, stream);
}
extern "C" int mst_row_tail_fwd_ride_shadows(const mst_row_tail_args* args, const mst_gemm_args* rider, uint32_t* queue, int sh_dtype,
                                             const float* sh_w, void* sh_wt16, const int64_t* sh_desc, const int64_t* sh_prefix, int64_t sh_n_mat,
                                             int64_t sh_tiles, mst_stream_t stream) {
  MST_CHECK_ARG(args != nullptr && rider != nullptr && queue != nullptr, "mst_row_tail_fwd_ride_shadows: null args / rider / queue word");
  MST_CHECK_ARG(sh_w && sh_wt16 && sh_desc && sh_prefix && sh_n_mat > 0 && sh_tiles > 0 && sh_tiles < (1ll << 30) && sh_dtype == args->dtype,
                "mst_row_tail_fwd_ride_shadows: bad shadow list (mst_transpose_shadows' arguments, in the tail's activation type)");
  TailShadow sh;
  sh.w = sh_w; sh.wt16 = sh_wt16; sh.desc = sh_desc; sh.prefix = sh_prefix; sh.n_mat = (int)sh_n_mat; sh.tiles = sh_tiles;
  sh.n_groups = (int)((sh_tiles + 15) / 16);
  return row_tail_fwd_impl(args, rider, queue, stream, sh);
}

static int row_tail_bwd_impl(const mst_row_tail_bwd_args* args, const mst_gemm_args* rider, uint32_t* queue, mst_stream_t stream) {
  MST_CHECK_ARG(args != nullptr, "mst_row_tail_bwd: null args");
  const mst_row_tail_bwd_args& q = *args;
  MST_CHECK_ARG(q.B > 0 && q.B <= 64, "mst_row_tail_bwd: 1..64 rows (got %lld)", (long long)q.B);
  MST_CHECK_ARG(q.D == 128 || q.D == 256, "mst_row_tail_bwd: width must be 128 or 256 (got %lld)", (long long)q.D);
  MST_CHECK_ARG(q.dy && q.h2 && q.h1 && q.a && q.mean1 && q.rstd1 && q.mean2 && q.rstd2 && q.g1 && q.g2 && q.W2t && q.W1t && q.Wpt && q.dh &&
                q.dhm && q.dx1 && q.dh1m && q.dpre && q.dh1 && q.datt && q.dg1 && q.db1 && q.dg2 && q.db2 && q.sync, "mst_row_tail_bwd: null pointer");
  MST_CHECK_ARG(q.rs_dy % 8 == 0 && q.rs_d % 8 == 0 && q.rs_a % 8 == 0 && q.rs_c % 8 == 0 && q.rs_dpre % 8 == 0 && q.rs_dh1 % 8 == 0 &&
                q.rs_datt % 8 == 0 && q.ldw2t % 8 == 0 && q.ldw1t % 8 == 0 && q.ldwpt % 8 == 0, "mst_row_tail_bwd: strides must be multiples of 8 elements");
  MST_CHECK_ARG((((uintptr_t)q.dy | (uintptr_t)q.h2 | (uintptr_t)q.h1 | (uintptr_t)q.a | (uintptr_t)q.W2t | (uintptr_t)q.W1t | (uintptr_t)q.Wpt |
                  (uintptr_t)q.dh | (uintptr_t)q.dhm | (uintptr_t)q.dx1 | (uintptr_t)q.dh1m | (uintptr_t)q.dpre | (uintptr_t)q.dh1 | (uintptr_t)q.datt) % 16) == 0,
                "mst_row_tail_bwd: operands must be 16-byte aligned");
  MST_CHECK_ARG(q.dropout_p >= 0.f && q.dropout_p < 1.f && q.phys_stride > 0 && q.stat_stride > 0, "mst_row_tail_bwd: bad dropout / strides");
  if (rider) {
    const int rc = check_rider("mst_row_tail_bwd_ride", *rider, q.dtype);
    if (rc) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  const mst_gemm_args none = {};
  return dispatch_act(q.dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    size_t lds = (size_t)2 * 64 * ((size_t)q.D + 8) * 2 + (size_t)16 * 2 * q.D * 4 + (size_t)4 * 64 * 16 * 4 + (size_t)2 * q.D * 4;
    int tiles = 0;
    if (rider) {  // (a workgroup is a participant or a rider: one region)
      size_t rlds;
      ride_shape(*rider, tiles, rlds);
      if (rlds > lds) lds = rlds;
    }
    static size_t opted[6] = {64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024};
    const int wi = (q.D == 256 ? 0 : 1) + (rider ? (tiles < 0 ? 4 : 2) : 0);
    typedef void (*kern_t)(mst_row_tail_bwd_args, mst_gemm_args, int, uint32_t*);
    const kern_t fns[6] = {&row_tail_bwd_kernel<T, 256>, &row_tail_bwd_kernel<T, 128>, &row_tail_bwd_kernel<T, 256, 128>, &row_tail_bwd_kernel<T, 128, 128>,
                           &row_tail_bwd_kernel<T, 256, 256>, &row_tail_bwd_kernel<T, 128, 256>};
    if (lds > opted[wi]) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[wi]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("row_tail_bwd_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
      opted[wi] = lds;
    }
    if (rider) {
      const unsigned grid = ride_grid(q.D / 16, tiles < 0 ? -tiles : tiles);
      hipLaunchKernelGGL(fns[wi], dim3(grid), dim3(1024), lds, s, q, *rider, tiles < 0 ? -tiles : tiles, queue);
    } else if (q.D == 256) hipLaunchKernelGGL((row_tail_bwd_kernel<T, 256>), dim3(16 * TAIL_OVERSUBSCRIBE), dim3(1024), lds, s, q, none, 0, (uint32_t*)nullptr);
    else hipLaunchKernelGGL((row_tail_bwd_kernel<T, 128>), dim3(8 * TAIL_OVERSUBSCRIBE), dim3(1024), lds, s, q, none, 0, (uint32_t*)nullptr);
    MST_CHECK_LAUNCH("row_tail_bwd_kernel");
    return MST_OK;
  });
}
extern "C" int mst_row_tail_bwd(const mst_row_tail_bwd_args* args, mst_stream_t stream) { return row_tail_bwd_impl(args, nullptr, nullptr, stream); }
extern "C" int mst_row_tail_bwd_ride(const mst_row_tail_bwd_args* args, const mst_gemm_args* rider, uint32_t* queue, mst_stream_t stream) {
  MST_CHECK_ARG(rider != nullptr && queue != nullptr, "mst_row_tail_bwd_ride: null rider / queue word");
  return row_tail_bwd_impl(args, rider, queue, stream);
}

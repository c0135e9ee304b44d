// gemm_wgrad.hip — weight gradients: dW[N,K] += scale * sum_m A[m,n] * B[m,k], db[n] += sum_m A[m,n].
//
// Autograd counterpart (trainer.py:176 loss.backward()) of every Dense call site in
// VarAutoEncoder/transformer.py:36-40,65-68 and model.py:70-71,214-227: A = dY [M,N], B = X [M,K].
//
// Design (gfx950): the contraction index m is the row index of BOTH operands, i.e. both MFMA
// operands are needed "transposed". The tiles are staged row-major exactly as they sit in HBM
// (coalesced 16-byte loads) and consumed through ds_read_tr16_b64, which hands each lane 4
// consecutive m of one column — two of them form one 16x16x32 fragment. M is split across
// workgroups (≫256 of them) and partial tiles are accumulated with fp32 atomics into the flat
// gradient bucket, which the step zeroes once. Several layers' problems go into ONE launch
// (mst_wgrad_batch) so the split factor, and with it the atomic traffic, stays small.
#include <type_traits>
#include "common.hpp"
#include "partial_sums.hpp"
#include "outer_jobs.hpp"

namespace mst {

constexpr int WG_MAXP = 16;  // problems per launch (the whole backward pass of configs[1] is 15)
struct WgradBatch {
  int n;
  uint32_t narrow;                // bit p: problem p runs 256 x 128 tiles inside the 256 x 256 launch (K <= 128: half the MFMAs of a
                                  // 256 x 256 tile on the same bytes); the slot layout stays that of the wide tiles
  int split_p[WG_MAXP];           // M split of each problem: min(split, what its row count supports) — a 64-row problem
                                  // gets ONE slab instead of `split` workgroups of which all but one exit at once
  int64_t item_prefix[WG_MAXP + 1];  // prefix sums of tiles_p * split_p: work items of problem p are [prefix[p], prefix[p+1])
  float* partial;                 // optional scratch [items][BN*BKO]: the slabs' tiles go there instead of into dW by atomics
  mst_wgrad_args p[WG_MAXP];
  int64_t tile_prefix[WG_MAXP + 1];  // prefix sums of (tiles_n * tiles_k) per problem
};

constexpr int BMR = 64;  // m rows per LDS stage

// LDS tiles are row-major with rows padded by 16 elements (32 bytes). ds_read_tr16_b64 has each 16-lane group read
// 4 consecutive rows x 32 bytes; with 288-byte (or 160-byte) rows those four pieces fall on four different 32-byte
// bank windows, so the reads are conflict-free (unpadded 256-byte rows measured 75 % of LDS cycles as conflicts), and
// every fragment address of a stage is ONE per-lane base plus a compile-time offset. (An XOR swizzle of the chunk
// index is conflict-free too, but its offsets are not additive: the kernel spent 265 VALU instructions per 64-row
// stage and wave, twice the MFMA time, mostly on LDS and global address arithmetic.)
constexpr int LDS_PAD = 16;

__device__ __forceinline__ i16x4 tr_read(const void* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(uintptr_t)p);
}

// (Direct global -> LDS staging — global_load_lds_dwordx4 into an image of 1-KiB slots + 32 B of pad, two buffers, one stage
// ahead — was built and measured: stamps of one workgroup showed the MFMA phase of a stage going from 0.95 to 1.85 us while
// the LDS-DMA writes of the next stage land, 3.1 us per stage against 2.65 for the register-staged loop. Removed; what did
// pay is the interleaved register staging in wgrad_body.)
// diagnostic build only (-DMST_WGRAD_STAMPS): one workgroup leaves s_memtime stamps per stage (tools/bench_wgrad_stamps.py)
#ifdef MST_WGRAD_STAMPS
#ifndef MST_WGRAD_STAMP_WG
#define MST_WGRAD_STAMP_WG 64
#endif
__device__ uint64_t g_wgrad_stamps[4 + 64 * 4];
__device__ uint64_t g_wgrad_wg[512 * 2];  // [item] = realtime stamps (100 MHz) at the workgroup's start and end
#define WG_STAMP(slot) do { if (blockIdx.x == MST_WGRAD_STAMP_WG && threadIdx.x == 0 && (slot) < 4 + 64 * 4) g_wgrad_stamps[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG_STAMP(slot) do { } while (0)
#endif

// AU8: this problem's A operand is uint8 (mst_wgrad_args.a_u8). A template switch on purpose: the 256x256 form sits at
// exactly 256 VGPRs, and a run-time branch in the stage loop (an extra register array, or partial-register updates)
// cost EVERY problem of the launch (92 -> 160 us for the step's batch); as two bodies in one kernel the 16-bit problems
// keep their code and the two embedding-table problems take the other copy.
template <typename T, int BN, int BKO, int WGN, int WGK, bool AU8, int SLOT = BN * BKO>
__device__ __forceinline__ void wgrad_body(const WgradBatch& b, unsigned char* smem, int pi, int64_t item) {
  constexpr int NT = WGN * WGK * 64;
  constexpr int WTN = BN / WGN, WTK = BKO / WGK;
  constexpr int TN = WTN / 16, TK = WTK / 16;
  constexpr int A_CPR = BN / 8, B_CPR = BKO / 8;          // 16-byte chunks per tile row
  constexpr int A_CH = BMR * A_CPR / NT, B_CH = BMR * B_CPR / NT;
  static_assert(BMR * A_CPR % NT == 0 && BMR * B_CPR % NT == 0, "tile/threads mismatch");
  static_assert(NT % A_CPR == 0 && NT % B_CPR == 0, "column ownership must be loop-invariant");
  typedef typename Act<T>::vec8 vec8;

  constexpr int LDA_S = BN + LDS_PAD, LDB_S = BKO + LDS_PAD;  // LDS row strides in elements
  constexpr int A_RSTEP = NT / A_CPR, B_RSTEP = NT / B_CPR;   // row distance between a thread's consecutive chunks
  T* sA = reinterpret_cast<T*>(smem);          // [2][BMR][LDA_S]
  T* sB = sA + 2 * BMR * LDA_S;                // [2][BMR][LDB_S]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WGK, wk = wave % WGK;
  const mst_wgrad_args& a = b.p[pi];
  // (32-bit quotients throughout: items, tiles, M and K are below 2^31 — checked on the host — and a 64-bit division is ~200
  // instructions; ten of them stood between a workgroup's start and its first load)
  const uint32_t tiles_p = (uint32_t)(b.tile_prefix[pi + 1] - b.tile_prefix[pi]);
  const uint32_t in_p = (uint32_t)(item - b.item_prefix[pi]);
  const uint32_t split_u = in_p / tiles_p;
  const int64_t local = in_p - split_u * tiles_p;
  const int split_id = (int)split_u, split_here = b.split_p[pi];
  const uint32_t tiles_k = ((uint32_t)a.K + BKO - 1) / BKO;
  const uint32_t tile_n = (uint32_t)local / tiles_k;
  const int64_t n0 = (int64_t)tile_n * BN, k0 = (int64_t)((uint32_t)local - tile_n * tiles_k) * BKO;

  const int64_t m_chunk = (int64_t)((((uint32_t)a.M + (uint32_t)split_here - 1) / (uint32_t)split_here + BMR - 1) / BMR * BMR);
  const int64_t m_begin = (int64_t)split_id * m_chunk;
  const int64_t m_end = (m_begin + m_chunk < a.M) ? m_begin + m_chunk : a.M;
  if (m_begin >= m_end) return;  // uniform for the whole workgroup

  typedef typename std::conditional<AU8, uint8_t, T>::type TA;  // element of A as it sits in HBM
  typedef typename std::conditional<AU8, u32x2, u32x4>::type RA;  // one 8-element chunk of A in registers
  const TA* __restrict__ A = reinterpret_cast<const TA*>(a.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(a.B);
  const bool do_bias = (a.db != nullptr) && (k0 == 0);

  // chunk ownership: thread tid stages chunks (row a_r0 + i*A_RSTEP, columns a_c..a_c+7), i < A_CH — the column is
  // loop-invariant, so the edge test is one flag and the global pointer just advances by whole rows
  const int a_r0 = tid / A_CPR, a_c = (tid % A_CPR) * 8;
  const int b_r0 = tid / B_CPR, b_c = (tid % B_CPR) * 8;
  const bool a_ok = n0 + a_c < a.N, b_ok = k0 + b_c < a.K;
  const bool remap_a = a.a_rows_per_group > 0, remap_b = a.b_rows_per_group > 0;
  // interior tile of a problem without row remap: plain loads, no per-chunk select (uniform flag)
  const bool plain = !remap_a && !remap_b && n0 + BN <= a.N && k0 + BKO <= a.K;
  const int a_off = (int)(a_r0 * a.lda) + a_c, b_off = (int)(b_r0 * a.ldb) + b_c;  // per-thread element offsets

  RA ra[A_CH];
  u32x4 rb[B_CH];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  RA zeroA;
  if constexpr (AU8) zeroA = u32x2{0u, 0u}; else zeroA = zero4;

  // Row remaps (decoder rows 1..T of each sample): the physical row of the thread's first chunk and its offset inside
  // the group are carried from stage to stage, so the per-chunk mapping is an add and a compare. (Recomputing it with
  // 64-bit divisions per chunk and stage made the two remapped problems' workgroups finish at 95 us when the median
  // workgroup of the whole-step launch finished at 70.) Groups shorter than a stage keep the division.
  const bool div_a = remap_a && a.a_rows_per_group < BMR, div_b = remap_b && a.b_rows_per_group < BMR;
  int64_t pa0 = 0, pb0 = 0, oa0 = 0, ob0 = 0;
  if (remap_a) {
    const uint32_t m = (uint32_t)(m_begin + a_r0), g = m / (uint32_t)a.a_rows_per_group;
    oa0 = m - g * (uint32_t)a.a_rows_per_group;
    pa0 = (int64_t)g * a.a_group_stride + a.a_group_offset + oa0;  // remap_row
  }
  if (remap_b) {
    const uint32_t m = (uint32_t)(m_begin + b_r0), g = m / (uint32_t)a.b_rows_per_group;
    ob0 = m - g * (uint32_t)a.b_rows_per_group;
    pb0 = (int64_t)g * a.b_group_stride + a.b_group_offset + ob0;
  }

  auto load_tile = [&](int64_t mb) {
    const bool full = mb + BMR <= m_end;  // uniform: only the last stage of a slab can be partial
    if (plain && full) {
      // uniform row pointer (scalar registers) + the thread's 32-bit offset
#pragma unroll
      for (int i = 0; i < A_CH; ++i) ra[i] = *reinterpret_cast<const RA*>(A + (mb + i * A_RSTEP) * a.lda + n0 + a_off);
#pragma unroll
      for (int i = 0; i < B_CH; ++i) rb[i] = *reinterpret_cast<const u32x4*>(B + (mb + i * B_RSTEP) * a.ldb + k0 + b_off);
      return;
    }
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int64_t m = mb + a_r0 + i * A_RSTEP;
      int64_t pm = m;
      if (div_a) pm = remap_row(m, a.a_rows_per_group, a.a_group_stride, a.a_group_offset);
      else if (remap_a) pm = pa0 + i * A_RSTEP + ((oa0 + i * A_RSTEP >= a.a_rows_per_group) ? a.a_group_stride - a.a_rows_per_group : 0);
      ra[i] = (a_ok && m < m_end) ? *reinterpret_cast<const RA*>(A + pm * a.lda + n0 + a_c) : zeroA;
    }
    if (remap_a && !div_a) {
      oa0 += BMR; pa0 += BMR;
      if (oa0 >= a.a_rows_per_group) { oa0 -= a.a_rows_per_group; pa0 += a.a_group_stride - a.a_rows_per_group; }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int64_t m = mb + b_r0 + i * B_RSTEP;
      int64_t pm = m;
      if (div_b) pm = remap_row(m, a.b_rows_per_group, a.b_group_stride, a.b_group_offset);
      else if (remap_b) pm = pb0 + i * B_RSTEP + ((ob0 + i * B_RSTEP >= a.b_rows_per_group) ? a.b_group_stride - a.b_rows_per_group : 0);
      rb[i] = (b_ok && m < m_end) ? *reinterpret_cast<const u32x4*>(B + pm * a.ldb + k0 + b_c) : zero4;
    }
    if (remap_b && !div_b) {
      ob0 += BMR; pb0 += BMR;
      if (ob0 >= a.b_rows_per_group) { ob0 -= a.b_rows_per_group; pb0 += a.b_group_stride - a.b_rows_per_group; }
    }
  };
  T* const wA = sA + a_r0 * LDA_S + a_c;  // this thread's first chunk in buffer 0
  T* const wB = sB + b_r0 * LDB_S + b_c;
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      u32x4 v;
      if constexpr (AU8) v = expand_u8x8<T>(ra[i]); else v = ra[i];
      *reinterpret_cast<u32x4*>(wA + buf * BMR * LDA_S + i * A_RSTEP * LDA_S) = v;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) *reinterpret_cast<u32x4*>(wB + buf * BMR * LDB_S + i * B_RSTEP * LDB_S) = rb[i];
  };

  f32x4 acc[TN][TK];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TK; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of A: one more MFMA per A fragment against an all-ones B fragment (every output
  // column then holds the sum) on the waves of the first K tile — no VALU work at all
  const bool bias_wave = do_bias && wk == 0;
  f32x4 acc_b[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc_b[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  vec8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (T)1.f;

  // tr-read lane roles: lane i = lane&15 of 16-lane group g = lane>>4 supplies the address of one row, columns
  // 4*(i&3).., and receives column i. Which 8 of a k-step's 32 rows feed group g's k slots is free as long as A and B
  // agree; rows 4g..4g+3 (elements 0..3) and 16+4g..16+4g+3 (elements 4..7) make the two groups of a 32-lane half
  // cover all eight 32-byte bank windows of the padded rows (rows 8g.. had them collide pairwise: 33 % conflict cycles).
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pcol = (li & 3) * 4;
  const T* const fA = sA + (4 * g + q) * LDA_S + wn * WTN + pcol;  // + compile-time offsets below
  const T* const fB = sB + (4 * g + q) * LDB_S + wk * WTK + pcol;

  const int64_t nsteps = (m_end - m_begin + BMR - 1) / BMR;
  auto compute = [&](int cur) {
    const T* cA = fA + cur * BMR * LDA_S;
    const T* cB = fB + cur * BMR * LDB_S;
#pragma unroll
    for (int ms = 0; ms < BMR / 32; ++ms) {
      vec8 af[TN], bf[TK];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const i16x4 lo = tr_read(cA + (ms * 32) * LDA_S + j * 16);
        const i16x4 hi = tr_read(cA + (ms * 32 + 16) * LDA_S + j * 16);
        const i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[j] = __builtin_bit_cast(vec8, v);
      }
#pragma unroll
      for (int i = 0; i < TK; ++i) {
        const i16x4 lo = tr_read(cB + (ms * 32) * LDB_S + i * 16);
        const i16x4 hi = tr_read(cB + (ms * 32 + 16) * LDB_S + i * 16);
        const i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bf[i] = __builtin_bit_cast(vec8, v);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TK; ++i) acc[j][i] = Act<T>::mfma16(af[j], bf[i], acc[j][i]);
      if (bias_wave) {
#pragma unroll
        for (int j = 0; j < TN; ++j) acc_b[j] = Act<T>::mfma16(af[j], ones, acc_b[j]);
      }
    }
  };
  // One stage of global loads in flight, staged through registers into the other LDS buffer. (Two stages in flight —
  // a second register set, 256 VGPRs — measured the same 58 us on the 128x128 tiles: the stage time, ~2.2 us for 0.25 us
  // of MFMA work per wave, is not a latency that more loads in flight would hide; what did move it was fewer operand
  // bytes per FLOP — the 256x128 and 256x256 tiles of launch_wgrad.)
  // ---- interleaved register staging (slabs of whole 64-row stages; row remaps in groups of >= 64 rows).
  // Stamps of one workgroup of the whole-step launch (tools/bench_wgrad_stamps.py) showed the stage as a SUM of phases:
  // 0.89 us issuing the 8 loads of the next stage (the wave stalls on the memory pipeline's back-pressure while the MFMA
  // pipe idles — both waves of a SIMD do the same thing at the same time), 0.95 us of MFMAs, 0.41 us waiting for the loads
  // and writing them to LDS, 0.19 us barrier. Here the staging is cut into 8 pieces — { ds_write chunk p of stage t+1 (in
  // registers since the previous stage), reload the register with chunk p of stage t+2 } — placed between the 8 MFMA groups
  // of stage t: a stalled load issue of one wave overlaps the other wave's MFMAs, the loads get a whole stage to arrive, and
  // nothing but the barrier separates the last MFMA of a stage from the first of the next.
#ifndef MST_WGRAD_IL
#define MST_WGRAD_IL 1
#endif
  bool done_il = false;
  if constexpr (MST_WGRAD_IL && A_CH + B_CH <= TN * (BMR / 32)) {
    if ((m_end - m_begin) % BMR == 0 && !div_a && !div_b) {
      done_il = true;
      // every load is unconditional: a chunk beyond N / K reads the tile's first column instead and is zeroed when it is
      // written to LDS; past the last stage the loads repeat the last one (a conditional load costs a vmcnt(0) drain)
      const int a_cc = a_ok ? a_c : 0, b_cc = b_ok ? b_c : 0;
      int64_t pa = pa0, oa = oa0, pb = pb0, ob = ob0;  // remapped operands: physical row / offset in its group of the thread's first chunk
      int64_t mb_next = m_begin, loaded = 0;           // the stage the next loads fetch
      auto ld = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < A_CH) {
          constexpr int r = p * A_RSTEP;
          const int64_t pm = remap_a ? pa + r + ((oa + r >= a.a_rows_per_group) ? a.a_group_stride - a.a_rows_per_group : 0) : mb_next + a_r0 + r;
          ra[p] = *reinterpret_cast<const RA*>(A + pm * a.lda + n0 + a_cc);
        } else {
          constexpr int r = (p - A_CH) * B_RSTEP;
          const int64_t pm = remap_b ? pb + r + ((ob + r >= a.b_rows_per_group) ? a.b_group_stride - a.b_rows_per_group : 0) : mb_next + b_r0 + r;
          rb[p - A_CH] = *reinterpret_cast<const u32x4*>(B + pm * a.ldb + k0 + b_cc);
        }
      };
      auto advance = [&]() {  // (uniform)
        if (loaded + 1 < nsteps) {
          ++loaded; mb_next += BMR;
          if (remap_a) { oa += BMR; pa += BMR; if (oa >= a.a_rows_per_group) { oa -= a.a_rows_per_group; pa += a.a_group_stride - a.a_rows_per_group; } }
          if (remap_b) { ob += BMR; pb += BMR; if (ob >= a.b_rows_per_group) { ob -= a.b_rows_per_group; pb += a.b_group_stride - a.b_rows_per_group; } }
        }
      };
      auto st = [&](auto pc, int buf) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < A_CH) {
          u32x4 v;
          if constexpr (AU8) v = expand_u8x8<T>(ra[p]); else v = ra[p];
          *reinterpret_cast<u32x4*>(wA + buf * BMR * LDA_S + p * A_RSTEP * LDA_S) = a_ok ? v : zero4;
        } else {
          *reinterpret_cast<u32x4*>(wB + buf * BMR * LDB_S + (p - A_CH) * B_RSTEP * LDB_S) = b_ok ? rb[p - A_CH] : zero4;
        }
      };
      constexpr int NPC = A_CH + B_CH;  // pieces of a stage: one per MFMA row of the stage, as far as they go
      auto all_pieces = [&](auto&& f) {
        f(std::integral_constant<int, 0>()); f(std::integral_constant<int, 1>()); f(std::integral_constant<int, 2>()); f(std::integral_constant<int, 3>());
        if constexpr (NPC > 4) { f(std::integral_constant<int, 4>()); f(std::integral_constant<int, 5>()); }
        if constexpr (NPC > 6) { f(std::integral_constant<int, 6>()); f(std::integral_constant<int, 7>()); }
      };
      static_assert(NPC == 4 || NPC == 6 || NPC == 8, "piece list");
      WG_STAMP(0);
      all_pieces([&](auto pc) { ld(pc); });
      advance();
      all_pieces([&](auto pc) { st(pc, 0); });
      all_pieces([&](auto pc) { ld(pc); });
      advance();
      __syncthreads();
      WG_STAMP(1);
      for (int64_t t = 0; t < nsteps; ++t) {
        const int cur = (int)(t & 1);
        const T* cA = fA + cur * BMR * LDA_S;
        const T* cB = fB + cur * BMR * LDB_S;
        WG_STAMP(4 + (int)t * 4);
        auto half = [&](auto msc) {
          constexpr int ms = decltype(msc)::value;
          vec8 af[TN], bf[TK];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const i16x4 lo = tr_read(cA + (ms * 32) * LDA_S + j * 16);
            const i16x4 hi = tr_read(cA + (ms * 32 + 16) * LDA_S + j * 16);
            const i16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            af[j] = __builtin_bit_cast(vec8, w);
          }
#pragma unroll
          for (int i = 0; i < TK; ++i) {
            const i16x4 lo = tr_read(cB + (ms * 32) * LDB_S + i * 16);
            const i16x4 hi = tr_read(cB + (ms * 32 + 16) * LDB_S + i * 16);
            const i16x8 w = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bf[i] = __builtin_bit_cast(vec8, w);
          }
          auto group = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < TN) {
#pragma unroll
              for (int i = 0; i < TK; ++i) acc[j][i] = Act<T>::mfma16(af[j], bf[i], acc[j][i]);
              if (bias_wave) acc_b[j] = Act<T>::mfma16(af[j], ones, acc_b[j]);
              if constexpr (ms * TN + j < NPC) {
                __builtin_amdgcn_sched_barrier(0);
                st(std::integral_constant<int, ms * TN + j>(), cur ^ 1);
                ld(std::integral_constant<int, ms * TN + j>());
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          };
          static_assert(TN <= 4, "the group list covers four rows of MFMAs");
          group(std::integral_constant<int, 0>()); group(std::integral_constant<int, 1>());
          group(std::integral_constant<int, 2>()); group(std::integral_constant<int, 3>());
        };
        static_assert(BMR / 32 == 2, "two k-steps per stage");
        half(std::integral_constant<int, 0>());
        half(std::integral_constant<int, 1>());
        advance();
        WG_STAMP(4 + (int)t * 4 + 1);
        WG_STAMP(4 + (int)t * 4 + 2);
        __syncthreads();
        WG_STAMP(4 + (int)t * 4 + 3);
      }
      WG_STAMP(2);
    }
  }
  if (!done_il) {
  WG_STAMP(0);
  load_tile(m_begin);
  store_tile(0);
  __syncthreads();
  WG_STAMP(1);
  for (int64_t t = 0; t < nsteps; ++t) {
    const int cur = (int)(t & 1);
    if (t + 1 < nsteps) load_tile(m_begin + (t + 1) * BMR);
    WG_STAMP(4 + (int)t * 4);
    compute(cur);
    WG_STAMP(4 + (int)t * 4 + 1);
    if (t + 1 < nsteps) store_tile(cur ^ 1);
    WG_STAMP(4 + (int)t * 4 + 2);
    __syncthreads();
    WG_STAMP(4 + (int)t * 4 + 3);
  }
  WG_STAMP(2);
  }

  // D[row = n (4*(lane>>4)+r)][col = k (lane&15)].
  if (b.partial) {
    // Two-pass reduction: this M-slab's tile goes to its own slot of the scratch buffer with plain stores and
    // wgrad_reduce_kernel sums the slots in a fixed order. fp32 atomics on dW cost ~48 us of the 120 us whole-step
    // launch (4 N K split = 62 MB at the ~1.3 TB/s the chip adds atomically) and made the gradients depend on
    // arrival order in the last bits.
    // The accumulators sit 4 rows x 64 bytes per wave-instruction; stored like that (or added atomically like that)
    // the 256 KiB tile leaves the CU in 64-byte pieces and the launch spends ~50 us on it. Instead each group of WTN
    // rows is transposed through LDS (the stage buffers are dead) and leaves as whole 1-KiB rows, 16 bytes per lane.
    float* slot = b.partial + item * (int64_t)SLOT;  // (rows of BKO floats; a narrower form uses the front of its slot)
    constexpr int LDF = BKO + 4;
    static_assert((size_t)WTN * LDF * 4 <= (size_t)2 * BMR * (BN + BKO + 2 * LDS_PAD) * 2, "row-group staging must fit the stage buffers");
    float* sF = reinterpret_cast<float*>(smem);
    for (int pass = 0; pass < WGN; ++pass) {
      if (wn == pass) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TK; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) sF[(j * 16 + 4 * g + r) * LDF + wk * WTK + i * 16 + li] = acc[j][i][r];
      }
      __syncthreads();
      for (int c = tid; c < WTN * (BKO / 4); c += NT) {
        const int row = c / (BKO / 4), c4 = (c % (BKO / 4)) * 4;
        *reinterpret_cast<f32x4*>(slot + (pass * WTN + row) * BKO + c4) = *reinterpret_cast<const f32x4*>(sF + row * LDF + c4);
      }
      __syncthreads();
    }
  } else
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int i = 0; i < TK; ++i) {
      const int64_t k = k0 + wk * WTK + i * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = n0 + wn * WTN + j * 16 + 4 * g + r;
        if (n < a.N && k < a.K) atomicAdd(a.dW + n * a.ldw + k, acc[j][i][r] * a.scale);
      }
    }
  }

  if (bias_wave && li == 0) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = n0 + wn * WTN + j * 16 + 4 * g + r;
        if (n < a.N) atomicAdd(a.db + n, acc_b[j][r] * a.scale);
      }
  }
  WG_STAMP(3);
}

template <typename T, int BN, int BKO, int WGN, int WGK>
__global__ __launch_bounds__(WGN * WGK * 64) void wgrad_kernel(WgradBatch b) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // locate (problem, slab, tile)
  // Workgroup ids are dealt round-robin to the 8 XCDs. Work items are numbered problem, then M-slab, then tile, and
  // every XCD takes a contiguous eighth of them: the tiles of a problem that read the same rows then mostly share
  // one XCD's L2, and the operands cross the fabric about once (117 MB for the encoder layer's four problems)
  // instead of once per XCD that owns a tile needing them (312 MB with the tile-major order).
  const int64_t n_items = b.item_prefix[b.n];
  const int64_t per_xcd = (n_items + 7) >> 3;
  const int64_t item = (int64_t)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if (item >= n_items) return;
#ifdef MST_WGRAD_STAMPS
  if (threadIdx.x == 0 && item < 512) g_wgrad_wg[2 * item] = __builtin_amdgcn_s_memrealtime();
#endif
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXP; ++i)
    if (i < b.n && item >= b.item_prefix[i]) pi = i;
  bool done = false;
  if constexpr (BN == 256 && BKO == 256) {
    // (the whole-step form: problems with K <= 128 run 256 x 128 tiles, in the same launch and the same slot layout)
    if ((b.narrow >> pi) & 1u) {
      if (b.p[pi].a_u8) wgrad_body<T, 256, 128, WGN, WGK, true, 256 * 256>(b, smem, pi, item);
      else wgrad_body<T, 256, 128, WGN, WGK, false, 256 * 256>(b, smem, pi, item);
      done = true;
    }
  }
  if (!done) {
    if (b.p[pi].a_u8) wgrad_body<T, BN, BKO, WGN, WGK, true>(b, smem, pi, item);  // (uniform for the workgroup)
    else wgrad_body<T, BN, BKO, WGN, WGK, false>(b, smem, pi, item);
  }
#ifdef MST_WGRAD_STAMPS
  if (threadIdx.x == 0 && item < 512) g_wgrad_wg[2 * item + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// dW[n, k] += scale * sum over the M-slabs (in slab order) of the tiles wgrad_kernel left in the scratch buffer.
// grid = (tile elements / 1024, tiles); a thread owns 4 consecutive k of one n.
template <int BN, int BKO>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradBatch b, PartialSumBatch ps, OuterBatch ob) {
  if (blockIdx.y >= (unsigned)b.tile_prefix[b.n]) {  // rows of workgroups past the tiles: the caller's column-sum / outer-product jobs
    __shared__ f32x4 red[16][16];
    const int wg = (int)((blockIdx.y - (unsigned)b.tile_prefix[b.n]) * gridDim.x + blockIdx.x);
    if (wg < ps.wg_prefix[ps.n]) partial_sums_wg(ps, wg, red);
    else if (wg - ps.wg_prefix[ps.n] < ob.wg_prefix[ob.n]) outer_jobs_wg(ob, wg - ps.wg_prefix[ps.n], reinterpret_cast<float(*)[64]>(&red[0][0]));
    return;
  }
  const int64_t tile_lin = blockIdx.y;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXP; ++i)
    if (i < b.n && tile_lin >= b.tile_prefix[i]) pi = i;
  const mst_wgrad_args& a = b.p[pi];
  const int64_t local = tile_lin - b.tile_prefix[pi];
  const int64_t tiles_p = b.tile_prefix[pi + 1] - b.tile_prefix[pi];
  const bool nar = (b.narrow >> pi) & 1u;
  const int bk = nar ? 128 : BKO;  // this problem's tile width in k (rows of the slot are bk floats long)
  // (shifts and 32-bit quotients: this kernel's threads do eleven loads each, and the 64-bit divisions of the tile decode were
  // most of its instructions)
  const int bk_log = nar ? 7 : (BKO == 256 ? 8 : BKO == 128 ? 7 : 6);
  static_assert(BKO == 256 || BKO == 128 || BKO == 64, "tile width in k must be a power of two");
  const uint32_t tiles_k = ((uint32_t)a.K + bk - 1) >> bk_log;
  const uint32_t tile_n = (uint32_t)local / tiles_k;
  const int64_t n0 = (int64_t)tile_n * BN, k0 = (int64_t)((uint32_t)local - tile_n * tiles_k) << bk_log;
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;  // element of the tile
  if (e >= BN * bk) return;
  const int nl = e >> bk_log, kl = e & (bk - 1);
  const int64_t n = n0 + nl, k = k0 + kl;
  if (n >= a.N || k >= a.K) return;
  // slabs that own rows of this problem (wgrad_kernel returns early, writing nothing, for the others)
  const int sp = b.split_p[pi];
  const int64_t m_chunk = (int64_t)((((uint32_t)a.M + (uint32_t)sp - 1) / (uint32_t)sp + BMR - 1) / BMR * BMR);
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < sp && (int64_t)s * m_chunk < a.M; ++s)
    sum += *reinterpret_cast<const f32x4*>(b.partial + (b.item_prefix[pi] + (int64_t)s * tiles_p + local) * (BN * BKO) + e);
  float* d = a.dW + n * a.ldw + k;
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (k + c < a.K) d[c] += sum[c] * a.scale;
}

static_assert(sizeof(WgradBatch) + sizeof(PartialSumBatch) + sizeof(OuterBatch) <= 4096, "kernel arguments of the reduction pass");

// *ps_done: the column-sum jobs were taken along by the reduction pass
template <typename T>
static int launch_wgrad(const WgradBatch& b, int big, const PartialSumBatch* ps, bool* ps_done, hipStream_t s, const OuterBatch& ob) {
  const int64_t total = cdiv(b.item_prefix[b.n], 8) * 8;  // padded to whole XCD rounds (see the kernel)
  if (big == 3) {
    constexpr int BN = 256, BKO = 256;
    size_t lds = (size_t)2 * BMR * (BN + BKO + 2 * LDS_PAD) * 2;
    static bool opted = false;
    if (!opted) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BN, BKO, 4, 2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("wgrad_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
      opted = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<T, BN, BKO, 4, 2>), dim3((unsigned)total), dim3(512), lds, s, b);
    if (b.partial) {
      MST_CHECK_LAUNCH("wgrad_kernel");
      constexpr unsigned GX = BN * BKO / 1024;
      PartialSumBatch none;
      none.n = 0;
      for (int i = 0; i <= PS_MAXJ; ++i) none.wg_prefix[i] = 0;
      const PartialSumBatch& pb = ps ? *ps : none;
      const unsigned extra = (unsigned)cdiv(pb.wg_prefix[pb.n] + ob.wg_prefix[ob.n], GX);
      hipLaunchKernelGGL((wgrad_reduce_kernel<BN, BKO>), dim3(GX, (unsigned)b.tile_prefix[b.n] + extra), dim3(256), 0, s, b, pb, ob);
      *ps_done = true;  // (the column sums and the outer products)
    }
  } else if (big == 2) {
    constexpr int BN = 256, BKO = 128;
    size_t lds = (size_t)2 * BMR * (BN + BKO + 2 * LDS_PAD) * 2;
    static bool opted = false;
    if (!opted) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BN, BKO, 4, 2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("wgrad_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
      opted = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<T, BN, BKO, 4, 2>), dim3((unsigned)total), dim3(512), lds, s, b);
  } else if (big) {
    constexpr int BN = 128, BKO = 128;
    size_t lds = (size_t)2 * BMR * (BN + BKO + 2 * LDS_PAD) * 2;
    static bool opted = false;  // 73.7 KB of dynamic LDS: above the 64 KB default
    if (!opted) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BN, BKO, 2, 2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("wgrad_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
      opted = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<T, BN, BKO, 2, 2>), dim3((unsigned)total), dim3(256), lds, s, b);
  } else {
    constexpr int BN = 64, BKO = 64;
    size_t lds = (size_t)2 * BMR * (BN + BKO + 2 * LDS_PAD) * 2;
    hipLaunchKernelGGL((wgrad_kernel<T, BN, BKO, 2, 2>), dim3((unsigned)total), dim3(256), lds, s, b);
  }
  MST_CHECK_LAUNCH("wgrad_kernel");
  return MST_OK;
}

static int check_wgrad(const mst_wgrad_args& a) {
  MST_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "mst_gemm_wgrad: M,N,K must be positive");
  MST_CHECK_ARG(a.M < (1ll << 30) && a.N < (1ll << 30) && a.K < (1ll << 30) && a.a_rows_per_group < (1ll << 30) && a.b_rows_per_group < (1ll << 30),
                "mst_gemm_wgrad: M, N, K and the remap group sizes must stay below 2^30 (the kernels' tile decode is 32-bit)");
  // N and K may be ragged as long as the 16-byte chunk that straddles the edge stays inside the row
  // (callers keep pad columns zero); outputs beyond N / K are never written.
  MST_CHECK_ARG(a.lda % 8 == 0 && a.ldb % 8 == 0 && a.lda >= roundup(a.N, 8) && a.ldb >= roundup(a.K, 8),
                "mst_gemm_wgrad: lda/ldb must be multiples of 8 and >= roundup8(N)/roundup8(K) (got N=%lld K=%lld lda=%lld ldb=%lld)",
                (long long)a.N, (long long)a.K, (long long)a.lda, (long long)a.ldb);
  MST_CHECK_ARG(a.A && a.B && a.dW, "mst_gemm_wgrad: null operand");
  MST_CHECK_ARG(a.ldw >= a.K, "mst_gemm_wgrad: ldw < K");
  MST_CHECK_ARG(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.B % 16 == 0), "mst_gemm_wgrad: operands must be 16-byte aligned");
  return MST_OK;
}

}  // namespace mst

using namespace mst;

extern "C" int mst_gemm_wgrad_batch_ws(const mst_wgrad_args* list, int n, float* scratch, int64_t scratch_bytes,
                                       mst_stream_t stream) {
  return mst_gemm_wgrad_batch_sums(list, n, scratch, scratch_bytes, nullptr, 0, stream);
}

extern "C" int mst_gemm_wgrad_batch_sums(const mst_wgrad_args* list, int n, float* scratch, int64_t scratch_bytes,
                                         const mst_partial_sum* sums, int n_sums, mst_stream_t stream) {
  return mst_gemm_wgrad_batch_flush(list, n, scratch, scratch_bytes, sums, n_sums, nullptr, 0, stream);
}

extern "C" int mst_gemm_wgrad_batch_flush(const mst_wgrad_args* list, int n, float* scratch, int64_t scratch_bytes,
                                          const mst_partial_sum* sums, int n_sums, const mst_outer_job* outers, int n_outers,
                                          mst_stream_t stream) {
  MST_CHECK_ARG(list != nullptr && n >= 1 && n <= WG_MAXP, "mst_gemm_wgrad_batch: need 1..%d problems", WG_MAXP);
  OuterBatch ob;
  {
    int rc = pack_outer_jobs(outers, n_outers, ob);
    if (rc) return rc;
  }
  MST_CHECK_ARG(n_sums >= 0 && (n_sums == 0 || sums != nullptr), "mst_gemm_wgrad_batch_sums: bad column-sum job list");
  PartialSumBatch ps;
  if (n_sums > 0) {
    int rc = pack_partial_sums(sums, n_sums, ps);
    if (rc) return rc;
  }
  MST_CHECK_ARG(!scratch || ((uintptr_t)scratch % 16 == 0 && scratch_bytes > 0), "mst_gemm_wgrad_batch_ws: bad scratch buffer");
  WgradBatch b;
  b.n = n;
  int64_t out_elems = 0, maxM = 0;
  for (int i = 0; i < n; ++i) {
    int rc = check_wgrad(list[i]);
    if (rc) return rc;
    MST_CHECK_ARG(list[i].dtype == list[0].dtype, "mst_gemm_wgrad_batch: mixed dtypes");
    b.p[i] = list[i];
    out_elems += list[i].N * list[i].K;
    if (list[i].M > maxM) maxM = list[i].M;
  }
  // tile size: 128x128 when that still leaves enough tiles to fill the chip at a modest split
  // Tile shape by total output size (enough tiles for one resident round at a split of a few): the kernel is bound
  // by operand traffic into the CUs, so bytes per FLOP decide: 64x64 tiles (4 waves, up to 5 workgroups per CU),
  // 128x128 (4 waves, 2 per CU), 256x128 and 256x256 (8 waves, 1 per CU). Whole step of configs[1] (2.2 M outputs):
  // 1.000 / 0.982 / 0.971 ms per step with 128x128 / 256x128 / 256x256.
  int big = 0;
  if (out_elems >= (int64_t)256 * 256 * 24) big = 3;
  else if (out_elems >= (int64_t)256 * 128 * 36) big = 2;  // (a 0.79 M-output batch measured 58 us with 128x128, 62 with 256x128)
  else if (out_elems >= (int64_t)128 * 128 * 24) big = 1;
  // (128 x 128 tiles with at most TWO M-slabs per tile added into dW by fp32 atomics — deterministic, because dW starts at zero and
  // fl(0 + s1) + s2 == fl(0 + s2) + s1; no slab scratch, no reduction pass — measured 170 us against 70 + 17: with one 4-wave
  // workgroup per CU a 64-row stage is a dependent chain of ~1.3 us whatever the tile size. docs/kernel_notes.md, round 4)
  const int bn = big >= 2 ? 256 : (big ? 128 : 64), bk = big == 3 ? 256 : (big ? 128 : 64);
  static const bool mixed = !(getenv("MST_WGRAD_MIXED") && getenv("MST_WGRAD_MIXED")[0] == '0');
  b.tile_prefix[0] = 0;
  int bk_p[WG_MAXP];
  b.narrow = 0u;
  for (int i = 0; i < n; ++i) {
    bk_p[i] = (big == 3 && mixed && list[i].K <= 128) ? 128 : bk;
    if (bk_p[i] != bk) b.narrow |= 1u << i;
    b.tile_prefix[i + 1] = b.tile_prefix[i] + cdiv(list[i].N, bn) * cdiv(list[i].K, bk_p[i]);
  }
  for (int i = n + 1; i <= WG_MAXP; ++i) b.tile_prefix[i] = b.tile_prefix[n];
  const int64_t tiles = b.tile_prefix[n];
  // One full resident round of workgroups, no more (the workgroup that does not fit starts a second round: 105 vs
  // 83 us). Every problem is split min(S, rows / 128) ways — a 64-row problem (the top encoder layer's position-0 path)
  // gets one slab, where a batch-wide split left 26 % of the launch's workgroups with nothing to do — and S is the
  // largest value for which the items still fit.
  // A stage of a narrower (256 x 128) tile is cheaper, so its problems get proportionally FEWER slabs (longer ones): the
  // workgroups of both kinds then finish together, and the wide tiles' slabs get shorter.
  const int64_t slots = big >= 2 ? 256 : (big ? 512 : 1024);
  auto split_of = [&](int i, int64_t S) -> int64_t {
    const int64_t cap = cdiv(list[i].M, 2 * BMR);
#ifndef MST_WGRAD_NARROW_SPLIT
#define MST_WGRAD_NARROW_SPLIT 1
#endif
    // (per-workgroup stamps, tools/bench_wgrad_wgs.py: a narrow tile's stage costs ~0.93 us, a wide one's ~2.0)
    int64_t sp = (bk_p[i] < bk) ? (MST_WGRAD_NARROW_SPLIT == 0 ? S : MST_WGRAD_NARROW_SPLIT == 1 ? (S * 5 + 4) / 8 : S / 2) : S;
    if (sp < 1) sp = 1;
    return sp < cap ? sp : cap;
  };
  int64_t split = 1, n_items = 0;
  for (int64_t S = cdiv(maxM, 2 * BMR); S >= 1; --S) {
    int64_t items = 0;
    for (int i = 0; i < n; ++i) items += (b.tile_prefix[i + 1] - b.tile_prefix[i]) * split_of(i, S);
    if (items <= slots || S == 1) { split = S; n_items = items; break; }
  }
  b.item_prefix[0] = 0;
  for (int i = 0; i < n; ++i) {
    b.split_p[i] = (int)split_of(i, split);
    b.item_prefix[i + 1] = b.item_prefix[i] + (b.tile_prefix[i + 1] - b.tile_prefix[i]) * b.split_p[i];
  }
  for (int i = n; i < WG_MAXP; ++i) b.split_p[i] = 1;
  for (int i = n + 1; i <= WG_MAXP; ++i) b.item_prefix[i] = b.item_prefix[n];
  // two-pass reduction through the caller's scratch buffer when it is big enough (256x256 tiles only: that is where
  // 4 N K split bytes of atomics are tens of microseconds)
  b.partial = nullptr;
  if (scratch && big == 3 && n_items * (int64_t)(256 * 256) * 4 <= scratch_bytes) b.partial = scratch;
  hipStream_t s = (hipStream_t)stream;
  bool ps_done = false;
  int rc = dispatch_act(list[0].dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    return launch_wgrad<T>(b, big, n_sums > 0 ? &ps : nullptr, &ps_done, s, ob);
  });
  if (rc == MST_OK && n_sums > 0 && !ps_done) rc = mst_partial_sums(sums, n_sums, stream);  // no reduction pass: own launch
  if (rc == MST_OK && n_outers > 0 && !ps_done) rc = mst_outer_jobs(outers, n_outers, stream);
  return rc;
}

extern "C" int mst_gemm_wgrad_batch(const mst_wgrad_args* list, int n, mst_stream_t stream) {
  return mst_gemm_wgrad_batch_ws(list, n, nullptr, 0, stream);
}

extern "C" int mst_gemm_wgrad(const mst_wgrad_args* args, mst_stream_t stream) {
  MST_CHECK_ARG(args != nullptr, "mst_gemm_wgrad: null args");
  return mst_gemm_wgrad_batch(args, 1, stream);
}

#ifdef MST_WGRAD_STAMPS
extern "C" int mst_debug_wgrad_wg(uint64_t* host_out) {  // diagnostic builds only: 512 x (start, end) realtime stamps
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mst::g_wgrad_wg), sizeof(uint64_t) * 1024) == hipSuccess ? 0 : -1;
}
extern "C" int mst_debug_wgrad_stamps(uint64_t* host_out) {  // diagnostic builds only
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mst::g_wgrad_stamps), sizeof(uint64_t) * (4 + 64 * 4)) == hipSuccess ? 0 : -1;
}
#endif

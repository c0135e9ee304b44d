// gemm_tile.hpp — the tile-level device code of the NT GEMM kernels (gemm_nt.hip): the K loop of one output tile and its epilogue.
// A header so that launches of OTHER kernels can run GEMM tiles as extra workgroups (row_tail.hip: the idle XCDs of the
// one-launch position-0 tails take the decoder's K | Q | V projection and its input gradient off the step's dependent chain).
#pragma once
#include <math.h>
#include <type_traits>
#include "common.hpp"

namespace mst {

// K depth of one LDS tile: template parameter BK (elements), CHUNKS = BK / 8 16-byte chunks per tile row. 64 for the
// big launches; 256 for the skinny ones (M <= 64, K >= 512: a handful of workgroups whose time is the number of
// dependent K tiles, each one exposed memory round trip).

// Shared epilogue of the GEMM kernels (called after a workgroup barrier: `smem` is free to reuse).
// ROWOPS: the row-indexed adds (rowadd / grpadd: the two embedding GEMMs of a step) AND the A / C row remaps (their 64-bit
// divisions) are compiled in. They are a template
// switch, not a run-time one, because the launch-floor-bound GEMMs of the step (M = 64: four workgroups, every
// instruction line a cold fetch) measurably pay for code they jump over: +0.4 ... +2.6 us per launch with the row-op code
// present in the one kernel, against -10 us on the embedding GEMM that uses it.
// PATH, likewise: 1 = every tile of the launch is interior and eligible for the fast row loop below (the host checks:
// gemm_fast_eligible), 2 = the guarded general loop only. Two small kernels instead of one with both bodies.
// DROP, likewise: dropout / self_resid compiled in (half of the step's GEMM launches are gradient GEMMs without it).
template <typename T, int BM, int BN, int WGM, int WGN, bool C_F32, bool ROWOPS, int PATH, bool DROP>
__device__ __forceinline__ void gemm_epilogue(const mst_gemm_args& a, unsigned char* smem,
                                              f32x4 (&acc)[(BN / WGN) / 16][(BM / WGM) / 16], int64_t m0, int64_t n0,
                                              const float (&bias_pre)[8] /* gemm_bias_preload's */) {
  constexpr int NT = WGM * WGN * 64;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int frow = lane & 15, fq = lane >> 4;
  // ------------------------------------------------------------------ epilogue
  // The accumulators go to LDS as fp32 (one wave-row group of the tile per pass) and every thread then finishes
  // 8 consecutive columns of a row at a time: bias / residual / gate / output move as 16-byte, row-contiguous
  // accesses. The element loop is branch-free (absent features are neutral constants: bias 0, ReLU floor -inf,
  // residual 0, gate 1); optional features cost one wave-uniform branch per 8-column chunk, and only tiles that
  // cross the M or N edge take the guarded path. (Finishing elements in accumulator layout with per-element
  // feature tests made the kernel issue-bound: ~2400 VALU per 128 MFMA.)
  const int n_store = (int)(((a.N + 3) / 4 * 4) < a.ldc ? ((a.N + 3) / 4 * 4) : a.ldc);
  const float inv_keep = dropout_inv_keep(a.dropout_p);
  const uint64_t dseed = a.dropout_seed ^ ((a.dropout_p > 0.f && a.dropout_seed_ptr) ? a.dropout_seed_ptr[0] : 0ull);
  const uint32_t dkey = dropout_key(dseed, a.dropout_site), dthr = dropout_thr(a.dropout_p);
  constexpr int LDS_F = BN + 4;   // fp32 row stride: rows stay 16-byte aligned, banks are spread
  constexpr int CPR = BN / 8;     // 8-column chunks per tile row
  static_assert(NT % CPR == 0, "a thread must keep its column chunk across rows");
  float* sF = reinterpret_cast<float*>(smem);  // the K-loop's tiles are dead: every wave passed the loop's last barrier
  const int ch = tid % CPR;
  const int nc = (int)n0 + ch * 8;             // first column of this thread's chunk (N < 2^31)
  const int N32 = (int)a.N;
  const bool edge = PATH != 1 && PATH != 4 && ((m0 + BM > a.M) || ((int)n0 + BN > N32) || (a.ldc % 8 != 0) ||
                                  (a.resid && ((a.ldr % 8 != 0) || ((uintptr_t)a.resid % 16 != 0))) ||
                                  (a.gate && ((a.ldg % 8 != 0) || ((uintptr_t)a.gate % 16 != 0))));
  const bool has_drop = DROP && a.dropout_p > 0.f;
  const bool has_rowops = ROWOPS && (a.rowadd || a.grpadd);
  // row-indexed adds (positional table row m % period, class row grp_index[m / period]) ride on the fast path when a
  // tile cannot straddle a period: the class row is then one per tile and the positional rows advance with the tile rows
  const bool fast = PATH == 1 || PATH == 4;
  constexpr bool SPLIT = PATH == 4;  // stage one wave-row block per pass: (BM / WGM) x (BN + 4) floats of LDS instead of BM x (BN + 4)
  float ga8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) ga8[e] = 0.f;
  if (ROWOPS && fast && a.grpadd) {
    const float* gp8 = a.grpadd + (int64_t)a.grp_index[m0 / a.rowadd_period] * a.ldga + nc;
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp8), g1 = *reinterpret_cast<const f32x4*>(gp8 + 4);
    ga8[0] = g0[0]; ga8[1] = g0[1]; ga8[2] = g0[2]; ga8[3] = g0[3]; ga8[4] = g1[0]; ga8[5] = g1[1]; ga8[6] = g1[2]; ga8[7] = g1[3];
  }
  const bool relu = a.act == MST_ACT_RELU;
  const float alpha = a.alpha;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = bias_pre[e];
  const T* resid = reinterpret_cast<const T*>(a.resid);
  const T* gate = reinterpret_cast<const T*>(a.gate);

  // every wave stages its accumulators at once (the launch sizes LDS for the whole BM x (BN+4) fp32 tile): one barrier
  // per workgroup instead of two per wave-row pass
  if constexpr (!SPLIT) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        *reinterpret_cast<f32x4*>(sF + (wm * WTM + i * 16 + frow) * LDS_F + wn * WTN + j * 16 + fq * 4) = acc[j][i];
    __syncthreads();
  }
  for (int pass = 0; pass < WGM; ++pass) {
    if constexpr (SPLIT) {
      if (pass > 0) __syncthreads();  // the previous block's readers are done
      if (wm == pass) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            *reinterpret_cast<f32x4*>(sF + (i * 16 + frow) * LDS_F + wn * WTN + j * 16 + fq * 4) = acc[j][i];
      }
      __syncthreads();
    }
    const float* sFp = sF + (SPLIT ? 0 : pass * WTM * LDS_F);  // this row block of the staged tile
    // Fast path (interior tile, 16-bit output, no row remap / row-indexed adds, < 2^32 output elements): the row
    // loop carries pointers and a 32-bit dropout counter forward by constant strides. At two waves per SIMD the
    // epilogue is VALU-bound (measured per workgroup: 7.7 us of a 13.9 us life in FFN1, most of it 64-bit address
    // and counter arithmetic per 8-column chunk).
    if constexpr (PATH == 1 || PATH == 4) {
      constexpr int RSTEP = NT / CPR;
      const int row0 = tid / CPR;
      const int64_t mf = m0 + pass * WTM + row0;
      // a C row remap whose groups are whole tiles moves the tile as a block: physical row = remap(m0) + (m - m0)
      const int64_t pmf = (ROWOPS ? remap_row(m0, a.c_rows_per_group, a.c_group_stride, a.c_group_offset) : m0) + pass * WTM + row0;
      T* cp = reinterpret_cast<T*>(a.C) + pmf * a.ldc + nc;
      // (the residual is indexed by the LOGICAL row — a strided view remaps it by its stride — unless resid_phys says it shares C's rows)
      const T* rp = resid ? resid + ((ROWOPS && a.resid_phys) ? pmf : mf) * a.ldr + nc : nullptr;
      const T* gp = gate ? gate + mf * a.ldg + nc : nullptr;
      uint32_t w = (uint32_t)((uint64_t)(pmf * a.N + nc) >> 2);  // counter = PHYSICAL output row
      const uint32_t wstep = (uint32_t)((uint64_t)(RSTEP * a.N) >> 2);
      const float* sp = sFp + row0 * LDS_F + ch * 8;
      constexpr int ITERS = WTM / RSTEP;
      // every LDS read and every residual / gate load of the thread's chunks is issued before the first chunk is
      // finished (at two waves per SIMD a chunk-by-chunk loop exposes one LDS + one global round trip per chunk)
      f32x4 v0[ITERS], v1[ITERS], ra0[ITERS], ra1[ITERS];
      u32x4 rv[ITERS], gv[ITERS];
      const float* rap = (ROWOPS && a.rowadd) ? a.rowadd + (m0 % a.rowadd_period + pass * WTM + row0) * a.ldra + nc : nullptr;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        v0[it] = *reinterpret_cast<const f32x4*>(sp + it * RSTEP * LDS_F);
        v1[it] = *reinterpret_cast<const f32x4*>(sp + it * RSTEP * LDS_F + 4);
        if (rp) rv[it] = *reinterpret_cast<const u32x4*>(rp + (int64_t)it * RSTEP * a.ldr);
        if (gp) gv[it] = *reinterpret_cast<const u32x4*>(gp + (int64_t)it * RSTEP * a.ldg);
        if (rap) {
          ra0[it] = *reinterpret_cast<const f32x4*>(rap + (int64_t)it * RSTEP * a.ldra);
          ra1[it] = *reinterpret_cast<const f32x4*>(rap + (int64_t)it * RSTEP * a.ldra + 4);
        }
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        float t[8] = {v0[it][0], v0[it][1], v0[it][2], v0[it][3], v1[it][0], v1[it][1], v1[it][2], v1[it][3]};
        if (ROWOPS && a.grpadd) {
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] += ga8[e];
        }
        if (a.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] += bias8[e];
        }
        if (alpha != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] *= alpha;
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = fmaxf(t[e], 0.f);
        }
        if (DROP && (has_drop || a.self_resid)) {
          float u0[4] = {t[0], t[1], t[2], t[3]}, u1[4] = {t[4], t[5], t[6], t[7]};
          if (has_drop) {
            dropout_apply4_32(dkey, w, dthr, inv_keep, u0);
            dropout_apply4_32(dkey, w + 1, dthr, inv_keep, u1);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            t[e] = a.self_resid ? t[e] + u0[e] : u0[e];
            t[4 + e] = a.self_resid ? t[4 + e] + u1[e] : u1[e];
          }
        }
        if (rap) {
          const float r8[8] = {ra0[it][0], ra0[it][1], ra0[it][2], ra0[it][3], ra1[it][0], ra1[it][1], ra1[it][2], ra1[it][3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] += r8[e];
        }
        if (rp) {
          Pack8 p8; p8.u = rv[it];
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] += bits_to_f32<T>(p8.h[e]);
        }
        if (gp) {  // ReLU backward: pass where the forward activation was positive
          Pack8 p8; p8.u = gv[it];
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = (bits_to_f32<T>(p8.h[e]) > 0.f) ? t[e] : 0.f;
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = (uint32_t)f32_to_bits<T>(t[2 * e]) | ((uint32_t)f32_to_bits<T>(t[2 * e + 1]) << 16);
        *reinterpret_cast<u32x4*>(cp) = o;
        cp += (int64_t)RSTEP * a.ldc;
        w += wstep;
      }
    } else
    if (nc < n_store) {
#pragma unroll 2
      for (int row = tid / CPR; row < WTM; row += NT / CPR) {
        const int64_t m = m0 + pass * WTM + row;
        if (m >= a.M) break;
        const int64_t pm = ROWOPS ? remap_row(m, a.c_rows_per_group, a.c_group_stride, a.c_group_offset) : m;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sFp + row * LDS_F + ch * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sFp + row * LDS_F + ch * 8 + 4);
        float t[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        // Optional features are wave-uniform branches per 8-column chunk (a branch costs less than 8 neutral
        // operations; per-ELEMENT tests in accumulator layout had made the kernel issue-bound).
        // t = alpha * (acc + bias [+ class row]) -> ReLU
        if (has_rowops && a.grpadd) {
          const float* ga_row = a.grpadd + (int64_t)a.grp_index[m / a.rowadd_period] * a.ldga + nc;
#pragma unroll
          for (int e = 0; e < 8; ++e) if (nc + e < N32) t[e] += ga_row[e];
        }
        if (a.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] += bias8[e];
        }
        if (alpha != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] *= alpha;
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) t[e] = fmaxf(t[e], 0.f);
        }
        if (DROP && (has_drop || a.self_resid)) {
          float u0[4] = {t[0], t[1], t[2], t[3]}, u1[4] = {t[4], t[5], t[6], t[7]};
          if (has_drop) {  // N % 4 == 0 when dropout is on: (row*N + nc) starts a 4-decision word
            const uint64_t w = (uint64_t)(pm * a.N + nc) >> 2;  // counter = PHYSICAL output row: survives row remaps
            dropout_apply4(dkey, w, dthr, inv_keep, u0);
            dropout_apply4(dkey, w + 1, dthr, inv_keep, u1);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            t[e] = a.self_resid ? t[e] + u0[e] : u0[e];
            t[4 + e] = a.self_resid ? t[4 + e] + u1[e] : u1[e];
          }
        }
        if (has_rowops && a.rowadd) {
          const float* ra_row = a.rowadd + (m % a.rowadd_period) * a.ldra + nc;
#pragma unroll
          for (int e = 0; e < 8; ++e) if (nc + e < N32) t[e] += ra_row[e];
        }
        if (resid) {
          const T* rp = resid + ((ROWOPS && a.resid_phys) ? pm : m) * a.ldr + nc;
          if (!edge) {
            Pack8 p8; p8.u = *reinterpret_cast<const u32x4*>(rp);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] += bits_to_f32<T>(p8.h[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (nc + e < a.ldr) t[e] += to_f32(rp[e]);
          }
        }
        if (gate) {  // ReLU backward: pass where the forward activation was positive
          const T* gp = gate + m * a.ldg + nc;
          if (!edge) {
            Pack8 p8; p8.u = *reinterpret_cast<const u32x4*>(gp);
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (bits_to_f32<T>(p8.h[e]) > 0.f) ? t[e] : 0.f;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (nc + e < a.ldg) t[e] = (to_f32(gp[e]) > 0.f) ? t[e] : 0.f;
          }
        }
        if (edge) {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (nc + e >= N32) t[e] = 0.f;
        }
        if (C_F32) {
          float* cp = reinterpret_cast<float*>(a.C) + pm * a.ldc + nc;
          *reinterpret_cast<f32x4*>(cp) = f32x4{t[0], t[1], t[2], t[3]};
          if (nc + 8 <= n_store) *reinterpret_cast<f32x4*>(cp + 4) = f32x4{t[4], t[5], t[6], t[7]};
        } else {
          T* cp = reinterpret_cast<T*>(a.C) + pm * a.ldc + nc;
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = (uint32_t)f32_to_bits<T>(t[2 * e]) | ((uint32_t)f32_to_bits<T>(t[2 * e + 1]) << 16);
          if (!edge) {
            *reinterpret_cast<u32x4*>(cp) = o;
          } else {
            *reinterpret_cast<u32x2*>(cp) = u32x2{o[0], o[1]};
            if (nc + 8 <= n_store) *reinterpret_cast<u32x2*>(cp + 4) = u32x2{o[2], o[3]};
          }
        }
      }
    }
  }
}

// The workgroup's tile (the XCD-aware order of gemm_mainloop, which recomputes it) and the epilogue's bias for this thread's
// 8-column chunk, requested BEFORE the K loop: the parameters are cold lines after every optimizer step, and a load issued at
// the epilogue's start is a round trip that nothing covers once the tile's MFMAs are done.
template <int BM, int BN>
__device__ __forceinline__ void gemm_tile_origin(const mst_gemm_args& a, int64_t& m0, int64_t& n0, int64_t bid_in = -1) {
  const int64_t tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM, nwg = tiles_m * tiles_n;
  int64_t bid = bid_in < 0 ? (int64_t)blockIdx.x : bid_in;  // (bid_in: the tile's index inside its own problem, mst_gemm_nt_pair)
  const int64_t q = nwg / 8, r = nwg % 8, x = bid % 8, y = bid / 8;
  bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  m0 = (bid / tiles_n) * BM;
  n0 = (bid % tiles_n) * BN;
}
template <int BM, int BN>
__device__ __forceinline__ void gemm_bias_preload(const mst_gemm_args& a, float (&bias8)[8], int64_t bid_in = -1) {
  int64_t m0, n0;
  gemm_tile_origin<BM, BN>(a, m0, n0, bid_in);
  const int nc = (int)n0 + ((int)threadIdx.x % (BN / 8)) * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = (a.bias && nc + e < (int)a.N) ? a.bias[nc + e] : 0.f;
}

// The tile's K loop, shared by the kernels below: locates the workgroup's tile (m0, n0) and leaves the fp32
// accumulators in `acc`; on return every wave has passed the loop's last barrier, so `smem` is free to reuse.
// AREMAP: the A row remap is compiled in (a 64-bit division per staged chunk of the prologue).
// AU8: A holds uint8 elements (mst_gemm_args.a_u8): a chunk is an 8-byte load, widened when it is written to LDS.
// A_IN_LDS: both K stages of the A tile already sit in the stage buffers (K == 2 * BK: stage t in buffer t, the layout store_tile
// writes) — the operand was produced by this workgroup (gemm_bce_dgrad_ln_kernel); only B is loaded.
// FULL: every tile of the launch is interior and K is a multiple of BK (host check): the stage loads carry no row / K guards. A
// guarded load is a compare, an exec-mask branch and a zero fill of its four registers — 33 vector instructions per stage of 8
// MFMAs per wave in the 64 x 64 form, as many issue cycles as the MFMAs themselves (SQ counters: the K = 768 input-gradient
// launches kept the vector ALUs busy 65 % of the time).
template <typename T, int BM, int BN, int WGM, int WGN, int BK, bool AREMAP = true, bool AU8 = false, bool A_IN_LDS = false, bool FULL = false>
__device__ __forceinline__ void gemm_mainloop(const mst_gemm_args& a, unsigned char* smem,
                                              f32x4 (&acc)[(BN / WGN) / 16][(BM / WGM) / 16], int64_t& m0, int64_t& n0,
                                              int64_t bid_in = -1) {
  constexpr int CHUNKS = BK / 8;
  constexpr int NT = WGM * WGN * 64;
  // XOR swizzle of a row's 16-byte chunk index (conflict-free ds_read_b128 of 16 rows at one k): the row's low 3 bits with
  // 8+ chunks per row (128-byte rows and wider); with 4 chunks (32-deep stages, 64-byte rows) four rows share a 256-byte
  // bank window, so bits 2..3 of the row — and the result stays inside the row's own four slots
  auto swz = [](int row) { return CHUNKS >= 8 ? (row & 7) : ((row >> 2) & (CHUNKS - 1)); };
  static_assert(CHUNKS >= 4, "K stages are at least 32 deep");
  constexpr int WTM = BM / WGM, WTN = BN / WGN;  // wave tile
  constexpr int TM = WTM / 16, TN = WTN / 16;    // 16x16 sub-tiles per wave
  constexpr int A_CH = (BM * CHUNKS + NT - 1) / NT, B_CH = BN * CHUNKS / NT;  // a short A tile may not occupy every thread
  static_assert((BM * CHUNKS % NT == 0 || BM * CHUNKS < NT) && BN * CHUNKS % NT == 0, "tile/threads mismatch");
  typedef typename Act<T>::vec8 vec8;

  u32x4* sA = reinterpret_cast<u32x4*>(smem);                    // [2][BM*CHUNKS]
  u32x4* sB = sA + 2 * BM * CHUNKS;                              // [2][BN*CHUNKS]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-aware tile order: blocks sharing an XCD (blockIdx % 8) walk neighbouring M tiles of one
  // N panel, so the weight panel and the A rows they share stay in that XCD's L2.
  const int64_t tiles_n = (a.N + BN - 1) / BN;
  const int64_t tiles_m = (a.M + BM - 1) / BM;
  const int64_t nwg = tiles_m * tiles_n;
  int64_t bid = bid_in < 0 ? (int64_t)blockIdx.x : bid_in;
  {
    const int64_t q = nwg / 8, r = nwg % 8, x = bid % 8, y = bid / 8;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  m0 = (bid / tiles_n) * BM;
  n0 = (bid % tiles_n) * BN;

  typedef typename std::conditional<AU8, uint8_t, T>::type TA;
  const TA* __restrict__ A = reinterpret_cast<const TA*>(a.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(a.B);

  // per-thread staging assignment: chunk c -> (row = c / CHUNKS, ch = c % CHUNKS)
  const TA* a_ptr[A_CH];
  bool a_ok[A_CH];
  int a_lds[A_CH], a_ch[A_CH];
  // (a tile that cannot straddle a remap group — groups a multiple of the tile height, as in every remapped launch of the step —
  // takes ONE 64-bit division for its first row instead of one per staged chunk)
  const bool a_tile_remap = AREMAP && a.a_rows_per_group > 0 && a.a_rows_per_group % BM == 0;
  const int64_t a_pm0 = a_tile_remap ? remap_row(m0, a.a_rows_per_group, a.a_group_stride, a.a_group_offset) : 0;
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    int c = tid + i * NT, row = c / CHUNKS, ch = c % CHUNKS;
    int64_t m = m0 + row;
    a_ok[i] = (FULL || m < a.M) && c < BM * CHUNKS;
    int64_t pm = a_tile_remap ? (a_ok[i] ? a_pm0 + row : 0)
                              : (AREMAP ? remap_row(a_ok[i] ? m : 0, a.a_rows_per_group, a.a_group_stride, a.a_group_offset) : (a_ok[i] ? m : 0));
    a_ptr[i] = A + pm * a.lda + ch * 8;
    a_ch[i] = ch * 8;
    a_lds[i] = row * CHUNKS + (ch ^ swz(row));
  }
  const T* b_ptr[B_CH];
  bool b_ok[B_CH];
  int b_lds[B_CH], b_ch[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    int c = tid + i * NT, row = c / CHUNKS, ch = c % CHUNKS;
    int64_t n = n0 + row;
    b_ok[i] = FULL || n < a.N;
    b_ptr[i] = B + (b_ok[i] ? n : 0) * a.ldb + ch * 8;
    b_ch[i] = ch * 8;
    b_lds[i] = row * CHUNKS + (ch ^ swz(row));
  }

  u32x4 ra[A_CH], rb[B_CH];
  u32x2 ra8[AU8 ? A_CH : 1];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_tile = [&](int64_t k0) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      if constexpr (A_IN_LDS) continue;
      if constexpr (FULL && BM * CHUNKS % NT == 0) {
        if constexpr (AU8) ra8[i] = *reinterpret_cast<const u32x2*>(a_ptr[i] + k0);
        else ra[i] = *reinterpret_cast<const u32x4*>(a_ptr[i] + k0);
        continue;
      }
      if constexpr (AU8) ra8[i] = (a_ok[i] && (FULL || k0 + a_ch[i] < a.K)) ? *reinterpret_cast<const u32x2*>(a_ptr[i] + k0) : u32x2{0u, 0u};
      else ra[i] = (a_ok[i] && (FULL || k0 + a_ch[i] < a.K)) ? *reinterpret_cast<const u32x4*>(a_ptr[i] + k0) : zero4;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      if constexpr (FULL) rb[i] = *reinterpret_cast<const u32x4*>(b_ptr[i] + k0);
      else rb[i] = (b_ok[i] && (k0 + b_ch[i] < a.K)) ? *reinterpret_cast<const u32x4*>(b_ptr[i] + k0) : zero4;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      if constexpr (A_IN_LDS) continue;
      if constexpr (AU8) ra[i] = expand_u8x8<T>(ra8[i]);
      if (BM * CHUNKS >= NT || tid < BM * CHUNKS) sA[buf * BM * CHUNKS + a_lds[i]] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) sB[buf * BN * CHUNKS + b_lds[i]] = rb[i];
  };

#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int64_t nk = (a.K + BK - 1) / BK;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int64_t t = 0; t < nk; ++t) {
    const int cur = (int)(t & 1);
    if (t + 1 < nk) load_tile((t + 1) * BK);
    const u32x4* cA = sA + cur * BM * CHUNKS;
    const u32x4* cB = sB + cur * BN * CHUNKS;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      vec8 xf[TM], wf[TN];
      const int kc = ks * 4 + fq;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int row = wm * WTM + i * 16 + frow;
        u32x4 v = cA[row * CHUNKS + (kc ^ swz(row))];
        xf[i] = __builtin_bit_cast(vec8, v);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int row = wn * WTN + j * 16 + frow;
        u32x4 v = cB[row * CHUNKS + (kc ^ swz(row))];
        wf[j] = __builtin_bit_cast(vec8, v);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = Act<T>::mfma16(wf[j], xf[i], acc[j][i]);
    }
    if (t + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
}

}  // namespace mst

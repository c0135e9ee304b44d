// gemm_nt.hip — C[M,N] = epilogue(A[M,K] · B[N,K]^T), 16-bit operands, fp32 accumulate on MFMA.
//
// Replaces gluon.nn.Dense(flatten=False) (y = x W^T + b, W stored [units, in_units]) at
// VarAutoEncoder/transformer.py:36-40,65-68 and model.py:214-227, and — called with the W^T
// shadow as B — the data-gradient of the same layers.
//
// Design (gfx950): both operands are K-contiguous, so every MFMA fragment is one 16-byte LDS
// read. The product is formed "swapped" (MFMA A-operand = weight rows n, B-operand = activation
// rows m) so an accumulator register quad is 4 consecutive n of one output row: the epilogue
// reads bias/residual/gate and writes C with 8-byte (16-bit C) or 16-byte (fp32 C) accesses.
// Tiles are staged global → registers → LDS (XOR-swizzled 16-byte chunks, conflict-free
// ds_read_b128), double-buffered with one barrier per 64-deep K tile; the global loads of tile
// t+1 are issued before the MFMAs of tile t and written to LDS after them.
#include <math.h>
#include <type_traits>
#include "common.hpp"
#include "gemm_tile.hpp"
#include "bce_math.hpp"
#include "step_begin.hpp"
#include "shadows.hpp"

namespace mst {

template <typename T, int BM, int BN, int WGM, int WGN, bool C_F32, int BK, bool ROWOPS, int PATH, bool DROP, bool AU8 = false>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_nt_kernel(mst_gemm_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4 acc[(BN / WGN) / 16][(BM / WGM) / 16];
  int64_t m0, n0;
  float bias_pre[8];
  gemm_bias_preload<BM, BN>(a, bias_pre);
  gemm_mainloop<T, BM, BN, WGM, WGN, BK, ROWOPS, AU8, false, PATH == 1 || PATH == 4>(a, smem, acc, m0, n0);  // (PATH 1, 4: whole tiles, whole K stages)
  // (the launch allocates max(K-loop tiles, BM x (BN+4) fp32 staging) bytes of LDS: launch_gemm)
  gemm_epilogue<T, BM, BN, WGM, WGN, C_F32, ROWOPS, PATH, DROP>(a, smem, acc, m0, n0, bias_pre);
}

// Two GEMMs of one kernel form in ONE launch (mst_gemm_nt_pair): the first `tiles0` workgroups are the first problem's tiles,
// the rest the second's. The piano-roll ends' two embedding GEMMs (model.py:81-91 and :241-245: the same uint8 frames against the
// encoder's and the decoder's table) — small launches whose cost is mostly the launch.
// n_begin > 0 (mst_gemm_nt_pair_begin): the first n_begin workgroups of the grid are the step's bookkeeping (step_begin.hpp).
template <typename T, int BM, int BN, int WGM, int WGN, int BK, int PATH>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_nt_pair_kernel(mst_gemm_args a0, mst_gemm_args a1, int tiles0, int tiles1, StepBegin sb,
                                                                       int n_begin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int)blockIdx.x < n_begin) {
    step_begin_wg<WGM * WGN * 64>(sb, (int)blockIdx.x, n_begin);
    return;
  }
  f32x4 acc[(BN / WGN) / 16][(BM / WGM) / 16];
  int64_t m0, n0;
  float bias_pre[8];
  const int64_t tile = (int64_t)blockIdx.x - n_begin;  // (n_begin is a multiple of 8: a tile keeps its XCD)
  if (tile >= tiles0 + tiles1) {
    // behind the GEMM tiles: the transposed 16-bit shadows of the matrices only the backward pass reads, rebuilt from the weights the
    // previous step's optimizer left (mst_step_begin_args.sh_*) — as a launch of their own behind the optimizer they cost 6.5 us
    shadow_tile_wg<T>(sb.sh_w, reinterpret_cast<T*>(sb.sh_wt16), sb.sh_desc, sb.sh_prefix, sb.sh_n_mat, tile - tiles0 - tiles1,
                      reinterpret_cast<float(*)[33]>(smem));
    return;
  }
  // (two straight-line copies of the tile, each reading its own argument block from the kernel arguments)
  if (tile < tiles0) {
    gemm_bias_preload<BM, BN>(a0, bias_pre, tile);
    gemm_mainloop<T, BM, BN, WGM, WGN, BK, true, true, false, true>(a0, smem, acc, m0, n0, tile);  // (whole tiles, whole K stages: host check)
    gemm_epilogue<T, BM, BN, WGM, WGN, false, true, PATH, false>(a0, smem, acc, m0, n0, bias_pre);
  } else {
    const int64_t bid = tile - tiles0;
    gemm_bias_preload<BM, BN>(a1, bias_pre, bid);
    gemm_mainloop<T, BM, BN, WGM, WGN, BK, true, true, false, true>(a1, smem, acc, m0, n0, bid);
    gemm_epilogue<T, BM, BN, WGM, WGN, false, true, PATH, false>(a1, smem, acc, m0, n0, bias_pre);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm fused into the epilogue of a GEMM whose tile spans the whole output row (BN == N).
//   mode 1 (forward):  h = epi(acc) is written to C as usual (the backward pass needs the pre-norm tensor) and
//                      y = LayerNorm(h) goes to ln.out, mean / rstd to ln.mean / ln.rstd — what mst_layernorm_fwd would
//                      compute from C (two-pass statistics on the 16-bit-rounded row, gluon.nn.LayerNorm eps).
//   mode 2 (backward): dy = epi(acc) is NOT stored; dx = LayerNorm-backward(dy; x, mean, rstd, gamma) goes to C, the
//                      dropout-masked copy (mask_mode 1) to ln.out, dgamma / dbeta are accumulated — mst_layernorm_bwd
//                      on the GEMM's result, without the round trip through HBM and without its launch.
// Supported epilogue features: bias, alpha, dropout / self_resid, residual, C row remap (the others are rejected on the
// host). One thread finishes 8 columns of a row; the N/8 threads of a row are consecutive lanes, so row sums are
// xor-shuffles inside a 32- or 16-lane group.
// diagnostic build only (-DMST_FFN_STAMPS): one workgroup leaves s_memtime stamps per stage (tools/bench_ffn_stamps.py)
#ifdef MST_FFN_STAMPS
__device__ uint64_t g_ffn_stamps[8 + 48 * 4];  // [0..3] kernel phases, [5..7] LayerNorm epilogue, [8 + 4k..] stage k, [190, 191] realtime
#define FFN_STAMP(slot) do { if (blockIdx.x == 64 && threadIdx.x == 0 && (slot) < 8 + 48 * 4) g_ffn_stamps[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#define FFN_RT(slot) do { if (blockIdx.x == 64 && threadIdx.x == 0) g_ffn_stamps[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FFN_STAMP(slot) do { } while (0)
#define FFN_RT(slot) do { } while (0)
#endif
template <int LANES>
__device__ __forceinline__ float row_sum(float v) { return group_sum<LANES>(v); }

template <typename T, int BM, int BN, int WGM, int WGN, int MODE>
__device__ __forceinline__ void gemm_epilogue_ln(const mst_gemm_args& a, const mst_ln_args& l, unsigned char* smem,
                                                 f32x4 (&acc)[(BN / WGN) / 16][(BM / WGM) / 16], int64_t m0,
                                                 const T* lds_resid = nullptr, int lds_resid_ld = 0,
                                                 T* lds_out = nullptr, int lds_out_ld = 0, const float* lds_par = nullptr,
                                                 const uint64_t* dseed_pre = nullptr /* the step's dropout seed, already loaded */) {
  // lds_resid: the workgroup's BM residual rows already sit in LDS (row stride lds_resid_ld elements, outside the staging
  // tile): they are read from there instead of from a.resid
  // lds_par: [bias | gamma | beta] (3 x BN floats) already in LDS (outside the staging tile): in a kernel that is one
  // workgroup per CU these cold parameter lines (the optimizer rewrote them) are an exposed round trip at the epilogue's start
  // lds_out: the result rows ALSO go to this LDS tile (outside the staging tile; rows past M as zeros): the LayerNorm output
  // (mode 1) or the input gradient — its masked copy when there is one — (mode 2), for a GEMM that follows in the same launch
  constexpr int NT = WGM * WGN * 64;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int TM = WTM / 16, TN = WTN / 16;
  constexpr int LDS_F = BN + 4, CPR = BN / 8, RSTEP = NT / CPR, ITERS = BM / RSTEP;
  static_assert(CPR == 32 || CPR == 16, "a row must be a 32- or 16-lane group");
  static_assert(BM % RSTEP == 0, "rows per thread must be whole");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int frow = lane & 15, fq = lane >> 4;
  float* sF = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      *reinterpret_cast<f32x4*>(sF + (wm * WTM + i * 16 + frow) * LDS_F + wn * WTN + j * 16 + fq * 4) = acc[j][i];
  __syncthreads();
  FFN_STAMP(5);

  const int ch = tid % CPR, nc = ch * 8, row0 = tid / CPR;
  const float inv_n = 1.f / (float)BN;
  const float inv_keep = dropout_inv_keep(a.dropout_p);
  const bool has_drop = a.dropout_p > 0.f;
  const uint64_t dseed = dseed_pre ? *dseed_pre : a.dropout_seed ^ ((has_drop && a.dropout_seed_ptr) ? a.dropout_seed_ptr[0] : 0ull);
  const uint32_t dkey = dropout_key(dseed, a.dropout_site), dthr = dropout_thr(a.dropout_p);
  const T* resid = reinterpret_cast<const T*>(a.resid);
  float bias8[8], gam8[8], bet8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (lds_par) {
      bias8[e] = lds_par[nc + e];
      gam8[e] = lds_par[BN + nc + e];
      bet8[e] = (MODE == 1) ? lds_par[2 * BN + nc + e] : 0.f;
    } else {
      bias8[e] = a.bias ? a.bias[nc + e] : 0.f;
      gam8[e] = l.gamma[nc + e];
      bet8[e] = (MODE == 1) ? l.beta[nc + e] : 0.f;
    }
  }
  float dg8[8], db8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { dg8[e] = 0.f; db8[e] = 0.f; }

  // every global load of the thread's rows (residual; backward: x, mean, rstd) is issued before the first row is finished:
  // in the step these lines are cold, and a row-by-row loop exposed one memory round trip per row at 8 waves per CU
  u32x4 rv[ITERS], xv[ITERS];
  float mean_r[ITERS], rstd_r[ITERS];
  // the row remap once per tile where a tile cannot straddle a group (the step's remapped launch: rows 1..T of T + 1, T a multiple
  // of the tile height): per row it is two 64-bit divisions, ~200 instructions each, in front of every row's loads
  const bool tile_remap = a.c_rows_per_group <= 0 || a.c_rows_per_group % BM == 0;
  const int64_t pm0 = remap_row(m0, a.c_rows_per_group, a.c_group_stride, a.c_group_offset);
  auto phys_row = [&](int64_t m) { return tile_remap ? pm0 + (m - m0) : remap_row(m, a.c_rows_per_group, a.c_group_stride, a.c_group_offset); };
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int64_t m = m0 + row0 + it * RSTEP;
    rv[it] = u32x4{0u, 0u, 0u, 0u}; xv[it] = rv[it]; mean_r[it] = 0.f; rstd_r[it] = 0.f;
    if (m < a.M) {
      const int64_t pm = phys_row(m);
      if (lds_resid) rv[it] = *reinterpret_cast<const u32x4*>(lds_resid + (row0 + it * RSTEP) * lds_resid_ld + nc);
      else if (resid) rv[it] = *reinterpret_cast<const u32x4*>(resid + m * a.ldr + nc);
      if (MODE == 2) {
        xv[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(l.x) + pm * l.ld_x + nc);
        mean_r[it] = l.mean[pm];
        rstd_r[it] = l.rstd[pm];
      }
    }
  }
  FFN_STAMP(6);
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int row = row0 + it * RSTEP;
    const int64_t m = m0 + row;
    if (m < a.M) {  // uniform for the lanes of a row
      const int64_t pm = phys_row(m);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(sF + row * LDS_F + nc);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(sF + row * LDS_F + nc + 4);
      float t[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      // ---- the GEMM's own epilogue (same order as gemm_epilogue): bias, alpha, dropout / self_resid, residual
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (t[e] + bias8[e]) * a.alpha;
      if (MODE == 1 && (has_drop || a.self_resid)) {
        float u0[4] = {t[0], t[1], t[2], t[3]}, u1[4] = {t[4], t[5], t[6], t[7]};
        if (has_drop) {
          const uint64_t w = (uint64_t)(pm * a.N + nc) >> 2;
          dropout_apply4(dkey, w, dthr, inv_keep, u0);
          dropout_apply4(dkey, w + 1, dthr, inv_keep, u1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          t[e] = a.self_resid ? t[e] + u0[e] : u0[e];
          t[4 + e] = a.self_resid ? t[4 + e] + u1[e] : u1[e];
        }
      }
      if (resid || lds_resid) {
        Pack8 p8; p8.u = rv[it];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] += bits_to_f32<T>(p8.h[e]);
      }
      // the value the unfused pipeline would have stored and re-read: round to the activation type first
      Pack8 hb;
#pragma unroll
      for (int e = 0; e < 8; ++e) { hb.h[e] = f32_to_bits<T>(t[e]); t[e] = bits_to_f32<T>(hb.h[e]); }
      if (MODE == 1) {
        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.C) + pm * a.ldc + nc) = hb.u;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += t[e];
        const float mean = row_sum<CPR>(s) * inv_n;
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { t[e] -= mean; ss += t[e] * t[e]; }
        const float rstd = 1.f / sqrtf(row_sum<CPR>(ss) * inv_n + l.eps);
        Pack8 yb;
#pragma unroll
        for (int e = 0; e < 8; ++e) yb.h[e] = f32_to_bits<T>(t[e] * rstd * gam8[e] + bet8[e]);
        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(l.out) + pm * l.ld_out + nc) = yb.u;
        if (lds_out) *reinterpret_cast<u32x4*>(lds_out + row * lds_out_ld + nc) = yb.u;
        if (ch == 0) { l.mean[pm] = mean; l.rstd[pm] = rstd; }
      } else {
        const int64_t rid = pm;  // x, the statistics and the forward's dropout counter live at the PHYSICAL row of C
        const float mean = mean_r[it], rstd = rstd_r[it];
        Pack8 xb; xb.u = xv[it];
        float xh[8], g[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[e] = (bits_to_f32<T>(xb.h[e]) - mean) * rstd;
          g[e] = t[e] * gam8[e];
          s1 += g[e];
          s2 += g[e] * xh[e];
          dg8[e] += t[e] * xh[e];
          db8[e] += t[e];
        }
        s1 = row_sum<CPR>(s1) * inv_n;
        s2 = row_sum<CPR>(s2) * inv_n;
        uint32_t keep8 = 0xFFu;
        if (l.mask_mode != 0 && has_drop) {
          const uint64_t w = (uint64_t)(rid * BN + nc) >> 2;  // the forward's counter: forward row id, N == BN columns
          keep8 = dropout_keep4k(dkey, w, dthr) | (dropout_keep4k(dkey, w + 1, dthr) << 4);
        }
        Pack8 ob, mb;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float o = rstd * (g[e] - s1 - xh[e] * s2);
          float om = 0.f;
          if (l.mask_mode != 0) {
            const float k = has_drop ? (((keep8 >> e) & 1u) ? inv_keep : 0.f) : 1.f;
            if (l.mask_mode == 1) om = o * k; else o = o * (1.f + k);
          }
          ob.h[e] = f32_to_bits<T>(o);
          mb.h[e] = f32_to_bits<T>(om);
        }
        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.C) + pm * a.ldc + nc) = ob.u;
        if (l.mask_mode == 1) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(l.out) + m * l.ld_out + nc) = mb.u;
        if (lds_out) *reinterpret_cast<u32x4*>(lds_out + row * lds_out_ld + nc) = l.mask_mode == 1 ? mb.u : ob.u;
      }
    } else if (lds_out) {
      *reinterpret_cast<u32x4*>(lds_out + row * lds_out_ld + nc) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  FFN_STAMP(7);
  if (MODE == 2) {
    // dgamma / dbeta: sum the RSTEP row groups through LDS (the staged tile is dead), one atomic per column per workgroup
    __syncthreads();
    float* red = sF;  // [2][RSTEP][BN]
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[row0 * BN + nc + e] = dg8[e];
      red[(RSTEP + row0) * BN + nc + e] = db8[e];
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += NT) {
      const int which = c / BN, col = c % BN;
      float s = 0.f;
      for (int r = 0; r < RSTEP; ++r) s += red[(which * RSTEP + r) * BN + col];
      if (l.partials) l.partials[(int64_t)blockIdx.x * 2 * BN + c] = s;  // [dgamma | dbeta], summed by partial_sums_kernel
      else atomicAdd((which ? l.dbeta : l.dgamma) + col, s);
    }
  }
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_nt_ln_kernel(mst_gemm_args a, mst_ln_args l) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4 acc[(BN / WGN) / 16][(BM / WGM) / 16];
  int64_t m0, n0;
  // bias | gamma | beta wait in LDS behind the K-loop tiles / the staging tile (launch_gemm_ln sizes it): cold lines, requested now
  constexpr size_t LOOP_B = (size_t)2 * (BM + BN) * 64 * 2, EPI_B = (size_t)BM * (BN + 4) * 4;
  float* sPar = reinterpret_cast<float*>(smem + (LOOP_B > EPI_B ? LOOP_B : EPI_B));
  for (int i = threadIdx.x; i < BN; i += WGM * WGN * 64) {
    sPar[i] = a.bias ? a.bias[i] : 0.f;
    sPar[BN + i] = l.gamma[i];
    sPar[2 * BN + i] = (MODE == 1) ? l.beta[i] : 0.f;
  }
  gemm_mainloop<T, BM, BN, WGM, WGN, 64>(a, smem, acc, m0, n0);
  gemm_epilogue_ln<T, BM, BN, WGM, WGN, MODE>(a, l, smem, acc, m0, nullptr, 0, nullptr, 0, sPar);
}

// ---------------------------------------------------------------------------------------------------------------
// Output layer + per-pitch BCE in ONE launch (mst_gemm_sigmoid_bce): the decoder's Dense[D -> P] (model.py:253-256) with
// sigmoid + BinaryCrossEntropy (loss.py:27-80) in its epilogue. A tile is 64 frames x BN pitches (128 or 256: the LDS-staged
// (time x pitch) tile) — the whole row of pitches at configs[1], one of P / 256 column tiles of it at configs[2]'s 2048 (the
// loss is a plain sum over frames and pitches, so column tiles only share the sample's atomic) — and the logits never reach
// HBM: the epilogue turns the fp32 accumulators into the
// logit gradient (the backward pass's operand), optionally the probabilities (reconstruction output), and the sample's
// loss sum — the arithmetic of sigmoid_bce_kernel on the logit rounded to the activation type, which is what the two-launch
// form reads back. A tile holds rows of ONE sample (the host requires T % 64 == 0): one atomic per workgroup.
// keepA (KEEP): the logit-gradient tile ALSO goes to LDS as the A operand of a GEMM that follows in the same launch, in
// gemm_mainloop's stage layout (BK = 64: columns [64 s, 64 s + 64) in stage buffer s, 16-byte chunks XOR-swizzled by the row)
template <typename T, int BN, bool KEEP>
__device__ __forceinline__ void gemm_bce_tile(const mst_gemm_args& a, const mst_bce_args& q, unsigned char* smem, float* red, u32x4* keepA) {
  constexpr int BM = 64, WGM = 2, WGN = 4, NT = 512;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 16, TN = WTN / 16;
  constexpr int LDS_F = BN + 4, CPR = BN / 8, RSTEP = NT / CPR, ITERS = BM / RSTEP;
  f32x4 acc[TN][TM];
  int64_t m0, n0;
  float bias8[8];
  gemm_bias_preload<BM, BN>(a, bias8);
  gemm_mainloop<T, BM, BN, WGM, WGN, 64>(a, smem, acc, m0, n0);
  const int64_t P = a.N;                       // pitches per frame (a multiple of BN: this tile holds columns [n0, n0 + BN))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN, frow = lane & 15, fq = lane >> 4;
  float* sF = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      *reinterpret_cast<f32x4*>(sF + (wm * WTM + i * 16 + frow) * LDS_F + wn * WTN + j * 16 + fq * 4) = acc[j][i];
  const int64_t b = m0 / q.T;                  // the tile's sample
  const int64_t per_sample = q.T * P;
  float w = 0.f;
  if (q.downweight) {                          // loss.py:58-81: w_b = n_pos / (n_neg + 1e-12) over the SAMPLE's labels
    const uint8_t* lab = q.labels + b * per_sample;
    int cnt = 0;
    for (int64_t i = (int64_t)tid * 8; i < per_sample; i += (int64_t)NT * 8)
      cnt += __popcll(*reinterpret_cast<const uint64_t*>(lab + i) & 0x0101010101010101ull);
    float c = wave_sum((float)cnt);
    if (lane == 0) red[wave] = c;
    __syncthreads();
    float np = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) np += red[i];
    w = np / (((float)q.T * (float)P - np) + 1e-12f);
  }
  __syncthreads();                              // staged tile visible (and `red` free again)
  const int ch = tid % CPR, nc = ch * 8, row0 = tid / CPR;
  const int64_t gc = n0 + nc;                   // this thread's 8 pitches in the frame
  const float inv_n = 1.f / ((float)q.T * (float)P), ls = q.label_smoothing;
  float lsum = 0.f;
  const float s1 = (1.f - ls) + 0.5f * ls, s0 = 0.5f * ls;
  auto sweep = [&](auto dwc) {
    constexpr bool DW = decltype(dwc)::value;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int row = row0 + it * RSTEP;
      const int64_t m = m0 + row;
      const bool live = m < a.M;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(sF + row * LDS_F + nc);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(sF + row * LDS_F + nc + 4);
      const float t8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const uint64_t lab8 = live ? *reinterpret_cast<const uint64_t*>(q.labels + m * P + gc) : 0ull;
      Pack8 pb, gb, xb;
      float x8[8];
      bool in_dom = true;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xb.h[e] = f32_to_bits<T>((t8[e] + bias8[e]) * a.alpha);  // the logit as the unfused pipeline stores it
        x8[e] = bits_to_f32<T>(xb.h[e]);
        in_dom = in_dom && bce_fast_domain(x8[e]);
      }
      const bool fast = __all(in_dom || !live);  // wave-uniform (bce_math.hpp: three transcendental instructions per element, not six)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = (float)((lab8 >> (8 * e)) & 0xFFull);
        float p, bce, dbce;
        bce_fast<DW>(x8[e], y, s1, s0, w, p, bce, dbce);
        if (!fast) {  // a saturated logit somewhere in this wave: ITS element takes the reference's operation order (an element's
                      // result depends on its own logit only, so the fused and the two-launch forms agree bit for bit)
          float p2, b2, d2;
          bce_exact<DW>(x8[e], y, ls, w, p2, b2, d2);
          if (!bce_fast_domain(x8[e])) { p = p2; bce = b2; dbce = d2; }
        }
        lsum += live ? bce : 0.f;
        pb.h[e] = f32_to_bits<T>(p);
        gb.h[e] = f32_to_bits<T>(dbce * inv_n * q.gscale);
      }
      if (!live) continue;
      if (a.C) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.C) + m * a.ldc + gc) = gb.u;
      if constexpr (KEEP) keepA[(ch >> 3) * (BM * 8) + row * 8 + ((ch & 7) ^ (row & 7))] = gb.u;  // (KEEP: P == BN)
      if (q.probs) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(q.probs) + m * q.ldp + gc) = pb.u;
      if (q.logits) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(q.logits) + m * q.ldl + gc) = xb.u;
    }
  };
  if (q.downweight) sweep(std::true_type()); else sweep(std::false_type());
  lsum = wave_sum(lsum);
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) tot += red[i];
    atomicAdd(q.loss + b, tot * inv_n);
  }
}

template <typename T, int BN>
__global__ __launch_bounds__(512) void gemm_bce_kernel(mst_gemm_args a, mst_bce_args q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[512 / 64];
  gemm_bce_tile<T, BN, false>(a, q, smem, red, nullptr);
}

// mst_gemm_sigmoid_bce_dgrad_ln: the loss launch above followed IN THE SAME WORKGROUP by the first launch of the backward pass — the
// output layer's input gradient d(dec_out) = dlogits W_out (K = the 128 pitches of the tile the workgroup has just produced) with the
// last decoder layer's LayerNorm-3 backward in its epilogue (mst_gemm_nt_ln mode 2). The logit gradient still goes to HBM (the
// weight-gradient launch reads it) but is not read back here, and a launch of the dependent chain disappears.
// LDS: [0, 48 K) the first GEMM's stages, then its fp32 staging tile (33.8 K), later the LayerNorm epilogue's; [48 K, 64 K) the kept
// logit-gradient tile (two 64 x 64 stages); [64 K, 96 K) the second GEMM's weight stages; then bias | gamma | beta.
template <typename T>
__global__ __launch_bounds__(512) void gemm_bce_dgrad_ln_kernel(mst_gemm_args a, mst_bce_args q, mst_gemm_args g2, mst_ln_args l) {
  constexpr int BM = 64, BN = 128, WGM = 2, WGN = 4;
  constexpr size_t OFF2 = 48 * 1024, END2 = OFF2 + (size_t)2 * (BM + BN) * 64 * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[512 / 64];
  float* sPar = reinterpret_cast<float*>(smem + END2);
  for (int i = threadIdx.x; i < BN; i += 512) {  // (cold lines: requested now, read by the LayerNorm epilogue)
    sPar[i] = g2.bias ? g2.bias[i] : 0.f;
    sPar[BN + i] = l.gamma[i];
    sPar[2 * BN + i] = 0.f;
  }
  gemm_bce_tile<T, BN, true>(a, q, smem, red, reinterpret_cast<u32x4*>(smem + OFF2));
  __syncthreads();  // the kept tile is complete, the staging tile dead
  f32x4 acc[(BN / WGN) / 16][(BM / WGM) / 16];
  int64_t m0, n0;
  gemm_mainloop<T, BM, BN, WGM, WGN, 64, true, false, true>(g2, smem + OFF2, acc, m0, n0);
  gemm_epilogue_ln<T, BM, BN, WGM, WGN, 2>(g2, l, smem, acc, m0, nullptr, 0, nullptr, 0, sPar);
}

template <typename T>
static int launch_gemm_bce_dgrad_ln(const mst_gemm_args& a, const mst_bce_args& q, const mst_gemm_args& g2, const mst_ln_args& l, hipStream_t s) {
  const size_t lds = (size_t)48 * 1024 + (size_t)2 * (64 + 128) * 64 * 2 + (size_t)3 * 128 * 4;
  static bool opted = false;
  if (!opted) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bce_dgrad_ln_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("gemm_bce_dgrad_ln_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
    opted = true;
  }
  hipLaunchKernelGGL((gemm_bce_dgrad_ln_kernel<T>), dim3((unsigned)cdiv(a.M, 64)), dim3(512), lds, s, a, q, g2, l);
  MST_CHECK_LAUNCH("gemm_bce_dgrad_ln_kernel");
  return MST_OK;
}

template <typename T, int BN>
static int launch_gemm_bce(const mst_gemm_args& a, const mst_bce_args& q, hipStream_t s) {
  const size_t lds_loop = (size_t)2 * (64 + BN) * 64 * 2, lds_epi = (size_t)64 * (BN + 4) * 4;
  const size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  if (lds > 64 * 1024) {
    static bool opted = false;
    if (!opted) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bce_kernel<T, BN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("gemm_bce_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
      opted = true;
    }
  }
  hipLaunchKernelGGL((gemm_bce_kernel<T, BN>), dim3((unsigned)(cdiv(a.M, 64) * (a.N / BN))), dim3(512), lds, s, a, q);
  MST_CHECK_LAUNCH("gemm_bce_kernel");
  return MST_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The whole feed-forward block of a Transformer layer in ONE launch (mst_ffn_ln_fwd):
//     a  = dropout(relu(x W1^T + b1))                 (transformer.py:38-40 / 152-153)
//     h2 = epi(a W2^T + b2) with the layer's residual form, y = LayerNorm(h2)      (transformer.py:157-158 / 199-200)
// = mst_gemm_nt(ff1) followed by mst_gemm_nt_ln(ff2, mode 1), bit for bit: the same MFMA sequence per output element (K
// in the same order), the same epilogues. A workgroup owns 64 rows for both GEMMs. The hidden activation is produced
// in chunks of BN (= the model width) columns: one chunk is a 64 x BN tile of the first GEMM (K = BN), finished in
// registers (bias, ReLU, dropout, rounding), parked in LDS as 16-bit — from where it is the A operand of the second GEMM's
// K-slice for that chunk, and is copied out to `a` for the backward pass with full-line stores. So the hidden tensor
// (33 MB at configs[1]) is written once and never read back, the 64 x BN accumulators of the second GEMM stay in
// registers across the F / BN chunks, and three launches (FFN1, FFN2, LayerNorm) become one. Weights stream through a
// double-buffered LDS stage exactly as in gemm_mainloop (both GEMMs of a chunk are stages of ONE pipelined stream);
// every workgroup reads both matrices once (1 MB at configs[1]: ~8 us at the ~127 GB/s a CU pulls from L2).
// MODE 2 is the block's backward pass with the same skeleton (mst_ffn_ln_bwd): the first GEMM is the FFN2 dgrad
// (d(pre-activation) = (dff W2) * alpha, then the ReLU gate a > 0), the second the FFN1 dgrad with the LayerNorm backward in
// its epilogue (mst_gemm_nt_ln mode 2). The gate is applied in a row-layout pass over the parked chunk (coalesced 16-byte
// reads of `a`), which is also the pass that stores the chunk for the weight-gradient launch.
// LEAD (backward only): the block's input tile is not loaded but COMPUTED — the layer's leading LayerNorm backward on the
// workgroup's 64 rows (mst_ffn_ln_bwd_lead), one launch and one 8 + 8 MB round trip less.
// FULL: M is a multiple of 64 (no row guards). The guards, like every other conditional load in the stage loop, are not
// free: hipcc cannot count outstanding loads across a branch and falls back to s_waitcnt vmcnt(0), which drains the
// weight ring — the launch is bound by a single workgroup's serial latency (35 us for ONE workgroup, 42 for 256), so
// every such drain is a full L2 round trip on the critical path. Hence also: bias of the first GEMM read from LDS
// (it was a global load + vmcnt(0) inside the chunk epilogue), prefetches issued unconditionally (clamped).
// EXTRA: one more width x width GEMM on the workgroup's rows in the same launch (gx; its weights are extra stages of the
// same stream). Forward (mst_proj_ffn_ln_fwd): the attention output projection + residual + LayerNorm in FRONT — the input
// tile is the attention output, the block's input x1 = LayerNorm(h1) is computed by mst_gemm_nt_ln's forward epilogue
// (gx, lnx) into the x tile (and stored, with h1 and the statistics, for the backward pass). (The mirror image — the projection's
// dgrad behind the backward block — and a form with every wave loading its own weight fragments straight into MFMA operand
// registers were built, measured slower / not worth a third shadow layout, and removed: docs/kernel_notes.md.)
template <typename T, int BN, int WGM, int WGN, int MODE, bool LEAD, bool FULL, bool EXTRA = false>
__global__ __launch_bounds__(WGM * WGN * 64) void ffn_ln_kernel(mst_gemm_args g1, mst_gemm_args g2, mst_ln_args ln, mst_ln_bwd_in lead,
                                                                mst_gemm_args gx, mst_ln_args lnx) {
  constexpr bool HEAD = EXTRA;
  static_assert(!EXTRA || MODE == 1, "the extra GEMM is the forward form's head");
  constexpr int BM = 64, BK = 64, CHUNKS = BK / 8;
  constexpr int NT = WGM * WGN * 64;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, TM = WTM / 16, TN = WTN / 16;
  constexpr int B_CH = BN * CHUNKS / NT;  // 16-byte pieces of a weight stage per thread
  constexpr int LDA = BN + 8;                  // row stride (elements) of the two activation tiles: conflict-free b128 reads
  constexpr int KST = BN / BK;                 // K stages of one GEMM of a chunk (K = BN for both)
  static_assert(BN * CHUNKS % NT == 0 && (BM * BN / 8) % NT == 0, "tile/threads mismatch");
  typedef typename Act<T>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // [weight stages 2 x BN x 64][hidden chunk 64 x LDA][x tile 64 x LDA]; the LayerNorm epilogue's fp32 staging tile reuses
  // the first two regions (both dead by then)
  u32x4* sB = reinterpret_cast<u32x4*>(smem);
  T* sH = reinterpret_cast<T*>(smem + (size_t)2 * BN * BK * 2);
  T* sX = sH + BM * LDA;
  float* sBias1 = reinterpret_cast<float*>(sX + BM * LDA);  // [F] the first GEMM's bias (zeros without one)
  float* sPar = sBias1 + g1.N;   // [2][3 BN]: bias | gamma | beta of the final epilogue, then of the head's (EXTRA forward)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN, frow = lane & 15, fq = lane >> 4;
#ifndef MST_XCD_ROWS
#define MST_XCD_ROWS 1
#endif
  // (row tiles in XCD-contiguous eighths, like the GEMMs' tiles and the attention workgroups: common.hpp xcd_chunk)
  int64_t m0 = (MST_XCD_ROWS ? xcd_chunk(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x) * BM;
  // Row groups (g1's A remap, the only remap the block takes): the block's M rows are rows [offset, offset + rows_per_group) of
  // every group of `stride` physical rows — the last decoder layer skips each sample's position-0 row, whose output is dropped
  // before the loss (model.py:253): 64 x 256 rows are 256 tiles, one resident round, where 64 x 257 were 257. Groups are whole
  // tiles (host check), so the tile moves as a block and every row address below is m0 + row as before; the dropout counters and
  // the LayerNorm statistics stay indexed by the PHYSICAL row.
  if (g1.a_rows_per_group > 0) {
    const uint32_t grp = (uint32_t)m0 / (uint32_t)g1.a_rows_per_group;
    m0 += (int64_t)grp * (g1.a_group_stride - g1.a_rows_per_group) + g1.a_group_offset;
    // (the epilogues' row guards compare physical rows against M: every row of a whole tile exists)
    g1.M = g2.M = gx.M = (g1.M / g1.a_rows_per_group) * g1.a_group_stride;
  }
  const int64_t F = g1.N;
  FFN_STAMP(0); FFN_RT(190);
  const int64_t Mg = FULL ? (int64_t)1 << 62 : g1.M;  // row guards compare against this (FULL: always true, folded away)
  for (int i = tid * 4; i < (int)F; i += NT * 4)
    *reinterpret_cast<f32x4*>(sBias1 + i) = g1.bias ? *reinterpret_cast<const f32x4*>(g1.bias + i) : f32x4{0.f, 0.f, 0.f, 0.f};
  // (the step's dropout seed words too: a scalar load at an epilogue's start is one more exposed round trip)
  const uint64_t seed2 = g2.dropout_seed ^ ((g2.dropout_p > 0.f && g2.dropout_seed_ptr) ? g2.dropout_seed_ptr[0] : 0ull);
  const uint64_t seedx = EXTRA ? gx.dropout_seed ^ ((gx.dropout_p > 0.f && gx.dropout_seed_ptr) ? gx.dropout_seed_ptr[0] : 0ull) : 0ull;
  for (int i = tid; i < BN; i += NT) {
    sPar[i] = g2.bias ? g2.bias[i] : 0.f;
    sPar[BN + i] = ln.gamma[i];
    sPar[2 * BN + i] = (MODE == 1) ? ln.beta[i] : 0.f;
    if constexpr (HEAD) {
      sPar[3 * BN + i] = gx.bias ? gx.bias[i] : 0.f;
      sPar[4 * BN + i] = lnx.gamma[i];
      sPar[5 * BN + i] = lnx.beta[i];
    }
  }
  const int n_chunks = (int)(F / BN);
  // Chunk order rotated per workgroup: every workgroup streams BOTH weight matrices in full, and 256 of them walking the
  // same lines in lockstep hit the same L2 channels at the same time. Workgroup i of an XCD starts at hidden chunk
  // i mod n_chunks; the second GEMM's sum over the chunks then runs in rotated order (fp32, a different rounding order
  // than the three-launch form; `a` itself is unchanged).
#ifndef MST_FFN_ROT
#define MST_FFN_ROT 1
#endif
#ifndef MST_FFN_EARLY_STORE
#define MST_FFN_EARLY_STORE 0  /* measured 43.9 vs 42.6 us at width 256: slower, kept as a switch */
#endif
  const int rot = MST_FFN_ROT ? (int)((blockIdx.x / 8) % (unsigned)n_chunks) : 0;
  auto phys = [&](int c) { const int pc = c + rot; return pc >= n_chunks ? pc - n_chunks : pc; };
  // ... and the K stages inside each GEMM of a chunk start at a per-workgroup offset too (KST is a power of two)
#ifndef MST_FFN_ROTK
#define MST_FFN_ROTK 0  /* measured: no effect (43.9 vs 43.6 us); off keeps `a` bit-identical to the three-launch form */
#endif
  static_assert((KST & (KST - 1)) == 0, "stage rotation masks with KST - 1");
  const int rotk = MST_FFN_ROTK ? (int)((blockIdx.x / 8 / (unsigned)n_chunks) & (KST - 1)) : 0;
  auto kstage = [&](int s) { return (s + rotk) & (KST - 1); };  // stage s of a GEMM reads K slice kstage(s)
  const T* __restrict__ W1 = reinterpret_cast<const T*>(g1.B);
  const T* __restrict__ W2 = reinterpret_cast<const T*>(g2.B);
  const T* __restrict__ WX = reinterpret_cast<const T*>(gx.B);  // EXTRA: chunk -1 (head) / chunk n_chunks (tail) of the stream

  // ---- weight stream: stage s of chunk c is GEMM 1 (s < KST: W1 rows c*BN.., columns s*64..) or GEMM 2 (W2 rows 0..BN-1,
  // columns c*BN + (s-KST)*64..). Per-thread element offsets are constants; the uniform base moves.
  uint32_t off1[B_CH], off2[B_CH], offx[B_CH];
  int b_lds[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int c = tid + i * NT;
    const int row = c / CHUNKS, ch = c % CHUNKS;
    off1[i] = (uint32_t)row * (uint32_t)g1.ldb + (uint32_t)ch * 8u;
    off2[i] = (uint32_t)row * (uint32_t)g2.ldb + (uint32_t)ch * 8u;
    offx[i] = EXTRA ? (uint32_t)row * (uint32_t)gx.ldb + (uint32_t)ch * 8u : 0u;
    b_lds[i] = row * CHUNKS + (ch ^ (row & 7));
  }
  // The stream runs AHEAD stages in front of the MFMAs, in a register ring: with one 8-wave workgroup per CU (BN = 256:
  // 133 KB of LDS) nothing else hides a weight load's ~1.5 us, and a single stage of lookahead (gemm_mainloop's scheme,
  // which relies on 2-5 co-resident workgroups) made every stage as long as that latency: 52 us for the launch.
  constexpr int SPC = 2 * KST;                 // stages per chunk (a multiple of the ring: slots are compile-time)
  constexpr int RING = BN >= 256 ? 4 : 2, AHEAD = RING - 1;
  static_assert(SPC % RING == 0, "ring slots must repeat per chunk");
  u32x4 ring[RING][B_CH];
  auto load_stage = [&](int c, int s, u32x4 (&rb)[B_CH]) {  // (c, s) uniform
    if (HEAD && c < 0) {  // the extra GEMM's K stage s
      const T* base = WX + kstage(s) * BK;
#pragma unroll
      for (int i = 0; i < B_CH; ++i) rb[i] = *reinterpret_cast<const u32x4*>(base + offx[i]);
    } else if (s < KST) {
      const T* base = W1 + (int64_t)phys(c) * BN * g1.ldb + kstage(s) * BK;
#pragma unroll
      for (int i = 0; i < B_CH; ++i) rb[i] = *reinterpret_cast<const u32x4*>(base + off1[i]);
    } else {
      const T* base = W2 + (int64_t)phys(c) * BN + kstage(s - KST) * BK;
#pragma unroll
      for (int i = 0; i < B_CH; ++i) rb[i] = *reinterpret_cast<const u32x4*>(base + off2[i]);
    }
  };
  auto load_piece = [&](int c, int s, u32x4 (&rb)[B_CH], auto ic) {  // one 16-byte piece of load_stage
    constexpr int i = decltype(ic)::value;
    const T* base;
    uint32_t off;
    if (HEAD && c < 0) { base = WX + kstage(s) * BK; off = offx[i]; }
    else if (s < KST) { base = W1 + (int64_t)phys(c) * BN * g1.ldb + kstage(s) * BK; off = off1[i]; }
    else { base = W2 + (int64_t)phys(c) * BN + kstage(s - KST) * BK; off = off2[i]; }
    rb[i] = *reinterpret_cast<const u32x4*>(base + off);
  };
  auto store_stage = [&](int buf, const u32x4 (&rb)[B_CH]) {
#pragma unroll
    for (int i = 0; i < B_CH; ++i) sB[buf * BN * CHUNKS + b_lds[i]] = rb[i];
  };
  // the first AHEAD stages are requested before the input tile is built: their latency runs under it
  {
    auto pro = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if constexpr (HEAD) {
        static_assert(!HEAD || AHEAD <= KST, "the head GEMM's stages cover the prologue");
        if (j < AHEAD) load_stage(-1, j, ring[j % RING]);
      } else {
        if (j < AHEAD && (j < SPC || n_chunks > 1)) load_stage(j / SPC, j % SPC, ring[j % RING]);
      }
    };
    pro(std::integral_constant<int, 0>()); pro(std::integral_constant<int, 1>()); pro(std::integral_constant<int, 2>());
    pro(std::integral_constant<int, 3>()); pro(std::integral_constant<int, 4>()); pro(std::integral_constant<int, 5>());
    pro(std::integral_constant<int, 6>());
    static_assert(AHEAD <= 7, "the prologue list covers seven stages");
  }
  if constexpr (LEAD) {
    // ---- the input tile = LayerNorm backward of the incoming gradient (the arithmetic of gemm_epilogue_ln's mode 2 on dy)
    constexpr int CPR = BN / 8, RSTEP = NT / CPR, ITERS = BM / RSTEP;
    const int ch = tid % CPR, nc = ch * 8, row0 = tid / CPR;
    const float inv_n = 1.f / (float)BN;
    const bool has_drop = lead.dropout_p > 0.f && lead.mask_mode == 1;
    const uint64_t dseed = lead.dropout_seed ^ ((has_drop && lead.dropout_seed_ptr) ? lead.dropout_seed_ptr[0] : 0ull);
    const uint32_t dkey = dropout_key(dseed, lead.dropout_site), dthr = dropout_thr(lead.dropout_p);
    const float inv_keep = dropout_inv_keep(lead.dropout_p);
    float gam8[8], dg8[8], db8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gam8[e] = lead.gamma[nc + e]; dg8[e] = 0.f; db8[e] = 0.f; }
    u32x4 dyv[ITERS], xv[ITERS];
    float mean_r[ITERS], rstd_r[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int64_t m = m0 + row0 + it * RSTEP;
      dyv[it] = u32x4{0u, 0u, 0u, 0u}; xv[it] = dyv[it]; mean_r[it] = 0.f; rstd_r[it] = 0.f;
      if (m < Mg) {
        dyv[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(lead.dy) + m * lead.ld_dy + nc);
        xv[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(lead.x) + m * lead.ld_x + nc);
        mean_r[it] = lead.mean[m];
        rstd_r[it] = lead.rstd[m];
      }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int row = row0 + it * RSTEP;
      const int64_t m = m0 + row;
      Pack8 db, xb;
      db.u = dyv[it]; xb.u = xv[it];
      float xh[8], g[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = bits_to_f32<T>(db.h[e]);
        xh[e] = (bits_to_f32<T>(xb.h[e]) - mean_r[it]) * rstd_r[it];
        g[e] = d * gam8[e];
        s1 += g[e];
        s2 += g[e] * xh[e];
        dg8[e] += d * xh[e];
        db8[e] += d;
      }
      s1 = row_sum<CPR>(s1) * inv_n;
      s2 = row_sum<CPR>(s2) * inv_n;
      uint32_t keep8 = 0xFFu;
      if (has_drop) {
        const uint64_t w = (uint64_t)(m * BN + nc) >> 2;
        keep8 = dropout_keep4k(dkey, w, dthr) | (dropout_keep4k(dkey, w + 1, dthr) << 4);
      }
      Pack8 ob, mb;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float o = rstd_r[it] * (g[e] - s1 - xh[e] * s2);
        const float k = has_drop ? (((keep8 >> e) & 1u) ? inv_keep : 0.f) : 1.f;
        ob.h[e] = f32_to_bits<T>(o);
        mb.h[e] = f32_to_bits<T>(o * k);
      }
      if (m < Mg) {
        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(lead.dx) + m * lead.ld_dx + nc) = ob.u;
        if (lead.mask_mode == 1) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(lead.dx_masked) + m * lead.ld_dxm + nc) = mb.u;
      }
      *reinterpret_cast<u32x4*>(sX + row * LDA + nc) = (m < Mg) ? (lead.mask_mode == 1 ? mb.u : ob.u) : u32x4{0u, 0u, 0u, 0u};
    }
    // dgamma / dbeta: the RSTEP row groups summed through LDS (the weight-stage region is not in use yet)
    float* red = reinterpret_cast<float*>(smem);  // [2][RSTEP][BN]
    static_assert((size_t)2 * RSTEP * BN * 4 <= (size_t)2 * BN * BK * 2, "reduction scratch must fit the weight stages");
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[row0 * BN + nc + e] = dg8[e];
      red[(RSTEP + row0) * BN + nc + e] = db8[e];
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += NT) {
      const int which = c / BN, col = c % BN;
      float sm = 0.f;
      for (int r = 0; r < RSTEP; ++r) sm += red[(which * RSTEP + r) * BN + col];
      if (lead.partials) lead.partials[(int64_t)blockIdx.x * 2 * BN + c] = sm;
      else atomicAdd((which ? lead.dbeta : lead.dgamma) + col, sm);
    }
    __syncthreads();  // the scratch becomes the first weight stage
  } else
  // ---- the x tile (rows past M read as zero)
  {
    // (HEAD: the attention output tile, the extra GEMM's A operand; the block's own input is computed from it below)
    const T* X = reinterpret_cast<const T*>(HEAD ? gx.A : g1.A);
    const int64_t ldx = HEAD ? gx.lda : g1.lda;
    constexpr int CPR = BN / 8;
#pragma unroll
    for (int i = 0; i < BM * CPR / NT; ++i) {
      const int c = tid + i * NT, row = c / CPR, ch = c % CPR;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m0 + row < Mg) v = *reinterpret_cast<const u32x4*>(X + (m0 + row) * ldx + ch * 8);
      *reinterpret_cast<u32x4*>(sX + row * LDA + ch * 8) = v;
    }
  }
  // one 64-deep K stage: acc += A[64, 64] (LDS tile `sA`, columns k0..) x stage `buf`
  // hook(k), k < 2 * TN: called behind the k-th row of MFMAs of the stage — the staged form hangs the weight staging
  // there (MST_FFN_IL), piece by piece, instead of issuing it in front of / behind the whole stage
  auto mma_stage = [&](f32x4 (&acc)[TN][TM], const T* sA, int k0, int buf, const u32x4 (&rb)[B_CH], auto&& hook) {
    const u32x4* cB = sB + buf * BN * CHUNKS;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      vec8 xf[TM], wf[TN];
      const int kc = ks * 4 + fq;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        xf[i] = __builtin_bit_cast(vec8, *reinterpret_cast<const u32x4*>(sA + (wm * WTM + i * 16 + frow) * LDA + k0 + kc * 8));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 16 + frow;
        wf[j] = __builtin_bit_cast(vec8, cB[row * CHUNKS + (kc ^ (row & 7))]);
      }
      auto row = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < TN) {
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[j][i] = Act<T>::mfma16(wf[j], xf[i], acc[j][i]);
          if (ks == 0) hook(std::integral_constant<int, j>()); else hook(std::integral_constant<int, TN + j>());
        }
      };
      static_assert(TN <= 4, "row list");
      row(std::integral_constant<int, 0>()); row(std::integral_constant<int, 1>());
      row(std::integral_constant<int, 2>()); row(std::integral_constant<int, 3>());
    }
  };
  auto no_hook = [](auto) {};

  f32x4 acc1[TN][TM], acc2[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc2[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float p1 = g1.dropout_p;
  const bool drop1 = MODE == 1 && p1 > 0.f;  // (the backward form takes neither dropout nor an activation: host check)
  const uint64_t seed1 = g1.dropout_seed ^ ((drop1 && g1.dropout_seed_ptr) ? g1.dropout_seed_ptr[0] : 0ull);
  const uint32_t dkey1 = dropout_key(seed1, g1.dropout_site), dthr1 = dropout_thr(p1);
  const float inv_keep1 = dropout_inv_keep(p1);
  const bool idx32 = (uint64_t)g1.M * (uint64_t)F < (1ull << 32);  // every element index of the hidden tensor fits 32 bits (uniform)
  const bool relu1 = MODE == 1 && g1.act == MST_ACT_RELU;
  const bool step_form1 = MODE == 1 && relu1 && drop1 && idx32 && g1.alpha == 1.f;  // (x * 1.0f == x bit for bit)
  // dropout counters of this thread's rows, premultiplied (dropout_apply4_pre): ((m0 + row) F / 2) * DROPOUT_MUL mod 2^32 (F % 4 == 0: host check)
  uint32_t rowmul[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rowmul[i] = (uint32_t)(m0 + wm * WTM + i * 16 + frow) * (uint32_t)(F >> 1) * DROPOUT_MUL;
  T* Aout = reinterpret_cast<T*>(g1.C);

  // the KST stages of the extra GEMM (stream position `cx` = -1: in front of the chunks) into acc1
  auto extra_gemm = [&](int cx) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc1[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    store_stage(0, ring[0]);
    __syncthreads();
    auto xstage = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s < KST) {
        constexpr int t = s + AHEAD;
        if constexpr (t < KST) load_stage(cx, t, ring[t % RING]);
        else if constexpr (HEAD) load_stage(0, t - KST, ring[t % RING]);  // the first chunk's stages follow (KST % RING == 0)
        mma_stage(acc1, sX, kstage(s) * BK, s & 1, ring[s % RING], no_hook);
        if constexpr (s + 1 < KST) store_stage((s + 1) & 1, ring[(s + 1) % RING]);
        __syncthreads();
      }
    };
    static_assert(!EXTRA || (KST <= 4 && KST % RING == 0), "the extra GEMM stage list / ring slots");
    xstage(std::integral_constant<int, 0>()); xstage(std::integral_constant<int, 1>());
    xstage(std::integral_constant<int, 2>()); xstage(std::integral_constant<int, 3>());
  };
  if constexpr (HEAD) {
    // h1 = epi(att Wp^T) (+ x), x1 = LayerNorm(h1): mst_gemm_nt_ln's forward epilogue; x1 also lands in the x tile
    extra_gemm(-1);
    gemm_epilogue_ln<T, BM, BN, WGM, WGN, 1>(gx, lnx, smem, acc1, m0, nullptr, 0, sX, LDA, sPar + 3 * BN, &seedx);
    __syncthreads();  // the staging tile (over the weight stages) is dead, the x tile complete
  }
  store_stage(0, ring[0]);
  __syncthreads();  // (also publishes the x tile)
  FFN_STAMP(1);
  for (int c = 0; c < n_chunks; ++c) {
    const int pc = phys(c);  // the hidden chunk this iteration computes
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc1[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // backward: this chunk's gate rows (the forward's hidden activation), requested now, used after the first GEMM
    constexpr int OUT_CH = BM * (BN / 8) / NT;
    u32x4 gv[OUT_CH];
    if constexpr (MODE == 2) {
      const T* G = reinterpret_cast<const T*>(g1.gate);
#pragma unroll
      for (int i = 0; i < OUT_CH; ++i) {
        const int cc = tid + i * NT, row = cc / (BN / 8), ch = cc % (BN / 8);
        gv[i] = u32x4{0u, 0u, 0u, 0u};
        if (m0 + row < Mg) gv[i] = *reinterpret_cast<const u32x4*>(G + (m0 + row) * g1.ldg + (int64_t)pc * BN + ch * 8);
      }
    }
    auto stage = [&](auto sc) {
      constexpr int s = decltype(sc)::value;      // stage within the chunk: ring slot s % RING, LDS buffer s % 2
      if constexpr (s < SPC) {
#ifndef MST_FFN_IL
#define MST_FFN_IL 1
#endif
        constexpr bool IL = MST_FFN_IL && BN >= 256 && !MST_FFN_EARLY_STORE && B_CH <= 2 * TN;  // (width 128, two workgroups per CU: measured 1 us slower)
        // request stage s + AHEAD of the stream (it may belong to the next chunk)
        constexpr int t = s + AHEAD;
        // (unconditional: past the last chunk the clamped load fetches a stage nobody stores)
        const int tc = t < SPC ? c : (c + 1 < n_chunks ? c + 1 : c);
        const bool more = s + 1 < SPC || c + 1 < n_chunks;  // a next stage exists: its weights go to the other LDS buffer
        // interleaved form: piece k of { load of stage s + AHEAD, LDS store of stage s + 1 } behind the k-th row of MFMAs
#ifndef MST_FFN_COPY_IL
#define MST_FFN_COPY_IL 0  /* measured: -1 us on the isolated launch, +1..2 us at step level (in-call A/B): off */
#endif
        // (experiment) forward: the finished chunk's copy to `a` rides behind the last MFMA rows of the second GEMM's first
        // stage, a 16-byte piece of a row at a time, instead of standing between the barrier and that stage
        constexpr bool COPY_IL = MST_FFN_COPY_IL && MODE == 1 && OUT_CH <= 2 * TN;
        auto piece = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (IL && k < B_CH) {
            __builtin_amdgcn_sched_barrier(0);
            load_piece(tc, t % SPC, ring[t % RING], kc);
            if (more) sB[((s + 1) & 1) * BN * CHUNKS + b_lds[k]] = ring[(s + 1) % RING][k];
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (COPY_IL && s == KST && k >= 2 * TN - OUT_CH) {
            constexpr int i = k - (2 * TN - OUT_CH), CPRc = BN / 8;
            const int cc = tid + i * NT, row = cc / CPRc, ch = cc % CPRc;
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 v = *reinterpret_cast<const u32x4*>(sH + row * LDA + ch * 8);
            if (m0 + row < Mg) *reinterpret_cast<u32x4*>(Aout + (m0 + row) * g1.ldc + (int64_t)pc * BN + ch * 8) = v;
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        if constexpr (!IL) load_stage(tc, t % SPC, ring[t % RING]);
        FFN_STAMP(8 + (c * SPC + s) * 4);
        if (MST_FFN_EARLY_STORE) {
          // Experiment (off): the next stage's weights go to the OTHER LDS buffer ahead of this stage's MFMAs (legal: that
          // buffer was last read in the previous stage, which ended with a barrier) instead of after them.
          if (s + 1 < SPC || c + 1 < n_chunks) store_stage((s + 1) & 1, ring[(s + 1) % RING]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (s < KST) mma_stage(acc1, sX, kstage(s) * BK, s & 1, ring[s % RING], piece);
        else mma_stage(acc2, sH, kstage(s - KST) * BK, s & 1, ring[s % RING], piece);
        FFN_STAMP(8 + (c * SPC + s) * 4 + 1);
        if constexpr (s == KST - 1) {
          // ---- chunk epilogue of GEMM 1, in registers: bias, ReLU, dropout, rounding (the order of gemm_epilogue) -> sH.
          // (The previous chunk's GEMM-2 stages, which read sH, ended with a barrier.)
          // Two bodies behind ONE uniform branch: the training step's form (ReLU, alpha 1, dropout, 32-bit counters) without a
          // select or a multiplication per optional feature, and the general one. (Run-time feature flags inside the element loop
          // are if-converted into a v_cndmask each: this epilogue is VALU-issue-bound — 592 vector instructions per chunk and wave
          // before, ~300 in the first body.)
          auto chunk_epilogue = [&](auto step_form) {
            constexpr bool STEP = decltype(step_form)::value;
            const uint32_t cmul = ((uint32_t)pc * (BN / 2) + (uint32_t)((wn * WTN + fq * 4) / 2)) * DROPOUT_MUL;  // this chunk, this lane's columns
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const int n = wn * WTN + j * 16 + fq * 4;   // column within the chunk
              const int64_t col = (int64_t)pc * BN + n;    // hidden unit
              const f32x4 b4 = *reinterpret_cast<const f32x4*>(sBias1 + col);
#pragma unroll
              for (int i = 0; i < TM; ++i) {
                const int row = wm * WTM + i * 16 + frow;
                float tv[4];
                if constexpr (STEP) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) tv[e] = fmaxf(acc1[j][i][e] + b4[e], 0.f);
                  // element index (m0 + row) F + col; the word pair of its group of four = dropout_word32(index / 2), + 1
                  dropout_apply4_pre(dkey1, rowmul[i] + cmul + (uint32_t)(j * 8) * DROPOUT_MUL, dthr1, inv_keep1, tv);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    tv[e] = (acc1[j][i][e] + b4[e]) * g1.alpha;
                    if (relu1) tv[e] = fmaxf(tv[e], 0.f);
                  }
                  if (drop1) dropout_apply4(dkey1, (uint64_t)((m0 + row) * F + col) >> 2, dthr1, inv_keep1, tv);
                }
                uint16_t hb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) hb[e] = f32_to_bits<T>(tv[e]);
                *reinterpret_cast<u32x2*>(sH + row * LDA + n) =
                    u32x2{(uint32_t)hb[0] | ((uint32_t)hb[1] << 16), (uint32_t)hb[2] | ((uint32_t)hb[3] << 16)};
              }
            }
          };
          if (step_form1) chunk_epilogue(std::true_type()); else chunk_epilogue(std::false_type());
        }
        // the next stage of the stream (requested AHEAD iterations ago) -> the other LDS buffer
        if (!IL && !MST_FFN_EARLY_STORE && more) store_stage((s + 1) & 1, ring[(s + 1) % RING]);
        FFN_STAMP(8 + (c * SPC + s) * 4 + 2);
        __syncthreads();
        FFN_STAMP(8 + (c * SPC + s) * 4 + 3);
        if constexpr (s == KST - 1 && !COPY_IL) {
          // the finished chunk goes out to `a` (the backward pass needs it) as whole 16-byte pieces of rows, while the
          // second GEMM's stages run
          constexpr int CPR = BN / 8;
#pragma unroll
          for (int i = 0; i < BM * CPR / NT; ++i) {
            const int cc = tid + i * NT, row = cc / CPR, ch = cc % CPR;
            u32x4 v = *reinterpret_cast<const u32x4*>(sH + row * LDA + ch * 8);
            if constexpr (MODE == 2) {  // ReLU backward: pass where the forward activation was positive (gemm_epilogue's gate)
              Pack8 pv, pg;
              pv.u = v; pg.u = gv[i];
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (!(bits_to_f32<T>(pg.h[e]) > 0.f)) pv.h[e] = 0;
              v = pv.u;
              *reinterpret_cast<u32x4*>(sH + row * LDA + ch * 8) = v;
            }
            if (m0 + row < Mg) *reinterpret_cast<u32x4*>(Aout + (m0 + row) * g1.ldc + (int64_t)pc * BN + ch * 8) = v;
          }
          if constexpr (MODE == 2) __syncthreads();  // the gated chunk is what the second GEMM reads
        }
      }
    };
    static_assert(SPC <= 8, "the stage list below covers eight stages per chunk");
    stage(std::integral_constant<int, 0>()); stage(std::integral_constant<int, 1>());
    stage(std::integral_constant<int, 2>()); stage(std::integral_constant<int, 3>());
    stage(std::integral_constant<int, 4>()); stage(std::integral_constant<int, 5>());
    stage(std::integral_constant<int, 6>()); stage(std::integral_constant<int, 7>());
  }
  // ---- the second GEMM's epilogue + LayerNorm: exactly mst_gemm_nt_ln's (staging tile over the dead weight / hidden regions)
  // (a residual that IS the block's input — the encoder's x1 + dropout(ff) — is taken from the x tile in LDS)
  const bool resid_is_x = g2.resid == g1.A && g2.ldr == g1.lda;
  FFN_STAMP(2);
  gemm_epilogue_ln<T, BM, BN, WGM, WGN, MODE>(g2, ln, smem, acc2, m0, resid_is_x ? sX : nullptr, LDA, nullptr, 0, sPar, &seed2);
  FFN_STAMP(3); FFN_RT(191);
}

template <typename T, int BN>
static int launch_ffn_ln(const mst_gemm_args& g1, const mst_gemm_args& g2, const mst_ln_args& ln, const mst_ln_bwd_in* lead, hipStream_t s,
                         const mst_gemm_args* gx = nullptr, const mst_ln_args* lnx = nullptr) {
  constexpr int BM = 64;
  const int ex = gx ? 1 : 0;
  const size_t lds_loop = (size_t)2 * BN * 64 * 2 + (size_t)2 * BM * (BN + 8) * 2 + (size_t)g1.N * 4 + (size_t)6 * BN * 4;
  const size_t lds_epi = (size_t)BM * (BN + 4) * 4;
  const size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  const int full = g1.M % BM == 0 ? 1 : 0;
  // [forward | backward | backward with the leading LayerNorm] x [row guards | whole tiles], then the forward form with the projection head
  const int mi = ex ? 6 + full : (lead ? 2 : (ln.mode == 2 ? 1 : 0)) * 2 + full;
  typedef void (*kern_t)(mst_gemm_args, mst_gemm_args, mst_ln_args, mst_ln_bwd_in, mst_gemm_args, mst_ln_args);
  const kern_t fns[8] = {&ffn_ln_kernel<T, BN, 2, 4, 1, false, false>, &ffn_ln_kernel<T, BN, 2, 4, 1, false, true>,
                         &ffn_ln_kernel<T, BN, 2, 4, 2, false, false>, &ffn_ln_kernel<T, BN, 2, 4, 2, false, true>,
                         &ffn_ln_kernel<T, BN, 2, 4, 2, true, false>, &ffn_ln_kernel<T, BN, 2, 4, 2, true, true>,
                         &ffn_ln_kernel<T, BN, 2, 4, 1, false, false, true>, &ffn_ln_kernel<T, BN, 2, 4, 1, false, true, true>};
  static size_t opted[8] = {64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024, 64 * 1024};  // LDS each kernel is opted in for
  if (lds > opted[mi]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[mi]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("ffn_ln_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
    opted[mi] = lds;
  }
  const mst_ln_bwd_in none = {};
  const mst_gemm_args no_gemm = {};
  const mst_ln_args no_ln = {};
  hipLaunchKernelGGL(fns[mi], dim3((unsigned)cdiv(g1.M, BM)), dim3(512), lds, s, g1, g2, ln, lead ? *lead : none,
                     gx ? *gx : no_gemm, lnx ? *lnx : no_ln);
  MST_CHECK_LAUNCH("ffn_ln_kernel");
  return MST_OK;
}

template <typename T, int BM, int BN, int WGM, int WGN>
static int launch_gemm_ln(const mst_gemm_args& a, const mst_ln_args& l, hipStream_t s) {
  const size_t lds_loop = (size_t)2 * (BM + BN) * 64 * 2, lds_epi = (size_t)BM * (BN + 4) * 4;
  const size_t lds = (lds_loop > lds_epi ? lds_loop : lds_epi) + (size_t)3 * BN * 4;  // + bias | gamma | beta
  dim3 grid((unsigned)cdiv(a.M, BM)), block(WGM * WGN * 64);
  const int mi = l.mode == 2 ? 1 : 0;
  const void* fn = mi ? reinterpret_cast<const void*>(&gemm_nt_ln_kernel<T, BM, BN, WGM, WGN, 2>)
                      : reinterpret_cast<const void*>(&gemm_nt_ln_kernel<T, BM, BN, WGM, WGN, 1>);
  if (lds > 64 * 1024) {
    static bool opted[2] = {false, false};
    if (!opted[mi]) {
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("gemm_nt_ln_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
      opted[mi] = true;
    }
  }
  if (mi) hipLaunchKernelGGL((gemm_nt_ln_kernel<T, BM, BN, WGM, WGN, 2>), grid, block, lds, s, a, l);
  else hipLaunchKernelGGL((gemm_nt_ln_kernel<T, BM, BN, WGM, WGN, 1>), grid, block, lds, s, a, l);
  MST_CHECK_LAUNCH("gemm_nt_ln_kernel");
  return MST_OK;
}

// 128 x 128 tiles at THREE or four workgroups per CU: 32-deep K stages (32 KB) and the accumulators staged one 64-row block at
// a time (34 KB) instead of 64 KB + 68 KB. For launches of 513..768 tiles (the K | Q | V projections: 768) that is one
// resident round instead of a full one and a half-empty one at two per CU. Eligibility as launch_gemm's fast form without
// dropout / row ops; returns 1 when the launch is not eligible (caller falls back).
template <typename T>
static int launch_gemm_3cu(const mst_gemm_args& a, hipStream_t s) {
  constexpr int BM = 128, BN = 128, BK = 32;
  const bool rowops = a.rowadd || a.grpadd || a.a_rows_per_group > 0 || a.c_rows_per_group > 0;
  const bool ok = !a.c_f32 && !a.a_u8 && !rowops && a.dropout_p == 0.f && !a.self_resid && a.M % BM == 0 && a.N % BN == 0 && a.K % BK == 0 &&
                  a.ldc % 8 == 0 && (uint64_t)a.M * (uint64_t)a.N < (1ull << 32) &&
                  (!a.resid || (a.ldr % 8 == 0 && (uintptr_t)a.resid % 16 == 0)) && (!a.gate || (a.ldg % 8 == 0 && (uintptr_t)a.gate % 16 == 0));
  if (!ok) return 1;
  const size_t lds_loop = (size_t)2 * (BM + BN) * BK * 2, lds_epi = (size_t)(BM / 2) * (BN + 4) * 4;
  const size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  hipLaunchKernelGGL((gemm_nt_kernel<T, BM, BN, 2, 2, false, BK, false, 4, false>), dim3((unsigned)((a.M / BM) * (a.N / BN))), dim3(256), lds, s, a);
  MST_CHECK_LAUNCH("gemm_nt_kernel");
  return MST_OK;
}

// every tile interior and every optional operand 16-byte friendly: the launch takes the kernel that holds only the fast row loop
template <int BM, int BN>
static bool gemm_fast_form(const mst_gemm_args& a) {
  const bool rowops = a.rowadd || a.grpadd || a.a_rows_per_group > 0 || a.c_rows_per_group > 0;
  const int64_t phys_rows = a.c_rows_per_group > 0 ? (a.M / a.c_rows_per_group + 1) * a.c_group_stride + a.c_group_offset : a.M;
  return !a.c_f32 && a.M % BM == 0 && a.N % BN == 0 && a.ldc % 8 == 0 &&
         (a.c_rows_per_group <= 0 || a.c_rows_per_group % BM == 0) &&
         (uint64_t)phys_rows * (uint64_t)a.N < (1ull << 32) &&
         (!a.resid || (a.ldr % 8 == 0 && (uintptr_t)a.resid % 16 == 0)) &&
         (!a.gate || (a.ldg % 8 == 0 && (uintptr_t)a.gate % 16 == 0)) &&
         (!rowops || (a.rowadd_period % BM == 0 && (!a.rowadd || (a.ldra % 4 == 0 && (uintptr_t)a.rowadd % 16 == 0)) &&
                      (!a.grpadd || (a.ldga % 4 == 0 && (uintptr_t)a.grpadd % 16 == 0))));
}

template <typename T, int BM, int BN, int WGM, int WGN, int BK = 64>
static int launch_gemm(const mst_gemm_args& a, hipStream_t s) {
  const int64_t tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
  const size_t lds_loop = (size_t)2 * (BM + BN) * BK * 2, lds_epi = (size_t)BM * (BN + 4) * 4;
  const size_t lds = lds_loop > lds_epi ? lds_loop : lds_epi;
  dim3 grid((unsigned)tiles), block(WGM * WGN * 64);
  // ("row ops" in the kernel choice: row-indexed adds or a row remap of A or C)
  const bool rowops = a.rowadd || a.grpadd || a.a_rows_per_group > 0 || a.c_rows_per_group > 0;
  const bool fast = gemm_fast_form<BM, BN>(a) && a.K % BK == 0;  // (the fast kernels' K loop is unguarded too)
  const bool drop = a.dropout_p > 0.f || a.self_resid;
  // kernels: [row-ops][fast without dropout | fast with dropout | general 16-bit | general fp32]
  // (an epilogue finished in accumulator layout — 8-byte accesses, no LDS round trip — measured +4 us per step and was removed)
  const int variant = a.a_u8 ? (fast ? 8 : 9) : (a.c_f32 ? 3 : (fast ? (drop ? 1 : 0) : 2)) + (rowops ? 4 : 0);
  typedef void (*kern_t)(mst_gemm_args);
  // [8], [9]: uint8 A operand (the piano-roll embedding GEMMs: row ops, 16-bit C, no dropout), fast / general
  const kern_t fns[10] = {&gemm_nt_kernel<T, BM, BN, WGM, WGN, false, BK, false, 1, false>, &gemm_nt_kernel<T, BM, BN, WGM, WGN, false, BK, false, 1, true>,
                         &gemm_nt_kernel<T, BM, BN, WGM, WGN, false, BK, false, 2, true>, &gemm_nt_kernel<T, BM, BN, WGM, WGN, true, BK, false, 2, true>,
                         &gemm_nt_kernel<T, BM, BN, WGM, WGN, false, BK, true, 1, false>, &gemm_nt_kernel<T, BM, BN, WGM, WGN, false, BK, true, 1, true>,
                         &gemm_nt_kernel<T, BM, BN, WGM, WGN, false, BK, true, 2, true>, &gemm_nt_kernel<T, BM, BN, WGM, WGN, true, BK, true, 2, true>,
                         &gemm_nt_kernel<T, BM, BN, WGM, WGN, false, BK, true, 1, false, true>, &gemm_nt_kernel<T, BM, BN, WGM, WGN, false, BK, true, 2, false, true>};
  if (lds > 64 * 1024) {  // dynamic LDS above 64 KB has to be opted into, once per kernel
    static bool opted[10] = {false, false, false, false, false, false, false, false, false, false};
    if (!opted[variant]) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fns[variant]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("gemm_nt_kernel: LDS opt-in of %zu bytes: %s", lds, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
      opted[variant] = true;
    }
  }
  hipLaunchKernelGGL(fns[variant], grid, block, lds, s, a);
  MST_CHECK_LAUNCH("gemm_nt_kernel");
  return MST_OK;
}

}  // namespace mst

using namespace mst;

static int check_gemm_common(const mst_gemm_args& a) {
  MST_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "mst_gemm_nt: M,N,K must be positive (got %lld,%lld,%lld)",
                (long long)a.M, (long long)a.N, (long long)a.K);
  MST_CHECK_ARG(a.K % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0,
                "mst_gemm_nt: K, lda, ldb must be multiples of 8 (got %lld,%lld,%lld)", (long long)a.K,
                (long long)a.lda, (long long)a.ldb);
  MST_CHECK_ARG(a.A && a.B && a.C, "mst_gemm_nt: null operand");
  MST_CHECK_ARG(a.dropout_p >= 0.f && a.dropout_p < 1.f, "mst_gemm_nt: dropout_p must be in [0,1)");
  MST_CHECK_ARG(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.B % 16 == 0) && ((uintptr_t)a.C % 16 == 0),
                "mst_gemm_nt: operands must be 16-byte aligned");
  return MST_OK;
}

extern "C" int64_t mst_gemm_nt_ln_parts(int64_t M) { return M > 0 ? cdiv(M, 64) : 0; }  // launch_gemm_ln's 64-row tiles

static int ffn_ln_impl(const char* who, const mst_gemm_args* first, const mst_gemm_args* second, const mst_ln_args* ln, int mode,
                       mst_stream_t stream, const mst_ln_bwd_in* lead = nullptr, const mst_gemm_args* extra = nullptr,
                       const mst_ln_args* extra_ln = nullptr) {
  MST_CHECK_ARG(first != nullptr && second != nullptr && ln != nullptr, "%s: null args", who);
  const mst_gemm_args& a = *first;
  const mst_gemm_args& b = *second;
  const mst_ln_args& l = *ln;
  int rc = check_gemm_common(a);
  if (rc) return rc;
  rc = check_gemm_common(b);
  if (rc) return rc;
  MST_CHECK_ARG(a.dtype == b.dtype && a.M == b.M, "%s: the two GEMMs must share dtype and M", who);
  MST_CHECK_ARG((b.N == 256 || b.N == 128) && a.K == b.N, "%s: the model width (first K = second N) must be 128 or 256 (got %lld, %lld)", who,
                (long long)a.K, (long long)b.N);
  MST_CHECK_ARG(a.N == b.K && a.N % b.N == 0, "%s: the hidden width (first N = second K) must be a multiple of the model width", who);
  MST_CHECK_ARG(b.A == a.C && b.lda == a.ldc, "%s: the second GEMM's A operand must be the first one's output (it is consumed on chip)", who);
  MST_CHECK_ARG(!a.c_f32 && !b.c_f32 && !b.gate && !a.rowadd && !b.rowadd && !a.grpadd && !b.grpadd && !a.resid && !a.self_resid &&
                b.act == MST_ACT_NONE && a.c_rows_per_group <= 0 && b.a_rows_per_group <= 0 && b.c_rows_per_group <= 0,
                "%s: fp32 outputs, row-indexed adds, row remaps, a gate or activation on the second GEMM and a residual on the first are not supported", who);
  // ... except the first GEMM's A remap, which stands for the whole block: its M rows are rows [offset, offset + rows_per_group)
  // of every `stride` physical rows, in every operand of the launch (groups and M in whole 64-row tiles)
  MST_CHECK_ARG(a.a_rows_per_group <= 0 || (a.a_rows_per_group % 64 == 0 && a.M % a.a_rows_per_group == 0 && a.a_group_offset >= 0 &&
                                            a.a_group_stride >= a.a_rows_per_group + a.a_group_offset && a.M < (1ll << 31)),
                "%s: row groups must be whole 64-row tiles (rows per group %lld, stride %lld, offset %lld, M %lld)", who,
                (long long)a.a_rows_per_group, (long long)a.a_group_stride, (long long)a.a_group_offset, (long long)a.M);
  MST_CHECK_ARG(a.lda % 8 == 0 && a.ldc % 8 == 0 && a.ldc >= a.N && b.ldc % 8 == 0 && b.ldc >= b.N, "%s: leading dimensions must be multiples of 8", who);
  MST_CHECK_ARG((uint64_t)a.N * (uint64_t)a.ldb < (1ull << 32) && (uint64_t)b.N * (uint64_t)b.ldb < (1ull << 32), "%s: weight matrices too large", who);
  MST_CHECK_ARG(!b.resid || (b.ldr % 8 == 0 && b.ldr >= b.N && (uintptr_t)b.resid % 16 == 0), "%s: bad residual layout", who);
  MST_CHECK_ARG(!a.bias || (uintptr_t)a.bias % 16 == 0, "%s: the first GEMM's bias must be 16-byte aligned", who);
  MST_CHECK_ARG(a.dropout_p == 0.f || a.N % 4 == 0, "%s: dropout needs widths that are multiples of 4", who);
  MST_CHECK_ARG(l.mode == mode && l.gamma && l.mean && l.rstd, "%s: LayerNorm arguments of the wrong form", who);
  if (mode == 1) {
    MST_CHECK_ARG(!a.gate, "%s: a gate belongs to the backward form", who);
    MST_CHECK_ARG(l.beta && l.out && l.ld_out % 8 == 0 && l.ld_out >= b.N && (uintptr_t)l.out % 16 == 0,
                  "%s: the LayerNorm arguments are those of mst_gemm_nt_ln's forward form", who);
  } else {
    MST_CHECK_ARG(a.gate && a.ldg % 8 == 0 && a.ldg >= a.N && (uintptr_t)a.gate % 16 == 0, "%s: the first GEMM needs the forward activation as its gate", who);
    MST_CHECK_ARG(a.act == MST_ACT_NONE && a.dropout_p == 0.f && !b.self_resid, "%s: activation / dropout / self_resid belong to the forward form", who);
    MST_CHECK_ARG(l.x && l.ld_x % 8 == 0 && (uintptr_t)l.x % 16 == 0 && (l.partials || (l.dgamma && l.dbeta)) && (uintptr_t)l.partials % 16 == 0,
                  "%s: backward needs x and dgamma + dbeta (or partials)", who);
    MST_CHECK_ARG(l.mask_mode >= 0 && l.mask_mode <= 2 &&
                  (l.mask_mode != 1 || (l.out && l.ld_out % 8 == 0 && l.ld_out >= b.N && (uintptr_t)l.out % 16 == 0)),
                  "%s: mask_mode must be 0, 1 (with out) or 2", who);
  }
  if (lead) {
    const mst_ln_bwd_in& q = *lead;
    MST_CHECK_ARG(mode == 2, "%s: a leading LayerNorm belongs to the backward form", who);
    MST_CHECK_ARG(q.dy && q.x && q.gamma && q.mean && q.rstd && q.dx && (q.partials || (q.dgamma && q.dbeta)), "%s: leading LayerNorm: null pointer", who);
    MST_CHECK_ARG(q.ld_dy % 8 == 0 && q.ld_x % 8 == 0 && q.ld_dx % 8 == 0 && q.ld_dy >= b.N && q.ld_x >= b.N && q.ld_dx >= b.N &&
                  ((uintptr_t)q.dy | (uintptr_t)q.x | (uintptr_t)q.dx | (uintptr_t)q.partials) % 16 == 0, "%s: leading LayerNorm: bad layout", who);
    MST_CHECK_ARG(q.mask_mode == 0 || (q.mask_mode == 1 && q.dx_masked && q.ld_dxm % 8 == 0 && q.ld_dxm >= b.N && (uintptr_t)q.dx_masked % 16 == 0),
                  "%s: leading LayerNorm: mask_mode must be 0 or 1 (with dx_masked)", who);
    MST_CHECK_ARG(q.dropout_p >= 0.f && q.dropout_p < 1.f, "%s: leading LayerNorm: dropout_p must be in [0,1)", who);
    const void* tile = q.mask_mode == 1 ? q.dx_masked : q.dx;
    const int64_t tile_ld = q.mask_mode == 1 ? q.ld_dxm : q.ld_dx;
    MST_CHECK_ARG(a.A == tile && a.lda == tile_ld, "%s: the first GEMM's A operand must be the leading LayerNorm's (masked) output", who);
  }
  if (extra) {
    const mst_gemm_args& x = *extra;
    rc = check_gemm_common(x);
    if (rc) return rc;
    MST_CHECK_ARG(x.dtype == a.dtype && x.M == a.M && x.N == b.N && x.K == b.N, "%s: the extra GEMM is width x width on the same rows", who);
    MST_CHECK_ARG(!x.c_f32 && !x.gate && !x.rowadd && !x.grpadd && x.act == MST_ACT_NONE && x.a_rows_per_group <= 0 && x.c_rows_per_group <= 0 &&
                  x.ldc % 8 == 0 && x.ldc >= x.N && (uint64_t)x.N * (uint64_t)x.ldb < (1ull << 32),
                  "%s: the extra GEMM takes no gate, activation, row-indexed add or row remap", who);
    MST_CHECK_ARG(mode == 1 && extra_ln != nullptr, "%s: the projection (with its LayerNorm) rides in front of the forward form", who);
    const mst_ln_args& q = *extra_ln;
    MST_CHECK_ARG(q.mode == 1 && q.gamma && q.beta && q.mean && q.rstd && q.out == a.A && q.ld_out == a.lda,
                  "%s: the leading LayerNorm's output must be the first GEMM's A operand", who);
    MST_CHECK_ARG(!x.resid || (x.ldr % 8 == 0 && x.ldr >= x.N && (uintptr_t)x.resid % 16 == 0), "%s: bad residual layout", who);
  }
  hipStream_t s = (hipStream_t)stream;
  return dispatch_act(a.dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    if (b.N == 256) return launch_ffn_ln<T, 256>(a, b, l, lead, s, extra, extra_ln);
    return launch_ffn_ln<T, 128>(a, b, l, lead, s, extra, extra_ln);
  });
}

extern "C" int mst_proj_ffn_ln_fwd(const mst_gemm_args* proj, const mst_ln_args* ln1, const mst_gemm_args* ff1, const mst_gemm_args* ff2,
                                   const mst_ln_args* ln2, mst_stream_t stream) {
  MST_CHECK_ARG(proj != nullptr && ln1 != nullptr, "mst_proj_ffn_ln_fwd: null args");
  return ffn_ln_impl("mst_proj_ffn_ln_fwd", ff1, ff2, ln2, 1, stream, nullptr, proj, ln1);
}

extern "C" int mst_ffn_ln_bwd_lead(const mst_ln_bwd_in* lead, const mst_gemm_args* ff2_dgrad, const mst_gemm_args* ff1_dgrad,
                                   const mst_ln_args* ln, mst_stream_t stream) {
  MST_CHECK_ARG(lead != nullptr, "mst_ffn_ln_bwd_lead: null args");
  return ffn_ln_impl("mst_ffn_ln_bwd_lead", ff2_dgrad, ff1_dgrad, ln, 2, stream, lead);
}

extern "C" int mst_ffn_ln_fwd(const mst_gemm_args* ff1, const mst_gemm_args* ff2, const mst_ln_args* ln, mst_stream_t stream) {
  return ffn_ln_impl("mst_ffn_ln_fwd", ff1, ff2, ln, 1, stream);
}
extern "C" int mst_ffn_ln_bwd(const mst_gemm_args* ff2_dgrad, const mst_gemm_args* ff1_dgrad, const mst_ln_args* ln, mst_stream_t stream) {
  return ffn_ln_impl("mst_ffn_ln_bwd", ff2_dgrad, ff1_dgrad, ln, 2, stream);
}

static int check_gemm_ln(const mst_gemm_args& a, const mst_ln_args& l) {
  int rc = check_gemm_common(a);
  if (rc) return rc;
  MST_CHECK_ARG(a.N == 256 || a.N == 128, "mst_gemm_nt_ln: the row width N must be 128 or 256 (got %lld): use mst_gemm_nt + "
                "mst_layernorm_* for other widths", (long long)a.N);
  MST_CHECK_ARG(l.mode == 1 || l.mode == 2, "mst_gemm_nt_ln: mode must be 1 (forward) or 2 (backward)");
  MST_CHECK_ARG(!a.c_f32 && !a.gate && !a.rowadd && !a.grpadd && a.act == MST_ACT_NONE,
                "mst_gemm_nt_ln: fp32 output, gate, rowadd, grpadd and activations are not supported in the fused form");
  MST_CHECK_ARG(a.ldc % 8 == 0 && a.ldc >= a.N, "mst_gemm_nt_ln: ldc must be a multiple of 8 and >= N");
  MST_CHECK_ARG(!a.resid || (a.ldr % 8 == 0 && a.ldr >= a.N && (uintptr_t)a.resid % 16 == 0), "mst_gemm_nt_ln: bad residual layout");
  MST_CHECK_ARG(l.gamma && l.mean && l.rstd, "mst_gemm_nt_ln: gamma / mean / rstd are required");
  if (l.mode == 1) {
    MST_CHECK_ARG(l.beta && l.out && l.ld_out % 8 == 0 && l.ld_out >= a.N && (uintptr_t)l.out % 16 == 0, "mst_gemm_nt_ln: forward needs beta and out");
  } else {
    MST_CHECK_ARG(l.x && l.ld_x % 8 == 0 && (uintptr_t)l.x % 16 == 0 && (l.partials || (l.dgamma && l.dbeta)),
                  "mst_gemm_nt_ln: backward needs x and dgamma + dbeta (or partials)");
    MST_CHECK_ARG((uintptr_t)l.partials % 16 == 0, "mst_gemm_nt_ln: partials must be 16-byte aligned");
    MST_CHECK_ARG(l.mask_mode >= 0 && l.mask_mode <= 2, "mst_gemm_nt_ln: mask_mode must be 0, 1 or 2");
    MST_CHECK_ARG(l.mask_mode != 1 || (l.out && l.ld_out % 8 == 0 && l.ld_out >= a.N && (uintptr_t)l.out % 16 == 0),
                  "mst_gemm_nt_ln: mask_mode 1 needs out");
    MST_CHECK_ARG(!a.self_resid, "mst_gemm_nt_ln: self_resid belongs to the forward form");
  }
  return MST_OK;
}

extern "C" int mst_gemm_nt_ln(const mst_gemm_args* args, const mst_ln_args* ln, mst_stream_t stream) {
  MST_CHECK_ARG(args != nullptr && ln != nullptr, "mst_gemm_nt_ln: null args");
  const mst_gemm_args& a = *args;
  const mst_ln_args& l = *ln;
  int rc = check_gemm_ln(a, l);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  return dispatch_act(a.dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    // 8 waves on a 64-row x full-width tile (32-row tiles, two or three workgroups per CU, measured 8-25 % slower)
    if (a.N == 256) return launch_gemm_ln<T, 64, 256, 2, 4>(a, l, s);
    return launch_gemm_ln<T, 64, 128, 2, 4>(a, l, s);
  });
}

static int check_gemm_bce(const mst_gemm_args& a, const mst_bce_args& q) {
  MST_CHECK_ARG(a.M > 0 && a.K > 0 && a.K % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.A && a.B,
                "mst_gemm_sigmoid_bce: bad GEMM operands");
  MST_CHECK_ARG(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.B % 16 == 0), "mst_gemm_sigmoid_bce: operands must be 16-byte aligned");
  MST_CHECK_ARG(a.N == 128 || (a.N > 0 && a.N % 256 == 0), "mst_gemm_sigmoid_bce: the row of pitches must be 128 or a multiple of 256 wide (got %lld): use "
                "mst_gemm_nt + mst_sigmoid_bce for other widths", (long long)a.N);
  MST_CHECK_ARG(a.N <= 256 || !q.downweight, "mst_gemm_sigmoid_bce: the label down-weighting counts a sample's positives in every workgroup — rows wider than "
                "one tile (256) take mst_gemm_nt + mst_sigmoid_bce");
  MST_CHECK_ARG(q.T > 0 && q.T % 64 == 0 && a.M % q.T == 0, "mst_gemm_sigmoid_bce: T must be a multiple of 64 and divide M (a tile holds one sample's rows)");
  MST_CHECK_ARG(!a.c_f32 && !a.resid && !a.gate && !a.rowadd && !a.grpadd && a.act == MST_ACT_NONE && a.dropout_p == 0.f && !a.self_resid &&
                a.c_rows_per_group <= 0 && !a.a_u8, "mst_gemm_sigmoid_bce: only bias, alpha and an A row remap are supported");
  MST_CHECK_ARG(q.labels && q.loss && (uintptr_t)q.labels % 8 == 0, "mst_gemm_sigmoid_bce: labels / loss missing or labels not 8-byte aligned");
  MST_CHECK_ARG(!a.C || (a.ldc % 8 == 0 && a.ldc >= a.N && (uintptr_t)a.C % 16 == 0), "mst_gemm_sigmoid_bce: bad dlogits layout");
  MST_CHECK_ARG(!q.probs || (q.ldp % 8 == 0 && q.ldp >= a.N && (uintptr_t)q.probs % 16 == 0), "mst_gemm_sigmoid_bce: bad probs layout");
  MST_CHECK_ARG(!q.logits || (q.ldl % 8 == 0 && q.ldl >= a.N && (uintptr_t)q.logits % 16 == 0), "mst_gemm_sigmoid_bce: bad logits layout");
  return MST_OK;
}

extern "C" int mst_gemm_sigmoid_bce(const mst_gemm_args* args, const mst_bce_args* bce, mst_stream_t stream) {
  MST_CHECK_ARG(args != nullptr && bce != nullptr, "mst_gemm_sigmoid_bce: null args");
  const mst_gemm_args& a = *args;
  const mst_bce_args& q = *bce;
  int rc = check_gemm_bce(a, q);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  return dispatch_act(a.dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    if (a.N % 256 == 0) return launch_gemm_bce<T, 256>(a, q, s);
    return launch_gemm_bce<T, 128>(a, q, s);
  });
}

extern "C" int mst_gemm_sigmoid_bce_dgrad_ln(const mst_gemm_args* args, const mst_bce_args* bce, const mst_gemm_args* dgrad,
                                             const mst_ln_args* ln, mst_stream_t stream) {
  MST_CHECK_ARG(args != nullptr && bce != nullptr && dgrad != nullptr && ln != nullptr, "mst_gemm_sigmoid_bce_dgrad_ln: null args");
  const mst_gemm_args &a = *args, &g2 = *dgrad;
  const mst_bce_args& q = *bce;
  const mst_ln_args& l = *ln;
  int rc = check_gemm_bce(a, q);
  if (rc == MST_OK) rc = check_gemm_ln(g2, l);
  if (rc) return rc;
  // one launch: 128 pitches, width 128, whole 64-row tiles, and the second GEMM's A operand IS the first one's logit gradient
  static const bool off = getenv("MST_BCE_DGRAD") && getenv("MST_BCE_DGRAD")[0] == '0';
  const bool one = !off && a.N == 128 && g2.N == 128 && g2.K == 128 && g2.M == a.M && a.M % 64 == 0 && l.mode == 2 && a.C && g2.A == a.C &&
                   g2.lda == a.ldc && g2.dtype == a.dtype && g2.a_rows_per_group <= 0 && !g2.a_u8;
  if (!one) {
    rc = mst_gemm_sigmoid_bce(args, bce, stream);
    return rc != MST_OK ? rc : mst_gemm_nt_ln(dgrad, ln, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  return dispatch_act(a.dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    return launch_gemm_bce_dgrad_ln<T>(a, q, g2, l, s);
  });
}

extern "C" int mst_gemm_nt(const mst_gemm_args* args, mst_stream_t stream) {
  MST_CHECK_ARG(args != nullptr, "mst_gemm_nt: null args");
  const mst_gemm_args& a = *args;
  MST_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "mst_gemm_nt: M,N,K must be positive (got %lld,%lld,%lld)",
                (long long)a.M, (long long)a.N, (long long)a.K);
  MST_CHECK_ARG(a.K % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0,
                "mst_gemm_nt: K, lda, ldb must be multiples of 8 (got %lld,%lld,%lld)", (long long)a.K,
                (long long)a.lda, (long long)a.ldb);
  MST_CHECK_ARG(a.ldc % 4 == 0 && a.ldc >= a.N, "mst_gemm_nt: ldc must be a multiple of 4 and >= N");
  MST_CHECK_ARG(a.A && a.B && a.C, "mst_gemm_nt: null operand");
  MST_CHECK_ARG(!a.resid || (a.ldr % 4 == 0 && a.ldr >= a.N), "mst_gemm_nt: ldr must be a multiple of 4 and >= N");
  MST_CHECK_ARG(!a.gate || (a.ldg % 4 == 0 && a.ldg >= a.N), "mst_gemm_nt: ldg must be a multiple of 4 and >= N");
  MST_CHECK_ARG((!a.rowadd && !a.grpadd) || a.rowadd_period > 0, "mst_gemm_nt: rowadd_period must be > 0");
  MST_CHECK_ARG(!a.grpadd || a.grp_index, "mst_gemm_nt: grpadd needs grp_index");
  MST_CHECK_ARG(a.dropout_p >= 0.f && a.dropout_p < 1.f, "mst_gemm_nt: dropout_p must be in [0,1)");
  MST_CHECK_ARG(a.dropout_p == 0.f || a.N % 4 == 0, "mst_gemm_nt: dropout needs N to be a multiple of 4");
  MST_CHECK_ARG(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.B % 16 == 0) && ((uintptr_t)a.C % 16 == 0),
                "mst_gemm_nt: operands must be 16-byte aligned");
  MST_CHECK_ARG(!a.a_u8 || (!a.c_f32 && a.dropout_p == 0.f && !a.self_resid),
                "mst_gemm_nt: a uint8 A operand comes with a 16-bit C and without dropout / self_resid");
  hipStream_t s = (hipStream_t)stream;
  return dispatch_act(a.dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    const int64_t big_tiles = cdiv(a.M, 128) * cdiv(a.N, 128);
    // Tile shape is second-order here: 64x64, 128x64, 64x128 and 128x128 tiles measured within 5 % of each other on
    // every GEMM of the step (time = 4.7 us fixed + 1.7 us per 8.4 MB of output + 4.2 us per 2.1 GFLOP: with K <= 1024
    // a tile's main loop is 2-16 stages of one exposed L2 round trip each, at 12 TB/s of L2->LDS traffic for the
    // K = 1024 shapes). 128x128 is used where it still leaves >= 1.5 workgroups per CU.
    // Two 128x128 workgroups fit a CU (LDS), 512 on the chip: a launch of 516 (the decoder's M = 64 x 257 rows: 129 row
    // tiles x 4) runs a second resident round for four workgroups — 21 us against 13 us with 64x64 tiles.
    const int64_t last_round = big_tiles % 512;
    const bool stub_round = big_tiles > 512 && last_round > 0 && last_round < 128;
    // (a launch whose rows are whole 64-row tiles but not whole 128-row tiles — the decoder's 64 x 257 — keeps the
    // fast-epilogue kernel with 64x64 tiles)
    const bool ragged128 = a.M % 128 != 0 && a.M % 64 == 0 && a.N % 128 == 0 && !a.c_f32;
    static const bool three = !(getenv("MST_GEMM_3CU") && getenv("MST_GEMM_3CU")[0] == '0');
    if (three && big_tiles > 512 && big_tiles <= 768 && a.N >= 128 && !ragged128) {
      const int rc = launch_gemm_3cu<T>(a, s);
      if (rc <= 0) return rc;
    }
    if (big_tiles >= 384 && a.N >= 128 && !stub_round && !ragged128) return launch_gemm<T, 128, 128, 2, 2>(a, s);
    if (a.M <= 64 && a.K >= 512 && a.K % 256 == 0) return launch_gemm<T, 64, 64, 2, 2, 256>(a, s);
    // (32-deep K stages for launches of 1281..2048 64 x 64 tiles — eight workgroups per CU, one resident round for the decoder's
    // 257 x 6 projection tiles — measured no faster: +2 us per step)
    return launch_gemm<T, 64, 64, 2, 2>(a, s);
  });
}

extern "C" int mst_gemm_nt_pair(const mst_gemm_args* args0, const mst_gemm_args* args1, mst_stream_t stream) {
  return mst_gemm_nt_pair_begin(args0, args1, nullptr, stream);
}

extern "C" int mst_gemm_nt_pair_begin(const mst_gemm_args* args0, const mst_gemm_args* args1, const mst_step_begin_args* begin,
                                      mst_stream_t stream) {
  MST_CHECK_ARG(args0 != nullptr && args1 != nullptr, "mst_gemm_nt_pair: null args");
  const mst_gemm_args &a0 = *args0, &a1 = *args1;
  // one launch for two uint8-A problems of the fast row-op form (what the embedding GEMMs are); anything else is two launches
  // of mst_gemm_nt, which also reports what is wrong with an argument
  auto plain = [](const mst_gemm_args& a) {
    return a.a_u8 && !a.c_f32 && a.M > 0 && a.N > 0 && a.K > 0 && a.K % 8 == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 && a.ldc >= a.N &&
           a.A && a.B && a.C && (uintptr_t)a.A % 16 == 0 && (uintptr_t)a.B % 16 == 0 && (uintptr_t)a.C % 16 == 0 && !a.resid && !a.gate &&
           a.act == 0 && a.dropout_p == 0.f && !a.self_resid && ((!a.rowadd && !a.grpadd) || a.rowadd_period > 0) &&
           (!a.grpadd || a.grp_index) && gemm_fast_form<64, 64>(a) && a.K % 64 == 0 && cdiv(a.M, 64) * cdiv(a.N, 64) < (1 << 20);
  };
  static const bool off = getenv("MST_GEMM_PAIR") && getenv("MST_GEMM_PAIR")[0] == '0';
  if (off || a0.dtype != a1.dtype || !plain(a0) || !plain(a1) || (begin && begin->sh_w && begin->sh_dtype != a0.dtype)) {
    int rc = begin ? mst_step_begin_v(begin, stream) : MST_OK;  // (runs the shadow refresh as a launch of its own)
    if (rc == MST_OK) rc = mst_gemm_nt(args0, stream);
    return rc != MST_OK ? rc : mst_gemm_nt(args1, stream);
  }
  StepBegin sb = {};
  int n_begin = 0;
  if (begin) {
    int64_t grid = 0;
    const int rc = pack_step_begin(*begin, sb, &grid);
    if (rc) return rc;
    n_begin = (int)((grid + 7) / 8 * 8);
  }
  hipStream_t s = (hipStream_t)stream;
  return dispatch_act(a0.dtype, [&](auto tag) -> int {
    typedef decltype(tag) T;
    const int64_t sh_tiles = sb.sh_w ? sb.sh_tiles : 0;
    // long contractions (configs[2]: 2048 pitch columns per frame): 128 x 128 tiles — the uint8 frames cross L2 -> LDS once per
    // 128 output columns instead of once per 64 (three times instead of six for the two tables), 33 K stages amortise the tile's
    // prologue and epilogue; at configs[1]'s K = 128 the 64 x 64 form stays (1 536 short tiles fill the chip, 384 long ones do not)
    static const bool big_off = getenv("MST_GEMM_PAIR_BIG") && getenv("MST_GEMM_PAIR_BIG")[0] == '0';
    if (!big_off && a0.K >= 1024 && a1.K >= 1024 && gemm_fast_form<128, 128>(a0) && gemm_fast_form<128, 128>(a1)) {
      const int t0 = (int)((a0.M / 128) * (a0.N / 128)), t1 = (int)((a1.M / 128) * (a1.N / 128));
      const size_t lds_b = (size_t)2 * (128 + 128) * 64 * 2;  // (epilogue: one 64-row block of the tile at a time, PATH 4: 34 KB)
      static bool opted = false;
      if (!opted) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pair_kernel<T, 128, 128, 2, 2, 64, 4>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        if (e != hipSuccess) { set_error("gemm_nt_pair_kernel: LDS opt-in of %zu bytes: %s", lds_b, hipGetErrorString(e)); return MST_ERR_LAUNCH; }
        opted = true;
      }
      hipLaunchKernelGGL((gemm_nt_pair_kernel<T, 128, 128, 2, 2, 64, 4>), dim3((unsigned)(n_begin + t0 + t1 + sh_tiles)), dim3(256), lds_b, s, a0, a1,
                         t0, t1, sb, n_begin);
      MST_CHECK_LAUNCH("gemm_nt_pair_kernel");
      return MST_OK;
    }
    const int tiles0 = (int)((a0.M / 64) * (a0.N / 64)), tiles1 = (int)((a1.M / 64) * (a1.N / 64));
    const size_t lds = (size_t)2 * (64 + 64) * 64 * 2;  // K-loop stages; the 64 x 68 fp32 epilogue staging is smaller
    hipLaunchKernelGGL((gemm_nt_pair_kernel<T, 64, 64, 2, 2, 64, 1>), dim3((unsigned)(n_begin + tiles0 + tiles1 + sh_tiles)), dim3(256), lds, s, a0,
                       a1, tiles0, tiles1, sb, n_begin);
    MST_CHECK_LAUNCH("gemm_nt_pair_kernel");
    return MST_OK;
  });
}

#ifdef MST_FFN_STAMPS
extern "C" int mst_debug_ffn_stamps(uint64_t* host_out) {  // diagnostic builds only
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mst::g_ffn_stamps), sizeof(uint64_t) * (8 + 48 * 4)) == hipSuccess ? 0 : -1;
}
#endif

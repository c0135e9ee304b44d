// embed.hip — token-path input ends and padding masks.
//
//   mst_embed_fwd : out[b, s_off+t, :] = alpha*(table[tok[b,t]] + cls[c_b]) + pos[s_off+t]
//                   (gluon Embedding + broadcast_add + sqrt(D)*x + pos: model.py:86-91,241-245,
//                   transformer.py:237,270) and keymask = (tok != 0) (model.py:81-83)
//   mst_embed_bwd : scatter-add of alpha*dX into the table gradient, per-sample column sums into the
//                   class-embedding gradient (autograd of the above)
//   mst_group_colsum : dst[idx[b], :] += alpha * sum_t X[b, s_off+t, :]   (class-embedding gradient of
//                   the piano-roll input GEMM, whose epilogue added cls[c_b] to every frame)
//   mst_mask_from_lengths : SequenceMask(ones, seq_len + add) (model.py:246-247)
//
// The piano-roll path does not come through here: a multi-hot frame times the table is a Dense
// GEMM (gemm_nt.hip with the table's transposed shadow) whose epilogue does the same adds.
#include "common.hpp"

namespace mst {

template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(int64_t BT, int64_t T_len, int D, const int32_t* __restrict__ tokens,
                                                        const float* __restrict__ table, int64_t ldt,
                                                        const int32_t* __restrict__ classes,
                                                        const float* __restrict__ cls_table, int64_t ldc,
                                                        const float* __restrict__ pos, int64_t ldp, float alpha,
                                                        T* __restrict__ out, int64_t ld_out, int64_t S_out,
                                                        int64_t s_off, uint8_t* __restrict__ keymask) {
  // one wave per (b,t) row, lanes over D in float4 chunks
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave_global; r < BT; r += nwaves) {
    const int64_t b = r / T_len, t = r % T_len;
    const int tok = tokens[r];
    const float* trow = table + (int64_t)tok * ldt;
    const float* crow = cls_table ? cls_table + (int64_t)classes[b] * ldc : nullptr;
    const float* prow = pos + (s_off + t) * ldp;
    T* orow = out + (b * S_out + s_off + t) * ld_out;
    for (int d = lane * 4; d < D; d += 256) {
      f32x4 e = *reinterpret_cast<const f32x4*>(trow + d);
      f32x4 p = *reinterpret_cast<const f32x4*>(prow + d);
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      if (crow) c = *reinterpret_cast<const f32x4*>(crow + d);
      u32x2 o;
      o[0] = (uint32_t)f32_to_bits<T>(alpha * (e[0] + c[0]) + p[0]) | ((uint32_t)f32_to_bits<T>(alpha * (e[1] + c[1]) + p[1]) << 16);
      o[1] = (uint32_t)f32_to_bits<T>(alpha * (e[2] + c[2]) + p[2]) | ((uint32_t)f32_to_bits<T>(alpha * (e[3] + c[3]) + p[3]) << 16);
      *reinterpret_cast<u32x2*>(orow + d) = o;
    }
    if (keymask && lane == 0) keymask[b * S_out + s_off + t] = (tok != 0) ? 1 : 0;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_kernel(int64_t BT, int64_t T_len, int D, const int32_t* __restrict__ tokens,
                                                            float* __restrict__ dtable, int64_t ldt, float alpha,
                                                            const T* __restrict__ dX, int64_t ld_dx, int64_t S_out,
                                                            int64_t s_off) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r = wave_global; r < BT; r += nwaves) {
    const int64_t b = r / T_len, t = r % T_len;
    const int tok = tokens[r];
    const T* grow = dX + (b * S_out + s_off + t) * ld_dx;
    float* drow = dtable + (int64_t)tok * ldt;
    for (int d = lane; d < D; d += 64) atomicAdd(drow + d, alpha * to_f32(grow[d]));
  }
}

// dst[idx[b], d] += alpha * sum_{t<T} X[b, s_off + t, d].
// grid = (chunks of GC_ROWS frames, B). A thread owns 8 consecutive columns (one 16-byte load per row) of every RG-th
// row of the chunk; the RG row groups are combined through LDS and each column ends in ONE atomic per workgroup.
// (One 2-byte load per lane and an atomic per 32 frames measured 15 us for 8.4 MB.)
constexpr int GC_ROWS = 64;
template <typename T>
__global__ __launch_bounds__(256) void group_colsum_kernel(int64_t T_len, int D, const T* __restrict__ X, int64_t ldx,
                                                           int64_t S_out, int64_t s_off,
                                                           const int32_t* __restrict__ idx, float* __restrict__ dst,
                                                           int64_t ldd, float alpha) {
  extern __shared__ float gc_part[];  // [RG][D]
  const int cpr = D / 8;              // 16-byte chunks per row (host guarantees D % 8 == 0, cpr <= 256)
  const int RG = 256 / cpr;           // row groups
  const int64_t b = blockIdx.y;
  const int tid = threadIdx.x, ch = tid % cpr, rg = tid / cpr;
  const int64_t t0 = (int64_t)blockIdx.x * GC_ROWS;
  const int64_t t1 = t0 + GC_ROWS < T_len ? t0 + GC_ROWS : T_len;
  const T* base = X + (b * S_out + s_off) * ldx + ch * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (rg < RG) {
#pragma unroll 4
    for (int64_t t = t0 + rg; t < t1; t += RG) {
      Pack8 p;
      p.u = *reinterpret_cast<const u32x4*>(base + t * ldx);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += bits_to_f32<T>(p.h[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) gc_part[rg * D + ch * 8 + e] = acc[e];
  }
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    float v = 0.f;
    for (int r = 0; r < RG; ++r) v += gc_part[r * D + d];
    atomicAdd(dst + (int64_t)idx[b] * ldd + d, alpha * v);
  }
}

__global__ __launch_bounds__(256) void mask_from_lengths_kernel(int64_t B, int64_t S, const int32_t* __restrict__ lens,
                                                                int32_t add, uint8_t* __restrict__ keymask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < B * S) keymask[i] = ((i % S) < (int64_t)lens[i / S] + add) ? 1 : 0;
}

}  // namespace mst

using namespace mst;

extern "C" int mst_embed_fwd(int dtype, int64_t B, int64_t T, int64_t D, const int32_t* tokens, const float* table,
                             int64_t ldt, const int32_t* classes, const float* cls_table, int64_t ldc,
                             const float* pos, int64_t ldp, float alpha, void* out, int64_t ld_out, int64_t S_out,
                             int64_t s_off, uint8_t* keymask, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && T > 0 && D > 0 && D % 4 == 0, "mst_embed_fwd: B,T,D must be positive and D a multiple of 4");
  MST_CHECK_ARG(tokens && table && pos && out, "mst_embed_fwd: null pointer");
  MST_CHECK_ARG(!cls_table || classes, "mst_embed_fwd: cls_table needs classes");
  MST_CHECK_ARG(ldt % 4 == 0 && ldp % 4 == 0 && ld_out % 4 == 0 && (!cls_table || ldc % 4 == 0), "mst_embed_fwd: leading dims must be multiples of 4");
  MST_CHECK_ARG(s_off >= 0 && s_off + T <= S_out, "mst_embed_fwd: rows do not fit in S_out");
  const int64_t BT = B * T;
  const unsigned grid = (unsigned)(cdiv(BT, 4) < 4096 ? cdiv(BT, 4) : 4096);
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) TT;
    hipLaunchKernelGGL((embed_fwd_kernel<TT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, BT, T, (int)D, tokens, table,
                       ldt, classes, cls_table, ldc, pos, ldp, alpha, (TT*)out, ld_out, S_out, s_off, keymask);
    MST_CHECK_LAUNCH("embed_fwd_kernel");
    return MST_OK;
  });
}

extern "C" int mst_group_colsum(int dtype, int64_t B, int64_t T, int64_t D, const void* X, int64_t ldx, int64_t S_out,
                                int64_t s_off, const int32_t* idx, float* dst, int64_t ldd, float alpha,
                                mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && T > 0 && D > 0 && X && idx && dst, "mst_group_colsum: bad argument");
  MST_CHECK_ARG(B <= 65535, "mst_group_colsum: B too large for grid.y");
  MST_CHECK_ARG(D % 8 == 0 && D <= 2048 && ldx % 8 == 0 && ((uintptr_t)X % 16 == 0),
                "mst_group_colsum: D must be a multiple of 8 (<= 2048), rows 16-byte aligned");
  const size_t lds = sizeof(float) * (size_t)(256 / (D / 8)) * (size_t)D;
  return dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) TT;
    hipLaunchKernelGGL((group_colsum_kernel<TT>), dim3((unsigned)cdiv(T, GC_ROWS), (unsigned)B), dim3(256), lds, (hipStream_t)stream, T,
                       (int)D, (const TT*)X, ldx, S_out, s_off, idx, dst, ldd, alpha);
    MST_CHECK_LAUNCH("group_colsum_kernel");
    return MST_OK;
  });
}

extern "C" int mst_embed_bwd(int dtype, int64_t B, int64_t T, int64_t D, const int32_t* tokens, float* dtable,
                             int64_t ldt, const int32_t* classes, float* dcls, int64_t ldc, float alpha,
                             const void* dX, int64_t ld_dx, int64_t S_out, int64_t s_off, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && T > 0 && D > 0, "mst_embed_bwd: B,T,D must be positive");
  MST_CHECK_ARG(tokens && dtable && dX, "mst_embed_bwd: null pointer");
  MST_CHECK_ARG(!dcls || classes, "mst_embed_bwd: dcls needs classes");
  const int64_t BT = B * T;
  const unsigned grid = (unsigned)(cdiv(BT, 4) < 4096 ? cdiv(BT, 4) : 4096);
  int rc = dispatch_act(dtype, [&](auto tag) -> int {
    typedef decltype(tag) TT;
    hipLaunchKernelGGL((embed_scatter_kernel<TT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, BT, T, (int)D, tokens,
                       dtable, ldt, alpha, (const TT*)dX, ld_dx, S_out, s_off);
    MST_CHECK_LAUNCH("embed_scatter_kernel");
    return MST_OK;
  });
  if (rc) return rc;
  if (dcls) return mst_group_colsum(dtype, B, T, D, dX, ld_dx, S_out, s_off, classes, dcls, ldc, alpha, stream);
  return MST_OK;
}

extern "C" int mst_mask_from_lengths(int64_t B, int64_t S, const int32_t* lens, int32_t add, uint8_t* keymask,
                                     mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && S > 0 && lens && keymask, "mst_mask_from_lengths: bad argument");
  hipLaunchKernelGGL(mask_from_lengths_kernel, dim3((unsigned)cdiv(B * S, 256)), dim3(256), 0, (hipStream_t)stream, B, S, lens,
                     add, keymask);
  MST_CHECK_LAUNCH("mask_from_lengths_kernel");
  return MST_OK;
}

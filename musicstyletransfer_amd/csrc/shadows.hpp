// shadows.hpp — one 32 x 32 tile of the transposed 16-bit shadow refresh (mst_transpose_shadows): for matrix i, src fp32
// [rows, cols] at w + desc[4i], dst [cols, ld_t] at wt16 + desc[4i+1] with ld_t = roundup8(rows); pad columns rows..ld_t zeroed.
// A launch of its own (optim.hip) or extra workgroups behind the tiles of the step's first GEMM launch (gemm_nt.hip:
// mst_gemm_nt_pair_begin with mst_step_begin_args.sh_*).
#pragma once
#include "common.hpp"

namespace mst {

// 256 threads (t: the thread's index among them — a wider workgroup runs one tile per 256 threads, every thread reaching the barrier;
// live: this group has a tile); tile: 32 x 33 floats of LDS; tb: tile index within the list
template <typename T>
__device__ __forceinline__ void shadow_tile_wg(const float* __restrict__ w, T* __restrict__ wt16, const int64_t* __restrict__ desc,
                                               const int64_t* __restrict__ tile_prefix, int n_mat, int64_t tb, float (*tile)[33],
                                               bool live = true, int t = (int)threadIdx.x) {
  if (!live) { __syncthreads(); return; }
  int mi = 0;
  for (int i = 1; i < n_mat; ++i)
    if (tb >= tile_prefix[i]) mi = i;
  const int64_t src_off = desc[4 * mi], dst_off = desc[4 * mi + 1], rows = desc[4 * mi + 2], cols = desc[4 * mi + 3];
  const int64_t ld_t = (rows + 7) / 8 * 8;
  const int64_t local = tb - tile_prefix[mi];
  const int64_t tiles_c = (cols + 31) / 32;
  const int64_t r0 = (local / tiles_c) * 32, c0 = (local % tiles_c) * 32;
  const int tx = t & 31, ty = t >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int64_t r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? w[src_off + r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int64_t c = c0 + j, r = r0 + tx;  // dst row = c, dst col = r
    if (c < cols && r < ld_t) wt16[dst_off + c * ld_t + r] = from_f32<T>(tile[tx][j]);
  }
}

}  // namespace mst

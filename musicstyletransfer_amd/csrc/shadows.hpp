// shadows.hpp — one 32 x 32 tile of the transposed 16-bit shadow refresh (mst_transpose_shadows): for matrix i, src fp32
// [rows, cols] at w + desc[4i], dst [cols, ld_t] at wt16 + desc[4i+1] with ld_t = roundup8(rows); pad columns rows..ld_t zeroed.
// A launch of its own (optim.hip) or extra workgroups behind the tiles of the step's first GEMM launch (gemm_nt.hip:
// mst_gemm_nt_pair_begin with mst_step_begin_args.sh_*).
#pragma once
#include "common.hpp"

namespace mst {

// which matrix of the list tile `tb` belongs to (tb uniform over the wave, every lane active): ONE parallel load of the prefix table and a
// ballot — the loop `if (tb >= tile_prefix[i]) mi = i` is n_mat DEPENDENT round trips to L2 (~0.3 us each: 6 us per tile at 20 matrices,
// most of what a tile cost)
__device__ __forceinline__ int shadow_mat_of(const int64_t* __restrict__ tile_prefix, int n_mat, int64_t tb) {
  if (n_mat <= 64) {
    const int lane = (int)(threadIdx.x & 63);
    const int64_t p = lane < n_mat ? tile_prefix[lane] : (int64_t)1 << 62;
    return __popcll(__ballot(tb >= p)) - 1;  // (tile_prefix[0] == 0)
  }
  int mi = 0;
  for (int i = 1; i < n_mat; ++i)
    if (tb >= tile_prefix[i]) mi = i;
  return mi;
}

// 256 threads (t: the thread's index among them — a wider workgroup runs one tile per 256 threads, every thread reaching the barrier;
// live: this group has a tile); tile: 32 x 33 floats of LDS; tb: tile index within the list
template <typename T>
__device__ __forceinline__ void shadow_tile_wg(const float* __restrict__ w, T* __restrict__ wt16, const int64_t* __restrict__ desc,
                                               const int64_t* __restrict__ tile_prefix, int n_mat, int64_t tb, float (*tile)[33],
                                               bool live = true, int t = (int)threadIdx.x) {
  if (!live) { __syncthreads(); return; }
  const int mi = shadow_mat_of(tile_prefix, n_mat, tb);
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

// FOUR consecutive tiles (tb0 .. tb0 + 3, those below n_total) by one group of 256 threads, every load of the four requested before
// the first is used: a rider workgroup pays one memory round trip per 16 tiles instead of one per 4 (row_tail.hip). tiles: 4 x 32 x 33
// floats of LDS; every thread of the workgroup reaches the one barrier.
template <typename T>
__device__ __forceinline__ void shadow_tile_quad(const float* __restrict__ w, T* __restrict__ wt16, const int64_t* __restrict__ desc,
                                                 const int64_t* __restrict__ tile_prefix, int n_mat, int64_t tb0, int64_t n_total,
                                                 float (*tiles)[32][33], int t) {
  const int tx = t & 31, ty = t >> 5;  // 32 x 8
  int64_t src[4], dst[4], rows[4], cols[4], r0[4], c0[4];
  float v[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t tb = tb0 + k;
    rows[k] = 0; cols[k] = 0; src[k] = 0; dst[k] = 0; r0[k] = 0; c0[k] = 0;
    if (tb < n_total) {  // (uniform over the wave)
      const int mi = shadow_mat_of(tile_prefix, n_mat, tb);
      src[k] = desc[4 * mi]; dst[k] = desc[4 * mi + 1]; rows[k] = desc[4 * mi + 2]; cols[k] = desc[4 * mi + 3];
      const int64_t local = tb - tile_prefix[mi], tiles_c = (cols[k] + 31) / 32;
      r0[k] = (local / tiles_c) * 32; c0[k] = (local % tiles_c) * 32;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int64_t r = r0[k] + ty + 8 * jj, c = c0[k] + tx;
      v[k][jj] = (r < rows[k] && c < cols[k]) ? w[src[k] + r * cols[k] + c] : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) tiles[k][ty + 8 * jj][tx] = v[k][jj];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t ld_t = (rows[k] + 7) / 8 * 8;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = ty + 8 * jj;
      const int64_t c = c0[k] + j, r = r0[k] + tx;  // dst row = c, dst col = r
      if (c < cols[k] && r < ld_t) wt16[dst[k] + c * ld_t + r] = from_f32<T>(tiles[k][tx][j]);
    }
  }
}

}  // namespace mst

// beam.hip — one position of beam search on the device (reference VarAutoEncoder/sampler.py:198-257, token ends):
//   mst_beam_step   : per sample, the K best of its K x V continuations (score = summed -log p; a finished hypothesis continues
//                     with PAD only, at no cost: sampler.py:218-221), the re-ranked token rows, the words fed to the next position
//   mst_beam_gather : the decoder layers' K | Q | V cache rows of the re-ranked hypotheses (sampler.py:236-238)
// With these two the whole position — decoder step, ranking, cache reorder — is device work inside ONE captured graph; the host
// version spent 98 % of a position in its own top-k over beam x V and the copies around it (2.6 ms against 55 us of device time).
// Ranking is deterministic: candidates are ordered by (score, hypothesis * V + word), i.e. the stable argsort of the host form.
#include <math.h>
#include "common.hpp"

namespace mst {

struct BeamArgs {
  int64_t B, K, V, i, L;
  const float* probs; int64_t ldp;
  const float* scores_in; float* scores_out;
  const int32_t* seqs_in; int32_t* seqs_out;
  int32_t* hyp_src; int32_t* word; int32_t* active;
  int32_t eos, pad;
};

constexpr int BEAM_MAXK = 16;

// one 256-thread workgroup per sample
__global__ __launch_bounds__(256) void beam_step_kernel(BeamArgs q) {
  __shared__ float s_score[BEAM_MAXK];
  __shared__ int s_fin[BEAM_MAXK];
  __shared__ float s_val[BEAM_MAXK];
  __shared__ int s_idx[BEAM_MAXK];
  __shared__ float r_val[4];
  __shared__ int r_idx[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const int K = (int)q.K, V = (int)q.V;
  if (tid < K) {
    const int64_t h = b * K + tid;
    s_score[tid] = q.scores_in[h];
    const int32_t last = q.seqs_in[h * q.L + q.i - 1];
    s_fin[tid] = (last == q.eos) || (last == q.pad && q.i > 1);
  }
  __syncthreads();
  const int n_cand = K * V;
  // every candidate's score ONCE (the K selection rounds below used to recompute the logarithm and the c / V division of all
  // K V candidates each: 15.9 us per position); a thread keeps its candidates c = tid + 256 j in registers when they fit
  constexpr int OWN = 8;  // K V <= 2048 candidates: beam 4 x 293 tokens = 5 per thread
  const bool in_regs = n_cand <= OWN * 256;
  float own[OWN];
  uint32_t done = 0u;  // this thread's candidates already selected
  auto score = [&](int c) {
    const int k = c / V, w = c - k * V;
    if (s_fin[k]) return (w == q.pad) ? s_score[k] : INFINITY;
    const float v = s_score[k] - logf(fmaxf(q.probs[(b * K + k) * q.ldp + w], 1e-30f));
    return v == v ? v : INFINITY;  // a NaN (from a NaN score or probability) never wins and never leaves the selection without a winner
  };
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < OWN; ++j) {
      const int c = tid + 256 * j;
      own[j] = c < n_cand ? score(c) : INFINITY;
    }
  }
  for (int r = 0; r < K; ++r) {
    float best = INFINITY;
    int best_i = 0x7fffffff;
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < OWN; ++j) {  // (the general loop's rule, ties included: an all-infinite remainder still yields a valid index)
        const int c = tid + 256 * j;
        if (c < n_cand && !((done >> j) & 1u) && (own[j] < best || (own[j] == best && c < best_i))) { best = own[j]; best_i = c; }
      }
    } else {
      for (int c = tid; c < n_cand; c += 256) {
        bool taken = false;
        for (int u = 0; u < r; ++u) taken |= (s_idx[u] == c);
        if (taken) continue;
        const float val = score(c);
        if (val < best || (val == best && c < best_i)) { best = val; best_i = c; }
      }
    }
    // workgroup argmin by (value, index)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(best_i, off, 64);
      if (ov < best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
    }
    if (lane == 0) { r_val[wave] = best; r_idx[wave] = best_i; }
    __syncthreads();
    if (tid == 0) {
      float bv = r_val[0]; int bi = r_idx[0];
      for (int wv = 1; wv < 4; ++wv)
        if (r_val[wv] < bv || (r_val[wv] == bv && r_idx[wv] < bi)) { bv = r_val[wv]; bi = r_idx[wv]; }
      // (with NaN mapped to +inf above some candidate always wins, ties included; the clamp keeps the index arithmetic below
      // inside seqs / hyp_src whatever happens)
      s_val[r] = bv; s_idx[r] = ((unsigned)bi < (unsigned)n_cand) ? bi : 0;
    }
    __syncthreads();
    if (in_regs) {  // the winner leaves its owner's registers
      const int won = s_idx[r];
#pragma unroll
      for (int j = 0; j < OWN; ++j)
        if (tid + 256 * j == won) done |= 1u << j;
    }
  }
  // the re-ranked rows: new hypothesis r continues hypothesis s_idx[r] / V with word s_idx[r] % V
  int alive = 0;
  for (int r = 0; r < K; ++r) {
    const int c = s_idx[r];
    const int k = c / V, w = c - k * V;
    const int64_t dst = b * K + r, src = b * K + k;
    for (int64_t col = tid; col < q.i; col += 256) q.seqs_out[dst * q.L + col] = q.seqs_in[src * q.L + col];
    if (tid == 0) {
      q.seqs_out[dst * q.L + q.i] = w;
      q.scores_out[dst] = s_val[r];
      q.hyp_src[dst] = (int32_t)src;
      q.word[dst] = w;
      alive += (w != q.eos && w != q.pad) ? 1 : 0;
    }
  }
  if (tid == 0 && q.active && alive) atomicAdd(q.active + q.i, alive);
}

// out[j, r, :] = in[src[j], r, :] for r < n_rows; rows of row_bytes bytes (a multiple of 16), t_max rows per hypothesis
// (skip_begin, skip_bytes: a byte range of every row that is NOT copied — the Q third of a K | Q | V cache row: a past position's
// query is never read again, attn_decode_kernel takes q from the newest row only)
__global__ __launch_bounds__(256) void beam_gather_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const int32_t* __restrict__ src,
                                                          int64_t N, int64_t n_rows, int64_t row_bytes, int64_t t_max, int64_t skip_begin,
                                                          int64_t skip_bytes) {
  const int64_t pieces = (row_bytes - skip_bytes) / 16, total = N * n_rows * pieces, skip_pc = skip_begin / 16, skip_n = skip_bytes / 16;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
    int64_t pc = p % pieces;
    const int64_t rr = (p / pieces) % n_rows, j = p / (pieces * n_rows);
    if (pc >= skip_pc) pc += skip_n;
    const int64_t s = src[j];
    reinterpret_cast<u32x4*>(out + (j * t_max + rr) * row_bytes)[pc] = reinterpret_cast<const u32x4*>(in + (s * t_max + rr) * row_bytes)[pc];
  }
}

// Ancestral sampling (sampler.py:155-190): one wave per sequence draws the next token from its distribution by inverse CDF —
// u = counter hash of (seed, position, sequence) in (0, 1] times the row sum; lane l owns the contiguous chunk of ceil(V / 64)
// tokens, a wave scan of the chunk sums finds the owning lane, which walks its chunk. A finished sequence (last token EOS, or
// PAD from position 2 on) continues with PAD at no cost. score += -log max(p[token], 1e-30); active[i] counts running sequences.
__global__ __launch_bounds__(256) void sample_step_kernel(int64_t N, int64_t V, int64_t i, int64_t L, const float* __restrict__ probs, int64_t ldp,
                                                          int32_t* __restrict__ seqs, float* __restrict__ scores, int32_t* __restrict__ word,
                                                          int32_t* __restrict__ active, uint64_t seed, int32_t eos, int32_t pad) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int32_t last = seqs[n * L + i - 1];
  const bool fin = (last == eos) || (last == pad && i > 1);
  const float* row = probs + n * ldp;
  const int chunk = (int)((V + 63) / 64);
  const int lo = lane * chunk, hi = (lo + chunk < V) ? lo + chunk : (int)V;
  float part = 0.f;
  for (int w = lo; w < hi; ++w) part += row[w];
  float incl = part;  // inclusive scan over the lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  const float total = __shfl(incl, 63, 64);
  const uint32_t h = dropout_hash(seed, (uint32_t)i, (uint64_t)n);
  const float target = ((float)(h >> 8) + 1.0f) * (1.0f / 16777216.0f) * total;  // in (0, total]
  const bool mine = (incl >= target) && (incl - part < target);
  int tok = -1;
  if (mine) {
    float acc = incl - part;
    tok = hi - 1;
    for (int w = lo; w < hi; ++w) {
      acc += row[w];
      if (acc >= target) { tok = w; break; }
    }
  }
  // exactly one lane owns the draw (rounding at a chunk boundary could leave none: the last token with mass then takes it)
  const unsigned long long owners = __ballot(mine && tok >= 0);
  int chosen = tok;
  if (owners == 0ull) chosen = (int)V - 1;
  const int src_lane = owners ? (int)__builtin_ctzll(owners) : 0;
  chosen = __shfl(chosen, src_lane, 64);
  if (lane == 0) {
    const int32_t w = fin ? pad : chosen;
    seqs[n * L + i] = w;
    word[n] = w;
    if (!fin) scores[n] += -logf(fmaxf(row[chosen] / fmaxf(total, 1e-30f), 1e-30f));
    if (active && w != eos && w != pad) atomicAdd(active + i, 1);
  }
}

}  // namespace mst

using namespace mst;

extern "C" int mst_sample_step(int64_t N, int64_t V, int64_t i, int64_t L, const float* probs, int64_t ldp, int32_t* seqs, float* scores,
                               int32_t* word, int32_t* active, uint64_t seed, int32_t eos, int32_t pad, mst_stream_t stream) {
  MST_CHECK_ARG(N > 0 && V > 0 && i >= 1 && i < L && probs && seqs && scores && word && ldp >= V, "mst_sample_step: bad argument");
  hipLaunchKernelGGL(sample_step_kernel, dim3((unsigned)cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, N, V, i, L, probs, ldp, seqs, scores, word,
                     active, seed, eos, pad);
  MST_CHECK_LAUNCH("sample_step_kernel");
  return MST_OK;
}

extern "C" int mst_beam_step(int64_t B, int64_t K, int64_t V, int64_t i, int64_t L, const float* probs, int64_t ldp, const float* scores_in,
                             float* scores_out, const int32_t* seqs_in, int32_t* seqs_out, int32_t* hyp_src, int32_t* word, int32_t* active,
                             int32_t eos, int32_t pad, mst_stream_t stream) {
  MST_CHECK_ARG(B > 0 && K > 0 && K <= BEAM_MAXK && V > 0 && i >= 1 && i < L && K * V < (1ll << 30), "mst_beam_step: bad sizes (beam <= %d)", BEAM_MAXK);
  MST_CHECK_ARG(probs && scores_in && scores_out && seqs_in && seqs_out && hyp_src && word && ldp >= V, "mst_beam_step: null pointer or ldp < V");
  MST_CHECK_ARG(seqs_in != seqs_out && scores_in != scores_out, "mst_beam_step: the re-ranked rows need buffers of their own");
  BeamArgs q = {B, K, V, i, L, probs, ldp, scores_in, scores_out, seqs_in, seqs_out, hyp_src, word, active, eos, pad};
  hipLaunchKernelGGL(beam_step_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, q);
  MST_CHECK_LAUNCH("beam_step_kernel");
  return MST_OK;
}

extern "C" int mst_beam_gather_cols(const void* in, void* out, const int32_t* src, int64_t N, int64_t n_rows, int64_t row_bytes, int64_t t_max,
                                    int64_t skip_begin, int64_t skip_bytes, mst_stream_t stream) {
  MST_CHECK_ARG(in && out && src && in != out && N > 0 && n_rows > 0 && n_rows <= t_max && row_bytes > 0 && row_bytes % 16 == 0,
                "mst_beam_gather: bad argument (rows of a multiple of 16 bytes, distinct buffers)");
  MST_CHECK_ARG(((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0), "mst_beam_gather: buffers must be 16-byte aligned");
  MST_CHECK_ARG(skip_begin >= 0 && skip_bytes >= 0 && skip_begin % 16 == 0 && skip_bytes % 16 == 0 && skip_begin + skip_bytes <= row_bytes &&
                    skip_bytes < row_bytes,
                "mst_beam_gather_cols: the skipped range must be whole 16-byte pieces inside the row and leave something to copy");
  const int64_t total = N * n_rows * ((row_bytes - skip_bytes) / 16);
  int64_t grid = cdiv(total, 256 * 4);
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(beam_gather_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)in, (uint8_t*)out, src, N,
                     n_rows, row_bytes, t_max, skip_begin, skip_bytes);
  MST_CHECK_LAUNCH("beam_gather_kernel");
  return MST_OK;
}
extern "C" int mst_beam_gather(const void* in, void* out, const int32_t* src, int64_t N, int64_t n_rows, int64_t row_bytes, int64_t t_max,
                               mst_stream_t stream) {
  return mst_beam_gather_cols(in, out, src, N, n_rows, row_bytes, t_max, 0, 0, stream);
}

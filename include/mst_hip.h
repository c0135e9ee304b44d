/*
 * mst_hip.h — C-ABI of libmst_hip.so: the MI355X (gfx950) kernels behind the
 * VarAutoEncoder training step of slyforce/MusicStyleTransfer.
 *
 * Every entry point takes raw DEVICE pointers, explicit sizes / leading
 * dimensions and a hipStream_t (passed as void*), returns 0 on success or a
 * negative mst_status, never allocates, never synchronises, and may be
 * captured into a hipGraph. mst_last_error() returns a thread-local message
 * for the last non-zero status.
 *
 * The reference has no native code (it is Python on MXNet 1.3); each function
 * below replaces the MXNet operator call site(s) cited next to it. Paths are
 * relative to /root/reference/music_style_transfer/.
 *
 * Conventions
 *   - "act" tensors (activations) are 16-bit: MST_BF16 or MST_F16, selected by
 *     the `dtype` argument. Parameters / gradients / optimizer state are fp32.
 *   - Row-major everywhere. ld* are leading dimensions in ELEMENTS.
 *   - Dense weights keep MXNet's [units, in_units] layout (y = x W^T + b).
 */
#ifndef MST_HIP_H
#define MST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mst_stream_t; /* hipStream_t */

enum mst_status {
  MST_OK = 0,
  MST_ERR_INVALID = -1, /* bad argument (shape/alignment/dtype)             */
  MST_ERR_LAUNCH = -2,  /* hip launch / runtime error (see mst_last_error)  */
  MST_ERR_UNSUPPORTED = -3
};

enum mst_dtype { MST_BF16 = 0, MST_F16 = 1, MST_F32 = 2 };

/* epilogue activation for mst_gemm_nt */
enum mst_act { MST_ACT_NONE = 0, MST_ACT_RELU = 1 };

int mst_version(void);
const char* mst_last_error(void);
/* number of HIP devices visible, or negative status (used by the loader to fail loudly) */
int mst_device_count(void);

/* ---- stream-capture helpers (hipGraph instead of a tracing compiler) ---- */
int mst_graph_begin(mst_stream_t stream);
int mst_graph_end(mst_stream_t stream, void** graph_exec_out);
int mst_graph_launch(void* graph_exec, mst_stream_t stream);
int mst_graph_destroy(void* graph_exec);

/* ---- hip events on an explicit stream (bench.py times kernels with these) ---- */
int mst_event_create(void** ev_out);
int mst_event_record(void* ev, mst_stream_t stream);
int mst_event_sync(void* ev);
int mst_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out);
int mst_event_destroy(void* ev);

/* ------------------------------------------------------------------------
 * K3/K7/K8/K10/K12 + piano-roll in/out Dense: C[M,N] = epi(A[M,K] · B[N,K]^T)
 * Replaces gluon.nn.Dense(flatten=False) at transformer.py:36-40,65-68,
 * model.py:70-71,214-227 and, with B = W^T shadow, their dgrad.
 *
 *   A, B      : act dtype; K must be a multiple of 8; lda, ldb multiples of 8
 *   C         : act dtype (c_f32 = 0) or fp32 (c_f32 = 1); ldc multiple of 4.
 *               Columns N..min(roundup4(N),ldc) are written as exact zeros.
 *   bias      : fp32 [N] or NULL
 *   resid     : act dtype [M, ldr] or NULL (residual branch, added after dropout)
 *   act       : mst_act applied to alpha*(acc+bias+grpadd)
 *   gate      : act dtype [M, ldg] or NULL; result is zeroed where gate <= 0
 *               (ReLU backward fused into the FFN2 dgrad, transformer.py:36-38)
 *   alpha     : scalar applied to (acc + bias + grpadd)  (sqrt(D), transformer.py:237,270)
 *   rowadd    : fp32 [rowadd_period, ldra] or NULL; rowadd[(m % rowadd_period)] added
 *               AFTER alpha·(acc+bias+grpadd) (positional table, transformer.py:204-211)
 *   grpadd    : fp32 table [*, ldga] gathered by grp_index[m / rowadd_period] (int32) or NULL;
 *               added BEFORE alpha (class embedding, model.py:89-91)
 *   a_rows_per_group/a_group_stride/a_group_offset : if a_rows_per_group > 0, logical row m of A
 *               lives at physical row (m / rpg)*stride + offset + (m % rpg)   (model.py:253 drops row 0)
 *   c_rows_per_group/...: same remap for C rows (model.py:244 concat after the initial state)
 *   dropout   : p in [0,1); if p > 0 the epilogue applies inverted dropout with the
 *               counter-based mask keep(seed, site, row*N+n), row = PHYSICAL output row (transformer.py:35,149,182)
 *   self_resid: out = t + dropout(t) (decoder LN3(ff + dropout(ff)), transformer.py:199-200)
 * Epilogue order: t = alpha*(acc + bias + grpadd) -> act -> [u = dropout(t); self_resid: u += t]
 *                 -> + rowadd -> + resid -> gate.
 * ------------------------------------------------------------------------ */
typedef struct mst_gemm_args {
  int32_t dtype;
  int32_t c_f32;
  int64_t M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  const float* bias;
  const void* resid; int64_t ldr;
  int32_t act;
  const void* gate; int64_t ldg;
  float alpha;
  const float* rowadd; int64_t ldra; int64_t rowadd_period;
  const float* grpadd; int64_t ldga; const int32_t* grp_index;
  int64_t a_rows_per_group, a_group_stride, a_group_offset;
  int64_t c_rows_per_group, c_group_stride, c_group_offset;
  float dropout_p; uint64_t dropout_seed; uint32_t dropout_site;
  int32_t self_resid;
  const uint64_t* dropout_seed_ptr; /* optional DEVICE word XORed into dropout_seed (per-step seed under graph replay) */
  int32_t a_u8; /* 1: A is uint8 [M, lda bytes] (piano-roll frames as they arrive from the batcher), widened to the
                 * activation type while the tile is staged into LDS: the frames are never stored in 16 bits. Offered
                 * for the embedding GEMMs' form: 16-bit C, no dropout / self_resid, lda % 8 == 0. */
  int32_t resid_phys; /* 1: the residual shares C's PHYSICAL rows (it follows the C row remap); 0: it is indexed by the logical row m,
                       * as a strided view would be */
} mst_gemm_args;

int mst_gemm_nt(const mst_gemm_args* args, mst_stream_t stream);

/* Two mst_gemm_nt problems in ONE launch: the piano-roll ends' two embedding GEMMs (model.py:81-91 encoder input and
 * model.py:241-245 decoder input, transformer.py:237,270: the same uint8 frames against two tables), whose outputs and epilogues
 * differ but whose kernel form is the same. Both problems uint8-A, 16-bit C, whole 64 x 64 tiles, row-indexed adds / C row
 * remap only: one launch; anything else: exactly mst_gemm_nt(args0) followed by mst_gemm_nt(args1). */
int mst_gemm_nt_pair(const mst_gemm_args* args0, const mst_gemm_args* args1, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * K12 + K14 in one launch: the decoder's output layer Dense[D -> P] (model.py:253-256) with sigmoid + BinaryCrossEntropy
 * (loss.py:27-80) in its epilogue, for P = 128 or 256 (a 64-row x P tile holds whole rows of pitches, and — T a multiple
 * of 64 — rows of one sample). `args` is the output layer's GEMM (A = decoder output with its row remap, B = weight, bias,
 * alpha); its C receives the LOGIT GRADIENT d(sum_b loss_b)/dlogit * gscale (NULL: forward only). The logits themselves
 * are not stored (unless `logits` is set); they are rounded to the activation type before the loss arithmetic, so the
 * result equals mst_gemm_nt followed by mst_sigmoid_bce. loss[b] must be zero on entry (mst_step_begin's zero list).
 * ------------------------------------------------------------------------ */
typedef struct mst_bce_args {
  const uint8_t* labels;      /* {0,1} [M, N] */
  int64_t T;                  /* rows per sample */
  float label_smoothing;
  int32_t downweight;         /* loss.py:50-54,58-81 negative-label down-weighting */
  float* loss;                /* fp32 [M / T], accumulated */
  void* probs; int64_t ldp;   /* optional act dtype [M, ldp]: sigmoid output */
  void* logits; int64_t ldl;  /* optional act dtype [M, ldl] */
  float gscale;
} mst_bce_args;
int mst_gemm_sigmoid_bce(const mst_gemm_args* args, const mst_bce_args* bce, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * Dense + LayerNorm in one launch (transformer.py:155,158,197,200 forward; their autograd backward): the GEMM's tile
 * spans the whole output row (N = 128 or 256), so the row statistics are available in its epilogue.
 *   mode 1, forward : C = h = epi(A B^T) as mst_gemm_nt (bias, alpha, dropout / self_resid, resid, C row remap);
 *                     out = LayerNorm(h; gamma, beta, eps); mean[r], rstd[r] with r = PHYSICAL C row.
 *                     Equals mst_gemm_nt followed by mst_layernorm_fwd on C.
 *   mode 2, backward: dy = epi(A B^T) (bias, alpha, resid) is NOT stored; C = dx = LayerNorm backward of dy with
 *                     x (the forward's pre-norm tensor), mean, rstd all taken at the PHYSICAL C row r (the forward
 *                     wrote them there), gamma; out = dropout-masked copy of dx (mask_mode 1) at the LOGICAL row m;
 *                     dgamma / dbeta accumulated into — or, with `partials` set, NOT touched: workgroup w (one per
 *                     64 output rows; mst_gemm_nt_ln_parts(M) of them) stores its column sums at
 *                     partials[w * 2N + 0..N) (dgamma) and [N..2N) (dbeta) with plain stores, and the caller adds
 *                     them up later with mst_partial_sums (256 same-address fp32 atomics per column cost ~5 us of
 *                     serialisation per launch at configs[1]).
 *                     mask modes and the dropout counter (p / seed / site in the mst_gemm_args dropout fields) are
 *                     those of mst_layernorm_bwd. Equals mst_gemm_nt followed by mst_layernorm_bwd on its result.
 * ------------------------------------------------------------------------ */
typedef struct mst_ln_args {
  int32_t mode;
  const float* gamma;
  const float* beta;
  float eps;
  void* out; int64_t ld_out;
  float* mean; float* rstd;
  const void* x; int64_t ld_x;
  float* dgamma; float* dbeta;
  int32_t mask_mode;
  float* partials;        /* mode 2, optional: [mst_gemm_nt_ln_parts(M)][2N] fp32, 16-byte aligned */
} mst_ln_args;

int mst_gemm_nt_ln(const mst_gemm_args* args, const mst_ln_args* ln, mst_stream_t stream);
/* mst_gemm_sigmoid_bce(args, bce) and mst_gemm_nt_ln(dgrad, ln) in ONE launch where `dgrad` is the output layer's input gradient
 * (model.py:253-256 backward): A = the logit gradient `args` has just produced (dgrad->A == args->C), 128 pitches, row width 128,
 * ln->mode 2 — the workgroup that holds a 64-row tile of logit gradients contracts it with W_out and runs the last decoder layer's
 * LayerNorm backward on the result; the logit gradient is still stored (the weight-gradient launch reads it). Any other shape:
 * exactly the two launches. */
int mst_gemm_sigmoid_bce_dgrad_ln(const mst_gemm_args* args, const mst_bce_args* bce, const mst_gemm_args* dgrad,
                                  const mst_ln_args* ln, mst_stream_t stream);
int64_t mst_gemm_nt_ln_parts(int64_t M);

/* The whole feed-forward block of a Transformer layer in one launch (transformer.py:38-40 with :152-158 or :194-200):
 *     mst_gemm_nt(ff1)              a  = dropout(relu(x W1^T + b1))       — written (the backward pass needs it), never re-read
 *     mst_gemm_nt_ln(ff2, ln) mode 1 h2 = epi(a W2^T + b2), y = LayerNorm(h2), mean, rstd
 * with identical results (same MFMA order per output element, same epilogues). ff2->A must be ff1->C; the model width
 * (ff1 K = ff2 N) is 128 or 256 and divides the hidden width; no remaps, gates or row-indexed adds. */
int mst_ffn_ln_fwd(const mst_gemm_args* ff1, const mst_gemm_args* ff2, const mst_ln_args* ln, mst_stream_t stream);
/* The block's backward pass, same kernel skeleton (autograd of the above):
 *     mst_gemm_nt(ff2_dgrad)               d(pre) = ((dff W2) * alpha) gated by a > 0     — written (the weight-gradient launch reads it)
 *     mst_gemm_nt_ln(ff1_dgrad, ln) mode 2 dx = LayerNorm-backward(d(pre) W1 + resid), masked copy, dgamma / dbeta (or partials:
 *                                          mst_gemm_nt_ln_parts(M) rows)
 * ff2_dgrad->gate is the forward's hidden activation; ff1_dgrad->A must be ff2_dgrad->C. */
int mst_ffn_ln_bwd(const mst_gemm_args* ff2_dgrad, const mst_gemm_args* ff1_dgrad, const mst_ln_args* ln, mst_stream_t stream);
/* The same with the layer's LEADING LayerNorm backward (the one whose output gradient the block receives: LayerNorm-2 of an
 * encoder layer, transformer.py:158) computed in the kernel's prologue instead of by an mst_layernorm_bwd launch in front:
 * dx = LayerNorm-backward(dy; x, mean, rstd, gamma) on the workgroup's 64 rows, dx and (mask_mode 1) its dropout-masked copy
 * stored for the residual branch / the weight gradients, the masked copy (or dx) handed to the first GEMM on chip —
 * ff2_dgrad->A must be that buffer. dgamma / dbeta as in mst_gemm_nt_ln (partials: mst_gemm_nt_ln_parts(M) rows). */
typedef struct mst_ln_bwd_in {
  const void* dy; int64_t ld_dy;
  const void* x;  int64_t ld_x;
  const float* gamma; const float* mean; const float* rstd;
  void* dx; int64_t ld_dx;
  void* dx_masked; int64_t ld_dxm;
  float* dgamma; float* dbeta; float* partials;
  int32_t mask_mode;     /* 0 or 1 */
  float dropout_p; uint64_t dropout_seed; const uint64_t* dropout_seed_ptr; uint32_t dropout_site;
} mst_ln_bwd_in;
int mst_ffn_ln_bwd_lead(const mst_ln_bwd_in* lead, const mst_gemm_args* ff2_dgrad, const mst_gemm_args* ff1_dgrad,
                        const mst_ln_args* ln, mst_stream_t stream);
/* The attention output projection joins the block (MultiHeadDotAttention's W_proj Dense, transformer.py:65-68,105-106, and the
 * residual LayerNorm behind it, transformer.py:154-156 / 191-193):
 * in FRONT of mst_ffn_ln_fwd —
 *     mst_gemm_nt_ln(proj, ln1) mode 1     h1 = epi(att Wp^T + bp) (+ dropout, + residual), x1 = LayerNorm(h1), mean1, rstd1
 * with x1 (= ln1->out, which must be ff1->A) handed to the block on chip and stored, with h1 and the statistics, for backward.
 * The projection is width x width (N = K = the model width) on the same M rows. Same results as the separate launches. */
int mst_proj_ffn_ln_fwd(const mst_gemm_args* proj, const mst_ln_args* ln1, const mst_gemm_args* ff1, const mst_gemm_args* ff2,
                        const mst_ln_args* ln2, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * Deferred column sums: dst[0..len) += scale * sum_{p < n_parts} src[p*stride + 0..len), parts added in index order
 * (deterministic). The LayerNorm-backward launches leave per-workgroup partial sums of dgamma / dbeta (`partials`
 * above and in mst_layernorm_bwd); one launch of this adds all of a backward pass's sites into the gradient bucket.
 * len % 4 == 0, stride % 4 == 0, src and dst 16-byte aligned; up to 20 jobs (host array) per launch.
 * ------------------------------------------------------------------------ */
typedef struct mst_partial_sum {
  const float* src; int64_t n_parts; int64_t stride; int64_t len;
  float* dst; float scale;
} mst_partial_sum;
int mst_partial_sums(const mst_partial_sum* jobs, int n, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * Weight gradient: dW[N,K] (+)= sum_m A[m,n] * B[m,k]   (A = dY [M,N], B = X [M,K])
 * plus optional bias gradient db[n] += sum_m A[m,n].
 * Autograd counterpart of the Dense call sites above (trainer.py:176).
 * dW, db are fp32 and are ACCUMULATED INTO (callers zero the flat gradient
 * bucket once per step); ldw in elements. lda, ldb multiples of 8 and >= roundup8(N) / roundup8(K)
 * (pad columns must hold zeros).
 * Row remaps as in mst_gemm_args apply to A (a_*) and B (b_*).
 * beta_scale multiplies the contribution (sqrt(D) for embedding-side GEMMs).
 * ------------------------------------------------------------------------ */
typedef struct mst_wgrad_args {
  int32_t dtype;
  int64_t M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  float* dW; int64_t ldw;
  float* db;
  float scale;
  int64_t a_rows_per_group, a_group_stride, a_group_offset;
  int64_t b_rows_per_group, b_group_stride, b_group_offset;
  int32_t a_u8; /* 1: A is uint8 [M, lda bytes] (the embedding tables' gradient contracts the piano-roll frames), widened on
                 * the way into LDS; lda % 8 == 0 */
} mst_wgrad_args;

int mst_gemm_wgrad(const mst_wgrad_args* args, mst_stream_t stream);
/* up to 16 problems (host array) in ONE launch — the engine hands over every weight gradient of the backward pass at
 * once: one resident round of workgroups, the M-split (and with it the fp32-atomic traffic) as small as it gets */
int mst_gemm_wgrad_batch(const mst_wgrad_args* list, int n, mst_stream_t stream);
/* The same with a caller-owned fp32 scratch buffer (16-byte aligned). When the batch is large enough for 256x256 tiles
 * and the buffer holds tiles * split * 256 KiB, the M-slabs' partial tiles are written there with plain stores and
 * summed in slab order by a second launch: no fp32 atomics on dW (deterministic gradients, and 62 MB of atomic traffic
 * less at configs[1]). Otherwise identical to mst_gemm_wgrad_batch. 64 MiB covers configs[1]. */
int mst_gemm_wgrad_batch_ws(const mst_wgrad_args* list, int n, float* scratch, int64_t scratch_bytes, mst_stream_t stream);
/* The same plus up to 20 column-sum jobs (mst_partial_sum below: the LayerNorm-backward launches' per-workgroup dgamma /
 * dbeta rows): they are executed by extra workgroups of the reduction pass when there is one, else by one
 * mst_partial_sums launch after the weight gradients — either way the flush of a backward pass is one call. */
int mst_gemm_wgrad_batch_sums(const mst_wgrad_args* list, int n, float* scratch, int64_t scratch_bytes,
                              const mst_partial_sum* sums, int n_sums, mst_stream_t stream);

/* Deferred batch outer products: out[j, i] += sum_b L[b, j] * R[b, i] (out fp32 [J, I] contiguous, accumulated into) and, with
 * obias, obias[j] += sum_b L[b, j]. L fp32 [B, J] contiguous; R [B, >= I] of r_dtype (MST_F32 / MST_BF16 / MST_F16) with row stride
 * r_stride elements. The parameter gradients of a layer that sees one row per sample — the latent block's latent_proj and
 * latent2hid (model.py:97-103,229-232; what mst_latent_bwd's second launch computes from mst_latent_bwd_vec's `scratch`) — which
 * nothing downstream but the optimizer reads. Up to 2 jobs per call. */
typedef struct mst_outer_job {
  const float* L; const void* R; int32_t r_dtype; int64_t r_stride;
  int64_t B, J, I;
  float* out; float* obias;
} mst_outer_job;
int mst_outer_jobs(const mst_outer_job* jobs, int n, mst_stream_t stream);
/* mst_gemm_wgrad_batch_sums plus up to 2 outer-product jobs: extra workgroups of the reduction pass when there is one, else one
 * mst_outer_jobs launch after the weight gradients. */
int mst_gemm_wgrad_batch_flush(const mst_wgrad_args* list, int n, float* scratch, int64_t scratch_bytes,
                               const mst_partial_sum* sums, int n_sums, const mst_outer_job* outers, int n_outers,
                               mst_stream_t stream);

/* ------------------------------------------------------------------------
 * K1/K2: token path input. out[b, s_off + t, :] = alpha*(table[tok[b,t]] + cls[classes[b]]) + pos[s_off+t]
 * (model.py:86-91, transformer.py:270; decoder: model.py:241-245 with cls = NULL, s_off = 1).
 * Also emits keymask[b, s_off + t] = (tok != 0) if keymask != NULL (model.py:81-83).
 * table, cls, pos fp32; tokens/classes int32; out act dtype [B, S_out, ld].
 * ------------------------------------------------------------------------ */
int mst_embed_fwd(int dtype, int64_t B, int64_t T, int64_t D,
                  const int32_t* tokens, const float* table, int64_t ldt,
                  const int32_t* classes, const float* cls_table, int64_t ldc,
                  const float* pos, int64_t ldp, float alpha,
                  void* out, int64_t ld_out, int64_t S_out, int64_t s_off,
                  uint8_t* keymask, mst_stream_t stream);

/* scatter-add of alpha*dX rows into dtable (fp32, accumulated) and dcls (fp32, accumulated) */
int mst_embed_bwd(int dtype, int64_t B, int64_t T, int64_t D,
                  const int32_t* tokens, float* dtable, int64_t ldt,
                  const int32_t* classes, float* dcls, int64_t ldc, float alpha,
                  const void* dX, int64_t ld_dx, int64_t S_out, int64_t s_off,
                  mst_stream_t stream);

/* dst[idx[b], :] += alpha * sum_{t<T} X[b, s_off+t, :]  (fp32 dst, accumulated): gradient of the class
 * embedding that the piano-roll input GEMM's epilogue added to every frame (model.py:89-91) */
int mst_group_colsum(int dtype, int64_t B, int64_t T, int64_t D, const void* X, int64_t ldx,
                     int64_t S_out, int64_t s_off, const int32_t* idx, float* dst, int64_t ldd,
                     float alpha, mst_stream_t stream);

/* key-validity mask from lengths: keymask[b,s] = s < lens[b] + add (model.py:246-247 SequenceMask) */
int mst_mask_from_lengths(int64_t B, int64_t S, const int32_t* lens, int32_t add,
                          uint8_t* keymask, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * K4/K11: MultiHeadDotAttention with the reference's key-row softmax
 * (transformer.py:85-126): logits[k,q] = K[k]·Q[q]/sqrt(dh) + (keymask[k] ? 0 : -1e9),
 * P = softmax over q, O[q] = sum_k P[k,q] V[k].
 *   qkv   : act dtype [B*S, ld_qkv]; K at column k_off + h*dh, Q at q_off + h*dh, V at v_off + h*dh
 *   lse   : fp32 [2, B, H, S] softmax row statistics (written by fwd, read by bwd): plane 0 = row max,
 *           plane 1 = log(row sum). Kept apart because a padded key row has max -1e9, where a single
 *           fp32 logsumexp would round log(S) away.
 *   out   : act dtype [B*S, ld_out], head h at column h*dh
 * dh in {16, 32, 64}.
 * ------------------------------------------------------------------------ */
int mst_attn_keysoftmax_fwd(int dtype, int64_t B, int64_t S, int64_t H, int64_t dh,
                            const void* qkv, int64_t ld_qkv, int64_t k_off, int64_t q_off, int64_t v_off,
                            const uint8_t* keymask, float* lse,
                            void* out, int64_t ld_out,
                            int64_t q_limit /* produce only queries [0, q_limit); <= 0 or >= S: all. The row statistics
                                               always cover every query (they normalise over the query axis) */,
                            mst_stream_t stream);

/* dqkv has the same layout as qkv; delta is fp32 scratch [B, H, S]. */
/* The layer's K | Q | V projection AND its attention in one call (transformer.py:88-104): qkv = x W^T + bias with
 * W [3 D, ld_w] (row c = output column c of the qkv layout, D = H * dh) and bias fp32 [3 D] is computed, written to `qkv`
 * (the backward pass reads it) and attended to as mst_attn_keysoftmax_fwd does. For head size 32 and sequences that take the
 * resident attention kernel the projection runs INSIDE the attention launch — every (batch, head) workgroup forms its own
 * [S, 3 * 32] tile of the product — which saves a launch and the re-read of qkv; other shapes run mst_gemm_nt followed by
 * mst_attn_keysoftmax_fwd. Same results either way up to the summation order of the fp32 accumulation. */
int mst_attn_qkv_fwd(int dtype, int64_t B, int64_t S, int64_t H, int64_t dh, const void* x, int64_t ld_x, const void* w,
                     int64_t ld_w, const float* bias, void* qkv, int64_t ld_qkv, int64_t k_off, int64_t q_off, int64_t v_off,
                     const uint8_t* keymask, float* lse, void* out, int64_t ld_out, int64_t q_limit, mst_stream_t stream);
int mst_attn_keysoftmax_bwd(int dtype, int64_t B, int64_t S, int64_t H, int64_t dh,
                            const void* qkv, int64_t ld_qkv, int64_t k_off, int64_t q_off, int64_t v_off,
                            const uint8_t* keymask, const float* lse,
                            const void* dout, int64_t ld_dout,
                            void* dqkv, int64_t ld_dqkv, float* delta,
                            int64_t q_limit /* the caller's promise that rows [q_limit, S) of every sample of dout are ZERO
                                               (the mirror of the forward's q_limit; <= 0 or >= S: dense). With
                                               q_limit <= 32 the resident kernel skips the work that multiplies those
                                               zeros — results are bit-identical to the dense computation; the
                                               streaming kernels read dout in full, so the rows must really be zero */,
                            mst_stream_t stream);

/* ------------------------------------------------------------------------
 * K6: y = LayerNorm(x) * gamma + beta over the last axis (eps, biased variance;
 * gluon.nn.LayerNorm at transformer.py:142,147,175,180). The residual add is fused
 * into the producing GEMM's epilogue, so x is the pre-norm sum.
 *   x, y : act dtype [M, ld]; mean, rstd : fp32 [M] (saved for backward)
 * bwd: dx (act dtype) and dgamma/dbeta (fp32, ACCUMULATED INTO). The gradient arriving through the
 *   residual branch is added by the consuming dgrad GEMM's `resid` epilogue, not here.
 *   mask_mode 0: dx only. 1: also dx_masked = dx * keep/(1-p), the gradient of the dropped-out
 *   GEMM output feeding this norm (transformer.py:155,158,197). 2: dx <- dx * (1 + keep/(1-p)),
 *   the decoder's LN3(ff + dropout(ff)) (transformer.py:200). The keep mask is regenerated from
 *   (dropout_seed, dropout_site, m*D + d), the same counter RNG the forward GEMM epilogue used.
 * ------------------------------------------------------------------------ */
int mst_layernorm_fwd(int dtype, int64_t M, int64_t D, const void* x, int64_t ldx,
                      const float* gamma, const float* beta, float eps,
                      void* y, int64_t ldy, float* mean, float* rstd,
                      int64_t row_id_stride /* statistics of row m are stored at index m*row_id_stride (1 = dense) */,
                      mst_stream_t stream);

int mst_layernorm_bwd(int dtype, int64_t M, int64_t D, const void* x, int64_t ldx,
                      const float* gamma, const float* mean, const float* rstd,
                      const void* dy, int64_t ldy, void* dx, int64_t ld_dx,
                      void* dx_masked, int64_t ld_dxm, float* dgamma, float* dbeta,
                      int mask_mode, float dropout_p, uint64_t dropout_seed, uint32_t dropout_site,
                      const uint64_t* dropout_seed_ptr,
                      int64_t row_id_stride /* row m here is row m*row_id_stride of the forward (mean/rstd index and
                                               dropout counter); 1 for a dense pass */,
                      float* partials /* optional [mst_layernorm_bwd_parts(M, D)][2D] fp32: per-workgroup column sums
                                         (dgamma | dbeta) stored instead of atomics on dgamma / dbeta; add them
                                         with mst_partial_sums */,
                      mst_stream_t stream);
int64_t mst_layernorm_bwd_parts(int64_t M, int64_t D);

/* ------------------------------------------------------------------------
 * K8/K9/K10 latent block (model.py:97-103,292,229-232; loss.py:8-12), fp32 math:
 *   fwd: h0 = enc_out[b, 0, :]                     (act dtype, row stride ld_enc*S)
 *        [mu | sigma] = h0 · Wl^T + bl              (Wl fp32 [2Z, De])
 *        z = mu + eps * sigma ; kl[b] = 0.5 * sum_z (sigma^2 + mu^2 - 1 - log(sigma^2))
 *        dec_in[b, 0, :] = alpha_d*(z · Wh^T + bh + cls_d[classes[b]]) + pos_d[0]   (Wh fp32 [Dd, Z])
 *   bwd: given d(dec_in[b,0,:]) and beta (KL weight): all parameter grads (accumulated, fp32),
 *        d(enc_out[b,0,:]) written (act dtype) into denc row 0 of each sample.
 * ------------------------------------------------------------------------ */
int mst_latent_fwd(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd,
                   const void* enc_out, int64_t enc_sample_stride,
                   const float* Wl, const float* bl, const float* eps,
                   const float* Wh, const float* bh,
                   const int32_t* classes, const float* cls_d, int64_t ld_cls,
                   const float* pos_d, float alpha_d,
                   float* mu, float* sigma, float* z, float* kl,
                   void* dec_in, int64_t dec_sample_stride, mst_stream_t stream);
/* ... and, on the same launch, the decoder's first K | Q | V projection of THAT row (transformer.py:88-93 on position 0):
 * qkv0[b * qkv_sample_stride + j] = dec_in[b, 0, :] . Wq[j, :] + bq[j], j < nq (Wq: the 16-bit weight, [nq, ld_wq]; fp32 dot products
 * over the row as stored). The other B T rows of that projection do not depend on the latent block: the step computes them as a
 * rider of the forward position-0 tail (mst_row_tail_fwd_ride). Dd must be 64, 128 or 256. */
int mst_latent_fwd_proj(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd,
                        const void* enc_out, int64_t enc_sample_stride,
                        const float* Wl, const float* bl, const float* eps,
                        const float* Wh, const float* bh,
                        const int32_t* classes, const float* cls_d, int64_t ld_cls,
                        const float* pos_d, float alpha_d,
                        float* mu, float* sigma, float* z, float* kl,
                        void* dec_in, int64_t dec_sample_stride,
                        const void* Wq, int64_t ld_wq, const float* bq, void* qkv0, int64_t qkv_sample_stride, int64_t nq,
                        mst_stream_t stream);

int mst_latent_bwd(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd,
                   const void* enc_out, int64_t enc_sample_stride,
                   const float* Wl, const float* eps, const float* Wh,
                   const int32_t* classes,
                   const float* mu, const float* sigma, const float* z,
                   const void* d_dec_in, int64_t dec_sample_stride, float alpha_d,
                   float kl_weight, float gscale /* encoder-side loss scale */,
                   float enc_scale /* gscale / decoder-side loss scale */,
                   float* dWl, float* dbl, float* dWh, float* dbh, float* dcls_d, int64_t ld_cls,
                   void* d_enc_out, int64_t denc_sample_stride, float* scratch /* fp32 [B*(Dd+2Z)] */,
                   mst_stream_t stream);
/* mst_latent_bwd's first launch on its own, with the decoder class table's gradient (dcls_d[classes[b], :] += t[b, :]) folded in:
 * leaves t = alpha_d * d(dec_in[b, 0, :]) at scratch[0 .. B*Dd) and d[mu | sigma] at scratch[B*Dd .. B*(Dd + 2Z)), writes d(enc_out)
 * row 0. The remaining parameter gradients are two mst_outer_job of the caller's weight-gradient flush:
 *   dWl[2Z, De] += dlat^T enc_out[:, 0, :], dbl += sum_b dlat;   dWh[Dd, Z] += t^T z, dbh += sum_b t. */
int mst_latent_bwd_vec(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const float* Wl, const float* eps, const float* Wh,
                       const int32_t* classes, const float* mu, const float* sigma, const void* d_dec_in, int64_t dec_sample_stride,
                       float alpha_d, float kl_weight, float gscale, float enc_scale, float* dcls_d, int64_t ld_cls,
                       void* d_enc_out, int64_t denc_sample_stride, float* scratch, mst_stream_t stream);
/* mst_latent_bwd_vec with d(dec_in[b, 0, :]) COMPUTED instead of read: dq0[b * dq_sample_stride + k] (k < nq: the gradient of the
 * decoder's first K | Q | V projection at position 0) against Wt [Dd, ld_wt >= nq] (the TRANSPOSED 16-bit weight, the operand of the
 * input-gradient GEMM) plus resid0[b * resid_sample_stride + :] (the residual branch's gradient row, may be NULL), rounded to the
 * activation type as that GEMM rounds it — whose other B T rows the step computes as a rider of the backward position-0 tail
 * (mst_row_tail_bwd_ride). nq must be 384 or 768 (decoder width 128 or 256). */
int mst_latent_bwd_vec_proj(int dtype, int64_t B, int64_t De, int64_t Z, int64_t Dd, const float* Wl, const float* eps, const float* Wh,
                            const int32_t* classes, const float* mu, const float* sigma,
                            const void* dq0, int64_t dq_sample_stride, const void* Wt, int64_t ld_wt, int64_t nq,
                            const void* resid0, int64_t resid_sample_stride,
                            float alpha_d, float kl_weight, float gscale, float enc_scale, float* dcls_d, int64_t ld_cls,
                            void* d_enc_out, int64_t denc_sample_stride, float* scratch, mst_stream_t stream);

/* standalone reparameterisation + KL (loss.VariationalKLLoss, loss.py:4-12; model.py:292) */
int mst_reparam_kl_fwd(int64_t B, int64_t Z, const float* mu, const float* sigma, const float* eps,
                       float* z, float* kl, mst_stream_t stream);
int mst_reparam_kl_bwd(int64_t B, int64_t Z, const float* mu, const float* sigma, const float* eps,
                       const float* dz, float kl_weight, float* dmu, float* dsigma, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * The position-0 tail of the top encoder layer in one launch (the model reads the encoder at position 0 only, model.py:97):
 *   h1 = resid + dropout(att Wp^T + bp);  x1 = LN1(h1);  a = dropout(relu(x1 W1^T + b1));  h2 = x1 + dropout(a W2^T + b2);
 *   x2 = LN2(h2)                                            (transformer.py:42-46,154-158 on B rows, B <= 64, D = 128 | 256)
 * = mst_gemm_nt (+resid) · mst_layernorm_fwd · mst_gemm_nt (ReLU) · mst_gemm_nt (+resid) · mst_layernorm_fwd on those rows,
 * same rounding points and dropout counters (sites site0, site0+1, site0+2; counter row = row * phys_stride), as D / 16
 * workgroups that own output-column slices and meet at three grid barriers. Row r of every tensor sits at element offset
 * r * (its row stride); statistics at index r * stat_stride. `sync`: THREE zeroed device words (mst_step_begin's zero list): barrier counter, claimed XCD, roles handed out — the launch oversubscribes the grid and keeps the workgroups of one XCD (one L2).
 * ------------------------------------------------------------------------ */
typedef struct mst_row_tail_args {
  int32_t dtype;
  int64_t B, D;
  const void* att; int64_t rs_att;
  const void* resid; int64_t rs_res;
  const void* Wp; int64_t ldwp; const float* bp;
  const float* g1; const float* be1;
  const void* W1; int64_t ldw1; const float* b1;
  const void* W2; int64_t ldw2; const float* b2;
  const float* g2; const float* be2;
  void* h1; void* x1; void* h2; void* x2; int64_t rs_d;
  void* a; int64_t rs_a;
  float* mean1; float* rstd1; float* mean2; float* rstd2; int64_t stat_stride;
  float eps;
  float dropout_p; uint64_t dropout_seed; const uint64_t* dropout_seed_ptr; uint32_t site0;
  int64_t phys_stride;
  uint32_t* sync;    /* THREE zeroed device words: [0] barrier counter (3 * D/16 after a complete launch), [1] claimed XCD + 1, [2] roles */
  uint32_t* status;  /* optional sticky device word: MST_TAIL_* flags are OR-ed into it when the launch could not do its work */
} mst_row_tail_args;
int mst_row_tail_fwd(const mst_row_tail_args* args, mst_stream_t stream);
/* RIDERS: the chain above keeps ONE XCD busy for ~26 us while seven idle. mst_row_tail_*_ride launch the same kernel with enough
 * workgroups for every CU; those that land on another XCD than the chain's compute the 128 x 128 tiles of `rider` — ONE
 * mst_gemm_nt problem that nothing in the chain reads (M, N multiples of 128, K of 64; 16-bit C; bias, alpha, a residual and
 * row remaps in whole tiles only) — from a work queue (`queue`: ONE zeroed device word, in another cache line than `sync`);
 * the chain's own workgroups pass by the queue when they are done, so every tile is computed by the end of the launch wherever
 * the workgroups land. The training step rides the decoder's K | Q | V projection of rows 1..T (transformer.py:88-93; its input
 * rows exist since the step's first launch) on the forward tail and that projection's input gradient on the backward tail:
 * two launches (12 + 10 us) less in the step's dependent chain. Same results as mst_gemm_nt, bit for bit. */
int mst_row_tail_fwd_ride(const mst_row_tail_args* args, const mst_gemm_args* rider, uint32_t* queue, mst_stream_t stream);
/* ... with the step's transposed-shadow refresh (mst_transpose_shadows' list, in the tail's activation type) BEHIND the rider's tiles in
 * the same queue, sixteen 32 x 32 tiles per ticket: the 16-bit transposed copies that only the backward pass reads (nothing in this launch
 * does) are rebuilt on compute units that idle until the chain ends, instead of 4.9 us on the step's first launch
 * (mst_step_begin_args.sh_*). The weights they are built from must be final (the previous step's optimizer launch precedes). */
int mst_row_tail_fwd_ride_shadows(const mst_row_tail_args* args, const mst_gemm_args* rider, uint32_t* queue, int sh_dtype,
                                  const float* sh_w, void* sh_wt16, const int64_t* sh_desc, const int64_t* sh_prefix, int64_t sh_n_mat,
                                  int64_t sh_tiles, mst_stream_t stream);
/* The same rows on the way back: autograd of mst_row_tail_fwd's chain for the gradient `dy` of the layer's output rows —
 *     mst_layernorm_bwd (LayerNorm-2; dh, its dropout-masked copy dhm, dgamma2 / dbeta2 +=)
 *  -> mst_gemm_nt(dhm, W2t, gate = a, alpha = 1 / (1 - p))   d(pre)                 [B, 4 D]
 *  -> mst_gemm_nt(d(pre), W1t, resid = dh)                   dx1
 *  -> mst_layernorm_bwd (LayerNorm-1; dh1 rows, masked copy dh1m, dgamma1 / dbeta1 +=)
 *  -> mst_gemm_nt(dh1m, Wpt)                                 datt rows
 * in one launch (same results to rounding of the LayerNorm sums). W2t [4 D, D], W1t [D, 4 D], Wpt [D, D]: the transposed
 * 16-bit weights (K-contiguous dgrad operands). dh / dhm / dx1 / dh1m: compact [B, >= D] scratch rows (row stride rs_c);
 * dh1 / datt: rows of the strided full-size buffers the following launches read. `sync`: three zeroed device words (as above). */
typedef struct mst_row_tail_bwd_args {
  int32_t dtype;
  int64_t B, D;
  const void* dy; int64_t rs_dy;
  const void* h2; const void* h1; int64_t rs_d;
  const void* a; int64_t rs_a;
  const float* mean1; const float* rstd1; const float* mean2; const float* rstd2; int64_t stat_stride;
  const float* g1; const float* g2;
  const void* W2t; int64_t ldw2t;
  const void* W1t; int64_t ldw1t;
  const void* Wpt; int64_t ldwpt;
  void* dh; void* dhm; void* dx1; void* dh1m; int64_t rs_c;
  void* dpre; int64_t rs_dpre;
  void* dh1; int64_t rs_dh1;
  void* datt; int64_t rs_datt;
  float* dg1; float* db1; float* dg2; float* db2;
  float dropout_p; uint64_t dropout_seed; const uint64_t* dropout_seed_ptr; uint32_t site0;
  int64_t phys_stride;
  uint32_t* sync;    /* three zeroed device words, as above ([0] ends at 2 * D/16) */
  uint32_t* status;  /* optional, as above */
} mst_row_tail_bwd_args;
int mst_row_tail_bwd(const mst_row_tail_bwd_args* args, mst_stream_t stream);
int mst_row_tail_bwd_ride(const mst_row_tail_bwd_args* args, const mst_gemm_args* rider, uint32_t* queue, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * Incremental decode (inference; model.py:259-272, transformer.py:70-77,242-249): the new position's query against the
 * K | Q | V rows cached so far. cache: act dtype [B, t_max, ld] with the training layout (k_off / q_off / v_off, head h at
 * columns h*dh); the new row is row n_keys - 1 and is already cached. out: act dtype [B, ld_out].
 *   mode 0: the reference's arithmetic — softmax over the QUERY axis, which holds the one new query: P = 1, out = sum of
 *           the cached value rows;   mode 1: softmax over the cached keys.
 * ------------------------------------------------------------------------ */
int mst_attn_decode(int dtype, int64_t B, int64_t H, int64_t dh, int64_t n_keys, int64_t t_max, const void* cache,
                    int64_t ld, int64_t k_off, int64_t q_off, int64_t v_off, int mode, void* out, int64_t ld_out,
                    mst_stream_t stream);

/* ------------------------------------------------------------------------
 * Beam search on the device (sampler.py:198-257, token ends), one position per call pair — capturable with the decode step.
 * mst_beam_step: B samples x K hypotheses (K <= 16); probs fp32 [B*K, ldp] = the decoder's distribution of position i; a
 *   hypothesis whose last token (seqs_in[., i-1]) is EOS — or PAD from position 2 on — is finished and continues with PAD only,
 *   at no cost (sampler.py:218-221); every other continuation costs -log max(p, 1e-30). Per sample the K candidates with the
 *   smallest (score, hypothesis * V + word) are kept: new hypothesis r of sample b gets scores_out, its source hypothesis in
 *   hyp_src (a row index into the B*K hypotheses), its word in word[] (int32, the next position's input) and its token row
 *   seqs_out[., 0..i] (int32 rows of L). active[i] (optional, zeroed by the caller) += hypotheses still running after position i.
 * mst_beam_gather: out[j, r, :] = in[src[j], r, :] for r < n_rows — the K | Q | V cache rows ([N, t_max, row_bytes]) of the
 *   re-ranked hypotheses (sampler.py:236-238); in and out are distinct buffers (the decode plan alternates between two).
 * ------------------------------------------------------------------------ */
int mst_beam_step(int64_t B, int64_t K, int64_t V, int64_t i, int64_t L, const float* probs, int64_t ldp, const float* scores_in,
                  float* scores_out, const int32_t* seqs_in, int32_t* seqs_out, int32_t* hyp_src, int32_t* word, int32_t* active,
                  int32_t eos, int32_t pad, mst_stream_t stream);
int mst_beam_gather(const void* in, void* out, const int32_t* src, int64_t N, int64_t n_rows, int64_t row_bytes, int64_t t_max,
                    mst_stream_t stream);
/* ... the same with bytes [skip_begin, skip_begin + skip_bytes) of every row left alone (whole 16-byte pieces): the Q third of the
 * cache rows, which no later position reads (a decode step takes its query from the newest row only) — a third less traffic. */
int mst_beam_gather_cols(const void* in, void* out, const int32_t* src, int64_t N, int64_t n_rows, int64_t row_bytes, int64_t t_max,
                         int64_t skip_begin, int64_t skip_bytes, mst_stream_t stream);
/* Ancestral sampling on the device (sampler.py:155-190): every sequence n of N draws token i from probs[n, :V] (fp32, need not be
 * normalised) by inverse CDF with u = counter hash of (seed, i, n); written to seqs[n, i] (int32 rows of L) and word[n] (the next
 * position's input); scores[n] += -log p; a finished sequence (EOS, or PAD from position 2 on) continues with PAD at no cost;
 * active[i] (optional, zeroed by the caller) += sequences still running. Capturable: the draw depends on device data and on the
 * host constants (seed, i) only. */
int mst_sample_step(int64_t N, int64_t V, int64_t i, int64_t L, const float* probs, int64_t ldp, int32_t* seqs, float* scores,
                    int32_t* word, int32_t* active, uint64_t seed, int32_t eos, int32_t pad, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * K12/K13: softmax over V + SoftmaxCrossEntropy (model.py:256; loss.py:15-23).
 *   logits : act dtype [M = B*T, ld]; labels int32 [M]
 *   loss[b] = (1/T) * sum_t -log p[b,t,label] * (label != 0)     (fp32 [B], written)
 *   probs  : optional fp32 [M, ldp] output (reconstruction)
 *   dlogits: optional act dtype [M, ld]: (p - onehot) * (label != 0) / T * gscale; pad cols zero
 *   tok_parts: optional fp32 [MST_CE_MAX_WORKGROUPS, 4], the masked token metrics of trainer.py:107-113 kept on the
 *            device: every workgroup ADDS {sum -log max(p[label], 1e-10), #(label is the arg-max), #(label among
 *            the top_k), #(label != 0)} of its rows to its own row (no atomics; the caller clears the buffer when it
 *            reads the metrics and sums the rows): ppl = exp([0]/[3]), acc = [1]/[3], topk = [2]/[3].
 * ------------------------------------------------------------------------ */
#define MST_CE_MAX_WORKGROUPS 4096
int mst_softmax_ce(int dtype, int64_t B, int64_t T, int64_t V,
                   const void* logits, int64_t ld, const int32_t* labels,
                   float* loss, float* probs, int64_t ldp,
                   void* dlogits, int64_t ldd, float gscale,
                   int pre_zeroed /* 1: loss[] is already zero (mst_step_begin's zero list), skip the memset node */,
                   float* tok_parts, int top_k, mst_stream_t stream);

/* The reference's own call forms of the two reconstruction losses, on PROBABILITIES (what Model(...) returns):
 *   SoftmaxCrossEntropy()(probs, labels)                  loss.py:16-23: -log(pick(pred, label)) * (label != 0), / padded T
 *   BinaryCrossEntropy(from_sigmoid=True)(probs, labels)  loss.py:40-56 without the sigmoid
 * probs: dtype MST_F32 / MST_BF16 / MST_F16, [B*T, ldp] resp. contiguous [B, n_per_sample]; loss fp32 [B], written.
 * Forward only (the training step differentiates the fused logit forms above). */
int mst_ce_from_probs(int dtype, int64_t B, int64_t T, int64_t V, const void* probs, int64_t ldp,
                      const int32_t* labels, float* loss, mst_stream_t stream);
int mst_bce_from_probs(int dtype, int64_t B, int64_t n_per_sample, const void* probs, const uint8_t* labels,
                       float label_smoothing, int downweight, float* loss, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * K14: sigmoid + BinaryCrossEntropy (loss.py:27-80), piano-roll head.
 *   logits : act dtype [B, T*P] viewed as [B*T, ld] with P valid columns
 *   labels : uint8 {0,1} [B*T, P]
 *   loss[b] = mean_{t,p} bce ; bce = -(s log(1e-12+p) + (1-s) log(1e-12+1-p)), s = (1-ls)*y + 0.5*ls
 *   negative down-weighting (downweight != 0): where y == 0, bce <- w_b * bce^2,
 *   w_b = n_pos_b / (n_neg_b + 1e-12) (loss.py:50-54,58-81); npos is int32 [B] scratch.
 *   probs  : optional act dtype [B*T, ldp] (sigmoid output = reconstructed piano-roll)
 *   dlogits: optional act dtype, d(sum_b loss_b)/dlogit * gscale
 * ------------------------------------------------------------------------ */
int mst_sigmoid_bce(int dtype, int64_t B, int64_t T, int64_t P,
                    const void* logits, int64_t ld, const uint8_t* labels,
                    float label_smoothing, int downweight, int32_t* npos,
                    float* loss, void* probs, int64_t ldp,
                    void* dlogits, int64_t ldd, float gscale,
                    int pre_zeroed /* as in mst_softmax_ce */, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * K15/K21: total[b] = recon[b] + kl_weight * kl[b]; metric_acc[0] += sum_b kl, [1] += sum_b total,
 * [2] += B (running sums kept on device: trainer.py:107-120,172,181-186).
 * ------------------------------------------------------------------------ */
int mst_loss_combine(int64_t B, const float* recon, const float* kl, float kl_weight,
                     float* total, float* metric_acc, mst_stream_t stream);

/* ------------------------------------------------------------------------
 * K16: multi-tensor Adam over one flat fp32 bucket with MXNet's update rule
 * (gluon.Trainer.step → optimizer.Adam → adam_update; trainer.py:94-101,177):
 *   g = clip(grad * rescale + wd * w, ±clip)  (clip < 0: no clipping)
 *   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g² ; w -= lr_t * m / (sqrt(v) + eps)
 *   lr_t = lr * sqrt(1-b2^t)/(1-b1^t), evaluated in double like the Python reference (lr, b1, b2 are doubles for
 *   that reason: 1 - 0.999f is already 1.3e-5 off); t lives in step_state[0] on the device and is advanced by a
 *   1-thread kernel issued ahead of the update in the same stream, so a captured graph replays
 *   with the right bias correction.
 * Also refreshes the 16-bit shadow copy w16 (act dtype, same offsets).
 * metrics (optional): the end-of-step bookkeeping of mst_loss_combine (total[b] = recon[b] + kl_weight * kl[b] and the
 * running metric sums; trainer.py:107-120,172,181-186) done by the first workgroup of this launch instead of a launch of
 * its own.
 * ------------------------------------------------------------------------ */
typedef struct mst_step_metrics {
  int64_t B; const float* recon; const float* kl; float kl_weight; float* total; float* metric;
  /* Step guard (all optional). status: two sticky device words {flags, number of skipped steps}. A step COUNTS only if
   * status[0] == 0 and every given *expect_ptr holds expect_val (the barrier counters of the one-launch position-0 tails,
   * mst_row_tail_*: a tail that could not finish leaves its counter short). Otherwise mst_adam_flat leaves parameters,
   * moments and the step count as they were, the metric sums are not touched, flag MST_STEP_INCOMPLETE is added to
   * status[0] when an expectation failed and status[1] is incremented: the batch is skipped, not mis-learned. The host reads
   * the words with the metrics, takes its fallback and clears them. */
  uint32_t* status; const uint32_t* expect_ptr0; uint32_t expect_val0; const uint32_t* expect_ptr1; uint32_t expect_val1;
  /* Non-finite guard (optional; needs status, which then has a third word): fin_recon / fin_kl = the step's fin_B per-sample
   * losses. If one of them is not finite — an activation overflowed (fp16 at long sequences: the key-row softmax lets a query
   * collect up to T key masses, attention outputs of several thousand), or sigma reached 0 under log(sigma^2) — the gradients
   * are NaN too and an update would destroy every parameter: mst_adam_flat then leaves parameters, moments and the step count
   * alone, adds nothing to the metric sums and increments status[2]. NOT sticky: the next batch is tried again. Every optimizer
   * launch of the step gets the same pointers (a second range without `recon` skips with the first). */
  const float* fin_recon; const float* fin_kl; int64_t fin_B;
} mst_step_metrics;
#define MST_TAIL_SPIN_FWD 1u     /* a grid barrier of mst_row_tail_fwd gave up waiting */
#define MST_TAIL_SPIN_BWD 2u     /* ... of mst_row_tail_bwd */
#define MST_STEP_INCOMPLETE 16u  /* an expect_ptr of the step guard did not hold its value at the end of the step */
int mst_adam_flat(int dtype, int64_t n, float* w, const float* grad, float* m, float* v,
                  void* w16, double lr, double beta1, double beta2, float eps, float wd,
                  float rescale, float clip, int32_t* step_state /* device int32[2]: {t, bits(lr_t)} */,
                  int advance_step /* 0: reuse the lr_t of the previous launch (second range of one step) */,
                  const mst_step_metrics* metrics, mst_stream_t stream);
/* mst_loss_combine with the step guard of mst_step_metrics (validation steps: a step whose position-0 tail failed adds
 * nothing to the running sums) */
int mst_loss_combine_v(const mst_step_metrics* metrics, mst_stream_t stream);

/* mst_adam_flat that also keeps the transposed 16-bit shadow of up to two matrices current (the piano-roll embedding tables,
 * which the step's first launch reads): for an element of matrix j — emb[4j..4j+3] = {source offset in the flat bucket, offset
 * in wt16, rows, cols}, host array — the new weight is also stored at wt16[dst + c * roundup8(rows) + r]. `base` is the flat
 * offset of `w` itself (a launch over a sub-range of the bucket). Everything else as mst_adam_flat with advance_step = 0. */
int mst_adam_flat_emb(int dtype, int64_t n, float* w, const float* grad, float* m, float* v, void* w16, double lr, double beta1,
                      double beta2, float eps, float wd, float rescale, float clip, int32_t* step_state,
                      const mst_step_metrics* metrics, int64_t base, const int64_t* emb, int64_t n_emb, void* wt16, mst_stream_t stream);

/* 16-bit shadow + transposed shadow refresh for a list of matrices.
 * desc: int64 [n_mat, 4] on device = {src_offset, dst_offset, rows, cols}; dst is [cols, ld_t] with
 * ld_t = roundup8(rows), pad columns zeroed. tiles: int64 prefix sums [n_mat+1] of 32x32 tile counts. */
int mst_transpose_shadows(int dtype, const float* w, void* wt16, const int64_t* desc,
                          const int64_t* tile_prefix, int64_t n_mat, int64_t total_tiles,
                          mst_stream_t stream);

/* out[i] = sum of squares of x[ranges[2i] .. ranges[2i+1]) (device int64 [n_segments, 2]): the per-parameter gradient
 * norms of the reference's periodic gradient log (trainer.py:257-270) in one launch over the flat bucket */
int mst_segment_sumsq(const float* x, const int64_t* ranges, int64_t n_segments, float* out, mst_stream_t stream);

/* fp32 -> act dtype cast of a flat range (initial shadow fill) */
int mst_cast_f32_to_act(int dtype, int64_t n, const float* src, void* dst, mst_stream_t stream);

/* counter-based dropout keep-mask materialisation (for the oracle comparison): keep[i] in {0,1} */
int mst_dropout_mask(int64_t n, float p, uint64_t seed, uint32_t site, uint8_t* keep, mst_stream_t stream);

/* stream-ordered memset to zero (gradient bucket, metric sums) */
int mst_zero(void* ptr, int64_t bytes, mst_stream_t stream);
/* device-resident per-step RNG seed: state = uint64[4] {seed for this step, step counter, base seed, 0};
 * one launch per step advances it, so dropout masks and eps differ on every replay of a captured graph
 * (word 3 is the workgroup arrival counter of mst_step_begin and is zero between launches) */
int mst_rng_advance(uint64_t* state, mst_stream_t stream);
/* Top-of-step bookkeeping in ONE launch (every kernel in the captured graph costs ~4.7 us): mst_rng_advance, Adam's
 * step counter / bias-corrected lr (then call mst_adam_flat with advance_step = 0), mst_randn into eps_out, and the two
 * mst_mask_from_lengths masks (model.py:246-247), and two optional buffers to clear (16-byte aligned, sizes multiples
 * of 16: the per-sample loss sums and the gradient bucket, instead of two memset nodes). Any pointer may be NULL to
 * skip that part. rng_state is the uint64[4] state of mst_rng_advance.
 * eps_index0 (even): eps_out[i] is draw number eps_index0 + i of the step's stream, so a data-parallel rank that
 * passes its first global sample index times Z draws exactly what a single process draws for those samples. */
int mst_step_begin(uint64_t* rng_state, int32_t* adam_state, double lr, double beta1, double beta2,
                   float* eps_out, int64_t n_eps, uint32_t eps_site, int64_t eps_index0,
                   const int32_t* lens, int64_t B, uint8_t* mask_e, int64_t Se, int32_t add_e,
                   uint8_t* mask_d, int64_t Sd, int32_t add_d,
                   void* zero_a, int64_t zero_a_bytes, void* zero_b, int64_t zero_b_bytes, mst_stream_t stream);
/* the same, arguments in a struct */
typedef struct mst_step_begin_args {
  uint64_t* rng_state; int32_t* adam_state; double lr, beta1, beta2;
  float* eps_out; int64_t n_eps; uint32_t eps_site; int64_t eps_index0;
  const int32_t* lens; int64_t B; uint8_t* mask_e; int64_t Se; int32_t add_e; uint8_t* mask_d; int64_t Sd; int32_t add_d;
  void* zero_a; int64_t zero_a_bytes; void* zero_b; int64_t zero_b_bytes;
  /* Optional transposed-shadow refresh riding on the same launch (mst_transpose_shadows' arguments; sh_w == NULL: none): the
   * 16-bit transposed copies of matrices that only the BACKWARD pass reads can be rebuilt from the fp32 weights at the start
   * of the next step instead of in a launch of their own behind the optimizer. Matrices the launch itself reads (the
   * piano-roll embedding tables of mst_gemm_nt_pair_begin) must not be listed: mst_adam_flat_emb keeps those current. */
  int32_t sh_dtype; const float* sh_w; void* sh_wt16; const int64_t* sh_desc; const int64_t* sh_prefix; int64_t sh_n_mat, sh_tiles;
} mst_step_begin_args;
int mst_step_begin_v(const mst_step_begin_args* args, mst_stream_t stream);
/* mst_step_begin and mst_gemm_nt_pair in ONE launch: nothing in the piano-roll ends' embedding GEMMs (the first arithmetic of the
 * step) reads what the bookkeeping writes, so its workgroups ride in front of the tiles of that launch. Where mst_gemm_nt_pair
 * would fall back to two launches this is exactly mst_step_begin_v(begin) followed by mst_gemm_nt_pair(args0, args1). */
int mst_gemm_nt_pair_begin(const mst_gemm_args* args0, const mst_gemm_args* args1, const mst_step_begin_args* begin,
                           mst_stream_t stream);
/* eps ~ N(0,1) (replaces mx.nd.random_normal, model.py:292): Box-Muller over the counter hash;
 * effective seed = seed ^ (seed_ptr ? *seed_ptr : 0) */
int mst_randn(int64_t n, float* out, uint64_t seed, const uint64_t* seed_ptr, uint32_t site, mst_stream_t stream);

/* elementwise act-dtype helpers used on gradient joins: y = a + b */
int mst_add_act(int dtype, int64_t n, const void* a, const void* b, void* y, mst_stream_t stream);

/* hardware layout self-test (MFMA fragment maps + ds_read_tr16_b64); out: int32[4] pass flags */
int mst_selftest(int32_t* out_flags_device, mst_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MST_HIP_H */

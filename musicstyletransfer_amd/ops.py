"""Thin torch-tensor front end over the C-ABI (include/mst_hip.h).

PyTorch is plumbing here: it owns device memory and streams; every function below passes raw
`data_ptr()`s, sizes and the current HIP stream to libmst_hip.so. Nothing in this module computes
on the CPU or through torch operators — if the library or a GPU is missing, calls raise.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import LnArgs, LnBwdIn, PartialSum, OuterJob, StepBeginArgs, StepMetrics, ACT_NONE, ACT_RELU, MST_BF16, MST_F16, GemmArgs, WgradArgs, call  # noqa: F401

_DT = {torch.bfloat16: MST_BF16, torch.float16: MST_F16}


def dt(x):
    try:
        return _DT[x if isinstance(x, torch.dtype) else x.dtype]
    except KeyError:
        raise TypeError(f"activation dtype must be bfloat16 or float16, got {x}") from None


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "mst ops need device tensors (there is no CPU fallback)"
    # launches go to torch's current stream of the CURRENT device: a tensor living on another card would be reached
    # across devices without peer access (a fault, not an error code), so refuse it here
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"tensor on cuda:{t.device.index} but the current device is cuda:{torch.cuda.current_device()}: "
                           "call torch.cuda.set_device() for this rank's GPU first")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def roundup(a, b):
    return (a + b - 1) // b * b


def ld(t):
    """leading dimension (elements) of a 2-D row-major view"""
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0)


# --------------------------------------------------------------------------- GEMMs
def _gemm_args(A, B, C_out, M=None, N=None, K=None, bias=None, resid=None, act=ACT_NONE, gate=None, alpha=1.0,
               rowadd=None, rowadd_period=0, grpadd=None, grp_index=None, a_remap=(0, 0, 0), c_remap=(0, 0, 0),
               dropout_p=0.0, dropout_seed=0, dropout_site=0, self_resid=False, dropout_seed_ptr=None, resid_phys=False):
    g = GemmArgs()
    g.a_u8 = 1 if A.dtype == torch.uint8 else 0  # piano-roll frames: widened to the activation type inside the kernel
    g.dtype = dt(B if g.a_u8 else A)
    g.c_f32 = 1 if C_out.dtype == torch.float32 else 0
    g.M = A.shape[0] if M is None else M
    g.N = B.shape[0] if N is None else N
    g.K = A.shape[1] if K is None else K
    g.A, g.lda = ptr(A), ld(A)
    g.B, g.ldb = ptr(B), ld(B)
    g.C, g.ldc = ptr(C_out), ld(C_out)
    g.bias = ptr(bias)
    g.resid, g.ldr = ptr(resid), (ld(resid) if resid is not None else 0)
    g.act = act
    g.gate, g.ldg = ptr(gate), (ld(gate) if gate is not None else 0)
    g.alpha = alpha
    g.rowadd, g.ldra = ptr(rowadd), (ld(rowadd) if rowadd is not None else 0)
    g.rowadd_period = rowadd_period
    g.grpadd, g.ldga = ptr(grpadd), (ld(grpadd) if grpadd is not None else 0)
    g.grp_index = ptr(grp_index)
    g.a_rows_per_group, g.a_group_stride, g.a_group_offset = a_remap
    g.c_rows_per_group, g.c_group_stride, g.c_group_offset = c_remap
    g.dropout_p, g.dropout_seed, g.dropout_site = dropout_p, dropout_seed, dropout_site
    g.self_resid = 1 if self_resid else 0
    g.resid_phys = 1 if resid_phys else 0  # the residual shares C's physical rows (follows c_remap) instead of being indexed by m
    g.dropout_seed_ptr = ptr(dropout_seed_ptr)
    return g


def gemm_nt(A, B, C_out, **kw):
    """C[M,N] = epilogue(A[M,K] @ B[N,K]^T); see mst_gemm_args in include/mst_hip.h."""
    call("mst_gemm_nt", C.byref(_gemm_args(A, B, C_out, **kw)), stream())


def gemm_nt_pair(first, second, begin=None):
    """two gemm_nt problems, each a dict(A=, B=, C_out=, **kw), in one launch where their form allows (mst_gemm_nt_pair);
    begin: keyword arguments of step_begin — the step's bookkeeping rides on the same launch (mst_gemm_nt_pair_begin)"""
    if begin is None:
        call("mst_gemm_nt_pair", C.byref(_gemm_args(**first)), C.byref(_gemm_args(**second)), stream())
    else:
        call("mst_gemm_nt_pair_begin", C.byref(_gemm_args(**first)), C.byref(_gemm_args(**second)), C.byref(_step_begin_args(**begin)),
             stream())


def can_ride(M, N, K, T=0):
    """shapes a GEMM riding on a position-0 tail launch takes (mst_row_tail_*_ride): whole 128 x 128 tiles, 64-deep K stages,
    row-remap groups (T rows) in whole tiles"""
    return M % 128 == 0 and N % 128 == 0 and K % 64 == 0 and (T == 0 or T % 128 == 0)


def can_row_tail(B, D):
    """shapes the one-launch position-0 tail of the top encoder layer exists for (mst_row_tail_fwd)"""
    return D in (128, 256) and 0 < B <= 64


def row_tail_fwd(att, resid, Wp, bp, g1, be1, W1, b1, W2, b2, g2, be2, h1, x1, a, h2, x2, mean1, rstd1, mean2, rstd2, sync, stat_stride,
                 phys_stride, eps=1e-5, dropout_p=0.0, dropout_seed_ptr=None, site0=0, status=None, rider=None, queue=None, shadows=None):
    """att / resid / h1 / x1 / a / h2 / x2: [B, width] row views (stride(0) = the row stride in elements) of the layer's
    buffers; sync: THREE zeroed int32 device words (barrier counter — 3 * D / 16 after a complete launch —, claimed XCD,
    roles handed out); status: optional sticky int32 device word the kernel ORs _lib.TAIL_* flags into when it cannot
    finish. rider: dict(A=, B=, C_out=, **gemm_nt keywords) — one GEMM nothing in the chain reads, computed by the launch's
    workgroups on the other seven XCDs (mst_row_tail_fwd_ride); queue: its tile queue, ONE zeroed int32 device word in a
    cache line of its own (not sync's); shadows: dict(w=, wt16=, desc=, prefix=, n_mat=, tiles=) — transpose_shadows' list, rebuilt by
    the riders behind the GEMM's tiles (mst_row_tail_fwd_ride_shadows; needs a rider)"""
    assert sync.numel() >= 3 and (rider is None or queue is not None) and (shadows is None or rider is not None)
    q = _lib.RowTailArgs()
    q.dtype, q.B, q.D = dt(att), att.shape[0], Wp.shape[0]
    q.att, q.rs_att, q.resid, q.rs_res = ptr(att), att.stride(0), ptr(resid), resid.stride(0)
    q.Wp, q.ldwp, q.bp, q.g1, q.be1 = ptr(Wp), ld(Wp), ptr(bp), ptr(g1), ptr(be1)
    q.W1, q.ldw1, q.b1, q.W2, q.ldw2, q.b2, q.g2, q.be2 = ptr(W1), ld(W1), ptr(b1), ptr(W2), ld(W2), ptr(b2), ptr(g2), ptr(be2)
    assert h1.stride(0) == x1.stride(0) == h2.stride(0) == x2.stride(0)
    q.h1, q.x1, q.h2, q.x2, q.rs_d, q.a, q.rs_a = ptr(h1), ptr(x1), ptr(h2), ptr(x2), h1.stride(0), ptr(a), a.stride(0)
    q.mean1, q.rstd1, q.mean2, q.rstd2, q.stat_stride = ptr(mean1), ptr(rstd1), ptr(mean2), ptr(rstd2), stat_stride
    q.eps, q.dropout_p, q.dropout_seed, q.dropout_seed_ptr, q.site0 = eps, dropout_p, 0, ptr(dropout_seed_ptr), site0
    q.phys_stride, q.sync, q.status = phys_stride, ptr(sync), ptr(status)
    if shadows is not None:
        call("mst_row_tail_fwd_ride_shadows", C.byref(q), C.byref(_gemm_args(**rider)), ptr(queue), dt(shadows["wt16"]), ptr(shadows["w"]),
             ptr(shadows["wt16"]), ptr(shadows["desc"]), ptr(shadows["prefix"]), shadows["n_mat"], shadows["tiles"], stream())
    elif rider is not None:
        call("mst_row_tail_fwd_ride", C.byref(q), C.byref(_gemm_args(**rider)), ptr(queue), stream())
    else:
        call("mst_row_tail_fwd", C.byref(q), stream())


def row_tail_bwd(dy, h2, h1, a, mean1, rstd1, mean2, rstd2, g1, g2, W2t, W1t, Wpt, dh, dhm, dx1, dh1m, dpre, dh1, datt, dg1, db1, dg2, db2,
                 sync, stat_stride, phys_stride, dropout_p=0.0, dropout_seed_ptr=None, site0=0, status=None, rider=None, queue=None):
    """backward of row_tail_fwd's chain in one launch (mst_row_tail_bwd); dy / h2 / h1 / a / dh1 / datt: [B, width] row views,
    dh / dhm / dx1 / dh1m / dpre: compact [B, width] scratch; sync: THREE zeroed int32 device words (the barrier counter
    ends at 2 * D / 16); status, rider, queue: as row_tail_fwd"""
    assert sync.numel() >= 3 and (rider is None or queue is not None)
    q = _lib.RowTailBwdArgs()
    q.dtype, q.B, q.D = dt(dy), dy.shape[0], Wpt.shape[0]
    q.dy, q.rs_dy = ptr(dy), dy.stride(0)
    assert h2.stride(0) == h1.stride(0)
    q.h2, q.h1, q.rs_d, q.a, q.rs_a = ptr(h2), ptr(h1), h2.stride(0), ptr(a), a.stride(0)
    q.mean1, q.rstd1, q.mean2, q.rstd2, q.stat_stride = ptr(mean1), ptr(rstd1), ptr(mean2), ptr(rstd2), stat_stride
    q.g1, q.g2 = ptr(g1), ptr(g2)
    q.W2t, q.ldw2t, q.W1t, q.ldw1t, q.Wpt, q.ldwpt = ptr(W2t), ld(W2t), ptr(W1t), ld(W1t), ptr(Wpt), ld(Wpt)
    assert dh.stride(0) == dhm.stride(0) == dx1.stride(0) == dh1m.stride(0)
    q.dh, q.dhm, q.dx1, q.dh1m, q.rs_c = ptr(dh), ptr(dhm), ptr(dx1), ptr(dh1m), dh.stride(0)
    q.dpre, q.rs_dpre, q.dh1, q.rs_dh1, q.datt, q.rs_datt = ptr(dpre), dpre.stride(0), ptr(dh1), dh1.stride(0), ptr(datt), datt.stride(0)
    q.dg1, q.db1, q.dg2, q.db2 = ptr(dg1), ptr(db1), ptr(dg2), ptr(db2)
    q.dropout_p, q.dropout_seed, q.dropout_seed_ptr, q.site0 = dropout_p, 0, ptr(dropout_seed_ptr), site0
    q.phys_stride, q.sync, q.status = phys_stride, ptr(sync), ptr(status)
    if rider is not None:
        call("mst_row_tail_bwd_ride", C.byref(q), C.byref(_gemm_args(**rider)), ptr(queue), stream())
    else:
        call("mst_row_tail_bwd", C.byref(q), stream())


def can_fuse_bce(P, T, downweight=False):
    """shapes the output-layer GEMM + BCE launch exists for (mst_gemm_sigmoid_bce): a row of 128 pitches, or of any multiple of
    256 (column tiles; the label down-weighting, which counts a sample's positives in every workgroup, only on rows of one tile)"""
    return (P == 128 or (P % 256 == 0 and (P == 256 or not downweight))) and T % 64 == 0


def bce_fusion_pays(P):
    """Where the engine routes the output layer through the fused launch. Measured at configs[2] (2048 pitches, B 64, T 256): the
    fused launch 110 us against mst_gemm_nt 37 us + mst_sigmoid_bce 59 us — eight column tiles per row block, each a
    one-workgroup-per-CU launch slot whose epilogue is the loss arithmetic with the matrix pipe idle, lose more than the 134 MB of
    logits written and read back cost; at one tile per row (128 / 256 pitches) the fused launch also carries the backward pass's
    first GEMM (configs[1]: 24 us for all three)."""
    return P <= 256


def _gemm_bce_args(A, B, labels, loss, T, dlogits=None, probs=None, logits=None, label_smoothing=0.0, downweight=False, gscale=1.0, **kw):
    g = _gemm_args(A, B, dlogits if dlogits is not None else A, N=B.shape[0], **kw)
    if dlogits is None:
        g.C, g.ldc = None, 0
    q = _lib.BceArgs()
    q.labels, q.T, q.label_smoothing, q.downweight, q.loss = ptr(labels), T, label_smoothing, 1 if downweight else 0, ptr(loss)
    q.probs, q.ldp = ptr(probs), (ld(probs) if probs is not None else 0)
    q.logits, q.ldl = ptr(logits), (ld(logits) if logits is not None else 0)
    q.gscale = gscale
    return g, q


def gemm_sigmoid_bce(A, B, labels, loss, T, dgrad=None, **kw):
    """loss[b] += BCE(sigmoid(A @ B^T + bias), labels) and dlogits = its gradient, in one launch; **kw: dlogits, probs, logits,
    label_smoothing, downweight, gscale and the GEMM's M, K, bias, a_remap.
    dgrad: keyword arguments of gemm_nt_ln_bwd whose A operand is that dlogits — the output layer's input gradient + LayerNorm
    backward ride on the same launch where the shapes allow (mst_gemm_sigmoid_bce_dgrad_ln)"""
    g, q = _gemm_bce_args(A, B, labels, loss, T, **kw)
    if dgrad is None:
        call("mst_gemm_sigmoid_bce", C.byref(g), C.byref(q), stream())
    else:
        g2, l = _gemm_ln_bwd_args(**dgrad)
        call("mst_gemm_sigmoid_bce_dgrad_ln", C.byref(g), C.byref(q), C.byref(g2), C.byref(l), stream())


def can_fuse_ln(D):
    """row widths the LayerNorm-fused GEMM exists for (mst_gemm_nt_ln)"""
    return D in (128, 256)


def ln_bwd_fusion_pays(D):
    """Where the engine routes Dense-dgrad + LayerNorm-backward through the fused launch. Measured at configs[1] (step
    level, ms per step): no fusion 0.953, width 128 only 0.945, widths 128 and 256 0.951 — at width 256 the 64 x 256
    tile leaves one 8-wave workgroup per CU and the GEMM part loses what the saved LayerNorm launch gains (still so
    once the parameter gradients left the atomics: 0.917 without, 0.922 with the width-256 fusion)."""
    return D == 128


def gemm_nt_ln_fwd(A, B, H_out, gamma, beta, Y_out, mean, rstd, eps=1e-5, **kw):
    """H_out = epilogue(A @ B^T) and Y_out = LayerNorm(H_out) in one launch (mst_gemm_nt_ln, mode 1); the statistics
    are indexed by H_out's physical row"""
    g = _gemm_args(A, B, H_out, **kw)
    l = LnArgs()
    l.mode, l.gamma, l.beta, l.eps = 1, ptr(gamma), ptr(beta), eps
    l.out, l.ld_out = ptr(Y_out), ld(Y_out)
    l.mean, l.rstd = ptr(mean), ptr(rstd)
    call("mst_gemm_nt_ln", C.byref(g), C.byref(l), stream())


def can_fuse_ffn(D, F):
    """shapes the one-launch feed-forward block exists for (mst_ffn_ln_fwd)"""
    return D in (128, 256) and F % D == 0


def ffn_fusion_pays(D, F):
    """where the engine routes the feed-forward block through the one-launch form: every shape it exists for (configs[1],
    ms per step in one gpurun call: 0.875 unfused, 0.869 width 256 only, 0.864 both widths)"""
    return can_fuse_ffn(D, F)


def _row_groups(n_rows, row_groups):
    """(M, remap) of a block that works on rows [offset, offset + rows_per_group) of every `stride` physical rows"""
    rpg, stride, off = row_groups
    assert n_rows % stride == 0 and rpg % 64 == 0 and off + rpg <= stride
    return (n_rows // stride) * rpg, (rpg, stride, off)


def ffn_ln_fwd(x, W1, a_out, W2, h_out, gamma, beta, y_out, mean, rstd, eps=1e-5, ff1=None, ff2=None, proj=None, row_groups=None):
    """a_out = epilogue1(x @ W1^T), h_out = epilogue2(a_out @ W2^T), y_out = LayerNorm(h_out) in one launch
    (mst_ffn_ln_fwd); ff1 / ff2: the keyword arguments gemm_nt would get for the two GEMMs.
    proj: dict(att, W, h1, gamma, beta, mean, rstd, **gemm_nt keywords) -> the attention output projection and its
    LayerNorm run in front, in the same launch (mst_proj_ffn_ln_fwd): h1 = epilogue(att @ W^T), x = LayerNorm(h1) (x is then
    an OUTPUT as well as the block's input)
    row_groups = (rows_per_group, stride, offset): the launch works on those rows of every operand only (the last decoder
    layer without each sample's position-0 row); the other rows are neither read nor written"""
    ff1, ff2 = dict(ff1 or {}), dict(ff2 or {})
    if row_groups is not None:
        M, remap = _row_groups(x.shape[0], row_groups)
        ff1.update(M=M, a_remap=remap)
        ff2.update(M=M)
        if proj is not None:
            proj = dict(proj, M=M)
    g1 = _gemm_args(x, W1, a_out, **ff1)
    g2 = _gemm_args(a_out, W2, h_out, **ff2)
    l = LnArgs()
    l.mode, l.gamma, l.beta, l.eps = 1, ptr(gamma), ptr(beta), eps
    l.out, l.ld_out = ptr(y_out), ld(y_out)
    l.mean, l.rstd = ptr(mean), ptr(rstd)
    if proj is None:
        call("mst_ffn_ln_fwd", C.byref(g1), C.byref(g2), C.byref(l), stream())
        return
    kw = {k: v for k, v in proj.items() if k not in ("att", "W", "h1", "gamma", "beta", "mean", "rstd")}
    g0 = _gemm_args(proj["att"], proj["W"], proj["h1"], **kw)
    l0 = LnArgs()
    l0.mode, l0.gamma, l0.beta, l0.eps = 1, ptr(proj["gamma"]), ptr(proj["beta"]), eps
    l0.out, l0.ld_out = ptr(x), ld(x)
    l0.mean, l0.rstd = ptr(proj["mean"]), ptr(proj["rstd"])
    call("mst_proj_ffn_ln_fwd", C.byref(g0), C.byref(l0), C.byref(g1), C.byref(g2), C.byref(l), stream())


def ffn_ln_bwd(dff, W2t, dpre_out, gate, W1t, dx_out, x, gamma, mean, rstd, dgamma, dbeta, alpha=1.0, dx_masked=None, mask_mode=0,
               partials=None, lead=None, row_groups=None, **kw):
    """dpre_out = ((dff @ W2t^T) * alpha) gated by gate > 0; dx_out = LayerNorm-backward(dpre_out @ W1t^T + resid; x, mean, rstd,
    gamma) in one launch (mst_ffn_ln_bwd). W2t / W1t: the transposed 16-bit weights ([F, D] and [D, F]); **kw: resid and the
    dropout fields of the LayerNorm-backward mask, as for gemm_nt_ln_bwd.
    lead: dict(dy, x, gamma, mean, rstd, dx, [dx_masked, dropout_*], [dgamma, dbeta | partials]) -> the layer's leading
    LayerNorm backward runs in the prologue (mst_ffn_ln_bwd_lead); dff must then be lead's dx_masked (or dx)."""
    rg = {}
    if row_groups is not None:  # (as in ffn_ln_fwd: the same rows of every operand)
        M, remap = _row_groups(dff.shape[0], row_groups)
        rg, kw = dict(M=M, a_remap=remap), dict(kw, M=M)
    g1 = _gemm_args(dff, W2t, dpre_out, gate=gate, alpha=alpha, **rg)
    g2 = _gemm_args(dpre_out, W1t, dx_out, **kw)
    l = LnArgs()
    l.mode, l.gamma = 2, ptr(gamma)
    l.mean, l.rstd = ptr(mean), ptr(rstd)
    l.x, l.ld_x = ptr(x), ld(x)
    l.dgamma, l.dbeta = ptr(dgamma), ptr(dbeta)
    l.out, l.ld_out = ptr(dx_masked), (ld(dx_masked) if dx_masked is not None else 0)
    l.mask_mode = mask_mode
    l.partials = ptr(partials)
    if partials is not None:
        assert partials.shape[0] >= gemm_nt_ln_parts(g1.M) and partials.shape[1] == 2 * g2.N and partials.is_contiguous()
    if lead is None:
        call("mst_ffn_ln_bwd", C.byref(g1), C.byref(g2), C.byref(l), stream())
        return
    q = LnBwdIn()
    q.dy, q.ld_dy = ptr(lead["dy"]), ld(lead["dy"])
    q.x, q.ld_x = ptr(lead["x"]), ld(lead["x"])
    q.gamma, q.mean, q.rstd = ptr(lead["gamma"]), ptr(lead["mean"]), ptr(lead["rstd"])
    q.dx, q.ld_dx = ptr(lead["dx"]), ld(lead["dx"])
    dxm = lead.get("dx_masked")
    q.dx_masked, q.ld_dxm = ptr(dxm), (ld(dxm) if dxm is not None else 0)
    q.dgamma, q.dbeta, q.partials = ptr(lead.get("dgamma")), ptr(lead.get("dbeta")), ptr(lead.get("partials"))
    q.mask_mode = 1 if dxm is not None else 0
    q.dropout_p, q.dropout_seed = lead.get("dropout_p", 0.0), lead.get("dropout_seed", 0)
    q.dropout_seed_ptr, q.dropout_site = ptr(lead.get("dropout_seed_ptr")), lead.get("dropout_site", 0)
    if lead.get("partials") is not None:
        assert lead["partials"].shape[0] >= gemm_nt_ln_parts(g1.M) and lead["partials"].shape[1] == 2 * g2.N
    call("mst_ffn_ln_bwd_lead", C.byref(q), C.byref(g1), C.byref(g2), C.byref(l), stream())


def gemm_nt_ln_bwd(A, B, dX_out, x, gamma, mean, rstd, dgamma, dbeta, dx_masked=None, mask_mode=0, partials=None, **kw):
    """dX_out = LayerNorm-backward(epilogue(A @ B^T); x, mean, rstd, gamma) in one launch (mst_gemm_nt_ln, mode 2); x, mean,
    rstd are indexed by dX_out's physical row, dx_masked by the logical row; the dropout fields among **kw are those of
    the LayerNorm-backward mask. partials: [gemm_nt_ln_parts(M), 2N] fp32 -> per-workgroup column sums are stored there
    instead of being added to dgamma / dbeta (finish with partial_sums)"""
    g, l = _gemm_ln_bwd_args(A, B, dX_out, x, gamma, mean, rstd, dgamma, dbeta, dx_masked, mask_mode, partials, **kw)
    call("mst_gemm_nt_ln", C.byref(g), C.byref(l), stream())


def _gemm_ln_bwd_args(A, B, dX_out, x, gamma, mean, rstd, dgamma, dbeta, dx_masked=None, mask_mode=0, partials=None, **kw):
    g = _gemm_args(A, B, dX_out, **kw)
    l = LnArgs()
    l.mode, l.gamma = 2, ptr(gamma)
    l.mean, l.rstd = ptr(mean), ptr(rstd)
    l.x, l.ld_x = ptr(x), ld(x)
    l.dgamma, l.dbeta = ptr(dgamma), ptr(dbeta)
    l.out, l.ld_out = ptr(dx_masked), (ld(dx_masked) if dx_masked is not None else 0)
    l.mask_mode = mask_mode
    l.partials = ptr(partials)
    if partials is not None:
        assert partials.shape[0] >= gemm_nt_ln_parts(g.M) and partials.shape[1] == 2 * g.N and partials.is_contiguous()
    return g, l


def gemm_nt_ln_parts(M):
    return int(_lib.load().mst_gemm_nt_ln_parts(M))


def layernorm_bwd_parts(M, D):
    return int(_lib.load().mst_layernorm_bwd_parts(M, D))


PARTIAL_SUM_MAX_JOBS = 20


def partial_sum_job(src, n_parts, dst, scale=1.0, col_off=0, length=None):
    """dst[0:length] += scale * sum_p src[p, col_off:col_off+length] (src: [>= n_parts, stride] fp32, contiguous rows)"""
    j = PartialSum()
    length = dst.numel() if length is None else length
    assert src.dtype == dst.dtype and src.element_size() == 4 and src.shape[0] >= n_parts
    j.src = src.data_ptr() + 4 * col_off
    j.n_parts, j.stride, j.len = n_parts, src.stride(0), length
    j.dst, j.scale = ptr(dst), scale
    return j


def partial_sums(jobs):
    """mst_partial_sums: every job's parts added in index order, 20 jobs per launch"""
    for i in range(0, len(jobs), PARTIAL_SUM_MAX_JOBS):
        chunk = jobs[i:i + PARTIAL_SUM_MAX_JOBS]
        call("mst_partial_sums", (PartialSum * len(chunk))(*chunk), len(chunk), stream())


def wgrad_problem(A, B, dW, db=None, M=None, N=None, K=None, scale=1.0, a_remap=(0, 0, 0), b_remap=(0, 0, 0)):
    w = WgradArgs()
    w.a_u8 = 1 if A.dtype == torch.uint8 else 0
    w.dtype = dt(B if w.a_u8 else A)
    w.M = A.shape[0] if M is None else M
    w.N = dW.shape[0] if N is None else N
    w.K = dW.shape[1] if K is None else K
    w.A, w.lda = ptr(A), ld(A)
    w.B, w.ldb = ptr(B), ld(B)
    w.dW, w.ldw = ptr(dW), ld(dW)
    w.db = ptr(db)
    w.scale = scale
    w.a_rows_per_group, w.a_group_stride, w.a_group_offset = a_remap
    w.b_rows_per_group, w.b_group_stride, w.b_group_offset = b_remap
    return w


WGRAD_MAX_PROBLEMS = 16
OUTER_MAX_JOBS = 2


def outer_job(L, R, out, obias=None):
    """out[J, I] += L[B, J]^T R[B, I], obias[J] += sum_b L[b, :] (mst_outer_job): L fp32 contiguous, R a [B, I] row view (fp32 or
    the activation type) with any row stride, out / obias fp32 contiguous"""
    q = OuterJob()
    assert L.dtype == torch.float32 and L.is_contiguous() and out.dtype == torch.float32 and out.is_contiguous() and R.stride(1) == 1
    q.L, q.R, q.r_stride = ptr(L), ptr(R), R.stride(0)
    q.r_dtype = 2 if R.dtype == torch.float32 else dt(R)
    q.B, q.J, q.I = L.shape[0], L.shape[1], R.shape[1]
    assert out.numel() == q.J * q.I and R.shape[0] == q.B and (obias is None or obias.numel() == q.J)
    q.out, q.obias = ptr(out), ptr(obias)
    return q


def outer_jobs(jobs):
    for i in range(0, len(jobs), OUTER_MAX_JOBS):
        chunk = jobs[i:i + OUTER_MAX_JOBS]
        call("mst_outer_jobs", (OuterJob * len(chunk))(*chunk), len(chunk), stream())


def gemm_wgrad_batch(problems, scratch=None, sums=None, outers=None):
    """dW_i[N,K] += A_i^T @ B_i for a list of problems, 16 per launch. scratch: optional fp32 work buffer for the
    atomic-free two-pass reduction of big batches (mst_gemm_wgrad_batch_ws). sums: column-sum jobs (partial_sum_job),
    outers: batch outer products (outer_job), to execute along with the weight gradients (mst_gemm_wgrad_batch_flush)."""
    sums, outers = list(sums or []), list(outers or [])
    for i in range(0, len(problems), WGRAD_MAX_PROBLEMS):
        chunk = problems[i:i + WGRAD_MAX_PROBLEMS]
        arr = (WgradArgs * len(chunk))(*chunk)
        if (sums or outers) and i + WGRAD_MAX_PROBLEMS >= len(problems):  # the last launch takes (the first 20 of) the sums along
            take, sums = sums[:PARTIAL_SUM_MAX_JOBS], sums[PARTIAL_SUM_MAX_JOBS:]
            tko, outers = outers[:OUTER_MAX_JOBS], outers[OUTER_MAX_JOBS:]
            call("mst_gemm_wgrad_batch_flush", arr, len(chunk), ptr(scratch),
                 (scratch.numel() * scratch.element_size() if scratch is not None else 0),
                 (PartialSum * len(take))(*take) if take else None, len(take),
                 (OuterJob * len(tko))(*tko) if tko else None, len(tko), stream())
        elif scratch is None:
            call("mst_gemm_wgrad_batch", arr, len(chunk), stream())
        else:
            call("mst_gemm_wgrad_batch_ws", arr, len(chunk), ptr(scratch), scratch.numel() * scratch.element_size(), stream())
    if sums:
        partial_sums(sums)
    if outers:
        outer_jobs(outers)


def gemm_wgrad(A, B, dW, db=None, **kw):
    gemm_wgrad_batch([wgrad_problem(A, B, dW, db, **kw)])


# --------------------------------------------------------------------------- attention
def attn_fwd(qkv, keymask, lse, out, B, S, H, dh, k_off, q_off, v_off, q_limit=0):
    call("mst_attn_keysoftmax_fwd", dt(qkv), B, S, H, dh, ptr(qkv), ld(qkv), k_off, q_off, v_off, ptr(keymask),
         ptr(lse), ptr(out), ld(out), q_limit, stream())


def attn_qkv_fwd(x, W, bias, qkv, keymask, lse, out, B, S, H, dh, k_off, q_off, v_off, q_limit=0):
    """qkv = x W^T + bias (written: the backward pass reads it) and attn_fwd on it, the projection inside the attention launch
    where the shape allows (mst_attn_qkv_fwd)"""
    call("mst_attn_qkv_fwd", dt(qkv), B, S, H, dh, ptr(x), ld(x), ptr(W), ld(W), ptr(bias), ptr(qkv), ld(qkv), k_off, q_off, v_off,
         ptr(keymask), ptr(lse), ptr(out), ld(out), q_limit, stream())


def attn_bwd(qkv, keymask, lse, dout, dqkv, delta, B, S, H, dh, k_off, q_off, v_off, q_limit=0):
    """q_limit > 0: rows [q_limit, S) of every sample of dout are zero (see the header)"""
    call("mst_attn_keysoftmax_bwd", dt(qkv), B, S, H, dh, ptr(qkv), ld(qkv), k_off, q_off, v_off, ptr(keymask),
         ptr(lse), ptr(dout), ld(dout), ptr(dqkv), ld(dqkv), ptr(delta), q_limit, stream())


def attn_decode(cache3, n_keys, H, dh, k_off, q_off, v_off, out, mode=0):
    """cache3: [B, t_max, ld] K|Q|V rows fed so far (row n_keys - 1 = the new position); out [B, ld_out]"""
    B, t_max = cache3.shape[0], cache3.shape[1]
    assert cache3.stride(2) == 1 and cache3.stride(0) == t_max * cache3.stride(1)
    call("mst_attn_decode", dt(cache3), B, H, dh, n_keys, t_max, ptr(cache3), cache3.stride(1), k_off, q_off, v_off, mode, ptr(out),
         ld(out), stream())


# --------------------------------------------------------------------------- LayerNorm
def beam_step(probs, scores_in, scores_out, seqs_in, seqs_out, hyp_src, word, i, K, eos, pad, active=None):
    """one position of beam search (mst_beam_step): probs fp32 [B*K, >= V]; seqs int32 [B*K, L]; word int32 [B*K(, 1)]"""
    N, V = probs.shape[0], probs.shape[1]
    call("mst_beam_step", N // K, K, V, i, seqs_in.shape[1], ptr(probs), probs.stride(0), ptr(scores_in), ptr(scores_out), ptr(seqs_in),
         ptr(seqs_out), ptr(hyp_src), ptr(word), ptr(active), eos, pad, stream())


def sample_step(probs, seqs, scores, word, i, seed, eos, pad, active=None):
    """ancestral sampling of position i for every sequence (mst_sample_step)"""
    call("mst_sample_step", probs.shape[0], probs.shape[1], i, seqs.shape[1], ptr(probs), probs.stride(0), ptr(seqs), ptr(scores), ptr(word),
         ptr(active), seed, eos, pad, stream())


def beam_gather(cache_in, cache_out, src, n_rows, skip_cols=None):
    """cache_out[j, :n_rows] = cache_in[src[j], :n_rows] for [N, t_max, width] caches (mst_beam_gather);
    skip_cols = (first, count): those columns of every row are left alone (mst_beam_gather_cols)"""
    N, t_max, width = cache_in.shape
    es = cache_in.element_size()
    if skip_cols is None:
        call("mst_beam_gather", ptr(cache_in), ptr(cache_out), ptr(src), N, n_rows, width * es, t_max, stream())
    else:
        call("mst_beam_gather_cols", ptr(cache_in), ptr(cache_out), ptr(src), N, n_rows, width * es, t_max, skip_cols[0] * es,
             skip_cols[1] * es, stream())


def layernorm_fwd(x, gamma, beta, y, mean, rstd, D=None, eps=1e-5, M=None, row_id_stride=1):
    M = x.shape[0] if M is None else M
    D = x.shape[1] if D is None else D
    call("mst_layernorm_fwd", dt(x), M, D, ptr(x), ld(x), ptr(gamma), ptr(beta), eps, ptr(y), ld(y), ptr(mean),
         ptr(rstd), row_id_stride, stream())


def layernorm_bwd(x, gamma, mean, rstd, dy, dx, dgamma, dbeta, D=None, dx_masked=None, mask_mode=0, dropout_p=0.0,
                  dropout_seed=0, dropout_site=0, dropout_seed_ptr=None, M=None, row_id_stride=1, partials=None):
    """partials: [layernorm_bwd_parts(M, D), 2D] fp32 -> per-workgroup column sums instead of atomics on dgamma / dbeta"""
    M = x.shape[0] if M is None else M
    D = x.shape[1] if D is None else D
    if partials is not None:
        assert partials.shape[0] >= layernorm_bwd_parts(M, D) and partials.shape[1] == 2 * D and partials.is_contiguous()
    call("mst_layernorm_bwd", dt(x), M, D, ptr(x), ld(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dy), ld(dy), ptr(dx),
         ld(dx), ptr(dx_masked), (ld(dx_masked) if dx_masked is not None else 0), ptr(dgamma), ptr(dbeta), mask_mode,
         dropout_p, dropout_seed, dropout_site, ptr(dropout_seed_ptr), row_id_stride, ptr(partials), stream())


# --------------------------------------------------------------------------- embedding / masks
def embed_fwd(tokens, table, pos, out3, s_off, alpha, classes=None, cls_table=None, keymask=None):
    """out3: [B, S_out, ld] activation buffer; tokens int32 [B, T]."""
    B, T = tokens.shape
    D = table.shape[1]
    call("mst_embed_fwd", dt(out3), B, T, D, ptr(tokens), ptr(table), table.stride(0), ptr(classes), ptr(cls_table),
         (cls_table.stride(0) if cls_table is not None else 0), ptr(pos), pos.stride(0), alpha, ptr(out3),
         out3.stride(1), out3.shape[1], s_off, ptr(keymask), stream())


def embed_bwd(tokens, dtable, dX3, s_off, alpha, classes=None, dcls=None):
    B, T = tokens.shape
    D = dtable.shape[1]
    call("mst_embed_bwd", dt(dX3), B, T, D, ptr(tokens), ptr(dtable), dtable.stride(0), ptr(classes), ptr(dcls),
         (dcls.stride(0) if dcls is not None else 0), alpha, ptr(dX3), dX3.stride(1), dX3.shape[1], s_off, stream())


def group_colsum(X3, T, D, s_off, idx, dst, alpha):
    B = X3.shape[0]
    call("mst_group_colsum", dt(X3), B, T, D, ptr(X3), X3.stride(1), X3.shape[1], s_off, ptr(idx), ptr(dst),
         dst.stride(0), alpha, stream())


def mask_from_lengths(lens, add, keymask):
    B, S = keymask.shape
    call("mst_mask_from_lengths", B, S, ptr(lens), add, ptr(keymask), stream())


# --------------------------------------------------------------------------- latent block
def latent_fwd(enc_out3, Wl, bl, eps, Wh, bh, classes, cls_d, pos_d, alpha_d, mu, sigma, z, kl, dec_in3, proj=None):
    """proj = (Wq [nq, Dd] 16-bit, bq, qkv3 [B, S, >= nq]): the decoder's first K | Q | V projection of the row this launch
    produces (position 0) on the same launch (mst_latent_fwd_proj)"""
    B = enc_out3.shape[0]
    De, Z, Dd = Wl.shape[1], Wh.shape[1], Wh.shape[0]
    args = (dt(enc_out3), B, De, Z, Dd, ptr(enc_out3), enc_out3.stride(0), ptr(Wl), ptr(bl), ptr(eps),
            ptr(Wh), ptr(bh), ptr(classes), ptr(cls_d), cls_d.stride(0), ptr(pos_d), alpha_d, ptr(mu), ptr(sigma), ptr(z),
            ptr(kl), ptr(dec_in3), dec_in3.stride(0))
    if proj is None:
        call("mst_latent_fwd", *args, stream())
    else:
        Wq, bq, qkv3 = proj
        call("mst_latent_fwd_proj", *args, ptr(Wq), ld(Wq), ptr(bq), ptr(qkv3), qkv3.stride(0), Wq.shape[0], stream())


def latent_bwd(enc_out3, Wl, eps, Wh, classes, mu, sigma, z, d_dec_in3, alpha_d, kl_weight, gscale, dWl, dbl, dWh, dbh,
               dcls_d, d_enc_out3, scratch, enc_scale=1.0):
    B = enc_out3.shape[0]
    De, Z, Dd = Wl.shape[1], Wh.shape[1], Wh.shape[0]
    call("mst_latent_bwd", dt(enc_out3), B, De, Z, Dd, ptr(enc_out3), enc_out3.stride(0), ptr(Wl), ptr(eps), ptr(Wh),
         ptr(classes), ptr(mu), ptr(sigma), ptr(z), ptr(d_dec_in3), d_dec_in3.stride(0), alpha_d, kl_weight, gscale,
         enc_scale, ptr(dWl), ptr(dbl), ptr(dWh), ptr(dbh), ptr(dcls_d), dcls_d.stride(0), ptr(d_enc_out3), d_enc_out3.stride(0),
         ptr(scratch), stream())


def latent_bwd_vec(Wl, eps, Wh, classes, mu, sigma, d_dec_in3, alpha_d, kl_weight, gscale, dcls_d, d_enc_out3, scratch, enc_scale=1.0,
                   proj=None):
    """latent_bwd's first launch with the decoder class table's gradient folded in (mst_latent_bwd_vec); the other parameter
    gradients are latent_outer_jobs(...) of the caller's weight-gradient flush.
    proj = (dqkv3 [B, S, >= nq], Wt [Dd, nq] the transposed 16-bit weight, resid3 [B, S, Dd] or None): d(dec_in[:, 0, :]) is computed
    here from the projection's gradient at position 0 instead of being read from d_dec_in3 (mst_latent_bwd_vec_proj)"""
    B, De, Z, Dd = d_dec_in3.shape[0], Wl.shape[1], Wh.shape[1], Wh.shape[0]
    if proj is not None:
        dq3, Wt, r3 = proj
        call("mst_latent_bwd_vec_proj", dt(dq3), B, De, Z, Dd, ptr(Wl), ptr(eps), ptr(Wh), ptr(classes), ptr(mu), ptr(sigma),
             ptr(dq3), dq3.stride(0), ptr(Wt), ld(Wt), 3 * Dd, ptr(r3), (r3.stride(0) if r3 is not None else 0), alpha_d, kl_weight,
             gscale, enc_scale, ptr(dcls_d), dcls_d.stride(0), ptr(d_enc_out3), d_enc_out3.stride(0), ptr(scratch), stream())
        return
    call("mst_latent_bwd_vec", dt(d_dec_in3), B, De, Z, Dd, ptr(Wl), ptr(eps), ptr(Wh), ptr(classes), ptr(mu), ptr(sigma),
         ptr(d_dec_in3), d_dec_in3.stride(0), alpha_d, kl_weight, gscale, enc_scale, ptr(dcls_d), dcls_d.stride(0), ptr(d_enc_out3),
         d_enc_out3.stride(0), ptr(scratch), stream())


def latent_outer_jobs(scratch, enc_out3, z, dWl, dbl, dWh, dbh):
    """the two outer products latent_bwd_vec leaves to the flush: dWl += dlat^T enc_out[:, 0], dWh += t^T z (and the biases)"""
    B, Dd, Z2 = enc_out3.shape[0], dWh.shape[0], dWl.shape[0]
    t = scratch[: B * Dd].view(B, Dd)
    dlat = scratch[B * Dd: B * (Dd + Z2)].view(B, Z2)
    return [outer_job(dlat, enc_out3[:, 0, :], dWl, dbl), outer_job(t, z, dWh, dbh)]


def reparam_kl_fwd(mu, sigma, eps, z, kl):
    B, Z = mu.shape
    call("mst_reparam_kl_fwd", B, Z, ptr(mu), ptr(sigma), ptr(eps), ptr(z), ptr(kl), stream())


def reparam_kl_bwd(mu, sigma, eps, dz, kl_weight, dmu, dsigma):
    B, Z = mu.shape
    call("mst_reparam_kl_bwd", B, Z, ptr(mu), ptr(sigma), ptr(eps), ptr(dz), kl_weight, ptr(dmu), ptr(dsigma), stream())


# --------------------------------------------------------------------------- loss heads
def softmax_ce(logits, labels, loss, B, T, V, probs=None, dlogits=None, gscale=1.0, pre_zeroed=False, tok_parts=None, top_k=5):
    """tok_parts: fp32 [CE_MAX_WORKGROUPS, 4] running partial sums of the masked token metrics (see the header)"""
    if tok_parts is not None:
        assert tok_parts.dtype == torch.float32 and tok_parts.is_contiguous() and tuple(tok_parts.shape) == (_lib.CE_MAX_WORKGROUPS, 4)
    call("mst_softmax_ce", dt(logits), B, T, V, ptr(logits), ld(logits), ptr(labels), ptr(loss), ptr(probs),
         (ld(probs) if probs is not None else 0), ptr(dlogits), (ld(dlogits) if dlogits is not None else 0), gscale,
         1 if pre_zeroed else 0, ptr(tok_parts), top_k, stream())


_PROB_DT = {torch.float32: _lib.MST_F32, torch.bfloat16: MST_BF16, torch.float16: MST_F16}


def ce_from_probs(probs2, labels, loss, B, T, V):
    """SoftmaxCrossEntropy on probabilities (the reference's call form); probs2: [B*T, ld] fp32 / 16-bit"""
    call("mst_ce_from_probs", _PROB_DT[probs2.dtype], B, T, V, ptr(probs2), ld(probs2), ptr(labels), ptr(loss), stream())


def bce_from_probs(probs, labels, loss, label_smoothing=0.0, downweight=False):
    """BinaryCrossEntropy(from_sigmoid=True): probs / labels contiguous [B, ...] with the same number of elements per sample"""
    B = probs.shape[0]
    assert probs.is_contiguous() and labels.is_contiguous() and labels.dtype == torch.uint8 and labels.numel() == probs.numel()
    call("mst_bce_from_probs", _PROB_DT[probs.dtype], B, probs.numel() // B, ptr(probs), ptr(labels), label_smoothing,
         1 if downweight else 0, ptr(loss), stream())


def sigmoid_bce(logits, labels, loss, B, T, P, label_smoothing=0.0, downweight=False, npos=None, probs=None,
                dlogits=None, gscale=1.0, pre_zeroed=False):
    call("mst_sigmoid_bce", dt(logits), B, T, P, ptr(logits), ld(logits), ptr(labels), label_smoothing,
         1 if downweight else 0, ptr(npos), ptr(loss), ptr(probs), (ld(probs) if probs is not None else 0),
         ptr(dlogits), (ld(dlogits) if dlogits is not None else 0), gscale, 1 if pre_zeroed else 0, stream())


def _step_metrics(metrics):
    """dict(recon, kl, kl_weight, total, metric[, status, expect=[(int32 device word, value), ...]]) -> StepMetrics; without
    `recon` only the step guard is carried (a second optimizer range of the same step)"""
    mt = StepMetrics()
    recon = metrics.get("recon")
    mt.B, mt.recon, mt.kl = (recon.numel() if recon is not None else 0), ptr(recon), ptr(metrics.get("kl"))
    mt.kl_weight, mt.total, mt.metric = metrics.get("kl_weight", 0.0), ptr(metrics.get("total")), ptr(metrics.get("metric"))
    mt.status = ptr(metrics.get("status"))
    exp = list(metrics.get("expect") or [])
    assert len(exp) <= 2 and (not exp or metrics.get("status") is not None)
    for i, (word, val) in enumerate(exp):
        setattr(mt, f"expect_ptr{i}", ptr(word))
        setattr(mt, f"expect_val{i}", int(val))
    fin = metrics.get("finite")  # (recon, kl): the non-finite guard (needs status: three int32 words)
    if fin is not None:
        assert metrics.get("status") is not None and metrics["status"].numel() >= 3
        mt.fin_recon, mt.fin_kl, mt.fin_B = ptr(fin[0]), ptr(fin[1]), fin[0].numel()
    return mt


def loss_combine(recon, kl, kl_weight, total=None, metric_acc=None, guard=None):
    """guard: dict(status=, expect=) — the step guard of mst_step_metrics (mst_loss_combine_v)"""
    if guard:
        mt = _step_metrics(dict(recon=recon, kl=kl, kl_weight=kl_weight, total=total, metric=metric_acc, **guard))
        call("mst_loss_combine_v", C.byref(mt), stream())
        return
    call("mst_loss_combine", recon.shape[0], ptr(recon), ptr(kl), kl_weight, ptr(total), ptr(metric_acc), stream())


# --------------------------------------------------------------------------- optimizer / shadows
def adam_flat(w, grad, m, v, w16, step_state, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0, rescale=1.0, clip=-1.0,
              advance_step=True, metrics=None, emb=None):
    """metrics: dict(recon, kl, kl_weight, total, metric) -> loss_combine's bookkeeping runs in this launch;
    + status / expect: the step guard (_step_metrics)"""
    mt = _step_metrics(metrics) if metrics is not None else None
    if emb is not None:  # (the optimizer keeps the transposed shadows of these matrices current: mst_adam_flat_emb)
        assert not advance_step
        flat = [int(x) for spec in emb["specs"] for x in spec]
        call("mst_adam_flat_emb", dt(w16), w.numel(), ptr(w), ptr(grad), ptr(m), ptr(v), ptr(w16), lr, beta1, beta2, eps, wd, rescale, clip,
             ptr(step_state), C.byref(mt) if mt is not None else None, emb.get("base", 0), (_lib.c_i64 * len(flat))(*flat), len(emb["specs"]),
             ptr(emb["wt16"]), stream())
        return
    call("mst_adam_flat", dt(w16), w.numel(), ptr(w), ptr(grad), ptr(m), ptr(v), ptr(w16), lr, beta1, beta2, eps, wd,
         rescale, clip, ptr(step_state), 1 if advance_step else 0, C.byref(mt) if mt is not None else None, stream())


def transpose_shadows(w, wt16, desc, tile_prefix, n_mat, total_tiles):
    call("mst_transpose_shadows", dt(wt16), ptr(w), ptr(wt16), ptr(desc), ptr(tile_prefix), n_mat, total_tiles, stream())


def segment_sumsq(x, ranges, out):
    """out[i] = sum of squares of x[ranges[i,0]:ranges[i,1]]; ranges: device int64 [n, 2]"""
    call("mst_segment_sumsq", ptr(x), ptr(ranges), ranges.shape[0], ptr(out), stream())


def cast_to_act(src, dst):
    call("mst_cast_f32_to_act", dt(dst), src.numel(), ptr(src), ptr(dst), stream())


def add_act(a, b, y):
    call("mst_add_act", dt(a), a.numel(), ptr(a), ptr(b), ptr(y), stream())


def dropout_mask(n, p, seed, site, keep):
    call("mst_dropout_mask", n, p, seed, site, ptr(keep), stream())


def zero(t):
    call("mst_zero", ptr(t), t.numel() * t.element_size(), stream())


def rng_advance(state):
    call("mst_rng_advance", ptr(state), stream())


def randn(out, seed=0, seed_ptr=None, site=0):
    call("mst_randn", out.numel(), ptr(out), seed, ptr(seed_ptr), site, stream())


def _step_begin_args(rng_state=None, adam_state=None, lr=0.0, beta1=0.9, beta2=0.999, eps_out=None, eps_site=0x7FFF0000, eps_index0=0,
                     lens=None, mask_e=None, add_e=0, mask_d=None, add_d=1, zero_a=None, zero_b=None, shadows=None):
    """shadows: dict(w, wt16, desc, prefix, n_mat, tiles) — a transposed-shadow refresh (transpose_shadows' arguments) hosted by the
    launch that carries the bookkeeping"""
    q = StepBeginArgs()
    nbytes = lambda t: t.numel() * t.element_size() if t is not None else 0
    q.rng_state, q.adam_state, q.lr, q.beta1, q.beta2 = ptr(rng_state), ptr(adam_state), lr, beta1, beta2
    q.eps_out, q.n_eps, q.eps_site, q.eps_index0 = ptr(eps_out), (eps_out.numel() if eps_out is not None else 0), eps_site, eps_index0
    q.lens, q.B = ptr(lens), (lens.shape[0] if lens is not None else 0)
    q.mask_e, q.Se, q.add_e = ptr(mask_e), (mask_e.shape[1] if mask_e is not None else 0), add_e
    q.mask_d, q.Sd, q.add_d = ptr(mask_d), (mask_d.shape[1] if mask_d is not None else 0), add_d
    q.zero_a, q.zero_a_bytes, q.zero_b, q.zero_b_bytes = ptr(zero_a), nbytes(zero_a), ptr(zero_b), nbytes(zero_b)
    if shadows:
        q.sh_dtype, q.sh_w, q.sh_wt16 = dt(shadows["wt16"]), ptr(shadows["w"]), ptr(shadows["wt16"])
        q.sh_desc, q.sh_prefix, q.sh_n_mat, q.sh_tiles = ptr(shadows["desc"]), ptr(shadows["prefix"]), shadows["n_mat"], shadows["tiles"]
    return q


def step_begin(**kw):
    """mst_step_begin (keyword arguments: _step_begin_args)"""
    call("mst_step_begin_v", C.byref(_step_begin_args(**kw)), stream())


def selftest():
    flags = torch.zeros(4, dtype=torch.int32, device="cuda")
    call("mst_selftest", ptr(flags), stream())
    torch.cuda.synchronize()
    return flags.cpu().tolist()


# --------------------------------------------------------------------------- graphs / events
class Graph:
    """hipGraph captured on the current torch stream through the library's own capture helpers."""

    def __init__(self):
        self.exec = C.c_void_p()

    def capture(self, fn):
        st = stream()
        call("mst_graph_begin", st)
        try:
            fn()
        finally:
            call("mst_graph_end", st, C.byref(self.exec))
        return self

    def launch(self):
        call("mst_graph_launch", self.exec, stream())

    def __del__(self):
        try:
            if self.exec:
                _lib.load().mst_graph_destroy(self.exec)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.ev = C.c_void_p()
        call("mst_event_create", C.byref(self.ev))

    def record(self):
        call("mst_event_record", self.ev, stream())

    def sync(self):
        call("mst_event_sync", self.ev)

    def elapsed_ms(self, later):
        ms = C.c_float()
        call("mst_event_elapsed_ms", self.ev, later.ev, C.byref(ms))
        return ms.value

    def __del__(self):
        try:
            _lib.load().mst_event_destroy(self.ev)
        except Exception:
            pass


def sqrt_d(d):
    return math.sqrt(float(d))

"""ctypes binding of libmst_hip.so (include/mst_hip.h).

The product path has NO CPU fallback: if the shared object is missing, or a kernel returns a
non-zero status, an exception is raised. `load()` only dlopen()s and checks the exported symbols
(so it works on a GPU-less box for the "does it build / export" tests); anything that launches a
kernel needs a visible HIP device.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libmst_hip.so")

MST_BF16, MST_F16, MST_F32 = 0, 1, 2
ACT_NONE, ACT_RELU = 0, 1
CE_MAX_WORKGROUPS = 4096  # MST_CE_MAX_WORKGROUPS: rows of mst_softmax_ce's token-metric partials

c_i32, c_i64, c_f32, c_f64, c_u64, c_u32 = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint64, C.c_uint32
vp = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", c_i32), ("c_f32", c_i32),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("A", vp), ("lda", c_i64),
        ("B", vp), ("ldb", c_i64),
        ("C", vp), ("ldc", c_i64),
        ("bias", vp),
        ("resid", vp), ("ldr", c_i64),
        ("act", c_i32),
        ("gate", vp), ("ldg", c_i64),
        ("alpha", c_f32),
        ("rowadd", vp), ("ldra", c_i64), ("rowadd_period", c_i64),
        ("grpadd", vp), ("ldga", c_i64), ("grp_index", vp),
        ("a_rows_per_group", c_i64), ("a_group_stride", c_i64), ("a_group_offset", c_i64),
        ("c_rows_per_group", c_i64), ("c_group_stride", c_i64), ("c_group_offset", c_i64),
        ("dropout_p", c_f32), ("dropout_seed", c_u64), ("dropout_site", c_u32),
        ("self_resid", c_i32),
        ("dropout_seed_ptr", vp),
        ("a_u8", c_i32),
        ("resid_phys", c_i32),
    ]


class BceArgs(C.Structure):
    _fields_ = [("labels", vp), ("T", c_i64), ("label_smoothing", c_f32), ("downweight", c_i32), ("loss", vp),
                ("probs", vp), ("ldp", c_i64), ("logits", vp), ("ldl", c_i64), ("gscale", c_f32)]


class RowTailArgs(C.Structure):
    _fields_ = [
        ("dtype", c_i32), ("B", c_i64), ("D", c_i64),
        ("att", vp), ("rs_att", c_i64), ("resid", vp), ("rs_res", c_i64),
        ("Wp", vp), ("ldwp", c_i64), ("bp", vp), ("g1", vp), ("be1", vp),
        ("W1", vp), ("ldw1", c_i64), ("b1", vp), ("W2", vp), ("ldw2", c_i64), ("b2", vp), ("g2", vp), ("be2", vp),
        ("h1", vp), ("x1", vp), ("h2", vp), ("x2", vp), ("rs_d", c_i64), ("a", vp), ("rs_a", c_i64),
        ("mean1", vp), ("rstd1", vp), ("mean2", vp), ("rstd2", vp), ("stat_stride", c_i64),
        ("eps", c_f32), ("dropout_p", c_f32), ("dropout_seed", c_u64), ("dropout_seed_ptr", vp), ("site0", c_u32),
        ("phys_stride", c_i64), ("sync", vp), ("status", vp),
    ]


class RowTailBwdArgs(C.Structure):
    _fields_ = [
        ("dtype", c_i32), ("B", c_i64), ("D", c_i64),
        ("dy", vp), ("rs_dy", c_i64), ("h2", vp), ("h1", vp), ("rs_d", c_i64), ("a", vp), ("rs_a", c_i64),
        ("mean1", vp), ("rstd1", vp), ("mean2", vp), ("rstd2", vp), ("stat_stride", c_i64), ("g1", vp), ("g2", vp),
        ("W2t", vp), ("ldw2t", c_i64), ("W1t", vp), ("ldw1t", c_i64), ("Wpt", vp), ("ldwpt", c_i64),
        ("dh", vp), ("dhm", vp), ("dx1", vp), ("dh1m", vp), ("rs_c", c_i64), ("dpre", vp), ("rs_dpre", c_i64),
        ("dh1", vp), ("rs_dh1", c_i64), ("datt", vp), ("rs_datt", c_i64),
        ("dg1", vp), ("db1", vp), ("dg2", vp), ("db2", vp),
        ("dropout_p", c_f32), ("dropout_seed", c_u64), ("dropout_seed_ptr", vp), ("site0", c_u32),
        ("phys_stride", c_i64), ("sync", vp), ("status", vp),
    ]


class LnArgs(C.Structure):
    _fields_ = [
        ("mode", c_i32),
        ("gamma", vp), ("beta", vp), ("eps", c_f32),
        ("out", vp), ("ld_out", c_i64),
        ("mean", vp), ("rstd", vp),
        ("x", vp), ("ld_x", c_i64),
        ("dgamma", vp), ("dbeta", vp),
        ("mask_mode", c_i32),
        ("partials", vp),
    ]


class LnBwdIn(C.Structure):
    _fields_ = [
        ("dy", vp), ("ld_dy", c_i64),
        ("x", vp), ("ld_x", c_i64),
        ("gamma", vp), ("mean", vp), ("rstd", vp),
        ("dx", vp), ("ld_dx", c_i64),
        ("dx_masked", vp), ("ld_dxm", c_i64),
        ("dgamma", vp), ("dbeta", vp), ("partials", vp),
        ("mask_mode", c_i32),
        ("dropout_p", c_f32), ("dropout_seed", c_u64), ("dropout_seed_ptr", vp), ("dropout_site", c_u32),
    ]


# flags of the sticky step-status word (include/mst_hip.h: MST_TAIL_SPIN_*, MST_STEP_INCOMPLETE)
TAIL_SPIN_FWD, TAIL_SPIN_BWD, STEP_INCOMPLETE = 1, 2, 16


class StepMetrics(C.Structure):
    _fields_ = [("B", c_i64), ("recon", vp), ("kl", vp), ("kl_weight", c_f32), ("total", vp), ("metric", vp),
                ("status", vp), ("expect_ptr0", vp), ("expect_val0", c_u32), ("expect_ptr1", vp), ("expect_val1", c_u32),
                ("fin_recon", vp), ("fin_kl", vp), ("fin_B", c_i64)]


class PartialSum(C.Structure):
    _fields_ = [
        ("src", vp), ("n_parts", c_i64), ("stride", c_i64), ("len", c_i64),
        ("dst", vp), ("scale", c_f32),
    ]


class StepBeginArgs(C.Structure):
    _fields_ = [
        ("rng_state", vp), ("adam_state", vp), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
        ("eps_out", vp), ("n_eps", c_i64), ("eps_site", c_u32), ("eps_index0", c_i64),
        ("lens", vp), ("B", c_i64), ("mask_e", vp), ("Se", c_i64), ("add_e", c_i32), ("mask_d", vp), ("Sd", c_i64), ("add_d", c_i32),
        ("zero_a", vp), ("zero_a_bytes", c_i64), ("zero_b", vp), ("zero_b_bytes", c_i64),
        ("sh_dtype", c_i32), ("sh_w", vp), ("sh_wt16", vp), ("sh_desc", vp), ("sh_prefix", vp), ("sh_n_mat", c_i64), ("sh_tiles", c_i64),
    ]


class OuterJob(C.Structure):
    _fields_ = [
        ("L", vp), ("R", vp), ("r_dtype", c_i32), ("r_stride", c_i64),
        ("B", c_i64), ("J", c_i64), ("I", c_i64),
        ("out", vp), ("obias", vp),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("dtype", c_i32),
        ("M", c_i64), ("N", c_i64), ("K", c_i64),
        ("A", vp), ("lda", c_i64),
        ("B", vp), ("ldb", c_i64),
        ("dW", vp), ("ldw", c_i64),
        ("db", vp),
        ("scale", c_f32),
        ("a_rows_per_group", c_i64), ("a_group_stride", c_i64), ("a_group_offset", c_i64),
        ("b_rows_per_group", c_i64), ("b_group_stride", c_i64), ("b_group_offset", c_i64),
        ("a_u8", c_i32),
    ]


# name -> (restype, argtypes). Every symbol include/mst_hip.h declares must appear here.
SIGNATURES = {
    "mst_version": (C.c_int, []),
    "mst_last_error": (C.c_char_p, []),
    "mst_device_count": (C.c_int, []),
    "mst_graph_begin": (C.c_int, [vp]),
    "mst_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
    "mst_graph_launch": (C.c_int, [vp, vp]),
    "mst_graph_destroy": (C.c_int, [vp]),
    "mst_event_create": (C.c_int, [C.POINTER(vp)]),
    "mst_event_record": (C.c_int, [vp, vp]),
    "mst_event_sync": (C.c_int, [vp]),
    "mst_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(c_f32)]),
    "mst_event_destroy": (C.c_int, [vp]),
    "mst_gemm_nt": (C.c_int, [C.POINTER(GemmArgs), vp]),
    "mst_gemm_nt_pair": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), vp]),
    "mst_gemm_nt_pair_begin": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), C.POINTER(StepBeginArgs), vp]),
    "mst_step_begin_v": (C.c_int, [C.POINTER(StepBeginArgs), vp]),
    "mst_gemm_sigmoid_bce": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(BceArgs), vp]),
    "mst_gemm_sigmoid_bce_dgrad_ln": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(BceArgs), C.POINTER(GemmArgs), C.POINTER(LnArgs), vp]),
    "mst_row_tail_fwd": (C.c_int, [C.POINTER(RowTailArgs), vp]),
    "mst_row_tail_fwd_ride": (C.c_int, [C.POINTER(RowTailArgs), C.POINTER(GemmArgs), vp, vp]),
    "mst_row_tail_fwd_ride_shadows": (C.c_int, [C.POINTER(RowTailArgs), C.POINTER(GemmArgs), vp, C.c_int, vp, vp, vp, vp, c_i64, c_i64, vp]),
    "mst_row_tail_bwd": (C.c_int, [C.POINTER(RowTailBwdArgs), vp]),
    "mst_row_tail_bwd_ride": (C.c_int, [C.POINTER(RowTailBwdArgs), C.POINTER(GemmArgs), vp, vp]),
    "mst_ffn_ln_fwd": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), C.POINTER(LnArgs), vp]),
    "mst_ffn_ln_bwd": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), C.POINTER(LnArgs), vp]),
    "mst_ffn_ln_bwd_lead": (C.c_int, [C.POINTER(LnBwdIn), C.POINTER(GemmArgs), C.POINTER(GemmArgs), C.POINTER(LnArgs), vp]),
    "mst_proj_ffn_ln_fwd": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(LnArgs), C.POINTER(GemmArgs), C.POINTER(GemmArgs), C.POINTER(LnArgs), vp]),
    "mst_gemm_nt_ln": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(LnArgs), vp]),
    "mst_gemm_nt_ln_parts": (c_i64, [c_i64]),
    "mst_partial_sums": (C.c_int, [C.POINTER(PartialSum), C.c_int, vp]),
    "mst_layernorm_bwd_parts": (c_i64, [c_i64, c_i64]),
    "mst_gemm_wgrad": (C.c_int, [C.POINTER(WgradArgs), vp]),
    "mst_gemm_wgrad_batch": (C.c_int, [C.POINTER(WgradArgs), C.c_int, vp]),
    "mst_gemm_wgrad_batch_ws": (C.c_int, [C.POINTER(WgradArgs), C.c_int, vp, c_i64, vp]),
    "mst_gemm_wgrad_batch_sums": (C.c_int, [C.POINTER(WgradArgs), C.c_int, vp, c_i64, C.POINTER(PartialSum), C.c_int, vp]),
    "mst_gemm_wgrad_batch_flush": (C.c_int, [C.POINTER(WgradArgs), C.c_int, vp, c_i64, C.POINTER(PartialSum), C.c_int,
                                             C.POINTER(OuterJob), C.c_int, vp]),
    "mst_outer_jobs": (C.c_int, [C.POINTER(OuterJob), C.c_int, vp]),
    "mst_latent_bwd_vec": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, c_i64, vp, vp, vp, vp, vp, vp, vp, c_i64, c_f32, c_f32, c_f32,
                                     c_f32, vp, c_i64, vp, c_i64, vp, vp]),
    "mst_latent_bwd_vec_proj": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, c_i64, vp, vp, vp, vp, vp, vp, vp, c_i64, vp, c_i64, c_i64, vp, c_i64,
                                          c_f32, c_f32, c_f32, c_f32, vp, c_i64, vp, c_i64, vp, vp]),
    "mst_embed_fwd": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, vp, vp, c_i64, vp, vp, c_i64, vp, c_i64, c_f32,
                                vp, c_i64, c_i64, c_i64, vp, vp]),
    "mst_embed_bwd": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, vp, vp, c_i64, vp, vp, c_i64, c_f32,
                                vp, c_i64, c_i64, c_i64, vp]),
    "mst_group_colsum": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, vp, c_i64, c_i64, c_i64, vp, vp, c_i64, c_f32, vp]),
    "mst_mask_from_lengths": (C.c_int, [c_i64, c_i64, vp, c_i32, vp, vp]),
    "mst_attn_keysoftmax_fwd": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, c_i64, vp, c_i64, c_i64, c_i64, c_i64,
                                          vp, vp, vp, c_i64, c_i64, vp]),
    "mst_attn_qkv_fwd": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, c_i64, vp, c_i64, vp, c_i64, vp, vp, c_i64, c_i64, c_i64, c_i64, vp, vp, vp,
                                   c_i64, c_i64, vp]),
    "mst_attn_keysoftmax_bwd": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, c_i64, vp, c_i64, c_i64, c_i64, c_i64,
                                          vp, vp, vp, c_i64, vp, c_i64, vp, c_i64, vp]),
    "mst_attn_decode": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, c_i64, c_i64, vp, c_i64, c_i64, c_i64, c_i64, C.c_int, vp, c_i64, vp]),
    "mst_layernorm_fwd": (C.c_int, [C.c_int, c_i64, c_i64, vp, c_i64, vp, vp, c_f32, vp, c_i64, vp, vp, c_i64, vp]),
    "mst_layernorm_bwd": (C.c_int, [C.c_int, c_i64, c_i64, vp, c_i64, vp, vp, vp, vp, c_i64, vp, c_i64,
                                    vp, c_i64, vp, vp, C.c_int, c_f32, c_u64, c_u32, vp, c_i64, vp, vp]),
    "mst_latent_fwd": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, c_i64, vp, c_i64, vp, vp, vp, vp, vp, vp, vp, c_i64,
                                 vp, c_f32, vp, vp, vp, vp, vp, c_i64, vp]),
    "mst_latent_fwd_proj": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, c_i64, vp, c_i64, vp, vp, vp, vp, vp, vp, vp, c_i64,
                                      vp, c_f32, vp, vp, vp, vp, vp, c_i64, vp, c_i64, vp, vp, c_i64, c_i64, vp]),
    "mst_latent_bwd": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, c_i64, vp, c_i64, vp, vp, vp, vp, vp, vp, vp,
                                 vp, c_i64, c_f32, c_f32, c_f32, c_f32, vp, vp, vp, vp, vp, c_i64, vp, c_i64, vp, vp]),
    "mst_reparam_kl_fwd": (C.c_int, [c_i64, c_i64, vp, vp, vp, vp, vp, vp]),
    "mst_reparam_kl_bwd": (C.c_int, [c_i64, c_i64, vp, vp, vp, vp, c_f32, vp, vp, vp]),
    "mst_softmax_ce": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, vp, c_i64, vp, vp, vp, c_i64, vp, c_i64, c_f32, C.c_int, vp,
                                 C.c_int, vp]),
    "mst_ce_from_probs": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, vp, c_i64, vp, vp, vp]),
    "mst_bce_from_probs": (C.c_int, [C.c_int, c_i64, c_i64, vp, vp, c_f32, C.c_int, vp, vp]),
    "mst_sigmoid_bce": (C.c_int, [C.c_int, c_i64, c_i64, c_i64, vp, c_i64, vp, c_f32, C.c_int, vp, vp, vp, c_i64,
                                  vp, c_i64, c_f32, C.c_int, vp]),
    "mst_beam_step": (C.c_int, [c_i64, c_i64, c_i64, c_i64, c_i64, vp, c_i64, vp, vp, vp, vp, vp, vp, vp, c_i32, c_i32, vp]),
    "mst_beam_gather": (C.c_int, [vp, vp, vp, c_i64, c_i64, c_i64, c_i64, vp]),
    "mst_beam_gather_cols": (C.c_int, [vp, vp, vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, vp]),
    "mst_sample_step": (C.c_int, [c_i64, c_i64, c_i64, c_i64, vp, c_i64, vp, vp, vp, vp, c_u64, c_i32, c_i32, vp]),
    "mst_loss_combine": (C.c_int, [c_i64, vp, vp, c_f32, vp, vp, vp]),
    "mst_loss_combine_v": (C.c_int, [C.POINTER(StepMetrics), vp]),
    "mst_adam_flat": (C.c_int, [C.c_int, c_i64, vp, vp, vp, vp, vp, c_f64, c_f64, c_f64, c_f32, c_f32, c_f32, c_f32,
                                vp, C.c_int, C.POINTER(StepMetrics), vp]),
    "mst_adam_flat_emb": (C.c_int, [C.c_int, c_i64, vp, vp, vp, vp, vp, c_f64, c_f64, c_f64, c_f32, c_f32, c_f32, c_f32, vp,
                                    C.POINTER(StepMetrics), c_i64, C.POINTER(c_i64), c_i64, vp, vp]),
    "mst_transpose_shadows": (C.c_int, [C.c_int, vp, vp, vp, vp, c_i64, c_i64, vp]),
    "mst_segment_sumsq": (C.c_int, [vp, vp, c_i64, vp, vp]),
    "mst_cast_f32_to_act": (C.c_int, [C.c_int, c_i64, vp, vp, vp]),
    "mst_dropout_mask": (C.c_int, [c_i64, c_f32, c_u64, c_u32, vp, vp]),
    "mst_add_act": (C.c_int, [C.c_int, c_i64, vp, vp, vp, vp]),
    "mst_selftest": (C.c_int, [vp, vp]),
    "mst_zero": (C.c_int, [vp, c_i64, vp]),
    "mst_rng_advance": (C.c_int, [vp, vp]),
    "mst_randn": (C.c_int, [c_i64, vp, c_u64, vp, c_u32, vp]),
    "mst_step_begin": (C.c_int, [vp, vp, c_f64, c_f64, c_f64, vp, c_i64, c_u32, c_i64, vp, c_i64, vp, c_i64, c_i32, vp, c_i64, c_i32,
                                 vp, c_i64, vp, c_i64, vp]),
}


class MstError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libmst_hip.so and bind every declared symbol. Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MstError(
            f"{LIB_PATH} is missing: build it with `python -m musicstyletransfer_amd.csrc.build` "
            "(there is no CPU fallback for the training step)")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7. It must be in the process BEFORE this library
    # is dlopen()ed, so that our NEEDED libamdhip64.so.7 binds to that same runtime instance (same SONAME):
    # torch's hipStream_t handles and device pointers are then valid in our launches. Loaded the other way
    # round, two HIP runtimes coexist and the second reports "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MstError(f"libmst_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mst_last_error()
        raise MstError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    check(rc, name)

"""ctypes binding of libmade_hip.so (include/made_hip.h).

The product path has no CPU fallback: if the shared library is missing, `lib()` raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C mgsv_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MADE_LIB_PATH: another build of the same library (A/B measurements of two builds on one box, tools/ab_libs.sh); it must export the same
# ABI version, checked below, so a stale or older build fails here instead of reading shifted struct fields


def variant_env(name: str, default: str = "") -> str:
    """The value of a measurement knob (MADE_* variables that select a variant an A/B measurement once needed: docs/EXPERIMENTS.md) -- honoured
    only under MADE_DEBUG_VARIANTS=1, like the library's own (csrc/common.h made_variant_env): production has one code path."""
    if os.environ.get("MADE_DEBUG_VARIANTS", "0") in ("", "0"):
        return default
    return os.environ.get(name, default)


LIB_PATH = os.environ.get("MADE_LIB_PATH") or os.path.join(_HERE, "libmade_hip.so")
ABI_VERSION = 8                          # include/made_hip.h MADE_ABI_VERSION the ctypes mirrors in this file were written for

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_QUICKGELU, ACT_SIGMOID = 0, 1, 2, 3, 4
GATE_NONE, GATE_RELU_OUT, GATE_GELU_Z, GATE_QUICKGELU_Z, GATE_SIGMOID_OUT = 0, 1, 2, 3, 4

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float


class MadeDropout(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("site", C.c_uint32), ("p", f32), ("seed_device", vp)]


class MadeLinearSeg(C.Structure):
    _fields_ = [("col_begin", i64), ("out", vp), ("out_dtype", i32), ("transposed", i32), ("ldo", i64),
                ("rows_per_batch", i64), ("out_batch_stride", i64), ("out_z_stride", i64),
                ("use_a2", i32), ("_pad", i32)]


class MadeLinearArgs(C.Structure):
    _fields_ = [("A", vp), ("a_dtype", i32), ("w_dtype", i32), ("lda", i64),
                ("A2", vp), ("lda2", i64), ("a2_row_mod", i64),
                ("a2_replace", i32), ("drop_col_div", i32),
                ("a_row_mask", vp),
                ("W", vp), ("ldw", i64),
                ("bias", vp),
                ("M", i64), ("N", i64), ("K", i64),
                ("batch", i64), ("a_z_stride", i64), ("w_z_stride", i64),
                ("act", i32), ("r_dtype", i32),
                ("R", vp), ("ldr", i64), ("r_row_mod", i64),
                ("out_row_mask", vp),
                ("tile_skip_mask", vp),
                ("nseg", i32), ("split_k", i32),
                ("split_ws", vp),
                ("seg", MadeLinearSeg * 4),
                ("G", vp), ("g_dtype", i32), ("gate", i32), ("ldg", i64), ("gate_scale", f32), ("z_dtype", i32),
                ("Zout", vp), ("ldz", i64), ("drop", MadeDropout), ("drop_ld", i64), ("row_index", vp), ("n_rows", vp),
                ("bias_row_scale", vp), ("bias_z_stride", i64)]


class MadeFinishArgs(C.Structure):
    _fields_ = [("ws", vp), ("split_k", i64), ("M", i64), ("N", i64),
                ("bias", vp), ("act", i32), ("r_dtype", i32),
                ("R", vp), ("ldr", i64), ("r_row_mod", i64),
                ("out", vp), ("out_dtype", i32), ("_pad0", i32), ("ldo", i64),
                ("ln1_g", vp), ("ln1_b", vp), ("ln1_out", vp), ("ln1_dtype", i32), ("_pad1", i32), ("ln1_ld", i64),
                ("ln2_g", vp), ("ln2_b", vp), ("ln2_out", vp), ("ln2_dtype", i32), ("_pad2", i32), ("ln2_ld", i64),
                ("eps", f32), ("_pad3", i32)]


class MadeDecStageArgs(C.Structure):
    _fields_ = [("Zin", vp), ("ldz", i64), ("ln_g", vp), ("ln_b", vp),
                ("ln2_g", vp), ("ln2_b", vp), ("x2_out", vp), ("ldx2", i64),
                ("add", vp), ("add_row_mod", i64), ("x_out", vp), ("ldx", i64),
                ("W", vp), ("ldw", i64), ("bias", vp), ("R", vp), ("ldr", i64), ("out", vp), ("ldo", i64),
                ("out_dtype", i32), ("act", i32), ("res_from_x", i32), ("eps", f32),
                ("M", i64), ("N", i64), ("K", i64),
                ("zin_dtype", i32), ("drop_col_div", i32), ("a_out", vp), ("lda_out", i64), ("drop", MadeDropout), ("drop_ld", i64)]


class MadeDecStageBwdArgs(C.Structure):
    _fields_ = [("xa", vp), ("gamma_a", vp), ("ldxa", i64), ("xb", vp), ("gamma_b", vp), ("ldxb", i64), ("dy", vp), ("lddy", i64),
                ("add", vp), ("ldadd", i64), ("dx_out", vp), ("lddx", i64), ("a_out", vp), ("lda_out", i64),
                ("dgamma_a", vp), ("dbeta_a", vp), ("dgamma_b", vp), ("dbeta_b", vp), ("drop_a", MadeDropout), ("drop_a_ld", i64),
                ("W", vp), ("ldw", i64), ("G", vp), ("ldg", i64), ("gate_scale", f32), ("eps", f32),
                ("drop_o", MadeDropout), ("drop_o_ld", i64), ("drop_o_col_div", i32), ("_pad", i32),
                ("R", vp), ("ldr", i64), ("out", vp), ("ldo", i64), ("M", i64), ("N", i64), ("K", i64)]


class MadeAttnArgs(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("V", vp), ("O", vp),
                ("dtype", i32), ("hd", i32),
                ("B", i64), ("H", i64), ("Lq", i64), ("Lk", i64),
                ("q_bs", i64), ("ldq", i64), ("k_bs", i64), ("ldk", i64),
                ("v_bs", i64), ("ldv", i64), ("o_bs", i64), ("ldo", i64),
                ("key_mask", vp), ("q_mask", vp),
                ("scale", f32), ("_pad", i32),
                ("q_skip_mask", vp),
                ("lse", vp), ("drop", MadeDropout), ("batch_order", vp), ("keep_bits", vp), ("ld_bits", i64)]


class MadeAttnBwdArgs(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("V", vp), ("O", vp), ("dO", vp), ("dQ", vp), ("dK", vp), ("dV", vp),
                ("lse", vp), ("delta", vp), ("dtype", i32), ("hd", i32),
                ("B", i64), ("H", i64), ("Lq", i64), ("Lk", i64),
                ("q_bs", i64), ("ldq", i64), ("k_bs", i64), ("ldk", i64), ("v_bs", i64), ("ldv", i64), ("o_bs", i64), ("ldo", i64),
                ("do_bs", i64), ("lddo", i64), ("dq_bs", i64), ("lddq", i64), ("dk_bs", i64), ("lddk", i64), ("dv_bs", i64), ("lddv", i64),
                ("key_mask", vp), ("q_skip_mask", vp), ("scale", f32), ("_pad", i32), ("drop", MadeDropout),
                ("batch_order", vp), ("keep_bits", vp), ("ld_bits", i64)]


class MadeXpoolFusedArgs(C.Structure):
    _fields_ = [("Q", vp), ("ldq", i64), ("K", vp), ("U", vp), ("k_bs", i64), ("ldk", i64), ("u_bs", i64), ("ldu", i64),
                ("key_mask", vp), ("ln2_g", vp), ("ln2_b", vp), ("Wl", vp), ("ldw", i64), ("bl", vp), ("ln3_g", vp), ("ln3_b", vp),
                ("vn", vp), ("ldvn", i64), ("sims", vp), ("ld_sims", i64),
                ("Nv", i64), ("Nm", i64), ("S", i64), ("D", i64), ("scale", f32), ("eps", f32), ("ws", vp), ("prepare_ws", i32), ("_pad", i32)]


class MadeXpoolAttnArgs(C.Structure):
    _fields_ = [("Q", vp), ("ldq", i64), ("K", vp), ("U", vp), ("k_bs", i64), ("ldk", i64), ("u_bs", i64), ("ldu", i64),
                ("key_mask", vp), ("out", vp), ("ldo", i64), ("Nv", i64), ("Nm", i64), ("S", i64), ("D", i64),
                ("scale", f32), ("eps", f32), ("normalize", i32), ("_pad", i32), ("ws", vp)]


class MadeXpoolSimsArgs(C.Structure):
    _fields_ = [("Q", vp), ("ldq", i64), ("K", vp), ("UU", vp), ("k_bs", i64), ("ldk", i64), ("u_bs", i64), ("ldu", i64),
                ("key_mask", vp), ("av", vp), ("bv", vp), ("ln3_g", vp), ("ln3_b", vp), ("vn", vp), ("ldvn", i64),
                ("sims", vp), ("ld_sims", i64), ("Nv", i64), ("Nm", i64), ("S", i64), ("D", i64),
                ("scale", f32), ("eps", f32), ("ws", vp), ("prepare_ws", i32), ("_pad", i32)]


class MadeXpoolInbatchArgs(C.Structure):
    _fields_ = [("Q", vp), ("ldq", i64), ("K", vp), ("U", vp), ("k_bs", i64), ("ldk", i64), ("u_bs", i64), ("ldu", i64),
                ("key_mask", vp), ("out", vp), ("out_dtype", i32), ("_pad", i32), ("o_bs", i64), ("ldo", i64),
                ("Nv", i64), ("Nm", i64), ("S", i64), ("D", i64), ("scale", f32), ("_pad2", i32), ("ws", vp)]


class MadeWideAttnArgs(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("Kadd", vp), ("V", vp), ("O", vp), ("key_mask", vp),
                ("dtype", i32), ("o_dtype", i32),
                ("B", i64), ("NQ1", i64), ("NQ2", i64), ("L", i64), ("D", i64),
                ("q_bs", i64), ("q_s1", i64), ("q_s2", i64), ("k_bs", i64), ("ldk", i64), ("kadd_bs", i64), ("ldkadd", i64),
                ("v_bs", i64), ("ldv", i64), ("o_bs", i64), ("o_s1", i64), ("o_s2", i64),
                ("scale", f32), ("_pad", i32),
                ("n_split", i64), ("part_o", vp), ("part_ml", vp), ("drop", MadeDropout), ("sum_out", vp),
                ("lse_out", vp)]


class MadeWideAttnBwdArgs(C.Structure):
    _fields_ = ([(k, vp) for k in ("Q", "dO", "O", "K", "V", "key_mask", "lse", "ssum", "extra", "dattc")] + [("ld_dattc", i64), ("vbias", vp), ("hd", i64),
                ("Pd", vp), ("dS", vp), ("p_bs", i64), ("ld_p", i64), ("dQ", vp), ("dq_bs", i64), ("ld_dq", i64)] +
                [(k, i64) for k in ("B", "NQ", "L", "D", "q_bs", "ld_q", "do_bs", "ld_do", "o_bs", "ld_o", "k_bs", "ldk", "v_bs", "ldv")] +
                [("scale", f32), ("_pad", i32), ("n_split", i64), ("part_dq", vp), ("drop", MadeDropout)])


class MadeGemmTNArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("ab_dtype", i32), ("c_dtype", i32),
                ("M", i64), ("N", i64), ("K", i64), ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("batch1", i64), ("batch2", i64),
                ("a_zs1", i64), ("a_zs2", i64), ("b_zs1", i64), ("b_zs2", i64), ("c_zs1", i64), ("c_zs2", i64),
                ("row_mask", vp), ("mask_zs1", i64), ("mask_zs2", i64),
                ("alpha", f32), ("accumulate", i32), ("split_m", i64),
                ("colsum", vp), ("colsum_zs1", i64), ("colsum_zs2", i64), ("row_group_valid", vp),
                ("row_index", vp), ("n_rows", vp)]


class MadeGemmTNProblem(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("colsum", vp), ("N", i64), ("K", i64), ("lda", i64), ("ldb", i64), ("ldc", i64)]


class MadeGemmTNGroup(C.Structure):
    _fields_ = [("n_problems", i32), ("alpha", f32), ("M", i64), ("split_m", i64), ("row_index", vp), ("n_rows", vp),
                ("p", MadeGemmTNProblem * 8), ("tile_end", i32 * 8), ("tile_size", i32), ("_pad", i32),
                ("workspace", vp), ("workspace_bytes", i64)]


class MadeAdamGroup(C.Structure):
    _fields_ = [("begin", i64), ("end", i64), ("lr", f32), ("max_norm", f32)]


class MadeAdamDeviceState(C.Structure):
    _fields_ = [("step", i64), ("lr", f32 * 4), ("bc1", f32), ("bc2_sqrt", f32)]


class MadeRepackDesc(C.Structure):
    _fields_ = [("src", vp), ("w", vp), ("wt", vp), ("rows", i64), ("cols", i64), ("wt_ld", i64), ("tile_begin", i64),
                ("dtype", i32), ("_pad", i32)]


# name -> (restype, argtypes); every symbol include/made_hip.h declares
SIGNATURES = {
    "made_abi_version": (C.c_int, []),
    "made_set_f32_products": (C.c_int, [C.c_int]),
    "made_get_f32_products": (C.c_int, []),
    "made_last_error": (C.c_char_p, []),
    "made_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "made_tape_begin": (C.c_int, []),
    "made_tape_end": (C.c_int, [C.POINTER(C.c_uint64)]),
    "made_tape_replay": (C.c_int, [C.c_uint64]),
    "made_tape_free": (C.c_int, [C.c_uint64]),
    "made_tape_interleave": (C.c_int, [C.c_uint64, i32]),
    "made_tape_count": (C.c_int, [C.c_uint64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "made_tape_replay_range": (C.c_int, [C.c_uint64, i64, i64]),
    "made_tape_op": (C.c_int, [C.c_uint64, i64, C.POINTER(i32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "made_stream_wait": (C.c_int, [vp, vp]),
    "made_tape_event": (C.c_int, [i32, i32, vp]),
    "made_tape_callback": (C.c_int, [C.CFUNCTYPE(C.c_int, vp), vp]),
    "made_memset_async": (C.c_int, [vp, i32, i64, vp]),
    "made_copy_async": (C.c_int, [vp, vp, i64, vp]),
    "made_store_words": (C.c_int, [vp, vp, i32, vp]),
    "made_linear": (C.c_int, [C.POINTER(MadeLinearArgs), vp]),
    "made_splitk_finish": (C.c_int, [C.POINTER(MadeFinishArgs), vp]),
    "made_dec_stage": (C.c_int, [C.POINTER(MadeDecStageArgs), vp]),
    "made_dec_stage_bwd": (C.c_int, [C.POINTER(MadeDecStageBwdArgs), vp]),
    "made_attention": (C.c_int, [C.POINTER(MadeAttnArgs), vp]),
    "made_attention_wide": (C.c_int, [C.POINTER(MadeWideAttnArgs), vp]),
    "made_attention_wide_bwd": (C.c_int, [C.POINTER(MadeWideAttnBwdArgs), vp]),
    "made_layernorm": (C.c_int, [vp, i32, i64, i64, i64, vp, vp, vp, i32, i64, i64, i64, f32, vp, vp]),
    "made_layernorm_add": (C.c_int, [vp, i32, i64, vp, vp, vp, i32, i64, vp, i32, i64, vp, i64, i64, i64, f32, vp, vp]),
    "made_cast_mask_rows": (C.c_int, [vp, i64, vp, vp, i32, i64, i64, i64, vp]),
    "made_pack_music_records": (C.c_int, [vp, i32, i64, vp, i64, vp, i64, vp, i32, i64, i64, i64, i64, i64, vp]),
    "made_masked_mean": (C.c_int, [vp, i32, i64, i64, vp, vp, i64, i64, i64, vp]),
    "made_l2norm_rows": (C.c_int, [vp, i32, i64, vp, vp, i32, i64, i64, i64, f32, vp]),
    "made_sine_pe": (C.c_int, [vp, vp, vp, i32, i64, i64, i64, vp]),
    "made_masked_softmax": (C.c_int, [vp, i64, vp, i64, vp, i32, i64, i64, i64, i64, i64, f32, vp]),
    "made_xpool_tail": (C.c_int, [vp, i32, i64, vp, vp, vp, i64, vp, vp, i64, i64, i64, i64, f32, vp]),
    "made_clip_loss": (C.c_int, [vp, i64, i64, vp, f32, i32, vp, vp, vp]),
    "made_hungarian_match": (C.c_int, [vp, vp, vp, i64, i64, i64, i64, i32, f32, f32, f32, vp, i32, vp, vp, vp, vp, vp]),
    "made_attention_bwd": (C.c_int, [C.POINTER(MadeAttnBwdArgs), vp]),
    "made_layernorm_bwd": (C.c_int, [vp, i32, i64, i64, i64, vp, vp, i32, i64, vp, i32, i64, vp, i32, i64, vp, i64,
                                     C.POINTER(MadeDropout), i64, vp, vp, i64, i64, f32, vp, vp]),
    "made_layernorm_bwd2": (C.c_int, [vp, vp, i64, vp, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, C.POINTER(MadeDropout), i64, i32,
                                      vp, vp, vp, vp, i64, i64, f32, vp]),
    "made_pool_bwd": (C.c_int, [vp, vp, vp, vp, i32, i64, i64, vp, i32, i64, i64, vp, i32, i64, i64, i64, i64, i64, f32, vp]),
    "made_l2norm_bwd": (C.c_int, [vp, i32, i64, vp, i64, i64, vp, i64, i32, vp, i32, i64, i64, i64, f32, vp]),
    "made_clip_loss_bwd": (C.c_int, [vp, i64, i64, vp, f32, vp, vp, vp, vp, i32, vp, vp, vp]),
    "made_xpool_tail_bwd": (C.c_int, [vp, i32, i64, vp, vp, vp, i64, vp, i64, vp, i32, i64, vp, C.POINTER(MadeDropout),
                                      vp, vp, vp, i64, vp, i64, f32, i64, i64, i64, f32, vp]),
    "made_softmax_bwd": (C.c_int, [vp, i64, vp, i64, vp, i64, vp, f32, C.POINTER(MadeDropout), vp, vp, vp, i32, i64, i64,
                                   i64, i64, i64, i64, i64, vp]),
    "made_head_bias": (C.c_int, [vp, i32, i64, vp, vp, i64, i64, i64, vp]),
    "made_head_bias_bwd": (C.c_int, [vp, i32, i64, vp, vp, vp, vp, i64, i64, i64, vp]),
    "made_add3": (C.c_int, [vp, i32, vp, i32, vp, i32, vp, i32, i64, i64, vp]),
    "made_colsum": (C.c_int, [vp, i32, i64, i64, i64, vp, vp]),
    "made_posbn_relu_fwd": (C.c_int, [vp, i32, i64, vp, vp, vp, vp, C.c_float, C.c_float, i32, vp, vp, vp, i32, i64, i64, i64, i64, vp]),
    "made_posbn_relu_bwd": (C.c_int, [vp, i32, i64, vp, i32, i64, vp, i32, i64, vp, vp, vp, i32, vp, i32, i64, vp, vp, i64, i64, i64, vp]),
    "made_set_criterion_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i32, f32, vp, vp,
                                         vp, vp, i64, i32, vp, vp, vp]),
    "made_adam_step": (C.c_int, [vp, vp, vp, vp, i64, C.POINTER(MadeAdamGroup), i32, f32, f32, f32, i64, f32, vp, vp]),
    "made_adam_step_device": (C.c_int, [vp, vp, vp, vp, i64, C.POINTER(MadeAdamGroup), i32, f32, f32, f32, vp, i32, f32, vp, vp]),
    "made_repack": (C.c_int, [vp, i32, i64, vp]),
    "made_row_groups": (C.c_int, [vp, i64, vp, vp]),
    "made_row_index": (C.c_int, [vp, i64, vp, vp, vp]),
    "made_linear_variant": (C.c_int, [C.POINTER(MadeLinearArgs)]),
    "made_row_affine": (C.c_int, [vp, i32, i64, vp, vp, i64, i32, vp, i32, i64, i64, i64, vp]),
    "made_gate_rows": (C.c_int, [vp, i32, i64, vp, i32, i64, i32, f32, vp, i64, i64, vp, i32, i64, vp, i64, i64, vp]),
    "made_xpool_fused": (C.c_int, [C.POINTER(MadeXpoolFusedArgs), vp]),
    "made_xpool_attention": (C.c_int, [C.POINTER(MadeXpoolAttnArgs), vp]),
    "made_xpool_sims": (C.c_int, [C.POINTER(MadeXpoolSimsArgs), vp]),
    "made_xpool_sims_ws_bytes": (C.c_int64, [i64, i64, i64]),
    "made_xpool_inbatch": (C.c_int, [C.POINTER(MadeXpoolInbatchArgs), vp]),
    "made_xpool_inbatch_ws_bytes": (C.c_int64, [i64, i64]),
    "made_batch_order": (C.c_int, [vp, i64, i64, vp, vp]),
    "made_recall_ranks": (C.c_int, [vp, i64, vp, vp, i64, i64, i64, vp, vp, vp]),
    "made_span_iou": (C.c_int, [vp, vp, vp, vp, i64, i64, i32, f32, vp, vp, vp]),
    "made_gemm_tn": (C.c_int, [C.POINTER(MadeGemmTNArgs), vp]),
    "made_gemm_tn_grouped": (C.c_int, [C.POINTER(MadeGemmTNGroup), vp]),
    "made_gemm_tn_grouped_workspace": (i64, [C.POINTER(MadeGemmTNGroup)]),
    "made_concat_cols": (C.c_int, [vp, i64, vp, i64, vp, i64, vp]),
    "made_pooled_cosine": (C.c_int, [vp, i64, vp, i32, vp, i64, i64, i64, i64, i64, vp]),
    "made_scale_exp": (C.c_int, [vp, vp, vp, i64, vp]),
    "made_span_convert": (C.c_int, [vp, vp, i64, i32, vp]),
    "made_span_pairwise": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, vp]),
    "made_span_iou_se": (C.c_int, [vp, vp, vp, i64, f32, i32, i32, vp, vp]),
    "made_set_criterion": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i32, f32, vp, vp, vp, vp]),
}

_lib = None


class MadeError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libmade_hip.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise MadeError(f"{LIB_PATH} not found: build the HIP library first (make -C mgsv_amd/csrc); "
                            "the MaDe hot path has no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.made_abi_version() != ABI_VERSION:
            raise MadeError(f"ABI version mismatch: {LIB_PATH} is version {l.made_abi_version()}, this binding was written for {ABI_VERSION} "
                            "(rebuild: make -C mgsv_amd/csrc)")
        _lib = l
    return _lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = lib().made_last_error().decode("utf-8", "replace")
        raise MadeError(f"{what or 'libmade_hip'} failed (status {status}): {msg}")


def device_info():
    l = lib()
    name = C.create_string_buffer(64)
    cu, is950 = C.c_int(0), C.c_int(0)
    check(l.made_device_info(name, 64, C.byref(cu), C.byref(is950)), "made_device_info")
    return name.value.decode(), cu.value, bool(is950.value)

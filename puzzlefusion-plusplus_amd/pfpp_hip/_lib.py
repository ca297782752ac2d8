"""ctypes binding of libpfpp_hip.so (the C ABI declared in include/pfpp.h).

There is no CPU fallback: if the shared object is missing the import of any
kernel wrapper raises, and every wrapper refuses non-CUDA tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# PFPP_LIB_PATH (lab): another build of the library (A/B of build flags on one box)
LIB_PATH = Path(os.environ["PFPP_LIB_PATH"]).resolve() if os.environ.get("PFPP_LIB_PATH") else Path(__file__).resolve().parent / "libpfpp_hip.so"

_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_f32 = C.c_float


class GemmArgs(C.Structure):
    """mirror of struct pfpp_gemm_args (include/pfpp.h)"""

    _fields_ = [
        ("A", _p), ("W", _p), ("C", _p),
        ("w_hi", _p), ("w_lo", _p),
        ("a_hi", _p), ("a_lo", _p), ("c_hi", _p), ("c_lo", _p),
        ("bias", _p), ("scale", _p), ("shift", _p), ("residual", _p),
        ("M", _i64), ("N", _i64), ("K", _i64),
        ("lda", _i64), ("ldw", _i64), ("ldc", _i64), ("ldr", _i64),
        ("w_kmajor", _i32), ("act", _i32), ("pool", _i32),
        ("batch", _i32), ("zdiv", _i32), ("precision", _i32),
        ("sA0", _i64), ("sA1", _i64), ("sW0", _i64), ("sW1", _i64), ("sC0", _i64), ("sC1", _i64),
        ("sV0", _i64), ("sV1", _i64),
        ("alpha", _f32),
        ("a_mul", _p), ("a_add", _p), ("stats", _p), ("stats_copies", _i32), ("c_min", _p),
        ("split_ws", _p), ("split_ws_bytes", _i64), ("split_cnt", _p), ("split_cnt_len", _i64),
        ("gather_idx", _p), ("gather_xyz", _p), ("gather_ctr", _p),
        ("gather_N", _i32), ("gather_S", _i32), ("gather_ns", _i32),
    ]


class GemmGradArgs(C.Structure):
    """mirror of struct pfpp_gemm_grad_args (include/pfpp.h)"""

    _fields_ = [
        ("A", _p), ("W", _p), ("C", _p),
        ("M", _i64), ("N", _i64), ("K", _i64),
        ("lda", _i64), ("ldw", _i64), ("ldc", _i64),
        ("a_kmajor", _i32), ("w_kmajor", _i32), ("accumulate", _i32), ("split_k", _i32), ("batch", _i32),
        ("sA", _i64), ("sW", _i64), ("sC", _i64),
        ("a_scale", _f32), ("w_scale", _f32), ("alpha", _f32),
        ("colsum", _p),
    ]


_u64 = C.c_uint64
_u32 = C.c_uint32


class SaTrainArgs(C.Structure):
    """mirror of struct pfpp_sa_train_args (include/pfpp.h)"""

    _fields_ = [
        ("xyz", _p), ("new_xyz", _p), ("feats", _p), ("idx", _p),
        ("w_hi", _p * 3), ("w_lo", _p * 3), ("bias", _p * 3), ("a_mul", _p * 2), ("a_add", _p * 2),
        ("stats", _p), ("stats_copies", _i64),
        ("y_out", _p), ("y_in", _p), ("u_in", _p), ("out_max", _p), ("out_min", _p),
        ("F", _i64), ("N", _i64), ("S", _i64), ("ns", _i64), ("D", _i64), ("C1", _i64), ("C2", _i64), ("C3", _i64),
        ("stage", _i32), ("max_workgroups", _i64), ("sched", _p),
    ]


class PlanesC(C.Structure):
    """mirror of struct pfpp_planes (include/pfpp.h)"""

    _fields_ = [("hi", _p), ("lo", _p), ("scale", _f32)]


class SampleLevel(C.Structure):
    """mirror of struct pfpp_sample_level (include/pfpp.h)"""

    _fields_ = [("S", _i64), ("nsample", _i64), ("r2", _f32), ("fps_idx", _p), ("new_xyz", _p), ("ball_idx", _p)]


class SlabJob(C.Structure):
    """pfpp_slab_job"""
    _fields_ = [("ws", _p), ("C", _p), ("csum_ws", _p), ("csum", _p), ("M", C.c_int32), ("N", C.c_int32), ("ldc", C.c_int64),
                ("splits", C.c_int32), ("accumulate", C.c_int32)]


class GemmPlanesArgs(C.Structure):
    """mirror of struct pfpp_gemm_planes_args (include/pfpp.h)"""

    _fields_ = [
        ("a_hi", _p), ("a_lo", _p), ("w_hi", _p), ("w_lo", _p), ("C", _p), ("bias", _p), ("residual", _p),
        ("M", _i64), ("N", _i64), ("K", _i64),
        ("lda", _i64), ("ldw", _i64), ("ldc", _i64), ("ldr", _i64),
        ("a_kmajor", _i32), ("w_kmajor", _i32), ("act", _i32), ("accumulate", _i32), ("splits", _i32), ("variant", _i32),
        ("alpha", _f32), ("single_pass", _i32),
        ("ws", _p), ("ws_bytes", _i64),
        ("colsum", _p), ("colsum_alpha", _f32),
        ("defer", _p),
    ]


_pl = C.POINTER(PlanesC)


class PwC(C.Structure):
    """mirror of struct pfpp_pw (include/pfpp.h)"""

    _fields_ = [("f32", _p), ("hi", _p), ("lo", _p), ("scale", _f32), ("ldw", _i64), ("fhi", _p), ("flo", _p)]


class ElayerParams(C.Structure):
    """mirror of struct pfpp_elayer_params (include/pfpp.h)"""

    _fields_ = ([(n, PwC) for n in ("qkv1", "o1", "qkv2", "o2", "ff1", "ff2")] + [(n, _p) for n in ("bo1", "bo2", "g3", "b3", "bff1", "bff2")])


class TlayersEvalArgs(C.Structure):
    """mirror of struct pfpp_tlayers_eval_args (include/pfpp.h)"""

    _fields_ = [
        ("n_layers", _i32), ("layers", C.POINTER(ElayerParams)),
        ("M", _i64), ("C", _i64), ("H", _i64), ("L", _i64), ("inner", _i64), ("Fv", _i64), ("B", _i64),
        ("h", _p), ("mods", _p), ("frag_b", _p), ("seq_off", _p), ("seq_len", _p),
        ("n_seq", _i64), ("max_len", _i64), ("att_scale", _f32), ("single_pass", _i32),
        ("norm", PlanesC), ("att", PlanesC), ("u", PlanesC), ("qkv", _p),
        ("split_ws", _p), ("split_ws_bytes", _i64), ("split_cnt", _p), ("split_cnt_len", _i64),
        ("lnlin_max_rows", _i64), ("wd_gemm", _i32),
    ]


class HeadParams(C.Structure):
    """mirror of struct pfpp_head_params (include/pfpp.h)"""

    _fields_ = [("w0", PlanesC), ("w2", PlanesC), ("w4", _p), ("b0", _p), ("b2", _p), ("b4", _p), ("f0", PlanesC), ("f2", PlanesC)]


class HeadGrads(C.Structure):
    """mirror of struct pfpp_head_grads (include/pfpp.h)"""

    _fields_ = [("w4", _p), ("b4", _p), ("b2", _p), ("b0", _p)]


class TlayerParams(C.Structure):
    """mirror of struct pfpp_tlayer_params (include/pfpp.h)"""

    _fields_ = ([(n, PlanesC) for n in ("qkv1", "o1", "qkv2", "o2", "ff1", "ff2")] +
                [(n, _p) for n in ("bo1", "bo2", "g3", "b3", "bff1", "bff2")])


class TlayerGrads(C.Structure):
    """mirror of struct pfpp_tlayer_grads (include/pfpp.h)"""

    _fields_ = [(n, _p) for n in ("qkv1_w", "o1_w", "o1_b", "qkv2_w", "o2_w", "o2_b", "g3", "b3", "ff1_w", "ff1_b", "ff2_w", "ff2_b")]


class TlayerAdamw(C.Structure):
    """mirror of struct pfpp_tlayer_adamw (include/pfpp.h)"""

    _fields_ = [("p", _p), ("g", _p), ("m", _p), ("v", _p), ("hi", _p), ("lo", _p), ("n", _i64)]


class TlayersArgs(C.Structure):
    """mirror of struct pfpp_tlayers_args (include/pfpp.h)"""

    _fields_ = [
        ("n_layers", _i32), ("layers", C.POINTER(TlayerParams)), ("grads", C.POINTER(TlayerGrads)), ("adamw", C.POINTER(TlayerAdamw)),
        ("M", _i64), ("C", _i64), ("H", _i64), ("L", _i64), ("inner", _i64), ("Fv", _i64), ("B", _i64),
        ("h_in", _p), ("mods", _p), ("frag_b", _p), ("seq_off", _p), ("seq_len", _p),
        ("n_seq", _i64), ("max_len", _i64),
        ("att_scale", _f32), ("p_tok", _f32), ("p_lay", _f32),
        ("seed", C.c_uint64),
        ("fwd_arena", _p), ("fwd_layer_bytes", _i64),
        ("ws_main", _p), ("ws_side", _p), ("ws_bytes", _i64),
        ("bwd_arena", _p), ("bwd_bytes", _i64),
        ("grad_scale", _f32),
        ("dh", _p), ("dhp", PlanesC), ("dhp_out", C.POINTER(PlanesC)), ("dmods", _p), ("dtok", _p),
        ("lr", _f32), ("beta1", _f32), ("beta2", _f32), ("eps", _f32), ("weight_decay", _f32), ("bc1", _f32), ("bc2", _f32),
        ("opt_g_scale", _f32), ("opt_zero_grad", _i32), ("overflow", _p),
        ("frag_ws", _p), ("frag_ws_bytes", _i64),
        ("ada_se", _p), ("ada_dse", _p), ("ada_w", _p), ("ada_gw", _p), ("ada_gb", _p),
        ("ada_adamw_w", C.POINTER(TlayerAdamw)), ("ada_adamw_b", C.POINTER(TlayerAdamw)),
    ]


class DwJob(C.Structure):
    """mirror of struct pfpp_dw_job (include/pfpp.h)"""

    _fields_ = [("dy", PlanesC), ("x", PlanesC), ("gw", _p), ("gb", _p), ("M", _i64), ("N", _i64)]


class ReblockJob(C.Structure):
    """mirror of struct pfpp_reblock_job (include/pfpp.h)"""

    _fields_ = [("w", PlanesC), ("N", _i64), ("K", _i64), ("ldw", _i64), ("fhi", _p), ("flo", _p), ("transposed", _i32)]


# name -> argtypes (all return int); must list every symbol include/pfpp.h declares
SIGNATURES = {
    "pfpp_tlayers_eval": [C.POINTER(TlayersEvalArgs), _p],
    "pfpp_embed_tokens_small": [_p, _p, _p, _p, _p, C.POINTER(PwC), _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    "pfpp_gemm_small": [_pl, _i64, C.POINTER(PwC), _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p],
    "pfpp_gemm_wd": [_pl, _i64, C.POINTER(PwC), _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p],
    "pfpp_gemm_wd_supported": [_i64, _i64, _i64],
    "pfpp_gemm_wd_f16": [_pl, _i64, C.POINTER(PwC), _p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p],
    "pfpp_reblock_planes": [C.POINTER(ReblockJob), _i32, _p],
    "pfpp_layernorm_linear_small": [_p, _p, _i64, _p, _p, _p, _i64, C.POINTER(PwC), _p, _p, _i64, _pl, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_heads_fwd": [_p, C.POINTER(HeadParams), C.POINTER(HeadParams), _i64, _i64, _p, _p, _p, _p, _p, _p, _i64, _p],
    "pfpp_heads_bwd": [_p, _p, C.POINTER(HeadParams), C.POINTER(HeadParams), _i64, _i64, _p, _p, _p, _p, _p, _p, _p,
                       C.POINTER(HeadGrads), C.POINTER(HeadGrads), _f32, _p, _i64, _p],
    "pfpp_tlayers_fwd": [C.POINTER(TlayersArgs), _i32, _i32, _p],
    "pfpp_tlayers_bwd": [C.POINTER(TlayersArgs), _i32, _i32, _p, _p],
    "pfpp_set_attention_mode": [C.c_int],
    "pfpp_se3_rotate_gather": [_p, _p, _p, _p, _i64, _i64, _p],
    "pfpp_pose_apply": [_p, _p, _p, _p, _i64, _i64, C.c_int, _p],
    "pfpp_fps": [_p, _p, _p, _i64, _i64, _i64, _p],
    "pfpp_ball_query": [_p, _p, _p, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_sample_levels": [_p, _i64, _i64, C.POINTER(SampleLevel), _p],
    "pfpp_group_gather": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p],
    "pfpp_gemm": [C.POINTER(GemmArgs), _p],
    "pfpp_sa_mlp3_fused": [_p] * 16 + [_i64] * 7 + [_p],
    "pfpp_sa_mlp2_fused": [_p] * 13 + [_i64] * 7 + [_p],
    "pfpp_sa_mlp2_fused_p": [_p] * 13 + [_pl] + [_i64] * 7 + [_p],
    "pfpp_sa_mlp2_table_p": [_p] * 11 + [_pl] + [_i64] * 8 + [_p],
    "pfpp_sa_table_planes": [_p] * 7 + [_pl] + [_i64] * 6 + [_p],
    "pfpp_vq_encode": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "pfpp_scatter_rows": [_p, _p, _p, _i64, _i64, _p],
    "pfpp_token_features": [_p, _p, _p, _p, _p, _p, _i64, _i64, _p],
    "pfpp_token_features_slots": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p],
    "pfpp_token_combine_slots": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    "pfpp_token_combine_bwd_slots": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    "pfpp_mse_loss_masked": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _f32, _p],
    "pfpp_token_combine": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "pfpp_token_combine_list": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    "pfpp_layernorm_grouped": [_p, _p, _p, _i64, _p, _i64, _i64, _i64, _f32, _p],
    "pfpp_layernorm_grouped_split": [_p, _p, _p, _p, _i64, _p, _i64, _i64, _i64, _f32, _p],
    "pfpp_silu_embed": [_p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "pfpp_layernorm": [_p, _p, _p, _i64, _p, _p, _i64, _i64, _i64, _f32, _p],
    "pfpp_layernorm_split": [_p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _i64, _f32, _p],
    "pfpp_attn_blockdiag": [_p, _p, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_attn_blockdiag_split": [_p, _p, _p, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_attn_dense": [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_attn_dense_split": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_softmax_rows": [_p, _p, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_mean_pool": [_p, _p, _i64, _i64, _i64, _p],
    "pfpp_ddpm_step": [_p, _p, _p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _p],
    "pfpp_add_noise": [_p, _p, _p, _p, _p, _i64, _i64, _p],
    "pfpp_verifier_embed": [_p, _p, _p, _p, _i64, _i64, _i64, _p],
    "pfpp_pose_compose": [_p, _p, _p, _p, _p, _i64, _p],
    "pfpp_pose_apply_points": [_p, _p, _p, _p, _i64, C.c_int, _p],
    "pfpp_edge_histogram": [_p, _p, _p, _p, _p, _i64, _i64, _p],
    "pfpp_fragment_prepare": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p],
    "pfpp_estimate_normals": [_p, _p, _i64, _i64, _i64, _p],
    "pfpp_merge_keep_mask": [_p, _p, _p, _i64, _i64, _f32, _p],
    "pfpp_fps_start": [_p, _p, _p, _i64, _i64, _i64, _p, _p],
    "pfpp_nn_dist": [_p, _p, _p, _i64, _i64, _i64, _p],
    "pfpp_quat_to_euler_xyz": [_p, _p, _i64, C.c_int, _p],
    # ---- training (a17)
    "pfpp_gemm_grad": [C.POINTER(GemmGradArgs), _p],
    "pfpp_gemm_grad_group": [C.POINTER(GemmGradArgs), C.c_int, _p],
    "pfpp_colsum": [_p, _p, _i64, _i64, _i64, _i64, _i64, _i64, C.c_int, _p],
    "pfpp_dropout": [_p, _p, _p, _i64, _f32, _u64, _u32, _p],
    "pfpp_dropout_mask": [_p, _i64, _f32, _u64, _u32, _p],
    "pfpp_geglu": [_p, _p, _i64, _i64, _f32, _u64, _u32, _p],
    "pfpp_geglu_bwd": [_p, _p, _p, _i64, _i64, _f32, _u64, _u32, _p],
    "pfpp_act": [_p, _p, _i64, C.c_int, _p],
    "pfpp_act_bwd": [_p, _p, _p, _i64, C.c_int, _p],
    "pfpp_layernorm_bwd": [_p, _p, _p, _i64, _p, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _f32, _p],
    "pfpp_layernorm_bwd_dropout": [_p, _p, _p, _i64, _p, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _f32, _p, _f32, _u64, _u32, _p],
    "pfpp_dropout_layernorm": [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _f32, _f32, _u64, _u32, _p],
    "pfpp_attn_blockdiag_bwd": [_p, _p, _p, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_attn_dense_train": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_attn_dense_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, _p],
    "pfpp_attn_dense_bwd_parts": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, C.c_int, _p],
    "pfpp_mean_pool_bwd": [_p, _p, _i64, _i64, _i64, _p],
    "pfpp_token_combine_bwd": [_p, _p, _p, _p, _i64, _i64, _i64, _p],
    "pfpp_silu_embed_bwd": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "pfpp_silu_embed_bwd_mark": [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p],
    "pfpp_adamw_rows_active": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, C.c_int, _p, _p],
    "pfpp_embed_pack_weights": [_p, _p, _p, _p, _p, _p, _p, _i64, _p],
    "pfpp_token_features_t": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p],
    "pfpp_token_embed_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, C.c_float, _p],
    "pfpp_ada_linear_bwd": [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    "pfpp_mse_loss": [_p, _p, _p, _p, _p, _i64, _i64, _f32, _p],
    "pfpp_bn_stats": [_p, _i64, _i64, _i64, _p, _p, _p, _p, _f32, _p, _p],
    "pfpp_bn_finalize": [_p, _i64, _i64, _i64, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _p, _p],
    "pfpp_bn_minmax_apply": [_p, _p, _p, _p, _p, _i64, _i64, _p],
    "pfpp_bn_apply": [_p, _i64, _i64, _i64, _p, _p, _p, _p, _f32, _p, _i64, _i64, _p],
    "pfpp_adamw": [_p, _p, _p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _p],
    "pfpp_adamw_zero": [_p, _p, _p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, C.c_int, _p],
    "pfpp_sa_train_stage": [C.POINTER(SaTrainArgs), _p],
    "pfpp_sa_pad_schedule": [_p, _i64, _i64, _p, _p],
    "pfpp_adamw_guarded": [_p, _p, _p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, C.c_int, _p, _p],
    "pfpp_adamw_rows": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _i64, C.c_int, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, C.c_int, _p, _p],
    # ---- plane GEMM and plane-producing forms of the training kernels
    "pfpp_gemm_planes": [C.POINTER(GemmPlanesArgs), _p],
    "pfpp_slab_reduce_group": [C.POINTER(SlabJob), C.c_int32, _p],
    "pfpp_gemm_dw_group": [C.POINTER(DwJob), _i32, _i64, _i32, _p],
    "pfpp_split_planes": [_p, _i64, _pl, _p],
    "pfpp_colsum_planes": [_p, _p, _p, _i64, _i64, _i64, _f32, _p],
    "pfpp_geglu_p": [_p, _p, _i64, _i64, _f32, _u64, _u32, _pl, _p],
    "pfpp_geglu_bwd_p": [_p, _p, _p, _i64, _i64, _f32, _u64, _u32, _pl, _p],
    "pfpp_dropout_layernorm_p": [_p, _p, _p, _p, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _f32, _f32, _u64, _u32, _pl, _p],
    "pfpp_layernorm_bwd_p": [_p, _p, _p, _i64, _p, _p, _i64, _i64, _p, _p, _p, _i64, _i64, _i64, _f32, _p, _f32, _u64, _u32, _i32,
                             _pl, _pl, _p],
    "pfpp_attn_dense_train_p": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, _pl, _p],
    "pfpp_attn_dense_bwd_p": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, _pl, _p],
    "pfpp_attn_blockdiag_bwd_p": [_p, _p, _p, _i64, _i64, _i64, _i64, _f32, _pl, _p],
}
PLAIN = {
    "pfpp_version": ([], C.c_int),
    "pfpp_last_error": ([], C.c_char_p),
    "pfpp_build_info": ([], C.c_char_p),
    "pfpp_abi_sizeof": ([C.c_char_p], C.c_int64),
    "pfpp_last_gemm_kernel": ([], C.c_char_p),
    "pfpp_device_cu_count": ([], C.c_int),
    "pfpp_get_attention_mode": ([], C.c_int),
    "pfpp_ada_linear_bwd_scratch_floats": ([_i64, _i64], C.c_int64),
    "pfpp_token_features_t_cols": ([_i64, _i64], C.c_int64),
    "pfpp_tlayers_fwd_bytes": ([_i64, _i64, _i64, _i64], C.c_int64),
    "pfpp_tlayers_frag_bytes": ([_i64, _i64], C.c_int64),
    "pfpp_tlayers_fwd_hout_offset": ([_i64, _i64, _i64, _i64], C.c_int64),
    "pfpp_tlayers_bwd_bytes": ([_i64, _i64, _i64, _i64], C.c_int64),
    "pfpp_bn_stats_workspace": ([_i64, _i64], C.c_int64),
    "pfpp_fragment_prepare_workspace": ([_i64, _i64], C.c_int64),
}

# struct name in include/pfpp.h (without the pfpp_ prefix) -> its mirror here; load() compares the sizes
STRUCT_MIRRORS = {
    "sample_level": SampleLevel, "gemm_args": GemmArgs, "planes": PlanesC, "slab_job": SlabJob, "gemm_planes_args": GemmPlanesArgs,
    "sa_train_args": SaTrainArgs, "gemm_grad_args": GemmGradArgs, "tlayer_params": TlayerParams, "tlayer_grads": TlayerGrads,
    "tlayer_adamw": TlayerAdamw, "tlayers_args": TlayersArgs, "pw": PwC, "elayer_params": ElayerParams,
    "tlayers_eval_args": TlayersEvalArgs, "head_params": HeadParams, "head_grads": HeadGrads, "reblock_job": ReblockJob, "dw_job": DwJob,
}

ACT = {"none": 0, "relu": 1, "silu": 2, "gelu": 3, "geglu": 4}

_lib = None


class PfppError(RuntimeError):
    pass


def _bind_torch_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  The kernels must run on THAT runtime
    (same device context, streams and allocations as the tensors they are handed), so it is
    loaded with RTLD_GLOBAL before libpfpp_hip.so: the library's NEEDED libamdhip64 entry then
    resolves to the already-loaded copy whatever the import order of the process was."""
    import os

    import torch

    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


ABI_VERSION = 2


def build_info(lib) -> dict:
    """pfpp_build_info() as a dict ("abi", "arch", "fma_mix_insts", "packed_fp32_ops", "chain_prio")"""
    fn = lib.pfpp_build_info
    fn.argtypes, fn.restype = [], C.c_char_p
    return dict(kv.split("=", 1) for kv in fn().decode().split(";") if "=" in kv)


def _attest(lib) -> None:
    """Refuse a libpfpp_hip.so of another ABI revision or one compiled without the two correctness switches of pfpp_hip/build.py:
    `-fma-mix-insts` (one fp16 rounding per hi / lo split) and `-packed-fp32-ops` (round 5: with v_pk_*_f32 in the
    library, fps_kernel picked a wrong farthest point next to a co-running GEMM once in 10^2 .. 10^4 launches, and 117 of 150 training
    steps differed between two runs).  PFPP_PACKED_FP32=1 (the lab switch that BUILDS such a library) also lets it load."""
    lib.pfpp_version.argtypes, lib.pfpp_version.restype = [], C.c_int
    if not hasattr(lib, "pfpp_build_info") or lib.pfpp_version() != ABI_VERSION:
        raise PfppError(f"{LIB_PATH}: ABI version {lib.pfpp_version()} != {ABI_VERSION} — rebuild it (python __graft_entry__.py)")
    info = build_info(lib)
    if info.get("fma_mix_insts") != "off":
        raise PfppError(f"{LIB_PATH} was not built by pfpp_hip/build.py: fma_mix_insts={info.get('fma_mix_insts')} "
                        "(the hi / lo splits need `-Xclang -target-feature -Xclang -fma-mix-insts`; DESIGN.md 6.1)")
    if info.get("packed_fp32_ops") != "off" and os.environ.get("PFPP_PACKED_FP32") != "1":
        raise PfppError(f"{LIB_PATH} was built with packed fp32 instructions (packed_fp32_ops={info.get('packed_fp32_ops')}): wrong "
                        "farthest-point samples next to co-running kernels (DESIGN.md 6).  Rebuild with pfpp_hip/build.py, or set "
                        "PFPP_PACKED_FP32=1 to load it anyway (lab)")


def load() -> C.CDLL:
    """dlopen libpfpp_hip.so and attach prototypes; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise PfppError(
            f"{LIB_PATH} is missing: the HIP kernels are the only implementation of this path. "
            "Build them with `python __graft_entry__.py` (hipcc --offload-arch=gfx950)."
        )
    _bind_torch_hip_runtime()
    lib = C.CDLL(str(LIB_PATH))
    _attest(lib)          # ABI revision and code-generation switches first: a foreign build must not get as far as a launch
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, (argtypes, restype) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    for cname, mirror in STRUCT_MIRRORS.items():
        want = lib.pfpp_abi_sizeof(cname.encode())
        if want != C.sizeof(mirror):
            raise PfppError(f"struct pfpp_{cname}: the library has {want} bytes, the ctypes mirror {mirror.__name__} {C.sizeof(mirror)} — "
                            "rebuild libpfpp_hip.so (python __graft_entry__.py) or update pfpp_hip/_lib.py")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().pfpp_last_error()
        raise PfppError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")

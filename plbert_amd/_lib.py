"""ctypes binding of libplbert_hip.so — the C ABI declared in include/plbert.h.

There is no CPU fallback: if the library is missing or fails to load, every entry point of the
product path raises.  torch is imported first so the library resolves HIP against the same
libamdhip64 torch already mapped (one HIP runtime per process; torch's stream handles are then
valid in our launches).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL: see module docstring)

HERE = os.path.dirname(os.path.abspath(__file__))
# PLBERT_HIP_LIB: load another build of the same C ABI (kernel timing experiments, tools/build_dbg.sh)
LIB_PATH = os.environ.get("PLBERT_HIP_LIB") or os.path.join(HERE, "libplbert_hip.so")

PLB_PARAM_NAMES = [
    "encoder.embeddings.word_embeddings.weight",
    "encoder.embeddings.position_embeddings.weight",
    "encoder.embeddings.token_type_embeddings.weight",
    "encoder.embeddings.LayerNorm.weight",
    "encoder.embeddings.LayerNorm.bias",
    "encoder.encoder.embedding_hidden_mapping_in.weight",
    "encoder.encoder.embedding_hidden_mapping_in.bias",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.full_layer_layer_norm.weight",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.full_layer_layer_norm.bias",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.query.weight",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.key.weight",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.value.weight",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.query.bias",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.key.bias",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.value.bias",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.dense.weight",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.dense.bias",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.LayerNorm.weight",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.attention.LayerNorm.bias",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.ffn.weight",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.ffn.bias",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.ffn_output.weight",
    "encoder.encoder.albert_layer_groups.0.albert_layers.0.ffn_output.bias",
    "phoneme_predictor.weight",
    "phoneme_predictor.bias",
    "encoder.pooler.weight",
    "encoder.pooler.bias",
    "token_predictor.weight",
    "token_predictor.bias",
]
PLB_NPARAM = len(PLB_PARAM_NAMES)

# every symbol include/plbert.h declares (tests check the library exports all of them)
PUBLIC_SYMBOLS = [
    "plb_last_error", "plb_create", "plb_destroy", "plb_param_layout", "plb_workspace_bytes", "plb_bind",
    "plb_sync_weights", "plb_forward", "plb_pooler", "plb_loss_fwd_bwd", "plb_loss_fwd", "plb_loss_fwd_bwd_dual", "plb_adamw_step",
    "plb_set_fp8", "plb_fp8_state", "plb_fp8_stats", "plb_token_head_steps", "plb_set_token_head_steps", "plb_comm_unique_id", "plb_comm_init", "plb_comm_destroy",
    "plb_comm_info", "plb_comm_pieces", "plb_last_application_rows", "plb_status", "plb_status_ex", "plb_poll_status", "plb_status_export", "plb_status_import", "plb_broadcast_params", "plb_set_grad_overlap", "plb_allreduce_grads", "plb_apply_mask",
    "plb_mask_batch", "plb_profile_enable", "plb_profile_num_classes", "plb_profile_class_name", "plb_profile_read",
    # test / tuning hooks (documented as such at the end of the header)
    "plb_debug_skip_piece", "plb_debug_ln_fault", "plb_debug_hb_audit", "plb_debug_hb_report", "plb_comm_trace", "plb_comm_trace_read",
    "plb_set_gemm_nt_tile", "plb_set_gemm_nt_prefetch", "plb_set_attn_bwd_fused", "plb_set_prune_last",
]


class PlbConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("embedding_size", C.c_int32), ("hidden_size", C.c_int32),
        ("num_attention_heads", C.c_int32), ("intermediate_size", C.c_int32), ("num_hidden_layers", C.c_int32),
        ("max_position_embeddings", C.c_int32), ("type_vocab_size", C.c_int32), ("layer_norm_eps", C.c_float),
        ("num_phonemes", C.c_int32), ("num_tokens", C.c_int32), ("max_batch", C.c_int32), ("max_seq", C.c_int32),
        ("inference_only", C.c_int32),
    ]


# ---- internal launch structs (csrc/plbert_kernels.h) — used by the kernel-level GPU tests ------------
class PlbGemmNT(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("Mstore", C.c_int),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("ldr", C.c_int), ("aux", C.c_void_p), ("ldaux", C.c_int),
        ("C", C.c_void_p), ("ldc", C.c_int), ("C2", C.c_void_p), ("ldc2", C.c_int), ("Cf", C.c_void_p), ("ldcf", C.c_int),
        ("colpart", C.c_void_p),
        ("ce_cols", C.c_int), ("ce_tgt", C.c_void_p), ("ce_pmax", C.c_void_p), ("ce_psum", C.c_void_p),
        ("ce_tlogit", C.c_void_p), ("ce_lse", C.c_void_p), ("ce_w", C.c_void_p),
        ("deq_a", C.c_void_p), ("deq_b", C.c_void_p), ("C8", C.c_void_p), ("ldc8", C.c_int), ("q_scale", C.c_void_p),
        ("q_amax", C.c_void_p), ("c8_bf8", C.c_int),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_mean", C.c_void_p), ("ln_rstd", C.c_void_p),
        ("ln_eps", C.c_float), ("ln_xchg", C.c_void_p), ("ln_err", C.c_void_p), ("ln_fault", C.c_int),
    ]


class PlbGemmTN(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int), ("Ncols", C.c_int), ("B", C.c_void_p), ("ldb", C.c_int),
        ("Mtot", C.c_int), ("N", C.c_int), ("K", C.c_int), ("rows_per_split", C.c_int), ("splits", C.c_int),
        ("slab", C.c_void_p), ("deq_a", C.c_void_p), ("deq_b", C.c_void_p),
    ]


class PlbAttn(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("ldqkv", C.c_int), ("lengths", C.c_void_p),
        ("B", C.c_int), ("S", C.c_int), ("NH", C.c_int), ("H", C.c_int), ("scale", C.c_float),
        ("ctx", C.c_void_p), ("ldctx", C.c_int), ("lse", C.c_void_p),
        ("dctx", C.c_void_p), ("lddctx", C.c_int), ("delta", C.c_void_p), ("dqkv", C.c_void_p), ("lddqkv", C.c_int),
        ("colpart", C.c_void_p), ("colpart_accumulate", C.c_int),
        ("ctx8", C.c_void_p), ("ldctx8", C.c_int), ("ctx_scale", C.c_void_p), ("ctx_amax", C.c_void_p),
        ("dqkv8", C.c_void_p), ("lddqkv8", C.c_int), ("dqkv_scale", C.c_void_p), ("dqkv_amax", C.c_void_p),
        ("qoff", C.c_void_p), ("q", C.c_void_p), ("ldq", C.c_int), ("nq_total", C.c_int), ("dq", C.c_void_p), ("lddq", C.c_int),
        ("dq8", C.c_void_p), ("lddq8", C.c_int),
    ]


class PlbEmbed(C.Structure):
    _fields_ = [
        ("ids", C.c_void_p), ("T", C.c_int), ("S", C.c_int), ("E", C.c_int), ("V", C.c_int),
        ("word", C.c_void_p), ("pos", C.c_void_p), ("type0", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
        ("eps", C.c_float), ("out", C.c_void_p), ("ldo", C.c_int), ("dout", C.c_void_p), ("lddo", C.c_int),
        ("dx", C.c_void_p), ("dword", C.c_void_p), ("dpos", C.c_void_p), ("partials", C.c_void_p), ("nblocks", C.c_int),
    ]


class PlbLayerNorm(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
        ("y", C.c_void_p), ("ldy", C.c_int), ("mean", C.c_void_p), ("rstd", C.c_void_p),
        ("T", C.c_int), ("H", C.c_int), ("Tzero", C.c_int),
        ("dy", C.c_void_p), ("lddy", C.c_int), ("dx", C.c_void_p), ("lddx", C.c_int),
        ("partials", C.c_void_p), ("nblocks", C.c_int), ("accumulate", C.c_int),
        ("out8", C.c_void_p), ("ld8", C.c_int), ("q_scale", C.c_void_p), ("q_amax", C.c_void_p),
    ]


_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """The loaded library; raises HipLibraryMissing (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} is missing: build it with `python -m plbert_amd.build` (needs hipcc, gfx950). "
            "There is no CPU fallback for the PL-BERT hot path.")
    try:
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as ex:
        raise HipLibraryMissing(f"cannot load {LIB_PATH}: {ex}") from ex
    vp, i32, i64p, f32, f64 = C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_float, C.c_double
    L.plb_last_error.restype = C.c_char_p
    L.plb_last_error.argtypes = []
    L.plb_create.restype = C.c_int
    L.plb_create.argtypes = [C.POINTER(PlbConfig), C.POINTER(vp)]
    L.plb_destroy.restype = None
    L.plb_destroy.argtypes = [vp]
    L.plb_param_layout.restype = C.c_int
    L.plb_param_layout.argtypes = [vp, i64p, i64p, i64p, i64p]
    L.plb_workspace_bytes.restype = C.c_int64
    L.plb_workspace_bytes.argtypes = [vp]
    L.plb_bind.restype = C.c_int
    L.plb_bind.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int64]
    L.plb_sync_weights.restype = C.c_int
    L.plb_sync_weights.argtypes = [vp, vp]
    L.plb_forward.restype = C.c_int
    L.plb_forward.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.plb_loss_fwd_bwd.restype = C.c_int
    L.plb_loss_fwd_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.plb_loss_fwd_bwd_dual.restype = C.c_int
    L.plb_loss_fwd_bwd_dual.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.plb_pooler.restype = C.c_int
    L.plb_pooler.argtypes = [vp, vp, i32, i32, vp, vp]
    L.plb_loss_fwd.restype = C.c_int
    L.plb_loss_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.plb_set_fp8.restype = C.c_int
    L.plb_set_fp8.argtypes = [vp, i32, vp]
    L.plb_fp8_state.restype = C.c_int
    L.plb_fp8_state.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.plb_token_head_steps.restype = i32
    L.plb_token_head_steps.argtypes = [vp]
    L.plb_set_token_head_steps.restype = C.c_int
    L.plb_set_token_head_steps.argtypes = [vp, i32]
    L.plb_comm_unique_id.restype = C.c_int
    L.plb_comm_unique_id.argtypes = [vp]
    L.plb_comm_init.restype = C.c_int
    L.plb_comm_init.argtypes = [vp, vp, i32, i32]
    L.plb_comm_destroy.restype = C.c_int
    L.plb_comm_destroy.argtypes = [vp]
    L.plb_comm_info.restype = C.c_int
    L.plb_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.plb_status.restype = C.c_int
    L.plb_status.argtypes = [vp, C.POINTER(i32)]
    L.plb_status_ex.restype = C.c_int
    L.plb_status_ex.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.plb_poll_status.restype = C.c_int
    L.plb_poll_status.argtypes = [vp, C.POINTER(i32)]
    L.plb_fp8_stats.restype = C.c_int
    L.plb_fp8_stats.argtypes = [vp, C.POINTER(f32), C.POINTER(f32), vp]
    L.plb_last_application_rows.restype = C.c_int
    L.plb_last_application_rows.argtypes = [vp, i64p, i64p]
    L.plb_status_export.restype = C.c_int
    L.plb_status_export.argtypes = [vp, vp, vp]
    L.plb_status_import.restype = C.c_int
    L.plb_status_import.argtypes = [vp, vp, vp]
    L.plb_debug_hb_audit.restype = C.c_int
    L.plb_debug_hb_audit.argtypes = [vp, i32, i32]
    L.plb_debug_hb_report.restype = C.c_int
    L.plb_debug_hb_report.argtypes = [vp, i64p, C.POINTER(i32), C.c_char_p, i32]
    L.plb_comm_trace.restype = C.c_int
    L.plb_comm_trace.argtypes = [vp, i32]
    L.plb_comm_trace_read.restype = C.c_int
    L.plb_comm_trace_read.argtypes = [vp, i32, C.POINTER(i32), i64p, i64p, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]
    L.plb_debug_ln_fault.restype = None
    L.plb_debug_ln_fault.argtypes = [C.c_int, C.c_int]
    L.plb_debug_skip_piece.restype = None
    L.plb_debug_skip_piece.argtypes = [C.c_int]
    L.plb_comm_pieces.restype = C.c_int
    L.plb_comm_pieces.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_int64)]
    L.plb_broadcast_params.restype = C.c_int
    L.plb_broadcast_params.argtypes = [vp, i32, vp]
    L.plb_set_grad_overlap.restype = C.c_int
    L.plb_set_grad_overlap.argtypes = [vp, i32]
    L.plb_allreduce_grads.restype = C.c_int
    L.plb_allreduce_grads.argtypes = [vp, vp]
    L.plb_apply_mask.restype = C.c_int
    L.plb_apply_mask.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int64, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.plb_adamw_step.restype = C.c_int
    L.plb_adamw_step.argtypes = [vp, f64, f64, f64, f64, f64, i32, f64, vp]
    L.plb_mask_batch.restype = C.c_int
    L.plb_mask_batch.argtypes = [vp, vp, i32, i32, C.c_uint64, C.c_uint32, f32, f32, f32, i32, i32, vp, vp, vp, vp, vp]
    L.plb_profile_enable.restype = None
    L.plb_profile_enable.argtypes = [C.c_int]
    L.plb_profile_num_classes.restype = C.c_int
    L.plb_profile_num_classes.argtypes = []
    L.plb_profile_class_name.restype = C.c_char_p
    L.plb_profile_class_name.argtypes = [C.c_int]
    L.plb_profile_read.restype = C.c_int
    L.plb_profile_read.argtypes = [C.POINTER(C.c_double), i64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    # internal launchers (kernel-level tests)
    L.plb_launch_gemm_nt.restype = C.c_int
    L.plb_launch_gemm_nt.argtypes = [C.POINTER(PlbGemmNT), C.c_int, C.c_int, vp]
    L.plb_set_gemm_nt_tile.restype = None
    L.plb_set_gemm_nt_tile.argtypes = [C.c_int]
    L.plb_set_gemm_nt_prefetch.restype = None
    L.plb_set_gemm_nt_prefetch.argtypes = [C.c_int]
    L.plb_launch_gemm_nt_fp8.restype = C.c_int
    L.plb_launch_gemm_nt_fp8.argtypes = [C.POINTER(PlbGemmNT), C.c_int, C.c_int, vp]
    L.plb_launch_gemm_nt_fp8_ln.restype = C.c_int
    L.plb_launch_gemm_nt_fp8_ln.argtypes = [C.POINTER(PlbGemmNT), C.c_int, C.c_int, vp]
    L.plb_launch_gemm_nt_fp8_gelud.restype = C.c_int
    L.plb_launch_gemm_nt_fp8_gelud.argtypes = [C.POINTER(PlbGemmNT), C.c_int, C.c_int, vp]
    L.plb_launch_gemm_tn.restype = C.c_int
    L.plb_launch_gemm_tn.argtypes = [C.POINTER(PlbGemmTN), vp]
    L.plb_launch_gemm_tn_big.restype = C.c_int
    L.plb_launch_gemm_tn_big.argtypes = [C.POINTER(PlbGemmTN), vp]
    L.plb_launch_gemm_tn_fp8.restype = C.c_int
    L.plb_launch_gemm_tn_fp8.argtypes = [C.POINTER(PlbGemmTN), vp]
    L.plb_launch_reduce_slabs.restype = C.c_int
    L.plb_launch_reduce_slabs.argtypes = [vp, C.c_int, C.c_size_t, vp, C.c_int, vp]
    L.plb_launch_attn_fwd.restype = C.c_int
    L.plb_launch_attn_fwd.argtypes = [C.POINTER(PlbAttn), vp]
    L.plb_launch_gemm_nt_ln.restype = C.c_int
    L.plb_launch_gemm_nt_ln.argtypes = [C.POINTER(PlbGemmNT), C.c_int, vp]
    L.plb_launch_gemm_nt_gelud.restype = C.c_int
    L.plb_launch_gemm_nt_gelud.argtypes = [C.POINTER(PlbGemmNT), C.c_int, vp]
    L.plb_launch_attn_bwd.restype = C.c_int
    L.plb_launch_attn_bwd.argtypes = [C.POINTER(PlbAttn), vp]
    L.plb_launch_attn_bwd_fused.restype = C.c_int
    L.plb_launch_attn_bwd_fused.argtypes = [C.POINTER(PlbAttn), vp]
    L.plb_set_prune_last.restype = None
    L.plb_set_prune_last.argtypes = [C.c_int]
    L.plb_set_attn_bwd_fused.restype = None
    L.plb_set_attn_bwd_fused.argtypes = [C.c_int]
    L.plb_launch_ln_fwd.restype = C.c_int
    L.plb_launch_ln_fwd.argtypes = [C.POINTER(PlbLayerNorm), vp]
    L.plb_launch_ln_bwd.restype = C.c_int
    L.plb_launch_ln_bwd.argtypes = [C.POINTER(PlbLayerNorm), vp]
    L.plb_launch_embed_scatter.restype = C.c_int
    L.plb_launch_embed_scatter.argtypes = [C.POINTER(PlbEmbed), C.c_int, vp]
    L.plb_launch_colsum.restype = C.c_int
    L.plb_launch_colsum.argtypes = [vp, C.c_int, C.c_size_t, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp]
    _lib = L
    return L


def profile_enable(on):
    lib().plb_profile_enable(int(bool(on)))


def profile_read():
    """{class name: dict(ms, launches, flops, bytes)} for the launches recorded since the last read."""
    L = lib()
    n = L.plb_profile_num_classes()
    ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    cnt = (C.c_int64 * n)()
    check(L.plb_profile_read(ms, cnt, fl, by), "plb_profile_read")
    return {L.plb_profile_class_name(i).decode(): dict(ms=ms[i], launches=int(cnt[i]), flops=fl[i], bytes=by[i])
            for i in range(n) if cnt[i]}


def check(rc, what):
    if rc != 0:
        msg = lib().plb_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed: {msg or 'rc=%d' % rc}")

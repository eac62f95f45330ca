"""ctypes binding of libxpretrain_hip.so (the C ABI in include/xpretrain_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C xpretrain_amd/csrc``.
Loading fails loudly if it is missing -- there is no fallback path.
"""
import ctypes as C
import os
import sys

# PyTorch must be imported BEFORE the library is dlopen'ed: torch bundles its own HIP runtime
# (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  Loaded first, it satisfies our DT_NEEDED by soname and
# the process has ONE runtime / one set of streams; loaded second, torch would bring a second runtime whose
# streams are not ordered with ours.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxpretrain_hip.so")

XP_BF16, XP_F32 = 0, 1
(EPI_NONE, EPI_BIAS, EPI_BIAS_QSCALE, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_GELU_BWD, EPI_PATCH, EPI_SCALE) = range(8)
ATTN_PROXY, ATTN_CAUSAL = 0, 1

i32, i64, f32, vp, sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class XpGemmDesc(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp),
        ("M", i64), ("N", i64), ("K", i64),
        ("lda", i64), ("ldb", i64), ("ldc", i64),
        ("a_kstrided", i32), ("b_kstrided", i32),
        ("in_dtype", i32), ("out_dtype", i32),
        ("epilogue", i32), ("split_k", i32),
        ("a_grp", i64), ("a_grp_stride", i64), ("a_off", i64),
        ("c_grp", i64), ("c_grp_stride", i64), ("c_off", i64),
        ("bias", vp),
        ("scale", f32), ("scale_cols", i64),
        ("resid", vp), ("ldr", i64),
        ("aux", vp), ("ldaux", i64),
        ("tab1", vp), ("tab2", vp), ("tab_L", i64),
        ("colsum_partials", vp),
        ("reserved0", i32), ("side_M", i32),
        ("resid_side", vp), ("out_side", vp), ("side_S", i64),
        ("a_frames", vp), ("a_frames_u8", i32), ("fr_H", i32), ("fr_W", i32), ("fr_P", i32),
        ("fr_mean", f32 * 3), ("fr_std", f32 * 3),
    ]


class XpReduceSeg(C.Structure):
    _fields_ = [("in_", vp), ("out", vp), ("stride", i64), ("nrows", i32), ("width", i32), ("accumulate", i32),
                ("reserved", i32)]


XP_REDUCE_MAX_SEGS = 16


class XpLayerDims(C.Structure):
    _fields_ = [("rows", i64), ("D", i64), ("Dff", i64), ("B", i64), ("S", i64), ("heads", i64),
                ("M", i64), ("N", i64), ("L", i64), ("attn_mode", i32), ("dtype", i32), ("q_scale", f32), ("ln_eps", f32)]


class XpLayerFwd(C.Structure):
    _fields_ = ([("dims", XpLayerDims)]
                + [(n, vp) for n in ("x", "Wqkv", "Wo", "W1", "W2", "ln1_w", "ln1_b", "bqkv", "bo", "ln2_w", "ln2_b", "b1", "b2",
                                     "pad_mask", "h1", "qkv", "attn_o", "x2", "h2", "pre", "act", "x3", "mean1", "rstd1",
                                     "mean2", "rstd2", "stats", "workspace")]
                + [("workspace_bytes", sz), ("side_in", vp), ("side_out", vp), ("side_S", i64), ("side_M", i32), ("reserved", i32),
                   ("side_x2", vp)])


class XpLayerBwd(C.Structure):
    _fields_ = ([("dims", XpLayerDims)]
                + [(n, vp) for n in ("x", "h1", "qkv", "attn_o", "x2", "h2", "pre", "act", "Wqkv", "Wo", "W1", "W2", "ln1_w",
                                     "ln2_w", "mean1", "rstd1", "mean2", "rstd2", "stats", "pad_mask", "dx3", "dx",
                                     "dln1_w", "dln1_b", "dwqkv", "dbqkv", "dwo", "dbo", "dln2_w", "dln2_b", "dw1", "db1",
                                     "dw2", "db2", "workspace")]
                + [("workspace_bytes", sz), ("side_in", vp), ("side_x2", vp), ("side_S", i64), ("side_M", i32), ("reserved", i32)])


class XpAdamTensor(C.Structure):
    _fields_ = [("p", vp), ("m", vp), ("v", vp), ("shadow", vp), ("numel", i64), ("shadow_dtype", i32), ("reserved", i32)]


class XpAdamGroup(C.Structure):
    _fields_ = [("lr", f32), ("beta1", f32), ("beta2", f32), ("eps", f32), ("weight_decay", f32), ("step_size", f32)]


XP_OPT_CHUNK, XP_OPT_MAX_TENSORS, XP_OPT_MAX_GROUPS = 65536, 256, 16

# name -> (restype, argtypes); mirrors include/xpretrain_hip.h one-to-one
SIGNATURES = {
    "xp_abi_version": (i32, []),
    "xp_last_error": (C.c_char_p, []),
    "xp_gemm": (i32, [C.POINTER(XpGemmDesc), vp]),
    "xp_gemm_auto_split": (i32, [C.POINTER(XpGemmDesc)]),
    "xp_gemm_auto_split_slack": (i32, [C.POINTER(XpGemmDesc)]),
    "xp_set_cu_budget": (i32, [i32]),
    "xp_get_cu_budget": (i32, []),
    "xp_gemm_colsum_rows": (i64, [C.POINTER(XpGemmDesc)]),
    "xp_gemm_tile_rows": (i32, [C.POINTER(XpGemmDesc)]),
    "xp_colsum_partial_rows": (i64, [i64, i64]),
    "xp_colsum_partials": (i32, [vp, i64, i64, i64, i32, vp, sz, vp]),
    "xp_reduce_rows_batch_workspace_bytes": (sz, [C.POINTER(XpReduceSeg), i32]),
    "xp_reduce_rows_batch": (i32, [C.POINTER(XpReduceSeg), i32, vp, sz, vp]),
    "xp_layernorm_bwd_partial_rows": (i64, [i64]),
    "xp_layernorm_bwd_partials": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, i32, i64, i64, i32, vp, sz, vp]),
    "xp_grad_sqnorm_partials": (i32, [vp, vp, i32, C.POINTER(vp), i32, vp, vp]),
    "xp_adamw_step": (i32, [vp, vp, i32, C.POINTER(vp), C.POINTER(C.c_uint8), i32, C.POINTER(XpAdamGroup), i32, vp, i32, f32,
                            vp, vp]),
    "xp_splitk_reduce": (i32, [vp, vp, i64, i32, i32, vp]),
    "xp_colsum_workspace_bytes": (sz, [i64, i64]),
    "xp_colsum": (i32, [vp, i64, i64, i64, i32, vp, i32, vp, sz, vp]),
    "xp_layernorm_fwd": (i32, [vp, i64, vp, vp, vp, i64, vp, vp, i64, i64, f32, i32, vp]),
    "xp_layernorm_fwd_side": (i32, [vp, i64, vp, vp, vp, i64, vp, vp, i64, i64, f32, i32, vp, vp, i64, i32, i32, vp]),
    "xp_layernorm_bwd_workspace_bytes": (sz, [i64, i64]),
    "xp_layernorm_bwd_partials_side": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, i32, i64, i64, i32, vp, i64, i32, i32,
                                             vp, sz, vp]),
    "xp_layernorm_bwd_side": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp, i32, i64, i64, i32, vp, i64, i32, i32,
                                    vp, sz, vp]),
    "xp_layernorm_bwd": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp, i32, i64, i64, i32, vp, sz, vp]),
    "xp_attn_workspace_bytes": (sz, [i32, i64, i64, i64, i64, i64]),
    "xp_attn_fwd": (i32, [vp, i64, vp, i64, vp, vp, i32, i64, i64, i64, i64, i64, i64, i32, vp, sz, vp]),
    "xp_attn_bwd": (i32, [vp, i64, vp, vp, i64, vp, vp, vp, f32, i32, i64, i64, i64, i64, i64, i64, i32, vp, sz, vp]),
    "xp_attn_bwd2": (i32, [vp, i64, vp, vp, i64, vp, vp, vp, f32, i32, i64, i64, i64, i64, i64, i64, i32, vp, sz, vp, vp]),
    "xp_attn_bwd_colsum_rows": (i64, [i32, i64, i64, i64, i64, i64, i64, i32]),
    "xp_im2col": (i32, [vp, vp, i64, i64, i64, i64, i32, vp]),
    "xp_im2col_u8": (i32, [vp, C.POINTER(f32), C.POINTER(f32), vp, i64, i64, i64, i64, i32, vp]),
    "xp_vip_proxy_rows": (i32, [vp, vp, vp, vp, i64, i64, i64, i64, i32, vp]),
    "xp_vip_embed_bwd_workspace_bytes": (sz, [i64, i64, i64, i64]),
    "xp_vip_embed_bwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i32, i32, vp, sz, vp]),
    "xp_text_embed_fwd": (i32, [vp, vp, vp, vp, i64, i64, i64, i64, i32, vp]),
    "xp_text_embed_bwd": (i32, [vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, vp]),
    "xp_argmax_rows": (i32, [vp, vp, i64, i64, vp]),
    "xp_gather_rows": (i32, [vp, vp, vp, i64, i64, i64, i32, vp]),
    "xp_scatter_rows": (i32, [vp, vp, vp, i64, i64, i64, i32, vp]),
    "xp_l2norm_fwd": (i32, [vp, vp, vp, i64, i64, i32, vp]),
    "xp_l2norm_bwd": (i32, [vp, vp, vp, vp, i64, i64, i32, vp]),
    "xp_cast": (i32, [vp, vp, i64, i32, vp]),
    "xp_cast_back": (i32, [vp, vp, i64, i32, i32, vp]),
    "xp_nce_loss_workspace_bytes": (sz, [i64, i64]),
    "xp_nce_loss": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i64, vp, sz, vp]),
    "xp_sim_matrix": (i32, [vp, vp, vp, i64, i64, i64, vp]),
    "xp_dsl_rerank": (i32, [vp, i64, i64, f32, i32, vp]),
    "xp_retrieval_ranks": (i32, [vp, vp, i64, i64, i32, vp, vp, vp]),
    "xp_vsc_fc_loss_workspace_bytes": (sz, [i64, i64]),
    "xp_vsc_fc_loss": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, vp, sz, vp]),
    "xp_encoder_layer_fwd_workspace_bytes": (sz, [C.POINTER(XpLayerDims)]),
    "xp_encoder_layer_fwd": (i32, [C.POINTER(XpLayerFwd), vp]),
    "xp_encoder_layer_bwd_workspace_bytes": (sz, [C.POINTER(XpLayerDims)]),
    "xp_encoder_layer_bwd": (i32, [C.POINTER(XpLayerBwd), vp]),
    "xp_side_stream": (vp, []),
    "xp_debug_set_gemm_trace": (i32, [vp]),
    "xp_debug_gemm_timer_arm": (i32, [i64, i64, i64, i32, i32, i32, i32, i32]),
    "xp_debug_gemm_timer_read": (i32, [C.POINTER(C.c_float), i32]),
    "xp_debug_set_attn_trace": (i32, [vp]),
    "xp_debug_gemm_occupancy": (i32, [i32]),
    "xp_probe_mfma_bf16": (i32, [vp, vp, vp, vp]),
    "xp_probe_mfma_f32": (i32, [vp, vp, vp, vp]),
    "xp_probe_tr16": (i32, [vp, vp, vp, vp]),
    "xp_probe_stream_copy": (i32, [vp, vp, i64, i32, i32, vp]),
    "xp_probe_stream_copy_fat": (i32, [vp, vp, i64, i32, i32, vp]),
    "xp_probe_pk_f32": (i32, [vp, i32, i32, C.c_uint32, vp]),
}

_lib = None


def lib():
    """The loaded library (raises RuntimeError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C xpretrain_amd/csrc`.  xpretrain_amd has no CPU / eager fallback.")
        l = C.CDLL(LIB_PATH)
        missing = [n for n in SIGNATURES if not hasattr(l, n)]
        if missing:
            raise RuntimeError(f"{LIB_PATH} is missing C-ABI symbols: {missing}")
        for n, (res, args) in SIGNATURES.items():
            fn = getattr(l, n)
            fn.restype, fn.argtypes = res, args
        if l.xp_abi_version() != 1:
            raise RuntimeError(f"ABI version mismatch: library {l.xp_abi_version()} != binding 1")
        _lib = l
    return _lib


# XPRETRAIN_DEBUG=flag[,flag...]: the debug / test facilities (csrc/common.cpp lists the library's flags).  Python side:
#   sync      synchronise and check the device after every library call (errors then surface at the call that caused them)
#   op_by_op  encoder layers as separate operator calls instead of one C-ABI call per pass (the fingerprinting / calibration tools)
#   no_comm   GradBucketReducer packs and tracks its buckets but skips the all-reduce calls (cost attribution on one GPU)
DEBUG = frozenset(f for f in os.environ.get("XPRETRAIN_DEBUG", "").split(",") if f)
DEBUG_SYNC = "sync" in DEBUG


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().xp_last_error().decode()}")
    if DEBUG_SYNC:      # debugging aid: localise an asynchronous fault to the call that caused it
        sys.stderr.write(f"[xp] {what} ... "); sys.stderr.flush()
        torch.cuda.synchronize()
        sys.stderr.write("ok\n"); sys.stderr.flush()

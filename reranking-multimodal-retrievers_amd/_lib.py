"""ctypes binding of librerank_mi355.so (include/rerank_mi355.h).

The library is the product; this module only declares its C ABI.  There is no Python or
CPU fallback: if the shared library is missing or a call fails, we raise.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: the process-wide HIP runtime is torch's libamdhip64)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librerank_mi355.so")

RR_ABI_VERSION = 2
RR_OK, RR_ERR_BAD_ARG, RR_ERR_BAD_SHAPE, RR_ERR_BAD_DTYPE, RR_ERR_UNSUPPORTED = 0, -1, -2, -3, -4
RR_ERR_HIP, RR_ERR_OOM, RR_ERR_MISSING_WEIGHT, RR_ERR_NO_DEVICE, RR_ERR_RANGE = -5, -6, -7, -8, -9
RR_F32, RR_BF16, RR_F16 = 0, 1, 2
LOSS_KINDS = {"BCE": 0, "2H_BCE": 1, "negative_sampling": 2}
COMPUTE_DTYPES = {"bf16": 0, "fp16": 1}
MODEL_KINDS = {"full_context": 0, "interaction": 1, "mores": 2}
KERNEL_CLASSES = ["gemm", "attention", "layernorm", "embed", "tail", "head", "gemm_fp8"]


class RRConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("abi_version", "vocab_size", "hidden", "layers", "heads", "intermediate", "max_pos", "type_vocab")] + \
               [("ln_eps", C.c_float), ("li_dim", C.c_int32)] + \
               [(n, C.c_int32) for n in ("ce_hidden", "ce_layers", "ce_heads", "ce_intermediate", "ce_max_pos",
                                         "has_vision", "vision_hidden", "prefix_len", "n_patches", "map_layers",
                                         "cross_attn_len", "loss_kind")] + \
               [("pos_weight", C.c_float), ("device", C.c_int32), ("compute_dtype", C.c_int32),
                ("model_kind", C.c_int32), ("vit_layers", C.c_int32), ("vit_heads", C.c_int32),
                ("vit_intermediate", C.c_int32), ("vit_image_size", C.c_int32), ("vit_patch_size", C.c_int32),
                ("fp8", C.c_int32)]


class RRProfile(C.Structure):
    _fields_ = [("ms", C.c_double * 7), ("launches", C.c_int64 * 7), ("flops", C.c_double * 7),
                ("bytes", C.c_double * 7)]


# exceptions mirror the reference's Python error behaviour (include/rerank_mi355.h rr_status comments)
_EXC = {RR_ERR_BAD_ARG: ValueError, RR_ERR_BAD_SHAPE: AssertionError, RR_ERR_BAD_DTYPE: ValueError,
        RR_ERR_UNSUPPORTED: NotImplementedError, RR_ERR_HIP: RuntimeError, RR_ERR_OOM: MemoryError,
        RR_ERR_MISSING_WEIGHT: KeyError, RR_ERR_NO_DEVICE: RuntimeError, RR_ERR_RANGE: OverflowError}

_P = C.c_void_p
_SIGS = {
    "rr_version": (C.c_char_p, []),
    "rr_status_string": (C.c_char_p, [C.c_int]),
    "rr_create": (C.c_int, [C.POINTER(RRConfig), C.POINTER(_P)]),
    "rr_destroy": (C.c_int, [_P]),
    "rr_last_error": (C.c_char_p, [_P]),
    "rr_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "rr_finalize_weights": (C.c_int, [_P]),
    "rr_num_required_weights": (C.c_int, [_P]),
    "rr_required_weight_name": (C.c_char_p, [_P, C.c_int]),
    "rr_workspace_bytes": (C.c_int64, [_P, C.c_int, C.c_int]),
    "rr_set_padded_seq_len": (C.c_int, [_P, C.c_int]),
    "rr_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "rr_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int)]),
    "rr_reserve": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "rr_activation_range_flag": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), _P]),
    "rr_forward_packed": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, _P, _P, C.c_int, _P, _P, _P]),
    "rr_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                             _P, _P, _P, _P, _P, _P]),
    "rr_encode_image": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "rr_forward_joint": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                   _P, _P, _P, _P, _P, _P]),
    "rr_forward_joint_fusion": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                          C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "rr_forward_interaction": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                         _P, _P, _P, _P, _P, _P]),
    "rr_forward_interaction_fusion": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, _P,
                                                C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "rr_head": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "rr_debug_read": (C.c_int64, [_P, C.c_char_p, _P, C.c_int64]),
    "rr_set_debug": (C.c_int, [_P, C.c_int]),
    "rr_set_profiling": (C.c_int, [_P, C.c_int]),
    "rr_get_profile": (C.c_int, [_P, C.POINTER(RRProfile), C.c_int]),
    "rr_op_gemm_bf16": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rr_op_gemm_resid_f32": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rr_op_attention_bf16": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, _P, C.c_int, _P]),
    "rr_set_gemm_variant": (C.c_int, [C.c_int]),
    "rr_set_op_dtype": (C.c_int, [C.c_int]),
    "rr_op_quantize_fp8": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rr_op_amax": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
    "rr_op_gemm_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rr_op_gemm_fp8_rc": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rr_op_gemm_fp8_gelu_e4m3": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rr_op_gemm_fp8_resid": (C.c_int, [_P, _P, _P, C.c_float, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rr_op_layernorm_q8": (C.c_int, [_P, _P, _P, C.c_float, C.c_int, C.c_int, _P, _P, _P, _P]),
    "rr_util_quantize_rows_e4m3": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "rr_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "rr_set_attn_stamps": (C.c_int, [_P]),
    "rr_set_attn_redo_stats": (C.c_int, [_P]),
    "rr_tok_create": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rr_tok_destroy": (C.c_int, [C.c_void_p]),
    "rr_tok_encode": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, _P, C.c_int]),
    "rr_tok_decode": (C.c_int, [C.c_void_p, _P, C.c_int, C.c_char_p, C.c_int]),
    "rr_tok_prepare_pairs": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, _P, _P, _P]),
    "rr_op_layernorm_stats": (C.c_int, [_P, _P, _P, C.c_float, C.c_int, C.c_int, _P, _P, _P, _P]),
    "rr_op_gemm_ln_resid_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rr_op_gemm_resid_lnprep": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P, _P]),
    "rr_op_gemm_resid_split": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P, _P]),
    "rr_op_gemm_lnfold": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "rr_op_split_residual_value": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P]),
    "rr_set_gemm_stamps": (C.c_int, [_P]),
    "rr_set_gemm_stagger": (C.c_int, [C.c_int]),
    "rr_op_layernorm": (C.c_int, [_P, _P, _P, C.c_float, C.c_int, C.c_int, _P, _P, _P]),
}
EXPORTED = sorted(_SIGS)

_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(status: int, handle=None, what: str = ""):
    if status == RR_OK:
        return
    lib = load()
    msg = lib.rr_last_error(handle).decode() if handle is not None else ""
    base = lib.rr_status_string(status).decode()
    raise _EXC.get(status, RuntimeError)(f"{what}: {base}: {msg}" if what else f"{base}: {msg}")


def ptr(t) -> int:
    """Device (or host) address of a contiguous torch tensor, 0 for None."""
    if t is None:
        return 0
    assert t.is_contiguous(), "librerank_mi355 takes contiguous tensors"
    return t.data_ptr()

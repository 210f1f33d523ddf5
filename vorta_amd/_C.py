"""ctypes binding of libvorta_hip.so (include/vorta_hip.h).  No torch types cross this boundary.

The library is the product: if it is missing or does not load, importing callers get a RuntimeError --
there is no CPU or PyTorch fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VORTA_HIP_LIB: experiments only (A/B of two builds in one session); the product is the in-tree library
LIB_PATH = os.environ.get("VORTA_HIP_LIB") or os.path.join(_HERE, "csrc", "libvorta_hip.so")

VORTA_OK, VORTA_EINVAL, VORTA_EUNSUPPORTED, VORTA_ELAUNCH = 0, -1, -2, -3
VORTA_BF16, VORTA_FP16, VORTA_FP32, VORTA_FP8E4M3, VORTA_INT8 = 0, 1, 2, 3, 4
ABI_VERSION = 8

_i32, _i64, _u32, _f32, _vp = C.c_int32, C.c_int64, C.c_uint32, C.c_float, C.c_void_p


class Tensor(C.Structure):
    _fields_ = [("ptr", _vp), ("stride_h", _i64), ("stride_s", _i64)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("dtype", _i32), ("head_dim", _i32), ("n_heads", _i32),
        ("q", Tensor), ("k", Tensor), ("v", Tensor), ("o", Tensor),
        ("head_list", _vp), ("n_heads_dev", _vp),
        ("n_q", _i32), ("q_group_len", _i32), ("q_row_offset", _i32), ("q_valid", _i32),
        ("q_rows", _vp), ("q_rows_stride_h", _i64),
        ("n_kv", _i32), ("kv_row_offset", _i32),
        ("kv_rows", _vp), ("kv_rows_stride_h", _i64), ("kv_rows_stride_g", _i64),
        ("dup_rows", _vp), ("dup_rows_stride_h", _i64),
        ("n_dup_pos", _i32), ("n_dup", _i32),
        ("scale", _f32), ("block_rows", _i32), ("n_splits", _i32),
        ("ws_o", _vp), ("ws_ml", _vp),
        ("n_kv_dev", _vp), ("q_valid_dev", _vp),
        ("variant", _i32), ("reserved", _i32),
        ("q_block_table", _vp), ("n_q_blocks", _i32), ("reserved2", _i32),
    ]


class CoresetArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("dtype", _i32), ("head_dim", _i32), ("n_heads", _i32),
        ("x", Tensor), ("head_list", _vp), ("n_heads_dev", _vp),
        ("latent", _i32 * 3), ("group", _i32 * 3),
        ("n_keep", _i32), ("tail_first", _i32), ("n_tail", _i32),
        ("row_map", _vp),
        ("keep_rows", _vp), ("keep_rows_stride_h", _i64),
        ("drop_rows", _vp), ("drop_rows_stride_h", _i64),
        ("keep_rows_kv", _vp), ("keep_rows_kv_stride_h", _i64),
    ]


class StaArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("latent", _i32 * 3), ("tile", _i32 * 3), ("window", _i32 * 3),
        ("t_eff", _i32), ("row_map", _vp), ("q_rows", _vp), ("kv_rows", _vp),
    ]


class RouterArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("dtype", _i32),
        ("batch", _i32), ("embed_dim", _i32), ("heads", _i32), ("n_experts", _i32),
        ("temb", _vp), ("weight", _vp), ("bias", _vp),
        ("tau", _f32),
        ("scores", _vp), ("expert_of_head", _vp), ("head_lists", _vp), ("head_counts", _vp),
        ("ws_logits", _vp),
    ]


class NormRopeArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("dtype", _i32), ("head_dim", _i32), ("heads", _i32),
        ("x", Tensor), ("weight", _vp), ("cos", _vp), ("sin", _vp),
        ("n_tokens", _i32), ("token_offset", _i32), ("rope_tokens", _i32),
        ("eps", _f32), ("across_heads", _i32),
    ]


class MixArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("dtype", _i32), ("head_dim", _i32), ("heads", _i32), ("n_experts", _i32),
        ("n_rows", _i32), ("x", Tensor * 3), ("out", Tensor), ("scores", _vp),
    ]


class PermuteArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("dtype", _i32), ("head_dim", _i32), ("heads", _i32), ("n_rows", _i32),
        ("n_tensors", _i32), ("src", Tensor * 4), ("dst", Tensor * 4), ("src_map", _vp), ("dst_map", _vp),
    ]


class Fp8QuantArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("dtype", _i32), ("head_dim", _i32), ("heads", _i32),
        ("n_tokens", _i32), ("qk_scale", _f32),
        ("q", Tensor), ("k", Tensor), ("v", Tensor), ("q8", Tensor), ("k8", Tensor), ("v8", Tensor),
        ("v_descale", _vp), ("ws", _vp), ("flags", _i32), ("seg_len", _i32), ("tail_first", _i32), ("tail_len", _i32),
        ("slot_first", _i32), ("slot_count", _i32),
        ("video_tokens", _i32), ("token_offset", _i32), ("total_tokens", _i32), ("src_map", _vp),
    ]


class Fp8VArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("dtype", _i32), ("head_dim", _i32), ("heads", _i32), ("n_tokens", _i32), ("flags", _i32),
        ("v", Tensor), ("v8", Tensor), ("src_map", _vp), ("amax", _vp), ("v_descale", _vp),
    ]


class AttnFp8Ext(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("out_dtype", _i32), ("v_descale", _vp), ("v_descale_stride_h", _i64),
        ("p_bias", _f32), ("defer", _f32), ("flags", _i32), ("reserved", _i32),
    ]


class I8QuantArgs(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("dtype", _i32), ("head_dim", _i32), ("heads", _i32), ("n_tokens", _i32),
        ("q", Tensor), ("k", Tensor), ("k8", Tensor), ("k_bias", _vp), ("k_bias_stride_h", _i64),
        ("q_prep", _vp), ("k_head_scale", _vp), ("ws", _vp), ("flags", _i32), ("seg_len", _i32), ("tail_first", _i32),
        ("tail_len", _i32), ("slot_first", _i32), ("slot_count", _i32),
    ]


class AttnI8Ext(C.Structure):
    _fields_ = [
        ("struct_size", _u32), ("flags", _i32), ("k_bias", _vp), ("k_bias_stride_h", _i64),
        ("q_prep", _vp), ("q_prep_stride_h", _i64), ("k_head_scale", _vp), ("v_descale", _vp), ("v_descale_stride_h", _i64),
        ("p_bias", _f32), ("defer", _f32),
    ]


# every symbol include/vorta_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "vorta_attn_fwd": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "vorta_attn_fwd_batch": (C.c_int, [C.POINTER(AttnArgs), _i32, _vp]),
    "vorta_attn_plan": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32)]),
    "vorta_attn_workspace_bytes": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vorta_fp8_quant_ws_floats": (C.c_int, [_i32, _i32]),
    "vorta_fp8_quant_ws_partials": (C.c_int, [_i32, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "vorta_fp8_quantize_qkv": (C.c_int, [C.POINTER(Fp8QuantArgs), _vp]),
    "vorta_fp8_v_absmax": (C.c_int, [C.POINTER(Fp8VArgs), _vp]),
    "vorta_fp8_v_convert": (C.c_int, [C.POINTER(Fp8VArgs), _vp]),
    "vorta_attn_fwd_fp8": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(AttnFp8Ext), _vp]),
    "vorta_attn_fwd_batch_fp8": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(AttnFp8Ext), _i32, _vp]),
    "vorta_i8_quantize_k": (C.c_int, [C.POINTER(I8QuantArgs), _vp]),
    "vorta_i8_tail_flags": (C.c_int, [C.POINTER(Tensor), _i32, _i32, _vp, C.c_float, _vp, _vp]),
    "vorta_split_heads": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "vorta_attn_fwd_i8": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(AttnI8Ext), _vp]),
    "vorta_attn_fwd_batch_i8": (C.c_int, [C.POINTER(AttnArgs), C.POINTER(AttnI8Ext), _i32, _vp]),
    "vorta_coreset_select": (C.c_int, [C.POINTER(CoresetArgs), _vp]),
    "vorta_sta_table_sizes": (C.c_int, [C.POINTER(StaArgs), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "vorta_sta_build_tables": (C.c_int, [C.POINTER(StaArgs), _vp]),
    "vorta_router_route": (C.c_int, [C.POINTER(RouterArgs), _vp]),
    "vorta_route_scores": (C.c_int, [C.POINTER(RouterArgs), _vp]),
    "vorta_route_plan": (C.c_int, [C.POINTER(RouterArgs), C.c_int32, _vp]),
    "vorta_qk_norm_rope": (C.c_int, [C.POINTER(NormRopeArgs), _vp]),
    "vorta_mix_experts": (C.c_int, [C.POINTER(MixArgs), _vp]),
    "vorta_seq_row_map": (C.c_int, [_vp, _i32, _i32, _i32, _vp]),
    "vorta_permute_heads": (C.c_int, [C.POINTER(PermuteArgs), _vp]),
    "vorta_abi_version": (C.c_int, []),
    "vorta_build_info": (C.c_char_p, []),
    "vorta_last_hip_error": (C.c_int, []),
    "vorta_sizeof": (C.c_int, [C.c_int]),
}

_lib = None


class VortaHipError(RuntimeError):
    pass


def lib():
    """Load libvorta_hip.so once; raise loudly if it is absent (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VortaHipError(
            f"{LIB_PATH} not found: build it with `python -m vorta_amd.build` (hipcc --offload-arch=gfx950). "
            "The routed attention path has no CPU/PyTorch fallback.")
    # Load order matters: PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7) but links it by
    # file name, so if this library pulled /opt/rocm's copy in first the process would end up with two HIP
    # runtimes (launches then fail with hipErrorNoDevice).  Importing torch first makes our NEEDED entry
    # resolve to the runtime torch already loaded -- one runtime, shared streams and allocations.
    import torch  # noqa: F401
    try:
        h = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise VortaHipError(f"could not load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(h, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if h.vorta_abi_version() != ABI_VERSION:
        raise VortaHipError(f"ABI mismatch: library {h.vorta_abi_version()} vs binding {ABI_VERSION}")
    for which, st in enumerate((Tensor, AttnArgs, CoresetArgs, StaArgs, RouterArgs, NormRopeArgs, MixArgs, Fp8QuantArgs,
                                AttnFp8Ext, PermuteArgs, Fp8VArgs, I8QuantArgs, AttnI8Ext)):
        if h.vorta_sizeof(which) != C.sizeof(st):
            raise VortaHipError(f"struct layout mismatch for {st.__name__}: "
                                f"C {h.vorta_sizeof(which)} vs ctypes {C.sizeof(st)}")
    _lib = h
    return h


_ERR = {VORTA_EINVAL: "VORTA_EINVAL (bad argument)", VORTA_EUNSUPPORTED: "VORTA_EUNSUPPORTED",
        VORTA_ELAUNCH: "VORTA_ELAUNCH (HIP launch failed)"}


def check(rc: int, what: str):
    if rc != VORTA_OK:
        extra = f", hipError {lib().vorta_last_hip_error()}" if rc == VORTA_ELAUNCH else ""
        # bad geometry / arguments surface as ValueError like the reference's _check_input (hunyuan.py:260-272)
        exc = ValueError if rc == VORTA_EINVAL else VortaHipError
        raise exc(f"{what} failed: {_ERR.get(rc, rc)}{extra}")

"""ctypes binding of libllx_hip.so (the C-ABI declared in include/llx.h).

The library is the only device arithmetic provider of this package: if it is missing or
cannot be loaded the product path raises -- there is no CPU / eager fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime this library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LLX_LIB_PATH") or os.path.join(_HERE, "libllx_hip.so")  # override: A/B of kernel builds

_lib = None

# name -> (restype, argtypes). Kept in step with include/llx.h (tests/test_abi.py checks both ways).
_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float
SIGNATURES: dict[str, tuple] = {
    "llx_version": (c_int, []),
    "llx_last_error_string": (c_char_p, []),
    "llx_device_info": (c_int, [c_int, c_char_p, c_int]),
    "llx_rmsnorm_fwd": (c_int, [_P, _P, _P, _P, _L, _L, _F, _P]),
    "llx_rmsnorm_fwd_quant": (c_int, [_P, _P, _P, _P, _P, _L, _P, _L, _L, _F, _P]),
    "llx_rmsnorm_bwd_workspace_bytes": (c_int64, [_L, _L]),
    "llx_rmsnorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _L, _L, _P]),
    "llx_attn_flags_bytes": (c_int64, [_L, _L]),
    "llx_attn_tile_flags": (c_int, [_P, _P, _P, _L, _L, _P]),
    "llx_attn_fwd": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _P, _P, _P, _L, _L, _L, _L, _L, _F, _P]),
    "llx_attn_dense_fwd": (c_int, [_P, _L, _L, _L, _P, _L, _L, _L, _P, _L, _L, _L, _P, _L, _L, _L, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _F, _P]),
    "llx_gemv_bf16": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _L, _P, _F, _I, _P, _L, _P, _L, _P, _L, _L, _P, _P, _L, _L, _P,
                              _P, _P, _P, _L, _L, _L, _P, _L, _F, _P]),
    "llx_mask_extent": (c_int, [_P, _L, _L, _L, _P, _P]),
    "llx_kv_scatter": (c_int, [_P, _P, _L, _L, _L, _P, _P, _L, _L, _L, _P, _L, _L, _L, _L, _L, _P]),
    "llx_attn_decode_workspace_bytes": (c_int64, [_L, _L, _L, _L]),
    "llx_attn_decode": (c_int, [_P, _L, _L, _L, _P, _P, _L, _L, _L, _P, _L, _L, _L, _P, _L, _L, _L, _P, _P, _L, _L, _L, _L, _L, _L, _L, _F, _P]),
    "llx_attn_bwd_workspace_bytes": (c_int64, [_L, _L, _L, _L]),
    "llx_attn_bwd_ds_bytes": (c_int64, [_L, _L, _L]),
    "llx_attn_bwd": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _P, _P, _L, _L, _P, _L, _L, _P, _L, _L,
                             _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _F, _P]),
    "llx_embedding_fwd": (c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _L, _P]),
    "llx_embedding_bwd": (c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _L, _P]),
    "llx_rope": (c_int, [_P, _L, _L, _P, _L, _L, _P, _L, _L, _L, _L, _I, _P]),
    "llx_swiglu_fwd": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _P]),
    "llx_swiglu_bwd": (c_int, [_P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _L, _L, _P]),
    "llx_scale": (c_int, [_P, _L, _P, _L, _P, _F, _P, _L, _L, _P]),
    "llx_add": (c_int, [_P, _P, _P, _L, _P]),
    "llx_transpose": (c_int, [_P, _L, _P, _L, _L, _L, _I, _P]),
    "llx_i8_to_bf16": (c_int, [_P, _P, _L, _P]),
    "llx_quantize_int8_rowwise": (c_int, [_P, _L, _P, _L, _P, _L, _L, _I, _P]),
    "llx_int8_mm_dequant": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _P, _P]),
    "llx_int8_mm_dequant_f32": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _P, _P]),
    "llx_int8_mm_dequant_ext": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _P, _P, _L, _P, _L, _L, _I, _P, _L, _P, _L, _L, _P]),
    "llx_mel_spectrogram": (c_int, [_P, _L, _L, _P, _P, _P, _P, _L, _L, _L, _P]),
    "llx_logmel_cmn": (c_int, [_P, _P, _L, _L, _L, _P]),
    "llx_gelu_fwd": (c_int, [_P, _L, _P, _L, _L, _L, _P]),
    "llx_gelu_bwd": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _P]),
    "llx_col2im3": (c_int, [_P, _P, _L, _L, _L, _L, _P]),
    "llx_conv_w_reorder": (c_int, [_P, _P, _L, _L, _I, _P]),
    "llx_lora_group_pack": (c_int, [_P, _P, _P, _P, _I, _L, _F, _P, _P, _P, _P, _P]),
    "llx_lora_groups_pack": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "llx_lora_pack": (c_int, [_P, _L, _P, _L, _L, _L, _L, _L, _F, _I, _P]),
    "llx_pad64": (c_int, [_P, _L, _P, _L, _L, _F, _I, _P]),
    "llx_rownorm2": (c_int, [_P, _L, _P, _L, _L, _P]),
    "llx_dora_colscale": (c_int, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _P]),
    "llx_colsum_mul_workspace_bytes": (c_int64, [_L]),
    "llx_colsum_mul": (c_int, [_P, _L, _P, _L, _P, _P, _P, _L, _L, _P]),
    "llx_colscale_bias": (c_int, [_P, _L, _P, _L, _P, _P, _L, _L, _P]),
    "llx_ce_workspace_bytes": (c_int64, [_L]),
    "llx_ce_fwd_bwd": (c_int, [_P, _L, _P, _L, _P, _P, _P, _L, _L, _P]),
    "llx_ce_fwd_bwd_rows": (c_int, [_P, _L, _P, _L, _P, _P, _P, _L, _L, _P, _P]),
    "llx_ce_fwd_bwd_part": (c_int, [_P, _L, _P, _L, _P, _P, _P, _L, _L, _L, _L, _P, _I, _P]),
    "llx_head_compact_index": (c_int, [_P, _P, _P, _P, _P, _L, _P]),
    "llx_gather_rows": (c_int, [_P, _L, _P, _P, _P, _L, _L, _L, _P]),
    "llx_scatter_rows": (c_int, [_P, _L, _P, _P, _P, _L, _L, _L, _P]),
    "llx_skinny_nt": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _P, _P]),
    "llx_skinny_nt_scaled": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _P, _P, _P, _L, _P]),
    "llx_rmsnorm_skinny_nt": (c_int, [_P, _P, _P, _L, _P, _P, _P, _L, _L, _L, _F, _P]),
    "llx_skinny_tn_workspace_bytes": (c_int64, [_L, _L, _L]),
    "llx_skinny_tn": (c_int, [_P, _P, _L, _P, _L, _L, _L, _L, _F, _I, _I, _P, _P, _I, _P]),
    "llx_skinny_tn_partial": (c_int, [_P, _P, _L, _L, _L, _L, _P, _P, _I, _P]),
    "llx_skinny_tn_partial_many": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "llx_skinny_u_workspace_bytes": (_L, [_L, _L]),
    "llx_skinny_tn_partial_many_u": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "llx_skinny_u_reduce": (c_int, [_P, _P, _L, _L, _L, _P, _I, _P]),
    "llx_skinny_tn_partial_many_us": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "llx_skinny_tn_reduce_many": (c_int, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "llx_gemm_nt_bf16": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _L, _P, _L, _L, _I, _P, _L, _P]),
    "llx_gemm_nt_bf16_splitk": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _I, _P, _P]),
    "llx_splitk_combine": (c_int, [_P, _I, _L, _L, _P, _P, _P, _L, _P]),
    "llx_gemm_nt_bf16_rows": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _L, _P, _L, _L, _I, _P, _L, _P, _P]),
    "llx_gemm_tn_bf16": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P]),
    "llx_gemm_tn_bf16_rows": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _P]),
    "llx_gemm_nt_bf16_rope": (c_int, [_P, _L, _P, _L, _P, _L, _L, _L, _L, _P, _L, _P, _L, _L, _P, _L, _L, _P]),
}

# include/llx_debug.h: diagnostic probes, bound for tools/ only
DEBUG_SIGNATURES: dict[str, tuple] = {
    "llx_debug_attn_fwd_occupancy": (c_int, []),
    "llx_debug_attn_bwd_set_stamps": (c_int, [_P]),
    "llx_debug_attn_fwd_stamps": (c_int, [_P, _P, _P, _P, _L, _L, _L, _P, _P]),
}


class LlxError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LlxError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C llama-x_amd/csrc`). There is no fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in {**SIGNATURES, **DEBUG_SIGNATURES}.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().llx_last_error_string()
        raise LlxError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream() -> c_void_p:
    """hipStream_t of torch's current stream on the current device (the fast raw accessor when torch exposes it)."""
    if _raw_stream is not None:
        return c_void_p(_raw_stream(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> c_void_p:
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise LlxError("llx ops run on the HIP device only (tensor on %s); there is no CPU path" % t.device)

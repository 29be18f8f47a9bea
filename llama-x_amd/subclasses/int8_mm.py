"""``torchao::int8_mm_dequant(Tensor A, Tensor B, Tensor A_scale, Tensor B_scale) -> Tensor`` on gfx950.

Same schema, assertions and Meta behaviour as the reference's Triton op (/root/reference/subclasses/int8_mm.py:121-149);
the device implementation is the i8-MFMA GEMM of llama-x_amd/csrc/gemm_bf16.hip (v_mfma_i32_16x16x64_i8, int32
accumulate, fp32 row*col scale epilogue, one rounding to the scale dtype).  As in the reference there is no CPU kernel.

Scale dtypes: bf16 (the training path: llx_int8_mm_dequant), fp32 (llx_int8_mm_dequant_f32: the fp32 product leaves unrounded)
and fp16 (the fp32 kernel on the exactly widened scales, then the ONE rounding to fp16 the reference's store makes) - the
reference returns ``dtype=A_scale.dtype`` for any float scale (int8_mm.py:126,136,143).  Shapes the MFMA tiling does not take
as they are (K not a multiple of 128, N not a multiple of 8, unaligned rows) are zero-padded on the way in: zero int8 products
change no integer sum, padded output columns are sliced off - the result is bit-identical to the unpadded definition.
"""
import torch
from torch import Tensor

from llx import _lib as L

_lib = torch.library.Library("torchao", "FRAGMENT")
_lib.define("int8_mm_dequant(Tensor A, Tensor B, Tensor A_scale, Tensor B_scale) -> Tensor")


def int8_mm_dequant(A: Tensor, B: Tensor, A_scale_rowwise: Tensor, B_scale_colwise: Tensor) -> Tensor:
    assert A.dtype is torch.int8 and B.dtype is torch.int8
    assert A_scale_rowwise.dtype is B_scale_colwise.dtype
    assert A.shape[1] == B.shape[0]
    assert A_scale_rowwise.squeeze().shape == (A.shape[0],)
    assert B_scale_colwise.squeeze().shape == (B.shape[1],)
    assert A_scale_rowwise.is_contiguous()
    assert B_scale_colwise.is_contiguous()
    return torch.ops.torchao.int8_mm_dequant(A, B, A_scale_rowwise, B_scale_colwise)


@torch.library.impl(_lib, "int8_mm_dequant", "Meta")
def _meta(A, B, A_scale_rowwise, B_scale_colwise):
    return torch.empty((A.shape[0], B.shape[1]), device=A.device, dtype=A_scale_rowwise.dtype)


def _launch(A: Tensor, Bt: Tensor, a_scale: Tensor, b_scale: Tensor, out: Tensor | None = None, a2: Tensor | None = None,
            b2: Tensor | None = None, epilogue: int = 0, e: Tensor | None = None, rope: tuple | None = None) -> Tensor:
    from llx import kernels as K

    if K.GEMM_TRACE is None:
        return _launch_impl(A, Bt, a_scale, b_scale, out, a2, b2, epilogue, e, rope)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    res = _launch_impl(A, Bt, a_scale, b_scale, out, a2, b2, epilogue, e, rope)
    ev[1].record()
    M, Kd = A.shape
    N = Bt.shape[0]
    # algorithmic work of the i8 kernel: the int8 product only (the bf16 LoRA extension riding in the same launch is not counted)
    K.GEMM_TRACE.append((ev[0], ev[1], 2.0 * M * N * Kd, 1.0 * (M * Kd + N * Kd) + 2.0 * M * N * (2 if epilogue == 1 else 1), "i8",
                         K.gemm_kernel_launches(M, N, epilogue)))
    return res


def _launch_impl(A: Tensor, Bt: Tensor, a_scale: Tensor, b_scale: Tensor, out: Tensor | None = None, a2: Tensor | None = None,
                 b2: Tensor | None = None, epilogue: int = 0, e: Tensor | None = None, rope: tuple | None = None) -> Tensor:
    """A [M,K] int8 rows, Bt [N,K] int8 rows (= B^T), scales bf16 -> out [M,N] bf16.
    a2 [M,K2], b2 [N,K2] (bf16, K2 % 64 == 0): a LoRA term a2 @ b2^T added on top of the dequantised product in the same launch;
    epilogue 1 (+ e [M,N]), 7 (SwiGLU forward, e = OUTPUT h [M,N/2]) or rope = (table, S, cols) as in llx.kernels.gemm_nt."""
    M, Kd = A.shape
    N = Bt.shape[0]
    if a_scale.dtype is not torch.bfloat16:
        raise L.LlxError(f"int8 GEMM with fused neighbours: scales must be bf16 (got {a_scale.dtype})")
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.bfloat16)
    if a2 is None and epilogue == 0 and rope is None:
        L.check(L.load().llx_int8_mm_dequant(L.ptr(A), A.stride(0), L.ptr(Bt), Bt.stride(0), L.ptr(out), out.stride(0), M, N, Kd,
                                             L.ptr(a_scale), L.ptr(b_scale), L.stream()), "llx_int8_mm_dequant")
        return out
    K2 = 0
    if a2 is not None:
        assert b2 is not None and a2.dtype is torch.bfloat16 and b2.dtype is torch.bfloat16 and a2.shape == (M, b2.shape[1]) and b2.shape[0] == N
        assert a2.stride(1) == 1 and b2.stride(1) == 1
        K2 = a2.shape[1]
    lde, table, rs, rc = 0, None, 0, 0
    if rope is not None:
        assert epilogue == 0
        table, rs, rc = rope
        epilogue = 8
        assert table.dtype is torch.float32 and table.is_contiguous() and table.shape[0] >= rs and table.shape[1:] == (64, 2)
    elif epilogue == 1:
        assert e is not None and e.shape == (M, N) and e.stride(1) == 1 and e.dtype is torch.bfloat16
        lde = e.stride(0)
    elif epilogue == 7:
        assert e is not None and N % 256 == 0 and e.shape == (M, N // 2) and e.stride(1) == 1 and e.dtype is torch.bfloat16
        lde = e.stride(0)
    else:
        assert epilogue == 0, f"int8 GEMM: epilogue {epilogue} unsupported"
    L.check(L.load().llx_int8_mm_dequant_ext(L.ptr(A), A.stride(0), L.ptr(Bt), Bt.stride(0), L.ptr(out), out.stride(0), M, N, Kd,
                                             L.ptr(a_scale), L.ptr(b_scale), L.ptr(a2), a2.stride(0) if a2 is not None else 0, L.ptr(b2),
                                             b2.stride(0) if b2 is not None else 0, K2, epilogue, L.ptr(e), lde, L.ptr(table), rs, rc,
                                             L.stream()), "llx_int8_mm_dequant_ext")
    return out


def _launch_f32(A: Tensor, Bt: Tensor, a_scale: Tensor, b_scale: Tensor) -> Tensor:
    """fp32 scales -> fp32 product (no rounding after the two fp32 multiplies of subclasses/int8_mm.py:112-114)."""
    M, Kd = A.shape
    N = Bt.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=torch.float32)
    L.check(L.load().llx_int8_mm_dequant_f32(L.ptr(A), A.stride(0), L.ptr(Bt), Bt.stride(0), L.ptr(out), out.stride(0), M, N, Kd,
                                             L.ptr(a_scale), L.ptr(b_scale), L.stream()), "llx_int8_mm_dequant_f32")
    return out


def _rows_for_kernel(X: Tensor, k_pad: int, row_pad: int = 0) -> Tensor:
    """int8 [R, K] rows with unit column stride, 16-byte aligned rows and base, K (and optionally R) zero-padded."""
    if k_pad or row_pad:
        return torch.nn.functional.pad(X, (0, k_pad, 0, row_pad))
    if X.stride(1) != 1 or X.stride(0) % 16 != 0 or X.data_ptr() % 16 != 0:
        return X.contiguous()
    return X


@torch.library.impl(_lib, "int8_mm_dequant", "CUDA")  # "CUDA" is the HIP dispatch key on PyTorch-ROCm
def _hip(A, B, A_scale_rowwise, B_scale_colwise):
    dtype = A_scale_rowwise.dtype
    if dtype not in (torch.bfloat16, torch.float32, torch.float16):
        raise L.LlxError(f"int8_mm_dequant: scale dtype {dtype} unsupported (bf16 / fp32 / fp16)")
    M, Kd = A.shape
    N = B.shape[1]
    k_pad, n_pad = -Kd % 128, -N % 8
    A = _rows_for_kernel(A, k_pad)
    # the kernel wants B's columns K-contiguous: B = W.T with strides (1, K) (reference call site subclasses/int8.py:113)
    Bt = _rows_for_kernel(B.t(), k_pad, n_pad)
    sa, sb = A_scale_rowwise.reshape(-1), B_scale_colwise.reshape(-1)
    if n_pad:
        sb = torch.nn.functional.pad(sb, (0, n_pad))
    if dtype is torch.bfloat16:
        out = _launch(A, Bt, sa, sb)
    else:
        out = _launch_f32(A, Bt, sa.float(), sb.float().contiguous())
        if dtype is torch.float16:
            out = out.half()
    return out[:, :N] if n_pad else out

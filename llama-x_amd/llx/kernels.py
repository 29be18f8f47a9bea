"""Functional (non-autograd) wrappers over the C-ABI: shape/dtype checks in Python, raw pointers below.

Every function launches on torch's current HIP stream and returns torch tensors it allocated through
torch's caching allocator; the C side never allocates or synchronises (include/llx.h).
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Sequence

import torch
from torch import Tensor

from . import _lib as L

BF16 = torch.bfloat16
EPI_NONE, EPI_RESIDUAL, EPI_BIAS, EPI_BIAS_GELU, EPI_COLSCALE = 0, 1, 2, 3, 4
EPI_SWIGLU_BWD = 6  # the product is dh; e = gate|up [M, 2N]; out = dg|du [M, 2N] (dh itself is not stored)
EPI_SWIGLU_FWD = 7  # b = [W_gate; W_up]; out = gate|up [M, N]; e = OUTPUT h [M, N/2] = silu(g) * u
SK_PAD = 64
# bench.py sets this to a list to collect (start_event, end_event, algorithmic ops, algorithmic bytes, "bf16" | "i8", kernel launches) per GEMM call
GEMM_TRACE = None
# likewise for the attention kernels: (start_event, end_event, "fwd" | "bwd", B, S, H) per call (the algorithmic FLOPs are the caller's to price:
# they depend on the mask)
ATTN_TRACE = None


def _lib():
    return L.load()


def gemm_kernel_launches(M: int, N: int, epilogue: int) -> int:
    """Kernel launches behind one GEMM call: 2 when the launcher re-tiles the columns of a half-empty last round with 256 x 128 tiles
    (csrc/gemm_bf16.hip launch_gemm), else 1.  Accounting only (the GEMM trace of bench.py counts kernel launches, as rocprofv3 does)."""
    import os

    gm, gn = -(-M // 256), -(-N // 256)
    tail = (gm * gn) % 256
    split = (epilogue != EPI_SWIGLU_FWD and os.environ.get("LLX_GEMM_TAIL", "1") != "0" and os.environ.get("LLX_GEMM_PIPE", "1") != "0"
             and N % 256 == 0 and gm * gn > 256 and 0 < tail <= 128 and tail % gm == 0)
    return 2 if split else 1


def _rows2d(x: Tensor) -> Tensor:
    """View [..., C] as [R, C] with a single row stride (no copy when the leading dims are jointly contiguous)."""
    if x.dim() == 2:
        return x if x.stride(1) == 1 else x.contiguous()
    y = x.reshape(-1, x.shape[-1])
    return y if y.stride(1) == 1 else y.contiguous()


def _chk_bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.dtype is BF16, f"llx kernels compute in bf16 (got {t.dtype})"
    L.require_cuda(*ts)


# ------------------------------------------------------------------------------------------------- rmsnorm
def rmsnorm_fwd(x: Tensor, w: Tensor, eps: float, quant: bool = False):
    """(y, rstd) - with quant=True also (q int8 [rows, dim], qscale bf16 [rows]) = quantize_int8_rowwise(y) from the same pass."""
    _chk_bf16(x, w)
    x2 = _rows2d(x)
    assert x2.stride(0) == x2.shape[1], "rmsnorm input rows must be dense"
    rows, dim = x2.shape
    assert w.shape == (dim,) and w.is_contiguous()
    y = torch.empty_like(x2)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    if quant:
        q = torch.empty(rows, dim, device=x.device, dtype=torch.int8)
        qs = torch.empty(rows, device=x.device, dtype=BF16)
        L.check(_lib().llx_rmsnorm_fwd_quant(L.ptr(x2), L.ptr(w), L.ptr(y), L.ptr(rstd), L.ptr(q), dim, L.ptr(qs), rows, dim, eps, L.stream()),
                "llx_rmsnorm_fwd_quant")
        return y.view(x.shape), rstd, q, qs
    L.check(_lib().llx_rmsnorm_fwd(L.ptr(x2), L.ptr(w), L.ptr(y), L.ptr(rstd), rows, dim, eps, L.stream()), "llx_rmsnorm_fwd")
    return y.view(x.shape), rstd


def rmsnorm_skinny_ok(dim: int) -> bool:
    return dim % 512 == 0 and dim <= 4096


def rmsnorm_skinny_nt(x: Tensor, w: Tensor, eps: float, a_cat: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    """(y, rstd, t): y = rmsnorm(x), t = y @ a_cat^T [rows, 64] (zero beyond column R) from one read of x (csrc/skinny.hip)."""
    _chk_bf16(x, w, a_cat)
    x2 = _rows2d(x)
    rows, dim = x2.shape
    assert x2.stride(0) == dim and w.shape == (dim,) and w.is_contiguous() and rmsnorm_skinny_ok(dim)
    assert a_cat.dim() == 2 and a_cat.shape[1] == dim and a_cat.stride(1) == 1 and a_cat.shape[0] <= SK_PAD
    y = torch.empty_like(x2)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    t = torch.empty(rows, SK_PAD, device=x.device, dtype=BF16)
    L.check(_lib().llx_rmsnorm_skinny_nt(L.ptr(x2), L.ptr(w), L.ptr(a_cat), a_cat.stride(0), L.ptr(y), L.ptr(rstd), L.ptr(t), rows, dim,
                                         a_cat.shape[0], eps, L.stream()), "llx_rmsnorm_skinny_nt")
    return y.view(x.shape), rstd, t


def rmsnorm_bwd(dy: Tensor, x: Tensor, w: Tensor, rstd: Tensor, need_dw: bool, dres: Optional[Tensor] = None,
                dw_out: Optional[Tensor] = None) -> tuple[Tensor, Optional[Tensor]]:
    """dx (+ dres, the gradient coming around the residual connection, joined in the same pass), dw (into dw_out when given)."""
    _chk_bf16(dy, x, w, dres)
    x2, dy2 = _rows2d(x), _rows2d(dy)
    if dy2.stride(0) != dy2.shape[1]:
        dy2 = dy2.contiguous()
    rows, dim = x2.shape
    dx = torch.empty_like(x2)
    dw = ws = None
    if need_dw:
        dw = dw_out if dw_out is not None else torch.empty(dim, device=x.device, dtype=BF16)
        assert dw.shape == (dim,) and dw.is_contiguous() and dw.dtype is BF16
        ws = torch.empty(_lib().llx_rmsnorm_bwd_workspace_bytes(rows, dim), device=x.device, dtype=torch.uint8)
    if dres is not None:
        dres = _rows2d(dres)
        assert dres.shape == x2.shape and dres.stride(0) == dim
    L.check(_lib().llx_rmsnorm_bwd(L.ptr(dy2), L.ptr(x2), L.ptr(w), L.ptr(rstd), L.ptr(dx), L.ptr(dw), 0, L.ptr(ws), L.ptr(dres), rows, dim,
                                   L.stream()), "llx_rmsnorm_bwd")
    return dx.view(x.shape), dw


# ------------------------------------------------------------------------------------------------- gemm
def gemm_nt(a: Tensor, b: Tensor, *, out: Optional[Tensor] = None, a2: Optional[Tensor] = None, b2: Optional[Tensor] = None,
            epilogue: int = EPI_NONE, e: Optional[Tensor] = None, rope: Optional[tuple[Tensor, int, int]] = None,
            k2_eff: Optional[float] = None, m_valid: Optional[Tensor] = None, m_expect: Optional[float] = None) -> Tensor:
    """out[M,N] = a[M,K] @ b[N,K]^T (+ a2[M,K2] @ b2[N,K2]^T) with a fused epilogue; bf16, fp32 accumulate.
    m_valid (device int32 scalar): only the first m_valid rows are wanted - row tiles past them are skipped by the kernel (no host sync);
    m_expect: accounting only, the row count the GEMM trace should book for such a launch.
    rope = (fp32 table [>= S, 64, 2], S, cols): apply_rope on columns [0, cols) of out in the epilogue (row m = position m % S).
    k2_eff: accounting only - the K-extension's true contraction length (LoRA rank; the operands are zero padded to 64 columns and
    block diagonal for fused groups), used by the GEMM trace so that multiplying zeros is not counted as algorithmic work."""
    _chk_bf16(a, b, a2, b2, e, out)
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1], (a.shape, b.shape)
    assert a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    if epilogue == EPI_SWIGLU_BWD:
        assert out is not None and e is not None and out.shape == (M, 2 * N) and e.shape == (M, 2 * N) and out.stride(1) == 1 and e.stride(1) == 1
    else:
        if out is None:
            out = torch.empty(M, N, device=a.device, dtype=BF16)
        assert out.shape == (M, N) and out.stride(1) == 1
    K2 = 0
    if a2 is not None:
        assert b2 is not None and a2.shape[0] == M and b2.shape[0] == N and a2.shape[1] == b2.shape[1]
        assert a2.stride(1) == 1 and b2.stride(1) == 1
        K2 = a2.shape[1]
    lde = 0
    if epilogue == EPI_RESIDUAL:
        assert e is not None and e.shape == (M, N) and e.stride(1) == 1
        lde = e.stride(0)
    elif epilogue == EPI_SWIGLU_BWD:
        lde = e.stride(0)
    elif epilogue == EPI_SWIGLU_FWD:
        assert e is not None and N % 256 == 0 and e.shape == (M, N // 2) and e.stride(1) == 1
        lde = e.stride(0)
    elif epilogue != EPI_NONE:
        assert e is not None and e.shape == (N,) and e.is_contiguous()
    ev = None
    if GEMM_TRACE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if rope is not None:
        table, rs, rc = rope
        assert epilogue == EPI_NONE and table.dtype is torch.float32 and table.is_contiguous() and table.shape[0] >= rs and table.shape[1:] == (64, 2)
        L.check(_lib().llx_gemm_nt_bf16_rope(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(out), out.stride(0), M, N, K,
                                             L.ptr(a2), a2.stride(0) if a2 is not None else 0, L.ptr(b2), b2.stride(0) if b2 is not None else 0, K2,
                                             L.ptr(table), rs, rc, L.stream()), "llx_gemm_nt_bf16_rope")
    elif m_valid is not None:
        assert m_valid.dtype is torch.int32 and m_valid.numel() == 1 and m_valid.device == a.device
        L.check(_lib().llx_gemm_nt_bf16_rows(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(out), out.stride(0), M, N, K,
                                             L.ptr(a2), a2.stride(0) if a2 is not None else 0, L.ptr(b2), b2.stride(0) if b2 is not None else 0, K2,
                                             epilogue, L.ptr(e), lde, L.ptr(m_valid), L.stream()), "llx_gemm_nt_bf16_rows")
    else:
        L.check(_lib().llx_gemm_nt_bf16(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(out), out.stride(0), M, N, K,
                                        L.ptr(a2), a2.stride(0) if a2 is not None else 0, L.ptr(b2), b2.stride(0) if b2 is not None else 0, K2,
                                        epilogue, L.ptr(e), lde, L.stream()), "llx_gemm_nt_bf16")
    if ev is not None:
        ev[1].record()
        kk = K + (K2 if k2_eff is None else min(float(k2_eff), K2))
        Mw = M if m_expect is None else float(m_expect)  # rows actually wanted (row-limited launches: the labelled rows)
        GEMM_TRACE.append((ev[0], ev[1], 2.0 * Mw * N * kk, 2.0 * (Mw * kk + N * kk + Mw * N * (2 if epilogue == EPI_RESIDUAL else 1)), "bf16",
                           gemm_kernel_launches(M, N, 8 if rope is not None else epilogue)))
    return out


def gemm_nt_splitk(a: Tensor, b: Tensor, splits: int, *, m_valid: Optional[Tensor] = None, inv: Optional[Tensor] = None,
                   dev_scalar: Optional[Tensor] = None, m_expect: Optional[float] = None) -> Tensor:
    """bf16(dev_scalar * bf16(a[M,K] @ b[N,K]^T)) with the contraction cut in `splits` ranges computed side by side (one launch, fp32
    partial products, summed by a second small kernel) - for products with few output tiles and a long K.  m_valid: only the first
    m_valid rows of a exist (device int32); inv: out row i = product row inv[i], zero where inv[i] < 0 (the scatter of compacted rows)."""
    _chk_bf16(a, b)
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1] and a.stride(1) == 1 and b.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    part = torch.empty(splits, M, N, device=a.device, dtype=torch.float32)
    ev = None
    if GEMM_TRACE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    L.check(_lib().llx_gemm_nt_bf16_splitk(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(part), M, N, K, splits, L.ptr(m_valid), L.stream()),
            "llx_gemm_nt_bf16_splitk")
    if ev is not None:
        ev[1].record()
        Mw = M if m_expect is None else float(m_expect)
        GEMM_TRACE.append((ev[0], ev[1], 2.0 * Mw * N * K, 2.0 * (Mw * K + N * K) + 4.0 * splits * Mw * N, "bf16", 1))
    out = torch.empty(M, N, device=a.device, dtype=BF16)
    assert inv is None or (inv.dtype is torch.int32 and inv.numel() == M)
    assert dev_scalar is None or (dev_scalar.dtype is torch.float32 and dev_scalar.numel() == 1)
    L.check(_lib().llx_splitk_combine(L.ptr(part), splits, M, N, L.ptr(inv), L.ptr(dev_scalar), L.ptr(out), N, L.stream()), "llx_splitk_combine")
    return out


def transpose(x: Tensor, pad_to: int = 1) -> Tensor:
    """[R,C] -> [C,Rp] bf16 copy (Rp = R rounded up to ``pad_to``, zero filled); int8 sources are widened to bf16."""
    L.require_cuda(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype in (BF16, torch.int8)
    R, C = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    out = (torch.zeros if Rp != R else torch.empty)(C, Rp, device=x.device, dtype=BF16)
    L.check(_lib().llx_transpose(L.ptr(x), x.stride(0), L.ptr(out), Rp, R, C, int(x.dtype is torch.int8), L.stream()), "llx_transpose")
    return out


def pad64(x: Tensor, scale_: float = 1.0, transposed: bool = False) -> Tensor:
    """[R,C<=64] -> [R,64] (or, transposed, [C<=64,R] -> [R,64]) bf16(scale*x), zero padded."""
    _chk_bf16(x)
    assert x.dim() == 2 and x.stride(1) == 1
    R, C = (x.shape[1], x.shape[0]) if transposed else x.shape
    out = torch.empty(R, SK_PAD, device=x.device, dtype=BF16)
    L.check(_lib().llx_pad64(L.ptr(x), x.stride(0), L.ptr(out), R, C, scale_, int(transposed), L.stream()), "llx_pad64")
    return out


def lora_pack(x: Tensor, out: Tensor, row_off: int, col_off: int, scale_: float = 1.0, transposed: bool = False) -> None:
    """Scatter scale*x (or its transpose) into ``out`` at (row_off, col_off); ``out`` is a pre-zeroed operand image."""
    _chk_bf16(x, out)
    assert x.dim() == 2 and x.stride(1) == 1 and out.dim() == 2 and out.stride(1) == 1
    R, C = x.shape
    rr, cc = (C, R) if transposed else (R, C)
    assert row_off + rr <= out.shape[0] and col_off + cc <= out.shape[1], (x.shape, out.shape, row_off, col_off, transposed)
    L.check(_lib().llx_lora_pack(L.ptr(x), x.stride(0), L.ptr(out), out.stride(0), R, C, row_off, col_off, scale_, int(transposed), L.stream()),
            "llx_lora_pack")


def lora_group_pack(lora_as: list, lora_bs: list, K_in: int, scale_: float):
    """(a_cat [R,K], b2 [N,64], bT [R,N], a2t [K,64]) for the members' LoRA factors, built by one launch."""
    import ctypes

    nm = len(lora_as)
    Ns = [b.shape[0] for b in lora_bs]
    ranks = [a.shape[0] for a in lora_as]
    for a, b in zip(lora_as, lora_bs):
        _chk_bf16(a, b)
        assert a.is_contiguous() and b.is_contiguous() and a.shape[1] == K_in and b.shape[1] == a.shape[0]
    N, R = sum(Ns), sum(ranks)
    dev = lora_as[0].device
    buf = torch.empty(R * K_in + N * SK_PAD + R * N + K_in * SK_PAD, device=dev, dtype=BF16)
    a_cat = buf[: R * K_in].view(R, K_in)
    o = R * K_in
    b2 = buf[o : o + N * SK_PAD].view(N, SK_PAD)
    o += N * SK_PAD
    bT = buf[o : o + R * N].view(R, N)
    o += R * N
    a2t = buf[o:].view(K_in, SK_PAD)
    PA = (ctypes.c_void_p * nm)(*[a.data_ptr() for a in lora_as])
    PB = (ctypes.c_void_p * nm)(*[b.data_ptr() for b in lora_bs])
    NS = (ctypes.c_int64 * nm)(*Ns)
    RS = (ctypes.c_int64 * nm)(*ranks)
    L.check(_lib().llx_lora_group_pack(PA, PB, NS, RS, nm, K_in, scale_, L.ptr(a_cat), L.ptr(b2), L.ptr(bT), L.ptr(a2t), L.stream()),
            "llx_lora_group_pack")
    return a_cat, b2, bT, a2t


def lora_groups_pack(groups: list) -> list:
    """[(lora_as, lora_bs, K_in, scale), ...] (<= 4 linear groups, e.g. the four of one transformer layer) -> [(a_cat, b2, bT, a2t), ...]
    as lora_group_pack builds them, from ONE launch and one buffer."""
    import ctypes

    ng = len(groups)
    assert 1 <= ng <= 4
    sizes, metas = [], []
    for las, lbs, K_in, _ in groups:
        for a, b in zip(las, lbs):
            _chk_bf16(a, b)
            assert a.is_contiguous() and b.is_contiguous() and a.shape[1] == K_in and b.shape[1] == a.shape[0]
        N, R = sum(b.shape[0] for b in lbs), sum(a.shape[0] for a in las)
        metas.append((N, R, K_in))
        sizes.append(R * K_in + N * SK_PAD + R * N + K_in * SK_PAD)
    dev = groups[0][0][0].device
    buf = torch.empty(sum(sizes), device=dev, dtype=BF16)
    out, o = [], 0
    for (N, R, K_in) in metas:
        a_cat = buf[o : o + R * K_in].view(R, K_in); o += R * K_in
        b2 = buf[o : o + N * SK_PAD].view(N, SK_PAD); o += N * SK_PAD
        bT = buf[o : o + R * N].view(R, N); o += R * N
        a2t = buf[o : o + K_in * SK_PAD].view(K_in, SK_PAD); o += K_in * SK_PAD
        out.append((a_cat, b2, bT, a2t))
    PA, PB = (ctypes.c_void_p * (4 * ng))(), (ctypes.c_void_p * (4 * ng))()
    NS, RS = (ctypes.c_int64 * (4 * ng))(), (ctypes.c_int64 * (4 * ng))()
    NM, KS, SC = (ctypes.c_int * ng)(), (ctypes.c_int64 * ng)(), (ctypes.c_float * ng)()
    OA, OB, OT, O2 = ((ctypes.c_void_p * ng)() for _ in range(4))
    for j, ((las, lbs, K_in, sc), imgs) in enumerate(zip(groups, out)):
        NM[j], KS[j], SC[j] = len(las), K_in, sc
        for i, (a, b) in enumerate(zip(las, lbs)):
            PA[4 * j + i], PB[4 * j + i], NS[4 * j + i], RS[4 * j + i] = a.data_ptr(), b.data_ptr(), b.shape[0], a.shape[0]
        OA[j], OB[j], OT[j], O2[j] = (t.data_ptr() for t in imgs)
    L.check(_lib().llx_lora_groups_pack(PA, PB, NS, RS, NM, KS, SC, OA, OB, OT, O2, ng, L.stream()), "llx_lora_groups_pack")
    return out


def gemm_tn(a: Tensor, b: Tensor, m_valid: Optional[Tensor] = None, m_expect: Optional[float] = None) -> Tensor:
    """a[M,N1]^T @ b[M,N2] -> [N1,N2] (weight gradients of dense linears / convolutions): the TN MFMA kernel reads both operands as they
    lie (token-major rows, row-strided views allowed).  Shapes it does not take (a dimension below 8 or not a multiple of 8, unaligned
    views) go through transposed, zero-padded copies and the NT kernel.  m_valid (device int32 scalar): only the first min(M, m_valid)
    rows enter the product (the compacted labelled rows of the LM head); m_expect is its value for the FLOP accounting of a traced step."""
    _chk_bf16(a, b)
    assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0] and a.stride(1) == 1 and b.stride(1) == 1
    M, N1 = a.shape
    N2 = b.shape[1]
    ok = (N1 % 8 == 0 and N2 % 8 == 0 and N1 >= 8 and N2 >= 8 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0
          and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)
    if not ok:
        if m_valid is not None:
            raise L.LlxError("gemm_tn: a device-side row count needs the TN kernel's shapes (dimensions multiples of 8, 16-byte aligned rows)")
        return gemm_nt(transpose(a, 64), transpose(b, 64))
    out = torch.empty(N1, N2, device=a.device, dtype=BF16)
    ev = None
    if GEMM_TRACE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if m_valid is not None:
        assert m_valid.dtype is torch.int32 and m_valid.is_cuda and m_valid.numel() == 1
    L.check(_lib().llx_gemm_tn_bf16_rows(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(out), out.stride(0), M, N1, N2,
                                         L.ptr(m_valid) if m_valid is not None else None, L.stream()), "llx_gemm_tn_bf16")
    if ev is not None:
        ev[1].record()
        Me = M if m_expect is None else m_expect
        GEMM_TRACE.append((ev[0], ev[1], 2.0 * Me * N1 * N2, 2.0 * (Me * N1 + Me * N2 + N1 * N2), "bf16_tn", 1))
    return out


def i8_to_bf16(x: Tensor) -> Tensor:
    L.require_cuda(x)
    assert x.dtype is torch.int8 and x.is_contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=BF16)
    L.check(_lib().llx_i8_to_bf16(L.ptr(x), L.ptr(out), x.numel(), L.stream()), "llx_i8_to_bf16")
    return out


def scale(x: Tensor, *, dev_scalar: Optional[Tensor] = None, host_scale: float = 1.0, colscale: Optional[Tensor] = None,
          out: Optional[Tensor] = None) -> Tensor:
    _chk_bf16(x, colscale, out)
    x2 = _rows2d(x)
    rows, cols = x2.shape
    if out is None:
        out = torch.empty(rows, cols, device=x.device, dtype=BF16)
    o2 = _rows2d(out)
    if dev_scalar is not None:
        assert dev_scalar.dtype is torch.float32 and dev_scalar.numel() == 1
    L.check(_lib().llx_scale(L.ptr(x2), x2.stride(0), L.ptr(o2), o2.stride(0), L.ptr(dev_scalar), host_scale, L.ptr(colscale), rows, cols,
                             L.stream()), "llx_scale")
    return out.view(x.shape) if (out.is_contiguous() and out.numel() == x.numel()) else out


def add(x: Tensor, y: Tensor, out: Optional[Tensor] = None) -> Tensor:
    _chk_bf16(x, y, out)
    assert x.shape == y.shape
    x, y = x.contiguous(), y.contiguous()
    z = out if (out is not None and out.is_contiguous()) else torch.empty_like(x)
    L.check(_lib().llx_add(L.ptr(x), L.ptr(y), L.ptr(z), x.numel(), L.stream()), "llx_add")
    if out is not None and z is not out:
        scale(z, out=out)  # strided destination: one more pass through the strided copy kernel
        return out
    return z


_ONES: dict = {}


def colsum(dy: Tensor) -> Tensor:
    """Column sums of [M,N] -> [N] (bias gradients), as a rank-1 skinny_tn against a column of ones."""
    M, N = dy.shape
    key = (M, str(dy.device))
    ones = _ONES.get(key)
    if ones is None:
        ones = torch.zeros(M, SK_PAD, device=dy.device, dtype=BF16)
        ones[:, 0] = 1
        _ONES.clear()
        _ONES[key] = ones
    out = torch.empty(1, N, device=dy.device, dtype=BF16)
    skinny_tn(ones, dy, 1, 1.0, out, transpose_out=False)
    return out.view(N)


# ------------------------------------------------------------------------------------------------- DoRA
def rownorm2(w: Tensor) -> Tensor:
    """fp32 [N]: squared L2 norm of every row of the bf16 matrix w [N, K]."""
    _chk_bf16(w)
    assert w.dim() == 2 and w.stride(1) == 1
    out = torch.empty(w.shape[0], device=w.device, dtype=torch.float32)
    L.check(_lib().llx_rownorm2(L.ptr(w), w.stride(0), L.ptr(out), w.shape[0], w.shape[1], L.stream()), "llx_rownorm2")
    return out


def dora_colscale(wn2: Tensor, G: Tensor, b2: Tensor, AAt: Tensor, m: Tensor, c: Tensor, inv_norm: Tensor, R: int) -> None:
    """c[n] = m[n] / ||W_n + s B_n A|| (bf16), inv_norm[n] = 1 / norm (fp32) for the N rows these views cover (DoRA, lora.py:55-59)."""
    _chk_bf16(G, b2, AAt, m, c)
    N = m.shape[0]
    assert wn2.dtype is torch.float32 and inv_norm.dtype is torch.float32 and wn2.shape == (N,) and inv_norm.shape == (N,) and c.shape == (N,)
    assert G.shape == (N, SK_PAD) and b2.shape == (N, SK_PAD) and AAt.shape[1] == SK_PAD and AAt.shape[0] >= R
    for t in (wn2, G, b2, AAt, m, c, inv_norm):
        assert t.is_contiguous()
    L.check(_lib().llx_dora_colscale(L.ptr(wn2), L.ptr(G), L.ptr(b2), L.ptr(AAt), L.ptr(m), L.ptr(c), L.ptr(inv_norm), N, R, L.stream()),
            "llx_dora_colscale")


def colsum_mul(a: Tensor, b: Tensor, colscale: Optional[Tensor] = None) -> Tensor:
    """bf16 [N]: colscale[n] * sum_m a[m, n] * b[m, n] (fp32 accumulate, deterministic)."""
    _chk_bf16(a, b)
    assert a.shape == b.shape and a.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, N = a.shape
    if colscale is not None:
        assert colscale.dtype is torch.float32 and colscale.shape == (N,) and colscale.is_contiguous()
    out = torch.empty(N, device=a.device, dtype=BF16)
    ws = torch.empty(_lib().llx_colsum_mul_workspace_bytes(N), device=a.device, dtype=torch.uint8)
    L.check(_lib().llx_colsum_mul(L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(colscale), L.ptr(out), L.ptr(ws), M, N, L.stream()),
            "llx_colsum_mul")
    return out


def colscale_bias(x: Tensor, colscale: Tensor, bias: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """bf16(bf16(x * colscale[n]) + bias[n]) on [M, N] rows (the two roundings of DoRALinear's rescale and bias add)."""
    _chk_bf16(x, colscale, bias, out)
    assert x.dim() == 2 and x.stride(1) == 1 and colscale.shape == (x.shape[1],) and colscale.is_contiguous()
    if bias is not None:
        assert bias.shape == (x.shape[1],) and bias.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=BF16)
    assert out.shape == x.shape and out.stride(1) == 1
    L.check(_lib().llx_colscale_bias(L.ptr(x), x.stride(0), L.ptr(out), out.stride(0), L.ptr(colscale), L.ptr(bias), x.shape[0], x.shape[1],
                                     L.stream()), "llx_colscale_bias")
    return out


# ------------------------------------------------------------------------------------------------- embedding
def embedding_fwd(ids: Tensor, table: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[b, s, :] = table[ids[b, s], :]; ``out`` may be a strided [B, S, D] view (e.g. behind an audio prefix)."""
    _chk_bf16(table)
    L.require_cuda(ids)
    assert ids.dtype is torch.int64 and ids.dim() == 2
    ids = ids.contiguous()
    B, S = ids.shape
    V, D = table.shape
    assert table.is_contiguous()
    if out is None:
        out = torch.empty(B, S, D, device=table.device, dtype=BF16)
    assert out.shape == (B, S, D) and out.stride(2) == 1
    L.check(_lib().llx_embedding_fwd(L.ptr(ids), L.ptr(table), L.ptr(out), B * S, D, V, S, out.stride(0), out.stride(1), L.stream()),
            "llx_embedding_fwd")
    return out


def embedding_bwd(ids: Tensor, dy: Tensor, vocab: int) -> Tensor:
    """fp32 [V, D] gradient of the table."""
    _chk_bf16(dy)
    ids = ids.contiguous()
    B, S = ids.shape
    D = dy.shape[-1]
    assert dy.shape == (B, S, D) and dy.stride(2) == 1
    dt = torch.zeros(vocab, D, device=dy.device, dtype=torch.float32)
    L.check(_lib().llx_embedding_bwd(L.ptr(ids), L.ptr(dy), L.ptr(dt), B * S, D, vocab, S, dy.stride(0), dy.stride(1), L.stream()),
            "llx_embedding_bwd")
    return dt


# ------------------------------------------------------------------------------------------------- rope
def rope_(x: Tensor, table: Tensor, nheads: int, backward: bool = False) -> Tensor:
    """In-place interleaved-pair rotation of the first ``nheads`` 128-wide heads of every row of x [B, S, *]."""
    _chk_bf16(x)
    assert x.dim() == 3 and x.stride(2) == 1 and x.shape[2] >= nheads * 128
    assert table.dtype is torch.float32 and table.is_contiguous() and table.shape[0] >= x.shape[1] and table.shape[1:] == (64, 2)
    B, S, _ = x.shape
    L.check(_lib().llx_rope(L.ptr(x), x.stride(0), x.stride(1), L.ptr(x), x.stride(0), x.stride(1), L.ptr(table), B, S, nheads, 128,
                            int(backward), L.stream()), "llx_rope")
    return x


# ------------------------------------------------------------------------------------------------- swiglu
def swiglu_fwd(g: Tensor, u: Tensor) -> Tensor:
    _chk_bf16(g, u)
    assert g.shape == u.shape and g.dim() == 2 and g.stride(1) == 1 and u.stride(1) == 1
    rows, cols = g.shape
    h = torch.empty(rows, cols, device=g.device, dtype=BF16)
    L.check(_lib().llx_swiglu_fwd(L.ptr(g), g.stride(0), L.ptr(u), u.stride(0), L.ptr(h), cols, rows, cols, L.stream()), "llx_swiglu_fwd")
    return h


def swiglu_bwd(dh: Tensor, g: Tensor, u: Tensor, dg: Tensor, du: Tensor) -> None:
    _chk_bf16(dh, g, u, dg, du)
    rows, cols = g.shape
    for t in (dh, g, u, dg, du):
        assert t.shape == (rows, cols) and t.stride(1) == 1
    L.check(_lib().llx_swiglu_bwd(L.ptr(dh), dh.stride(0), L.ptr(g), g.stride(0), L.ptr(u), u.stride(0), L.ptr(dg), dg.stride(0), L.ptr(du),
                                  du.stride(0), rows, cols, L.stream()), "llx_swiglu_bwd")


# ------------------------------------------------------------------------------------------------- attention
class MaskSpec:
    """Per-token mask metadata (replaces FlexAttention's BlockMask in the ``block_mask=`` slot).

    allow(q, k) = (k <= q or k < prefix_len[b]) and (doc_ids is None or doc_ids[b, q] == doc_ids[b, k]).
    ``doc_ids`` int32 [B, S] (or [S], broadcast), ``prefix_len`` int32 [B].  Tile classes are built once on device.
    """

    def __init__(self, doc_ids: Optional[Tensor] = None, prefix_len: Optional[Tensor] = None):
        self.doc_ids = doc_ids
        self.prefix_len = prefix_len
        self._flags = None
        self._key = None

    def prepared(self, B: int, S: int, device) -> "MaskSpec":
        key = (B, S, str(device))
        if self._key == key:
            return self
        d = self.doc_ids
        if d is not None:
            d = d.to(device=device, dtype=torch.int32)
            if d.dim() == 1:
                d = d.view(1, -1)
            assert d.shape[1] == S
            d = d.expand(B, S).contiguous()
        p = self.prefix_len
        if p is not None:
            p = torch.as_tensor(p, device=device).to(torch.int32).reshape(-1)
            if p.numel() == 1:
                p = p.expand(B)
            assert p.numel() == B
            p = p.contiguous()
        self.doc_ids, self.prefix_len = d, p
        self._flags = None
        if d is not None or p is not None:
            fl = torch.empty(_lib().llx_attn_flags_bytes(B, S), device=device, dtype=torch.uint8)
            L.check(_lib().llx_attn_tile_flags(L.ptr(d), L.ptr(p), L.ptr(fl), B, S, L.stream()), "llx_attn_tile_flags")
            self._flags = fl
        self._key = key
        return self


def maskspec_from_dense(mask: Tensor, B: int, S: int) -> Optional[MaskSpec]:
    """The MaskSpec whose rule  allow(q, k) = (k <= q or k < prefix_len[b]) and doc_ids[b, q] == doc_ids[b, k]  reproduces a dense
    bool mask (the reference's small-shape route ``layer(x, rope, mask=...)``, modelling/llama.py:135-137,163-172) EXACTLY, or None.
    Recognised: causal, prefix-LM (per-sample prefix), contiguous documents, and their combination.  Documents are read off the
    sub-diagonal (q-1 and q share a document iff mask[q, q-1]), the prefix off the part above the diagonal (its last visible column);
    the rule is then evaluated densely and compared bit for bit - any other mask (per-head masks, non-contiguous document ids,
    arbitrary patterns) gives None.  One host synchronisation (the comparison) per distinct mask tensor; cached on the tensor."""
    if mask.dtype is not torch.bool or mask.shape[-2:] != (S, S):
        return None
    m = mask
    while m.dim() < 4:
        m = m.unsqueeze(0)
    if m.dim() != 4 or m.shape[1] != 1 or m.shape[0] not in (1, B):
        return None
    m = m[:, 0].expand(B, S, S)
    idx = torch.arange(S, device=m.device)
    upper = m & (idx[None, :, None] < idx[None, None, :])            # allowed pairs with k > q: only the prefix term can make them
    seen = upper.any(dim=1)                                           # [B, S]: column k visible from some earlier row
    prefix = torch.where(seen.any(dim=1), S - seen.flip(1).float().argmax(dim=1), torch.zeros(B, device=m.device, dtype=torch.int64)).to(torch.int64)
    link = torch.diagonal(m, offset=-1, dim1=1, dim2=2)               # [B, S-1]: mask[q, q-1]
    doc = torch.cat([torch.zeros(B, 1, dtype=torch.int64, device=m.device), (~link).to(torch.int64).cumsum(1)], dim=1)
    rule = ((idx[None, None, :] <= idx[None, :, None]) | (idx[None, None, :] < prefix[:, None, None])) & (doc[:, :, None] == doc[:, None, :])
    if not bool(torch.equal(rule, m)):
        return None
    has_doc, has_prefix = bool(doc.any()), bool(prefix.any())
    if not has_doc and not has_prefix:
        return MaskSpec()  # plain causal
    return MaskSpec(doc.to(torch.int32) if has_doc else None, prefix.to(torch.int32) if has_prefix else None)


def attn_fwd(q: Tensor, k: Tensor, v: Tensor, mask: Optional[MaskSpec] = None) -> tuple[Tensor, Tensor]:
    """q [B,S,H,128], k/v [B,S,KVH,128] (last two dims dense; batch/seq strides free) -> o [B,S,H,128], lse [B,H,S]."""
    _chk_bf16(q, k, v)
    B, S, H, hd = q.shape
    KVH = k.shape[2]
    for t in (q, k, v):
        assert t.stride(3) == 1 and t.stride(2) == hd
    o = torch.empty(B, S, H, hd, device=q.device, dtype=BF16)
    lse = torch.empty(B, H, S, device=q.device, dtype=torch.float32)
    d = p = fl = None
    if mask is not None:
        mask = mask.prepared(B, S, q.device)
        d, p, fl = mask.doc_ids, mask.prefix_len, mask._flags
    ev = _trace_begin(ATTN_TRACE)
    L.check(_lib().llx_attn_fwd(L.ptr(q), q.stride(0), q.stride(1), L.ptr(k), k.stride(0), k.stride(1), L.ptr(v), v.stride(0), v.stride(1),
                                L.ptr(o), o.stride(0), o.stride(1), L.ptr(lse), L.ptr(d), L.ptr(p), L.ptr(fl), B, S, H, KVH, hd,
                                1.0 / math.sqrt(hd), L.stream()), "llx_attn_fwd")
    _trace_end(ATTN_TRACE, ev, "fwd", B, S, H)
    return o, lse


def _trace_begin(trace):
    if trace is None:
        return None
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    return ev


def _trace_end(trace, ev, *info):
    if ev is not None:
        ev[1].record()
        trace.append((ev[0], ev[1], *info))


import os as _os

_ATTN_BWD_DS = _os.environ.get("LLX_ATTN_BWD_DS", "0") == "1"
_ATTN_BWD_DS_MAX = int(float(_os.environ.get("LLX_ATTN_BWD_DS_MAX_GB", "16")) * 2**30)


def attn_bwd(q: Tensor, k: Tensor, v: Tensor, o: Tensor, do: Tensor, lse: Tensor, dq: Tensor, dk: Tensor, dv: Tensor,
             mask: Optional[MaskSpec] = None, rope: Optional[Tensor] = None) -> None:
    """rope (fp32 table [>= S, 64, 2]): q, k are the rotated projections; dq, dk come out as gradients of the un-rotated ones."""
    _chk_bf16(q, k, v, o, do, dq, dk, dv)
    if rope is not None:
        assert rope.dtype is torch.float32 and rope.is_contiguous() and rope.shape[0] >= q.shape[1] and rope.shape[1:] == (64, 2)
    B, S, H, hd = q.shape
    KVH = k.shape[2]
    for t in (q, k, v, o, do, dq, dk, dv):
        assert t.stride(3) == 1 and t.stride(2) == hd
    delta = torch.empty(_lib().llx_attn_bwd_workspace_bytes(B, S, H, KVH) // 4, device=q.device, dtype=torch.float32)  # delta + dK/dV partials
    # LLX_ATTN_BWD_DS=1: dS^T scratch (bf16 [B, H, Sp, Sp], 1.07 GB at S = 4096; capped by LLX_ATTN_BWD_DS_MAX_GB, default 16): with it every
    # product of the backward is computed once and dQ becomes a tiled product over the stored dS^T.  Measured at S = 4096: the
    # two routes take the same time (the dS^T round trip is bound by the CUs' fill rate), so the default is the route without scratch,
    # whose dQ kernel recomputes S and dP.
    ds = None
    ds_bytes = _lib().llx_attn_bwd_ds_bytes(B, S, H)
    if _ATTN_BWD_DS and ds_bytes <= _ATTN_BWD_DS_MAX:
        ds = torch.empty(ds_bytes // 2, device=q.device, dtype=BF16)
    d = p = fl = None
    if mask is not None:
        mask = mask.prepared(B, S, q.device)
        d, p, fl = mask.doc_ids, mask.prefix_len, mask._flags
    ev = _trace_begin(ATTN_TRACE)
    L.check(_lib().llx_attn_bwd(L.ptr(q), q.stride(0), q.stride(1), L.ptr(k), k.stride(0), k.stride(1), L.ptr(v), v.stride(0), v.stride(1),
                                L.ptr(o), o.stride(0), o.stride(1), L.ptr(do), do.stride(0), do.stride(1), L.ptr(lse), L.ptr(delta),
                                L.ptr(dq), dq.stride(0), dq.stride(1), L.ptr(dk), dk.stride(0), dk.stride(1), L.ptr(dv), dv.stride(0),
                                dv.stride(1), L.ptr(d), L.ptr(p), L.ptr(fl), L.ptr(rope), L.ptr(ds), B, S, H, KVH, hd, 1.0 / math.sqrt(hd),
                                L.stream()),
            "llx_attn_bwd")
    _trace_end(ATTN_TRACE, ev, "bwd", B, S, H)


def attn_dense_fwd(q: Tensor, k: Tensor, v: Tensor, mask: Tensor) -> Tensor:
    """Inference attention with an explicit bool mask: q [B,H,Sq,128], k/v [B,KVH,Skv,128] (any strides, last dim dense),
    mask broadcastable to [B,H,Sq,Skv] -> o [B,H,Sq,128].  Forward only."""
    _chk_bf16(q, k, v)
    L.require_cuda(mask)
    B, H, Sq, hd = q.shape
    KVH, Skv = k.shape[1], k.shape[2]
    assert mask.dtype is torch.bool and mask.shape[-2:] == (Sq, Skv)
    for t in (q, k, v):
        assert t.stride(3) == 1
    m = mask
    while m.dim() < 4:
        m = m.unsqueeze(0)
    m = m.expand(B if m.shape[0] != 1 else 1, H if m.shape[1] != 1 else 1, Sq, Skv)
    if m.stride(3) != 1:
        m = m.contiguous()
    m_sb = m.stride(0) if m.shape[0] != 1 else 0
    m_sh = m.stride(1) if m.shape[1] != 1 else 0
    o = torch.empty(B, H, Sq, hd, device=q.device, dtype=BF16)
    L.check(_lib().llx_attn_dense_fwd(L.ptr(q), q.stride(0), q.stride(1), q.stride(2), L.ptr(k), k.stride(0), k.stride(1), k.stride(2),
                                      L.ptr(v), v.stride(0), v.stride(1), v.stride(2), L.ptr(o), o.stride(0), o.stride(1), o.stride(2),
                                      L.ptr(m), m_sb, m_sh, m.stride(2), B, H, KVH, Sq, Skv, hd, 1.0 / math.sqrt(hd), L.stream()),
            "llx_attn_dense_fwd")
    return o


# ------------------------------------------------------------------------------------------------- decode path (csrc/decode.hip)
GV_NONE, GV_RESIDUAL, GV_QKV, GV_SWIGLU = 0, 1, 2, 3


def gemv(ws: Sequence[Tensor], x: Tensor, *, norm: Optional[tuple[Tensor, float]] = None, epilogue: int = GV_NONE, out: Optional[Tensor] = None,
         res: Optional[Tensor] = None, qkv: Optional[tuple] = None, lora: Optional[tuple] = None) -> Tensor:
    """out = epilogue([rmsnorm(x) | x] @ cat(ws)^T) for M = x.shape[0] <= 4 rows: every CU streams weight rows (llx_gemv_bf16).
    ws: 1-3 weights [n_s, K]; norm = (weight, eps); res [M, N] for GV_RESIDUAL; qkv = (rope_table, n_q, n_k, k_cache, v_cache, input_pos)
    for GV_QKV (caches [1, KVH, Smax, 128]; returns q [M, n_q]); GV_SWIGLU returns h [M, n_0]; lora = (b factors, t [M, sum r], scale)."""
    _chk_bf16(x, *ws)
    M, Kd = x.shape
    assert 1 <= len(ws) <= 3 and all(w.dim() == 2 and w.shape[1] == Kd and w.stride(1) == 1 for w in ws) and x.stride(1) == 1
    ns = [w.shape[0] for w in ws] + [0] * (3 - len(ws))
    N = sum(ns)
    wp = [L.ptr(w) for w in ws] + [None] * (3 - len(ws))
    lw = [w.stride(0) for w in ws] + [0] * (3 - len(ws))
    n_out = {GV_NONE: N, GV_RESIDUAL: N, GV_QKV: qkv[1] if qkv else 0, GV_SWIGLU: ns[0]}[epilogue]
    if out is None:
        out = torch.empty(M, n_out, device=x.device, dtype=BF16)
    assert out.shape == (M, n_out) and out.stride(1) == 1
    nw, eps = (norm[0], float(norm[1])) if norm is not None else (None, 0.0)
    rope = kc = vc = pos = None
    n_q = n_k = c_sh = c_ss = 0
    if epilogue == GV_QKV:
        rope, n_q, n_k, kc, vc, pos = qkv
        assert rope.dtype is torch.float32 and rope.is_contiguous() and rope.shape[0] >= M and rope.shape[1:] == (64, 2)
        assert kc.shape == vc.shape and kc.dim() == 4 and kc.shape[0] == 1 and kc.shape[3] == 128 and kc.stride() == vc.stride() and kc.stride(3) == 1
        assert pos.dtype is torch.int64 and pos.shape == (M,) and pos.is_contiguous() and pos.is_cuda
        c_sh, c_ss = kc.stride(1), kc.stride(2)
    if epilogue == GV_RESIDUAL:
        assert res is not None and res.shape == (M, N) and res.stride(1) == 1 and res.dtype is BF16
    bs, ranks, t, ldt, lscale = [None] * 3, [0] * 3, None, 0, 0.0
    if lora is not None:
        b_list, t, lscale = lora
        assert len(b_list) == len(ws) and t.dtype is BF16 and t.shape[0] == M and t.stride(1) == 1
        for i, b in enumerate(b_list):
            assert b.dtype is BF16 and b.is_contiguous() and b.shape[0] == ns[i]
            bs[i], ranks[i] = b, b.shape[1]
        assert t.shape[1] == sum(ranks)
        ldt = t.stride(0)
    L.check(_lib().llx_gemv_bf16(wp[0], lw[0], ns[0], wp[1], lw[1], ns[1], wp[2], lw[2], ns[2], L.ptr(x), x.stride(0), M, Kd, L.ptr(nw), eps, epilogue,
                                 L.ptr(out), out.stride(0), L.ptr(res), res.stride(0) if res is not None else 0, L.ptr(rope), n_q, n_k, L.ptr(kc), L.ptr(vc),
                                 c_sh, c_ss, L.ptr(pos), L.ptr(bs[0]), L.ptr(bs[1]), L.ptr(bs[2]), ranks[0], ranks[1], ranks[2], L.ptr(t), ldt, float(lscale),
                                 L.stream()), "llx_gemv_bf16")
    return out


def mask_extent(mask: Tensor) -> Tensor:
    """Device int32 [1]: 1 + the largest key index any row of the bool mask [..., Skv] allows (0 if none)."""
    L.require_cuda(mask)
    assert mask.dtype is torch.bool
    m2 = mask.reshape(-1, mask.shape[-1])
    if m2.stride(1) != 1:
        m2 = m2.contiguous()
    ext = torch.empty(1, device=mask.device, dtype=torch.int32)
    L.check(_lib().llx_mask_extent(L.ptr(m2), m2.stride(0), m2.shape[0], m2.shape[1], L.ptr(ext), L.stream()), "llx_mask_extent")
    return ext


def kv_scatter(k: Tensor, v: Tensor, k_cache: Tensor, v_cache: Tensor, input_pos: Tensor) -> None:
    """k_cache[:, :, input_pos] = k ; v_cache[:, :, input_pos] = v  (KVCache.update, modelling/llama.py:83-90); k / v [B, KVH, L, 128] views."""
    _chk_bf16(k, v, k_cache, v_cache)
    L.require_cuda(input_pos)
    B, KVH, Lq, hd = k.shape
    assert v.shape == k.shape and k.stride() == v.stride() and k.stride(3) == 1 and k_cache.stride() == v_cache.stride() and k_cache.stride(3) == 1
    assert k_cache.shape[0] == B and k_cache.shape[1] == KVH and k_cache.shape[3] == hd and input_pos.shape == (Lq,)
    pos = input_pos.to(torch.int64).contiguous()
    L.check(_lib().llx_kv_scatter(L.ptr(k), L.ptr(v), k.stride(0), k.stride(1), k.stride(2), L.ptr(k_cache), L.ptr(v_cache), k_cache.stride(0),
                                  k_cache.stride(1), k_cache.stride(2), L.ptr(pos), B, KVH, Lq, k_cache.shape[2], hd, L.stream()), "llx_kv_scatter")


_DECODE_WS: dict = {}


_DECODE_WGS = int(_os.environ.get("LLX_DECODE_WGS", "512"))


def attn_decode(q: Tensor, k_cache: Tensor, v_cache: Tensor, mask: Tensor, extent: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """SDPA over the cache for a few query tokens: q [B, H, M, 128] (any strides, last dim dense), caches [B, KVH, Skv, 128], bool mask
    broadcastable to [B, H, M, Skv] -> o [B, H, M, 128] stored as [B, M, H*128] (the layout wo reads).  M * H / KVH <= 16."""
    _chk_bf16(q, k_cache, v_cache)
    L.require_cuda(mask)
    B, H, M, hd = q.shape
    KVH, Skv = k_cache.shape[1], k_cache.shape[2]
    assert mask.dtype is torch.bool and mask.shape[-2:] == (M, Skv) and k_cache.stride() == v_cache.stride()
    m = mask
    while m.dim() < 4:
        m = m.unsqueeze(0)
    if m.stride(3) != 1:
        m = m.contiguous()
    m_sb = m.stride(0) if m.shape[0] != 1 else 0
    m_sh = m.stride(1) if m.shape[1] != 1 else 0
    nsplit = max(1, min(-(-Skv // 32), -(-_DECODE_WGS // (B * KVH)), 128))  # workgroups = nsplit * B * KVH: two or more per CU keep more rows in flight
    nbytes = _lib().llx_attn_decode_workspace_bytes(B, H, M, nsplit)
    key = (q.device, torch.cuda.current_stream(q.device).cuda_stream)
    ws = _DECODE_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _DECODE_WS[key] = torch.empty(nbytes // 4, device=q.device, dtype=torch.float32)
    if out is None:
        out = torch.empty(B, M, H * hd, device=q.device, dtype=BF16)
    o4 = out.view(B, M, H, hd)
    L.check(_lib().llx_attn_decode(L.ptr(q), q.stride(0), q.stride(1), q.stride(2), L.ptr(k_cache), L.ptr(v_cache), k_cache.stride(0), k_cache.stride(1),
                                   k_cache.stride(2), L.ptr(out), o4.stride(0), o4.stride(2), o4.stride(1), L.ptr(m), m_sb, m_sh, m.stride(2), L.ptr(extent),
                                   L.ptr(ws), B, H, KVH, M, Skv, nsplit, hd, 1.0 / math.sqrt(hd), L.stream()), "llx_attn_decode")
    return out


# ------------------------------------------------------------------------------------------------- cross entropy
def head_compact_index(labels: Tensor) -> tuple[Tensor, Tensor, Tensor, Tensor]:
    """(idx, inv, labels_c, count) for the labelled rows (labels != -100) in order; everything stays on the device."""
    L.require_cuda(labels)
    labels = labels.reshape(-1).contiguous()
    assert labels.dtype is torch.int64
    T = labels.numel()
    idx = torch.empty(T, device=labels.device, dtype=torch.int32)
    inv = torch.empty(T, device=labels.device, dtype=torch.int32)
    labels_c = torch.empty(T, device=labels.device, dtype=torch.int64)
    count = torch.empty(1, device=labels.device, dtype=torch.int32)
    L.check(_lib().llx_head_compact_index(L.ptr(labels), L.ptr(idx), L.ptr(inv), L.ptr(labels_c), L.ptr(count), T, L.stream()), "llx_head_compact_index")
    return idx, inv, labels_c, count


def gather_rows(src: Tensor, idx: Tensor, count: Tensor) -> Tensor:
    """dst[j] = src[idx[j]] for j < count (zero rows up to the next multiple of 256, later rows uninitialised)."""
    _chk_bf16(src)
    assert src.dim() == 2 and src.stride(1) == 1 and idx.dtype is torch.int32 and idx.numel() == src.shape[0]
    T, D = src.shape
    dst = torch.empty(T, D, device=src.device, dtype=BF16)
    L.check(_lib().llx_gather_rows(L.ptr(src), src.stride(0), L.ptr(idx), L.ptr(count), L.ptr(dst), D, T, D, L.stream()), "llx_gather_rows")
    return dst


def scatter_rows(src: Tensor, inv: Tensor, dev_scalar: Optional[Tensor] = None) -> Tensor:
    """dst[i] = bf16(dev_scalar * src[inv[i]]) where inv[i] >= 0, zero rows elsewhere."""
    _chk_bf16(src)
    assert src.dim() == 2 and src.stride(1) == 1 and inv.dtype is torch.int32 and inv.numel() == src.shape[0]
    assert dev_scalar is None or (dev_scalar.dtype is torch.float32 and dev_scalar.numel() == 1)
    T, D = src.shape
    dst = torch.empty(T, D, device=src.device, dtype=BF16)
    L.check(_lib().llx_scatter_rows(L.ptr(src), src.stride(0), L.ptr(inv), L.ptr(dev_scalar), L.ptr(dst), D, T, D, L.stream()), "llx_scatter_rows")
    return dst


def ce_fwd_bwd(logits: Tensor, labels: Tensor, write_grad: bool, rows: Optional[Tensor] = None) -> tuple[Tensor, Optional[Tensor]]:
    """Mean CE over labels != -100.  With write_grad the logits buffer is overwritten by d loss / d logits.
    rows (device int32): the rows are compacted (head_compact_index) and only the first `rows` (rounded up to 256) exist."""
    _chk_bf16(logits)
    L.require_cuda(labels)
    lg = _rows2d(logits)
    T, V = lg.shape
    labels = labels.reshape(-1).contiguous()
    assert labels.dtype is torch.int64 and labels.numel() == T
    loss = torch.empty((), device=logits.device, dtype=torch.float32)
    ws = torch.empty(_lib().llx_ce_workspace_bytes(T), device=logits.device, dtype=torch.uint8)
    if rows is not None:
        L.check(_lib().llx_ce_fwd_bwd_rows(L.ptr(lg), lg.stride(0), L.ptr(lg) if write_grad else None, lg.stride(0), L.ptr(labels), L.ptr(loss),
                                           L.ptr(ws), T, V, L.ptr(rows), L.stream()), "llx_ce_fwd_bwd_rows")
    else:
        L.check(_lib().llx_ce_fwd_bwd(L.ptr(lg), lg.stride(0), L.ptr(lg) if write_grad else None, lg.stride(0), L.ptr(labels), L.ptr(loss),
                                      L.ptr(ws), T, V, L.stream()), "llx_ce_fwd_bwd")
    return loss, (lg if write_grad else None)


def ce_chunk(logits: Tensor, labels_all: Tensor, ws: Tensor, loss: Tensor, row0: int, write_grad: bool, rows: Optional[Tensor], first: bool,
             last: bool) -> Optional[Tensor]:
    """One chunk of ce_fwd_bwd over a row set walked in chunks: logits [n, V] are rows row0 .. row0+n of the set, labels_all / ws (float32,
    T + 2) / loss cover the whole set.  Returns the chunk's d loss / d logits (in place) when write_grad."""
    _chk_bf16(logits)
    n, V = logits.shape
    T = labels_all.numel()
    L.check(_lib().llx_ce_fwd_bwd_part(L.ptr(logits), logits.stride(0), L.ptr(logits) if write_grad else None, logits.stride(0), L.ptr(labels_all),
                                       L.ptr(loss), L.ptr(ws), T, V, row0, n, L.ptr(rows), (1 if first else 0) | (2 if last else 0), L.stream()),
            "llx_ce_fwd_bwd_part")
    return logits if write_grad else None


# ------------------------------------------------------------------------------------------------- skinny (LoRA)
def skinny_nt(x: Tensor, w: Tensor, kranges: Optional[Sequence[int]] = None, colscale: Optional[Tensor] = None):
    """[M,K] @ [R,K]^T -> [M,64] bf16, zero beyond column R.
    kranges: (lo, hi) x 4, multiples of 64 - rows 16*nb..16*nb+15 of a block-diagonal w are zero outside k in [lo_nb, hi_nb).
    colscale [K] bf16: returns (out, g) with g [M,K] = bf16(x * colscale) written from the same read of x."""
    _chk_bf16(x, w, colscale)
    assert x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1] and x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    R = w.shape[0]
    out = torch.empty(M, SK_PAD, device=x.device, dtype=BF16)
    kr = None
    if kranges is not None:
        assert len(kranges) == 8
        kr = (ctypes.c_int32 * 8)(*[int(v) for v in kranges])
    if colscale is not None:
        assert colscale.shape == (K,) and colscale.is_contiguous()
        g = torch.empty(M, K, device=x.device, dtype=BF16)
        L.check(_lib().llx_skinny_nt_scaled(L.ptr(x), x.stride(0), L.ptr(w), w.stride(0), L.ptr(out), M, K, R, kr, L.ptr(colscale), L.ptr(g), K,
                                            L.stream()), "llx_skinny_nt_scaled")
        return out, g
    L.check(_lib().llx_skinny_nt(L.ptr(x), x.stride(0), L.ptr(w), w.stride(0), L.ptr(out), M, K, R, kr, L.stream()), "llx_skinny_nt")
    return out


def skinny_tn(u: Tensor, y: Tensor, R: int, scale_: float, out: Tensor, transpose_out: bool, accumulate: bool = False,
              segs: Optional[Sequence[tuple[int, int, int, int]]] = None, pending: Optional[list] = None, defer: bool = False,
              u_from: Optional[Tensor] = None, scaled: Optional[tuple[Tensor, Tensor]] = None) -> Tensor:
    """out ([R,N], or [N,R] when transpose_out) (+)= scale * u[:, :R]^T @ y, u [M,64], y [M,N].
    segs: members (n_lo, n_hi, r_lo, r_hi) of a fused group (block-diagonal product): out is then a flat buffer that receives the
    members' [n, r] blocks one after another, each contiguous.
    pending: a list - only the first stage (fp32 split partials) is launched and the second stage is appended to it; ``out`` is valid
    once skinny_tn_flush(pending) has run (one launch for up to 4 products: the adapter gradients of a transformer block).
    defer (with pending): the first stage waits too, until skinny_tn_partials(pending) / the flush launches it together with the other
    queued products (u and y must stay untouched until then).
    u_from (with pending, not deferred): the batched B^T image [R, N] of the group - the first stage then also emits the column-tile
    partials of y @ u_from^T from the y tiles it stages (y is read once for both products); skinny_u_reduce(pending[-1]) finishes it.
    scaled = (colscale [N] bf16, g [M, N] bf16 output; with u_from): the first stage also writes g = bf16(y * colscale) - the scaled
    gradient an int8 linear's data gradient multiplies (subclasses/int8.py:127)."""
    _chk_bf16(u, y, out)
    M, N = y.shape
    assert u.shape == (M, SK_PAD) and u.is_contiguous() and y.stride(1) == 1 and out.stride(-1) == 1
    sp, ns = None, 0
    if segs is not None:
        ns = len(segs)
        assert 1 <= ns <= 4 and out.is_contiguous() and out.numel() == sum((b - a) * (d - c) for a, b, c, d in segs)
        sp = (ctypes.c_int32 * (4 * ns))(*[int(v) for sgm in segs for v in sgm])
    else:
        assert out.shape == ((N, R) if transpose_out else (R, N))
    ws = torch.empty(_lib().llx_skinny_tn_workspace_bytes(M, N, R), device=y.device, dtype=torch.uint8)
    out_ld = out.stride(0) if out.dim() == 2 else 0
    if pending is None:
        assert u_from is None and scaled is None, "the fused u / scaled-copy outputs need the queued (pending=) form"
        L.check(_lib().llx_skinny_tn(L.ptr(u), L.ptr(y), y.stride(0), L.ptr(out), out_ld, M, N, R, scale_, int(transpose_out), int(accumulate),
                                     L.ptr(ws), sp, ns, L.stream()), "llx_skinny_tn")
        return out
    # queued: [ws, out, out_ld, M, N, R, scale, transpose_out, accumulate, segs, n_segs, first stage not launched yet?, u, y, Bt, upart]
    upart = None
    if u_from is not None:
        assert not defer and u_from.dtype is BF16 and u_from.dim() == 2 and u_from.shape == (R, N) and u_from.stride(1) == 1
        upart = torch.empty(_lib().llx_skinny_u_workspace_bytes(M, N) // 4, device=y.device, dtype=torch.float32)
    if scaled is not None:
        assert u_from is not None and scaled[0].dtype is BF16 and scaled[0].numel() == N and scaled[1].shape == (M, N) and scaled[1].stride(1) == 1
    pending.append([ws, out, out_ld, M, N, R, scale_, int(transpose_out), int(accumulate), sp, ns, True, u, y, u_from, upart, scaled])
    if not defer:
        skinny_tn_partials(pending)
    if len(pending) == 4:
        skinny_tn_flush(pending)
    return out


def skinny_tn_partials(pending: list) -> None:
    """First stage (fp32 split partials) of every queued product that has not had it yet, in ONE launch (up to 4 products): the dB and
    dA products of a linear group are queued with defer=True and launched together."""
    todo = [c for c in pending if c[11]]
    while todo:
        chunk, todo = todo[:4], todo[4:]
        n = len(chunk)
        if n == 1 and chunk[0][15] is None:
            c = chunk[0]
            L.check(_lib().llx_skinny_tn_partial(L.ptr(c[12]), L.ptr(c[13]), c[13].stride(0), c[3], c[4], c[5], L.ptr(c[0]), c[9], c[10], L.stream()),
                    "llx_skinny_tn_partial")
        else:
            UU = (ctypes.c_void_p * n)(*[c[12].data_ptr() for c in chunk])
            YY = (ctypes.c_void_p * n)(*[c[13].data_ptr() for c in chunk])
            LY = (ctypes.c_int64 * n)(*[c[13].stride(0) for c in chunk])
            MM = (ctypes.c_int64 * n)(*[c[3] for c in chunk])
            NN = (ctypes.c_int64 * n)(*[c[4] for c in chunk])
            RR = (ctypes.c_int64 * n)(*[c[5] for c in chunk])
            WS = (ctypes.c_void_p * n)(*[c[0].data_ptr() for c in chunk])
            SG = (ctypes.c_void_p * n)(*[ctypes.cast(c[9], ctypes.c_void_p).value if c[9] is not None else None for c in chunk])
            NS = (ctypes.c_int * n)(*[c[10] for c in chunk])
            BT = (ctypes.c_void_p * n)(*[(c[14].data_ptr() if c[15] is not None else None) for c in chunk])
            LB = (ctypes.c_int64 * n)(*[(c[14].stride(0) if c[15] is not None else 0) for c in chunk])
            UP = (ctypes.c_void_p * n)(*[(c[15].data_ptr() if c[15] is not None else None) for c in chunk])
            CS = (ctypes.c_void_p * n)(*[(c[16][0].data_ptr() if c[16] is not None else None) for c in chunk])
            GG = (ctypes.c_void_p * n)(*[(c[16][1].data_ptr() if c[16] is not None else None) for c in chunk])
            LG = (ctypes.c_int64 * n)(*[(c[16][1].stride(0) if c[16] is not None else 0) for c in chunk])
            L.check(_lib().llx_skinny_tn_partial_many_us(n, UU, YY, LY, MM, NN, RR, WS, SG, NS, BT, LB, UP, CS, GG, LG, L.stream()), "llx_skinny_tn_partial_many")
        for c in chunk:
            c[11] = False
            c[12] = c[13] = None  # the operands are not needed past the first stage


def skinny_u_reduce(entry: list) -> Tensor:
    """u [M, 64] bf16 (= y @ Bt^T, columns >= R zero) from the column-tile partials the first stage of `entry` (a pending item queued with
    u_from) has written; call it right after that skinny_tn (the partial workspace is reused by the next product of the same parity)."""
    assert entry[15] is not None and not entry[11], "the product's first stage must have run with u_from"
    M, N, R = entry[3], entry[4], entry[5]
    out = torch.empty(M, SK_PAD, device=entry[15].device, dtype=BF16)
    L.check(_lib().llx_skinny_u_reduce(L.ptr(entry[15]), L.ptr(out), M, N, R, entry[9], entry[10], L.stream()), "llx_skinny_u_reduce")
    return out


def skinny_tn_flush(pending: list) -> None:
    """Second stage of every product queued by skinny_tn(..., pending=...) in one launch (at most 4 per launch)."""
    skinny_tn_partials(pending)
    while pending:
        chunk, pending[:] = pending[:4], pending[4:]
        n = len(chunk)
        WS = (ctypes.c_void_p * n)(*[c[0].data_ptr() for c in chunk])
        OUT = (ctypes.c_void_p * n)(*[c[1].data_ptr() for c in chunk])
        LD = (ctypes.c_int64 * n)(*[c[2] for c in chunk])
        MM = (ctypes.c_int64 * n)(*[c[3] for c in chunk])
        NN = (ctypes.c_int64 * n)(*[c[4] for c in chunk])
        RR = (ctypes.c_int64 * n)(*[c[5] for c in chunk])
        SC = (ctypes.c_float * n)(*[c[6] for c in chunk])
        TR = (ctypes.c_int * n)(*[c[7] for c in chunk])
        AC = (ctypes.c_int * n)(*[c[8] for c in chunk])
        SG = (ctypes.c_void_p * n)(*[ctypes.cast(c[9], ctypes.c_void_p).value if c[9] is not None else None for c in chunk])
        NS = (ctypes.c_int * n)(*[c[10] for c in chunk])
        L.check(_lib().llx_skinny_tn_reduce_many(n, WS, OUT, LD, MM, NN, RR, SC, TR, AC, SG, NS, L.stream()), "llx_skinny_tn_reduce_many")

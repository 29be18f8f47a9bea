"""Autograd layer over the HIP kernels: per-linear plans (dense / LoRA / int8) and fused block functions.

One ``torch.autograd.Function`` per residual branch keeps the host-side graph small (2 nodes per transformer
layer) and lets the forward write q|k|v and gate|up into fused buffers that the attention / SwiGLU kernels
read through strides.  Frozen base weights get a cached transposed copy (HBM is 288 GB; a second 16 GB image
of Llama-3.1-8B is cheap) so that every data-gradient is the same tuned NT GEMM as the forward.

Reference semantics reproduced here: modelling/llama.py:108-174,196-219 ; modelling/lora.py:40-44 ;
subclasses/int8.py:106-130.
"""
from __future__ import annotations

import os

from typing import Optional, Sequence

import torch
from torch import Tensor, nn
from torch.autograd import Function

from . import kernels as K
from ._lib import LlxError

BF16 = torch.bfloat16


# =================================================================================================
# cached derived images of (mostly frozen) weights
# =================================================================================================
def _cached(t: Tensor, tag: str, build):
    """Cache ``build()`` on the tensor object, keyed by its version counter (in-place updates invalidate)."""
    store = t.__dict__.setdefault("_llx_cache", {})
    ver = t._version
    hit = store.get(tag)
    if hit is not None and hit[0] == ver and hit[1].device == t.device:
        return hit[1]
    val = build()
    store[tag] = (ver, val)
    return val


def weight_t(w: Tensor) -> Tensor:
    """[N,K] -> cached [K,N] bf16 image (B operand of the dgrad GEMM)."""
    return _cached(w, "wt", lambda: K.transpose(w.detach()))


# =================================================================================================
# LinearPlan: how one nn.Linear (possibly LoRA-dressed, possibly int8) runs forward / backward
# =================================================================================================
class LinearPlan:
    """Execution plan of one ``nn.Linear`` child as the reference would run it:

    * plain:      F.linear(x, W, b)                                              (modelling/llama.py:118-120 ...)
    * LoRA:       F.linear(x, W, b) + x @ A^T @ B^T * (alpha/r)                  (modelling/lora.py:40-44)
    * int8 W:     _Int8Linear (weight-only or dynamic-activation)                (subclasses/int8.py:106-130)

    ``tensors()`` lists the tensors autograd must see; ``backward`` returns their gradients in that order.
    """

    def __init__(self, m: nn.Linear):
        from subclasses.int8 import Int8LinearWeight  # local import: subclasses imports llx too

        self.N, self.K = m.out_features, m.in_features
        self.weight = m.weight
        self.bias = m.bias
        self.rank = int(getattr(m, "rank", 0) or 0)
        self.lora_a = getattr(m, "lora_a", None) if self.rank > 0 else None
        self.lora_b = getattr(m, "lora_b", None) if self.rank > 0 else None
        self.scale = float(getattr(m, "scale", 1.0))
        self.dora_m = getattr(m, "m", None) if self.rank > 0 else None  # DoRALinear's magnitude vector (modelling/lora.py:51)
        self.int8 = isinstance(self.weight, Int8LinearWeight)
        self.dynamic = bool(self.int8 and self.weight.dynamic_int8_act)
        if self.dora_m is not None and self.int8:
            # the reference cannot run this combination either: (weight + delta).norm() has no Int8LinearWeight dispatch (subclasses/int8.py:102)
            raise LlxError("DoRA on an Int8LinearWeight is not supported (nor by the reference: no dispatch for weight + delta)")
        if self.rank > 64:
            raise LlxError(f"LoRA rank {self.rank} > 64 is not supported by the skinny kernels")
        if self.rank > 0 and self.lora_a.dtype is not BF16:
            raise LlxError("LoRA factors must be bf16")

    # ---- autograd-visible tensors
    def tensors(self) -> list[Tensor]:
        ts = []
        if not self.int8:
            ts.append(self.weight)
        if self.bias is not None:
            ts.append(self.bias)
        if self.rank > 0:
            ts += [self.lora_a, self.lora_b]
        if self.dora_m is not None:
            ts.append(self.dora_m)
        return ts

    # ---- forward: y = linear(x) [+ residual]; returns (y, saved) where saved is the LoRA intermediate t = x @ A^T
    # (DoRA: the tuple (t, z, c, inv_norm) - un-scaled output, column scale m / ||W + sBA||, 1 / norm)
    def forward(self, x: Tensor, out: Optional[Tensor] = None, residual: Optional[Tensor] = None, gelu: bool = False):
        t = b2 = None
        if self.rank > 0:
            t = K.skinny_nt(x, self.lora_a.detach())
            b2 = K.pad64(self.lora_b.detach(), self.scale)
        if self.dora_m is not None:
            if gelu:
                raise LlxError("DoRA + GELU epilogue is not supported")
            w = self.weight.detach()
            c, inv = dora_colscale([self], w, self.lora_a.detach(), b2)
            z = K.gemm_nt(x, w, a2=t, b2=b2, k2_eff=self.rank)
            direct = residual is None
            y = K.colscale_bias(z, c, self.bias.detach() if self.bias is not None else None, out=out if direct else None)
            if residual is not None:
                y = K.add(y, residual, out=out)
            return y, (t, z, c, inv)
        if self.int8:
            from subclasses.int8 import int8_linear_forward

            if self.bias is not None or gelu:
                raise LlxError("int8 linear with bias is not on the fused path")
            # (x @ W8^T) * scale is rounded to bf16 first (subclasses/int8.py:118); adapter and residual are added after
            direct = self.rank == 0 and residual is None
            y = int8_linear_forward(x, self.weight, out=out if direct else None)
            if self.rank > 0:
                y = K.gemm_nt(t, b2, out=out if residual is None else None, epilogue=K.EPI_RESIDUAL, e=y, k2_eff=0)
            if residual is not None:
                y = K.add(y, residual, out=out)
            return y, t
        w = self.weight.detach()
        if self.bias is not None:
            if residual is not None:
                raise LlxError("bias + residual epilogue is not supported")
            y = K.gemm_nt(x, w, out=out, a2=t, b2=b2, epilogue=K.EPI_BIAS_GELU if gelu else K.EPI_BIAS, e=self.bias.detach(), k2_eff=self.rank)
        elif residual is not None:
            y = K.gemm_nt(x, w, out=out, a2=t, b2=b2, epilogue=K.EPI_RESIDUAL, e=residual, k2_eff=self.rank)
        else:
            y = K.gemm_nt(x, w, out=out, a2=t, b2=b2, k2_eff=self.rank)
        return y, t

    # ---- backward: grads of tensors() and (optionally) dx (accumulated into dx_out when dx_accum)
    def backward(self, dy: Tensor, x: Tensor, t: Optional[Tensor], needs: Sequence[bool], need_dx: bool = True,
                 dx_out: Optional[Tensor] = None, dx_accum: bool = False):
        grads: list[Optional[Tensor]] = []
        ni = iter(needs)
        u = None
        dy_out, gm = dy, None
        if self.dora_m is not None:  # out = z * c (+ bias): d m = colsum(dy * z) / norm, and everything upstream sees dz = dy * c
            t, z, c, inv = t
            if needs[-1]:
                gm = K.colsum_mul(dy, z, inv)
            dy = K.scale(dy, colscale=c)
        if self.rank > 0:
            u = K.skinny_nt(dy, K.transpose(self.lora_b.detach()))  # dy @ B  -> [M,64]
        # --- parameter gradients, in tensors() order
        if not self.int8:
            grads.append(K.gemm_tn(dy, x) if next(ni) else None)
        if self.bias is not None:
            grads.append(K.colsum(dy_out) if next(ni) else None)  # the bias is added after the DoRA rescale
        if self.rank > 0:
            need_a, need_b = next(ni), next(ni)
            ga = gb = None
            if need_a:
                ga = _grad_dst([self.lora_a], (self.rank, self.K))
                if ga is None:
                    ga = torch.empty(self.rank, self.K, device=dy.device, dtype=BF16)
                K.skinny_tn(u, x, self.rank, self.scale, ga, transpose_out=False)
            if need_b:
                gb = _grad_dst([self.lora_b], (self.N, self.rank))
                if gb is None:
                    gb = torch.empty(self.N, self.rank, device=dy.device, dtype=BF16)
                K.skinny_tn(t, dy, self.rank, self.scale, gb, transpose_out=True)
            grads += [ga, gb]
        if self.dora_m is not None:
            grads.append(gm)
        # --- data gradient
        dx = None
        if need_dx:
            b2 = K.pad64(self.lora_a.detach(), self.scale, transposed=True) if self.rank > 0 else None
            if self.int8:
                from subclasses.int8 import int8_weight_t

                g = K.scale(dy, colscale=self.weight.scale)  # (g * scale) rounded to bf16 (subclasses/int8.py:127)
                wt = int8_weight_t(self.weight)
            else:
                g = dy
                wt = weight_t(self.weight)
            if dx_accum:
                dx = K.gemm_nt(g, wt, out=dx_out, a2=u, b2=b2, epilogue=K.EPI_RESIDUAL, e=dx_out, k2_eff=self.rank)
            else:
                dx = K.gemm_nt(g, wt, out=dx_out, a2=u, b2=b2, k2_eff=self.rank)
        return dx, grads


def dora_colscale(members: Sequence[LinearPlan], w_cat: Tensor, a_cat: Tensor, b2: Tensor) -> tuple[Tensor, Tensor]:
    """DoRA's per-output-row factor for one linear or a fused group (modelling/lora.py:55-59, norm detached from A and B):
    c [N] bf16 = m / ||W + s B A||_row and inv_norm [N] fp32, from ||W_n||^2 (cached with the frozen weight), G = W A^T and A A^T -
    W is read once by the skinny MFMA kernel, no [out, in] temporary is formed.  b2 = the [N, 64] image of s * B (block diagonal)."""
    N, R = w_cat.shape[0], a_cat.shape[0]
    wn2 = _cached_multi([m.weight for m in members], "wn2", lambda: K.rownorm2(w_cat))
    G = K.skinny_nt(w_cat, a_cat)
    AAt = K.skinny_nt(a_cat, a_cat)
    c = torch.empty(N, device=w_cat.device, dtype=BF16)
    inv = torch.empty(N, device=w_cat.device, dtype=torch.float32)
    off = 0
    for m in members:
        sl = slice(off, off + m.N)
        K.dora_colscale(wn2[sl], G[sl], b2[sl], AAt, m.dora_m.detach(), c[sl], inv[sl], R)
        off += m.N
    return c, inv


def _cached_multi(tensors: Sequence[Tensor], tag: str, build):
    """Cache on the first tensor, keyed by the version counters (and devices) of all of them."""
    store = tensors[0].__dict__.setdefault("_llx_cache", {})
    key = tuple((t._version, str(t.device)) for t in tensors)
    hit = store.get(tag)
    if hit is not None and hit[0] == key:
        return hit[1]
    val = build()
    store[tag] = (key, val)
    return val


class GroupPlan:
    """Linears that read the SAME input (wq|wk|wv, w1|w3), executed as one GEMM over concatenated weight images.

    Forward: y[M, sum N] = x @ [W_0; W_1; ...]^T with the members' LoRA adapters batched into one skinny product
    (t = x @ [A_0; A_1; ...]^T, 64 columns) and one block-diagonal K-extension operand; backward: one dgrad GEMM over the
    concatenated transposed image (K = sum N) and one skinny product per adapter factor.  Better tile quantisation
    (1792 tiles = 7 full waves for gate|up instead of 2 x 3.5), a third of the launches, x / dy read once.
    Falls back to per-member plans when the members cannot be fused (mixed kinds, biases, sum of ranks > 64)."""

    def __init__(self, mods: Sequence[nn.Linear]):
        self.members = [LinearPlan(m) for m in mods]
        m0 = self.members[0]
        self.K = m0.K
        self.Ns = [m.N for m in self.members]
        self.N = sum(self.Ns)
        self.n_off = [sum(self.Ns[:i]) for i in range(len(self.Ns))]
        self.ranks = [m.rank for m in self.members]
        self.R = sum(self.ranks)
        self.r_off = [sum(self.ranks[:i]) for i in range(len(self.ranks))]
        same_kind = all((m.int8, m.dynamic) == (m0.int8, m0.dynamic) for m in self.members)
        same_scale = len({m.scale for m in self.members if m.rank > 0}) <= 1
        lora_all_or_none = all(r > 0 for r in self.ranks) or self.R == 0
        dora_all_or_none = len({m.dora_m is not None for m in self.members}) == 1
        self.dora = all(m.dora_m is not None for m in self.members)
        self.fused = (same_kind and same_scale and lora_all_or_none and dora_all_or_none and self.R <= 64
                      and all(m.bias is None for m in self.members) and all(m.K == self.K for m in self.members))
        self.int8, self.dynamic = m0.int8, m0.dynamic
        self.packed = None  # (a_cat, b2, bT, a2t) when prepack() built the LoRA operand images of several groups in one launch
        self.scale = next((m.scale for m in self.members if m.rank > 0), 1.0)
        # true contraction length of the (block-diagonal, zero-padded) LoRA K-extension: forward sum_i N_i r_i / N, dgrad R
        self.k2_fwd = sum(n * r for n, r in zip(self.Ns, self.ranks)) / self.N

    def tensors(self) -> list[Tensor]:
        return [t for m in self.members for t in m.tensors()]

    # ---- cached concatenated images of the (normally frozen) base weights
    def _weights(self):
        return [(m.weight.int_data if m.int8 else m.weight) for m in self.members]

    def w_cat(self) -> Tensor:
        ws = self._weights()
        if self.int8 and self.dynamic:
            return _cached_multi(ws, "cat_i8", lambda: torch.cat([w.detach() for w in ws], 0))
        if self.int8:
            return _cached_multi(ws, "cat_bf16", lambda: K.i8_to_bf16(torch.cat([w.detach() for w in ws], 0)))
        return _cached_multi(ws, "cat", lambda: torch.cat([w.detach() for w in ws], 0) if len(ws) > 1 else ws[0].detach())

    def wt_cat(self) -> Tensor:
        ws = self._weights()
        return _cached_multi(ws, "cat_t", lambda: K.transpose(torch.cat([w.detach() for w in ws], 0) if len(ws) > 1 else ws[0].detach()))

    def scale_cat(self) -> Tensor:
        ss = [m.weight.scale for m in self.members]
        return _cached_multi(ss, "cat_scale", lambda: torch.cat(ss, 0) if len(ss) > 1 else ss[0])

    # ---- forward
    def forward(self, x: Tensor, out: Optional[Tensor] = None, residual: Optional[Tensor] = None, swiglu_h: Optional[Tensor] = None,
                rope: Optional[tuple[Tensor, int, int]] = None, xq: Optional[tuple[Tensor, Tensor]] = None, t_pre: Optional[Tensor] = None):
        """out: [M, sum N] (row-strided view allowed; allocated when None).  ``residual`` [M, sum N] is added in the GEMM
        epilogue (x + linear(..), modelling/llama.py:172-173).  xq = quantize_int8_rowwise(x) / t_pre = x @ A_cat^T when the producer of
        x already made them (the RMSNorm forward).  Returns (out, saved) - saved feeds backward()."""
        if out is None:
            out = torch.empty(x.shape[0], self.N, device=x.device, dtype=BF16)
        return out, self._forward(x, out, residual, swiglu_h, rope, xq, t_pre)

    def norm_can_make_t(self, dim: int) -> bool:
        """The preceding RMSNorm can emit t = xn @ A_cat^T itself: the operand images exist already (prepack) and the width fits."""
        return self.fused and self.R > 0 and self.packed is not None and K.rmsnorm_skinny_ok(dim) and _FUSE_NORM_SKINNY

    def wants_quantized_input(self) -> bool:
        """The fused group runs torchao::int8_mm_dequant on row-wise quantised activations (subclasses/int8.py:110-113)."""
        return self.fused and self.int8 and self.dynamic

    def rope_fusable(self) -> bool:
        """apply_rope can ride in the projection GEMM's epilogue (one fused bf16 GEMM writes the whole q|k|v row)."""
        return self.fused and (not self.int8 or self.dynamic) and not self.dora

    def swiglu_fusable(self) -> bool:
        """silu(g) * u can ride in the gate|up GEMM's epilogue."""
        return self.fused and (not self.int8 or self.dynamic) and not self.dora and len(self.members) == 2 and self.Ns[0] % 128 == 0

    def _forward(self, x: Tensor, out: Tensor, residual: Optional[Tensor], swiglu_h: Optional[Tensor] = None,
                 rope: Optional[tuple[Tensor, int, int]] = None, xq: Optional[tuple[Tensor, Tensor]] = None, t_pre: Optional[Tensor] = None):
        if not self.fused:
            assert residual is None or len(self.members) == 1
            return [m.forward(x, out=out[:, o : o + n], residual=residual)[1] for m, o, n in zip(self.members, self.n_off, self.Ns)]
        t = b2 = None
        if self.R > 0:
            # the four operand images (forward: a_cat, b2; backward: bT, a2t) come out of one launch and ride along in `saved`
            a_cat, b2, bT, a2t = self.packed or K.lora_group_pack([m.lora_a.detach() for m in self.members],
                                                                  [m.lora_b.detach() for m in self.members], self.K, self.scale)
            t = (t_pre if t_pre is not None else K.skinny_nt(x, a_cat), bT, a2t)
        if self.dora:
            # DoRA members: the un-scaled product z is kept for d m; rescale (and residual) are their own HBM-bound passes, so RoPE /
            # SwiGLU run stand-alone after this group (rope_fusable / swiglu_fusable are False)
            assert swiglu_h is None and rope is None
            c, inv = dora_colscale(self.members, self.w_cat(), a_cat, b2)
            z = K.gemm_nt(x, self.w_cat(), a2=t[0], b2=b2, k2_eff=self.k2_fwd)
            K.scale(z, colscale=c, out=out)
            if residual is not None:
                K.add(out, residual, out=out)
            return (*t, z, c, inv)
        if not self.int8:
            if residual is not None:
                K.gemm_nt(x, self.w_cat(), out=out, a2=t[0] if t else None, b2=b2, epilogue=K.EPI_RESIDUAL, e=residual, k2_eff=self.k2_fwd)
            elif swiglu_h is not None:  # gate|up group: h = silu(g) * u leaves the same GEMM (g and u are stored as usual)
                K.gemm_nt(x, self.w_cat(), out=out, a2=t[0] if t else None, b2=b2, epilogue=K.EPI_SWIGLU_FWD, e=swiglu_h, k2_eff=self.k2_fwd)
            else:
                K.gemm_nt(x, self.w_cat(), out=out, a2=t[0] if t else None, b2=b2, rope=rope, k2_eff=self.k2_fwd)
            return t
        if self.dynamic:
            from subclasses.int8 import quantize_int8_rowwise
            from subclasses.int8_mm import _launch as i8_gemm

            xi, xs = xq if xq is not None else quantize_int8_rowwise(x)
            # one launch: int8 product dequantised in place, the adapter as bf16 K-extension, then residual / SwiGLU / RoPE epilogue
            a2, bb = (t[0], b2) if self.R > 0 else (None, None)
            if residual is not None:
                i8_gemm(xi, self.w_cat(), xs, self.scale_cat(), out=out, a2=a2, b2=bb, epilogue=K.EPI_RESIDUAL, e=residual)
            elif swiglu_h is not None:
                i8_gemm(xi, self.w_cat(), xs, self.scale_cat(), out=out, a2=a2, b2=bb, epilogue=K.EPI_SWIGLU_FWD, e=swiglu_h)
            else:
                i8_gemm(xi, self.w_cat(), xs, self.scale_cat(), out=out, a2=a2, b2=bb, rope=rope)
            return t
        else:
            direct = self.R == 0 and residual is None
            y0 = K.gemm_nt(x, self.w_cat(), out=out if direct else None, epilogue=K.EPI_COLSCALE, e=self.scale_cat())
        # (x @ W8^T) * scale is rounded to bf16 first (subclasses/int8.py:118); adapter and residual are added after
        if self.R > 0:
            y0 = K.gemm_nt(t[0], b2, out=out if residual is None else None, epilogue=K.EPI_RESIDUAL, e=y0, k2_eff=0)
        if residual is not None:
            K.add(y0, residual, out=out)
        return t

    # ---- backward: returns (dx, grads aligned with tensors())
    def _tn_segs(self) -> Optional[list[tuple[int, int, int, int]]]:
        """(n_lo, n_hi, r_lo, r_hi) of every adapted member for the segment form of skinny_tn; None when it does not apply."""
        if getattr(self, "_segs", None) is None:
            segs = [(no, no + n, ro, ro + r) for ro, r, no, n in zip(self.r_off, self.ranks, self.n_off, self.Ns) if r > 0]
            ok = 1 <= len(segs) <= 4 and all(a % 256 == 0 and (b % 256 == 0 or b == self.N) for a, b, _, _ in segs)
            self._segs = segs if ok else False
        return self._segs or None

    def _kranges(self) -> Optional[list[int]]:
        """k range of each 16-row block of the batched B^T (block-diagonal: member i owns rows r_off[i].. and k in
        n_off[i]..n_off[i]+N_i); None when a boundary is not a multiple of 64 (the kernel then treats B^T as dense)."""
        if getattr(self, "_kr", None) is None and len(self.members) == 1:
            self._kr = False  # a single member's B^T is dense
        if getattr(self, "_kr", None) is None:
            kr: list[int] = []
            for nb in range(4):
                lo, hi = None, None
                for ro, r, no, n in zip(self.r_off, self.ranks, self.n_off, self.Ns):
                    if r > 0 and ro < 16 * nb + 16 and ro + r > 16 * nb:
                        lo = no if lo is None else min(lo, no)
                        hi = no + n if hi is None else max(hi, no + n)
                kr += [lo or 0, hi or 0] if lo is not None else [0, 0]
            ok = all(v % 64 == 0 for v in kr)
            self._kr = kr if ok else False
        return self._kr or None

    def backward(self, dy: Tensor, x: Tensor, saved, needs: Sequence[bool], need_dx: bool, dx_out: Optional[Tensor] = None,
                 swiglu: Optional[tuple[Tensor, Tensor]] = None, pending: Optional[list] = None):
        """swiglu = (gate|up activations [M, 2K], dg|du output [M, 2K]): the data gradient of this linear is the gradient of
        silu(g)*u, and the dgrad GEMM applies the SwiGLU backward in its epilogue (returns the dg|du tensor instead of dx).
        pending: the second stage of the adapter-gradient products is queued there instead of launched (K.skinny_tn_flush: the block's
        backward runs the second stages of its two groups - four products - in one launch before it returns the gradients)."""
        if not self.fused:
            grads, dx, first = [], None, True
            ni = 0
            for m, o, n, t in zip(self.members, self.n_off, self.Ns, saved):
                cnt = len(m.tensors())
                d, g = m.backward(dy[:, o : o + n], x, t, needs[ni : ni + cnt], need_dx, dx_out, not first)
                ni += cnt
                grads += g
                dx = d if d is not None else dx
                first = False
            return dx, grads
        gM = None
        ni = iter(needs)
        need = []
        for m in self.members:
            cnt = len(m.tensors())
            need.append([next(ni) for _ in range(cnt)])
        ia = -3 if self.dora else -2  # per-member needs end with (..., lora_a, lora_b[, m])
        dy_out = dy
        z = inv = None
        if self.dora:  # out = z * c: d m = colsum(dy * z) / norm; everything upstream sees dz = dy * c
            t, bT, a2t, z, c, inv = saved
            dy = K.scale(dy_out, colscale=c)
        else:
            t, bT, a2t = saved if saved is not None else (None, None, None)
        u = gA = gBt = g_scaled = None
        fused_u_here = False
        gB_views: Optional[list[Optional[Tensor]]] = None
        need_a, need_b = self.R > 0 and any(nd[ia] for nd in need), self.R > 0 and any(nd[ia + 1] for nd in need)
        # (issuing the dB chain, the dA chain or both on a side stream - parallel branches of the replayed hipGraph - was measured three
        # ways against the data-gradient GEMM and against each other: within +-0.2 ms per step of this plain order, so it stays plain)
        # dB = s t^T.dy (and DoRA's d m): nothing here needs u
        if self.dora and any(nd[-1] for nd in need):
            gM = K.colsum_mul(dy_out, z, inv)
        if need_b:
            segs = self._tn_segs()
            if segs is not None:  # each member's dB lands in its own contiguous block of one flat buffer: no slicing copies
                flat = _grad_dst([m.lora_b for m in self.members if m.rank > 0])  # the arena keeps a group's B factors back to back
                if flat is None:
                    flat = torch.empty(sum((b - a) * (d - c_) for a, b, c_, d in segs), device=dy.device, dtype=BF16)
                fuse_u = _FUSE_U and pending is not None and tuple(bT.shape) == (self.R, self.N)
                if fuse_u:
                    # dB's first stage also emits u = dy @ B from the dy tiles it stages (dy read once instead of twice) - and, for an
                    # int8 base, (dy * scale), the operand of its data gradient (subclasses/int8.py:127)
                    sc = None
                    if self.int8 and need_dx and _FUSE_DY_SCALE:
                        g_scaled = torch.empty_like(dy)
                        sc = (self.scale_cat(), g_scaled)
                    K.skinny_tn(t, dy, self.R, self.scale, flat, transpose_out=True, segs=segs, pending=pending, u_from=bT, scaled=sc)
                    u = K.skinny_u_reduce(pending[-1])
                    fused_u_here = True
                else:
                    K.skinny_tn(t, dy, self.R, self.scale, flat, transpose_out=True, segs=segs, pending=pending, defer=need_a and _BATCH_TN_PARTIAL)
                gB_views, off = [], 0
                for m_, (a, b, c_, d) in zip([m for m in self.members if m.rank > 0], segs):
                    gB_views.append(flat[off : off + (b - a) * (d - c_)].view(b - a, d - c_))
                    off += (b - a) * (d - c_)
            else:
                gBt = torch.empty(self.N, self.R, device=dy.device, dtype=BF16)
                K.skinny_tn(t, dy, self.R, self.scale, gBt, transpose_out=True)  # (sliced below: needs the finished product)
        if self.R > 0:  # u = dy.B, then dA = s u^T.x
            if u is not None:
                pass  # came out of the dB pass above
            elif self.int8 and need_dx and _FUSE_DY_SCALE:
                # ... and (dy * scale), the int8 base's data-gradient operand (subclasses/int8.py:127), from the same read of dy
                u, g_scaled = K.skinny_nt(dy, bT, self._kranges(), colscale=self.scale_cat())
            else:
                u = K.skinny_nt(dy, bT, self._kranges())  # [M,64]: column block i = dy_i @ B_i
            if need_a:
                gA = _grad_dst([m.lora_a for m in self.members if m.rank > 0], (self.R, self.K))  # ... and its A factors
                if gA is None:
                    gA = torch.empty(self.R, self.K, device=dy.device, dtype=BF16)
                # (with the fused u the dB stage has run already and dA's first stage launches on its own: batched with the NEXT group's
                # dB + u stage it made one long launch of unequal blocks - 105 us where the two take 25 + 39)
                K.skinny_tn(u, x, self.R, self.scale, gA, transpose_out=False, pending=pending)  # otherwise: launches the deferred dB stage with its own
        dx = None
        if need_dx:
            if self.int8:
                g = g_scaled if g_scaled is not None else K.scale(dy, colscale=self.scale_cat())  # (g * scale) rounded (subclasses/int8.py:127)
            else:
                g = dy
            if swiglu is not None:
                dx = K.gemm_nt(g, self.wt_cat(), out=swiglu[1], a2=u, b2=a2t if self.R > 0 else None, epilogue=K.EPI_SWIGLU_BWD, e=swiglu[0], k2_eff=self.R)
            else:
                dx = K.gemm_nt(g, self.wt_cat(), out=dx_out, a2=u, b2=a2t if self.R > 0 else None, k2_eff=self.R)
        grads: list[Optional[Tensor]] = []
        gb_i = 0
        for m, nd, ro, no, n in zip(self.members, need, self.r_off, self.n_off, self.Ns):
            j = 0
            if not m.int8:
                grads.append(K.gemm_tn(dy[:, no : no + n], x) if nd[j] else None)
                j += 1
            if m.rank > 0:
                grads.append(gA[ro : ro + m.rank] if nd[j] else None)
                if gB_views is not None:
                    grads.append(gB_views[gb_i] if nd[j + 1] else None)
                    gb_i += 1
                else:
                    grads.append(gBt[no : no + n, ro : ro + m.rank].contiguous() if nd[j + 1] else None)
                j += 2
            if m.dora_m is not None:
                grads.append(gM[no : no + n] if nd[j] else None)
        return dx, grads


def _grad_dst(params: Sequence[Optional[Tensor]], shape: Optional[tuple] = None) -> Optional[Tensor]:
    """Where backward may write the gradients of `params` directly: the slice of the trainable arena's gradient buffer
    (llx.arena.TrainableArena) that covers them back to back, or None - not in an arena, not adjacent in this order, one of them
    already has a ``.grad`` (accumulation micro-step: autograd has to ADD, so the product goes to a fresh buffer), or another autograd
    node of this backward has already been given the slot (a parameter with two consumers: the second product must not overwrite the
    first; the claim is released by the arena's settle() / zero_grad() / optimizer-step hook)."""
    g0 = first = end = None
    for p in params:
        slot = getattr(p, "_llx_slot", None)
        if slot is None or p.grad is not None or getattr(p, "_llx_claimed", False):
            return None
        G, off, n = slot
        if g0 is None:
            g0, first = G, off
        elif G is not g0 or off != end:
            return None
        end = off + n
    if g0 is None:
        return None
    for p in params:
        p._llx_claimed = True
    out = g0[first:end]
    return out.view(shape) if shape is not None else out


def _enc(o, flat: list):
    # module-level on purpose: a recursive CLOSURE over `flat` would be a reference cycle (function <-> its own cell) that keeps every
    # saved activation alive until the cyclic GC happens to run - exactly the memory activation checkpointing is there to release
    if isinstance(o, Tensor):
        flat.append(o)
        return ("t", len(flat) - 1)
    if isinstance(o, (tuple, list)):
        return ("l", [_enc(x, flat) for x in o])
    return ("c", o)


def _dec(sp, flat):
    kind, v = sp
    if kind == "t":
        return flat[v]
    if kind == "l":
        return tuple(_dec(x, flat) for x in v)
    return v


def _save(ctx, *objs) -> None:
    """Route EVERY tensor a backward needs through ``save_for_backward`` (nested tuples / lists / None allowed): saved-tensor hooks
    (non-reentrant ``checkpoint`` at modelling/llama.py, CPU offload) then see the activations and can drop / recompute them,
    and autograd's in-place version checks cover them.  Only the nesting structure stays on ``ctx``."""
    flat: list[Tensor] = []
    ctx._llx_spec = [_enc(o, flat) for o in objs]
    ctx.save_for_backward(*flat)


def _load(ctx) -> list:
    flat = ctx.saved_tensors
    return [_dec(sp, flat) for sp in ctx._llx_spec]


def prepack(plans: Sequence[GroupPlan]) -> None:
    """Build the LoRA operand images of several fused groups (the four of a transformer layer) in one launch instead of one per group."""
    todo = [p for p in plans if p.fused and p.R > 0]
    for i in range(0, len(todo), 4):
        chunk = todo[i : i + 4]
        if len(chunk) < 2:
            break
        imgs = K.lora_groups_pack([([m.lora_a.detach() for m in p.members], [m.lora_b.detach() for m in p.members], p.K, p.scale) for p in chunk])
        for p, im in zip(chunk, imgs):
            p.packed = im


def _plans_tensors(plans: Sequence[LinearPlan]) -> tuple[list[Tensor], list[int]]:
    ts, counts = [], []
    for p in plans:
        t = p.tensors()
        ts += t
        counts.append(len(t))
    return ts, counts


# =================================================================================================
# generic single-linear function (LM head without labels, stand-alone LoRALinear / int8 F.linear)
# =================================================================================================
class LinearFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, plan: LinearPlan, *tensors):
        K.L.require_cuda(x)
        x2 = K._rows2d(x)
        y, t = plan.forward(x2)
        ctx.plan = plan
        _save(ctx, x2, t)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], plan.N)

    @staticmethod
    def backward(ctx, dy: Tensor):
        plan: LinearPlan = ctx.plan
        dy2 = K._rows2d(dy)
        needs = ctx.needs_input_grad[2:]
        x2, t = _load(ctx)
        dx, grads = plan.backward(dy2, x2, t, needs, need_dx=ctx.needs_input_grad[0])
        return (dx.view(ctx.xshape) if dx is not None else None, None, *grads)


def linear(x: Tensor, m: nn.Linear) -> Tensor:
    plan = LinearPlan(m)
    return LinearFn.apply(x, plan, *plan.tensors())


# =================================================================================================
# RMSNorm
# =================================================================================================
class RMSNormFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, eps: float):
        y, rstd = K.rmsnorm_fwd(x.contiguous(), w.detach(), eps)
        ctx.save_for_backward(x, w, rstd)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, w, rstd = ctx.saved_tensors
        dx, dw = K.rmsnorm_bwd(dy.contiguous(), x.contiguous(), w.detach(), rstd, ctx.needs_input_grad[1], dw_out=_grad_dst([w]))
        return dx, dw, None


def rmsnorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    return RMSNormFn.apply(x, w, eps)


# =================================================================================================
# attention residual branch:  [x +] wo( attn( rope(wq xn), rope(wk xn), wv xn ) ),  xn = [rmsnorm(x)]
# =================================================================================================
class AttnBlockMeta:
    def __init__(self, qkv: GroupPlan, wo: GroupPlan, num_heads, num_kv_heads, head_dim, mask, eps, fuse_norm, fuse_residual):
        self.qkv, self.wo = qkv, wo
        self.H, self.KVH, self.hd = num_heads, num_kv_heads, head_dim
        self.mask, self.eps = mask, eps
        self.fuse_norm, self.fuse_residual = fuse_norm, fuse_residual


class AttnBlockFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, rope: Tensor, norm_w: Optional[Tensor], meta: AttnBlockMeta, *tensors):
        K.L.require_cuda(x)
        B, S, D = x.shape
        H, KVH, hd = meta.H, meta.KVH, meta.hd
        x2 = K._rows2d(x.contiguous())
        xq = t_pre = None
        if meta.fuse_norm and meta.qkv.wants_quantized_input() and _FUSE_NORM_QUANT:
            xn, rstd, *xq = K.rmsnorm_fwd(x2, norm_w.detach(), meta.eps, quant=True)  # the norm also emits quantize_int8_rowwise(xn)
        elif meta.fuse_norm and meta.qkv.norm_can_make_t(D):
            xn, rstd, t_pre = K.rmsnorm_skinny_nt(x2, norm_w.detach(), meta.eps, meta.qkv.packed[0])  # ... or the adapters' xn @ A_cat^T
        elif meta.fuse_norm:
            xn, rstd = K.rmsnorm_fwd(x2, norm_w.detach(), meta.eps)
        else:
            xn, rstd = x2, None
        W = (H + 2 * KVH) * hd
        qkv = torch.empty(B * S, W, device=x.device, dtype=BF16)
        fuse_rope = meta.qkv.rope_fusable() and _FUSE_ROPE
        _, tqkv = meta.qkv.forward(xn, qkv, rope=(rope, S, (H + KVH) * hd) if fuse_rope else None, xq=xq, t_pre=t_pre)
        qkv3 = qkv.view(B, S, W)
        if not fuse_rope:
            K.rope_(qkv3, rope, H + KVH)
        q = qkv3[..., : H * hd].unflatten(-1, (H, hd))
        k = qkv3[..., H * hd : (H + KVH) * hd].unflatten(-1, (KVH, hd))
        v = qkv3[..., (H + KVH) * hd :].unflatten(-1, (KVH, hd))
        o, lse = K.attn_fwd(q, k, v, meta.mask)
        o2 = o.view(B * S, H * hd)
        y, to = meta.wo.forward(o2, None, x2 if meta.fuse_residual else None)
        ctx.meta = meta
        # x2 is x itself (or its contiguous copy) and xn == x2 without the fused norm: saved once, by identity, below
        _save(ctx, x, rope, norm_w, x2, xn if meta.fuse_norm else None, rstd, qkv3, o, lse, tqkv, to)
        return y.view(B, S, D)

    @staticmethod
    def backward(ctx, dy: Tensor):
        meta: AttnBlockMeta = ctx.meta
        x, rope, norm_w, x2, xn, rstd, qkv3, o, lse, tqkv, to = _load(ctx)
        if xn is None:
            xn = x2
        B, S, D = x.shape
        H, KVH, hd = meta.H, meta.KVH, meta.hd
        dy2 = K._rows2d(dy.contiguous())
        needs = list(ctx.needs_input_grad[4:])
        n_qkv = len(meta.qkv.tensors())
        nqkv, no = needs[:n_qkv], needs[n_qkv:]
        pend = [] if _BATCH_TN_REDUCE else None  # second stages of the four adapter-gradient products of this block: one launch at the end
        # wo
        do2, g_o = meta.wo.backward(dy2, o.view(B * S, H * hd), to, no, True, pending=pend)
        # attention
        W = (H + 2 * KVH) * hd
        dqkv = torch.empty(B, S, W, device=x.device, dtype=BF16)
        q = qkv3[..., : H * hd].unflatten(-1, (H, hd))
        k = qkv3[..., H * hd : (H + KVH) * hd].unflatten(-1, (KVH, hd))
        v = qkv3[..., (H + KVH) * hd :].unflatten(-1, (KVH, hd))
        dq = dqkv[..., : H * hd].unflatten(-1, (H, hd))
        dk = dqkv[..., H * hd : (H + KVH) * hd].unflatten(-1, (KVH, hd))
        dv = dqkv[..., (H + KVH) * hd :].unflatten(-1, (KVH, hd))
        if _FUSE_ROPE:  # apply_rope's transpose rides in the dQ epilogue and the dK/dV reduce
            K.attn_bwd(q, k, v, o, do2.view(B, S, H, hd), lse, dq, dk, dv, meta.mask, rope=rope)
        else:
            K.attn_bwd(q, k, v, o, do2.view(B, S, H, hd), lse, dq, dk, dv, meta.mask)
            K.rope_(dqkv, rope, H + KVH, backward=True)
        d2 = dqkv.view(B * S, W)
        need_dx = ctx.needs_input_grad[0]
        need_dxn = need_dx or (meta.fuse_norm and ctx.needs_input_grad[2])  # the norm weight gradient needs d(xn) too
        dxn = torch.empty(B * S, D, device=x.device, dtype=BF16) if need_dxn else None
        _, g_qkv = meta.qkv.backward(d2, xn, tqkv, nqkv, need_dxn, dxn, pending=pend)
        if pend:
            K.skinny_tn_flush(pend)
        dx = dnw = None
        if meta.fuse_norm:
            if need_dxn:
                dx, dnw = K.rmsnorm_bwd(dxn, x2, norm_w.detach(), rstd, ctx.needs_input_grad[2], dy2 if (need_dx and meta.fuse_residual) else None,
                                            dw_out=_grad_dst([norm_w]))
        else:
            dx = dxn
            if need_dx and meta.fuse_residual:
                dx = K.add(dx, dy2)
        return (dx.view(B, S, D) if (dx is not None and need_dx) else None, None, dnw, None, *g_qkv, *g_o)


# =================================================================================================
# MLP residual branch:  [x +] w2( silu(w1 xn) * w3 xn ),  xn = [rmsnorm(x)]
# =================================================================================================
_FUSE_SWIGLU_FWD = os.environ.get("LLX_FUSE_SWIGLU_FWD", "1") != "0"  # A/B knob: 0 = stand-alone swiglu_fwd kernel
_BATCH_TN_REDUCE = os.environ.get("LLX_BATCH_TN_REDUCE", "1") != "0"  # A/B knob: 0 = every adapter-gradient product reduces its partials at once
_FUSE_NORM_SKINNY = os.environ.get("LLX_FUSE_NORM_SKINNY", "1") != "0"  # A/B knob: 0 = RMSNorm, then the stand-alone skinny product
_FUSE_DY_SCALE = os.environ.get("LLX_FUSE_DY_SCALE", "1") != "0"  # A/B knob: 0 = stand-alone dy * scale pass for an int8 base's data gradient
_FUSE_U = os.environ.get("LLX_FUSE_U", "1") != "0"  # A/B knob: 0 = u = dy @ B as its own pass over dy (skinny_nt) instead of riding in dB's first stage
_BATCH_TN_PARTIAL = os.environ.get("LLX_BATCH_TN_PARTIAL", "1") != "0"  # A/B knob: 0 = dB's first stage launched on its own, before u
_HEAD_COMPACT = os.environ.get("LLX_HEAD_COMPACT", "1") != "0"  # LM head + loss over the labelled rows only (HeadLossFn)
# K ranges of the head's d-hidden GEMM (1 = unsplit).  The row count is only known on the device, so the split is static: with 4 ranges
# a round of 256 tiles lasts a quarter of the unsplit tile time, i.e. the time follows the labelled-row count in steps of 1/4 round
# (3071 rows: 12 x 16 x 4 = 768 tiles = 3 full rounds; measured -0.5 ms against one long round, 2 ranges = 1.5 rounds gain nothing)
_HEAD_SPLITK = max(1, int(os.environ.get("LLX_HEAD_SPLITK", "4")))
_HEAD_CHUNK_FORCED = "LLX_HEAD_CHUNK_ROWS" in os.environ
_HEAD_CHUNK_ROWS = max(256, int(os.environ.get("LLX_HEAD_CHUNK_ROWS", "8192")) // 256 * 256)  # rows per logits buffer of the chunked head
_FUSE_NORM_QUANT = os.environ.get("LLX_FUSE_NORM_QUANT", "1") != "0"  # A/B knob: 0 = stand-alone activation quantiser after the RMSNorm
_FUSE_ROPE = os.environ.get("LLX_FUSE_ROPE", "1") != "0"  # A/B knob: 0 = stand-alone rope kernel after the projection / before its dgrad


class MLPBlockMeta:
    def __init__(self, w13: GroupPlan, w2: GroupPlan, eps, fuse_norm, fuse_residual):
        self.w13, self.w2 = w13, w2
        self.eps, self.fuse_norm, self.fuse_residual = eps, fuse_norm, fuse_residual


class MLPBlockFn(Function):
    @staticmethod
    def forward(ctx, x: Tensor, norm_w: Optional[Tensor], meta: MLPBlockMeta, *tensors):
        K.L.require_cuda(x)
        shape = x.shape
        x2 = K._rows2d(x.contiguous())
        xq = t_pre = None
        if meta.fuse_norm and meta.w13.wants_quantized_input() and _FUSE_NORM_QUANT:
            xn, rstd, *xq = K.rmsnorm_fwd(x2, norm_w.detach(), meta.eps, quant=True)  # the norm also emits quantize_int8_rowwise(xn)
        elif meta.fuse_norm and meta.w13.norm_can_make_t(x2.shape[1]):
            xn, rstd, t_pre = K.rmsnorm_skinny_nt(x2, norm_w.detach(), meta.eps, meta.w13.packed[0])  # ... or the adapters' xn @ A_cat^T
        elif meta.fuse_norm:
            xn, rstd = K.rmsnorm_fwd(x2, norm_w.detach(), meta.eps)
        else:
            xn, rstd = x2, None
        T, I = x2.shape[0], meta.w13.Ns[0]
        gu = torch.empty(T, 2 * I, device=x.device, dtype=BF16)
        if meta.w13.swiglu_fusable() and _FUSE_SWIGLU_FWD:
            h = torch.empty(T, I, device=x.device, dtype=BF16)
            _, t13 = meta.w13.forward(xn, gu, swiglu_h=h, xq=xq, t_pre=t_pre)  # SwiGLU in the epilogue of the gate|up GEMM
        else:
            _, t13 = meta.w13.forward(xn, gu, xq=xq, t_pre=t_pre)
            h = K.swiglu_fwd(gu[:, :I], gu[:, I:])
        y, t2 = meta.w2.forward(h, None, x2 if meta.fuse_residual else None)
        ctx.meta = meta
        _save(ctx, x, norm_w, x2, xn if meta.fuse_norm else None, rstd, gu, h, t13, t2)
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy: Tensor):
        meta: MLPBlockMeta = ctx.meta
        x, norm_w, x2, xn, rstd, gu, h, t13, t2 = _load(ctx)
        if xn is None:
            xn = x2
        T, I = x2.shape[0], meta.w13.Ns[0]
        dy2 = K._rows2d(dy.contiguous())
        needs = list(ctx.needs_input_grad[3:])
        n_13 = len(meta.w13.tensors())
        n13, n2 = needs[:n_13], needs[n_13:]
        dgu = torch.empty(T, 2 * I, device=x.device, dtype=BF16)
        pend = [] if _BATCH_TN_REDUCE else None  # second stages of the four adapter-gradient products of this block: one launch at the end
        if meta.w2.fused:  # dh = dy.W2 (+LoRA) never reaches HBM: the dgrad GEMM's epilogue turns it into dg | du
            _, g_2 = meta.w2.backward(dy2, h, t2, n2, True, swiglu=(gu, dgu), pending=pend)
        else:
            dh, g_2 = meta.w2.backward(dy2, h, t2, n2, True)
            K.swiglu_bwd(dh, gu[:, :I], gu[:, I:], dgu[:, :I], dgu[:, I:])
        need_dx = ctx.needs_input_grad[0]
        need_dxn = need_dx or (meta.fuse_norm and ctx.needs_input_grad[1])
        dxn = torch.empty_like(x2) if need_dxn else None
        _, g_13 = meta.w13.backward(dgu, xn, t13, n13, need_dxn, dxn, pending=pend)
        if pend:
            K.skinny_tn_flush(pend)
        dx = dnw = None
        if meta.fuse_norm:
            if need_dxn:
                dx, dnw = K.rmsnorm_bwd(dxn, x2, norm_w.detach(), rstd, ctx.needs_input_grad[1], dy2 if (need_dx and meta.fuse_residual) else None,
                                            dw_out=_grad_dst([norm_w]))
        else:
            dx = dxn
            if need_dx and meta.fuse_residual:
                dx = K.add(dx, dy2)
        return (dx.view(x.shape) if (dx is not None and need_dx) else None, dnw, None, *g_13, *g_2)


# =================================================================================================
# final norm + LM head + cross-entropy (modelling/llama.py:216-218)
# =================================================================================================
class HeadLossFn(Function):
    """Labelled-row compaction (default; LLX_HEAD_COMPACT=0 turns it off): F.cross_entropy(ignore_index=-100) gives a position whose
    label is -100 no loss term and a zero gradient row (modelling/llama.py:216-218), so with a frozen plain head the logits GEMM, the
    loss and the d-hidden GEMM run over the labelled rows only - gathered in order on the device, the count read by the kernels from
    device memory (no host sync, capturable), d hidden scattered back with zero rows in between.  Same per-row arithmetic, same loss
    terms: every gradient is bit-identical to the uncompacted path; an SFT batch with a masked prompt saves that share of the head."""

    @staticmethod
    def forward(ctx, x: Tensor, norm_w: Tensor, labels: Tensor, eps: float, plan: LinearPlan, *tensors):
        K.L.require_cuda(x, labels)
        x2 = K._rows2d(x.contiguous())
        xn, rstd = K.rmsnorm_fwd(x2, norm_w.detach(), eps)
        need_grad = any(ctx.needs_input_grad)
        ctx.plan, ctx.eps = plan, eps
        # a plain head (no adapter, no bias, bf16).  Frozen: only d hidden is wanted from its backward.  Trainable (the reference's default,
        # train_metamathqa.py:177-180): its weight gradient dW = d logits^T . norm(x) runs over the compacted rows too - the TN kernel reads
        # the row count from device memory.
        plain = not plan.int8 and plan.rank == 0 and plan.bias is None and plan.dora_m is None
        ctx.plain_head = plain and not any(ctx.needs_input_grad[5:])
        ctx.plain_trainable = plain and not ctx.plain_head and _HEAD_COMPACT
        ctx.compact = _HEAD_COMPACT and plain
        ctx.m_expect = None
        T, V = x2.shape[0], plan.N
        # T-chunked head: one [rows, V] logits buffer is limited by the GEMM's 32-bit tile offsets (4 GiB = 16.7 k rows of a 128 k
        # vocabulary; the reference's packed [1, bs * S] batches get there at bs >= 5): beyond that the rows are walked in chunks - one
        # logits buffer, one CE launch set and (in backward) one d-hidden product per chunk, the loss normalised by the global count of
        # labelled rows.  Same per-row arithmetic: bit-identical to the unchunked path (LLX_HEAD_CHUNK_ROWS forces chunking for tests).
        chunk = _HEAD_CHUNK_ROWS if (_HEAD_CHUNK_FORCED or T * V * 2 >= 2**32) else 0
        if chunk and T > chunk:
            if not ctx.plain_head:
                raise LlxError(f"LM head over {T} rows x {V} vocabulary entries exceeds one logits buffer (4 GiB) and a trainable / adapted / "
                               "quantised head is not chunked: freeze the plain head or lower the number of positions per step")
            inv = cnt = None
            rows_in, labels_in = xn, labels.reshape(-1).contiguous()
            if ctx.compact:
                idx, inv, labels_in, cnt = K.head_compact_index(labels)
                rows_in = K.gather_rows(xn, idx, cnt)
            w = plan.weight.detach()
            ws = torch.empty(T + 2, device=x.device, dtype=torch.float32)
            loss = torch.empty((), device=x.device, dtype=torch.float32)
            parts = []
            for r0 in range(0, T, chunk):
                r1 = min(T, r0 + chunk)
                cnt_c = torch.clamp(cnt - r0, min=0, max=r1 - r0).to(torch.int32) if cnt is not None else None  # labelled rows inside this chunk
                lg = torch.empty(r1 - r0, V, device=x.device, dtype=BF16)
                K.gemm_nt(rows_in[r0:r1], w, out=lg, m_valid=cnt_c)
                parts.append((K.ce_chunk(lg, labels_in, ws, loss, r0, need_grad, cnt, r0 == 0, r1 == T), cnt_c))
            ctx.chunk = chunk
            _save(ctx, x, norm_w, x2, None, rstd, tuple(p[0] for p in parts), (inv, cnt, tuple(p[1] for p in parts)))
            return loss
        ctx.chunk = 0
        if ctx.compact:
            idx, inv, labels_c, cnt = K.head_compact_index(labels)
            xc = K.gather_rows(xn, idx, cnt)
            w = plan.weight.detach()
            logits = torch.empty(xc.shape[0], w.shape[0], device=x.device, dtype=BF16)
            ctx.m_expect = float(cnt.item()) if K.GEMM_TRACE is not None else None  # accounting of the traced eager step only
            K.gemm_nt(xc, w, out=logits, m_valid=cnt, m_expect=ctx.m_expect)
            loss, dlogits = K.ce_fwd_bwd(logits, labels_c, write_grad=need_grad, rows=cnt)
            _save(ctx, x, norm_w, x2, xc if ctx.plain_trainable else None, rstd, dlogits, (inv, cnt))
            return loss
        logits, t = plan.forward(xn)
        loss, dlogits = K.ce_fwd_bwd(logits, labels, write_grad=need_grad)
        _save(ctx, x, norm_w, x2, xn, rstd, dlogits, t)
        return loss

    @staticmethod
    def backward(ctx, gout: Tensor):
        plan: LinearPlan = ctx.plan
        x, norm_w, x2, xn, rstd, dlogits, t = _load(ctx)
        needs = ctx.needs_input_grad[5:]
        g32 = gout.detach().to(torch.float32).reshape(1)
        if ctx.plain_head or ctx.plain_trainable:
            # d hidden = d logits . W: [T, D] is ONE round of 256 tiles whatever the row count, with a contraction over the whole
            # vocabulary - cut in K ranges computed side by side, so that fewer rows (12 x 16 tiles at 3071 labelled rows) still fill
            # the chip with full rounds of short tiles.  The compacted and the uncompacted path use the same split, so they stay
            # bit-identical to each other.
            if ctx.chunk:  # chunked rows (see forward): d hidden chunk by chunk, then one scatter / scale over all rows
                inv, cnt, cnts = t
                dx = dnw = None
                if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                    wt = weight_t(plan.weight)
                    T = x2.shape[0]
                    split = wt.shape[1] % (64 * _HEAD_SPLITK) == 0
                    dxc = torch.empty(T, wt.shape[0], device=x.device, dtype=BF16)
                    for i, (dl, cnt_c) in enumerate(zip(dlogits, cnts)):
                        r0 = i * ctx.chunk
                        part = K.gemm_nt_splitk(dl, wt, _HEAD_SPLITK, m_valid=cnt_c) if split else K.gemm_nt(dl, wt, m_valid=cnt_c)
                        dxc[r0 : r0 + dl.shape[0]].copy_(part)
                    dxn = K.scatter_rows(dxc, inv, g32) if inv is not None else K.scale(dxc, dev_scalar=g32)
                    dx, dnw = K.rmsnorm_bwd(dxn, x2, norm_w.detach(), rstd, ctx.needs_input_grad[1], dw_out=_grad_dst([norm_w]))
                return (dx.view(x.shape) if dx is not None else None, dnw, None, None, None, *([None] * len(needs)))
            inv, cnt = t if ctx.compact else (None, None)
            dx = dnw = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                wt = weight_t(plan.weight)
                if wt.shape[1] % (64 * _HEAD_SPLITK) == 0:
                    dxn = K.gemm_nt_splitk(dlogits, wt, _HEAD_SPLITK, m_valid=cnt, inv=inv, dev_scalar=g32, m_expect=ctx.m_expect)
                elif ctx.compact:
                    dxn = K.scatter_rows(K.gemm_nt(dlogits, wt, m_valid=cnt, m_expect=ctx.m_expect), inv, g32)
                else:
                    dxn = K.scale(K.gemm_nt(dlogits, wt), dev_scalar=g32)
                dx, dnw = K.rmsnorm_bwd(dxn, x2, norm_w.detach(), rstd, ctx.needs_input_grad[1], dw_out=_grad_dst([norm_w]))
            gw = None
            if ctx.plain_trainable:  # dW over the labelled rows (xn holds their gathered, normed activations), scaled by the incoming gradient
                gw = K.scale(K.gemm_tn(dlogits, xn, m_valid=cnt, m_expect=ctx.m_expect), dev_scalar=g32)
            return (dx.view(x.shape) if dx is not None else None, dnw, None, None, None, *([gw] + [None] * (len(needs) - 1) if needs else []))
        dxn, grads = plan.backward(dlogits, xn, t, needs, need_dx=ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        grads = [None if g is None else K.scale(g, dev_scalar=g32) for g in grads]
        dx = dnw = None
        if dxn is not None:
            K.scale(dxn, dev_scalar=g32, out=dxn)
            dx, dnw = K.rmsnorm_bwd(dxn, x2, norm_w.detach(), rstd, ctx.needs_input_grad[1], dw_out=_grad_dst([norm_w]))
        return (dx.view(x.shape) if dx is not None else None, dnw, None, None, None, *grads)

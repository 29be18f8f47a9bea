"""Decode fast path: a transformer layer / LM head for a handful of query tokens against the KV cache, as weight-streaming kernels.

Reference semantics: the cached branch of Attention.forward (modelling/llama.py:126-127,135-137) inside TransformerLayer.forward
(:163-174) and the head of Llama.forward (:216) - same values as the generic inference path of modelling/llama.py::_run_dense, which
stays for every call this path does not take (more than 4 tokens, more than 16 query rows per kv head, biases, int8 / DoRA linears).

Per layer (M <= 4 tokens, batch 1 as KVCache is built, :189-192):
    q            = gemv([wq; wk; wv], rmsnorm(x))  + RoPE on q, k + k, v scattered into the caches        1 launch
    o            = SDPA(q, k_cache, v_cache, mask)  split over the cache, 4 heads per K/V read             2 launches
    x            = x + gemv(wo, o)                                                                         1 launch
    h            = silu(gemv(w1, rmsnorm(x))) * gemv(w3, rmsnorm(x))                                       1 launch
    x            = x + gemv(w2, h)                                                                         1 launch
LoRA-dressed linears (modelling/lora.py:43) add one small launch per group for t = x @ A^T; the B factors ride in the main launch.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from . import kernels as K
from . import ops

BF16 = torch.bfloat16
MAX_TOKENS = 4


def _plain(m: nn.Linear) -> bool:
    """A linear this path streams: bf16 weight, no bias, optionally a LoRA adapter (rank a multiple of 8)."""
    from subclasses.int8 import Int8LinearWeight

    if m.bias is not None or isinstance(m.weight, Int8LinearWeight) or m.weight.dtype is not BF16 or getattr(m, "m", None) is not None:
        return False
    rank = int(getattr(m, "rank", 0) or 0)
    return rank == 0 or (rank % 8 == 0 and m.lora_a.dtype is BF16)


def layer_ok(layer, x: Tensor, mask: Optional[Tensor]) -> bool:
    att = layer.attention
    if att.kv_cache is None or mask is None or x.dim() != 3 or x.shape[0] != 1 or x.dtype is not BF16 or not x.is_cuda:
        return False
    M = x.shape[1]
    if M > MAX_TOKENS or M * (att.num_heads // att.num_kv_heads) > 16 or att.head_dim != 128 or mask.dtype is not torch.bool:
        return False
    if x.shape[2] % 8 != 0 or layer.feed_forward.w2.in_features % 8 != 0:
        return False
    ff = layer.feed_forward
    lins = (att.wq, att.wk, att.wv, att.wo, ff.w1, ff.w3, ff.w2)
    if not all(_plain(m) for m in lins):
        return False
    for grp in ((att.wq, att.wk, att.wv), (ff.w1, ff.w3)):  # one t vector and one scale per fused group
        ranks = {int(getattr(m, "rank", 0) or 0) > 0 for m in grp}
        if len(ranks) != 1 or len({float(getattr(m, "scale", 1.0)) for m in grp}) != 1:
            return False
    return att.wq.out_features % 4 == 0 and att.wk.out_features % 4 == 0 and ff.w1.out_features == ff.w3.out_features


def _lora(mods, x: Tensor, norm):
    """(b factors, t = [rmsnorm(x) | x] @ [A_0; A_1; ..]^T, scale) or None."""
    if int(getattr(mods[0], "rank", 0) or 0) == 0:
        return None
    t = K.gemv([m.lora_a.detach() for m in mods], x, norm=norm)
    return [m.lora_b.detach() for m in mods], t, float(mods[0].scale)


def mask_extent(mask: Tensor) -> Tensor:
    """Extent of the call's mask, computed once and shared by all layers (cached on the mask tensor)."""
    return ops._cached(mask, "extent", lambda: K.mask_extent(mask))


def layer_forward(layer, x: Tensor, rope: Tensor, mask: Tensor, input_pos: Tensor) -> Tensor:
    att, ff = layer.attention, layer.feed_forward
    M, D = x.shape[1], x.shape[2]
    H, KVH, hd = att.num_heads, att.num_kv_heads, att.head_dim
    x2 = x.view(M, D)
    n1 = (layer.attention_norm.weight.detach(), layer.attention_norm.eps)
    qkv_mods = (att.wq, att.wk, att.wv)
    cache = att.kv_cache
    pos = input_pos.to(torch.int64).contiguous()
    q = K.gemv([m.weight.detach() for m in qkv_mods], x2, norm=n1, epilogue=K.GV_QKV,
               qkv=(rope, H * hd, KVH * hd, cache.k_cache, cache.v_cache, pos), lora=_lora(qkv_mods, x2, n1))
    o = K.attn_decode(q.view(1, M, H, hd).transpose(1, 2), cache.k_cache, cache.v_cache, mask, mask_extent(mask))  # [1, M, H*hd]
    o2 = o.view(M, H * hd)
    x1 = K.gemv([att.wo.weight.detach()], o2, epilogue=K.GV_RESIDUAL, res=x2, lora=_lora((att.wo,), o2, None))
    n2 = (layer.ffn_norm.weight.detach(), layer.ffn_norm.eps)
    h = K.gemv([ff.w1.weight.detach(), ff.w3.weight.detach()], x1, norm=n2, epilogue=K.GV_SWIGLU, lora=_lora((ff.w1, ff.w3), x1, n2))
    x3 = K.gemv([ff.w2.weight.detach()], h, epilogue=K.GV_RESIDUAL, res=x1, lora=_lora((ff.w2,), h, None))
    return x3.view(1, M, D)


def head_ok(model, x: Tensor) -> bool:
    return (x.dim() == 3 and x.shape[0] == 1 and x.shape[1] <= MAX_TOKENS and x.is_cuda and x.dtype is BF16 and _plain(model.output)
            and x.shape[2] % 8 == 0)


def head_forward(model, x: Tensor) -> Tensor:
    """logits = output(norm(x)) (modelling/llama.py:216) for M <= 4 rows: the 1 GB head weight streamed once."""
    M, D = x.shape[1], x.shape[2]
    x2 = x.reshape(M, D)
    nw = (model.norm.weight.detach(), model.norm.eps)
    logits = K.gemv([model.output.weight.detach()], x2, norm=nw, lora=_lora((model.output,), x2, nw))
    return logits.view(1, M, -1)

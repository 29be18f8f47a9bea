"""Llama decoder with the reference's module API (/root/reference/modelling/llama.py:17-292), executing on
hand-written gfx950 kernels.

Class names, constructor signatures, parameter / state-dict names and forward signatures match the reference so
training scripts can switch packages unchanged.  What differs is below the module boundary: every residual branch of
a layer is ONE autograd node (llx/ops.py) that calls the HIP kernels of llama-x_amd/csrc through the C-ABI; there
is no SDPA / FlexAttention / aten matmul on the hot path and no CPU fallback (CPU tensors raise).

``block_mask`` takes a :class:`llx.kernels.MaskSpec` (per-token ``doc_ids`` / per-sample ``prefix_len``) in place of
FlexAttention's BlockMask; the mask rule is the reference's ``mask_mod`` (train_metamathqa.py:67-68) plus the
prefix-LM term of README.md:16.
"""
import json
import os
from typing import NamedTuple

import torch
from torch import Tensor, nn
from torch.utils.checkpoint import checkpoint

from llx import decode as D
from llx import kernels as K
from llx import ops
from llx._lib import LlxError
from llx.kernels import MaskSpec  # noqa: F401  (re-exported: the block_mask type of this package)


class LlamaConfig(NamedTuple):
    embed_dim: int
    num_layers: int
    head_dim: int
    num_heads: int
    num_kv_heads: int
    intermediate_dim: int
    max_seq_len: int = 2048
    vocab_size: int = 128_256  # Llama3
    attn_dropout: float = 0.0
    rope_base: int = 50_000
    is_llama3_1: bool = False
    activation_checkpointing: bool = False


# ------------------------------------------------------------------------------------------------- RoPE
def scale_llama3_1_rope(freqs: Tensor) -> Tensor:
    """Llama-3.1 frequency rescale (factor 8, low 1, high 4, original context 8192), host side, fp32."""
    factor, low, high, ctx_len = 8, 1, 4, 8192
    wavelen = 2 * torch.pi / freqs
    ratio = (ctx_len / wavelen - low) / (high - low)
    blended = (1 - ratio) * freqs / factor + ratio * freqs
    long_wave = torch.where(wavelen > ctx_len / low, freqs / factor, blended)
    return torch.where(wavelen < ctx_len / high, freqs, long_wave).to(freqs.dtype)


def build_rope(config: LlamaConfig) -> Tensor:
    """fp32 [max_seq_len, head_dim/2, 2] table of (cos, sin); built on the host for bit parity of the table."""
    half = torch.arange(0, config.head_dim, 2, dtype=torch.float32) / config.head_dim
    theta = 1.0 / (config.rope_base**half)
    if config.is_llama3_1:
        theta = scale_llama3_1_rope(theta)
    angle = torch.outer(torch.arange(config.max_seq_len, dtype=torch.float32), theta)
    return torch.stack([angle.cos(), angle.sin()], dim=-1)


class _RopeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, table: Tensor):
        B, S, H, hd = x.shape
        ctx.table, ctx.shape = table, x.shape
        y = x.contiguous().clone().view(B, S, H * hd)
        return K.rope_(y, table, H).view(B, S, H, hd)

    @staticmethod
    def backward(ctx, g: Tensor):
        B, S, H, hd = ctx.shape
        y = g.contiguous().clone().view(B, S, H * hd)
        return K.rope_(y, ctx.table, H, backward=True).view(B, S, H, hd), None


def _rope_f32(rope: Tensor) -> Tensor:
    # model.bfloat16() also casts the buffer (train_metamathqa.py:176): keep those rounded values, widen for the kernel
    return rope if rope.dtype is torch.float32 else ops._cached(rope, "f32", lambda: rope.float())


def apply_rope(x: Tensor, rope: Tensor) -> Tensor:
    """Interleaved-pair rotation of x [B, S, H, 128] by the first S rows of ``rope`` (fp32 math, one rounding)."""
    return _RopeFn.apply(x, _rope_f32(rope).contiguous())


# ------------------------------------------------------------------------------------------------- leaf modules
class Linear(nn.Linear):
    """nn.Linear whose stand-alone forward runs the llx GEMM (still an nn.Linear for quantize_/adapter surgery)."""

    def forward(self, x: Tensor) -> Tensor:
        return ops.linear(x, self)


class RMSNorm(nn.RMSNorm):
    def forward(self, x: Tensor) -> Tensor:
        return ops.rmsnorm(x, self.weight, self.eps)


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids: Tensor, table: Tensor):
        ctx.save_for_backward(ids)
        ctx.vocab = table.shape[0]
        return K.embedding_fwd(ids, table.detach())

    @staticmethod
    def backward(ctx, g: Tensor):
        (ids,) = ctx.saved_tensors
        return None, K.embedding_bwd(ids, g.contiguous(), ctx.vocab).to(g.dtype)


class Embedding(nn.Embedding):
    def forward(self, ids: Tensor) -> Tensor:
        return _EmbeddingFn.apply(ids, self.weight)


class KVCache(nn.Module):
    def __init__(self, batch_size: int, config: LlamaConfig, dtype: torch.dtype):
        super().__init__()
        shape = (batch_size, config.num_kv_heads, config.max_seq_len, config.head_dim)
        self.register_buffer("k_cache", torch.zeros(shape, dtype=dtype), persistent=False)
        self.register_buffer("v_cache", torch.zeros(shape, dtype=dtype), persistent=False)

    def update(self, input_pos: Tensor, k: Tensor, v: Tensor):
        # input_pos: [S], k/v: [B, KVH, S, hd]
        assert input_pos.shape[0] == k.shape[2], (input_pos.shape, k.shape)
        if k.is_cuda and k.dtype is torch.bfloat16 and k.shape[0] == self.k_cache.shape[0] and k.stride() == v.stride() and k.stride(3) == 1:
            K.kv_scatter(k, v, self.k_cache, self.v_cache, input_pos)  # both caches from one HIP launch, straight from the q|k|v buffer's views
        else:
            self.k_cache[:, :, input_pos] = k
            self.v_cache[:, :, input_pos] = v
        return self.k_cache, self.v_cache


def _as_maskspec(block_mask):
    if block_mask is None or isinstance(block_mask, MaskSpec):
        return block_mask
    raise LlxError(
        "block_mask must be an llx MaskSpec(doc_ids=..., prefix_len=...) - FlexAttention BlockMask objects are not "
        "dispatched on this platform (the mask rule is evaluated on device from per-token metadata)"
    )


# ------------------------------------------------------------------------------------------------- blocks
class Attention(nn.Module):
    def __init__(self, config: LlamaConfig) -> None:
        super().__init__()
        self.num_heads = config.num_heads
        self.num_kv_heads = config.num_kv_heads
        self.embed_dim = config.embed_dim
        self.attn_dropout = config.attn_dropout
        self.head_dim = config.head_dim

        self.wq = Linear(self.embed_dim, self.num_heads * self.head_dim, bias=False)
        self.wk = Linear(self.embed_dim, self.num_kv_heads * self.head_dim, bias=False)
        self.wv = Linear(self.embed_dim, self.num_kv_heads * self.head_dim, bias=False)
        self.wo = Linear(self.num_heads * self.head_dim, self.embed_dim, bias=False)
        self.kv_cache = None

    def plans(self):
        return ops.GroupPlan((self.wq, self.wk, self.wv)), ops.GroupPlan((self.wo,))

    def _run(self, x: Tensor, rope: Tensor, norm: nn.Module | None, residual: bool, mask, input_pos, block_mask, plans=None) -> Tensor:
        if mask is not None and self.kv_cache is None and torch.is_grad_enabled() and (
                x.requires_grad or any(p.requires_grad for p in self.parameters()) or (norm is not None and norm.weight.requires_grad)):
            # training through the reference's dense-mask route (llama.py:135-137): a mask that the MaskSpec rule reproduces exactly
            # (causal / prefix-LM / contiguous documents) runs on the fused kernels with their backward; anything else has no backward here
            spec = ops._cached(mask, "maskspec", lambda: (K.maskspec_from_dense(mask, x.shape[0], x.shape[1]),))[0]
            if spec is None:
                raise LlxError("training with a dense mask= needs a mask of the form (k <= q or k < prefix[b]) and same-document "
                               "(contiguous documents): pass block_mask=MaskSpec(doc_ids=..., prefix_len=...) for anything else")
            mask, block_mask = None, (spec if (spec.doc_ids is not None or spec.prefix_len is not None) else None)
        if self.kv_cache is not None or mask is not None:
            return self._run_dense(x, rope, norm, residual, mask, input_pos)
        if self.training and self.attn_dropout > 0.0:
            raise LlxError("attention dropout is not supported by the HIP attention kernel (reference default is 0.0)")
        qkv, wo = plans or self.plans()
        meta = ops.AttnBlockMeta(qkv, wo, self.num_heads, self.num_kv_heads, self.head_dim, _as_maskspec(block_mask),
                                 norm.eps if norm is not None else 0.0, norm is not None, residual)
        tensors = qkv.tensors() + wo.tensors()
        return ops.AttnBlockFn.apply(x, _rope_f32(rope), norm.weight if norm is not None else None, meta, *tensors)

    def _run_dense(self, x: Tensor, rope: Tensor, norm, residual: bool, mask, input_pos) -> Tensor:
        """Inference path: KV cache and/or an explicit bool mask (SDPA branch with is_causal=False, llama.py:126-127,135-137).
        Forward only: training uses block_mask=MaskSpec(...) which has a fused backward."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise LlxError("dense-mask / KV-cache attention is forward-only here: run it under torch.no_grad() "
                           "(for training use block_mask=MaskSpec(doc_ids=..., prefix_len=...))")
        B, L_, _ = x.shape
        H, KVH, hd = self.num_heads, self.num_kv_heads, self.head_dim
        xn = norm(x) if norm is not None else x
        x2 = K._rows2d(xn.contiguous())
        qkv = torch.empty(B * L_, (H + 2 * KVH) * hd, device=x.device, dtype=x.dtype)
        ops.GroupPlan((self.wq, self.wk, self.wv)).forward(x2, qkv)
        qkv3 = qkv.view(B, L_, -1)
        K.rope_(qkv3, _rope_f32(rope).contiguous(), H + KVH)
        q = qkv3[..., : H * hd].unflatten(-1, (H, hd)).transpose(1, 2)
        k = qkv3[..., H * hd : (H + KVH) * hd].unflatten(-1, (KVH, hd)).transpose(1, 2)
        v = qkv3[..., (H + KVH) * hd :].unflatten(-1, (KVH, hd)).transpose(1, 2)
        if self.kv_cache is not None:
            k, v = self.kv_cache.update(input_pos, k, v)
        if mask is None:  # no cache, no mask cannot reach here; a cache without mask attends to everything cached
            mask = torch.ones(L_, k.shape[2], dtype=torch.bool, device=x.device)
        if L_ * (H // KVH) <= 16 and k.stride() == v.stride():  # a few query tokens over a long cache: split-cache decode kernel
            o2 = K.attn_decode(q, k, v, mask, D.mask_extent(mask)).view(B * L_, H * hd)
        else:
            o2 = K.attn_dense_fwd(q, k, v, mask).transpose(1, 2).reshape(B * L_, H * hd)  # [B,H,L,hd] -> rows
        y, _ = ops.GroupPlan((self.wo,)).forward(o2, None, K._rows2d(x.contiguous()) if residual else None)
        return y.view(B, L_, -1)

    def forward(self, x: Tensor, rope: Tensor, *, mask: Tensor | None = None, input_pos: Tensor | None = None,
                block_mask=None) -> Tensor:
        return self._run(x, rope, None, False, mask, input_pos, block_mask)


class FeedForward(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.w1 = Linear(config.embed_dim, config.intermediate_dim, bias=False)
        self.w3 = Linear(config.embed_dim, config.intermediate_dim, bias=False)
        self.w2 = Linear(config.intermediate_dim, config.embed_dim, bias=False)
        self.act = nn.SiLU()

    def plans(self):
        return ops.GroupPlan((self.w1, self.w3)), ops.GroupPlan((self.w2,))

    def _run(self, x: Tensor, norm: nn.Module | None, residual: bool, plans=None) -> Tensor:
        w13, w2 = plans or self.plans()
        meta = ops.MLPBlockMeta(w13, w2, norm.eps if norm is not None else 0.0, norm is not None, residual)
        tensors = w13.tensors() + w2.tensors()
        return ops.MLPBlockFn.apply(x, norm.weight if norm is not None else None, meta, *tensors)

    def forward(self, x: Tensor) -> Tensor:
        return self._run(x, None, False)


class TransformerLayer(nn.Module):
    def __init__(self, config: LlamaConfig) -> None:
        super().__init__()
        self.attention_norm = RMSNorm(config.embed_dim, eps=1e-5)
        self.attention = Attention(config)
        self.ffn_norm = RMSNorm(config.embed_dim, eps=1e-5)
        self.feed_forward = FeedForward(config)

    def forward(self, x: Tensor, rope: Tensor, *, mask: Tensor | None = None, input_pos: Tensor | None = None,
                block_mask=None) -> Tensor:
        # x + attention(attention_norm(x)) and x + feed_forward(ffn_norm(x)), each as one fused autograd node
        if D.layer_ok(self, x, mask) and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
            return D.layer_forward(self, x, _rope_f32(rope).contiguous(), mask, input_pos)  # decode: every linear a weight stream
        pa = pf = None
        if x.is_cuda and self.attention.kv_cache is None and mask is None:
            pa, pf = self.attention.plans(), self.feed_forward.plans()
            ops.prepack((*pa, *pf))  # the LoRA operand images of the layer's four linear groups from one launch
        x = self.attention._run(x, rope, self.attention_norm, True, mask, input_pos, block_mask, pa)
        return self.feed_forward._run(x, self.ffn_norm, True, pf)


class Llama(nn.Module):
    def __init__(self, config: LlamaConfig) -> None:
        super().__init__()
        self.tok_embeddings = Embedding(config.vocab_size, config.embed_dim)
        self.layers = nn.ModuleList([TransformerLayer(config) for _ in range(config.num_layers)])
        self.norm = RMSNorm(config.embed_dim, eps=1e-5)
        self.output = Linear(config.embed_dim, config.vocab_size, bias=False)
        self.config = config

    def build_cache(self, inference: bool = False):
        self.register_buffer("rope", build_rope(self.config), persistent=False)
        if inference:
            for layer in self.layers:
                layer.attention.kv_cache = KVCache(1, self.config, self.tok_embeddings.weight.dtype)
            L = self.config.max_seq_len
            self.register_buffer("causal_mask", torch.tril(torch.ones(L, L, dtype=torch.bool)), persistent=False)

    def _run_layers(self, x: Tensor, rope: Tensor, lo: int = 0, hi: int | None = None, **kw) -> Tensor:
        for layer in self.layers[lo:hi]:
            if self.config.activation_checkpointing:
                x = checkpoint(layer, x, rope, use_reentrant=False, **kw)
            else:
                x = layer(x, rope, **kw)
        return x

    def _head(self, x: Tensor, labels: Tensor | None) -> Tensor:
        if labels is None:
            if D.head_ok(self, x) and not (torch.is_grad_enabled() and (x.requires_grad or self.output.weight.requires_grad or self.norm.weight.requires_grad)):
                return D.head_forward(self, x)
            return self.output(self.norm(x))
        plan = ops.LinearPlan(self.output)
        return ops.HeadLossFn.apply(x, self.norm.weight, labels, self.norm.eps, plan, *plan.tensors())

    def _embed(self, x: Tensor) -> tuple[Tensor, int]:
        """(hidden states [B, S, D], number of leading positions that are dropped before the head)."""
        return self.tok_embeddings(x), 0

    def forward(self, x: Tensor, *, input_pos: Tensor | None = None, block_mask=None, labels: Tensor | None = None) -> Tensor:
        mask = self.causal_mask[None, None, input_pos] if input_pos is not None else None  # inference path (generate)
        x, _ = self._embed(x)
        rope = self.rope[: x.shape[1]]
        x = self._run_layers(x, rope, mask=mask, input_pos=input_pos, block_mask=block_mask)
        return self._head(x, labels)

    @staticmethod
    def from_hf(model_id: str, **kwargs):
        config = _get_hf_config(model_id)._replace(**kwargs)
        with torch.device("meta"):
            model = Llama(config).eval()
        # the cache cannot be built under the meta device: load (assign) first, then build
        model.load_state_dict(_get_hf_state_dict(model_id), assign=True)
        model.build_cache()
        return model


# ------------------------------------------------------------------------------------------------- HF loading
def _hf_file(model_id: str, filename: str) -> str:
    """A local checkpoint directory is used as is; otherwise the hub is asked (needs network, as in the reference)."""
    if os.path.isdir(model_id):
        return os.path.join(model_id, filename)
    from huggingface_hub import hf_hub_download

    return hf_hub_download(model_id, filename)


def _hf_list(model_id: str) -> list[str]:
    if os.path.isdir(model_id):
        return sorted(os.listdir(model_id))
    from huggingface_hub import list_repo_files

    return list_repo_files(model_id)


def _get_hf_config(model_id: str) -> LlamaConfig:
    with open(_hf_file(model_id, "config.json")) as f:
        hf = json.load(f)
    assert hf["architectures"][0] == "LlamaForCausalLM"
    config = LlamaConfig(
        embed_dim=hf["hidden_size"],
        num_layers=hf["num_hidden_layers"],
        head_dim=hf.get("head_dim", hf["hidden_size"] // hf["num_attention_heads"]),
        num_heads=hf["num_attention_heads"],
        num_kv_heads=hf["num_key_value_heads"],
        intermediate_dim=hf["intermediate_size"],
        vocab_size=hf["vocab_size"],
    )
    if "rope_theta" in hf:
        config = config._replace(rope_base=hf["rope_theta"])
    if hf.get("rope_scaling", None) is not None:
        config = config._replace(is_llama3_1=hf["rope_scaling"]["rope_type"] == "llama3")
    return config


_HF_RENAMES = (
    ("embed_tokens", "tok_embeddings"),
    ("self_attn.q_proj", "attention.wq"),
    ("self_attn.k_proj", "attention.wk"),
    ("self_attn.v_proj", "attention.wv"),
    ("self_attn.o_proj", "attention.wo"),
    ("mlp.gate_proj", "feed_forward.w1"),
    ("mlp.up_proj", "feed_forward.w3"),
    ("mlp.down_proj", "feed_forward.w2"),
    ("input_layernorm", "attention_norm"),
    ("post_attention_layernorm", "ffn_norm"),
    ("lm_head", "output"),
)


def _rename_hf_key(key: str) -> str:
    key = key.removeprefix("model.")
    for old, new in _HF_RENAMES:
        key = key.replace(old, new)
    return key


def _get_hf_state_dict(model_id: str) -> dict:
    names = _hf_list(model_id)
    filenames = [n for n in names if n.endswith(".safetensors")] or [n for n in names if n.endswith(".bin")]
    if not filenames:
        raise RuntimeError(f"No weights found for {model_id=}")
    state = {}
    for name in filenames:
        path = _hf_file(model_id, name)
        if path.endswith(".safetensors"):
            import safetensors

            with safetensors.safe_open(path, framework="pt") as f:
                for k in f.keys():
                    state[k] = f.get_tensor(k)
        else:
            state.update(torch.load(path, map_location="cpu", weights_only=True, mmap=True))
    return {_rename_hf_key(k): v for k, v in state.items()}

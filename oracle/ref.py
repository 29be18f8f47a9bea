"""CPU oracle for the llama-x hot path.  TEST INFRASTRUCTURE ONLY.

A functional, parameter-dict restatement (PyTorch CPU ops + numpy) of what the reference computes on
its training hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module; the product (``llama-x_amd/``) never does.

Pinning: every function here is checked against the reference itself (imported on CPU from
/root/reference by ``oracle/gen_golden.py``) and against the committed fixtures in ``tests/golden``
(``tests/test_oracle_golden.py``).  The one exception is :func:`mel_spectrogram`: its arithmetic lives
in torchaudio (unpinned version, not installed, not under /root/reference) -> **parity unpinned**; it
follows torchaudio's documented semantics and is cross-checked against transformers.audio_utils.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace
from typing import Iterable, Iterator, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


# --------------------------------------------------------------------------------------------------
# configuration (field names/defaults of LlamaConfig, modelling/llama.py:17-29; AudioConfig, modelling/audio.py:12-17)
# --------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Cfg:
    embed_dim: int
    num_layers: int
    head_dim: int
    num_heads: int
    num_kv_heads: int
    intermediate_dim: int
    max_seq_len: int = 2048
    vocab_size: int = 128_256
    attn_dropout: float = 0.0
    rope_base: int = 50_000
    is_llama3_1: bool = False
    activation_checkpointing: bool = False

    def _replace(self, **kw):
        return replace(self, **kw)


@dataclass(frozen=True)
class AudioCfg:
    sample_rate: int = 16_000
    n_fft: int = 512
    win_length: int = 400
    hop_length: int = 160
    n_mels: int = 128


TINY = Cfg(embed_dim=512, num_layers=2, head_dim=128, num_heads=4, num_kv_heads=1, intermediate_dim=1792,
           max_seq_len=512, vocab_size=1024, rope_base=500_000, is_llama3_1=True)
LLAMA31_8B = Cfg(embed_dim=4096, num_layers=32, head_dim=128, num_heads=32, num_kv_heads=8, intermediate_dim=14336,
                 max_seq_len=4096, vocab_size=128_256, rope_base=500_000, is_llama3_1=True)


# --------------------------------------------------------------------------------------------------
# deterministic inputs: numpy PCG64 keyed by tensor name (no torch RNG, no reference code needed)
# --------------------------------------------------------------------------------------------------
def _seed_of(name: str, seed: int) -> int:
    import zlib

    return (zlib.crc32(name.encode()) + 0x9E3779B1 * (seed + 1)) % (2**63)


def randn(name: str, shape: Sequence[int], std: float = 1.0, seed: int = 1234, dtype=torch.float32) -> Tensor:
    g = np.random.Generator(np.random.PCG64(_seed_of(name, seed)))
    a = g.standard_normal(size=tuple(shape), dtype=np.float32) * np.float32(std)
    return torch.from_numpy(a).to(dtype)


def randint(name: str, shape: Sequence[int], low: int, high: int, seed: int = 1234) -> Tensor:
    g = np.random.Generator(np.random.PCG64(_seed_of(name, seed)))
    return torch.from_numpy(g.integers(low, high, size=tuple(shape), dtype=np.int64))


def uniform(name: str, shape: Sequence[int], lo: float, hi: float, seed: int = 1234) -> Tensor:
    g = np.random.Generator(np.random.PCG64(_seed_of(name, seed)))
    return torch.from_numpy(g.uniform(lo, hi, size=tuple(shape)).astype(np.float32))


def init_params(cfg: Cfg, *, seed: int = 1234, std: float = 0.02, dtype=torch.float32, audio: bool = False,
                n_mels: int = 128) -> dict[str, Tensor]:
    """Synthetic weights under the reference's state-dict key names (modelling/llama.py:102-105,146-148,158-160,180-183)."""
    D, I, V = cfg.embed_dim, cfg.intermediate_dim, cfg.vocab_size
    hq, hkv = cfg.num_heads * cfg.head_dim, cfg.num_kv_heads * cfg.head_dim
    p: dict[str, Tensor] = {}

    def w(name, shape, s=std):
        p[name] = randn(name, shape, s, seed).to(dtype)

    w("tok_embeddings.weight", (V, D))
    for i in range(cfg.num_layers):
        pre = f"layers.{i}."
        w(pre + "attention.wq.weight", (hq, D))
        w(pre + "attention.wk.weight", (hkv, D))
        w(pre + "attention.wv.weight", (hkv, D))
        w(pre + "attention.wo.weight", (D, hq))
        w(pre + "feed_forward.w1.weight", (I, D))
        w(pre + "feed_forward.w3.weight", (I, D))
        w(pre + "feed_forward.w2.weight", (D, I))
        p[pre + "attention_norm.weight"] = (1.0 + randn(pre + "attention_norm.weight", (D,), 0.05, seed)).to(dtype)
        p[pre + "ffn_norm.weight"] = (1.0 + randn(pre + "ffn_norm.weight", (D,), 0.05, seed)).to(dtype)
    p["norm.weight"] = (1.0 + randn("norm.weight", (D,), 0.05, seed)).to(dtype)
    w("output.weight", (V, D))
    if audio:
        w("audio_embed.0.weight", (D, n_mels, 3), 0.05)
        w("audio_embed.0.bias", (D,), 0.05)
        w("audio_embed.2.weight", (D, D, 3), 0.02)
        w("audio_embed.2.bias", (D,), 0.05)
    return p


LINEAR_SUFFIXES = ("attention.wq", "attention.wk", "attention.wv", "attention.wo",
                   "feed_forward.w1", "feed_forward.w3", "feed_forward.w2")


def init_lora(cfg: Cfg, rank: int, *, seed: int = 1234, dtype=torch.float32, b_std: float = 0.01) -> dict[str, Tensor]:
    """LoRA factors for every linear under ``layers`` (shapes modelling/lora.py:31-32).

    The reference draws A kaiming-normal(a=sqrt(5)) from torch's RNG and zeros B (modelling/lora.py:34-35);
    parity runs use these deterministic draws instead (A std = sqrt(2/(1+5))/sqrt(in) as kaiming-normal gives,
    B ~ N(0, b_std^2) so that dA != 0).
    """
    D, I = cfg.embed_dim, cfg.intermediate_dim
    hq, hkv = cfg.num_heads * cfg.head_dim, cfg.num_kv_heads * cfg.head_dim
    shapes = {"attention.wq": (hq, D), "attention.wk": (hkv, D), "attention.wv": (hkv, D), "attention.wo": (D, hq),
              "feed_forward.w1": (I, D), "feed_forward.w3": (I, D), "feed_forward.w2": (D, I)}
    p = {}
    for i in range(cfg.num_layers):
        for suf, (o, n) in shapes.items():
            key = f"layers.{i}.{suf}"
            p[key + ".lora_a"] = randn(key + ".lora_a", (rank, n), math.sqrt(2.0 / 6.0) / math.sqrt(n), seed).to(dtype)
            p[key + ".lora_b"] = randn(key + ".lora_b", (o, rank), b_std, seed).to(dtype)
    return p


def init_dora_m(p: dict, cfg: Cfg) -> dict[str, Tensor]:
    """DoRA magnitude vectors for every linear of the layers: the reference's initial value ||W||_row (modelling/lora.py:51)
    moved off it by a seeded +-10 % factor, so that tests see m != norm."""
    out = {}
    for i in range(cfg.num_layers):
        for suf in LINEAR_SUFFIXES:
            w = p[f"layers.{i}.{suf}.weight"].float()
            out[f"layers.{i}.{suf}.m"] = w.norm(p=2, dim=1) * (1 + 0.1 * randn(f"dora_m.{i}.{suf}", (w.shape[0],)))
    return out


# --------------------------------------------------------------------------------------------------
# RoPE  (modelling/llama.py:32-73)
# --------------------------------------------------------------------------------------------------
def llama31_rescale(freqs: Tensor) -> Tensor:
    """Piecewise frequency rescale of Llama-3.1 (modelling/llama.py:32-51), vectorised.

    Constants: factor 8, low 1, high 4, original context 8192.  Same fp32 operation order per element
    as the reference's per-frequency loop, so the result is bit-identical (checked in gen_golden.py).
    """
    factor, low, high, old_ctx = 8, 1, 4, 8192
    wavelen = 2 * torch.pi / freqs
    smooth = (old_ctx / wavelen - low) / (high - low)
    mid = (1 - smooth) * freqs / factor + smooth * freqs
    out = torch.where(wavelen < old_ctx / high, freqs, torch.where(wavelen > old_ctx / low, freqs / factor, mid))
    return out.to(freqs.dtype)


def rope_table(cfg: Cfg) -> Tensor:
    """fp32 [max_seq_len, head_dim/2, 2] table of (cos, sin) (modelling/llama.py:54-60)."""
    expo = torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim
    theta = 1.0 / (cfg.rope_base**expo)
    if cfg.is_llama3_1:
        theta = llama31_rescale(theta)
    pos = torch.arange(cfg.max_seq_len, dtype=torch.float32)
    ang = torch.outer(pos, theta)
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1)


def rope_apply(x: Tensor, table: Tensor) -> Tensor:
    """Interleaved-pair rotation in fp32, cast back (modelling/llama.py:63-73). x: [B,S,H,hd]; table: [S,hd/2,2]."""
    B, S, H, hd = x.shape
    t = table[:S].view(1, S, 1, hd // 2, 2)
    xf = x.float().view(B, S, H, hd // 2, 2)
    x0, x1 = xf[..., 0], xf[..., 1]
    c, s = t[..., 0], t[..., 1]
    out = torch.stack([x0 * c - x1 * s, x1 * c + x0 * s], dim=-1)
    return out.view(B, S, H, hd).to(x.dtype)


# --------------------------------------------------------------------------------------------------
# norms / masks / attention / MLP  (modelling/llama.py:93-174)
# --------------------------------------------------------------------------------------------------
def rmsnorm(x: Tensor, w: Tensor, eps: float = 1e-5) -> Tensor:
    """nn.RMSNorm(D, eps=1e-5) (modelling/llama.py:158,160,182): fp32 internally, one rounding at the end."""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w.float()
    return y.to(x.dtype)


def causal_mask(S: int) -> Tensor:
    """is_causal == tril mask (modelling/llama.py:135-137, :194)."""
    return torch.tril(torch.ones(S, S, dtype=torch.bool))


def document_mask(doc_ids: Tensor) -> Tensor:
    """mask_mod of the packed iterator (train_metamathqa.py:67-68): same document AND q >= kv. -> bool [S,S]."""
    S = doc_ids.shape[0]
    idx = torch.arange(S)
    return (doc_ids[:, None] == doc_ids[None, :]) & (idx[:, None] >= idx[None, :])


def prefix_lm_mask(S: int, prefix_len: Tensor | Sequence[int]) -> Tensor:
    """Prefix-LM mask (README.md:16 plan): allow(q,kv) = kv < P_b or q >= kv. -> bool [B,1,S,S]."""
    P = torch.as_tensor(prefix_len, dtype=torch.int64).view(-1, 1, 1, 1)
    q = torch.arange(S).view(1, 1, S, 1)
    kv = torch.arange(S).view(1, 1, 1, S)
    return (kv < P) | (q >= kv)


def sdpa(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor]) -> Tensor:
    """softmax(QK^T/sqrt(hd) + mask) V with GQA by head grouping (modelling/llama.py:129-137).

    q: [B,H,S,hd]; k,v: [B,KVH,S,hd]; mask: bool broadcastable to [B,H,S,S] (True = attend) or None = causal.
    Scores and softmax are fp32; the output is cast to q.dtype.
    """
    B, H, S, hd = q.shape
    g = H // k.shape[1]
    kf = k.float().repeat_interleave(g, dim=1)
    vf = v.float().repeat_interleave(g, dim=1)
    s = (q.float() @ kf.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if mask is None:
        mask = causal_mask(S)
    s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ vf).to(q.dtype)


def linear(x: Tensor, p: dict, key: str, lora_scale: float = 1.0) -> Tensor:
    """F.linear with optional LoRA / int8 dressing found in ``p`` under ``key`` (modelling/lora.py:40-44,
    subclasses/int8.py:106-121).  Adapter order follows LoRALinear.forward: base + ((x@A^T)@B^T)*scale."""
    if key + ".m" in p:  # DoRALinear (modelling/lora.py:47-62); dora_linear is defined below
        return dora_linear(x, p[key + ".weight"], p[key + ".lora_a"], p[key + ".lora_b"], p[key + ".m"], lora_scale, p.get(key + ".bias"))
    if key + ".int_data" in p:
        out = int8_linear(x, p[key + ".int_data"], p[key + ".scale"], dynamic=bool(p.get(key + ".dynamic", False)))
    else:
        out = F.linear(x, p[key + ".weight"], p.get(key + ".bias"))
    if key + ".lora_a" in p:
        out = out + x @ p[key + ".lora_a"].T @ p[key + ".lora_b"].T * lora_scale
    return out


def attention(x: Tensor, p: dict, pre: str, cfg: Cfg, table: Tensor, mask: Optional[Tensor], lora_scale: float = 1.0) -> Tensor:
    """Attention.forward (modelling/llama.py:108-140): q/k/v linears, RoPE on q,k, SDPA with GQA, wo."""
    B, S, _ = x.shape
    q = linear(x, p, pre + "wq", lora_scale).view(B, S, cfg.num_heads, cfg.head_dim)
    k = linear(x, p, pre + "wk", lora_scale).view(B, S, cfg.num_kv_heads, cfg.head_dim)
    v = linear(x, p, pre + "wv", lora_scale).view(B, S, cfg.num_kv_heads, cfg.head_dim)
    q = rope_apply(q, table).transpose(1, 2)
    k = rope_apply(k, table).transpose(1, 2)
    v = v.transpose(1, 2)
    o = sdpa(q, k, v, mask)
    o = o.transpose(1, 2).reshape(B, S, cfg.num_heads * cfg.head_dim)
    return linear(o, p, pre + "wo", lora_scale)


def feed_forward(x: Tensor, p: dict, pre: str, lora_scale: float = 1.0) -> Tensor:
    """w2(silu(w1 x) * w3 x) (modelling/llama.py:151-152)."""
    return linear(F.silu(linear(x, p, pre + "w1", lora_scale)) * linear(x, p, pre + "w3", lora_scale), p, pre + "w2", lora_scale)


def layer(x: Tensor, p: dict, i: int, cfg: Cfg, table: Tensor, mask: Optional[Tensor], lora_scale: float = 1.0) -> Tensor:
    """TransformerLayer.forward (modelling/llama.py:163-174): two pre-norm residual branches."""
    pre = f"layers.{i}."
    x = x + attention(rmsnorm(x, p[pre + "attention_norm.weight"]), p, pre + "attention.", cfg, table, mask, lora_scale)
    x = x + feed_forward(rmsnorm(x, p[pre + "ffn_norm.weight"]), p, pre + "feed_forward.", lora_scale)
    return x


def cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """F.cross_entropy(logits.view(-1,V).float(), labels.view(-1)) with ignore_index -100, mean (modelling/llama.py:218)."""
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), labels.reshape(-1))


def llama_forward(tokens: Tensor, p: dict, cfg: Cfg, *, mask: Optional[Tensor] = None, labels: Optional[Tensor] = None,
                  lora_scale: float = 1.0) -> Tensor:
    """Llama.forward (modelling/llama.py:196-219). mask None = causal; else dense bool mask (document / prefix)."""
    x = F.embedding(tokens, p["tok_embeddings.weight"])
    table = rope_table(cfg)[: x.shape[1]]
    for i in range(cfg.num_layers):
        x = layer(x, p, i, cfg, table, mask, lora_scale)
    x = F.linear(rmsnorm(x, p["norm.weight"]), p["output.weight"])
    if labels is not None:
        return cross_entropy(x, labels)
    return x


def llama_forward_cached(tokens: Tensor, p: dict, cfg: Cfg, cache: dict, input_pos: Tensor) -> Tensor:
    """Inference path of Llama.forward with KV caches (modelling/llama.py:83-90,126-127,135-137,189-194,205-207).

    ``cache[i] = (k_cache, v_cache)`` of shape [1, KVH, max_seq_len, hd] (zeros initially), updated in place at
    ``input_pos``; the mask is ``tril[input_pos]`` over the WHOLE cache length.  Reference quirk reproduced: the RoPE rows are
    ``rope[:L]`` of the current call (:207), i.e. positions restart at 0 for every call regardless of ``input_pos``."""
    L = tokens.shape[1]
    x = F.embedding(tokens, p["tok_embeddings.weight"])
    table = rope_table(cfg)[:L]
    mask = torch.tril(torch.ones(cfg.max_seq_len, cfg.max_seq_len, dtype=torch.bool))[None, None, input_pos]
    for i in range(cfg.num_layers):
        pre = f"layers.{i}."
        h = rmsnorm(x, p[pre + "attention_norm.weight"])
        B = h.shape[0]
        q = linear(h, p, pre + "attention.wq").view(B, L, cfg.num_heads, cfg.head_dim)
        k = linear(h, p, pre + "attention.wk").view(B, L, cfg.num_kv_heads, cfg.head_dim)
        v = linear(h, p, pre + "attention.wv").view(B, L, cfg.num_kv_heads, cfg.head_dim)
        q = rope_apply(q, table).transpose(1, 2)
        k = rope_apply(k, table).transpose(1, 2)
        v = v.transpose(1, 2)
        kc, vc = cache[i]
        kc[:, :, input_pos] = k
        vc[:, :, input_pos] = v
        o = sdpa(q, kc, vc, mask).transpose(1, 2).reshape(B, L, -1)
        x = x + linear(o, p, pre + "attention.wo")
        x = x + feed_forward(rmsnorm(x, p[pre + "ffn_norm.weight"]), p, pre + "feed_forward.")
    return F.linear(rmsnorm(x, p["norm.weight"]), p["output.weight"])


def new_cache(cfg: Cfg, dtype=torch.float32) -> dict:
    shape = (1, cfg.num_kv_heads, cfg.max_seq_len, cfg.head_dim)
    return {i: (torch.zeros(shape, dtype=dtype), torch.zeros(shape, dtype=dtype)) for i in range(cfg.num_layers)}


# --------------------------------------------------------------------------------------------------
# LoRA / DoRA  (modelling/lora.py:19-62)
# --------------------------------------------------------------------------------------------------
def lora_linear(x: Tensor, w: Tensor, a: Tensor, b: Tensor, scale: float, bias: Optional[Tensor] = None) -> Tensor:
    """LoRALinear.forward (modelling/lora.py:40-44): F.linear(x,W,b) + x @ A^T @ B^T * scale."""
    return F.linear(x, w, bias) + x @ a.T @ b.T * scale


def dora_linear(x: Tensor, w: Tensor, a: Tensor, b: Tensor, m: Tensor, scale: float, bias: Optional[Tensor] = None) -> Tensor:
    """DoRALinear.forward (modelling/lora.py:53-62): LoRA output rescaled per out-row by m / ||W + s*B@A||_2."""
    out = F.linear(x, w) + x @ a.T @ b.T * scale
    dw = b.detach() @ a.detach() * scale
    out = out * (m / (w + dw).norm(p=2, dim=1))
    if bias is not None:
        out = out + bias
    return out


# --------------------------------------------------------------------------------------------------
# int8  (subclasses/int8.py:10-16,106-130 ; subclasses/int8_mm.py:93-118)
# --------------------------------------------------------------------------------------------------
def quantize_int8_rowwise(x: Tensor) -> tuple[Tensor, Tensor]:
    """Row-wise absmax int8 (subclasses/int8.py:10-16): fp32 scale = absmax/127, divide by clip(scale,1e-12),
    round half-to-even, int8; the scale is returned in the input dtype."""
    xf = x.float()
    scale = xf.abs().amax(1) / 127
    q = (xf / scale.clip(1e-12).view(-1, 1)).round().to(torch.int8)
    return q, scale.to(x.dtype)


def int8_mm_dequant(a_i8: Tensor, b_i8: Tensor, a_scale: Tensor, b_scale: Tensor) -> Tensor:
    """torchao::int8_mm_dequant (subclasses/int8_mm.py:93-118): int32 accumulate, fp32 row*col scale, cast to the
    scale dtype.  Integer accumulation is order independent => bit-exact target."""
    # The integer sum as an fp64 matrix product: every partial sum is an integer below 127^2 * K < 2^53 for any K < 5e11, so fp64
    # accumulation is exact in any order (the int32 matmul of CPU torch has no BLAS path: minutes at the 8B shapes, seconds this way).
    K = a_i8.shape[-1]
    assert 127 * 127 * K < 2 ** 31, "the reference's int32 accumulator would overflow"
    acc = (a_i8.to(torch.float64) @ b_i8.to(torch.float64)).to(torch.int32)
    out = acc.float() * a_scale.float().view(-1, 1) * b_scale.float().view(1, -1)
    return out.to(a_scale.dtype)


def int8_linear_grad_input(grad_out: Tensor, w_i8: Tensor, w_scale: Tensor) -> Tensor:
    """_Int8Linear.backward (subclasses/int8.py:124-127): (g * scale) @ W_i8.to(g.dtype)."""
    return (grad_out * w_scale) @ w_i8.to(grad_out.dtype)


class _Int8LinearRef(torch.autograd.Function):
    """_Int8Linear (subclasses/int8.py:106-130): the forward may quantise the activations (non-differentiable rounding),
    the backward is always the bf16/fp32 dequantised product - so it has to be an explicit autograd function."""

    @staticmethod
    def forward(ctx, x, w_i8, w_scale, dynamic):
        ctx.save_for_backward(w_i8, w_scale)
        if dynamic:
            xi, xs = quantize_int8_rowwise(x.reshape(-1, w_i8.shape[1]))
            return int8_mm_dequant(xi, w_i8.T, xs, w_scale.to(xs.dtype)).view(*x.shape[:-1], -1)
        return (x @ w_i8.T.to(x.dtype)) * w_scale

    @staticmethod
    def backward(ctx, g):
        w_i8, w_scale = ctx.saved_tensors
        return int8_linear_grad_input(g, w_i8, w_scale), None, None, None


def int8_linear(x: Tensor, w_i8: Tensor, w_scale: Tensor, dynamic: bool = False, bias: Optional[Tensor] = None) -> Tensor:
    """F.linear on an Int8LinearWeight (subclasses/int8.py:60-64,106-121)."""
    out = _Int8LinearRef.apply(x, w_i8, w_scale, dynamic)
    return out + bias if bias is not None else out


def int8_dequantize(w_i8: Tensor, w_scale: Tensor) -> Tensor:
    """Int8LinearWeight.dequantize (subclasses/int8.py:50-51)."""
    return w_i8 * w_scale.view(-1, 1)


# --------------------------------------------------------------------------------------------------
# audio front end  (modelling/audio.py:26-77).  mel_spectrogram: PARITY UNPINNED (torchaudio absent).
# --------------------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f: np.ndarray) -> np.ndarray:
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    with np.errstate(divide="ignore"):
        mel = np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mel)
    return mel


def _mel_to_hz_slaney(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    f = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)


def mel_filterbank(ac: AudioCfg = AudioCfg()) -> Tensor:
    """Slaney-scale, slaney-normalised triangular filterbank, fp32 [n_fft/2+1, n_mels]
    (torchaudio.functional.melscale_fbanks semantics: f_min 0, f_max sr/2; modelling/audio.py:35)."""
    n_freqs = ac.n_fft // 2 + 1
    all_freqs = torch.linspace(0, ac.sample_rate // 2, n_freqs)
    m_pts = torch.linspace(float(_hz_to_mel_slaney(0.0)), float(_hz_to_mel_slaney(ac.sample_rate / 2)), ac.n_mels + 2)
    f_pts = torch.from_numpy(_mel_to_hz_slaney(m_pts.numpy())).float()
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.minimum(down, up), min=0.0)
    enorm = 2.0 / (f_pts[2 : ac.n_mels + 2] - f_pts[: ac.n_mels])
    return fb * enorm.unsqueeze(0)


def mel_spectrogram(audio: Tensor, ac: AudioCfg = AudioCfg()) -> Tensor:
    """MelSpectrogram(sample_rate,n_fft,win_length,hop_length,n_mels,norm="slaney",mel_scale="slaney")
    (modelling/audio.py:35): power STFT (hann periodic window, centre/reflect padding) -> mel. fp32 [B,n_mels,1+L//hop]."""
    win = torch.hann_window(ac.win_length, periodic=True, dtype=torch.float32)
    spec = torch.stft(audio.float(), n_fft=ac.n_fft, hop_length=ac.hop_length, win_length=ac.win_length, window=win,
                      center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.abs().pow(2.0)
    fb = mel_filterbank(ac)
    return (power.transpose(-1, -2) @ fb).transpose(-1, -2)


def log_mel_cmn(mel: Tensor) -> Tensor:
    """mel[..., :-1].clip(1e-12).log10() minus per-bin mean over time (modelling/audio.py:53-54)."""
    a = mel[..., :-1].clip(1e-12).log10()
    return a - a.mean(2, keepdim=True)


def audio_embed(feat: Tensor, p: dict) -> Tensor:
    """Conv1d(n_mels,D,3,1,1) GELU Conv1d(D,D,3,2,1) GELU, then [B,L,D] (modelling/audio.py:26-31,59-60)."""
    h = F.gelu(F.conv1d(feat, p["audio_embed.0.weight"], p["audio_embed.0.bias"], stride=1, padding=1))
    h = F.gelu(F.conv1d(h, p["audio_embed.2.weight"], p["audio_embed.2.bias"], stride=2, padding=1))
    return h.transpose(1, 2)


def llama_audio_forward(audio: Optional[Tensor], tokens: Tensor, p: dict, cfg: Cfg, ac: AudioCfg = AudioCfg(), *,
                        labels: Optional[Tensor] = None, mask: Optional[Tensor] = None, mel: Optional[Tensor] = None,
                        lora_scale: float = 1.0) -> Tensor:
    """LlamaAudio.forward (modelling/audio.py:38-77).  ``mel`` may be supplied to bypass the (unpinned) STFT front end.
    mask None = causal over [audio ; text] as the reference runs it; a prefix-LM mask may be passed."""
    x = F.embedding(tokens, p["tok_embeddings.weight"])
    n_audio = 0
    if audio is not None or mel is not None:
        if mel is None:
            mel = mel_spectrogram(audio, ac)
        feat = log_mel_cmn(mel).to(p["tok_embeddings.weight"].dtype)
        a = audio_embed(feat, p)
        n_audio = a.shape[1]
        x = torch.cat([a, x], dim=1)
    table = rope_table(cfg)[: x.shape[1]]
    for i in range(cfg.num_layers):
        x = layer(x, p, i, cfg, table, mask, lora_scale)
    x = x[:, n_audio:]
    x = F.linear(rmsnorm(x, p["norm.weight"]), p["output.weight"])
    if labels is not None:
        return cross_entropy(x, labels)
    return x


# --------------------------------------------------------------------------------------------------
# host-side index / label / mask construction (bit-exact contracts M1, M2, M4) and step semantics (M3)
# --------------------------------------------------------------------------------------------------
def next_multiple(x: int, n: int) -> int:
    """train_metamathqa.py:25-26."""
    return (x + n - 1) // n * n


def pad_batch(tokens_batch: Sequence[Tensor], seq_len_multiple: int = 256) -> tuple[Tensor, Tensor]:
    """One batch of _data_iter_padding (train_metamathqa.py:38-46): inputs = tokens[:-1] zero padded,
    labels = tokens[1:] padded with -100, both to max(next_multiple(len-1, multiple))."""
    B = len(tokens_batch)
    L = max(next_multiple(int(t.shape[0]) - 1, seq_len_multiple) for t in tokens_batch)
    inputs = torch.zeros(B, L, dtype=torch.int64)
    labels = torch.full((B, L), -100, dtype=torch.int64)
    for r, t in enumerate(tokens_batch):
        n = int(t.shape[0]) - 1
        inputs[r, :n] = t[:-1]
        labels[r, :n] = t[1:]
    return inputs, labels


def pack_documents(docs: Iterable[Tensor], seq_len: int, *, state: Optional[dict] = None) -> Iterator[tuple[Tensor, Tensor, Tensor]]:
    """Greedy packer of _data_iter_document_mask (train_metamathqa.py:51-83) over ONE pass of ``docs``
    (the caller owns shuffling/epochs).  Yields (inputs[S], labels[S], doc_ids[S]) each time the next document
    would overflow (test ``i + len(tokens) - 1 > seq_len``, :64).  Reproduces the reference quirks: the document
    counter is never reset across buffers (:56,:83) while doc_ids is re-zeroed per buffer (:75), so the unused
    tail carries id 0.  ``state`` carries (buffers, i, doc_idx) across passes like the reference's outer loop."""
    st = state if state is not None else {}
    if "inputs" not in st:
        st.update(inputs=torch.zeros(seq_len, dtype=torch.int64), labels=torch.full((seq_len,), -100, dtype=torch.int64),
                  doc_ids=torch.zeros(seq_len, dtype=torch.int64), i=0, doc_idx=0)
    for tokens in docs:
        if st["i"] + len(tokens) - 1 > seq_len:
            yield st["inputs"], st["labels"], st["doc_ids"]
            st.update(inputs=torch.zeros(seq_len, dtype=torch.int64), labels=torch.full((seq_len,), -100, dtype=torch.int64),
                      doc_ids=torch.zeros(seq_len, dtype=torch.int64), i=0)
        n = len(tokens) - 1
        i = st["i"]
        st["inputs"][i : i + n] = tokens[:-1]
        st["labels"][i : i + n] = tokens[1:]
        st["doc_ids"][i : i + n] = st["doc_idx"]
        st["i"] = i + n
        st["doc_idx"] += 1


def prepare_audio_batch(batch: Sequence[tuple[Tensor, list[int]]], audio_length: int, seq_len_multiple: int, pad_id: int):
    """LibriSpeech._prepare_batch (train_librispeech.py:68-86)."""
    audios, toks = zip(*batch)
    audio = torch.stack([F.pad(a, (0, audio_length - a.shape[0])) for a in audios], dim=0)
    L = math.ceil(max(len(t) for t in toks) / seq_len_multiple) * seq_len_multiple
    tokens, labels = [], []
    for t in toks:
        pad = L - len(t)
        tokens.append(list(t) + [pad_id] * pad)
        labels.append(list(t[1:]) + [-100] * (pad + 1))
    return audio, torch.tensor(tokens), torch.tensor(labels)


def list_transcripts(trans_files: Sequence[tuple[str, str, Sequence[str]]], tokenize) -> list[tuple[str, list[int]]]:
    """LibriSpeech.__init__ sample listing (train_librispeech.py:53-63), over an in-memory stand-in of the
    ``**/*.trans.txt`` glob: ``trans_files`` = (directory relative to data_dir, file name, lines).  Reproduces the reference's
    de-indented body (:58-61): the three statements after the inner ``for line`` loop run ONCE per transcript file, with the
    variables of its LAST line, so each file contributes exactly one sample.  Sorted as :63."""
    samples = []
    for rel_dir, _fname, lines in trans_files:
        for line in lines:
            audio_fname, text = line.rstrip().split(" ", 1)
        audio_path = f"{rel_dir}/{audio_fname}.flac" if rel_dir else f"{audio_fname}.flac"
        samples.append((audio_path, tokenize(f" {text.lower()}.")))
    samples.sort()
    return samples


def pack_utterances(samples: Sequence[tuple[str, list[int]]], load_audio, order: Iterable[int], *, audio_duration: float,
                    sample_rate: int, batch_size: int, seq_len_multiple: int, bos_id: int, eos_id: int, pad_id: int,
                    state: Optional[dict] = None) -> Iterator[tuple[Tensor, Tensor, Tensor]]:
    """LibriSpeech.__iter__ (train_librispeech.py:88-124) over ONE pass of ``order`` (the caller owns the
    ``torch.randperm`` / endless ``while True`` of :95-98; ``state`` carries batch/audio/tokens/duration across passes).
    ``load_audio(path) -> (waveform [channels, n], fs)`` stands in for ``torchaudio.load`` (:101).  Utterances are averaged
    over channels (:103), clips longer than ``audio_duration`` are skipped (:106-108); when the next clip would overflow the
    packed duration the pack is closed with eos (:110-112), appended to the batch, and a full batch is emitted through
    _prepare_batch (:114-116); the overflowing clip then opens the next pack (:118-124)."""
    st = state if state is not None else {}
    if "tokens" not in st:
        st.update(batch=[], audio=[], tokens=[bos_id], duration=0)
    for idx in order:
        this_audio_path, this_tokens = samples[int(idx)]
        this_audio, fs = load_audio(this_audio_path)
        assert fs == sample_rate
        this_audio = this_audio.mean(0)
        this_duration = this_audio.shape[0] / fs
        if this_duration > audio_duration:
            continue
        if st["duration"] + this_duration > audio_duration:
            audio = torch.cat(st["audio"], dim=0)
            st["tokens"].append(eos_id)
            st["batch"].append((audio, st["tokens"]))
            if len(st["batch"]) == batch_size:
                yield prepare_audio_batch(st["batch"], int(audio_duration * sample_rate), seq_len_multiple, pad_id)
                st["batch"] = []
            st["audio"] = []
            st["tokens"] = [bos_id]
            st["duration"] = 0
        st["audio"].append(this_audio)
        st["tokens"].extend(this_tokens)
        st["duration"] += this_duration


def lr_at(step: int, lr: float, n_steps: int, warmup: float, decay: float) -> float:
    """LRScheduler.get_lr (train_utils.py:38-58): trapezoid."""
    t1, t2, t3 = int(n_steps * warmup), int(n_steps * (1 - decay)), n_steps
    if step < t1:
        return lr * step / t1
    if step < t2:
        return lr
    if step < t3:
        return lr * (t3 - step) / (t3 - t2)
    return lr


def train_steps(p: dict, trainable: Sequence[str], batches: Sequence[tuple], cfg: Cfg, *, lr: float = 1e-4,
                weight_decay: float = 0.0, lora_scale: float = 1.0, grad_accum: int = 1, n_steps: Optional[int] = None,
                warmup: float = 0.0, decay: float = 0.0, clip: Optional[float] = None) -> list[float]:
    """Loop body of train_metamathqa.py:217-257: accumulate (loss/accum).backward(); set LR; optional clip; AdamW step."""
    params = [p[k].requires_grad_(True) for k in trainable]
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)
    n_steps = n_steps if n_steps is not None else len(batches) // grad_accum
    losses = []
    it = iter(batches)
    for step in range(n_steps):
        for _ in range(grad_accum):
            tokens, labels, mask = next(it)
            loss = llama_forward(tokens, p, cfg, mask=mask, labels=labels, lora_scale=lora_scale)
            (loss / grad_accum).backward()
        for g in opt.param_groups:
            g["lr"] = lr_at(step, lr, n_steps, warmup, decay)
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(params, clip)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    return losses

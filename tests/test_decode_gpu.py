"""GPU: the decode path (csrc/decode.hip, llx/decode.py) - weight-streaming GEMV with its fused prologue / epilogues, split-cache
attention, cache scatter - against the oracle (oracle/ref.py restating modelling/llama.py:76-90,126-127,135-137,189-207), kernel by
kernel and as one Llama-3.1-8B-dimension layer + head decoding against a 4k-token cache."""
import math

import pytest
import torch

from oracle import ref as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def K(cuda):
    from llx import kernels

    return kernels


def _close(a, b, rel, name):
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= rel * scale + 1e-6, f"{name}: max err {err:.4e} vs scale {scale:.4e} (allowed {rel * scale:.4e})"


def _rms(x, w, eps):
    return O.rmsnorm(x, w, eps)


@pytest.mark.parametrize("M,K_,ns", [(1, 4096, (4096,)), (1, 512, (512, 128, 128)), (2, 1792, (512,)), (4, 4096, (256, 64, 64)), (3, 520, (36,)), (1, 14336, (4096,))])
@pytest.mark.parametrize("norm", [False, True])
def test_gemv_plain_and_residual(K, cuda, M, K_, ns, norm):
    """out = [rmsnorm(x) | x] @ [W0; W1; W2]^T (+ residual): fp32 reference on the bf16-rounded operands; K with a partly filled last
    512-element piece (1792, 520), N not a multiple of 4 (36 -> guarded tail rows... 36 is; 3 rows of tokens use the 4-token build)."""
    ws = [O.randn(f"gv_w{i}_{K_}_{n}", (n, K_), 0.05).to(BF) for i, n in enumerate(ns)]
    x = O.randn(f"gv_x{M}_{K_}", (M, K_), 1.0).to(BF)
    nw = (1 + O.randn(f"gv_n{K_}", (K_,), 0.1)).to(BF)
    res = O.randn(f"gv_r{M}_{sum(ns)}", (M, sum(ns)), 1.0).to(BF)
    xin = _rms(x, nw, 1e-5) if norm else x
    want = xin.float() @ torch.cat(ws).float().T
    got = K.gemv([w.to(cuda) for w in ws], x.to(cuda), norm=(nw.to(cuda), 1e-5) if norm else None)
    assert got.shape == want.shape
    _close(got.float().cpu(), want, 0.01, "gemv")
    got_r = K.gemv([w.to(cuda) for w in ws], x.to(cuda), norm=(nw.to(cuda), 1e-5) if norm else None, epilogue=K.GV_RESIDUAL, res=res.to(cuda))
    want_r = want.to(BF).float() + res.float()  # bf16 linear output + bf16 residual, rounded (the GEMM epilogue's order)
    _close(got_r.float().cpu(), want_r, 0.01, "gemv + residual")
    # bit-level: same rounding points as the MFMA GEMM path on a well-conditioned case is not required; determinism is
    assert torch.equal(got, K.gemv([w.to(cuda) for w in ws], x.to(cuda), norm=(nw.to(cuda), 1e-5) if norm else None))


def test_gemv_four_row_build_is_bit_identical(K, cuda, tmp_path):
    """LLX_GEMV_RPW=4 (four output rows per wave, steps of 1024 elements - the knob is read once per process, hence the child process):
    every lane adds the same products in the same order as in the default two-row build, so the outputs agree bit for bit - plain,
    q|k|v (RoPE + cache write) and SwiGLU epilogues."""
    import os, subprocess, sys

    script = r"""
import sys, torch
sys.path[:0] = [sys.argv[2] + "/llama-x_amd", sys.argv[2]]
from llx import kernels as K
torch.manual_seed(5)
dev = "cuda"
x = torch.randn(1, 1024, device=dev).bfloat16(); nw = (1 + 0.1 * torch.randn(1024, device=dev)).bfloat16()
w = [(0.05 * torch.randn(n, 1024, device=dev)).bfloat16() for n in (512, 128, 128)]
out = {"plain": K.gemv(w, x, norm=(nw, 1e-5))}
rope = torch.randn(1, 64, 2, device=dev)
kc = torch.zeros(1, 1, 16, 128, device=dev, dtype=torch.bfloat16); vc = torch.zeros_like(kc)
pos = torch.tensor([3], device=dev)
out["q"] = K.gemv(w, x, norm=(nw, 1e-5), epilogue=K.GV_QKV, qkv=(rope, 512, 128, kc, vc, pos)); out["kc"] = kc; out["vc"] = vc
g = [(0.05 * torch.randn(768, 1024, device=dev)).bfloat16() for _ in range(2)]
out["h"] = K.gemv(g, x, norm=(nw, 1e-5), epilogue=K.GV_SWIGLU)
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[1])
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for rpw in ("2", "4"):
        path = str(tmp_path / f"gemv_rpw{rpw}.pt")
        env = dict(os.environ, LLX_GEMV_RPW=rpw)
        subprocess.run([sys.executable, "-c", script, path, root], check=True, env=env, timeout=300)
        res[rpw] = torch.load(path, weights_only=True)
    for k in res["2"]:
        assert torch.equal(res["2"][k], res["4"][k]), k
    assert float(res["2"]["h"].abs().max()) > 0 and float(res["2"]["kc"].abs().max()) > 0


def test_gemv_ragged_rows_and_rejects(K, cuda):
    from llx._lib import LlxError

    w = O.randn("gv_w_ragged", (38, 256), 0.05).to(BF)  # N = 38: the last row group has 2 live rows
    x = O.randn("gv_x_ragged", (1, 256), 1.0).to(BF)
    got = K.gemv([w.to(cuda)], x.to(cuda))
    _close(got.float().cpu(), x.float() @ w.float().T, 0.01, "ragged N")
    with pytest.raises(LlxError):
        K.gemv([w.to(cuda)], O.randn("gv_x5", (5, 256)).to(BF).to(cuda))  # M > 4 belongs to the MFMA GEMM
    with pytest.raises(LlxError):
        K.gemv([w[:, :100].contiguous().to(cuda)], x[:, :100].contiguous().to(cuda))  # K % 8 != 0


@pytest.mark.parametrize("M", [1, 3])
def test_gemv_swiglu_and_lora(K, cuda, M):
    """gate|up with the SwiGLU epilogue (roundings of the bf16 eager graph: g, u, silu(g) rounded, then the product) and LoRA adapters
    on both members (t = rmsnorm(x) @ [A1; A3]^T from a first launch, B factors in the main launch; modelling/lora.py:40-44)."""
    D, I, r = 512, 1792, 16
    w1, w3 = (O.randn(f"sw_w{i}", (I, D), 0.05).to(BF) for i in (1, 3))
    a1, a3 = (O.randn(f"sw_a{i}", (r, D), 0.05).to(BF) for i in (1, 3))
    b1, b3 = (O.randn(f"sw_b{i}", (I, r), 0.05).to(BF) for i in (1, 3))
    x = O.randn(f"sw_x{M}", (M, D), 1.0).to(BF)
    nw = (1 + O.randn("sw_n", (D,), 0.1)).to(BF)
    xn = _rms(x, nw, 1e-5).float()
    for lora in (False, True):
        g = xn @ w1.float().T + (2.0 * (xn @ a1.float().T) @ b1.float().T if lora else 0)
        u = xn @ w3.float().T + (2.0 * (xn @ a3.float().T) @ b3.float().T if lora else 0)
        want = (torch.nn.functional.silu(g.to(BF).float()).to(BF).float() * u.to(BF).float())
        lo = None
        if lora:
            t = K.gemv([a1.to(cuda), a3.to(cuda)], x.to(cuda), norm=(nw.to(cuda), 1e-5))
            _close(t.float().cpu(), torch.cat([xn @ a1.float().T, xn @ a3.float().T], 1), 0.01, "t = x A^T")
            lo = ([b1.to(cuda), b3.to(cuda)], t, 2.0)
        h = K.gemv([w1.to(cuda), w3.to(cuda)], x.to(cuda), norm=(nw.to(cuda), 1e-5), epilogue=K.GV_SWIGLU, lora=lo)
        assert h.shape == (M, I)
        _close(h.float().cpu(), want, 0.02, f"swiglu lora={lora}")


@pytest.mark.parametrize("M", [1, 2, 4])
def test_gemv_qkv_rope_and_cache_scatter(K, cuda, M):
    """The q|k|v projection of a decode step: RoPE on q and k with the table rows of the CALL (0..M-1, modelling/llama.py:207), k / v
    written into the caches at input_pos (KVCache.update), everything else in the caches untouched."""
    D, H, KVH, hd, Smax = 512, 4, 2, 128, 96
    wq, wk, wv = O.randn("qk_wq", (H * hd, D), 0.05).to(BF), O.randn("qk_wk", (KVH * hd, D), 0.05).to(BF), O.randn("qk_wv", (KVH * hd, D), 0.05).to(BF)
    x = O.randn(f"qk_x{M}", (M, D), 1.0).to(BF)
    nw = (1 + O.randn("qk_n", (D,), 0.1)).to(BF)
    table = O.rope_table(O.TINY._replace(max_seq_len=Smax))
    pos = torch.tensor([70, 3, 95, 41][:M])
    xn = _rms(x, nw, 1e-5).float()
    q = (xn @ wq.float().T).to(BF).view(1, M, H, hd)
    k = (xn @ wk.float().T).to(BF).view(1, M, KVH, hd)
    v = (xn @ wv.float().T).to(BF).view(1, M, KVH, hd)
    q_want, k_want = O.rope_apply(q, table), O.rope_apply(k, table)
    kc = O.randn("qk_kc", (1, KVH, Smax, hd), 1.0).to(BF)
    vc = O.randn("qk_vc", (1, KVH, Smax, hd), 1.0).to(BF)
    kc_d, vc_d = kc.to(cuda), vc.to(cuda)
    got = K.gemv([wq.to(cuda), wk.to(cuda), wv.to(cuda)], x.to(cuda), norm=(nw.to(cuda), 1e-5), epilogue=K.GV_QKV,
                 qkv=(table.to(cuda), H * hd, KVH * hd, kc_d, vc_d, pos.to(cuda)))
    assert got.shape == (M, H * hd)
    _close(got.float().cpu().view(1, M, H, hd), q_want.float(), 0.01, "q with RoPE")
    kc_w, vc_w = kc.clone(), vc.clone()
    kc_w[:, :, pos] = k_want.transpose(1, 2)
    vc_w[:, :, pos] = v.transpose(1, 2)
    others = torch.ones(Smax, dtype=torch.bool)
    others[pos] = False
    assert torch.equal(kc_d.cpu()[:, :, others], kc[:, :, others]) and torch.equal(vc_d.cpu()[:, :, others], vc[:, :, others]), "untouched cache rows"
    _close(kc_d.cpu()[:, :, pos].float(), kc_w[:, :, pos].float(), 0.01, "k cache rows")
    _close(vc_d.cpu()[:, :, pos].float(), vc_w[:, :, pos].float(), 0.01, "v cache rows")
    # the stand-alone scatter kernel (prefill calls): strided [B, KVH, L, hd] views of a fused q|k|v buffer
    L_ = 5
    qkv = O.randn("sc_qkv", (1, L_, (H + 2 * KVH) * hd), 1.0).to(BF).to(cuda)
    k5 = qkv[..., H * hd : (H + KVH) * hd].unflatten(-1, (KVH, hd)).transpose(1, 2)
    v5 = qkv[..., (H + KVH) * hd :].unflatten(-1, (KVH, hd)).transpose(1, 2)
    p5 = torch.tensor([9, 0, 33, 95, 50], device=cuda)
    kc2, vc2 = kc.to(cuda), vc.to(cuda)
    K.kv_scatter(k5, v5, kc2, vc2, p5)
    kr, vr = kc.clone(), vc.clone()
    kr[:, :, p5.cpu()] = k5.cpu()
    vr[:, :, p5.cpu()] = v5.cpu()
    assert torch.equal(kc2.cpu(), kr) and torch.equal(vc2.cpu(), vr)


@pytest.mark.parametrize("H,KVH,M,Skv,valid", [(32, 8, 1, 4096, 4096), (32, 8, 1, 8192, 4101), (4, 1, 1, 512, 65), (4, 1, 4, 512, 300), (8, 2, 2, 200, 200),
                                               (4, 4, 3, 333, 17)])
def test_attn_decode_against_oracle_sdpa(K, cuda, H, KVH, M, Skv, valid):
    """Split-cache decode attention vs the oracle's SDPA over the WHOLE cache with the reference's mask rows (tril[input_pos]); the
    extent bound (keys any row may see) is computed on the device; caches hold garbage beyond the valid range."""
    hd = 128
    q = O.randn(f"ad_q{H}{M}", (1, H, M, hd), 1.0).to(BF)
    kc = O.randn(f"ad_k{KVH}{Skv}", (1, KVH, Skv, hd), 1.0).to(BF)
    vc = O.randn(f"ad_v{KVH}{Skv}", (1, KVH, Skv, hd), 1.0).to(BF)
    kc[:, :, valid:] = 1e4  # stale / never written cache rows must not leak through the mask
    pos = torch.arange(valid - M, valid)
    mask = torch.tril(torch.ones(Skv, Skv, dtype=torch.bool))[None, None, pos]
    want = O.sdpa(q.float(), kc.float(), vc.float(), mask)  # [1, H, M, hd]
    md = mask.to(cuda)
    ext = K.mask_extent(md)
    assert int(ext.item()) == valid
    for e in (ext, None):
        got = K.attn_decode(q.to(cuda), kc.to(cuda), vc.to(cuda), md, e)
        assert got.shape == (1, M, H * hd)
        _close(got.float().cpu().view(1, M, H, hd).transpose(1, 2), want, 0.02, f"decode attention (extent={'device' if e is not None else 'none'})")
    # an arbitrary (non-causal) mask with holes, per-head rows, and q as a strided view of a [1, M, H*hd] projection buffer
    g = torch.Generator().manual_seed(5)
    m2 = torch.rand(1, H, M, Skv, generator=g) < 0.3
    m2[..., 0] = True
    qb = q.transpose(1, 2).reshape(1, M, H * hd).contiguous().to(cuda)
    got = K.attn_decode(qb.view(1, M, H, hd).transpose(1, 2), kc.to(cuda).clamp(-4, 4), vc.to(cuda), m2.to(cuda), K.mask_extent(m2.to(cuda)))
    want = O.sdpa(q.float(), kc.float().clamp(-4, 4), vc.float(), m2)
    _close(got.float().cpu().view(1, M, H, hd).transpose(1, 2), want, 0.02, "decode attention, per-head mask with holes")


def test_attn_decode_fully_masked_row_is_nan(K, cuda):
    q = O.randn("ad_qn", (1, 4, 2, 128)).to(BF).to(cuda)
    kc = O.randn("ad_kn", (1, 1, 64, 128)).to(BF).to(cuda)
    mask = torch.zeros(1, 1, 2, 64, dtype=torch.bool)
    mask[0, 0, 1, :10] = True
    got = K.attn_decode(q, kc, kc, mask.to(cuda), K.mask_extent(mask.to(cuda))).view(1, 2, 4, 128)
    assert torch.isnan(got[0, 0]).all() and not torch.isnan(got[0, 1]).any(), "softmax over an all-masked row is NaN in SDPA; other rows unaffected"


@pytest.mark.parametrize("lora", [False, True])
def test_decode_layer_at_8b_dimensions(cuda, lora):
    """N2 at the real size: ONE Llama-3.1-8B-dimension layer + norm + head decoding 1 token (then 3 tokens in one call) against a
    cache holding 4100 positions, vs the oracle's cached forward (restating modelling/llama.py:83-90,126-127,135-137,189-207 incl.
    the RoPE-row quirk).  The product takes the weight-streaming path (llx/decode.py); the generic inference path must agree."""
    from modelling import Llama, apply_linear_adapter_
    from tests.util import bf16_params, to_model_config

    t0, Smax = 4100, 4352
    cfg = O.LLAMA31_8B._replace(num_layers=1, max_seq_len=Smax, vocab_size=8)
    p = O.init_params(cfg)
    if lora:
        p.update(O.init_lora(cfg, 16))
    pb, pf = bf16_params(p)
    model = Llama(to_model_config(cfg)).bfloat16()
    model.load_state_dict({k: v for k, v in pb.items() if "lora_" not in k})
    if lora:
        apply_linear_adapter_(model.layers, "lora", rank=16, alpha=32.0)
        with torch.no_grad():
            for name, mod in model.layers.named_modules():
                if f"layers.{name}.lora_a" in pb:
                    mod.lora_a.copy_(pb[f"layers.{name}.lora_a"])
                    mod.lora_b.copy_(pb[f"layers.{name}.lora_b"])
    model.build_cache(inference=True)
    model = model.to(cuda).eval()
    kc0 = O.randn("dl_kc", (1, cfg.num_kv_heads, Smax, 128), 1.0).to(BF)
    vc0 = O.randn("dl_vc", (1, cfg.num_kv_heads, Smax, 128), 1.0).to(BF)
    kc0[:, :, t0:], vc0[:, :, t0:] = 0, 0
    cache_mod = model.layers[0].attention.kv_cache
    cache_mod.k_cache.copy_(kc0)
    cache_mod.v_cache.copy_(vc0)
    cache = {0: (kc0.float().clone(), vc0.float().clone())}
    tokens = O.randint("dl_tok", (1, 4), 0, 8)
    scale = 2.0 if lora else 1.0
    import oracle.ref as R

    def oracle_step(tok, pos):
        # O.llama_forward_cached with the adapter scale threaded through (its linear() calls default to lora_scale 1)
        L_ = tok.shape[1]
        x = torch.nn.functional.embedding(tok, pf["tok_embeddings.weight"])
        table = R.rope_table(cfg)[:L_]
        mask = torch.tril(torch.ones(Smax, Smax, dtype=torch.bool))[None, None, pos]
        pre = "layers.0."
        h = R.rmsnorm(x, pf[pre + "attention_norm.weight"])
        q = R.linear(h, pf, pre + "attention.wq", scale).view(1, L_, cfg.num_heads, 128)
        k = R.linear(h, pf, pre + "attention.wk", scale).view(1, L_, cfg.num_kv_heads, 128)
        v = R.linear(h, pf, pre + "attention.wv", scale).view(1, L_, cfg.num_kv_heads, 128)
        q, k, v = R.rope_apply(q, table).transpose(1, 2), R.rope_apply(k, table).transpose(1, 2), v.transpose(1, 2)
        kc, vc = cache[0]
        kc[:, :, pos], vc[:, :, pos] = k, v
        o = R.sdpa(q, kc, vc, mask).transpose(1, 2).reshape(1, L_, -1)
        x = x + R.linear(o, pf, pre + "attention.wo", scale)
        x = x + R.feed_forward(R.rmsnorm(x, pf[pre + "ffn_norm.weight"]), pf, pre + "feed_forward.", scale)
        return torch.nn.functional.linear(R.rmsnorm(x, pf["norm.weight"]), pf["output.weight"])

    import llx.decode as D

    with torch.no_grad():
        for tok, pos in ((tokens[:, :1], torch.tensor([t0])), (tokens[:, 1:], torch.tensor([t0 + 1, t0 + 2, t0 + 3]))):
            assert D.layer_ok(model.layers[0], model.tok_embeddings(tok.to(cuda)), model.causal_mask[None, None, pos.to(cuda)])
            got = model(tok.to(cuda), input_pos=pos.to(cuda))
            want = oracle_step(tok, pos)
            _close(got.float().cpu(), want, 0.03, f"decode logits at {pos.tolist()}")
            kc_dev = cache_mod.k_cache[:, :, pos.to(cuda)].float().cpu()
            _close(kc_dev, cache[0][0][:, :, pos], 0.02, "k cache rows written by the fused projection")
    # the generic inference path (MFMA GEMMs + dense-mask attention) on the same state gives the same logits
    cache_mod.k_cache.copy_(kc0)
    cache_mod.v_cache.copy_(vc0)
    with torch.no_grad():
        fast = model(tokens[:, :1].to(cuda), input_pos=torch.tensor([t0], device=cuda))
        cache_mod.k_cache.copy_(kc0)
        cache_mod.v_cache.copy_(vc0)
        old = D.MAX_TOKENS
        D.MAX_TOKENS = 0
        try:
            slow = model(tokens[:, :1].to(cuda), input_pos=torch.tensor([t0], device=cuda))
        finally:
            D.MAX_TOKENS = old
    _close(fast.float().cpu(), slow.float().cpu(), 0.03, "weight-streaming path vs generic inference path")

"""CPU: the oracle (oracle/ref.py) against the committed golden vectors that oracle/gen_golden.py captured from the
REFERENCE itself (imported from /root/reference in the authoring container).  Inputs are regenerated here from the
numpy PCG64 generator; nothing of the reference is needed at test time."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import ref as O
from oracle import script_cases as SC

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = O.TINY


def G(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name + ".npz")).items()}


def tokens_labels(B, S, seed=0):
    tokens = O.randint("tokens", (B, S), 0, CFG.vocab_size, seed)
    labels = torch.roll(tokens, -1, 1).clone()
    labels[:, : S // 4] = -100
    labels[:, -1] = -100
    return tokens, labels


def test_rope_tables_bit_exact():
    g = G("g01_rope")
    assert torch.equal(O.rope_table(CFG)[:8], g["table_tiny"])
    freqs = 1.0 / (500_000 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
    scaled = O.llama31_rescale(freqs)
    assert torch.equal(scaled, g["scaled_freqs"])
    # 29 unchanged / 6 smoothed / 29 divided by 8 at base 5e5
    assert int((scaled == freqs).sum()) == 29 and int((scaled == freqs / 8).sum()) == 29
    digest = hashlib.sha256(O.rope_table(O.LLAMA31_8B).numpy().tobytes()).digest()
    assert bytes(g["sha256_8b_table"].numpy().tolist()) == digest, "full Llama-3.1-8B RoPE table (checksum of 2 MiB)"


def test_apply_rope_bit_exact():
    g = G("g02_apply_rope")
    x = O.randn("rope_x", (2, 256, 5, 128))
    t = O.rope_table(CFG)
    assert torch.equal(O.rope_apply(x, t)[:, ::16, :, ::8], g["y_f32_slice"])
    assert torch.equal(O.rope_apply(x.bfloat16(), t)[:, ::16, :, ::8].float(), g["y_bf16_slice"])


def test_rmsnorm():
    g = G("g03_rmsnorm")
    w = 1 + O.randn("norm_w", (512,), 0.1)
    x = O.randn("norm_x", (300, 512))
    dy = O.randn("norm_dy", (300, 512))
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    y = O.rmsnorm(xr, wr)
    y.backward(dy)
    torch.testing.assert_close(y[::10, ::4], g["y"], atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(xr.grad[::10, ::4], g["dx"], atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(wr.grad, g["dw"], atol=1e-4, rtol=1e-5)
    assert torch.equal(O.rmsnorm(x.bfloat16(), w.bfloat16())[::10, ::4].float(), g["y_bf16"]), "single rounding in bf16"


def test_llama_fp32_logits_loss_grads():
    g = G("g07_llama_fp32")
    params = O.init_params(CFG)
    tokens, labels = tokens_labels(2, 256)
    torch.testing.assert_close(O.llama_forward(tokens, params, CFG)[:, ::8, ::8], g["logits_slice"], atol=2e-5, rtol=1e-4)
    pr = {k: v.clone().requires_grad_() for k, v in params.items()}
    loss = O.llama_forward(tokens, pr, CFG, labels=labels)
    loss.backward()
    torch.testing.assert_close(loss.detach(), g["loss"], atol=1e-6, rtol=1e-6)
    for name in ("layers.0.attention.wq.weight", "layers.1.feed_forward.w2.weight", "layers.0.attention_norm.weight", "norm.weight",
                 "layers.1.attention.wk.weight"):
        got = pr[name].grad
        got = got if got.dim() == 1 else got[::8, ::8]
        torch.testing.assert_close(got, g[name.replace(".", "_")], atol=1e-6, rtol=2e-4)


def test_modules_and_masks():
    g = G("g04_modules")
    params = O.init_params(CFG)
    t = O.rope_table(CFG)
    h = O.randn("hidden", (2, 256, 512), 0.5)
    torch.testing.assert_close(O.attention(h, params, "layers.0.attention.", CFG, t[:256], None)[:, ::8, ::8], g["attn"], atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(O.feed_forward(h, params, "layers.0.feed_forward.")[:, ::8, ::8], g["mlp"], atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(O.layer(h, params, 0, CFG, t[:256], None)[:, ::8, ::8], g["layer"], atol=2e-5, rtol=1e-4)
    gm = G("g04_masks")
    S = 384
    doc = gm["doc_ids"]
    tokens1, labels1 = tokens_labels(1, S)
    loss = O.llama_forward(tokens1, params, CFG, mask=O.document_mask(doc)[None, None], labels=labels1)
    torch.testing.assert_close(loss, gm["loss_doc"], atol=2e-6, rtol=2e-6)  # reference ran this through FlexAttention
    dense = O.prefix_lm_mask(S, torch.tensor([128]))
    hid = O.randn("hidden1", (1, S, 512), 0.5)
    torch.testing.assert_close(O.layer(hid, params, 0, CFG, t[:S], dense)[:, ::8, ::8], gm["prefix_layer"], atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(O.llama_forward(tokens1, params, CFG, mask=dense)[:, ::8, ::8], gm["prefix_logits"], atol=2e-5, rtol=1e-4)


def test_llama_bf16_statistical():
    g = G("g07_llama_bf16")
    pb = {k: v.bfloat16() for k, v in O.init_params(CFG).items()}
    tokens, labels = tokens_labels(2, 256)
    out = O.llama_forward(tokens, pb, CFG)[:, ::8, ::8].float()
    assert (out - g["logits_slice"]).abs().max() < 0.05  # bf16 eager on both sides, different accumulation order
    assert abs(O.llama_forward(tokens, pb, CFG, labels=labels).item() - g["loss"].item()) < 0.02


@pytest.mark.parametrize("rank", [8, 16])
def test_lora(rank):
    g = G(f"g08_lora_r{rank}")
    params = dict(O.init_params(CFG))
    params.update({k: v.clone().requires_grad_() for k, v in O.init_lora(CFG, rank).items()})
    tokens, labels = tokens_labels(2, 256)
    loss = O.llama_forward(tokens, params, CFG, labels=labels, lora_scale=1.0)
    loss.backward()
    torch.testing.assert_close(loss.detach(), g["loss"], atol=1e-6, rtol=1e-6)
    for name in ("layers.0.attention.wq.lora_a", "layers.0.attention.wq.lora_b", "layers.1.feed_forward.w2.lora_a",
                 "layers.1.feed_forward.w2.lora_b", "layers.0.attention.wv.lora_b"):
        torch.testing.assert_close(params[name].grad, g[name.replace(".", "_")], atol=1e-7, rtol=3e-4)


def test_dora():
    g = G("g08_dora")
    w, b = O.randn("dora_w", (256, 512), 0.05), O.randn("dora_b", (256,), 0.05)
    a, lb, x = O.randn("dora_a", (8, 512), 0.05), O.randn("dora_lb", (256, 8), 0.05), O.randn("dora_x", (40, 512))
    torch.testing.assert_close(g["m"], w.norm(p=2, dim=1), atol=1e-6, rtol=1e-6)  # init: m = ||W||_row (modelling/lora.py:51)
    torch.testing.assert_close(O.dora_linear(x, w, a, lb, g["m"], 2.0, b), g["y"], atol=1e-5, rtol=1e-5)


def test_dora_model():
    """apply_linear_adapter_(model.layers, "dora") through the whole model: loss and gradients of m / lora_a / lora_b."""
    g = G("g08_dora_model")
    params = dict(O.init_params(CFG))
    params.update({k: v.clone().requires_grad_() for k, v in O.init_lora(CFG, 8).items()})
    params.update({k: v.clone().requires_grad_() for k, v in O.init_dora_m(params, CFG).items()})
    tokens, labels = tokens_labels(2, 256)
    loss = O.llama_forward(tokens, params, CFG, labels=labels, lora_scale=2.0)
    loss.backward()
    torch.testing.assert_close(loss.detach(), g["loss"], atol=1e-6, rtol=1e-6)
    for name in ("layers.0.attention.wq.m", "layers.0.attention.wq.lora_a", "layers.1.feed_forward.w2.m", "layers.1.feed_forward.w1.lora_b",
                 "layers.0.attention.wv.m"):
        torch.testing.assert_close(params[name].grad, g[name.replace(".", "_")], atol=1e-7, rtol=3e-4)


def test_int8_bit_exact():
    g = G("g09_quant_bf16")
    w8 = O.randn("q_w", (96, 512), 0.05).bfloat16()
    w8[5] = 0
    q, s = O.quantize_int8_rowwise(w8)
    assert torch.equal(q, g["q"].to(torch.int8)) and torch.equal(s.float(), g["scale"])
    assert s.dtype is torch.bfloat16 and int(q[5].abs().sum()) == 0
    gl = G("g09_int8_linear")
    wq, ws = O.quantize_int8_rowwise(O.randn("i8_w", (256, 512), 0.05).bfloat16())
    x = O.randn("i8_x", (40, 512)).bfloat16().requires_grad_()
    y = O.int8_linear(x, wq, ws, dynamic=False)
    y.backward(O.randn("i8_g", (40, 256)).bfloat16())
    assert torch.equal(y.float(), gl["y"]) and torch.equal(x.grad.float(), gl["dx"])
    assert torch.equal(O.int8_dequantize(wq, ws)[::8, ::8].float(), gl["dequant_slice"])


def test_int8_mm_dequant_against_the_executed_reference_kernel():
    """A23: g10 holds what the reference's Triton kernel (subclasses/int8_mm.py:50-118) computed under TRITON_INTERPRET=1
    (oracle/gen_golden_scripts.py): fp32 output with fp32 scales, and the bf16 result for bf16 scales; B is the non-contiguous
    W.T view of subclasses/int8.py:113; M / N not multiples of the block sizes."""
    gm = G("g10_int8_mm")
    for name, (M, N, K, _blocks) in SC.INT8_MM_CASES.items():
        a8, w8, sa, sb = SC.int8_mm_inputs(name, M, N, K)
        c32 = O.int8_mm_dequant(a8, w8.T, sa, sb)
        assert c32.dtype is torch.float32 and torch.equal(c32, gm[f"{name}_c_f32"])
        cb = O.int8_mm_dequant(a8, w8.T, sa.bfloat16(), sb.bfloat16())
        assert cb.dtype is torch.bfloat16 and torch.equal(cb.float(), gm[f"{name}_c_bf16"])
        assert torch.equal(O.int8_mm_dequant(a8, w8.T.contiguous(), sa, sb), c32)


# ------------------------------------------------------------------------------------------------- G12: the training scripts' iterators
def test_padding_iterator_matches_reference_batches():
    g = G("g12_padding")
    torch.manual_seed(SC.PAD_SEED)
    for i, (inputs, labels) in enumerate(SC.oracle_padding_batches(SC.documents())):
        assert inputs.dtype is torch.int64 and torch.equal(inputs, g[f"inputs_{i}"]) and torch.equal(labels, g[f"labels_{i}"])


def test_document_packer_matches_reference_buffers_and_mask_bits():
    g = G("g12_document_mask")
    torch.manual_seed(SC.PACK_SEED)
    for i, (inputs, labels, doc_ids) in enumerate(SC.oracle_packed_buffers(SC.documents())):
        assert torch.equal(inputs.view(1, -1), g[f"inputs_{i}"]) and torch.equal(labels.view(1, -1), g[f"labels_{i}"])
        assert torch.equal(doc_ids, g[f"doc_ids_{i}"])
        bits = torch.from_numpy(np.packbits(O.document_mask(doc_ids).numpy(), axis=1))
        assert torch.equal(bits, g[f"mask_bits_{i}"]), "dense mask of the reference's mask_mod closure (train_metamathqa.py:67-68)"


def test_librispeech_batches_match_reference():
    g = np.load(os.path.join(GOLD, "g12_librispeech.npz"))
    tok = SC.ToyTokenizer()
    listing = O.list_transcripts(SC.TRANSCRIPTS, tok)
    assert [p for p, _ in listing] == g["listing_paths"].tolist() and [len(t) for _, t in listing] == g["listing_tokens"].tolist()
    a, t, lab = O.prepare_audio_batch(SC.prepare_batch_case(), int(SC.AUDIO_SECONDS * SC.AUDIO_RATE), SC.AUDIO_MULTIPLE, tok.pad_id)
    assert np.array_equal(a.numpy(), g["prep_audio"]) and np.array_equal(t.numpy(), g["prep_tokens"]) and np.array_equal(lab.numpy(), g["prep_labels"])
    torch.manual_seed(SC.AUDIO_SEED)
    for i, batch in enumerate(SC.oracle_utterance_batches(listing, SC.clips())):
        for nm, v in zip(("audio", "tokens", "labels"), batch):
            assert np.array_equal(v.numpy(), g[f"{nm}_{i}"]), (i, nm)


def test_lr_schedule_matches_reference():
    g = G("g12_lr_schedule")
    for i, (lr, n, wu, dc) in enumerate(SC.LR_CASES):
        assert [O.lr_at(s, lr, n, wu, dc) for s in range(n + 3)] == g[f"case_{i}"].tolist()


def test_audio_path_given_mel():
    g = G("g11_audio")
    pa = O.init_params(CFG, audio=True)
    audio = O.uniform("audio", (1, 16000), -0.1, 0.1)
    ttok, tlab = tokens_labels(1, 128)
    mel = O.mel_spectrogram(audio)
    assert mel.shape == (1, 128, 101)  # frames = 1 + L // hop
    torch.testing.assert_close(mel, g["mel"], atol=1e-6, rtol=1e-4)
    # downstream of the mel tensor everything was checked against the reference's own LlamaAudio.forward
    feat = O.log_mel_cmn(g["mel"])
    torch.testing.assert_close(feat, g["feat"], atol=1e-5, rtol=1e-5)
    assert feat.shape[-1] == 100 and abs(feat.mean(2)).max() < 1e-5
    torch.testing.assert_close(O.audio_embed(feat, pa)[:, ::2, ::8], g["audio_tokens"], atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(O.llama_audio_forward(None, ttok, pa, CFG, mel=g["mel"])[:, ::4, ::8], g["logits_slice"], atol=3e-5, rtol=1e-4)
    torch.testing.assert_close(O.llama_audio_forward(None, ttok, pa, CFG, mel=g["mel"], labels=tlab), g["loss"], atol=2e-6, rtol=2e-6)


def test_three_step_trajectory():
    g = G("g13_trajectory")
    params = dict(O.init_params(CFG))
    lp = O.init_lora(CFG, 8)
    params.update({k: v.clone() for k, v in lp.items()})
    batches = [(*tokens_labels(1, 256, seed=s), None) for s in range(3)]
    losses = O.train_steps(params, sorted(lp), batches, CFG, lr=1e-3)
    np.testing.assert_allclose(losses, g["losses"].numpy(), atol=2e-5)


def test_kv_cache_path():
    g = G("g14_kv_cache")
    params = O.init_params(CFG)
    tok, _ = tokens_labels(1, 96)
    cache = O.new_cache(CFG)
    pre = O.llama_forward_cached(tok[:, :64], params, CFG, cache, torch.arange(64))
    torch.testing.assert_close(pre[:, ::8, ::8], g["prefill_slice"], atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(pre, O.llama_forward(tok[:, :64], params, CFG), atol=2e-5, rtol=1e-4)  # prefill == causal forward
    for i, t in enumerate(range(64, 68)):
        d = O.llama_forward_cached(tok[:, t : t + 1], params, CFG, cache, torch.tensor([t]))
        torch.testing.assert_close(d[:, 0, ::8], g["decode_slices"][i], atol=2e-5, rtol=1e-4)

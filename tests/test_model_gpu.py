"""Model-level parity on the GPU: the product (modelling/ + subclasses/ on HIP kernels, bf16) against the CPU oracle
(fp32 math on the same bf16-rounded weights).  Tolerances: bf16 activations through 2 layers -> logits within 2e-2
absolute (values are O(0.3)); losses within 2e-3; LoRA gradients within 3% of the gradient's max magnitude."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref as O  # noqa: E402
from tests.util import bf16_params, build_model  # noqa: E402

CFG = O.TINY


def _data(B, S, seed=0):
    tokens = O.randint("tokens", (B, S), 0, CFG.vocab_size, seed)
    labels = torch.roll(tokens, -1, 1).clone()
    labels[:, : S // 4] = -100
    labels[:, -1] = -100
    return tokens, labels


def _close(a, b, rel, name):
    scale = b.abs().max().item()
    err = (a - b).abs().max().item()
    assert err <= rel * scale + 1e-6, f"{name}: max err {err:.4e} vs scale {scale:.4e} (allowed {rel * scale:.4e})"


def test_logits_causal(cuda):
    pb, pf = bf16_params(O.init_params(CFG))
    tokens, _ = _data(2, 256)
    ref = O.llama_forward(tokens, pf, CFG)
    model = build_model(CFG, pb, cuda)
    with torch.no_grad():
        out = model(tokens.to(cuda))
    assert out.dtype is torch.bfloat16 and out.shape == ref.shape
    _close(out.float().cpu(), ref, 0.03, "logits")


@pytest.mark.parametrize("S", [256, 384])
def test_loss_and_full_grads(cuda, S):
    """Dense (no adapter): every parameter trainable -> exercises wgrad, norm dw, embedding scatter, LM-head grads."""
    pb, pf = bf16_params(O.init_params(CFG))
    tokens, labels = _data(2, S)
    pr = {k: v.clone().requires_grad_() for k, v in pf.items()}
    ref = O.llama_forward(tokens, pr, CFG, labels=labels)
    ref.backward()
    model = build_model(CFG, pb, cuda)
    loss = model(tokens.to(cuda), labels=labels.to(cuda))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 2e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    for name, prm in model.named_parameters():
        assert prm.grad is not None, name
        _close(prm.grad.float().cpu(), pr[name].grad, 0.04, name)


@pytest.mark.parametrize("rank", [8, 16])
def test_lora_loss_and_grads(cuda, rank):
    p = O.init_params(CFG)
    p.update(O.init_lora(CFG, rank))
    pb, pf = bf16_params(p)
    tokens, labels = _data(2, 256)
    train = [k for k in pf if "lora_" in k or k.endswith("_norm.weight")]
    pr = {k: (v.clone().requires_grad_() if k in train else v) for k, v in pf.items()}
    ref = O.llama_forward(tokens, pr, CFG, labels=labels, lora_scale=1.0)
    ref.backward()
    model = build_model(CFG, pb, cuda, lora_rank=rank)
    for n, prm in model.named_parameters():
        if n.startswith(("tok_embeddings", "output", "norm")):
            prm.requires_grad_(False)
    loss = model(tokens.to(cuda), labels=labels.to(cuda))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 2e-3 * max(1.0, abs(ref.item()))
    for name, prm in model.named_parameters():
        if prm.requires_grad:
            assert prm.grad is not None, name
            _close(prm.grad.float().cpu(), pr[name].grad, 0.04, name)
        else:
            assert prm.grad is None, name


def test_document_and_prefix_masks(cuda):
    from modelling.llama import MaskSpec

    pb, pf = bf16_params(O.init_params(CFG))
    S = 384
    tokens, labels = _data(1, S)
    doc = torch.zeros(S, dtype=torch.int64)
    for c in (70, 150, 301):
        doc[c:] += 1
    doc[S - 20 :] = 0  # packer tail quirk
    model = build_model(CFG, pb, cuda)
    # document mask (train_metamathqa.py:67-68)
    ref = O.llama_forward(tokens, pf, CFG, mask=O.document_mask(doc)[None, None], labels=labels)
    loss = model(tokens.to(cuda), labels=labels.to(cuda), block_mask=MaskSpec(doc_ids=doc))
    assert abs(loss.item() - ref.item()) < 2e-3 * max(1.0, abs(ref.item()))
    # prefix-LM (README.md:16): pinned through the oracle's dense mask path
    P = torch.tensor([128])
    ref = O.llama_forward(tokens, pf, CFG, mask=O.prefix_lm_mask(S, P))
    with torch.no_grad():
        out = model(tokens.to(cuda), block_mask=MaskSpec(prefix_len=P))
    _close(out.float().cpu(), ref, 0.03, "prefix-LM logits")


@pytest.mark.parametrize("dynamic", [False, True])
def test_int8_lora(cuda, dynamic):
    p = O.init_params(CFG)
    p.update(O.init_lora(CFG, 16))
    pb, pf = bf16_params(p)
    # oracle side: quantise the bf16 weights exactly as Int8LinearWeight.from_float does, keep scales in bf16
    po = dict(pf)
    for i in range(CFG.num_layers):
        for suf in O.LINEAR_SUFFIXES:
            key = f"layers.{i}.{suf}"
            q, s = O.quantize_int8_rowwise(pb[key + ".weight"])
            po.pop(key + ".weight")
            po[key + ".int_data"], po[key + ".scale"], po[key + ".dynamic"] = q, s.float(), dynamic
    tokens, labels = _data(2, 256)
    train = [k for k in po if "lora_" in k]
    pr = {k: (v.clone().requires_grad_() if k in train else v) for k, v in po.items()}
    ref = O.llama_forward(tokens, pr, CFG, labels=labels)
    ref.backward()
    model = build_model(CFG, pb, cuda, lora_rank=16, quantize="int8", quantize_kwargs=dict(dynamic_int8_act=dynamic))
    for n, prm in model.named_parameters():
        if "lora_" not in n:
            prm.requires_grad_(False)
    w = model.layers[0].attention.wq.weight
    qref, sref = O.quantize_int8_rowwise(pb["layers.0.attention.wq.weight"])
    assert torch.equal(w.int_data.cpu(), qref) and torch.equal(w.scale.cpu(), sref), "int8 quantiser must be bit-exact"
    loss = model(tokens.to(cuda), labels=labels.to(cuda))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 5e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    for name, prm in model.named_parameters():
        if prm.requires_grad:
            _close(prm.grad.float().cpu(), pr[name].grad, 0.06, name)


def test_three_step_trajectory(cuda):
    """G13: 3 AdamW steps on LoRA factors follow the oracle's loss trajectory (train_metamathqa.py:217-257)."""
    p = O.init_params(CFG)
    p.update(O.init_lora(CFG, 8))
    pb, pf = bf16_params(p)
    batches = [(*_data(1, 256, seed=s), None) for s in range(3)]
    train = sorted(k for k in pf if "lora_" in k)
    ref_losses = O.train_steps({k: v.clone() for k, v in pf.items()}, train, [(t, l, m) for t, l, m in batches], CFG, lr=1e-3)
    model = build_model(CFG, pb, cuda, lora_rank=8)
    for n, prm in model.named_parameters():
        if "lora_" not in n:
            prm.requires_grad_(False)
    opt = torch.optim.AdamW([q for q in model.parameters() if q.requires_grad], lr=1e-3, weight_decay=0.0)
    losses = []
    for tokens, labels, _ in batches:
        loss = model(tokens.to(cuda), labels=labels.to(cuda))
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 5e-3 * max(1.0, abs(b)), (losses, ref_losses)


# ------------------------------------------------------------------------------------------------- audio (config C3 shape)
def test_mel_spectrogram_kernel(cuda):
    """K15: device mel vs the oracle's torch.stft restatement (itself cross-checked to 3.7e-6 against transformers.audio_utils;
    torchaudio absent => parity unpinned against the reference proper).  fp32 direct DFT vs FFT: 1e-4 of the spectrum's peak."""
    from llx.audio_ops import MelSpectrogram, logmel_cmn_padded

    audio = O.uniform("audio", (2, 16000), -0.1, 0.1)
    audio[1, 8000:] = 0.0  # silence exercises the 1e-12 clip
    ref = O.mel_spectrogram(audio)
    ms = MelSpectrogram().to(cuda)
    mel = ms(audio.to(cuda))
    assert mel.shape == ref.shape == (2, 128, 101)
    torch.testing.assert_close(mel.cpu(), ref, atol=1e-4 * ref.abs().max().item(), rtol=1e-3)
    feat = logmel_cmn_padded(mel).cpu().float()
    rf = O.log_mel_cmn(ref).transpose(1, 2)  # [B, T, n_mels]
    assert feat.shape == (2, 102, 128) and feat[:, 0].abs().sum() == 0 and feat[:, -1].abs().sum() == 0
    # bf16 storage of values up to ~|12|: half an ulp = 0.03; compare where the mel energy is above the fp32 noise floor
    strong = (ref[..., :-1] > 1e-9 * ref.max()).transpose(1, 2)
    assert ((feat[:, 1:-1] - rf).abs()[strong]).max() < 0.07


def test_audio_model_loss_and_conv_grads(cuda):
    p = O.init_params(CFG, audio=True)
    pb, pf = bf16_params(p)
    audio = O.uniform("audio", (1, 32000), -0.1, 0.1)  # 2 s -> 201 frames -> 200 -> 100 audio tokens
    tokens, labels = _data(1, 128)
    train = [k for k in pf if k.startswith("audio_embed")]
    pr = {k: (v.clone().requires_grad_() if k in train else v) for k, v in pf.items()}
    mel = O.mel_spectrogram(audio)
    ref_logits = O.llama_audio_forward(None, tokens, pf, CFG, mel=mel)
    ref = O.llama_audio_forward(None, tokens, pr, CFG, mel=mel, labels=labels)
    ref.backward()
    model = build_model(CFG, pb, cuda, audio=True)
    for n, q in model.named_parameters():
        q.requires_grad_(n.startswith("audio_embed"))
    with torch.no_grad():
        logits = model(audio.to(cuda), tokens.to(cuda))
    assert logits.shape == ref_logits.shape == (1, 128, CFG.vocab_size)
    _close(logits.float().cpu(), ref_logits, 0.04, "audio logits")
    loss = model(audio.to(cuda), tokens.to(cuda), labels=labels.to(cuda))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 3e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    for name, q in model.named_parameters():
        if q.requires_grad:
            _close(q.grad.float().cpu(), pr[name].grad, 0.06, name)
    # text-only call of the audio model takes the plain embedding path (audio=None, modelling/audio.py:51)
    with torch.no_grad():
        t_only = model(None, tokens.to(cuda))
    _close(t_only.float().cpu(), O.llama_forward(tokens, pf, CFG), 0.03, "audio model, audio=None")


def test_audio_prefix_lm_mask(cuda):
    """P1 end to end: prefix-LM over [audio ; text] with P = number of audio tokens."""
    from modelling.llama import MaskSpec

    p = O.init_params(CFG, audio=True)
    pb, pf = bf16_params(p)
    audio = O.uniform("audio", (1, 32000), -0.1, 0.1)
    tokens, labels = _data(1, 156)  # 100 audio + 156 text = 256
    mel = O.mel_spectrogram(audio)
    mask = O.prefix_lm_mask(256, [100])
    ref = O.llama_audio_forward(None, tokens, pf, CFG, mel=mel, labels=labels, mask=mask)
    model = build_model(CFG, pb, cuda, audio=True)
    with torch.no_grad():
        loss = model(audio.to(cuda), tokens.to(cuda), labels=labels.to(cuda), block_mask=MaskSpec(prefix_len=torch.tensor([100])))
    assert abs(loss.item() - ref.item()) < 3e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())


# ------------------------------------------------------------------------------------------------- M3 / N1 / A18
def test_trainer_accumulation_clip_schedule_and_checkpoint(cuda, tmp_path):
    """M3 step semantics (accumulate, LR before step, clip) against the oracle loop, then N1: save {step, model, optim},
    reload into a fresh int8+LoRA model with weights_only=True and continue identically."""
    from llx.data import LRScheduler
    from llx.train import Trainer

    p = O.init_params(CFG)
    p.update(O.init_lora(CFG, 8))
    pb, pf = bf16_params(p)
    batches = [_data(1, 256, seed=s) for s in range(4)]
    train = sorted(k for k in pf if "lora_" in k)
    ref_losses = O.train_steps({k: v.clone() for k, v in pf.items()}, train, [(t, l, None) for t, l in batches], CFG, lr=1e-3, grad_accum=2,
                               n_steps=2, warmup=0.5, decay=0.0, clip=1.0)
    model = build_model(CFG, pb, cuda, lora_rank=8)
    for n, q in model.named_parameters():
        q.requires_grad_("lora_" in n)
    opt = torch.optim.AdamW([q for q in model.parameters() if q.requires_grad], lr=1e-3, weight_decay=0.0)
    tr = Trainer(model, opt, lr_schedule=LRScheduler(1e-3, 2, 0.5, 0.0), grad_accum=2, clip_grad_norm=1.0)
    losses = []
    for s in range(2):
        mb = [(lambda m, t=t, l=l: m(t.to(cuda), labels=l.to(cuda))) for t, l in batches[2 * s : 2 * s + 2]]
        losses.append(tr.step(mb).item())
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 5e-3 * max(1.0, abs(b)), (losses, ref_losses)
    assert opt.param_groups[0]["lr"] == 1e-3  # step 1 of 2 with warmup 0.5 -> full lr (step 0 ran at lr 0)

    # --- checkpoint round trip with int8 weights (Int8LinearWeight flatten/unflatten + copy_)
    model8 = build_model(CFG, pb, cuda, lora_rank=8, quantize="int8")
    opt8 = torch.optim.AdamW([q for n, q in model8.named_parameters() if "lora_" in n], lr=1e-3)
    for n, q in model8.named_parameters():
        q.requires_grad_("lora_" in n)
    t8 = Trainer(model8, opt8)
    t0, l0 = batches[0]
    t8.step(lambda m: m(t0.to(cuda), labels=l0.to(cuda)))
    path = tmp_path / "last.pth"
    torch.save(t8.state_dict(), path)
    from subclasses import Int8LinearWeight

    torch.serialization.add_safe_globals([Int8LinearWeight])
    ckpt = torch.load(path, map_location="cpu", weights_only=True, mmap=True)
    fresh = build_model(CFG, {k: torch.zeros_like(v) for k, v in pb.items()}, cuda, lora_rank=8, quantize="int8")
    for n, q in fresh.named_parameters():
        q.requires_grad_("lora_" in n)
    optf = torch.optim.AdamW([q for n, q in fresh.named_parameters() if "lora_" in n], lr=1e-3)
    tf = Trainer(fresh, optf)
    tf.load_state_dict(ckpt)
    assert tf.step_idx == 1
    assert torch.equal(fresh.layers[0].attention.wq.weight.int_data, model8.layers[0].attention.wq.weight.int_data)
    t1, l1 = batches[1]
    a = t8.step(lambda m: m(t1.to(cuda), labels=l1.to(cuda)))
    b = tf.step(lambda m: m(t1.to(cuda), labels=l1.to(cuda)))
    assert torch.equal(a, b), "resumed run must continue bit-identically (deterministic kernels)"
    assert torch.equal(fresh.layers[1].feed_forward.w2.lora_b, model8.layers[1].feed_forward.w2.lora_b)


def test_dora_linear_standalone(cuda):
    """A18: DoRALinear.forward (second priority) = fused LoRA GEMM + row-norm rescale; compared with the oracle / golden."""
    import numpy as np
    import os

    from modelling import apply_linear_adapter_

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g08_dora.npz"))
    lin = torch.nn.Linear(512, 256, bias=True)
    lin.weight.data.copy_(O.randn("dora_w", (256, 512), 0.05))
    lin.bias.data.copy_(O.randn("dora_b", (256,), 0.05))
    lin = lin.bfloat16()
    apply_linear_adapter_(lin, "dora", rank=8, alpha=16.0)
    lin.lora_a.data.copy_(O.randn("dora_a", (8, 512), 0.05))
    lin.lora_b.data.copy_(O.randn("dora_lb", (256, 8), 0.05))
    lin = lin.to(cuda)
    x = O.randn("dora_x", (40, 512)).bfloat16()
    y = lin(x.to(cuda))
    ref = torch.from_numpy(g["y"])
    _close(y.float().cpu(), ref, 0.03, "DoRA forward vs reference golden")
    y.sum().backward()
    assert lin.m.grad is not None and lin.lora_a.grad is not None and lin.weight.grad is None


def test_kv_cache_prefill_and_decode(cuda):
    """N2 / A5 / M5: build_cache(inference=True); prefill with input_pos then single-token decode steps follow the oracle's
    restatement of the reference's cached path (KVCache.update + causal_mask[None, None, input_pos], modelling/llama.py:83-90,
    189-194,205-207 - including its quirk that RoPE rows restart at 0 on every call; pinned by fixture g14_kv_cache)."""
    pb, pf = bf16_params(O.init_params(CFG))
    tokens, _ = _data(1, 96)
    from tests.util import to_model_config
    from modelling import Llama

    model = Llama(to_model_config(CFG)).bfloat16()
    model.load_state_dict(pb, strict=False)
    model.build_cache(inference=True)
    model = model.to(cuda).eval()
    assert model.causal_mask.shape == (CFG.max_seq_len, CFG.max_seq_len) and model.layers[0].attention.kv_cache.k_cache.shape == (1, 1, CFG.max_seq_len, 128)
    cache = O.new_cache(CFG)
    with torch.no_grad():
        pre = model(tokens[:, :64].to(cuda), input_pos=torch.arange(64, device=cuda))
        ref = O.llama_forward_cached(tokens[:, :64], pf, CFG, cache, torch.arange(64))
        _close(pre.float().cpu(), ref, 0.03, "prefill logits")
        _close(pre.float().cpu(), O.llama_forward(tokens[:, :64], pf, CFG), 0.03, "prefill == causal forward")
        for t in range(64, 70):
            step = model(tokens[:, t : t + 1].to(cuda), input_pos=torch.tensor([t], device=cuda))
            want = O.llama_forward_cached(tokens[:, t : t + 1], pf, CFG, cache, torch.tensor([t]))
            _close(step.float().cpu(), want, 0.03, f"decode logits at {t}")
    # dense bool mask= on a layer (the reference's small-shape prefix-LM route), forward only
    layer = model.layers[0]
    layer.attention.kv_cache = None
    hid = O.randn("hidden1", (1, 384, 512), 0.5).bfloat16()
    dense = O.prefix_lm_mask(384, [128])
    with torch.no_grad():
        out = layer(hid.to(cuda), model.rope[:384], mask=dense.to(cuda))
    want = O.layer(hid.float(), pf, 0, CFG, O.rope_table(CFG)[:384], dense)
    _close(out.float().cpu(), want, 0.03, "layer with dense prefix-LM mask")
    from llx._lib import LlxError

    with pytest.raises(LlxError):
        layer(hid.to(cuda).requires_grad_(), model.rope[:384], mask=dense.to(cuda))


@pytest.mark.parametrize("S", [512, 2048])
def test_full_dimension_layer_parity(cuda, S):
    """One TransformerLayer at the REAL Llama-3.1-8B dimensions (D 4096, 32/8 heads, I 14336, LoRA r=16) against the oracle:
    exercises the production tile shapes (N = 6144 / 28672 fused groups with RoPE / SwiGLU epilogues, K-extension, block-diagonal
    LoRA operands, GQA 4:1, multi-tile causal attention; S = 2048 adds multi-round GEMM grids and 16-tile key sweeps)."""
    from modelling import apply_linear_adapter_
    from modelling.llama import LlamaConfig, TransformerLayer, build_rope

    cfg = O.LLAMA31_8B._replace(num_layers=1, max_seq_len=S)
    p = {k: v for k, v in O.init_params(cfg._replace(vocab_size=8)).items() if k.startswith("layers.0.")}
    p.update(O.init_lora(cfg, 16))
    pb, pf = bf16_params(p)
    x = O.randn("x_full", (1, S, cfg.embed_dim), 0.5).bfloat16()
    dy = O.randn("dy_full", (1, S, cfg.embed_dim), 0.1).bfloat16()
    train = [k for k in pf if "lora_" in k or k.endswith("_norm.weight")]
    pr = {k: (v.clone().requires_grad_() if k in train else v) for k, v in pf.items()}
    xr = x.float().requires_grad_()
    ref = O.layer(xr, pr, 0, cfg, O.rope_table(cfg)[:S], None, 1.0)
    ref.backward(dy.float())

    layer = TransformerLayer(LlamaConfig(**{f: getattr(cfg, f) for f in LlamaConfig._fields})).bfloat16()
    layer.load_state_dict({k[len("layers.0."):]: v for k, v in pb.items() if "lora_" not in k})
    apply_linear_adapter_(layer, "lora", rank=16, alpha=16.0)
    with torch.no_grad():
        for name, mod in layer.named_modules():
            if f"layers.0.{name}.lora_a" in pb:
                mod.lora_a.copy_(pb[f"layers.0.{name}.lora_a"])
                mod.lora_b.copy_(pb[f"layers.0.{name}.lora_b"])
    layer = layer.to(cuda)
    for n, q in layer.named_parameters():
        q.requires_grad_("lora_" in n or n.endswith("_norm.weight"))
    rope = build_rope(LlamaConfig(**{f: getattr(cfg, f) for f in LlamaConfig._fields})).to(cuda)
    xg = x.to(cuda).requires_grad_()
    out = layer(xg, rope[:S])
    out.backward(dy.to(cuda))
    _close(out.float().cpu(), ref.detach(), 0.02, "layer output at 8B dims")
    _close(xg.grad.float().cpu(), xr.grad, 0.04, "dx at 8B dims")
    for name, q in layer.named_parameters():
        if q.requires_grad:
            _close(q.grad.float().cpu(), pr["layers.0." + name].grad, 0.05, name)
    # determinism at full size: a second run is bit-identical (no atomics anywhere on the path)
    xg2 = x.to(cuda).requires_grad_()
    for q in layer.parameters():
        q.grad = None
    out2 = layer(xg2, rope[:S])
    out2.backward(dy.to(cuda))
    assert torch.equal(out2, out) and torch.equal(xg2.grad, xg.grad)


def test_packed_iterator_with_prefetch_trains(cuda):
    """N3: the document-mask packer (llx.data, bit-exact with the oracle's restatement of train_metamathqa.py:51-83) feeding the
    model through the pinned-memory prefetcher; losses equal the oracle's on the same packed buffers."""
    from llx import data as D

    pb, pf = bf16_params(O.init_params(CFG))
    docs = [O.randint(f"doc{i}", (n,), 1, CFG.vocab_size) for i, n in enumerate((90, 130, 64, 200, 17, 150, 99, 260, 40))]
    gen = torch.Generator().manual_seed(0)
    it = D.document_mask_iterator(list(docs), 384, generator=gen)
    host_batches = [next(it) for _ in range(2)]
    model = build_model(CFG, pb, cuda)
    pre = D.DevicePrefetcher(iter(host_batches), cuda)
    for (inputs, labels, ms), (hi, hl, hm) in zip(pre, host_batches):
        assert inputs.is_cuda and torch.equal(inputs.cpu(), hi)
        with torch.no_grad():
            loss = model(inputs, labels=labels, block_mask=ms)
        ref = O.llama_forward(hi, pf, CFG, mask=O.document_mask(hm.doc_ids.cpu().long().view(-1))[None, None], labels=hl)
        assert abs(loss.item() - ref.item()) < 3e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())

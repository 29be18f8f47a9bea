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


def _rows_close(a, b, name, min_cos=0.999, floor=1e-3):
    """Per-row check next to the max-norm one: every row (last dim) of `a` must point the same way as the reference row and have
    the same length - an error confined to small-magnitude rows (a dropped LoRA row block, a mis-rotated head) passes `_close` but
    not this.  Rows whose reference norm is below `floor` x the largest row norm carry rounding noise only and are skipped."""
    a2, b2 = a.reshape(-1, a.shape[-1]).double(), b.reshape(-1, b.shape[-1]).double()
    nb = b2.norm(dim=1)
    keep = nb > floor * nb.max()
    cos = (a2 * b2).sum(1) / (a2.norm(dim=1) * nb).clamp_min(1e-30)
    worst = cos[keep].min().item()
    assert worst >= min_cos, f"{name}: worst per-row cosine {worst:.5f} < {min_cos} (row {int(cos.masked_fill(~keep, 2).argmin())})"
    ratio = (a2.norm(dim=1) / nb.clamp_min(1e-30))[keep]
    assert (ratio - 1).abs().max().item() < 0.05, f"{name}: per-row norm ratio off by {(ratio - 1).abs().max().item():.4f}"


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
            if prm.grad.dim() == 2:
                _rows_close(prm.grad.float().cpu(), pr[name].grad, name, min_cos=0.995)
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
    """A18: DoRALinear.forward stand-alone (with a bias) = fused LoRA GEMM + HIP row-norm rescale; forward against the REFERENCE's
    golden output, gradients of x / m / lora_a / lora_b / bias against the oracle (modelling/lora.py:53-62)."""
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
    torch.testing.assert_close(lin.m.detach().float(), torch.from_numpy(g["m"]), rtol=2 ** -7, atol=0)
    lin = lin.to(cuda)
    lin.bias.requires_grad_(True)
    x = O.randn("dora_x", (40, 512)).bfloat16()
    xg = x.to(cuda).requires_grad_()
    y = lin(xg)
    ref = torch.from_numpy(g["y"])
    _close(y.float().cpu(), ref, 0.03, "DoRA forward vs reference golden")
    dy = O.randn("dora_dy", (40, 256)).bfloat16()
    y.backward(dy.to(cuda))
    t = {k: getattr(lin, k).detach().float().cpu().requires_grad_() for k in ("lora_a", "lora_b", "m", "bias")}
    xo = x.float().requires_grad_()
    yo = O.dora_linear(xo, lin.weight.detach().float().cpu(), t["lora_a"], t["lora_b"], t["m"], 2.0, t["bias"])
    yo.backward(dy.float())
    _close(y.float().cpu(), yo.detach(), 0.02, "DoRA forward vs oracle")
    _close(xg.grad.float().cpu(), xo.grad, 0.03, "DoRA dx")
    for k, v in t.items():
        _close(getattr(lin, k).grad.float().cpu(), v.grad, 0.03, "DoRA d" + k)
    assert lin.weight.grad is None


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
    # ... and under autograd (the reference trains through this route too, modelling/llama.py:135-137,163-172): a dense mask that the
    # MaskSpec rule reproduces exactly runs on the fused kernels with their backward; an arbitrary one has no backward and raises
    from llx._lib import LlxError

    xg = hid.to(cuda).requires_grad_()
    dy = O.randn("dense_dy", (1, 384, 512), 0.1).bfloat16()
    out_g = layer(xg, model.rope[:384], mask=dense.to(cuda))
    out_g.backward(dy.to(cuda))
    xr = hid.float().requires_grad_()
    pr = {k: (v.clone().requires_grad_() if k.startswith("layers.0.") and k.endswith("_norm.weight") else v) for k, v in pf.items()}
    ref = O.layer(xr, pr, 0, CFG, O.rope_table(CFG)[:384], dense)
    ref.backward(dy.float())
    _close(out_g.float().cpu(), ref.detach(), 0.03, "dense prefix-LM mask under autograd: output")
    _close(xg.grad.float().cpu(), xr.grad, 0.05, "dense prefix-LM mask under autograd: dx")
    _close(layer.attention_norm.weight.grad.float().cpu(), pr["layers.0.attention_norm.weight"].grad, 0.06, "d attention_norm.weight")
    g = torch.Generator().manual_seed(3)
    scattered = torch.rand(384, 384, generator=g) < 0.5
    scattered |= torch.eye(384, dtype=torch.bool)
    with pytest.raises(LlxError):
        layer(hid.to(cuda).requires_grad_(), model.rope[:384], mask=scattered.to(cuda))


def _mask_for(kind, S):
    """(dense bool oracle mask | None, MaskSpec | None) for the layer-parity cases."""
    from modelling.llama import MaskSpec

    if kind == "causal":
        return None, None
    if kind == "doc":  # packed documents of uneven length + the packer's id-0 tail (train_metamathqa.py:51-83)
        doc = torch.zeros(S, dtype=torch.int64)
        for c in (S // 16 + 5, S // 3 + 77, S // 2 - 130, (7 * S) // 8 + 9):
            doc[c:] += 1
        doc[S - 100 :] = 0
        return O.document_mask(doc), MaskSpec(doc_ids=doc)
    if kind == "prefix":  # prefix-LM: bidirectional over the first P positions (P not a multiple of the 64/128 tiles)
        P = S // 2 - 56
        return O.prefix_lm_mask(S, [P])[0, 0], MaskSpec(prefix_len=torch.tensor([P]))
    raise ValueError(kind)


@pytest.mark.parametrize("S,kind,base", [(512, "causal", "bf16"), (2048, "causal", "bf16"), (4096, "causal", "bf16"), (4096, "doc", "bf16"),
                                         (4096, "prefix", "bf16"), (8192, "causal", "bf16"), (8192, "prefix", "bf16"), (8192, "mixed-prefix-b2", "bf16"),
                                         (4096, "causal", "int8-dynamic"), (4096, "causal", "int8-weight-only")])
def test_full_dimension_layer_parity(cuda, S, kind, base):
    """One TransformerLayer at the REAL Llama-3.1-8B dimensions (D 4096, 32/8 heads, I 14336, LoRA r=16) against the oracle, at
    the sequence lengths BASELINE.json's configs run (S = 4096 headline, S = 8192 for configs[4]) and with the mask kinds they use
    (causal, packed-document, prefix-LM, and configs[4]'s B = 2 batch with a different prefix length per sample): exercises the
    production tile shapes (N = 6144 / 28672 fused groups with RoPE / SwiGLU epilogues, K-extension, block-diagonal LoRA operands,
    GQA 4:1, 64-128-tile causal key sweeps, multi-round GEMM grids).  base = int8-*: configs[3]'s frozen INT8 base
    (quantize_linear_ then the adapter, train_metamathqa.py:178-179) - dynamic: fused norm-quantiser, llx_int8_mm_dequant_ext with
    RoPE / SwiGLU / residual epilogues and the scaled adapter gradient pass; weight-only: the bf16 GEMM on the widened int8 image
    with the column-scale epilogue - against O.int8_linear per member linear."""
    from modelling import apply_linear_adapter_
    from modelling.llama import LlamaConfig, MaskSpec, TransformerLayer, build_rope
    from subclasses import quantize_linear_

    cfg = O.LLAMA31_8B._replace(num_layers=1, max_seq_len=S)
    p = {k: v for k, v in O.init_params(cfg._replace(vocab_size=8)).items() if k.startswith("layers.0.")}
    p.update(O.init_lora(cfg, 16))
    pb, pf = bf16_params(p)
    B = 2 if kind == "mixed-prefix-b2" else 1
    x = O.randn("x_full", (B, S, cfg.embed_dim), 0.5).bfloat16()
    dy = O.randn("dy_full", (B, S, cfg.embed_dim), 0.1).bfloat16()
    if base != "bf16":  # oracle side: quantise the bf16 weights exactly as Int8LinearWeight.from_float does (scales in bf16)
        for suf in O.LINEAR_SUFFIXES:
            key = f"layers.0.{suf}"
            q, sc = O.quantize_int8_rowwise(pb[key + ".weight"])
            pf.pop(key + ".weight")
            pf[key + ".int_data"], pf[key + ".scale"], pf[key + ".dynamic"] = q, sc.float(), base == "int8-dynamic"
    train = [k for k in pf if "lora_" in k or k.endswith("_norm.weight")]
    pr = {k: (v.clone().requires_grad_() if k in train else v) for k, v in pf.items()}
    xr = x.float().requires_grad_()
    if kind == "mixed-prefix-b2":  # configs[4]: per-sample prefix lengths {2048, 4096} in one batch
        P = torch.tensor([2048, 4096])
        spec = MaskSpec(prefix_len=P)
        outs = []
        for b in range(B):  # the oracle sample by sample (its [H, S, S] fp32 scores are 8.6 GB each); parameter gradients add up
            ob = O.layer(xr[b : b + 1], pr, 0, cfg, O.rope_table(cfg)[:S], O.prefix_lm_mask(S, P[b : b + 1])[0, 0], 1.0)
            ob.backward(dy[b : b + 1].float())
            outs.append(ob.detach())
        ref = torch.cat(outs)
    else:
        dense, spec = _mask_for(kind, S)
        ref = O.layer(xr, pr, 0, cfg, O.rope_table(cfg)[:S], dense, 1.0)
        ref.backward(dy.float())
        ref = ref.detach()

    layer = TransformerLayer(LlamaConfig(**{f: getattr(cfg, f) for f in LlamaConfig._fields})).bfloat16()
    layer.load_state_dict({k[len("layers.0."):]: v for k, v in pb.items() if "lora_" not in k})
    if base != "bf16":
        quantize_linear_(layer, "int8", dynamic_int8_act=base == "int8-dynamic")
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
    out = layer(xg, rope[:S], block_mask=spec)
    out.backward(dy.to(cuda))
    # Dynamic int8 activations: the oracle runs in fp32, the product rounds every activation to bf16 before the row-wise quantiser, so
    # a few per cent of the int8 codes differ by one step between the two - a difference of the size of the quantisation noise itself
    # (the heavy-tailed silu(g)*u rows carry ~3 % of it).  The max-norm bars widen accordingly; the per-row cosine bars stay tight
    # enough to catch any structural error (a wrong scale, a dropped row block, a mis-rotated head).
    dyn = base == "int8-dynamic"
    t_out, t_dx, t_g = (0.06, 0.08, 0.08) if dyn else (0.02, 0.04, 0.05)
    c_out, c_dx, c_g = (0.998, 0.995, 0.99) if dyn else (0.999, 0.998, 0.995)
    o_cpu, dx_cpu = out.float().cpu(), xg.grad.float().cpu()
    print(f"[{S}-{kind}-{base}] out err {(o_cpu - ref).abs().max() / ref.abs().max():.4f}, dx err {(dx_cpu - xr.grad).abs().max() / xr.grad.abs().max():.4f}")
    _close(o_cpu, ref, t_out, "layer output at 8B dims")
    _rows_close(o_cpu, ref, "layer output rows", min_cos=c_out)
    _close(dx_cpu, xr.grad, t_dx, "dx at 8B dims")
    _rows_close(dx_cpu, xr.grad, "dx rows", min_cos=c_dx)
    for name, q in layer.named_parameters():
        if q.requires_grad:
            _close(q.grad.float().cpu(), pr["layers.0." + name].grad, t_g, name)
            if q.grad.dim() == 2:
                _rows_close(q.grad.float().cpu(), pr["layers.0." + name].grad, name, min_cos=c_g)
    # determinism at full size: a second run is bit-identical (no atomics anywhere on the path)
    xg2 = x.to(cuda).requires_grad_()
    for q in layer.parameters():
        q.grad = None
    out2 = layer(xg2, rope[:S], block_mask=spec)
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


# ------------------------------------------------------------------------------------------------- configs[4]: mixed prefix, B > 1
def test_audio_text_mixed_prefix_batch(cuda):
    """BASELINE.json configs[4] in small: B = 2 early-fusion [audio ; text] sequences whose prefix lengths DIFFER per sample
    (sample 0: all 100 audio tokens bidirectional, sample 1: a 1.2 s clip zero-padded to 2 s -> P = 60), text zero-padded to a
    common length with -100 labels (LibriSpeech._prepare_batch, train_librispeech.py:68-86), LoRA r=8 on the layers and a trainable
    audio_embed: loss, conv gradients and LoRA gradients against the oracle's dense per-sample prefix-LM mask."""
    from llx import data as D
    from modelling.llama import MaskSpec

    p = O.init_params(CFG, audio=True)
    p.update(O.init_lora(CFG, 8))
    pb, pf = bf16_params(p)
    clips = [O.uniform("audio0", (32000,), -0.1, 0.1), O.uniform("audio1", (19200,), -0.1, 0.1)]
    toks = [[1] + O.randint("t0", (140,), 3, CFG.vocab_size).tolist() + [2], [1] + O.randint("t1", (87,), 3, CFG.vocab_size).tolist() + [2]]
    audio, tokens, labels = D.prepare_audio_batch(list(zip(clips, toks)), 32000, 156, 0)  # text padded to 156 -> S = 100 + 156 = 256
    assert tokens.shape == (2, 156) and (labels[1, 88:] == -100).all() and labels[0, 140] == 2
    P = torch.tensor([100, 60])
    S = 100 + tokens.shape[1]
    mask = O.prefix_lm_mask(S, P)
    assert mask.shape == (2, 1, S, S) and bool(mask[0, 0, 3, 99]) and not bool(mask[1, 0, 3, 99]) and bool(mask[1, 0, 3, 59])
    train = [k for k in pf if "lora_" in k or k.startswith("audio_embed")]
    pr = {k: (v.clone().requires_grad_() if k in train else v) for k, v in pf.items()}
    mel = O.mel_spectrogram(audio)
    ref = O.llama_audio_forward(None, tokens, pr, CFG, mel=mel, labels=labels, mask=mask, lora_scale=1.0)
    ref.backward()
    model = build_model(CFG, pb, cuda, lora_rank=8, audio=True)
    for n, q in model.named_parameters():
        q.requires_grad_("lora_" in n or n.startswith("audio_embed"))
    loss = model(audio.to(cuda), tokens.to(cuda), labels=labels.to(cuda), block_mask=MaskSpec(prefix_len=P))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 3e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    for name, q in model.named_parameters():
        if q.requires_grad:
            assert q.grad is not None, name
            _close(q.grad.float().cpu(), pr[name].grad, 0.06, name)
    # the prefix is per sample: changing sample 1's prefix moves sample 1's logits and leaves sample 0's bit-identical
    with torch.no_grad():
        a = model(audio.to(cuda), tokens.to(cuda), block_mask=MaskSpec(prefix_len=P))
        b = model(audio.to(cuda), tokens.to(cuda), block_mask=MaskSpec(prefix_len=torch.tensor([100, 100])))
    assert torch.equal(a[0], b[0]) and not torch.equal(a[1], b[1])


def test_audio_conv_stack_full_width(cuda):
    """A12 / K16 at the production width: Conv1d(128 -> 4096, k3, s1) + GELU + Conv1d(4096 -> 4096, k3, s2) + GELU over a 20 s clip
    as implicit GEMMs (M = 2000, K = 384, N = 4096 and M = 1000, K = 12288, N = 4096 over the stride-2 overlapping view; M is not a
    multiple of the 256-row tile), forward and the hand-written backward (conv weights + biases), against the oracle's F.conv1d."""
    from llx import audio_ops
    from modelling import LlamaAudio
    from tests.util import to_model_config

    D = 4096
    cfg = O.TINY._replace(embed_dim=D, num_layers=0, num_heads=32, num_kv_heads=8, intermediate_dim=256, vocab_size=64, max_seq_len=2048)
    gen = {"audio_embed.0.weight": O.randn("c1w", (D, 128, 3), 0.05), "audio_embed.0.bias": O.randn("c1b", (D,), 0.05),
           "audio_embed.2.weight": O.randn("c2w", (D, D, 3), 0.01), "audio_embed.2.bias": O.randn("c2b", (D,), 0.05),
           "tok_embeddings.weight": O.randn("emb", (64, D), 0.02)}
    pb, pf = bf16_params(gen)
    audio = O.uniform("audio", (1, 320000), -0.1, 0.1)
    tokens = O.randint("tokens", (1, 24), 0, 64)
    mel = O.mel_spectrogram(audio)
    pr = {k: v.clone().requires_grad_() for k, v in pf.items()}
    feat = O.log_mel_cmn(mel).bfloat16().float()  # the product casts the features to the embedding dtype (modelling/audio.py:55)
    ref = O.audio_embed(feat, pr)
    assert ref.shape == (1, 1000, D)
    dy = O.randn("conv_dy", (1, 1000, D), 0.1).bfloat16()
    ref.backward(dy.float())
    model = LlamaAudio(to_model_config(cfg)).bfloat16()
    model.load_state_dict(pb, strict=False)
    model.build_cache()
    model = model.to(cuda)
    x, n_audio = audio_ops.audio_prefix_and_embed(model, audio.to(cuda), tokens.to(cuda))
    assert n_audio == 1000 and x.shape == (1, 1024, D)
    _close(x[:, :1000].float().cpu(), ref.detach(), 0.02, "conv stack output at D=4096")
    _rows_close(x[:, :1000].float().cpu(), ref.detach(), "conv stack rows", min_cos=0.999)
    assert torch.equal(x[0, 1000:].cpu(), pb["tok_embeddings.weight"][tokens[0]])  # the gather lands behind the audio prefix
    g = torch.zeros_like(x)
    g[:, :1000] = dy.to(cuda)
    x.backward(g)
    for name in ("audio_embed.0.weight", "audio_embed.0.bias", "audio_embed.2.weight", "audio_embed.2.bias"):
        got = dict(model.named_parameters())[name].grad.float().cpu()
        _close(got, pr[name].grad, 0.03, name)
        if got.dim() == 3:
            _rows_close(got.flatten(1), pr[name].grad.flatten(1), name, min_cos=0.998, floor=0.02)


# ------------------------------------------------------------------------------------------------- K18 activation checkpointing
def test_activation_checkpointing_bit_identical_and_saves_memory(cuda):
    """LlamaConfig.activation_checkpointing=True (modelling/llama.py:209-212: non-reentrant checkpoint per layer): every activation
    of the fused blocks goes through save_for_backward, so the checkpoint drops them and recomputes - same loss and gradients bit for
    bit (deterministic kernels), lower peak memory."""
    cfg = CFG._replace(num_layers=6, max_seq_len=2048)
    p = O.init_params(cfg)
    p.update(O.init_lora(cfg, 8))
    pb, _ = bf16_params(p)
    tokens, labels = _data(2, 2048)
    res = {}
    for ckpt in (False, True):
        model = build_model(cfg._replace(activation_checkpointing=ckpt), pb, cuda, lora_rank=8)
        assert model.config.activation_checkpointing == ckpt
        for n, q in model.named_parameters():
            q.requires_grad_("lora_" in n or n.endswith("_norm.weight") or n.startswith("tok_embeddings"))
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = model(tokens.to(cuda), labels=labels.to(cuda))
        loss.backward()
        torch.cuda.synchronize()
        res[ckpt] = (loss.detach().clone(), {n: q.grad.clone() for n, q in model.named_parameters() if q.grad is not None},
                     torch.cuda.max_memory_allocated() - base)
        del model, loss
    (l0, g0, m0), (l1, g1, m1) = res[False], res[True]
    assert torch.equal(l0, l1), (l0, l1)
    assert g0.keys() == g1.keys() and len(g0) > 0
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    assert m1 < 0.75 * m0, f"checkpointing must lower peak activation memory: {m1 / 2**20:.0f} MiB vs {m0 / 2**20:.0f} MiB"


def test_audio_embed_is_checkpointed_with_the_layers(cuda):
    """LlamaAudio with activation_checkpointing=True also checkpoints the conv stack (modelling/audio.py:56-57): z1 / z2 / h1 of
    AudioPrefixFn are dropped after the forward and recomputed in backward - bit-identical loss and conv / embedding gradients,
    lower peak memory (a 20 s clip: 2000 frames at D = 512)."""
    cfg = CFG._replace(num_layers=1, max_seq_len=1280)
    pb, _ = bf16_params(O.init_params(cfg, audio=True))
    audio = O.uniform("ck_audio", (2, 320000), -0.1, 0.1)
    tokens, labels = _data(2, 128)
    res = {}
    for ckpt in (False, True):
        model = build_model(cfg._replace(activation_checkpointing=ckpt), pb, cuda, audio=True)
        for n, q in model.named_parameters():
            q.requires_grad_(n.startswith("audio_embed") or n.startswith("tok_embeddings"))
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        loss = model(audio.to(cuda), tokens.to(cuda), labels=labels.to(cuda))
        after_fwd = torch.cuda.memory_allocated() - base
        loss.backward()
        torch.cuda.synchronize()
        res[ckpt] = (loss.detach().clone(), {n: q.grad.clone() for n, q in model.named_parameters() if q.grad is not None}, after_fwd)
        del model, loss
    (l0, g0, m0), (l1, g1, m1) = res[False], res[True]
    assert torch.equal(l0, l1) and g0.keys() == g1.keys() and any(n.startswith("audio_embed") for n in g0)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
    assert m1 < 0.8 * m0, f"the conv stack's saved activations must be gone after the forward: {m1 / 2**20:.1f} MiB vs {m0 / 2**20:.1f} MiB"


# ------------------------------------------------------------------------------------------------- A18 DoRA inside the fused blocks
def test_dora_model_loss_and_grads(cuda):
    """apply_linear_adapter_(model.layers, "dora") (the scripts' --adapter dora): every linear of the fused attention / MLP blocks
    is a DoRALinear.  Loss against the reference's golden value (g08_dora_model, fp32) and loss + gradients of m / lora_a / lora_b
    against the oracle on the same bf16-rounded parameters."""
    import numpy as np
    import os

    from modelling import DoRALinear, apply_linear_adapter_

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g08_dora_model.npz"))
    p = O.init_params(CFG)
    p.update(O.init_lora(CFG, 8))
    p.update(O.init_dora_m(p, CFG))
    pb, pf = bf16_params(p)
    tokens, labels = _data(2, 256)
    train = [k for k in pf if "lora_" in k or k.endswith(".m")]
    pr = {k: (v.clone().requires_grad_() if k in train else v) for k, v in pf.items()}
    ref = O.llama_forward(tokens, pr, CFG, labels=labels, lora_scale=2.0)
    ref.backward()
    assert abs(ref.item() - float(g["loss"])) < 5e-3, "bf16-rounded parameters move the fp32 golden loss only slightly"
    model = build_model(CFG, {k: v for k, v in pb.items() if "lora_" not in k and not k.endswith(".m")}, torch.device("cpu"))
    apply_linear_adapter_(model.layers, "dora", rank=8, alpha=16.0)
    with torch.no_grad():
        for name, mod in model.layers.named_modules():
            if isinstance(mod, DoRALinear):
                torch.testing.assert_close(mod.m.float(), pf[f"layers.{name}.weight"].norm(dim=1), rtol=2 ** -7, atol=0)  # init = ||W||_row
                mod.lora_a.copy_(pb[f"layers.{name}.lora_a"])
                mod.lora_b.copy_(pb[f"layers.{name}.lora_b"])
                mod.m.copy_(pb[f"layers.{name}.m"])
    model = model.to(cuda)
    for n, q in model.named_parameters():
        q.requires_grad_("lora_" in n or n.endswith(".m"))
    loss = model(tokens.to(cuda), labels=labels.to(cuda))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 3e-3 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    for name, q in model.named_parameters():
        if q.requires_grad:
            assert q.grad is not None, name
            _close(q.grad.float().cpu(), pr[name].grad, 0.05, name)
    for key in ("layers_0_attention_wq_m", "layers_1_feed_forward_w2_m"):  # and the reference's own gradient of m, to bf16 tolerance
        name = key.replace("layers_0_", "layers.0.").replace("layers_1_", "layers.1.").replace("attention_wq_m", "attention.wq.m").replace("feed_forward_w2_m", "feed_forward.w2.m")
        _close(dict(model.named_parameters())[name].grad.float().cpu(), torch.from_numpy(g[key]), 0.08, name + " vs reference golden")


@pytest.mark.parametrize("pattern", ["prompt", "scattered"])
def test_head_compaction_matches_uncompacted_path(cuda, pattern, monkeypatch):
    """HeadLossFn over the labelled rows only (llx/ops.py, default) against the same model with LLX_HEAD_COMPACT=0: the loss sums the
    same row terms (different order: 2e-6 relative), every gradient is bit-identical - skipping the ignore_index rows changes nothing
    the reference computes (modelling/llama.py:216-218: their loss term and gradient row are zero)."""
    from llx import ops

    p = O.init_params(CFG)
    p.update(O.init_lora(CFG, 8))
    pb, _ = bf16_params(p)
    tokens, labels = _data(2, 256)
    if pattern == "scattered":
        labels = torch.roll(tokens, -1, 1).clone()
        labels[torch.rand(labels.shape, generator=torch.Generator().manual_seed(3)) < 0.5] = -100
    out = {}
    for compact in (True, False):
        monkeypatch.setattr(ops, "_HEAD_COMPACT", compact)
        model = build_model(CFG, pb, cuda, lora_rank=8)
        for n, prm in model.named_parameters():
            prm.requires_grad_("lora_" in n or n.endswith("norm.weight"))
        loss = model(tokens.to(cuda), labels=labels.to(cuda))
        loss.backward()
        out[compact] = (loss.item(), {n: prm.grad.clone() for n, prm in model.named_parameters() if prm.requires_grad})
    assert abs(out[True][0] - out[False][0]) <= 2e-6 * abs(out[False][0])
    assert out[True][1].keys() == out[False][1].keys() and len(out[True][1]) > 10
    for n in out[True][1]:
        assert torch.equal(out[True][1][n], out[False][1][n]), n
    # a trainable head (the reference's default, train_metamathqa.py:177-180): its weight gradient runs over the compacted rows as well
    # (llx_gemm_tn_bf16_rows reads the count on the device) - same terms as the all-rows product in another fp32 summation order
    got = {}
    for compact in (True, False):
        monkeypatch.setattr(ops, "_HEAD_COMPACT", compact)
        model = build_model(CFG, pb, cuda, lora_rank=8)
        loss = model(tokens.to(cuda), labels=labels.to(cuda))
        loss.backward()
        got[compact] = (loss.item(), model.output.weight.grad.clone(), model.norm.weight.grad.clone(), model.tok_embeddings.weight.grad.clone())
    assert abs(got[True][0] - out[False][0]) <= 2e-6 * abs(out[False][0])
    gw, gw0 = got[True][1].float(), got[False][1].float()
    torch.testing.assert_close(gw, gw0, atol=2 ** -7 * gw0.abs().max().item(), rtol=2 ** -6)
    assert torch.nn.functional.cosine_similarity(gw.flatten(), gw0.flatten(), dim=0).item() > 0.99999
    for i in (2, 3):  # d hidden: the compacted path cuts the vocabulary contraction in K ranges, the all-rows trainable path does not
        g1, g0 = got[True][i].float(), got[False][i].float()
        torch.testing.assert_close(g1, g0, atol=2 ** -6 * g0.abs().max().item(), rtol=2 ** -5)


# ------------------------------------------------------------------------------------------------- T-chunked LM head
@pytest.mark.parametrize("compact", [True, False])
def test_chunked_head_is_bit_identical_to_one_buffer(cuda, compact, monkeypatch):
    """HeadLossFn walks the rows in chunks when [T, V] logits do not fit one buffer (llx/ops.py: 4 GiB of bf16 = 16.7 k rows of the
    128 k vocabulary).  Forced here at T = 4096 with 1024-row chunks (one chunk ends inside the labelled rows, the last ones hold
    none on the compacted path): same per-row arithmetic, global count of labelled rows -> loss and every gradient bit-identical."""
    from llx import ops

    p = O.init_params(CFG)
    p.update(O.init_lora(CFG, 8))
    pb, _ = bf16_params(p)
    cfg = CFG._replace(max_seq_len=2048)
    tokens, labels = _data(2, 2048)
    labels[1, 1500:] = -100
    out = {}
    for chunked in (False, True):
        monkeypatch.setattr(ops, "_HEAD_COMPACT", compact)
        monkeypatch.setattr(ops, "_HEAD_CHUNK_FORCED", chunked)
        monkeypatch.setattr(ops, "_HEAD_CHUNK_ROWS", 1024)
        model = build_model(cfg, pb, cuda, lora_rank=8)
        for n, prm in model.named_parameters():
            prm.requires_grad_("lora_" in n or n.endswith("norm.weight"))
        loss = model(tokens.to(cuda), labels=labels.to(cuda))
        loss.backward()
        out[chunked] = (loss.detach().clone(), {n: prm.grad.clone() for n, prm in model.named_parameters() if prm.requires_grad})
    assert torch.equal(out[True][0], out[False][0]), (out[True][0], out[False][0])
    for n in out[True][1]:
        assert torch.equal(out[True][1][n], out[False][1][n]), n
    # a trainable head beyond one buffer is refused loudly instead of overflowing the tile offsets
    from llx._lib import LlxError

    monkeypatch.setattr(ops, "_HEAD_CHUNK_FORCED", True)
    model = build_model(cfg, pb, cuda, lora_rank=8)
    with pytest.raises(LlxError):
        model(tokens.to(cuda), labels=labels.to(cuda))


def test_chunked_head_at_20480_rows_of_the_full_vocabulary(cuda):
    """T = 20 480 positions x V = 128 256 (5.25 GB of logits: more than the 4 GiB one GEMM operand may span; the reference's packed
    [1, bs * S] batch at bs = 5, S = 4096): the chunked head runs, and equals the one-buffer head evaluated on four 5120-row slices
    recombined with their labelled-row counts (loss: weighted mean; d hidden: slice gradient x n_slice / n_total)."""
    from modelling import Llama, LlamaConfig

    T, D, V = 20480, 4096, 128_256
    cfg = LlamaConfig(embed_dim=D, num_layers=1, head_dim=128, num_heads=32, num_kv_heads=8, intermediate_dim=256, max_seq_len=64, vocab_size=V)
    with torch.device("meta"):
        model = Llama(cfg)
    model = model.to(torch.bfloat16).to_empty(device=cuda)
    g = torch.Generator(device=cuda).manual_seed(1)
    with torch.no_grad():
        model.output.weight.normal_(0.0, 0.02, generator=g)
        model.norm.weight.fill_(1.0)
    for q in model.parameters():
        q.requires_grad_(False)
    x = (torch.randn(1, T, D, device=cuda, generator=g) * 0.5).bfloat16()
    labels = torch.randint(0, V, (1, T), device=cuda, generator=g)
    labels[0, :3000] = -100
    labels[0, 11000:12500] = -100
    xa = x.clone().requires_grad_()
    loss = model._head(xa, labels)
    loss.backward()
    n_tot = int((labels != -100).sum())
    acc, parts = 0.0, []
    for r0 in range(0, T, 5120):
        xs = x[:, r0 : r0 + 5120].clone().requires_grad_()
        ls = labels[:, r0 : r0 + 5120]
        li = model._head(xs, ls)
        li.backward()
        n_i = int((ls != -100).sum())
        acc += float(li) * n_i / n_tot
        parts.append(xs.grad.float() * (n_i / n_tot))
    assert abs(float(loss) - acc) <= 2e-5 * abs(acc), (float(loss), acc)
    want = torch.cat(parts, dim=1)
    got = xa.grad.float()
    assert torch.equal(got[0, :3000], torch.zeros_like(got[0, :3000])), "ignore_index rows have a zero gradient"
    err = (got - want).abs().max().item()
    assert err <= 0.02 * want.abs().max().item(), (err, want.abs().max().item())


"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU, and pin oracle/ref.py against it.

Run here (the authoring container) only:   python oracle/gen_golden.py
/root/reference never travels; what is committed are the output vectors (inputs are regenerated from the numpy PCG64
generator in oracle/ref.py on both sides) together with this script.

How the reference is imported:
  * sys.dont_write_bytecode so nothing is written under /root/reference;
  * /root/reference is put first on sys.path so `import modelling` / `import subclasses` resolve to the reference;
  * torchaudio is not installed: a stand-in `torchaudio.transforms.MelSpectrogram` that returns the ORACLE's mel is
    registered so that modelling/audio.py imports.  Everything in LlamaAudio.forward downstream of the mel tensor is the
    reference's own code; the mel arithmetic itself stays "parity unpinned" (oracle/ref.py:mel_spectrogram).
Every comparison below asserts oracle == reference (tight fp32 tolerance, exact for integer/bit-level claims) before
the fixture is written, so a fixture can only exist if the oracle agreed with the reference when it was made.
"""
import hashlib
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ref as O  # noqa: E402

try:  # imported BEFORE the torchaudio stand-in is registered (transformers probes for the real package)
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
except Exception:  # noqa: BLE001
    mel_filter_bank = None

# ---- stand-in for the missing torchaudio (see module docstring)
_ta = types.ModuleType("torchaudio")
_tr = types.ModuleType("torchaudio.transforms")


class _MelStandIn(torch.nn.Module):
    def __init__(self, sample_rate, n_fft, win_length, hop_length, n_mels, norm, mel_scale):
        super().__init__()
        assert norm == "slaney" and mel_scale == "slaney"
        self.ac = O.AudioCfg(sample_rate, n_fft, win_length, hop_length, n_mels)
        self.spectrogram = types.SimpleNamespace(forward=lambda *a, **k: None)  # attribute touched at modelling/audio.py:36

    def forward(self, audio):
        return O.mel_spectrogram(audio, self.ac)


_tr.MelSpectrogram = _MelStandIn
_ta.transforms = _tr
sys.modules["torchaudio"] = _ta
sys.modules["torchaudio.transforms"] = _tr

sys.path.insert(0, REF)
import modelling as RM  # noqa: E402  (the reference's package)
import subclasses as RS  # noqa: E402
from modelling import llama as RL  # noqa: E402
from subclasses import int8 as RI8  # noqa: E402

assert RM.__file__.startswith(REF) and RS.__file__.startswith(REF)

torch.manual_seed(0)
torch.set_num_threads(8)
CFG = O.TINY
RCFG = RM.LlamaConfig(**{f: getattr(CFG, f) for f in RM.LlamaConfig._fields})
saved = {}


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach()
            v = v.float().numpy() if v.dtype is torch.bfloat16 else v.numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    saved[name] = sum(a.nbytes for a in out.values())


def close(a, b, tol=1e-5, what=""):
    a, b = a.detach().float(), b.detach().float()
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"{what}: oracle vs reference differ by {err} (scale {scale})"


def ref_model(params, dtype=torch.float32, audio=False):
    m = (RM.LlamaAudio if audio else RM.Llama)(RCFG)
    missing = m.load_state_dict({k: v.to(dtype) for k, v in params.items()}, strict=False)
    assert not missing.unexpected_keys
    m = m.to(dtype)
    m.build_cache()
    return m


def tokens_labels(B, S, seed=0):
    tokens = O.randint("tokens", (B, S), 0, CFG.vocab_size, seed)
    labels = torch.roll(tokens, -1, 1).clone()
    labels[:, : S // 4] = -100
    labels[:, -1] = -100
    return tokens, labels


# ------------------------------------------------------------------------------------------------- G1 rope tables
t_ref = RL.build_rope(RCFG)
t_or = O.rope_table(CFG)
assert torch.equal(t_ref, t_or), "rope table must be bit-identical"
freqs = 1.0 / (500_000 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
assert torch.equal(RL.scale_llama3_1_rope(freqs), O.llama31_rescale(freqs))
big = O.LLAMA31_8B
t8 = RL.build_rope(RM.LlamaConfig(**{f: getattr(big, f) for f in RM.LlamaConfig._fields}))
assert torch.equal(t8, O.rope_table(big))
save("g01_rope", table_tiny=t_ref[:8], scaled_freqs=RL.scale_llama3_1_rope(freqs),
     sha256_8b_table=np.frombuffer(hashlib.sha256(t8.numpy().tobytes()).digest(), dtype=np.uint8))

# ------------------------------------------------------------------------------------------------- G2 apply_rope
x = O.randn("rope_x", (2, 256, 5, 128))
y_ref = RL.apply_rope(x, t_ref[:256])
assert torch.equal(y_ref, O.rope_apply(x, t_or))
xb = x.bfloat16()
yb_ref = RL.apply_rope(xb, t_ref[:256])
assert torch.equal(yb_ref, O.rope_apply(xb, t_or))
save("g02_apply_rope", y_f32_slice=y_ref[:, ::16, :, ::8], y_bf16_slice=yb_ref[:, ::16, :, ::8])

# ------------------------------------------------------------------------------------------------- G3 RMSNorm
norm = torch.nn.RMSNorm(512, eps=1e-5)
w = 1 + O.randn("norm_w", (512,), 0.1)
norm.weight.data.copy_(w)
x = O.randn("norm_x", (300, 512))
dy = O.randn("norm_dy", (300, 512))
xr = x.clone().requires_grad_()
yr = norm(xr)
yr.backward(dy)
xo, wo = x.clone().requires_grad_(), w.clone().requires_grad_()
yo = O.rmsnorm(xo, wo)
yo.backward(dy)
close(yo, yr, 1e-6, "rmsnorm fwd")
close(xo.grad, xr.grad, 1e-5, "rmsnorm dx")
close(wo.grad, norm.weight.grad, 1e-5, "rmsnorm dw")
nb = torch.nn.RMSNorm(512, eps=1e-5).bfloat16()
nb.weight.data.copy_(w.bfloat16())
assert torch.equal(nb(x.bfloat16()), O.rmsnorm(x.bfloat16(), w.bfloat16())), "bf16 RMSNorm is a single rounding (SURVEY 8c)"
save("g03_rmsnorm", y=yr[::10, ::4], dx=xr.grad[::10, ::4], dw=norm.weight.grad, y_bf16=nb(x.bfloat16())[::10, ::4])

# ------------------------------------------------------------------------------------------------- G4-G7 model, fp32
params = O.init_params(CFG)
model = ref_model(params)
tokens, labels = tokens_labels(2, 256)
logits_ref = model(tokens)
logits_or = O.llama_forward(tokens, params, CFG)
close(logits_or, logits_ref, 2e-5, "llama logits")
pr = {k: v.clone().requires_grad_() for k, v in params.items()}
loss_or = O.llama_forward(tokens, pr, CFG, labels=labels)
loss_or.backward()
loss_ref = model(tokens, labels=labels)
loss_ref.backward()
close(loss_or, loss_ref, 1e-6, "llama loss")
gsel = {}
for name, prm in model.named_parameters():
    close(pr[name].grad, prm.grad, 5e-5, f"grad {name}")
for name in ("layers.0.attention.wq.weight", "layers.1.feed_forward.w2.weight", "layers.0.attention_norm.weight", "norm.weight",
             "layers.1.attention.wk.weight"):
    g = dict(model.named_parameters())[name].grad
    gsel[name.replace(".", "_")] = g if g.dim() == 1 else g[::8, ::8]
save("g07_llama_fp32", logits_slice=logits_ref[:, ::8, ::8], loss=loss_ref, **gsel)

# per-module outputs (G4 attention, G5 MLP, G6 layer) on a fixed hidden state
h = O.randn("hidden", (2, 256, 512), 0.5)
layer0 = model.layers[0]
rope = model.rope[:256]
a_ref = layer0.attention(h, rope)
a_or = O.attention(h, params, "layers.0.attention.", CFG, t_or[:256], None)
close(a_or, a_ref, 2e-5, "attention causal")
f_ref = layer0.feed_forward(h)
close(O.feed_forward(h, params, "layers.0.feed_forward."), f_ref, 2e-5, "feed_forward")
l_ref = layer0(h, rope)
close(O.layer(h, params, 0, CFG, t_or[:256], None), l_ref, 2e-5, "layer")
save("g04_modules", attn=a_ref[:, ::8, ::8], mlp=f_ref[:, ::8, ::8], layer=l_ref[:, ::8, ::8])

# document mask through FlexAttention (the reference's block_mask path) and prefix-LM through mask=
from torch.nn.attention.flex_attention import create_block_mask  # noqa: E402

S = 384
doc = torch.zeros(S, dtype=torch.int64)
for c in (70, 150, 301):
    doc[c:] += 1
doc[S - 20 :] = 0
tokens1, labels1 = tokens_labels(1, S)


def mask_mod(b, hh, q_idx, kv_idx):  # the reference's closure, train_metamathqa.py:67-68
    return (doc[q_idx] == doc[kv_idx]) & (q_idx >= kv_idx)


bm = create_block_mask(mask_mod, 1, None, S, S, device="cpu")
loss_doc_ref = model(tokens1, labels=labels1, block_mask=bm)
loss_doc_or = O.llama_forward(tokens1, params, CFG, mask=O.document_mask(doc)[None, None], labels=labels1)
close(loss_doc_or, loss_doc_ref, 2e-6, "document-mask loss")
P = torch.tensor([128])
dense = O.prefix_lm_mask(S, P)
hid = O.randn("hidden1", (1, S, 512), 0.5)
pl_ref = model.layers[0](hid, model.rope[:S], mask=dense)
close(O.layer(hid, params, 0, CFG, t_or[:S], dense), pl_ref, 2e-5, "prefix-LM layer via mask=")
logits_prefix_or = O.llama_forward(tokens1, params, CFG, mask=dense)
x_ = model.tok_embeddings(tokens1)
for lyr in model.layers:
    x_ = lyr(x_, model.rope[:S], mask=dense)
logits_prefix_ref = model.output(model.norm(x_))
close(logits_prefix_or, logits_prefix_ref, 2e-5, "prefix-LM logits")
save("g04_masks", loss_doc=loss_doc_ref, prefix_layer=pl_ref[:, ::8, ::8], prefix_logits=logits_prefix_ref[:, ::8, ::8], doc_ids=doc)

# ------------------------------------------------------------------------------------------------- G7 bf16 model
pb = {k: v.bfloat16() for k, v in params.items()}
mb = ref_model(pb, torch.bfloat16)
mb.rope = model.rope.clone()  # keep the fp32 table (build_cache after .to(dtype) already gives fp32)
with torch.no_grad():
    lb_ref = mb(tokens)
lb_or = O.llama_forward(tokens, pb, CFG)
# both sides are bf16 eager with different internal accumulation orders: statistical agreement only
assert (lb_or.float() - lb_ref.float()).abs().max() < 0.05
save("g07_llama_bf16", logits_slice=lb_ref[:, ::8, ::8], loss=mb(tokens, labels=labels))

# ------------------------------------------------------------------------------------------------- G8 LoRA / DoRA
for rank in (8, 16):
    lp = O.init_lora(CFG, rank)
    m = ref_model(params)
    for q in m.parameters():
        q.requires_grad_(False)
    RM.apply_linear_adapter_(m.layers, "lora", rank=rank, alpha=float(rank))
    with torch.no_grad():
        for name, mod in m.layers.named_modules():
            if f"layers.{name}.lora_a" in lp:
                mod.lora_a.copy_(lp[f"layers.{name}.lora_a"])
                mod.lora_b.copy_(lp[f"layers.{name}.lora_b"])
    loss = m(tokens, labels=labels)
    loss.backward()
    allp = dict(params)
    allp.update({k: v.clone().requires_grad_() for k, v in lp.items()})
    lo = O.llama_forward(tokens, allp, CFG, labels=labels, lora_scale=1.0)
    lo.backward()
    close(lo, loss, 1e-6, f"lora r{rank} loss")
    out = {"loss": loss}
    for name, q in m.named_parameters():
        if "lora_" in name:
            close(allp[name].grad, q.grad, 5e-5, f"lora grad {name}")
    for name in ("layers.0.attention.wq.lora_a", "layers.0.attention.wq.lora_b", "layers.1.feed_forward.w2.lora_a", "layers.1.feed_forward.w2.lora_b",
                 "layers.0.attention.wv.lora_b"):
        out[name.replace(".", "_")] = dict(m.named_parameters())[name].grad
    save(f"g08_lora_r{rank}", **out)

lin = torch.nn.Linear(512, 256, bias=True)
lin.weight.data.copy_(O.randn("dora_w", (256, 512), 0.05))
lin.bias.data.copy_(O.randn("dora_b", (256,), 0.05))
lin.__class__ = RM.DoRALinear
lin.init_adapter(rank=8, alpha=16.0)
lin.lora_a.data.copy_(O.randn("dora_a", (8, 512), 0.05))
lin.lora_b.data.copy_(O.randn("dora_lb", (256, 8), 0.05))
xd = O.randn("dora_x", (40, 512))
yd = lin(xd)
close(O.dora_linear(xd, lin.weight, lin.lora_a, lin.lora_b, lin.m, 2.0, lin.bias), yd, 1e-5, "dora")
save("g08_dora", y=yd, m=lin.m)

# DoRA on every linear of the layers (apply_linear_adapter_(model.layers, "dora"), the scripts' --adapter dora): loss and the gradients
# of m / lora_a / lora_b through the whole model
lp = O.init_lora(CFG, 8)
m = ref_model(params)
for q in m.parameters():
    q.requires_grad_(False)
RM.apply_linear_adapter_(m.layers, "dora", rank=8, alpha=16.0)
dp_ = dict(params)
dm_ = O.init_dora_m(params, CFG)
with torch.no_grad():
    for name, mod in m.layers.named_modules():
        if f"layers.{name}.lora_a" in lp:
            mod.lora_a.copy_(lp[f"layers.{name}.lora_a"])
            mod.lora_b.copy_(lp[f"layers.{name}.lora_b"])
            close(mod.m, mod.weight.norm(p=2, dim=1), 1e-6, "dora m init")  # the reference initialises m = ||W||_row (lora.py:51)
            mod.m.copy_(dm_[f"layers.{name}.m"])  # then moved off that value
            dp_[f"layers.{name}.m"] = mod.m.detach().clone().requires_grad_()
dp_.update({k: v.clone().requires_grad_() for k, v in lp.items()})
loss_d = m(tokens, labels=labels)
loss_d.backward()
lo_d = O.llama_forward(tokens, dp_, CFG, labels=labels, lora_scale=2.0)
lo_d.backward()
close(lo_d, loss_d, 1e-6, "dora model loss")
out = {"loss": loss_d}
for name, q in m.named_parameters():
    if q.requires_grad:
        close(dp_[name].grad, q.grad, 5e-5, f"dora grad {name}")
for name in ("layers.0.attention.wq.m", "layers.0.attention.wq.lora_a", "layers.1.feed_forward.w2.m", "layers.1.feed_forward.w1.lora_b",
             "layers.0.attention.wv.m"):
    out[name.replace(".", "_")] = dict(m.named_parameters())[name].grad
save("g08_dora_model", **out)

# ------------------------------------------------------------------------------------------------- G9 int8
for dt in (torch.float32, torch.bfloat16):
    w8 = O.randn("q_w", (96, 512), 0.05).to(dt)
    w8[5] = 0  # an all-zero row exercises the 1e-12 clip
    q_ref, s_ref = RI8.quantize_int8_rowwise(w8)
    q_or, s_or = O.quantize_int8_rowwise(w8)
    assert torch.equal(q_ref, q_or) and torch.equal(s_ref, s_or), "quantiser must be bit-exact"
    if dt is torch.bfloat16:
        save("g09_quant_bf16", q=q_ref, scale=s_ref)
wq = RI8.Int8LinearWeight.from_float(O.randn("i8_w", (256, 512), 0.05).bfloat16())
xi = O.randn("i8_x", (40, 512)).bfloat16().requires_grad_()
yi = torch.nn.functional.linear(xi, wq, None)  # the reference needs the bias argument spelled out (subclasses/int8.py:108)
gi = O.randn("i8_g", (40, 256)).bfloat16()
yi.backward(gi)
xo = xi.detach().clone().requires_grad_()
yo = O.int8_linear(xo, wq.int_data, wq.scale, dynamic=False)
yo.backward(gi)
assert torch.equal(yo, yi), "weight-only int8 forward"
assert torch.equal(xo.grad, xi.grad), "int8 backward"
assert torch.equal(O.int8_dequantize(wq.int_data, wq.scale), wq.dequantize())
moved = wq.to(torch.float32)  # _to_copy: scale dtype changes, int_data untouched
assert moved.scale.dtype is torch.float32 and moved.int_data.dtype is torch.int8 and moved.dtype is torch.float32
dst = torch.zeros(256, 512, dtype=torch.bfloat16)
dst.copy_(wq)
assert torch.equal(dst, wq.dequantize())
save("g09_int8_linear", y=yi, dx=xi.grad, dequant_slice=wq.dequantize()[::8, ::8])

# G10 (int8_mm_dequant) and G12 (the training scripts' iterators) are written by oracle/gen_golden_scripts.py, which EXECUTES the
# reference's Triton kernel under TRITON_INTERPRET=1 and imports the scripts behind stand-in modules.

# ------------------------------------------------------------------------------------------------- G11 audio path given mel
pa = O.init_params(CFG, audio=True)
ma = ref_model(pa, audio=True)
audio = O.uniform("audio", (1, 16000), -0.1, 0.1)  # 1 s -> 101 frames -> 100 -> 50 audio tokens
ttok, tlab = tokens_labels(1, 128)
la_ref = ma(audio, ttok)
la_or = O.llama_audio_forward(audio, ttok, pa, CFG)
close(la_or, la_ref, 3e-5, "audio logits")
loss_a_ref = ma(audio, ttok, labels=tlab)
close(O.llama_audio_forward(audio, ttok, pa, CFG, labels=tlab), loss_a_ref, 1e-6, "audio loss")
mel = O.mel_spectrogram(audio)
feat = O.log_mel_cmn(mel)
save("g11_audio", mel=mel, feat=feat, audio_tokens=ma.audio_embed(feat).transpose(1, 2)[:, ::2, ::8], logits_slice=la_ref[:, ::4, ::8], loss=loss_a_ref)

# mel cross-check against an independent implementation (transformers.audio_utils) - not the oracle, a sanity bound
if mel_filter_bank is not None:
    fb2 = mel_filter_bank(num_frequency_bins=257, num_mel_filters=128, min_frequency=0.0, max_frequency=8000.0, sampling_rate=16000,
                          norm="slaney", mel_scale="slaney")
    spec2 = spectrogram(audio[0].numpy().astype(np.float64), window_function(400, "hann", periodic=True), frame_length=400, hop_length=160,
                        fft_length=512, power=2.0, center=True, pad_mode="reflect", mel_filters=fb2)
    rel = np.abs(spec2 - mel[0].numpy()).max() / np.abs(spec2).max()
    print(f"mel cross-check vs transformers.audio_utils: max rel diff {rel:.2e}")
    assert rel < 1e-3
else:
    print("transformers.audio_utils not importable: mel cross-check skipped")

# ------------------------------------------------------------------------------------------------- G14 KV-cache inference
mi = ref_model(params)
mi.build_cache(inference=True)
mi.eval()
tok_i, _ = tokens_labels(1, 96)
cache = O.new_cache(CFG)
with torch.no_grad():
    pre_ref = mi(tok_i[:, :64], input_pos=torch.arange(64))
    pre_or = O.llama_forward_cached(tok_i[:, :64], params, CFG, cache, torch.arange(64))
    close(pre_or, pre_ref, 2e-5, "kv-cache prefill")
    close(pre_ref, logits_ref.new_tensor(O.llama_forward(tok_i[:, :64], params, CFG)), 2e-5, "prefill == causal forward")
    dec = []
    for t in range(64, 68):
        d_ref = mi(tok_i[:, t : t + 1], input_pos=torch.tensor([t]))
        d_or = O.llama_forward_cached(tok_i[:, t : t + 1], params, CFG, cache, torch.tensor([t]))
        close(d_or, d_ref, 2e-5, f"kv-cache decode {t}")
        dec.append(d_ref[:, 0, ::8])
save("g14_kv_cache", prefill_slice=pre_ref[:, ::8, ::8], decode_slices=torch.stack(dec))

# ------------------------------------------------------------------------------------------------- G13 3-step trajectory
lp = O.init_lora(CFG, 8)
m = ref_model(params)
for q in m.parameters():
    q.requires_grad_(False)
RM.apply_linear_adapter_(m.layers, "lora", rank=8, alpha=8.0)
with torch.no_grad():
    for name, mod in m.layers.named_modules():
        if f"layers.{name}.lora_a" in lp:
            mod.lora_a.copy_(lp[f"layers.{name}.lora_a"])
            mod.lora_b.copy_(lp[f"layers.{name}.lora_b"])
opt = torch.optim.AdamW([q for q in m.parameters() if q.requires_grad], lr=1e-3, weight_decay=0.0)
batches = [tokens_labels(1, 256, seed=s) for s in range(3)]
ref_losses = []
for tk, lb in batches:
    loss = m(tk, labels=lb)
    loss.backward()
    opt.step()
    opt.zero_grad()
    ref_losses.append(loss.item())
allp = dict(params)
allp.update({k: v.clone() for k, v in lp.items()})
or_losses = O.train_steps(allp, sorted(lp), [(tk, lb, None) for tk, lb in batches], CFG, lr=1e-3)
assert max(abs(a - b) for a, b in zip(or_losses, ref_losses)) < 1e-5, (or_losses, ref_losses)
save("g13_trajectory", losses=np.array(ref_losses, dtype=np.float64))

print("golden fixtures written:")
for k, v in saved.items():
    print(f"  {k}.npz  {v / 1024:.1f} KiB (uncompressed)")
assert not any(f.endswith(".pyc") for _, _, fs in os.walk(REF) for f in fs), "bytecode was written under /root/reference"

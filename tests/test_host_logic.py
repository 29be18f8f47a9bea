"""CPU: host-side logic of the product and the bit-exact integer contracts (index / label / mask construction)."""
import math

import pytest
import torch

from oracle import ref as O


# ---------------------------------------------------------------------------------------------- M1 padding iterator
def test_pad_batch_hand_case():
    toks = [torch.arange(1, 7), torch.arange(10, 13)]  # lengths 6 and 3 -> n = 5 and 2 ; multiple 4 -> L = 8
    inputs, labels = O.pad_batch(toks, seq_len_multiple=4)
    assert inputs.tolist() == [[1, 2, 3, 4, 5, 0, 0, 0], [10, 11, 0, 0, 0, 0, 0, 0]]
    assert labels.tolist() == [[2, 3, 4, 5, 6, -100, -100, -100], [11, 12, -100, -100, -100, -100, -100, -100]]
    assert inputs.dtype is torch.int64 and labels.dtype is torch.int64
    assert O.next_multiple(255, 256) == 256 and O.next_multiple(256, 256) == 256 and O.next_multiple(257, 256) == 512


def test_product_data_matches_oracle():
    from llx import data as D

    toks = [O.randint(f"d{i}", (n,), 1, 1000) for i, n in enumerate((17, 300, 64, 257, 5, 129))]
    a, b = D.pad_batch(toks[:3], 256), O.pad_batch(toks[:3], 256)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    pa = list(D.pack_documents(toks * 3, 512))
    pb = list(O.pack_documents(toks * 3, 512))
    assert len(pa) == len(pb) > 0
    for x, y in zip(pa, pb):
        assert all(torch.equal(u, v) for u, v in zip(x, y))
    batch = [(torch.ones(1000), [1, 5, 6, 7, 2]), (torch.ones(500), [1, 9, 2])]
    for u, v in zip(D.prepare_audio_batch(batch, 1600, 4, 0), O.prepare_audio_batch(batch, 1600, 4, 0)):
        assert torch.equal(u, v)
    for step in range(0, 120, 7):
        assert D.LRScheduler(1e-3, 100, 0.1, 0.2).get_lr(step) == O.lr_at(step, 1e-3, 100, 0.1, 0.2)


# ---------------------------------------------------------------------------------------------- M2 packer + mask
def test_pack_documents_quirks():
    docs = [torch.arange(100, 100 + n) for n in (5, 4, 6, 3, 4)]
    out = list(O.pack_documents(docs, seq_len=8))
    # doc0 (4 tokens) + doc1 (3) fit in 8; doc2 (5) would overflow (7 + 5 > 8) -> flush
    inp, lab, ids = out[0]
    assert inp.tolist() == [100, 101, 102, 103, 100, 101, 102, 0]
    assert lab.tolist() == [101, 102, 103, 104, 101, 102, 103, -100]
    assert ids.tolist() == [0, 0, 0, 0, 1, 1, 1, 0]  # unused tail carries id 0 (doc_ids re-zeroed, :75)
    inp2, lab2, ids2 = out[1]
    assert ids2.tolist() == [2, 2, 2, 2, 2, 3, 3, 0], "document counter is never reset across buffers (:56,:83)"
    # flush test is i + len(tokens) - 1 > seq_len: an exact fit does not flush
    exact = list(O.pack_documents([torch.arange(5), torch.arange(5), torch.arange(3)], seq_len=8))
    assert exact[0][2].tolist() == [0, 0, 0, 0, 1, 1, 1, 1]


def test_document_mask_bits():
    ids = torch.tensor([0, 0, 0, 1, 1, 0, 0])  # first buffer: tail id 0 shares the first document's id
    m = O.document_mask(ids)
    expect = torch.tensor([[int(ids[q] == ids[k] and q >= k) for k in range(7)] for q in range(7)], dtype=torch.bool)
    assert torch.equal(m, expect)
    assert m[5, 0] and m[6, 2] and not m[5, 3], "tail rows attend causally to document 0 (SURVEY M2 quirk)"
    p = O.prefix_lm_mask(5, [2])
    assert p[0, 0].int().tolist() == [[1, 1, 0, 0, 0], [1, 1, 0, 0, 0], [1, 1, 1, 0, 0], [1, 1, 1, 1, 0], [1, 1, 1, 1, 1]]
    assert torch.equal(O.causal_mask(4), torch.tril(torch.ones(4, 4, dtype=torch.bool)))


# ---------------------------------------------------------------------------------------------- M4 audio batch
def test_prepare_audio_batch():
    batch = [(torch.ones(5), [1, 7, 8, 2]), (torch.ones(3), [1, 9, 2, 4, 4, 2])]
    audio, tokens, labels = O.prepare_audio_batch(batch, audio_length=8, seq_len_multiple=4, pad_id=0)
    assert audio.shape == (2, 8) and audio[0].tolist() == [1, 1, 1, 1, 1, 0, 0, 0]
    assert tokens.tolist() == [[1, 7, 8, 2, 0, 0, 0, 0], [1, 9, 2, 4, 4, 2, 0, 0]]
    assert labels.tolist() == [[7, 8, 2, -100, -100, -100, -100, -100], [9, 2, 4, 4, 2, -100, -100, -100]]


# ---------------------------------------------------------------------------------------------- N3 LibriSpeech utterance packer
def _clips():
    """Synthetic utterances at 10 Hz 'sample rate': name -> (waveform [channels, n], fs); durations 0.5 / 0.4 / 1.2 / 0.3 / 0.6 s."""
    wav = lambda n, v, ch=1: torch.full((ch, n), float(v))  # noqa: E731
    return {"a.flac": (wav(5, 1), 10), "b.flac": (wav(4, 2, ch=2), 10), "c.flac": (wav(12, 3), 10), "d.flac": (wav(3, 4), 10),
            "e.flac": (wav(6, 5), 10)}


def test_utterance_packer_hand_case():
    """train_librispeech.py:88-124 on a hand-computed case: limit 1.0 s, batch of 2 packs, seq multiple 4, bos 1 / eos 2 / pad 0."""
    clips = _clips()
    samples = [("a.flac", [10, 11]), ("b.flac", [20]), ("c.flac", [30, 31]), ("d.flac", [40, 41, 42]), ("e.flac", [50])]
    kw = dict(audio_duration=1.0, sample_rate=10, batch_size=2, seq_len_multiple=4, bos_id=1, eos_id=2, pad_id=0)
    st = {}
    out = list(O.pack_utterances(samples, clips.__getitem__, [0, 1, 2, 3, 4, 0], state=st, **kw))
    # a (0.5) + b (0.4) = 0.9; c (1.2 s) is over-long: skipped; d (0.3) would make 1.2 > 1.0 -> pack {a,b} closes, d opens;
    # e (0.6): 0.3 + 0.6 = 0.9 fits; a again (0.5): 0.9 + 0.5 > 1.0 -> pack {d,e} closes -> batch of 2 emitted; a opens the next pack
    assert len(out) == 1
    audio, tokens, labels = out[0]
    assert audio.shape == (2, 10) and audio.dtype is torch.float32
    assert audio[0].tolist() == [1, 1, 1, 1, 1, 2, 2, 2, 2, 0]      # b is stereo: channel mean; zero padded to 1.0 s
    assert audio[1].tolist() == [4, 4, 4, 5, 5, 5, 5, 5, 5, 0]
    assert tokens.tolist() == [[1, 10, 11, 20, 2, 0, 0, 0], [1, 40, 41, 42, 50, 2, 0, 0]]
    assert labels.tolist() == [[10, 11, 20, 2, -100, -100, -100, -100], [40, 41, 42, 50, 2, -100, -100, -100]]
    assert st["tokens"] == [1, 10, 11] and st["duration"] == 0.5 and st["batch"] == [], "the overflowing clip opens the next pack"


def test_utterance_packer_product_matches_oracle(tmp_path):
    from llx import data as D

    clips = _clips()
    gen = torch.Generator().manual_seed(7)
    lens = torch.randint(2, 9, (40,), generator=gen).tolist()
    big = {f"u{i:02d}.flac": (torch.rand(1 + i % 2, n, generator=gen), 10) for i, n in enumerate(lens)}
    big["long.flac"] = (torch.rand(1, 15, generator=gen), 10)
    samples = [(k, [100 + i, 200 + i][: 1 + i % 2]) for i, k in enumerate(sorted(big))]
    kw = dict(audio_duration=1.0, batch_size=3, seq_len_multiple=4, bos_id=1, eos_id=2, pad_id=0)
    g1 = torch.Generator().manual_seed(3)
    it = iter(D.UtterancePacker(samples, big.__getitem__, sample_rate=10, generator=g1, **kw))
    got = [next(it) for _ in range(7)]  # spans more than one shuffled pass
    g2 = torch.Generator().manual_seed(3)
    want, st = [], {}
    while len(want) < 7:
        want += list(O.pack_utterances(samples, big.__getitem__, torch.randperm(len(samples), generator=g2), sample_rate=10, state=st, **kw))
    for a, b in zip(got, want):
        assert all(torch.equal(u, v) for u, v in zip(a, b))
    # transcript listing incl. the reference's de-indentation quirk (:55-61): one sample per *.trans.txt, from its LAST line
    d = tmp_path / "libri" / "84" / "121123"
    d.mkdir(parents=True)
    (d / "84-121123.trans.txt").write_text("84-121123-0000 GO DO YOU HEAR\n84-121123-0001 BUT IN LESS THAN FIVE\n")
    d2 = tmp_path / "libri" / "19" / "198"
    d2.mkdir(parents=True)
    (d2 / "19-198.trans.txt").write_text("19-198-0000 NORTHANGER ABBEY\n")
    tok = lambda s: [ord(c) for c in s]  # noqa: E731
    got_s = D.librispeech_samples(tmp_path / "libri", tok)
    want_s = O.list_transcripts([("84/121123", "84-121123.trans.txt", ["84-121123-0000 GO DO YOU HEAR\n", "84-121123-0001 BUT IN LESS THAN FIVE\n"]),
                                 ("19/198", "19-198.trans.txt", ["19-198-0000 NORTHANGER ABBEY\n"])], tok)
    assert got_s == want_s == [("19/198/19-198-0000.flac", tok(" northanger abbey.")), ("84/121123/84-121123-0001.flac", tok(" but in less than five."))]
    assert clips  # (fixture reused above)


def test_lr_schedule():
    f = lambda s: O.lr_at(s, 1.0, 100, 0.1, 0.2)  # noqa: E731
    assert f(0) == 0.0 and f(5) == 0.5 and f(10) == 1.0 and f(79) == 1.0 and f(80) == 1.0 and f(90) == 0.5 and f(100) == 1.0


# ---------------------------------------------------------------------------------------------- G12: product vs the reference's own batches
def _g12(name):
    import os

    import numpy as np

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))


def test_product_iterators_reproduce_reference_fixtures(tmp_path):
    """llx.data against tests/golden/g12_* = what the reference's train_metamathqa.py / train_librispeech.py / train_utils.py
    produced on the same seeded inputs (oracle/gen_golden_scripts.py): M1, M2 (+ mask bits through MaskSpec's rule), M4, the
    utterance packer incl. the transcript listing, and the LR schedule.  Bit-exact."""
    import numpy as np

    from llx import data as D
    from oracle import script_cases as SC

    docs = SC.documents()
    g = _g12("g12_padding")
    torch.manual_seed(SC.PAD_SEED)
    it = D.padding_iterator(list(docs), SC.PAD_BATCH, SC.PAD_MULTIPLE)
    for i in range(SC.PAD_N):
        inputs, labels, mask = next(it)
        assert mask is None and np.array_equal(inputs.numpy(), g[f"inputs_{i}"]) and np.array_equal(labels.numpy(), g[f"labels_{i}"])
    g = _g12("g12_document_mask")
    torch.manual_seed(SC.PACK_SEED)
    it = D.document_mask_iterator(list(docs), SC.PACK_SEQ)
    for i in range(SC.PACK_N):
        inputs, labels, spec = next(it)
        assert inputs.shape == (1, SC.PACK_SEQ) and np.array_equal(inputs.numpy(), g[f"inputs_{i}"]) and np.array_equal(labels.numpy(), g[f"labels_{i}"])
        assert spec.prefix_len is None and np.array_equal(spec.doc_ids.numpy(), g[f"doc_ids_{i}"])
        # the rule the attention kernels apply to a MaskSpec(doc_ids): same document AND q >= kv
        ids = spec.doc_ids
        idx = torch.arange(SC.PACK_SEQ)
        dense = (ids[:, None] == ids[None, :]) & (idx[:, None] >= idx[None, :])
        assert np.array_equal(np.packbits(dense.numpy(), axis=1), g[f"mask_bits_{i}"])
    g = _g12("g12_librispeech")
    tok = SC.ToyTokenizer()
    SC.write_transcripts(tmp_path)
    samples = D.librispeech_samples(tmp_path, tok)
    assert [p for p, _ in samples] == g["listing_paths"].tolist() and [len(t) for _, t in samples] == g["listing_tokens"].tolist()
    a, t, lab = D.prepare_audio_batch(SC.prepare_batch_case(), int(SC.AUDIO_SECONDS * SC.AUDIO_RATE), SC.AUDIO_MULTIPLE, tok.pad_id)
    assert np.array_equal(a.numpy(), g["prep_audio"]) and np.array_equal(t.numpy(), g["prep_tokens"]) and np.array_equal(lab.numpy(), g["prep_labels"])
    import os

    clips = SC.clips()
    torch.manual_seed(SC.AUDIO_SEED)
    it = iter(D.UtterancePacker(samples, lambda p: clips[os.path.basename(str(p))], audio_duration=SC.AUDIO_SECONDS, seq_len_multiple=SC.AUDIO_MULTIPLE,
                                batch_size=SC.AUDIO_BATCH, bos_id=tok.bos_id, eos_id=tok.eos_id, pad_id=tok.pad_id, sample_rate=SC.AUDIO_RATE))
    for i in range(SC.AUDIO_N):
        for nm, v in zip(("audio", "tokens", "labels"), next(it)):
            assert np.array_equal(v.numpy(), g[f"{nm}_{i}"]), (i, nm)
    g = _g12("g12_lr_schedule")
    for i, (lr, n, wu, dc) in enumerate(SC.LR_CASES):
        sch = D.LRScheduler(lr, n, wu, dc)
        assert [sch.get_lr(s) for s in range(n + 3)] == g[f"case_{i}"].tolist()
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=torch.tensor(0.5))
        sch.set_lr(opt, n // 2)
        assert isinstance(opt.param_groups[0]["lr"], torch.Tensor) and float(opt.param_groups[0]["lr"]) == np.float32(g[f"case_{i}"][n // 2])


def test_dense_mask_is_recognised_as_a_maskspec_rule():
    """llx.kernels.maskspec_from_dense: the reference's dense `mask=` route (modelling/llama.py:135-137) maps onto the MaskSpec rule when
    the mask IS that rule (causal, prefix-LM per sample, contiguous documents, both together), bit for bit; anything else gives None."""
    from llx import kernels as K

    S = 96
    idx = torch.arange(S)
    sp = K.maskspec_from_dense(O.causal_mask(S), 1, S)
    assert sp is not None and sp.doc_ids is None and sp.prefix_len is None
    sp = K.maskspec_from_dense(O.prefix_lm_mask(S, [40, 7]), 2, S)
    assert sp.doc_ids is None and sp.prefix_len.tolist() == [40, 7]
    sp = K.maskspec_from_dense(O.prefix_lm_mask(S, [1])[0, 0], 1, S)  # P = 1 is the causal mask
    assert sp is not None and sp.prefix_len is None
    doc = torch.zeros(S, dtype=torch.int64)
    doc[20:] += 1
    doc[50:] += 1
    sp = K.maskspec_from_dense(O.document_mask(doc), 1, S)
    assert sp.prefix_len is None and torch.equal(sp.doc_ids[0].long(), doc)
    both = ((idx[None, :] <= idx[:, None]) | (idx[None, :] < 30)) & (doc[:, None] == doc[None, :])
    sp = K.maskspec_from_dense(both[None, None], 1, S)
    assert sp.prefix_len.tolist() == [30] and torch.equal(sp.doc_ids[0].long(), doc)
    rule = ((idx[None, :] <= idx[:, None]) | (idx[None, :] < int(sp.prefix_len[0]))) & (sp.doc_ids[0][:, None] == sp.doc_ids[0][None, :])
    assert torch.equal(rule, both)
    tail = doc.clone()
    tail[80:] = 0  # the packer's first-buffer quirk: the tail shares id 0 with the first document - not a contiguous-document mask
    assert K.maskspec_from_dense(O.document_mask(tail), 1, S) is None
    assert K.maskspec_from_dense(torch.rand(S, S, generator=torch.Generator().manual_seed(0)) < 0.5, 1, S) is None
    assert K.maskspec_from_dense(O.causal_mask(S)[None, None].expand(1, 4, S, S), 1, S) is None  # per-head masks are not a MaskSpec


# ---------------------------------------------------------------------------------------------- product API surface
def test_module_api_and_state_dict_names():
    from modelling import AudioConfig, DoRALinear, Llama, LlamaAudio, LlamaConfig, LoRALinear, apply_linear_adapter_

    assert LlamaConfig._fields == ("embed_dim", "num_layers", "head_dim", "num_heads", "num_kv_heads", "intermediate_dim", "max_seq_len",
                                   "vocab_size", "attn_dropout", "rope_base", "is_llama3_1", "activation_checkpointing")
    d = LlamaConfig(64, 1, 128, 1, 1, 128)
    assert (d.max_seq_len, d.vocab_size, d.attn_dropout, d.rope_base, d.is_llama3_1, d.activation_checkpointing) == (2048, 128256, 0.0, 50000, False, False)
    assert AudioConfig() == (16000, 512, 400, 160, 128)
    cfg = LlamaConfig(256, 2, 128, 2, 1, 512, max_seq_len=64, vocab_size=300)
    m = Llama(cfg)
    m.build_cache()
    keys = set(m.state_dict().keys())
    expect = {"tok_embeddings.weight", "norm.weight", "output.weight"}
    for i in range(2):
        expect |= {f"layers.{i}.attention.{w}.weight" for w in ("wq", "wk", "wv", "wo")}
        expect |= {f"layers.{i}.feed_forward.{w}.weight" for w in ("w1", "w2", "w3")}
        expect |= {f"layers.{i}.attention_norm.weight", f"layers.{i}.ffn_norm.weight"}
    assert keys == expect, "rope / caches are non-persistent buffers"
    assert m.rope.shape == (64, 64, 2) and m.rope.dtype is torch.float32
    assert all(isinstance(x, torch.nn.Linear) for x in (m.output, m.layers[0].attention.wq, m.layers[0].feed_forward.w2))
    apply_linear_adapter_(m.layers, "lora", rank=4, alpha=8.0)
    wq = m.layers[0].attention.wq
    assert isinstance(wq, LoRALinear) and wq.scale == 2.0 and wq.lora_a.shape == (4, 256) and wq.lora_b.shape == (256, 4)
    assert not wq.weight.requires_grad and wq.lora_a.requires_grad and float(wq.lora_b.abs().sum()) == 0.0
    assert not isinstance(m.output, LoRALinear), "only model.layers is adapted (train_metamathqa.py:179)"
    apply_linear_adapter_(m.output, None)
    ma = LlamaAudio(cfg)
    assert {"audio_embed.0.weight", "audio_embed.0.bias", "audio_embed.2.weight", "audio_embed.2.bias"} <= set(ma.state_dict())
    lin = torch.nn.Linear(16, 8)
    apply_linear_adapter_(lin, "dora", rank=2)
    assert isinstance(lin, DoRALinear) and torch.allclose(lin.m, lin.weight.norm(p=2, dim=1))
    with pytest.raises(KeyError):
        apply_linear_adapter_(lin, "nope")


def test_rope_table_product_matches_oracle():
    from modelling.llama import LlamaConfig, build_rope, scale_llama3_1_rope

    cfg = LlamaConfig(4096, 1, 128, 32, 8, 14336, max_seq_len=512, rope_base=500000, is_llama3_1=True)
    assert torch.equal(build_rope(cfg), O.rope_table(O.LLAMA31_8B._replace(max_seq_len=512)))
    f = 1.0 / (500_000 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
    assert torch.equal(scale_llama3_1_rope(f), O.llama31_rescale(f))


def test_hf_key_rename_and_local_loading(tmp_path):
    import json

    from modelling.llama import Llama, _rename_hf_key

    assert _rename_hf_key("model.layers.3.self_attn.q_proj.weight") == "layers.3.attention.wq.weight"
    assert _rename_hf_key("model.layers.0.mlp.down_proj.weight") == "layers.0.feed_forward.w2.weight"
    assert _rename_hf_key("model.layers.0.post_attention_layernorm.weight") == "layers.0.ffn_norm.weight"
    assert _rename_hf_key("model.embed_tokens.weight") == "tok_embeddings.weight" and _rename_hf_key("lm_head.weight") == "output.weight"
    # local checkpoint directory (no network): config.json + safetensors with HF names
    from safetensors.torch import save_file

    hf_cfg = dict(architectures=["LlamaForCausalLM"], hidden_size=128, num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1,
                  intermediate_size=256, vocab_size=64, rope_theta=500000.0, rope_scaling=dict(rope_type="llama3"))
    (tmp_path / "config.json").write_text(json.dumps(hf_cfg))
    sd = {"model.embed_tokens.weight": torch.randn(64, 128), "model.norm.weight": torch.ones(128), "lm_head.weight": torch.randn(64, 128),
          "model.layers.0.input_layernorm.weight": torch.ones(128), "model.layers.0.post_attention_layernorm.weight": torch.ones(128)}
    for hf, shape in (("self_attn.q_proj", (128, 128)), ("self_attn.k_proj", (128, 128)), ("self_attn.v_proj", (128, 128)),
                      ("self_attn.o_proj", (128, 128)), ("mlp.gate_proj", (256, 128)), ("mlp.up_proj", (256, 128)), ("mlp.down_proj", (128, 256))):
        sd[f"model.layers.0.{hf}.weight"] = torch.randn(*shape)
    save_file(sd, str(tmp_path / "model.safetensors"))
    m = Llama.from_hf(str(tmp_path), max_seq_len=32)
    assert m.config.is_llama3_1 and m.config.rope_base == 500000.0 and m.config.max_seq_len == 32
    assert torch.equal(m.layers[0].feed_forward.w1.weight, sd["model.layers.0.mlp.gate_proj.weight"]) and m.rope.shape == (32, 64, 2)


# ---------------------------------------------------------------------------------------------- int8 subclass on the host
def test_int8_weight_subclass_host_behaviour():
    from subclasses import Int8LinearWeight, quantize_linear_
    from subclasses.int8 import quantize_int8_rowwise

    w = O.randn("q_w", (96, 512), 0.05).bfloat16()
    w[5] = 0
    q, s = quantize_int8_rowwise(w)
    qo, so = O.quantize_int8_rowwise(w)
    assert torch.equal(q, qo) and torch.equal(s, so), "host quantiser is bit-exact with the oracle"
    t = Int8LinearWeight.from_float(w, dynamic_int8_act=True)
    assert t.shape == (96, 512) and t.dtype is torch.bfloat16 and t.dynamic_int8_act and t.int_data.dtype is torch.int8
    assert torch.equal(t.dequantize(), O.int8_dequantize(q, s))
    names, attrs = t.__tensor_flatten__()
    assert names == ["int_data", "scale"] and attrs == [True]
    t2 = Int8LinearWeight.__tensor_unflatten__({"int_data": t.int_data, "scale": t.scale}, attrs)
    assert torch.equal(t2.int_data, t.int_data) and t2.dynamic_int8_act
    d = t.detach().clone()
    assert isinstance(d, Int8LinearWeight) and torch.equal(d.int_data, t.int_data)
    f = t.to(torch.float32)
    assert f.scale.dtype is torch.float32 and f.int_data.dtype is torch.int8 and f.dtype is torch.float32
    dst = torch.zeros(96, 512, dtype=torch.bfloat16)
    dst.copy_(t)
    assert torch.equal(dst, t.dequantize())
    t3 = Int8LinearWeight.from_float(torch.zeros(96, 512, dtype=torch.bfloat16))
    t3.copy_(w)  # float -> int8 re-quantises
    assert torch.equal(t3.int_data, q) and torch.equal(t3.scale, s)
    t3.copy_(t)
    with pytest.raises(NotImplementedError):
        torch.add(t, 1)
    lin = torch.nn.Sequential(torch.nn.Linear(512, 96, bias=False))
    lin[0].weight.data.copy_(w.float())
    lin = lin.bfloat16()
    quantize_linear_(lin, "int8")
    assert isinstance(lin[0].weight, Int8LinearWeight) and not lin[0].weight.requires_grad
    sd = lin.state_dict()
    lin2 = torch.nn.Sequential(torch.nn.Linear(512, 96, bias=False)).bfloat16()
    quantize_linear_(lin2, "int8")
    lin2.load_state_dict(sd)  # checkpoint round trip goes through copy_ (train_librispeech.py:200-204)
    assert torch.equal(lin2[0].weight.int_data, q)
    quantize_linear_(lin, None)
    with pytest.raises(KeyError):
        quantize_linear_(lin, "int4")


def test_int8_mm_dequant_meta_and_asserts():
    from subclasses.int8_mm import int8_mm_dequant

    a = torch.zeros(8, 128, dtype=torch.int8, device="meta")
    b = torch.zeros(16, 128, dtype=torch.int8, device="meta").T
    out = int8_mm_dequant(a, b, torch.zeros(8, dtype=torch.bfloat16, device="meta"), torch.zeros(16, dtype=torch.bfloat16, device="meta"))
    assert out.shape == (8, 16) and out.dtype is torch.bfloat16
    with pytest.raises(AssertionError):
        int8_mm_dequant(a.float(), b, torch.zeros(8, device="meta"), torch.zeros(16, device="meta"))
    with pytest.raises(NotImplementedError):  # no CPU kernel, exactly like the reference
        int8_mm_dequant(torch.zeros(8, 128, dtype=torch.int8), torch.zeros(128, 16, dtype=torch.int8), torch.zeros(8), torch.zeros(16))


def test_product_fails_loudly_without_gpu():
    """The product path has no CPU / eager fallback: CPU tensors raise instead of silently computing elsewhere."""
    from llx._lib import LlxError
    from modelling import Llama, LlamaConfig

    m = Llama(LlamaConfig(128, 1, 128, 1, 1, 256, max_seq_len=32, vocab_size=64)).bfloat16()
    m.build_cache()
    with pytest.raises(LlxError):
        m(torch.zeros(1, 8, dtype=torch.int64))
    with pytest.raises(LlxError):
        m.layers[0].attention.wq(torch.zeros(1, 8, 128, dtype=torch.bfloat16))


def test_group_plan_block_diagonal_descriptors():
    """Host side of the block-diagonal LoRA operands: the k ranges (skinny_nt) and member segments (skinny_tn) that a fused
    linear group hands to the kernels."""
    from torch import nn

    from llx.ops import GroupPlan
    from modelling import apply_linear_adapter_

    def group(ns, rank, k=256):
        mods = nn.ModuleList([nn.Linear(k, n, bias=False) for n in ns]).bfloat16()
        apply_linear_adapter_(mods, "lora", rank=rank, alpha=float(rank))
        return GroupPlan(list(mods))

    g = group((4096, 1024, 1024), 16)  # q|k|v at 8B dims
    assert g.fused and g.R == 48
    assert g._kranges() == [0, 4096, 4096, 5120, 5120, 6144, 0, 0]
    assert g._tn_segs() == [(0, 4096, 0, 16), (4096, 5120, 16, 32), (5120, 6144, 32, 48)]
    g = group((14336, 14336), 16)  # gate|up
    assert g._kranges() == [0, 14336, 14336, 28672, 0, 0, 0, 0]
    assert g._tn_segs() == [(0, 14336, 0, 16), (14336, 28672, 16, 32)]
    g = group((512, 512), 8)  # two rank-8 members share the first 16-row block of B^T: its k range is their union
    assert g._kranges() == [0, 1024, 0, 0, 0, 0, 0, 0]
    assert g._tn_segs() == [(0, 512, 0, 8), (512, 1024, 8, 16)]
    g = group((512, 128, 128), 16)  # tiny config: member boundaries at multiples of 64 but not of 256
    assert g._kranges() == [0, 512, 512, 640, 640, 768, 0, 0]
    assert g._tn_segs() is None  # skinny_tn falls back to the full [N, R] product and slices
    g = group((4096,), 16)  # a single member's B^T is dense
    assert g._kranges() is None and g._tn_segs() == [(0, 4096, 0, 16)]
    g = group((200, 312), 16)  # boundary not a multiple of 64: dense
    assert g._kranges() is None and g._tn_segs() is None

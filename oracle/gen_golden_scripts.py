"""Second golden generator: the reference's TRAINING SCRIPTS (integer contracts M1/M2/M4, the utterance packer, the LR
schedule) and the reference's int8 Triton kernel, both EXECUTED here on CPU.  Writes tests/golden/g12_*.npz and g10_int8_mm.npz.

Run here (the authoring container) only:   python oracle/gen_golden_scripts.py
/root/reference never travels: only output vectors are committed; the inputs are regenerated on both sides from the numpy PCG64
generator of oracle/ref.py (and, for shuffles, from torch.manual_seed on the same torch build that runs the tests).

How the scripts are imported (they are written for a GPU box with wandb / tiktoken / torchao / torchaudio installed):
  * sys.dont_write_bytecode, /root/reference first on sys.path;
  * inert stand-in MODULES for wandb, tiktoken(+.load), torchao.prototype.low_bit_optim and torchaudio (none of them is called
    by the functions pinned here, except torchaudio.load, which is given an in-memory loader over synthetic clips);
  * torch.Tensor.cuda is patched to the identity INSIDE THIS PROCESS ONLY (train_metamathqa.py:48,65,71 move batches with .cuda());
  * train_metamathqa.create_block_mask is wrapped so that the reference's own mask_mod closure (:67-68) is evaluated densely on CPU
    (the reference passes no device, so the real call would default to "cuda");
  * train_librispeech.get_tokenizer is replaced by a byte-level toy tokenizer (the real ones download from the HF hub), so
    LibriSpeech.__init__ (:37-66, incl. the transcript-listing quirk of :55-61), _prepare_batch (:68-86) and __iter__ (:88-124)
    run as written.
  * TRITON_INTERPRET=1: subclasses/int8_mm.py:50-118 is run by Triton's CPU interpreter through `_int8_mm_dequant_kernel.fn[grid]`
    (the autotuner wrapper needs a GPU driver; `.fn` is the jitted function itself) with explicit block sizes.  The interpreter
    rounds the final store to bf16 by truncation (an interpreter artefact), so the kernel is run with an fp32 output and fp32
    scale BUFFERS holding the bf16-representable scale values: the fp32 value each program computes before the store is then
    exactly what the bf16 kernel computes (`.to(tl.float32)` of a bf16 scale is exact), and the bf16 result is its
    round-to-nearest-even cast (what `tl.store` to a bf16 pointer does on hardware).
Every fixture is written only after the oracle (oracle/ref.py) reproduced the reference's output exactly.
"""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
os.environ["TRITON_INTERPRET"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ref as O  # noqa: E402
from oracle import script_cases as SC  # noqa: E402  (the shared, seeded inputs of the g12 cases)


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_module("wandb")
_module("tiktoken", Encoding=object)
_module("tiktoken.load", load_tiktoken_bpe=lambda *a, **k: {})
_lbo = _module("torchao.prototype.low_bit_optim", AdamW8bit=object, AdamW4bit=object)
_module("torchao.prototype", low_bit_optim=_lbo)
_module("torchao", prototype=sys.modules["torchao.prototype"])
_ta = _module("torchaudio")
_module("torchaudio.transforms", MelSpectrogram=torch.nn.Identity)
_ta.transforms = sys.modules["torchaudio.transforms"]

sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self  # generator process only

import train_librispeech as TL  # noqa: E402
import train_metamathqa as TM  # noqa: E402
import train_utils as TU  # noqa: E402
import triton  # noqa: E402
from subclasses import int8_mm as RMM  # noqa: E402

for mod in (TL, TM, TU, RMM):
    assert mod.__file__.startswith(REF)

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


def same(a, b, what):
    assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), f"{what}: oracle != reference"


# ------------------------------------------------------------------------------------------------- G12a  _data_iter_padding (M1)
docs = SC.documents()
torch.manual_seed(SC.PAD_SEED)
it = TM._data_iter_padding(list(docs), SC.PAD_BATCH, SC.PAD_MULTIPLE)
ref_batches = [next(it) for _ in range(SC.PAD_N)]  # spans two shuffles of the 11 documents
torch.manual_seed(SC.PAD_SEED)
or_batches = SC.oracle_padding_batches(docs)
out = {}
for i, ((ri, rl, rm), (oi, ol)) in enumerate(zip(ref_batches, or_batches)):
    assert rm is None
    same(oi, ri, f"padding inputs {i}")
    same(ol, rl, f"padding labels {i}")
    out[f"inputs_{i}"], out[f"labels_{i}"] = ri, rl
save("g12_padding", **out)

# ------------------------------------------------------------------------------------------------- G12b  _data_iter_document_mask (M2)
captured = []


def _dense_block_mask(mask_mod, B, H, Q, KV, **kw):
    """Stand-in for create_block_mask: evaluates the reference's closure on every (q, kv) pair; keeps the doc_ids it closed over."""
    q = torch.arange(Q).view(-1, 1).expand(Q, KV)
    kv = torch.arange(KV).view(1, -1).expand(Q, KV)
    dense = mask_mod(0, 0, q, kv)
    (cell,) = [c.cell_contents for c in mask_mod.__closure__ if isinstance(c.cell_contents, torch.Tensor)]
    captured.append((cell.clone(), dense.clone()))
    return ("dense", len(captured) - 1)


TM.create_block_mask = _dense_block_mask
torch.manual_seed(SC.PACK_SEED)
it = TM._data_iter_document_mask(list(docs), SC.PACK_SEQ)
ref_packs = [next(it) for _ in range(SC.PACK_N)]
torch.manual_seed(SC.PACK_SEED)
or_packs = SC.oracle_packed_buffers(docs)
out = {}
for i, ((ri, rl, (_, ci)), (oi, ol, od)) in enumerate(zip(ref_packs, or_packs)):
    ids, dense = captured[ci]
    same(oi.view(1, -1), ri, f"packed inputs {i}")
    same(ol.view(1, -1), rl, f"packed labels {i}")
    same(od, ids, f"doc ids {i}")
    same(O.document_mask(od), dense, f"dense document mask {i}")
    out[f"inputs_{i}"], out[f"labels_{i}"], out[f"doc_ids_{i}"] = ri, rl, ids
    out[f"mask_bits_{i}"] = np.packbits(dense.numpy(), axis=1)
assert int(captured[1][0].max()) > int(captured[0][0].max()), "document ids keep growing across buffers (:56,:83)"
save("g12_document_mask", **out)

# ------------------------------------------------------------------------------------------------- G12c  LibriSpeech (M4, packer)
TL.get_tokenizer = lambda name: SC.ToyTokenizer()
clips = SC.clips()
_ta.load = lambda path: clips[os.path.basename(str(path))]
with tempfile.TemporaryDirectory() as tmp:
    SC.write_transcripts(tmp)
    ds = TL.LibriSpeech(tmp, "toy", audio_duration=SC.AUDIO_SECONDS, seq_len_multiple=SC.AUDIO_MULTIPLE, batch_size=SC.AUDIO_BATCH,
                        audio_config=types.SimpleNamespace(sample_rate=SC.AUDIO_RATE))
    listing = O.list_transcripts(SC.TRANSCRIPTS, SC.ToyTokenizer())
    assert ds.samples == listing, "transcript listing (one sample per *.trans.txt, from its last line)"
    assert (ds.bos_id, ds.eos_id, ds.pad_id) == (SC.ToyTokenizer.bos_id, SC.ToyTokenizer.eos_id, SC.ToyTokenizer.pad_id)
    # _prepare_batch on a fixed batch
    pb = SC.prepare_batch_case()
    ra, rt, rl = ds._prepare_batch([(a, list(t)) for a, t in pb])
    oa, ot, ol = O.prepare_audio_batch(pb, int(SC.AUDIO_SECONDS * SC.AUDIO_RATE), SC.AUDIO_MULTIPLE, SC.ToyTokenizer.pad_id)
    same(oa, ra, "_prepare_batch audio"), same(ot, rt, "_prepare_batch tokens"), same(ol, rl, "_prepare_batch labels")
    out = dict(prep_audio=ra, prep_tokens=rt, prep_labels=rl,
               listing_paths=np.array([p for p, _ in ds.samples]), listing_tokens=np.array([len(t) for _, t in ds.samples]))
    # __iter__: batches over more than one shuffled pass
    torch.manual_seed(SC.AUDIO_SEED)
    it = iter(ds)
    ref_b = [next(it) for _ in range(SC.AUDIO_N)]
    torch.manual_seed(SC.AUDIO_SEED)
    or_b = SC.oracle_utterance_batches(listing, clips)
    for i, (r, o) in enumerate(zip(ref_b, or_b)):
        for nm, u, v in zip(("audio", "tokens", "labels"), r, o):
            same(v, u, f"utterance packer batch {i} {nm}")
            out[f"{nm}_{i}"] = u
save("g12_librispeech", **out)

# ------------------------------------------------------------------------------------------------- G12d  LRScheduler
rows = []
for lr, n, wu, dc in SC.LR_CASES:
    sch = TU.LRScheduler(lr, n, wu, dc)
    vals = [sch.get_lr(s) for s in range(n + 3)]
    assert vals == [O.lr_at(s, lr, n, wu, dc) for s in range(n + 3)], "LR schedule"
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=torch.tensor(0.5))
    sch.set_lr(opt, n // 2)  # the tensor-lr branch of set_lr (:62-64)
    assert float(opt.param_groups[0]["lr"]) == np.float32(vals[n // 2])
    rows.append(np.array(vals, dtype=np.float64))
save("g12_lr_schedule", **{f"case_{i}": r for i, r in enumerate(rows)})

# ------------------------------------------------------------------------------------------------- G10  int8_mm_dequant, executed
kern = RMM._int8_mm_dequant_kernel.fn  # the @triton.jit function under the autotuner (subclasses/int8_mm.py:50-118)


def run_reference_kernel(A, B, sa, sb, blocks):
    M, K = A.shape
    N = B.shape[1]
    C = torch.empty(M, N, dtype=torch.float32)
    BM, BN, BK = blocks
    grid = (triton.cdiv(M, BM) * triton.cdiv(N, BN),)
    kern[grid](A, B, C, sa, sb, M, N, K, *A.stride(), *B.stride(), *C.stride(), BLOCK_M=BM, BLOCK_N=BN, BLOCK_K=BK, EVEN_K=K % 2 == 0)
    return C


out = {}
for name, (M, N, K, blocks) in SC.INT8_MM_CASES.items():
    a8, w8, sa, sb = SC.int8_mm_inputs(name, M, N, K)
    B = w8.T  # the non-contiguous [K, N] view of subclasses/int8.py:113
    assert not B.is_contiguous() and B.stride() == (1, K)
    c32 = run_reference_kernel(a8, B, sa, sb, blocks)
    same(O.int8_mm_dequant(a8, B, sa, sb), c32, f"int8_mm_dequant fp32 scales [{name}]")
    sab, sbb = sa.bfloat16(), sb.bfloat16()
    c32b = run_reference_kernel(a8, B, sab.float(), sbb.float(), blocks)  # the kernel's fp32 value under bf16 scales
    same(O.int8_mm_dequant(a8, B, sab, sbb), c32b.bfloat16(), f"int8_mm_dequant bf16 scales [{name}]")
    meta = torch.ops.torchao.int8_mm_dequant(a8.to("meta"), B.to("meta"), sab.to("meta"), sbb.to("meta"))
    assert meta.shape == (M, N) and meta.dtype is torch.bfloat16
    out[f"{name}_c_f32"], out[f"{name}_c_bf16"] = c32, c32b.bfloat16()
# a second block configuration of the autotuner's list must give the same integers (order independence)
a8, w8, sa, sb = SC.int8_mm_inputs("ragged", *SC.INT8_MM_CASES["ragged"][:3])
assert torch.equal(run_reference_kernel(a8, w8.T, sa, sb, (64, 32, 32)), out["ragged_c_f32"])
save("g10_int8_mm", **out)

print("golden fixtures written:")
for k, v in saved.items():
    print(f"  {k}.npz  {v / 1024:.1f} KiB (uncompressed)")
assert not any(f.endswith(".pyc") for _, _, fs in os.walk(REF) for f in fs), "bytecode was written under /root/reference"

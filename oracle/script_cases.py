"""Seeded inputs and oracle-side drivers of the g12 / g10 fixtures.  TEST INFRASTRUCTURE ONLY (see oracle/ref.py).

Shared by ``oracle/gen_golden_scripts.py`` (which runs the REFERENCE's training scripts on these inputs and writes the fixtures)
and by ``tests/`` (which run the oracle and the product on the same inputs and compare with the fixtures).  Nothing here is
reference code: the inputs are synthetic, the drivers call ``oracle/ref.py``.
"""
from __future__ import annotations

import os

import torch

from . import ref as O

# ---------------------------------------------------------------------------------------------- documents (M1 / M2)
DOC_LENGTHS = (17, 300, 64, 257, 5, 129, 513, 2, 256, 90, 31)  # incl. len-1 == multiple, len-1 == multiple+1, a 2-token document
PAD_SEED, PAD_BATCH, PAD_MULTIPLE, PAD_N = 11, 3, 256, 7      # 3 batches per shuffle of 11 documents -> 7 batches span 3 shuffles
PACK_SEED, PACK_SEQ, PACK_N = 5, 640, 6                       # 6 buffers span more than one shuffle


def documents() -> list[torch.Tensor]:
    return [O.randint(f"g12_doc{i}", (n,), 1, 32000) for i, n in enumerate(DOC_LENGTHS)]


def oracle_padding_batches(docs, n_batches: int = PAD_N):
    """The endless loop of _data_iter_padding (train_metamathqa.py:29-48) around O.pad_batch: every pass re-shuffles the list
    left by the previous pass with torch.randperm from the GLOBAL generator and drops the ragged tail (:37).  Call under
    torch.manual_seed(PAD_SEED)."""
    docs = list(docs)
    out = []
    while len(out) < n_batches:
        docs = [docs[i] for i in torch.randperm(len(docs))]
        for i in range(0, len(docs) - PAD_BATCH + 1, PAD_BATCH):
            out.append(O.pad_batch(docs[i : i + PAD_BATCH], PAD_MULTIPLE))
    return out[:n_batches]


def oracle_packed_buffers(docs, n_buffers: int = PACK_N):
    """The endless loop of _data_iter_document_mask (train_metamathqa.py:51-83) around O.pack_documents (state carried across
    shuffles).  Call under torch.manual_seed(PACK_SEED)."""
    docs = list(docs)
    out, st = [], {}
    while len(out) < n_buffers:
        docs = [docs[i] for i in torch.randperm(len(docs))]
        out += [tuple(t.clone() for t in b) for b in O.pack_documents(docs, PACK_SEQ, state=st)]
    return out[:n_buffers]


# ---------------------------------------------------------------------------------------------- LibriSpeech (M4 / packer)
AUDIO_RATE, AUDIO_SECONDS, AUDIO_MULTIPLE, AUDIO_BATCH, AUDIO_SEED, AUDIO_N = 100, 2.0, 8, 2, 3, 5


class ToyTokenizer:
    """Byte-level stand-in for the hub tokenizers of llama_tokenizers.py (same call signature and id attributes)."""

    bos_id, eos_id, pad_id = 1, 2, 0

    def __call__(self, text: str, add_bos: bool = False, add_eos: bool = False):
        return [self.bos_id] * add_bos + [3 + b for b in text.encode()] + [self.eos_id] * add_eos


# (directory relative to data_dir, file name, lines): one sample per FILE survives the reference's listing (its last line)
TRANSCRIPTS = [
    (f"{100 + 7 * i}/{2000 + i}", f"{100 + 7 * i}-{2000 + i}.trans.txt",
     [f"{100 + 7 * i}-{2000 + i}-{j:04d} " + " ".join(("ALPHA", "BRAVO", "CHARLIE", "DELTA", "ECHO")[: 1 + (i + j) % 5]) + "\n"
      for j in range(1 + i % 3)])
    for i in range(14)
]
_CLIP_SAMPLES = (50, 80, 120, 30, 210, 60, 95, 140, 20, 75, 110, 45, 199, 66)  # 210 > 2.0 s at 100 Hz: dropped by the packer


def clips() -> dict:
    """file name -> (waveform [channels, n] fp32, sample rate), as torchaudio.load returns; every second clip is stereo."""
    out = {}
    for i, (rel, _f, lines) in enumerate(TRANSCRIPTS):
        stem = lines[-1].split(" ", 1)[0]
        out[f"{stem}.flac"] = (O.uniform(f"g12_clip{i}", (1 + i % 2, _CLIP_SAMPLES[i]), -1.0, 1.0), AUDIO_RATE)
    return out


def write_transcripts(root) -> None:
    for rel, fname, lines in TRANSCRIPTS:
        os.makedirs(os.path.join(root, rel), exist_ok=True)
        with open(os.path.join(root, rel, fname), "w") as f:
            f.writelines(lines)


def prepare_batch_case():
    """A closed batch as __iter__ hands it to _prepare_batch: (mono audio, [bos] + tokens + [eos]); one token row is exactly a
    multiple of AUDIO_MULTIPLE long (its labels still end with one ignore_index, train_librispeech.py:80)."""
    return [(O.uniform("g12_pb0", (150,), -1.0, 1.0), [1] + list(range(10, 24)) + [2]),   # 16 tokens = 2 x 8
            (O.uniform("g12_pb1", (200,), -1.0, 1.0), [1, 40, 41, 2]),
            (O.uniform("g12_pb2", (7,), -1.0, 1.0), [1] + list(range(50, 59)) + [2])]       # 11 tokens


def oracle_utterance_batches(listing, clip_table, n_batches: int = AUDIO_N):
    """The endless loop of LibriSpeech.__iter__ (train_librispeech.py:94-98) around O.pack_utterances.  Call under
    torch.manual_seed(AUDIO_SEED)."""
    out, st = [], {}
    kw = dict(audio_duration=AUDIO_SECONDS, sample_rate=AUDIO_RATE, batch_size=AUDIO_BATCH, seq_len_multiple=AUDIO_MULTIPLE,
              bos_id=ToyTokenizer.bos_id, eos_id=ToyTokenizer.eos_id, pad_id=ToyTokenizer.pad_id)
    while len(out) < n_batches:
        out += list(O.pack_utterances(listing, lambda p: clip_table[os.path.basename(str(p))], torch.randperm(len(listing)), state=st, **kw))
    return out[:n_batches]


# ---------------------------------------------------------------------------------------------- LR schedule
LR_CASES = ((1e-3, 100, 0.1, 0.2), (3e-4, 37, 0.0, 0.0), (1.0, 10, 0.5, 0.5), (2e-5, 1000, 0.03, 0.0))

# ---------------------------------------------------------------------------------------------- int8_mm_dequant (A23)
# name -> (M, N, K, (BLOCK_M, BLOCK_N, BLOCK_K) used when the reference kernel was run)
INT8_MM_CASES = {"ragged": (70, 96, 256, (32, 32, 64)),      # M not a block multiple (masked tail rows)
                 "odd_n": (33, 50, 128, (32, 64, 32)),       # M and N ragged
                 "square": (128, 128, 512, (64, 64, 128))}


def int8_mm_inputs(name: str, M: int, N: int, K: int):
    """(A int8 [M,K], W int8 [N,K] (B = W.T), fp32 a_scale [M], fp32 b_scale [N]); scale magnitudes as quantize_int8_rowwise gives."""
    a8 = O.randint(f"mm_a_{name}", (M, K), -127, 128).to(torch.int8)
    w8 = O.randint(f"mm_b_{name}", (N, K), -127, 128).to(torch.int8)
    return a8, w8, O.uniform(f"mm_sa_{name}", (M,), 0.001, 0.02), O.uniform(f"mm_sb_{name}", (N,), 0.001, 0.02)

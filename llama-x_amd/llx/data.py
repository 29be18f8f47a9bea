"""Host-side batch construction and step schedule of the reference's training scripts (the integer contracts M1-M4
of SURVEY 8a).  Pure host code: index / label / mask tensors are bit-exact with the reference's iterators.

  pad_batch            <- _data_iter_padding batch body          (train_metamathqa.py:38-46)
  pack_documents       <- _data_iter_document_mask                (train_metamathqa.py:51-83), yields doc_ids for MaskSpec
  prepare_audio_batch  <- LibriSpeech._prepare_batch              (train_librispeech.py:68-86)
  librispeech_samples  <- LibriSpeech.__init__ transcript listing (train_librispeech.py:53-66)
  UtterancePacker      <- LibriSpeech.__iter__                    (train_librispeech.py:88-124)
  LRScheduler          <- train_utils.py:38-66
"""
from __future__ import annotations

import math
from typing import Iterable, Iterator, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor


def next_multiple(x: int, n: int) -> int:
    return -(-x // n) * n


def pad_batch(tokens_batch: Sequence[Tensor], seq_len_multiple: int = 256) -> tuple[Tensor, Tensor]:
    rows = len(tokens_batch)
    width = max(next_multiple(t.shape[0] - 1, seq_len_multiple) for t in tokens_batch)
    inputs = torch.zeros(rows, width, dtype=torch.int64)
    labels = torch.full((rows, width), -100, dtype=torch.int64)
    for r, t in enumerate(tokens_batch):
        n = t.shape[0] - 1
        inputs[r, :n], labels[r, :n] = t[:-1], t[1:]
    return inputs, labels


def padding_iterator(tokens_list: list[Tensor], batch_size: int, seq_len_multiple: int = 256, generator=None):
    """Endless shuffled batches (inputs, labels, None), as _data_iter_padding yields them (host tensors)."""
    n = len(tokens_list)
    while True:
        order = torch.randperm(n, generator=generator).tolist()
        tokens_list = [tokens_list[i] for i in order]
        for i in range(0, n - batch_size + 1, batch_size):
            yield (*pad_batch(tokens_list[i : i + batch_size], seq_len_multiple), None)


class _PackState:
    def __init__(self, seq_len: int):
        self.seq_len, self.fill, self.doc_idx = seq_len, 0, 0  # doc_idx is never reset (reference quirk, :56/:83)
        self.fresh()

    def fresh(self):
        self.inputs = torch.zeros(self.seq_len, dtype=torch.int64)
        self.labels = torch.full((self.seq_len,), -100, dtype=torch.int64)
        self.doc_ids = torch.zeros(self.seq_len, dtype=torch.int64)  # re-zeroed per buffer: the unused tail carries id 0
        self.fill = 0


def pack_documents(docs: Iterable[Tensor], seq_len: int, state: _PackState | None = None) -> Iterator[tuple[Tensor, Tensor, Tensor]]:
    """Greedy packing of shifted documents into [seq_len] buffers; yields (inputs, labels, doc_ids) on overflow."""
    st = state or _PackState(seq_len)
    for tokens in docs:
        if st.fill + len(tokens) - 1 > seq_len:
            yield st.inputs, st.labels, st.doc_ids
            st.fresh()
        n = len(tokens) - 1
        sl = slice(st.fill, st.fill + n)
        st.inputs[sl], st.labels[sl], st.doc_ids[sl] = tokens[:-1], tokens[1:], st.doc_idx
        st.fill += n
        st.doc_idx += 1


def document_mask_iterator(tokens_list: list[Tensor], seq_len: int, generator=None):
    """Endless (inputs[1,S], labels[1,S], MaskSpec(doc_ids)) batches, as _data_iter_document_mask yields them."""
    from .kernels import MaskSpec

    st = _PackState(seq_len)
    while True:
        order = torch.randperm(len(tokens_list), generator=generator).tolist()
        tokens_list = [tokens_list[i] for i in order]
        for inputs, labels, doc_ids in pack_documents(tokens_list, seq_len, st):
            yield inputs.view(1, -1), labels.view(1, -1), MaskSpec(doc_ids=doc_ids.clone())


def prepare_audio_batch(batch: Sequence[tuple[Tensor, list[int]]], audio_length: int, seq_len_multiple: int, pad_id: int):
    audios, toks = zip(*batch)
    audio = torch.stack([F.pad(a, (0, audio_length - a.shape[0])) for a in audios])
    width = math.ceil(max(len(t) for t in toks) / seq_len_multiple) * seq_len_multiple
    tokens = torch.tensor([list(t) + [pad_id] * (width - len(t)) for t in toks])
    labels = torch.tensor([list(t[1:]) + [-100] * (width - len(t) + 1) for t in toks])
    return audio, tokens, labels


def librispeech_samples(data_dir, tokenize) -> list[tuple[str, list[int]]]:
    """(audio path relative to data_dir, token ids of " <lower-cased transcript>.") per transcript FILE, sorted.

    The reference's loop body after ``for line in open(file)`` is de-indented (train_librispeech.py:55-61), so only the LAST line
    of every ``*.trans.txt`` becomes a sample.  Kept as is: a drop-in must hand the model the same sample list."""
    from pathlib import Path

    root = Path(data_dir)
    out = []
    for trans in root.glob("**/*.trans.txt"):
        rows = [ln.rstrip().split(" ", 1) for ln in open(trans)]
        if not rows:
            continue
        stem, text = rows[-1]
        out.append((str((trans.parent / f"{stem}.flac").relative_to(root)), tokenize(f" {text.lower()}.")))
    out.sort()
    return out


class UtterancePacker:
    """Endless iterator of (audio fp32 [B, duration*sr], tokens int64 [B, St], labels int64 [B, St]) host batches: utterances in
    shuffled order are concatenated until the next one would exceed ``audio_duration`` seconds; a pack's tokens are
    ``[bos] + utterance tokens... + [eos]``; clips longer than ``audio_duration`` are dropped; ``batch_size`` closed packs make a
    batch (zero-padded audio, ``pad_id`` / -100 padded tokens / labels through :func:`prepare_audio_batch`).

    ``load_audio(relative_path) -> (waveform [channels, n], sample_rate)`` is injected: the reference calls ``torchaudio.load``
    (train_librispeech.py:101), which this image does not have.  Feed it to :class:`DevicePrefetcher` for pinned, overlapped H2D."""

    def __init__(self, samples: Sequence[tuple[str, list[int]]], load_audio, *, audio_duration: float, seq_len_multiple: int, batch_size: int,
                 bos_id: int, eos_id: int, pad_id: int, sample_rate: int = 16_000, generator=None):
        self.samples, self.load_audio = list(samples), load_audio
        self.limit, self.sr, self.mult, self.bs = audio_duration, sample_rate, seq_len_multiple, batch_size
        self.bos, self.eos, self.pad = bos_id, eos_id, pad_id
        self.generator = generator

    def __iter__(self):
        closed: list[tuple[Tensor, list[int]]] = []
        clips: list[Tensor] = []
        toks: list[int] = [self.bos]
        secs = 0
        n_pad = int(self.limit * self.sr)
        while True:
            for i in torch.randperm(len(self.samples), generator=self.generator).tolist():
                path, utt_tokens = self.samples[i]
                wav, fs = self.load_audio(path)
                assert fs == self.sr, (fs, self.sr)
                mono = wav.mean(0)
                dur = mono.shape[0] / fs
                if dur > self.limit:
                    continue  # over-long clip: skipped, the open pack is left untouched
                if secs + dur > self.limit:  # close the open pack; this clip opens the next one
                    closed.append((torch.cat(clips, dim=0), toks + [self.eos]))
                    clips, toks, secs = [], [self.bos], 0
                    if len(closed) == self.bs:
                        yield prepare_audio_batch(closed, n_pad, self.mult, self.pad)
                        closed = []
                clips.append(mono)
                toks = toks + list(utt_tokens)
                secs += dur


class LRScheduler:
    """Trapezoid: linear warm-up to lr over n_steps*warmup, flat, linear decay over the last n_steps*decay."""

    def __init__(self, lr: float, n_steps: int, warmup: float, decay: float) -> None:
        self.lr, self.t1, self.t2, self.t3 = lr, int(n_steps * warmup), int(n_steps * (1 - decay)), n_steps

    def get_lr(self, step: int) -> float:
        if step < self.t1:
            return self.lr * step / self.t1
        if step < self.t2 or step >= self.t3:
            return self.lr
        return self.lr * (self.t3 - step) / (self.t3 - self.t2)

    def set_lr(self, optim: torch.optim.Optimizer, step: int):
        lr = self.get_lr(step)
        for group in optim.param_groups:
            if isinstance(group["lr"], Tensor):
                group["lr"].fill_(lr)
            else:
                group["lr"] = lr


class DevicePrefetcher:
    """Host iterator -> device batches with the H2D copy of batch i+1 overlapped with the compute of batch i.

    The reference moves every batch with a blocking ``.cuda()`` inside the iterator (train_metamathqa.py:48,65,71) or from
    DataLoader workers with pin_memory (train_librispeech.py:184-192).  Here tensors are staged in pinned host memory and
    copied on a side HIP stream; ``next()`` makes the compute stream wait on the copy event only.  Non-tensor items (None,
    MaskSpec) pass through; a MaskSpec's metadata is moved with the batch.
    """

    def __init__(self, it, device, depth: int = 2):
        self.it, self.device, self.depth = iter(it), torch.device(device), depth
        self.stream = torch.cuda.Stream(device=self.device)
        self.queue = []
        for _ in range(depth):
            self._push()

    def _to_device(self, item):
        from .kernels import MaskSpec

        if isinstance(item, Tensor):
            host = item if item.is_pinned() else item.pin_memory()
            return host.to(self.device, non_blocking=True)
        if isinstance(item, MaskSpec):
            return MaskSpec(None if item.doc_ids is None else self._to_device(item.doc_ids.to(torch.int32)),
                            None if item.prefix_len is None else self._to_device(torch.as_tensor(item.prefix_len, dtype=torch.int32)))
        return item

    def _push(self):
        try:
            batch = next(self.it)
        except StopIteration:
            return
        with torch.cuda.stream(self.stream):
            moved = tuple(self._to_device(x) for x in batch)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.queue.append((moved, ev))

    def __iter__(self):
        return self

    def __next__(self):
        if not self.queue:
            raise StopIteration
        moved, ev = self.queue.pop(0)
        torch.cuda.current_stream(self.device).wait_event(ev)
        for x in moved:
            if isinstance(x, Tensor):
                x.record_stream(torch.cuda.current_stream(self.device))
        self._push()
        return moved

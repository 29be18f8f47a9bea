"""Audio front end of LlamaAudio on HIP kernels (mel spectrogram, conv stack as implicit GEMM, prefix assembly)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from ._lib import LlxError


class MelSpectrogram(nn.Module):
    """Stand-in for torchaudio.transforms.MelSpectrogram with the arguments the reference uses (modelling/audio.py:35)."""

    def __init__(self, sample_rate=16_000, n_fft=512, win_length=400, hop_length=160, n_mels=128, norm="slaney", mel_scale="slaney"):
        super().__init__()
        assert norm == "slaney" and mel_scale == "slaney"
        self.sample_rate, self.n_fft, self.win_length, self.hop_length, self.n_mels = sample_rate, n_fft, win_length, hop_length, n_mels

    def forward(self, audio: Tensor) -> Tensor:
        raise LlxError("mel front end kernels are not built yet")


def audio_prefix_and_embed(model, audio: Tensor, tokens: Tensor):
    raise LlxError("audio front end kernels are not built yet")

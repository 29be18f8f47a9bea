"""Early-fusion audio + text Llama (API of /root/reference/modelling/audio.py:12-101) on gfx950 kernels.

Front end: log-mel spectrogram (framed, windowed 512-point real DFT evaluated directly in fp32 on the VALU, |.|^2 and the
slaney mel filterbank in the same kernel: csrc/audio.hip `mel_power_kernel`) -> log10 / clip / CMN (`logmel_cmn_kernel`) ->
Conv1d(k3,s1)+GELU -> Conv1d(k3,s2)+GELU as implicit GEMMs over a time-major, zero-padded activation buffer
(the im2col matrix of a k=3 convolution is a strided VIEW of that buffer) with bias+GELU fused in the GEMM epilogue.
The second convolution writes straight into the [audio ; text] sequence buffer, so torch.cat (audio.py:63) never runs.
"""
from typing import NamedTuple

import torch
from torch import Tensor, nn

from llx import audio_ops
from llx._lib import LlxError

from .llama import Llama, LlamaConfig, _get_hf_config, _get_hf_state_dict


class AudioConfig(NamedTuple):
    sample_rate: int = 16_000
    n_fft: int = 512
    win_length: int = 400
    hop_length: int = 160
    n_mels: int = 128


class LlamaAudio(Llama):
    def __init__(self, config: LlamaConfig, audio_config: AudioConfig = AudioConfig()):
        super().__init__(config)
        self.audio_config = audio_config
        # inspired by Whisper encoder
        self.audio_embed = nn.Sequential(
            nn.Conv1d(audio_config.n_mels, config.embed_dim, 3, 1, 1),
            nn.GELU(),
            nn.Conv1d(config.embed_dim, config.embed_dim, 3, 2, 1),
            nn.GELU(),
        )

    def build_cache(self, inference: bool = False):
        super().build_cache(inference)
        self.melspec = audio_ops.MelSpectrogram(**self.audio_config._asdict(), norm="slaney", mel_scale="slaney")

    def _embed(self, tokens: Tensor, audio: Tensor | None = None) -> tuple[Tensor, int]:
        if audio is None:
            return self.tok_embeddings(tokens), 0
        # [audio tokens ; text tokens] are produced into one sequence buffer (reference: cat at audio.py:63)
        return audio_ops.audio_prefix_and_embed(self, audio, tokens)

    def forward(self, audio: Tensor | None, tokens: Tensor, *, input_pos: Tensor | None = None, labels: Tensor | None = None,
                block_mask=None) -> Tensor:
        mask = self.causal_mask[None, None, input_pos] if input_pos is not None else None  # inference path (generate)
        x, n_audio = self._embed(tokens, audio)
        rope = self.rope[: x.shape[1]]
        x = self._run_layers(x, rope, mask=mask, input_pos=input_pos, block_mask=block_mask)
        if n_audio:
            x = x[:, n_audio:]  # remove audio embs
        return self._head(x, labels)

    @staticmethod
    def from_hf(model_id: str, **kwargs):
        audio_kwargs = {k: kwargs.pop(k) for k in list(kwargs) if k in AudioConfig._fields}
        audio_config = AudioConfig(**audio_kwargs)
        config = _get_hf_config(model_id)._replace(**kwargs)
        with torch.device("meta"):
            model = LlamaAudio(config, audio_config).eval()
        incompat_keys = model.load_state_dict(_get_hf_state_dict(model_id), strict=False, assign=True)
        if incompat_keys:
            print(incompat_keys)
        # audio_embed has no checkpoint weights: materialise from meta and initialise
        model.audio_embed.to_empty(device="cpu")
        model.audio_embed.to(dtype=model.tok_embeddings.weight.dtype)
        for m in model.audio_embed.modules():
            if isinstance(m, nn.Conv1d):
                m.reset_parameters()
        model.build_cache()
        return model

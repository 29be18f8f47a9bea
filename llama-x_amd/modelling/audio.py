"""Early-fusion audio + text Llama (API of /root/reference/modelling/audio.py:12-101) on gfx950 kernels.

Front end: log-mel spectrogram (framed, windowed 512-point real DFT evaluated directly in fp32 on the VALU, |.|^2 and the
slaney mel filterbank in the same kernel: csrc/audio.hip `mel_power_kernel`) -> log10 / clip / CMN (`logmel_cmn_kernel`) ->
Conv1d(k3,s1)+GELU -> Conv1d(k3,s2)+GELU as implicit GEMMs over a time-major, zero-padded activation buffer
(the im2col matrix of a k=3 convolution is a strided VIEW of that buffer): the bias rides in the GEMM epilogue, GELU is its own
element-wise pass because the pre-activation z is what the backward needs (the fused bias+GELU epilogue exists for inference).
The second GELU writes straight into the [audio ; text] sequence buffer, so torch.cat (audio.py:63) never runs.  With
``config.activation_checkpointing`` the conv stack is checkpointed as the reference does (audio.py:56-57): its pre-activations are
recomputed in backward instead of being kept.
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
        # two-convolution front end; the Sequential's indices are the checkpoint keys audio_embed.{0,2}.{weight,bias}
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
        """A text checkpoint (hub id or local directory, as Llama.from_hf) plus a freshly initialised audio front end.
        Keyword arguments named like AudioConfig fields configure the front end, the rest override LlamaConfig fields."""
        audio_fields = {k: kwargs.pop(k) for k in tuple(kwargs) if k in AudioConfig._fields}  # (the reference pops while iterating: audio.py:81)
        config = _get_hf_config(model_id)._replace(**kwargs)
        with torch.device("meta"):
            model = LlamaAudio(config, AudioConfig(**audio_fields)).eval()
        report = model.load_state_dict(_get_hf_state_dict(model_id), strict=False, assign=True)
        absent = [k for k in report.missing_keys if not k.startswith("audio_embed.")]
        if absent or report.unexpected_keys:
            print(report)  # the text checkpoint is expected to lack exactly the audio_embed.* entries
        # the checkpoint has no audio weights: give the meta-device convolutions real storage in the model dtype and PyTorch's default init
        dtype = model.tok_embeddings.weight.dtype
        model.audio_embed.to_empty(device="cpu").to(dtype=dtype)
        for conv in (model.audio_embed[0], model.audio_embed[2]):
            conv.reset_parameters()
        model.build_cache()  # after materialisation: buffers cannot be built under the meta device
        return model

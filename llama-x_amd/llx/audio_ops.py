"""Audio front end of LlamaAudio on HIP kernels (reference modelling/audio.py:26-36,49-63).

  MelSpectrogram        stand-in for torchaudio.transforms.MelSpectrogram with the reference's arguments (device kernel)
  audio_prefix_and_embed  log-mel/CMN -> Conv1d+GELU x2 as implicit GEMMs -> audio tokens written in front of the token
                        embeddings in ONE [B, n_audio + S_text, D] buffer (no torch.cat), with a hand-written backward
                        for the (trainable) convolution weights.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import Tensor, nn

from . import _lib as L
from . import kernels as K
from .ops import _load, _save

BF16 = torch.bfloat16


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3)
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (math.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((math.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)


def _mel_filterbank(sample_rate: int, n_fft: int, n_mels: int) -> Tensor:
    """Slaney-scale, slaney-normalised triangles, fp32 [n_fft/2+1, n_mels] (f_min 0, f_max sr/2)."""
    n_freqs = n_fft // 2 + 1
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(float(_slaney_hz_to_mel(0.0)), float(_slaney_hz_to_mel(sample_rate / 2)), n_mels + 2)
    f_pts = torch.from_numpy(_slaney_mel_to_hz(m_pts.numpy())).float()
    diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - freqs.unsqueeze(1)
    fb = torch.clamp(torch.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]), min=0.0)
    return fb * (2.0 / (f_pts[2 : n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)


class MelSpectrogram(nn.Module):
    """forward(audio fp32 [B, L]) -> fp32 [B, n_mels, 1 + L // hop] power mel spectrogram (centre / reflect padding)."""

    def __init__(self, sample_rate=16_000, n_fft=512, win_length=400, hop_length=160, n_mels=128, norm="slaney", mel_scale="slaney"):
        super().__init__()
        if norm != "slaney" or mel_scale != "slaney" or n_fft != 512:
            raise L.LlxError("MelSpectrogram: only n_fft=512 with slaney norm/scale is built (the reference's configuration)")
        self.sample_rate, self.n_fft, self.win_length, self.hop_length, self.n_mels = sample_rate, n_fft, win_length, hop_length, n_mels
        j = np.arange(n_fft, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * j / n_fft), np.sin(2 * np.pi * j / n_fft)], axis=1).astype(np.float32)
        win = torch.zeros(n_fft)
        left = (n_fft - win_length) // 2
        win[left : left + win_length] = torch.hann_window(win_length, periodic=True, dtype=torch.float32)
        self.register_buffer("twiddle", torch.from_numpy(tw), persistent=False)
        self.register_buffer("window", win, persistent=False)
        self.register_buffer("fbank", _mel_filterbank(sample_rate, n_fft, n_mels).contiguous(), persistent=False)

    def _consts(self, device):
        # model.bfloat16() would round these buffers: keep fp32 copies on the right device
        for name in ("twiddle", "window", "fbank"):
            t = getattr(self, name)
            if t.dtype is not torch.float32 or t.device != device:
                raise L.LlxError(f"MelSpectrogram.{name} must stay fp32 on {device} (build_cache() after dtype/device moves)")
        return self.twiddle, self.window, self.fbank

    def forward(self, audio: Tensor) -> Tensor:
        L.require_cuda(audio)
        assert audio.dim() == 2 and audio.dtype is torch.float32, "audio is fp32 [B, samples]"
        audio = audio.contiguous()
        B, n = audio.shape
        frames = 1 + n // self.hop_length
        tw, win, fb = self._consts(audio.device)
        mel = torch.empty(B, self.n_mels, frames, device=audio.device, dtype=torch.float32)
        L.check(L.load().llx_mel_spectrogram(L.ptr(audio), B, n, L.ptr(tw), L.ptr(win), L.ptr(fb), L.ptr(mel), frames, self.hop_length,
                                             self.n_mels, L.stream()), "llx_mel_spectrogram")
        return mel


def logmel_cmn_padded(mel: Tensor) -> Tensor:
    """mel [B, n_mels, F] -> bf16 [B, F+1, n_mels]: zero row, (F-1) rows of log10(clip(mel[..., :-1])) - mean_t, zero row."""
    B, n_mels, F = mel.shape
    feat = torch.empty(B, F + 1, n_mels, device=mel.device, dtype=BF16)
    L.check(L.load().llx_logmel_cmn(L.ptr(mel.contiguous()), L.ptr(feat), B, F, n_mels, L.stream()), "llx_logmel_cmn")
    return feat


def _gelu_fwd(z: Tensor, out: Tensor):
    L.check(L.load().llx_gelu_fwd(L.ptr(z), z.stride(0), L.ptr(out), out.stride(0), z.shape[0], z.shape[1], L.stream()), "llx_gelu_fwd")


def _gelu_bwd(dy: Tensor, z: Tensor) -> Tensor:
    dz = torch.empty(z.shape, device=z.device, dtype=BF16)
    L.check(L.load().llx_gelu_bwd(L.ptr(dy), dy.stride(0), L.ptr(z), z.stride(0), L.ptr(dz), dz.stride(0), z.shape[0], z.shape[1], L.stream()),
            "llx_gelu_bwd")
    return dz


def _reorder(w: Tensor, to_gemm: bool) -> Tensor:
    """Conv1d weight [D, C, 3] <-> GEMM weight [D, 3*C] with (tap, channel) column order."""
    if to_gemm:
        D, C, _ = w.shape
        out = torch.empty(D, 3 * C, device=w.device, dtype=BF16)
    else:
        D, C = w.shape[0], w.shape[1] // 3
        out = torch.empty(D, C, 3, device=w.device, dtype=BF16)
    L.check(L.load().llx_conv_w_reorder(L.ptr(w.contiguous()), L.ptr(out), D, C, int(to_gemm), L.stream()), "llx_conv_w_reorder")
    return out


def _col2im3(dA: Tensor, C: int, P: int, stride: int) -> Tensor:
    out = torch.empty(P, C, device=dA.device, dtype=BF16)
    L.check(L.load().llx_col2im3(L.ptr(dA), L.ptr(out), dA.shape[0], C, P, stride, L.stream()), "llx_col2im3")
    return out


class AudioPrefixFn(torch.autograd.Function):
    """x[b] = [ gelu(conv2(gelu(conv1(feat[b])))) ; tok_embeddings(tokens[b]) ]  (modelling/audio.py:49,56-63)."""

    @staticmethod
    def forward(ctx, feat_pad: Tensor, tokens: Tensor, emb: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor):
        for t in (emb, w1, b1, w2, b2):
            if t.dtype is not BF16:
                raise L.LlxError("the audio front end computes in bf16: cast the model with .bfloat16()")
        B, P1, C = feat_pad.shape
        L1 = P1 - 2
        D = w1.shape[0]
        L2 = (L1 - 1) // 2 + 1
        St = tokens.shape[1]
        w1r, w2r = _reorder(w1.detach(), True), _reorder(w2.detach(), True)
        x = torch.empty(B, L2 + St, D, device=feat_pad.device, dtype=BF16)
        z1 = torch.empty(B, L1, D, device=x.device, dtype=BF16)
        z2 = torch.empty(B, L2, D, device=x.device, dtype=BF16)
        h1 = torch.zeros(B, L1 + 2, D, device=x.device, dtype=BF16)
        for b in range(B):
            A1 = torch.as_strided(feat_pad[b], (L1, 3 * C), (C, 1))
            K.gemm_nt(A1, w1r, out=z1[b], epilogue=K.EPI_BIAS, e=b1.detach())
            _gelu_fwd(z1[b], h1[b, 1 : L1 + 1])
            A2 = torch.as_strided(h1[b], (L2, 3 * D), (2 * D, 1))
            K.gemm_nt(A2, w2r, out=z2[b], epilogue=K.EPI_BIAS, e=b2.detach())
            _gelu_fwd(z2[b], x[b, :L2])
        K.embedding_fwd(tokens, emb.detach(), out=x[:, L2:])
        _save(ctx, tokens, feat_pad, z1, z2, h1, w1r, w2r)
        ctx.dims = (B, L1, L2, C, D, emb.shape[0])
        return x

    @staticmethod
    def backward(ctx, dx: Tensor):
        tokens, feat_pad, z1, z2, h1, w1r, w2r = _load(ctx)
        B, L1, L2, C, D, vocab = ctx.dims
        dx = dx.contiguous()
        need_emb, need_w1, need_b1, need_w2, need_b2 = ctx.needs_input_grad[2:]
        demb = K.embedding_bwd(tokens, dx[:, L2:], vocab).to(BF16) if need_emb else None
        dw1 = db1 = dw2 = db2 = None
        if need_w1 or need_b1 or need_w2 or need_b2:
            w2rt = K.transpose(w2r)  # [3D, D]
            for b in range(B):
                dz2 = _gelu_bwd(dx[b, :L2], z2[b])
                A2 = torch.as_strided(h1[b], (L2, 3 * D), (2 * D, 1))
                if need_w2:
                    g = K.gemm_tn(dz2, A2)
                    dw2 = g if dw2 is None else K.add(dw2, g)
                if need_b2:
                    g = K.colsum(dz2)
                    db2 = g if db2 is None else K.add(db2, g)
                if need_w1 or need_b1:
                    dA2 = K.gemm_nt(dz2, w2rt)  # [L2, 3D]
                    dh1 = _col2im3(dA2, D, L1 + 2, 2)
                    dz1 = _gelu_bwd(dh1[1 : L1 + 1], z1[b])
                    A1 = torch.as_strided(feat_pad[b], (L1, 3 * C), (C, 1))
                    if need_w1:
                        g = K.gemm_tn(dz1, A1)
                        dw1 = g if dw1 is None else K.add(dw1, g)
                    if need_b1:
                        g = K.colsum(dz1)
                        db1 = g if db1 is None else K.add(db1, g)
            dw1 = _reorder(dw1, False) if dw1 is not None else None
            dw2 = _reorder(dw2, False) if dw2 is not None else None
        return None, None, demb, dw1, db1, dw2, db2


def audio_prefix_and_embed(model, audio: Tensor, tokens: Tensor):
    """-> (x [B, n_audio + S_text, D], n_audio) for LlamaAudio.forward."""
    L.require_cuda(audio, tokens)
    conv1, conv2 = model.audio_embed[0], model.audio_embed[2]
    if (conv1.kernel_size, conv1.stride, conv1.padding, conv2.kernel_size, conv2.stride, conv2.padding) != ((3,), (1,), (1,), (3,), (2,), (1,)):
        raise L.LlxError("audio_embed must be Conv1d(k3,s1,p1) -> GELU -> Conv1d(k3,s2,p1) -> GELU (modelling/audio.py:26-31)")
    mel = model.melspec(audio)  # (B, n_mels, frames); the last frame is dropped inside logmel_cmn_padded (audio.py:53)
    feat_pad = logmel_cmn_padded(mel)
    args = (feat_pad, tokens, model.tok_embeddings.weight, conv1.weight, conv1.bias, conv2.weight, conv2.bias)
    if model.config.activation_checkpointing and torch.is_grad_enabled() and any(t.requires_grad for t in args[2:]):
        # the reference checkpoints the conv stack (modelling/audio.py:56-57): z1 / z2 / h1 (3 x [B, 4096 .. 8192, D] at 81.92 s clips) are
        # dropped after the forward and recomputed in backward - every tensor AudioPrefixFn keeps goes through save_for_backward
        from torch.utils.checkpoint import checkpoint

        x = checkpoint(AudioPrefixFn.apply, *args, use_reentrant=False)
    else:
        x = AudioPrefixFn.apply(*args)
    return x, x.shape[1] - tokens.shape[1]

"""HiFi-GAN vocoder of the pipeline (SURVEY f-4): mel spectrogram [B, T, 64] -> waveform [B, 160 T], MI355X-native.

Mirrors ``transformers.SpeechT5HifiGan`` -- the class AudioLDM2Pipeline holds as ``self.vocoder`` and calls in
``mel_spectrogram_to_waveform`` (/root/reference/pipeline/pipeline_audioldm2.py:583-590).  Parameter / buffer names follow
the transformers module (``conv_pre``, ``upsampler.N``, ``resblocks.N.convs1.M`` / ``convs2.M``, ``conv_post``, ``mean``,
``scale``), so its state dict loads with ``load_state_dict`` (after ``remove_weight_norm``).  torch.nn modules are parameter
containers only; every convolution is ONE ``apad_gemm`` launch in its APAD_A_CONV1D mode (implicit GEMM over channels-last
activations: Conv1d with dilation, or ConvTranspose1d as a gather with the stride's holes skipped) with the vocoder's
leaky-ReLU pre-activation applied while the input tile is staged, bias + residual in the epilogue; the mean of the three
residual-block branches is ``apad_mix3``; tanh rides in the last convolution's epilogue.  No PyTorch compute fallback.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

import torch
import torch.nn as nn

from . import ops
from .unet import _Packed


@dataclass
class HifiGanConfig:
    """defaults = the vocoder of cvssp/audioldm2(-large) (SpeechT5HifiGanConfig values of that checkpoint: 64 mel bins, 16 kHz,
    x160 up-sampling) -- not verifiable offline, and every field is honoured"""
    model_in_dim: int = 64
    sampling_rate: int = 16000
    upsample_initial_channel: int = 1024
    upsample_rates: Tuple[int, ...] = (5, 4, 2, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (16, 16, 8, 4, 4)
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    leaky_relu_slope: float = 0.1
    normalize_before: bool = False


class HifiGanResidualBlock(nn.Module):
    def __init__(self, channels, kernel_size=3, dilation=(1, 3, 5), leaky_relu_slope=0.1):
        super().__init__()
        self.leaky_relu_slope, self.kernel_size, self.dilation = leaky_relu_slope, kernel_size, tuple(dilation)
        pad = lambda d: (kernel_size * d - d) // 2
        self.convs1 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, dilation=d, padding=pad(d)) for d in dilation])
        self.convs2 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, dilation=1, padding=pad(1)) for _ in dilation])
        self._pk1 = [_Packed() for _ in dilation]
        self._pk2 = [_Packed() for _ in dilation]

    def forward(self, h):
        """h [B, T, C] channels-last"""
        k, s = self.kernel_size, self.leaky_relu_slope
        for c1, c2, d, p1, p2 in zip(self.convs1, self.convs2, self.dilation, self._pk1, self._pk2):
            t = ops.conv1d(h, _pack1d(c1, p1), c1.bias, k, dilation=d, pre_slope=s)
            h = ops.conv1d(t, _pack1d(c2, p2), c2.bias, k, pre_slope=s, residual=h)
        return h


def _pack1d(conv, pk, pad_out=0):
    """Conv1d weight [Cout, Cin, k] -> [Cout (+ zero rows), k * Cin] in (tap, cin) order"""
    def f(w):
        w2 = w.permute(0, 2, 1).reshape(w.shape[0], -1)
        if pad_out > w.shape[0]:
            w2 = torch.cat([w2, w2.new_zeros(pad_out - w.shape[0], w2.shape[1])], 0)
        return w2.contiguous()
    return pk.get(conv.weight, f)


def _pack1d_t(conv, pk):
    """ConvTranspose1d weight [Cin, Cout, k] -> [Cout, k * Cin] in (tap, cin) order"""
    return pk.get(conv.weight, lambda w: w.permute(1, 2, 0).reshape(w.shape[1], -1).contiguous())


class SpeechT5HifiGan(nn.Module):
    def __init__(self, config: HifiGanConfig = None):
        super().__init__()
        cfg = self.config = config or HifiGanConfig()
        self.num_kernels, self.num_upsamples = len(cfg.resblock_kernel_sizes), len(cfg.upsample_rates)
        if self.num_kernels != 3:
            raise NotImplementedError("the branch mean is apad_mix3: three residual-block kernel sizes (HiFi-GAN V1 / AudioLDM2)")
        c0 = cfg.upsample_initial_channel
        self.conv_pre = nn.Conv1d(cfg.model_in_dim, c0, 7, padding=3)
        self.upsampler = nn.ModuleList([
            nn.ConvTranspose1d(c0 // 2 ** i, c0 // 2 ** (i + 1), k, stride=r, padding=(k - r) // 2)
            for i, (r, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes))])
        self.resblocks = nn.ModuleList()
        for i in range(self.num_upsamples):
            ch = c0 // 2 ** (i + 1)
            for k, d in zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes):
                self.resblocks.append(HifiGanResidualBlock(ch, k, d, cfg.leaky_relu_slope))
        self.conv_post = nn.Conv1d(ch, 1, 7, padding=3)
        self.register_buffer("mean", torch.zeros(cfg.model_in_dim))
        self.register_buffer("scale", torch.ones(cfg.model_in_dim))
        self._pk_pre, self._pk_post, self._pk_postb = _Packed(), _Packed(), _Packed()
        self._pk_up = [_Packed() for _ in self.upsampler]

    @torch.no_grad()
    def forward(self, spectrogram):
        """[B, T, model_in_dim] (or un-batched [T, model_in_dim]) -> [B, T * prod(upsample_rates)] (or 1-D), as
        transformers.SpeechT5HifiGan.forward"""
        cfg = self.config
        if cfg.normalize_before:
            raise NotImplementedError("normalize_before=True (SpeechT5's own vocoder) is not on the AudioLDM2 path")
        batched = spectrogram.dim() == 3
        x = spectrogram if batched else spectrogram.unsqueeze(0)
        dtype = self.conv_pre.weight.dtype
        x = x.to(dtype).contiguous()  # channels-last already: [B, T, mel bins] (the reference transposes to [B, C, T] here)
        h = ops.conv1d(x, _pack1d(self.conv_pre, self._pk_pre), self.conv_pre.bias, 7)
        for i, (up, rate, k) in enumerate(zip(self.upsampler, cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            h = ops.conv1d(h, _pack1d_t(up, self._pk_up[i]), up.bias, k, transposed_stride=rate, pre_slope=cfg.leaky_relu_slope)
            r = [self.resblocks[i * 3 + j](h) for j in range(3)]
            h = ops.mix3(r[0], r[1], r[2], 1.0 / 3.0)
        # conv_post has ONE output channel: padded to 8 zero-extended rows (the GEMM's 16-byte output rows), column 0 is the wave
        wp = _pack1d(self.conv_post, self._pk_post, pad_out=8)
        bp = self._pk_postb.get(self.conv_post.bias, lambda b: torch.cat([b, b.new_zeros(7)]).contiguous())
        y = ops.conv1d(h, wp, bp, 7, pre_slope=0.01, act="tanh")  # F.leaky_relu's default slope, then conv_post, then tanh
        wave = y[:, :, 0].contiguous()
        return wave if batched else wave[0]

"""Mel-spectrogram VAE either side of the denoise loop (SURVEY f-4 / f-3), MI355X-native.

Mirrors ``diffusers.AutoencoderKL`` as AudioLDM2Pipeline holds it (``self.vae``):
``vae.decode(latents / vae.config.scaling_factor).sample`` after the loop
(/root/reference/pipeline/pipeline_audioldm2.py:1036-1038) and ``vae.encode(mel).latent_dist.sample() * scaling_factor`` in
the training step (/root/reference/train_apadapter_v2.py:895-897).  Parameter names follow the diffusers module
(``encoder.down_blocks.N.resnets.M.*``, ``decoder.mid_block.attentions.0.to_q`` ..., ``quant_conv``, ``post_quant_conv``) so a
diffusers state dict loads with ``load_state_dict``.  torch.nn modules are parameter containers; the arithmetic is the C ABI:

* every 3x3 convolution (conv_in / resnets / up- and down-samplers / conv_out) is one ``apad_gemm`` launch in its implicit-GEMM
  conv mode over NHWC activations -- the decoder's nearest x2 up-sampling folded into the gather, the encoder's
  bottom/right-only zero padding selected by ``conv_asym_pad``, 1-channel ends widened to the 8-element vector the kernel
  stages (zero weight columns / rows);
* GroupNorm (+SiLU) is ``apad_groupnorm``; 1x1 convolutions and the attention projections are plain ``apad_gemm``;
* the mid-block attention -- ONE head of dim 512 over the 4000 latent pixels, outside ``apad_attention``'s head-dim
  envelope -- runs per sample as ``apad_gemm`` (Q.K^T) -> ``apad_softmax_rows`` -> ``apad_gemm`` (P.V against V^T written by
  the projection's APAD_OUT_VT mode);
* the posterior draw is ``apad_gaussian_sample`` (clamp, exp, scale by ``scaling_factor`` in one pass).
No PyTorch compute fallback: CPU tensors raise.
"""
import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Tuple

import torch
import torch.nn as nn

from . import ops
from .unet import _Packed, _conv3x3, _w2d


@dataclass
class VaeConfig:
    """defaults = the VAE of cvssp/audioldm2(-large) (its AutoencoderKL config: 1-channel mel, 8 latent channels, x4 reduction) --
    not verifiable offline; every field is honoured"""
    in_channels: int = 1
    out_channels: int = 1
    latent_channels: int = 8
    block_out_channels: Tuple[int, ...] = (128, 256, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.4110932946205139


EPS = 1e-6  # resnet_eps / GroupNorm eps of diffusers' Encoder / Decoder


class VaeResnetBlock(nn.Module):
    """diffusers ResnetBlock2D without a time embedding"""

    def __init__(self, cin, cout, groups):
        super().__init__()
        self.groups = groups
        self.norm1 = nn.GroupNorm(groups, cin, eps=EPS)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=EPS)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
        self._pk1, self._pk2 = _Packed(), _Packed()

    def forward(self, x, B, H, W):
        h = ops.group_norm(x, self.norm1.weight, self.norm1.bias, self.groups, EPS, silu=True)
        h, _, _ = _conv3x3(self.conv1, self._pk1, h, B, H, W)
        h = ops.group_norm(h, self.norm2.weight, self.norm2.bias, self.groups, EPS, silu=True)
        sc = x if self.conv_shortcut is None else ops.linear(x, _w2d(self.conv_shortcut), self.conv_shortcut.bias)
        out, _, _ = _conv3x3(self.conv2, self._pk2, h, B, H, W, residual=sc)
        return out


class VaeAttention(nn.Module):
    """diffusers Attention in its deprecated-attention-block configuration: GroupNorm, biased projections, one head of dim C,
    residual connection"""

    def __init__(self, channels, groups):
        super().__init__()
        self.groups, self.heads = groups, 1
        self.group_norm = nn.GroupNorm(groups, channels, eps=EPS)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])
        self._pk = _Packed()

    def _qk(self):
        ws = (self.to_q.weight, self.to_k.weight, self.to_q.bias, self.to_k.bias)
        key = tuple((id(p), p.data_ptr(), p._version, p.dtype, p.device) for p in ws)
        if getattr(self, "_qk_key", None) != key:
            self._qk_val = (torch.cat([ws[0].detach(), ws[1].detach()], 0).contiguous(), torch.cat([ws[2].detach(), ws[3].detach()], 0).contiguous())
            self._qk_key = key
        return self._qk_val

    def forward(self, x, B, H, W):
        HW, C = H * W, x.shape[-1]
        if HW % 8:
            raise ValueError(f"VaeAttention: {H}x{W} latent pixels; the score GEMM needs a multiple of 8")
        h = ops.group_norm(x, self.group_norm.weight, self.group_norm.bias, self.groups, EPS, silu=False)
        wqk, bqk = self._qk()
        qk = ops.linear(h, wqk, bqk)  # [B, HW, 2C]: q | k in one launch
        Lpad = ops.round_up(HW, 32)
        vt = torch.zeros(B, 1, C, Lpad, dtype=x.dtype, device=x.device)
        ops.linear_vt(h, self.to_v.weight, B, HW, 1, vt, bias=self.to_v.bias)
        qk2 = qk.view(B * HW, 2 * C)
        scores = torch.empty(HW, HW, dtype=x.dtype, device=x.device)  # one sample at a time: 4000 x 4000 at the AudioLDM2 size
        o = torch.empty(B, HW, C, dtype=x.dtype, device=x.device)
        for b in range(B):
            rows = qk2[b * HW:(b + 1) * HW]
            ops.gemm(rows[:, :C], rows[:, C:], M=HW, N=HW, K=C, lda=2 * C, ldw=2 * C, out=scores, ldo=HW)
            ops.softmax_rows(scores, 1.0 / math.sqrt(C), out=scores)
            ops.gemm(scores, vt[b, 0], M=HW, N=C, K=HW, lda=HW, ldw=Lpad, out=o[b], ldo=C)
        return ops.linear(o, self.to_out[0].weight, self.to_out[0].bias, residual=x)


class VaeMidBlock(nn.Module):
    def __init__(self, channels, groups):
        super().__init__()
        self.attentions = nn.ModuleList([VaeAttention(channels, groups)])
        self.resnets = nn.ModuleList([VaeResnetBlock(channels, channels, groups), VaeResnetBlock(channels, channels, groups)])

    def forward(self, x, B, H, W):
        x = self.resnets[0](x, B, H, W)
        x = self.attentions[0](x, B, H, W)
        return self.resnets[1](x, B, H, W)


class _Resample(nn.Module):
    def __init__(self, channels, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=stride, padding=1 if stride == 1 else 0)
        self._pk = _Packed()


class UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, n_layers, groups, add_upsample):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetBlock(cin if i == 0 else cout, cout, groups) for i in range(n_layers)])
        self.upsamplers = nn.ModuleList([_Resample(cout)]) if add_upsample else None

    def forward(self, x, B, H, W):
        for r in self.resnets:
            x = r(x, B, H, W)
        if self.upsamplers is not None:
            u = self.upsamplers[0]
            x, H, W = _conv3x3(u.conv, u._pk, x, B, H, W, up=(2 * H, 2 * W))  # nearest x2 folded into the conv's gather
        return x, H, W


class DownEncoderBlock(nn.Module):
    def __init__(self, cin, cout, n_layers, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetBlock(cin if i == 0 else cout, cout, groups) for i in range(n_layers)])
        self.downsamplers = nn.ModuleList([_Resample(cout, stride=2)]) if add_downsample else None

    def forward(self, x, B, H, W):
        for r in self.resnets:
            x = r(x, B, H, W)
        if self.downsamplers is not None:
            d = self.downsamplers[0]
            x, H, W = _conv3x3(d.conv, d._pk, x, B, H, W, stride=2, asym_pad=True)  # F.pad(x, (0,1,0,1)) + stride-2 conv
        return x, H, W


def _pack_wide_in(w, cin_pad):
    """[Cout, Cin, 3, 3] -> [Cout, 9 * cin_pad] (ky, kx, cin) with zero columns for the padded input channels"""
    co, ci = w.shape[:2]
    wp = w.new_zeros(co, 3, 3, cin_pad)
    wp[..., :ci] = w.permute(0, 2, 3, 1)
    return wp.reshape(co, -1).contiguous()


def _pack_wide_out(w, cout_pad):
    """[Cout, Cin, 3, 3] -> [cout_pad, 9 * Cin] with zero rows for the padded output channels"""
    co = w.shape[0]
    wp = w.new_zeros(cout_pad, w.shape[1] * 9)
    wp[:co] = w.permute(0, 2, 3, 1).reshape(co, -1)
    return wp.contiguous()


class Encoder(nn.Module):
    def __init__(self, cfg: VaeConfig):
        super().__init__()
        w, g = list(cfg.block_out_channels), cfg.norm_num_groups
        self.groups = g
        self.conv_in = nn.Conv2d(cfg.in_channels, w[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([DownEncoderBlock(w[max(i - 1, 0)], w[i], cfg.layers_per_block, g, i != len(w) - 1) for i in range(len(w))])
        self.mid_block = VaeMidBlock(w[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, w[-1], eps=EPS)
        self.conv_out = nn.Conv2d(w[-1], 2 * cfg.latent_channels, 3, padding=1)
        self._pk_in, self._pk_out = _Packed(), _Packed()

    def forward(self, x):
        """x [B, Cin, H, W] -> moments [B, h*w, 2*latent] (NHWC), h, w"""
        B, Cin, H, W = x.shape
        cp = ops.round_up(Cin, 8)
        xin = torch.zeros(B, H * W, cp, dtype=x.dtype, device=x.device)  # layout: NCHW -> NHWC, channels widened to one 16-byte vector
        xin[..., :Cin] = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin)
        h, _, _ = ops.conv3x3(xin, self._pk_in.get(self.conv_in.weight, lambda w: _pack_wide_in(w, cp)), self.conv_in.bias, B, H, W)
        for blk in self.down_blocks:
            h, H, W = blk(h, B, H, W)
        h = self.mid_block(h, B, H, W)
        h = ops.group_norm(h, self.conv_norm_out.weight, self.conv_norm_out.bias, self.groups, EPS, silu=True)
        h, _, _ = _conv3x3(self.conv_out, self._pk_out, h, B, H, W)
        return h, H, W


class Decoder(nn.Module):
    def __init__(self, cfg: VaeConfig):
        super().__init__()
        w, g = list(cfg.block_out_channels), cfg.norm_num_groups
        rw = w[::-1]
        self.groups, self.out_channels = g, cfg.out_channels
        self.conv_in = nn.Conv2d(cfg.latent_channels, w[-1], 3, padding=1)
        self.mid_block = VaeMidBlock(w[-1], g)
        self.up_blocks = nn.ModuleList([UpDecoderBlock(rw[max(i - 1, 0)], rw[i], cfg.layers_per_block + 1, g, i != len(w) - 1) for i in range(len(w))])
        self.conv_norm_out = nn.GroupNorm(g, w[0], eps=EPS)
        self.conv_out = nn.Conv2d(w[0], cfg.out_channels, 3, padding=1)
        self._pk_in, self._pk_out, self._pk_bias = _Packed(), _Packed(), _Packed()

    def forward(self, z, B, H, W):
        """z [B, h*w, latent] NHWC (after post_quant_conv) -> [B, out_channels, H', W']"""
        h, _, _ = _conv3x3(self.conv_in, self._pk_in, z, B, H, W)
        h = self.mid_block(h, B, H, W)
        for blk in self.up_blocks:
            h, H, W = blk(h, B, H, W)
        h = ops.group_norm(h, self.conv_norm_out.weight, self.conv_norm_out.bias, self.groups, EPS, silu=True)
        co = self.out_channels
        cp = ops.round_up(co, 8)  # the GEMM stores 16-byte vectors: zero weight rows up to 8 output channels
        wp = self._pk_out.get(self.conv_out.weight, lambda w: _pack_wide_out(w, cp))
        bp = self._pk_bias.get(self.conv_out.bias, lambda b: torch.cat([b, b.new_zeros(cp - co)]).contiguous())
        out, _, _ = ops.conv3x3(h, wp, bp, B, H, W)
        return out[..., :co].permute(0, 2, 1).reshape(B, co, H, W).contiguous()  # layout: NHWC -> NCHW


class DiagonalGaussianDistribution:
    """the object diffusers' ``AutoencoderKL.encode(x).latent_dist`` is: moments kept NHWC on the device, ``sample`` one launch"""

    def __init__(self, moments, B, H, W, latent):
        self._m, self._geom = moments, (B, H, W, latent)

    def _nchw(self, t):
        B, H, W, Lc = self._geom
        return t.view(B, H * W, Lc).permute(0, 2, 1).reshape(B, Lc, H, W).contiguous()

    @property
    def mean(self):
        return self._nchw(self._m[..., :self._geom[3]].contiguous())

    def mode(self):
        return self.mean

    def sample(self, generator=None, noise=None, scale=1.0):
        """mean + std * noise (x ``scale``: callers that multiply by scaling_factor next can fold it in).  ``noise`` [B, latent, H, W]
        overrides the draw (parity tests)."""
        B, H, W, Lc = self._geom
        if noise is None:
            noise = torch.randn(B, Lc, H, W, generator=generator, device=self._m.device, dtype=self._m.dtype)
        n2 = noise.to(self._m.dtype).permute(0, 2, 3, 1).reshape(B * H * W, Lc).contiguous()
        return self._nchw(ops.gaussian_sample(self._m.view(B * H * W, 2 * Lc), n2, scale))


class AutoencoderKL(nn.Module):
    def __init__(self, config: VaeConfig = None):
        super().__init__()
        cfg = self.config = config or VaeConfig()
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """z [B, latent, h, w] (= latents / scaling_factor) -> .sample [B, out_channels, 4h, 4w]"""
        if not z.is_cuda:
            raise RuntimeError("AutoencoderKL.decode: expected a GPU tensor; the HIP path has no CPU fallback")
        B, Lc, H, W = z.shape
        zl = z.to(self.post_quant_conv.weight.dtype).permute(0, 2, 3, 1).reshape(B, H * W, Lc).contiguous()
        zl = ops.linear(zl, _w2d(self.post_quant_conv), self.post_quant_conv.bias)
        out = self.decoder(zl, B, H, W)
        return SimpleNamespace(sample=out) if return_dict else (out,)

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        """x [B, in_channels, H, W] mel -> .latent_dist (DiagonalGaussianDistribution over [B, latent, H/4, W/4])"""
        if not x.is_cuda:
            raise RuntimeError("AutoencoderKL.encode: expected a GPU tensor; the HIP path has no CPU fallback")
        B = x.shape[0]
        h, H, W = self.encoder(x.to(self.quant_conv.weight.dtype))
        m = ops.linear(h, _w2d(self.quant_conv), self.quant_conv.bias)
        dist = DiagonalGaussianDistribution(m, B, H, W, self.config.latent_channels)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    def geometry_dict(self):
        c = self.config
        return dict(in_channels=c.in_channels, out_channels=c.out_channels, latent_channels=c.latent_channels,
                    block_out_channels=tuple(c.block_out_channels), layers_per_block=c.layers_per_block,
                    norm_num_groups=c.norm_num_groups)

"""Oracle: the mel-spectrogram VAE either side of the denoise loop -- ``self.vae.decode(latents / scaling_factor).sample``
(/root/reference/pipeline/pipeline_audioldm2.py:1036-1038) and, on the training side,
``vae.encode(mel).latent_dist.sample() * scaling_factor`` (/root/reference/train_apadapter_v2.py:895-897).

The class is ``diffusers.AutoencoderKL`` (diffusers==0.21.2, /root/reference/requirements.txt:1): a third-party dependency
that is NOT vendored in /root/reference and NOT installed here, and the reference holds no test or golden vector for it.
PARITY UNPINNED: this is a restatement of that release's published architecture (``Encoder`` / ``Decoder`` /
``UNetMidBlock2D`` / ``DownEncoderBlock2D`` / ``UpDecoderBlock2D`` / ``ResnetBlock2D`` / ``Attention`` with the deprecated
attention-block settings / ``DiagonalGaussianDistribution``), written as plain torch.nn.functional over a flat state dict
with the diffusers key names:

  decode(z):  post_quant_conv (1x1) -> conv_in 3x3 -> mid [resnet, attention, resnet] -> up blocks (reversed widths,
              layers_per_block + 1 resnets each, nearest x2 + conv3x3 between) -> GroupNorm(eps 1e-6) -> SiLU -> conv_out 3x3
  encode(x):  conv_in 3x3 -> down blocks (layers_per_block resnets, then a stride-2 conv3x3 over the input zero-padded by one
              row / column at the bottom / right only) -> mid -> GroupNorm -> SiLU -> conv_out 3x3 (2 x latent channels)
              -> quant_conv (1x1) -> (mean, logvar clamped to [-30, 20]); sample = mean + exp(logvar / 2) * noise
  resnet:     GroupNorm(eps 1e-6) -> SiLU -> conv3x3 -> GroupNorm -> SiLU -> conv3x3, + input (through a 1x1 conv when the
              width changes)
  attention:  GroupNorm(eps 1e-6) over [B, C, HW] -> q, k, v = Linear(C, C) (with bias) -> ONE head of dim C, softmax(q k^T /
              sqrt(C)) v -> Linear(C, C) -> + input
TEST INFRASTRUCTURE ONLY.
"""
import math

import torch
import torch.nn.functional as F


def _resnet(sd, p, x, groups, eps):
    h = F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps))
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps))
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def _attention(sd, p, x, groups, eps):
    B, C, H, W = x.shape
    h = F.group_norm(x.view(B, C, H * W), groups, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], eps).transpose(1, 2)
    q = F.linear(h, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(h, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(h, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    s = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1)
    o = F.linear(s @ v, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def _mid(sd, p, x, groups, eps):
    x = _resnet(sd, p + "resnets.0.", x, groups, eps)
    x = _attention(sd, p + "attentions.0.", x, groups, eps)
    return _resnet(sd, p + "resnets.1.", x, groups, eps)


def decode(sd, cfg, z):
    """z [B, latent, h, w] (already divided by scaling_factor) -> mel [B, out_channels, h * 2^(levels-1), w * 2^(levels-1)]"""
    g, eps, widths, lpb = cfg["norm_num_groups"], 1e-6, list(cfg["block_out_channels"]), cfg["layers_per_block"]
    x = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _mid(sd, "decoder.mid_block.", x, g, eps)
    for i in range(len(widths)):
        for j in range(lpb + 1):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}.", x, g, eps)
        if i != len(widths) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def encode_moments(sd, cfg, x):
    """x [B, in_channels, H, W] -> (mean, logvar) each [B, latent, H / 2^(levels-1), W / 2^(levels-1)]"""
    g, eps, widths, lpb = cfg["norm_num_groups"], 1e-6, list(cfg["block_out_channels"]), cfg["layers_per_block"]
    x = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(len(widths)):
        for j in range(lpb):
            x = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}.", x, g, eps)
        if i != len(widths) - 1:
            x = F.pad(x, (0, 1, 0, 1))
            x = F.conv2d(x, sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    x = _mid(sd, "encoder.mid_block.", x, g, eps)
    x = F.silu(F.group_norm(x, g, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], eps))
    x = F.conv2d(x, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    x = F.conv2d(x, sd["quant_conv.weight"], sd["quant_conv.bias"])
    mean, logvar = x.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def encode_sample(sd, cfg, x, noise):
    """latent_dist.sample() with the normal draw supplied: mean + exp(logvar / 2) * noise"""
    mean, logvar = encode_moments(sd, cfg, x)
    return mean + torch.exp(0.5 * logvar) * noise

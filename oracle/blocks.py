"""Oracle: the UNet building blocks the reference imports from diffusers==0.21.2
(/root/reference/pipeline/modeling_audioldm2.py:22-41; requirements.txt:1).

PARITY UNPINNED: diffusers is not vendored in /root/reference and not installable here, and the reference
holds no tests for these blocks.  The arithmetic is restated from diffusers 0.21.2's published source
(models/attention.py BasicTransformerBlock/FeedForward/GEGLU, models/transformer_2d.py, models/resnet.py
ResnetBlock2D/Downsample2D/Upsample2D, models/embeddings.py Timesteps/TimestepEmbedding) and anchored on
the reference's call sites cited per function.  TEST INFRASTRUCTURE ONLY.

All functions take a flat ``sd`` (state-dict with diffusers key names) and a key prefix.
"""
import math
import torch
import torch.nn.functional as F

from .attention import attn_processor_2_0, ip_attn_processor_2_0


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    """diffusers get_timestep_embedding as called by Timesteps(block_out_channels[0], flip_sin_to_cos, freq_shift)
    (modeling_audioldm2.py:317, :761)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def time_embedding(sd, t_emb):
    """TimestepEmbedding: linear_1 -> SiLU -> linear_2 (modeling_audioldm2.py:322-328, :767)."""
    h = F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    return F.linear(F.silu(h), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])


def resnet_block(sd, p, x, temb, groups=32, eps=1e-5):
    """ResnetBlock2D, time_embedding_norm='default', output_scale_factor=1 (call sites
    modeling_audioldm2.py:1032-1043, :1139, :1255-1337, :1491)."""
    h = F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    h = F.conv2d(F.silu(h), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    t = F.linear(F.silu(temb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.conv2d(F.silu(h), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def downsample(sd, p, x):
    """Downsample2D(use_conv=True, padding=1, name='op'): 3x3 stride-2 conv (modeling_audioldm2.py:1062-1068)."""
    return F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2, padding=1)


def upsample(sd, p, x, output_size=None):
    """Upsample2D(use_conv=True): nearest x2, or nearest to ``output_size`` when the UNet forwards an
    upsample size (modeling_audioldm2.py:712-720, :847-848), then 3x3 conv."""
    if output_size is None:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    else:
        x = F.interpolate(x, size=tuple(output_size), mode="nearest")
    return F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)


def _attention(sd, p, hs, ehs, mask, heads, proc):
    """Dispatch one ``Attention`` module to its processor the way diffusers' Attention.forward does
    (``self.processor(self, hidden_states, encoder_hidden_states=..., attention_mask=...)``).
    proc: None -> AttnProcessor2_0; dict(scale=, num_tokens=) -> IPAttnProcessor2_0 with weights under
    ``p + 'processor.'`` (checkpoint key scheme inference.py:54-55)."""
    w = lambda n: sd[p + n]
    if proc is None:
        return attn_processor_2_0(hs, ehs, w("to_q.weight"), w("to_k.weight"), w("to_v.weight"),
                                  w("to_out.0.weight"), w("to_out.0.bias"), heads, mask)
    return ip_attn_processor_2_0(hs, ehs, w("to_q.weight"), w("to_k.weight"), w("to_v.weight"),
                                 w("to_out.0.weight"), w("to_out.0.bias"),
                                 w("processor.to_k_ip.weight"), w("processor.to_v_ip.weight"),
                                 heads, proc["num_tokens"], proc["scale"], mask)


def basic_transformer_block(sd, p, x, ehs, emask, heads, procs):
    """BasicTransformerBlock (pre-LN x3, GEGLU FF), only_cross_attention=False.  attn2 is self-attention when
    ehs is None (double_self_attention, modeling_audioldm2.py:1058)."""
    n = F.layer_norm(x, x.shape[-1:], sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    x = _attention(sd, p + "attn1.", n, None, None, heads, procs.get(p + "attn1.processor")) + x
    n = F.layer_norm(x, x.shape[-1:], sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    x = _attention(sd, p + "attn2.", n, ehs, emask, heads, procs.get(p + "attn2.processor")) + x
    n = F.layer_norm(x, x.shape[-1:], sd[p + "norm3.weight"], sd[p + "norm3.bias"], 1e-5)
    h = F.linear(n, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"])
    a, gate = h.chunk(2, dim=-1)
    h = a * F.gelu(gate)
    return F.linear(h, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"]) + x


def transformer_2d(sd, p, x, ehs, emask, heads, n_blocks, procs, groups=32):
    """Transformer2DModel, continuous input, use_linear_projection=False: GroupNorm(eps 1e-6) -> 1x1 proj_in
    -> [B,HW,C] -> blocks -> 1x1 proj_out -> + residual (call sites modeling_audioldm2.py:1047-1058)."""
    b, c, hh, ww = x.shape
    res = x
    h = F.group_norm(x, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    h = F.conv2d(h, _as_conv(sd[p + "proj_in.weight"]), sd[p + "proj_in.bias"])
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    for i in range(n_blocks):
        h = basic_transformer_block(sd, f"{p}transformer_blocks.{i}.", h, ehs, emask, heads, procs)
    h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    h = F.conv2d(h, _as_conv(sd[p + "proj_out.weight"]), sd[p + "proj_out.bias"])
    return h + res


def _as_conv(w):
    return w if w.dim() == 4 else w[:, :, None, None]

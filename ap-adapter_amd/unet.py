"""Host-side module tree of the AudioLDM2 UNet, MI355X-native.

Mirrors the surface the reference drives (/root/reference/pipeline/modeling_audioldm2.py):
``AudioLDM2UNet2DConditionModel.forward`` (:663-873), ``attn_processors`` (:517-538), ``set_attn_processor``
(:541-574), and diffusers' parameter names, so a diffusers state dict loads with ``load_state_dict``.  torch.nn
modules are used as PARAMETER CONTAINERS only; every forward below calls the C ABI through ``ops`` and works on
token-major NHWC activations ``[B, H*W, C]`` (so Transformer2DModel's permute/reshape is free and the 3x3
convolutions run as implicit GEMMs).  There is no PyTorch compute fallback.
"""
import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import autograd as AG
from . import ops
from .processors import AttnProcessor2_0


@dataclass
class UNetConfig:
    """AudioLDM2-large geometry (inferred from copied_cross_attention/ names and shapes)."""
    in_channels: int = 8
    out_channels: int = 8
    block_out_channels: Tuple[int, ...] = (128, 256, 384, 640)
    layers_per_block: int = 2
    transformer_layers_per_block: int = 2
    cross_attention_dim: Tuple[Optional[int], ...] = (None, 768, 1024, None)
    attention_head_dim: int = 8  # used as the head COUNT (modeling_audioldm2.py:280)
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                         "CrossAttnDownBlock2D")
    up_block_types: Tuple[str, ...] = ("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D")

    def geometry_dict(self):
        return dict(in_channels=self.in_channels, out_channels=self.out_channels,
                    block_out_channels=tuple(self.block_out_channels), layers_per_block=self.layers_per_block,
                    transformer_layers_per_block=self.transformer_layers_per_block,
                    cross_attention_dim=tuple(self.cross_attention_dim), heads=self.attention_head_dim,
                    norm_num_groups=self.norm_num_groups, flip_sin_to_cos=self.flip_sin_to_cos,
                    freq_shift=self.freq_shift, down_block_types=tuple(self.down_block_types),
                    up_block_types=tuple(self.up_block_types))


class _Packed:
    """Cache of a re-laid-out weight, invalidated when the parameter object, its storage or its version changes."""

    def __init__(self):
        self.key = None
        self.val = None

    def get(self, p, fn):
        key = (id(p), p.data_ptr(), p._version, p.dtype, p.device)
        if key != self.key:
            self.val = fn(p.detach())
            self.key = key
        return self.val


def _w2d(conv_or_linear):
    w = conv_or_linear.weight
    return w.detach().reshape(w.shape[0], -1) if w.dim() == 4 else w.detach()


class Conv3x3(nn.Module):
    """Parameter container ``conv`` ([Cout,Cin,3,3], diffusers layout) + packed [Cout, 9*Cin] (ky,kx,cin) copy."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
        self.stride = stride
        self._pk = _Packed()

    def packed(self):
        return self._pk.get(self.conv.weight, lambda w: w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())


def _conv3x3(mod_conv, pk, x, B, H, W, stride=1, **kw):
    wp = pk.get(mod_conv.weight, lambda w: w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())
    return ops.conv3x3(x, wp, mod_conv.bias, B, H, W, stride=stride, **kw)


class Attention(nn.Module):
    """What diffusers==0.21.2 ``Attention`` exposes to processors on this path."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.set_processor(AttnProcessor2_0())

    def set_processor(self, processor):
        if hasattr(self, "processor") and isinstance(self.processor, nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def get_processor(self, return_deprecated_lora=False):
        return self.processor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is None:
            return None
        if attention_mask.shape[-1] != target_length:
            attention_mask = torch.nn.functional.pad(attention_mask, (0, target_length), value=0.0)
        if out_dim == 3 and attention_mask.shape[0] < batch_size * self.heads:
            attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
        return attention_mask

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, residual=None, ln=None, **kw):
        """``ln`` = (gamma, beta, eps) of the block's pre-attention LayerNorm and ``residual`` are fused into this
        package's processors; a foreign processor gets the reference call (normalised input, plain output)."""
        proc = self.processor
        if getattr(proc, "fuses_residual", False):
            return proc(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                        attention_mask=attention_mask, _residual=residual, _ln=ln, **kw)
        if ln is not None:
            hidden_states = ops.layer_norm(hidden_states, *ln)
        if attention_mask is not None:  # the reference hands processors the bias in the hidden dtype (:741-747)
            attention_mask = attention_mask.to(hidden_states.dtype)
        out = proc(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)
        return out if residual is None else out + residual  # foreign (non-HIP) processor: its own tensors


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def _hs_weights(self, ln):
        """fragment-packed GEGLU projection (norm3 folded in) of apad_hs_geglu, re-packed when a parameter is re-assigned, moved, cast or updated"""
        ps = (self.net[0].proj.weight, self.net[0].proj.bias, ln[0], ln[1], self.net[2].weight)
        key = tuple((id(p), p.data_ptr(), p._version, p.dtype, p.device) for p in ps) + (float(ln[2]),)
        if getattr(self, "_hs_key", None) != key:
            self._hs_w = ops.hs_pack_geglu(self.net[0].proj.weight, self.net[0].proj.bias, ln=ln) + (ops.hs_pack_ff2(self.net[2].weight),)
            self._hs_key = key
        return self._hs_w

    def _packed_weights(self):
        """the weight stream of apad_geglu_mlp_packed (ops.mlp_pack), re-packed when a parameter is re-assigned, moved, cast or updated"""
        ps = (self.net[0].proj.weight, self.net[0].proj.bias, self.net[2].weight)  # (a bias-free GEGLU projection: ps[1] is None)
        key = tuple(None if p is None else (id(p), p.data_ptr(), p._version, p.dtype, p.device) for p in ps)
        if getattr(self, "_mlp3_key", None) != key:
            self._mlp3_w = ops.mlp_pack(ps[0].detach(), None if ps[1] is None else ps[1].detach(), ps[2].detach())
            self._mlp3_key = key
        return self._mlp3_w

    def forward(self, x, ln):
        """x un-normalised; ln = norm3.  C in ops.MLP_C: the whole feed-forward + residual in one launch (the 4C-wide
        activation never reaches HBM); otherwise LayerNorm + GEGLU projection in one launch, then the 4C->C GEMM +
        residual."""
        if AG.on(x):  # training: un-fused, the GEGLU projection is kept for the backward
            xn, xr = AG.layer_norm_res(x, *ln)  # (the residual gradient joins the LayerNorm backward launch)
            h = AG.geglu(AG.linear(xn, self.net[0].proj.weight, self.net[0].proj.bias))
            return AG.linear(h, self.net[2].weight, self.net[2].bias, residual=xr)
        if ops.HS_FF and ops.HS_ATTN and ops.hs_rows_ok(x) and self.net[0].proj.weight.shape[0] == 8 * ops.HS_C:
            # the 64-token level: LayerNorm + GEGLU projection in ONE launch, workgroup = (sample, hidden quarter) (csrc/hsattn.hip)
            pk, bb, w2p = self._hs_weights(ln)
            h = ops.hs_geglu(x, pk, bb, ln_eps=ln[2])
            if ops.HS_FF2:
                return ops.hs_ff2(h, w2p, self.net[2].bias, x, rowstat=True)
            return ops.linear(h, self.net[2].weight, self.net[2].bias, residual=x, rowstat=True)
        if x.shape[-1] in ops.MLP_C and x.dtype in ops.FUSED_DTYPES:
            if ops.MLP_PACKED and x.numel() // x.shape[-1] >= ops.MLP_PACKED_MIN_M and x.is_contiguous():
                wp, bp = self._packed_weights()
                return ops.geglu_mlp_packed(x, wp, bp, self.net[2].bias, ln=ln)
            return ops.geglu_mlp(x, self.net[0].proj.weight, self.net[0].proj.bias, self.net[2].weight, self.net[2].bias, ln=ln)
        if (ops.MLP_PACKED and x.shape[-1] in ops.GEGLU_PACKED_C and x.dtype in ops.FUSED_DTYPES and x.is_contiguous()
                and x.numel() // x.shape[-1] >= ops.GEGLU_PACKED_MIN_M):
            # the 384-wide level at full size: LayerNorm + GEGLU projection on the 64-token register-block kernel from packed weights
            ps = (self.net[0].proj.weight, self.net[0].proj.bias)
            key = tuple(None if p is None else (id(p), p.data_ptr(), p._version, p.dtype, p.device) for p in ps)
            if getattr(self, "_geglu3_key", None) != key:
                self._geglu3_w = ops.geglu_pack(ps[0].detach(), None if ps[1] is None else ps[1].detach())
                self._geglu3_key = key
            h = ops.layernorm_geglu_packed(x, self._geglu3_w[0], self._geglu3_w[1], ln=ln)
        else:
            h = ops.fused_linear(x, self.net[0].proj.weight, self.net[0].proj.bias, ln=ln, act="geglu")
        return ops.linear(h, self.net[2].weight, self.net[2].bias, residual=x, rowstat=True)  # (the next block's norm1 folds into its q|k|v)


def cfg_expand(x, cond):
    """The classifier-free-guidance batch is [latents] * 2 against [negative | positive] conditions (pipeline_audioldm2.py:1003,
    :953-955): until the first attention that reads a condition, both halves of the UNet batch hold the SAME rows.  The denoise
    step therefore enters the UNet with the un-duplicated latents (forward_nhwc, batch_repeat) and the hidden states are
    replicated here, in front of the first conditioned attention: conv_in, the first down block, the first resnet and the whole
    unconditioned (double-self-attention) transformer of the second one run once per clip instead of twice -- same values, row for
    row, as the duplicated batch (the kernels' per-row arithmetic does not depend on the batch size)."""
    if not _CFG_EXPAND[0] or cond is None or cond.dim() != 3 or cond.shape[0] <= x.shape[0]:
        return x  # (outside a shared-prefix forward a batch mismatch is the caller's error and surfaces in the kernels' checks)
    if cond.shape[0] % x.shape[0] != 0:
        raise ValueError(f"condition batch {cond.shape[0]} is not a multiple of the hidden-state batch {x.shape[0]}")
    return x.repeat(cond.shape[0] // x.shape[0], *([1] * (x.dim() - 1)))


CFG_SHARED_PREFIX = os.environ.get("APAD_CFG_SHARED_PREFIX", "1") == "1"  # A/B switch (read once)
_CFG_EXPAND = [False]  # set by forward_nhwc for the duration of a shared-prefix forward: only then does cfg_expand replicate rows
NO_CAT = os.environ.get("APAD_NO_CAT", "1") == "1"  # up blocks: two-source GroupNorm / shortcut instead of torch.cat (A/B switch)


class BasicTransformerBlock(nn.Module):
    """diffusers BasicTransformerBlock, pre-LN x3 + GEGLU FF; attn2 is self-attention when cross_attention_dim is
    None (double_self_attention, modeling_audioldm2.py:1058)."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, ehs, emask):
        x = self.attn1(x, residual=x, ln=(self.norm1.weight, self.norm1.bias, self.norm1.eps))
        x = cfg_expand(x, ehs)
        x = self.attn2(x, encoder_hidden_states=ehs, attention_mask=emask, residual=x,
                       ln=(self.norm2.weight, self.norm2.bias, self.norm2.eps))
        return self.ff(x, (self.norm3.weight, self.norm3.bias, self.norm3.eps))


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, num_layers, cross_attention_dim, groups):
        super().__init__()
        inner = heads * dim_head
        self.groups = groups
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def _hs_packed(self, which):
        """fragment-packed proj_in / proj_out weight for apad_hs_out, re-packed when the parameter is re-assigned, moved, cast or updated"""
        w = (self.proj_in if which == "in" else self.proj_out).weight
        key = (id(w), w.data_ptr(), w._version, w.dtype, w.device)
        cache = self.__dict__.setdefault("_hs_pk", {})
        if cache.get(which, (None,))[0] != key:
            cache[which] = (key, ops.hs_pack_rows(w.detach().reshape(w.shape[0], -1))[0])
        return cache[which][1]

    def forward(self, x, ehs, emask):
        if AG.on(x):
            h = AG.group_norm(x, self.norm.weight, self.norm.bias, self.groups, self.norm.eps, False)
            h = AG.linear(h, self.proj_in.weight, self.proj_in.bias)
        else:
            h = ops.group_norm(x, self.norm.weight, self.norm.bias, self.groups, self.norm.eps, silu=False)
            hs_io = ops.HS_ATTN and ops.hs_rows_ok(h) and tuple(self.proj_in.weight.shape[:2]) == (ops.HS_C, ops.HS_C) == tuple(self.proj_out.weight.shape[:2])
            if hs_io:  # the 64-token level: the 1x1 projections on the row-tile kernel of its attention sub-layers (csrc/hsattn.hip, apad_hs_out)
                h = ops.hs_out(h, self._hs_packed("in"), self.proj_in.bias, None, rowstat=True)
            else:
                h = ops.fused_linear(h, _w2d(self.proj_in), self.proj_in.bias, rowstat=True)
        for blk in self.transformer_blocks:
            h = blk(h, ehs, emask)
        if not AG.on(h, x) and h.shape[0] == x.shape[0] and ops.HS_ATTN and ops.hs_rows_ok(h) and ops.hs_rows_ok(x) \
                and tuple(self.proj_in.weight.shape[:2]) == (ops.HS_C, ops.HS_C) == tuple(self.proj_out.weight.shape[:2]):
            return ops.hs_out(h, self._hs_packed("out"), self.proj_out.bias, x)
        if h.shape[0] != x.shape[0]:  # the CFG batch was expanded inside (cfg_expand): the residual rows are the same for both halves
            if h.dtype in ops.FUSED_DTYPES and not AG.on(h, x):  # read modulo its rows by the GEMM's epilogue: no copy
                return ops.linear(h, _w2d(self.proj_out), self.proj_out.bias, residual=x.contiguous(),
                                  residual_row_mod=x.shape[0] * x.shape[1])
            x = x.repeat(h.shape[0] // x.shape[0], 1, 1)
        if AG.on(h, x):
            return AG.linear(h, self.proj_out.weight, self.proj_out.bias, residual=x)
        return ops.fused_linear(h, _w2d(self.proj_out), self.proj_out.bias, residual=x)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_channels, groups, eps):
        super().__init__()
        self.groups = groups
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
        self._pk1, self._pk2 = _Packed(), _Packed()

    def time_proj(self, emb_act):
        """emb_act = SiLU(time embedding) [rows, 512] -> [rows, Cout]"""
        return ops.linear(emb_act, self.time_emb_proj.weight, self.time_emb_proj.bias)

    def forward(self, x, B, H, W, tproj, rows_per_group, step_ptr=None, skip=None):
        """skip: the up blocks' second input -- the resnet of torch.cat([x, skip], -1) (modeling_audioldm2.py:1488) with the
        concatenation never materialised (16-bit inference): norm1 and the 1x1 shortcut read both tensors, and a skip of the
        CFG-shared prefix (half the batch) is read modulo its batch"""
        if skip is not None:
            h = ops.group_norm2(x, skip, self.norm1.weight, self.norm1.bias, self.groups, self.norm1.eps, silu=True)
            B = h.shape[0]
            h, _, _ = _conv3x3(self.conv1, self._pk1, h, B, H, W, rowgroup_bias=tproj, rows_per_group=rows_per_group, step_ptr=step_ptr)
            h = ops.group_norm(h, self.norm2.weight, self.norm2.bias, self.groups, self.norm2.eps, silu=True)
            sc = ops.linear2(x, skip, _w2d(self.conv_shortcut), self.conv_shortcut.bias)
            out, _, _ = _conv3x3(self.conv2, self._pk2, h, B, H, W, residual=sc)
            return out
        if AG.on(x):
            h = AG.group_norm(x, self.norm1.weight, self.norm1.bias, self.groups, self.norm1.eps, True)
            h, _, _ = AG.conv3x3(h, self.conv1.weight, self.conv1.bias, B, H, W, rowgroup_bias=tproj, rows_per_group=rows_per_group)
            h = AG.group_norm(h, self.norm2.weight, self.norm2.bias, self.groups, self.norm2.eps, True)
            sc = x if self.conv_shortcut is None else AG.linear(x, self.conv_shortcut.weight, self.conv_shortcut.bias)
            out, _, _ = AG.conv3x3(h, self.conv2.weight, self.conv2.bias, B, H, W, residual=sc)
            return out
        h = ops.group_norm(x, self.norm1.weight, self.norm1.bias, self.groups, self.norm1.eps, silu=True)
        h, _, _ = _conv3x3(self.conv1, self._pk1, h, B, H, W, rowgroup_bias=tproj, rows_per_group=rows_per_group,
                           step_ptr=step_ptr)
        h = ops.group_norm(h, self.norm2.weight, self.norm2.bias, self.groups, self.norm2.eps, silu=True)
        sc = x if self.conv_shortcut is None else ops.linear(x, _w2d(self.conv_shortcut), self.conv_shortcut.bias)
        out, _, _ = _conv3x3(self.conv2, self._pk2, h, B, H, W, residual=sc)
        return out


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)
        self._pk = _Packed()

    def forward(self, x, B, H, W):
        if AG.on(x):
            return AG.conv3x3(x, self.conv.weight, self.conv.bias, B, H, W, stride=2)
        return _conv3x3(self.conv, self._pk, x, B, H, W, stride=2)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)
        self._pk = _Packed()

    def forward(self, x, B, H, W, output_size=None):
        up = tuple(output_size) if output_size is not None else (2 * H, 2 * W)
        if AG.on(x):
            return AG.conv3x3(x, self.conv.weight, self.conv.bias, B, H, W, up=up)
        return _conv3x3(self.conv, self._pk, x, B, H, W, up=up)


class _Block(nn.Module):
    """Shared body of DownBlock2D / CrossAttnDownBlock2D / UNetMidBlock2DCrossAttn / CrossAttnUpBlock2D / UpBlock2D."""

    def __init__(self, cfg: UNetConfig, resnet_io, channels, with_attn, temb_channels):
        super().__init__()
        g, eps = cfg.norm_num_groups, cfg.norm_eps
        self.cad = tuple(cfg.cross_attention_dim)
        self.resnets = nn.ModuleList([ResnetBlock2D(ci, co, temb_channels, g, eps) for ci, co in resnet_io])
        self.has_cross_attention = with_attn
        if with_attn:
            heads = cfg.attention_head_dim
            n_layers = len(resnet_io) if not isinstance(self, UNetMidBlock2DCrossAttn) else len(resnet_io) - 1
            self.attentions = nn.ModuleList([
                Transformer2DModel(heads, channels // heads, channels, cfg.transformer_layers_per_block, self.cad[j], g)
                for _ in range(n_layers) for j in range(len(self.cad))])

    def _attn_stack(self, layer, x, ehs, emask, ehs1, emask1):
        n_per = len(self.cad)
        for idx, cad in enumerate(self.cad):  # routing modeling_audioldm2.py:1140-1149
            if cad is not None and idx <= 1:
                e, m = ehs, emask
            elif cad is not None and idx > 1:
                e, m = ehs1, emask1
            else:
                e, m = None, None
            x = self.attentions[layer * n_per + idx](x, e, m)
        return x


class DownBlock(_Block):
    def __init__(self, cfg, cin, cout, with_attn, add_downsample, temb_channels):
        io = [(cin if i == 0 else cout, cout) for i in range(cfg.layers_per_block)]
        super().__init__(cfg, io, cout, with_attn, temb_channels)
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None


class UNetMidBlock2DCrossAttn(_Block):
    def __init__(self, cfg, channels, temb_channels):
        super().__init__(cfg, [(channels, channels), (channels, channels)], channels, True, temb_channels)


class UpBlock(_Block):
    def __init__(self, cfg, cin, cout, prev, with_attn, add_upsample, temb_channels):
        n = cfg.layers_per_block + 1
        io = []
        for i in range(n):
            skip = cin if i == n - 1 else cout
            rin = prev if i == 0 else cout
            io.append((rin + skip, cout))
        super().__init__(cfg, io, cout, with_attn, temb_channels)
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_upsample else None


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)


class AudioLDM2UNet2DConditionModel(nn.Module):
    def __init__(self, config: Optional[UNetConfig] = None):
        super().__init__()
        cfg = config or UNetConfig()
        self.config = cfg
        boc = cfg.block_out_channels
        temb_dim = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb_dim)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, typ in enumerate(cfg.down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            self.down_blocks.append(DownBlock(cfg, in_ch, out_ch, typ == "CrossAttnDownBlock2D", i != len(boc) - 1, temb_dim))
        self.up_blocks = nn.ModuleList()
        self.mid_block = UNetMidBlock2DCrossAttn(cfg, boc[-1], temb_dim)
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i, typ in enumerate(cfg.up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlock(cfg, in_ch, out_ch, prev, typ == "CrossAttnUpBlock2D", i != len(boc) - 1, temb_dim))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, boc[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)
        self._pk_in, self._pk_out = _Packed(), _Packed()
        self._time_tables = None
        self.low_res_streams = None  # optional (stream, stream): see forward_nhwc
        self.low_res_levels = 1      # how many of the lowest-resolution levels run on the two streams

    # ---- processor plumbing (modeling_audioldm2.py:517-574) ----
    @property
    def attn_processors(self) -> Dict[str, object]:
        procs = {}

        def rec(name, module):
            if hasattr(module, "get_processor"):
                procs[f"{name}.processor"] = module.get_processor(return_deprecated_lora=True)
            for sub, child in module.named_children():
                rec(f"{name}.{sub}", child)

        for name, module in self.named_children():
            rec(name, module)
        return procs

    def set_attn_processor(self, processor):
        count = len(self.attn_processors.keys())
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(
                f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
                f" number of attention layers: {count}. Please make sure to pass {count} processor classes.")

        def rec(name, module):
            if hasattr(module, "set_processor"):
                module.set_processor(processor if not isinstance(processor, dict) else processor.pop(f"{name}.processor"))
            for sub, child in module.named_children():
                rec(f"{name}.{sub}", child)

        for name, module in self.named_children():
            rec(name, module)

    def set_kv_cache(self, enabled: bool, clear=True):
        """Hoist the timestep-invariant K/V projections of every cross-attention out of the denoise loop.  Hoisted results
        track what they were computed from (condition version, weight identities) and recompute themselves when it
        changes; ``clear`` drops them."""
        for p in self.attn_processors.values():
            if hasattr(p, "kv_cache_enabled"):
                p.kv_cache_enabled = enabled
                if clear:
                    p.clear_kv_cache()

    def drop_kv_owner(self, owner):
        """drop the hoisted results created on behalf of ``owner`` (processors.HOIST_OWNER at their creation)"""
        for p in set(self.attn_processors.values()):
            if hasattr(p, "drop_kv_owner"):
                p.drop_kv_owner(owner)

    def refresh_kv_cache(self):
        """Recompute IN PLACE every hoisted K/V whose condition tensor or weights changed: what a cached hipGraph needs
        before it is replayed on new conditions copied into its static buffers."""
        for p in set(self.attn_processors.values()):
            if hasattr(p, "refresh_kv_cache"):
                p.refresh_kv_cache()

    # ---- time embedding ----
    def _resnets(self):
        for name, m in self.named_modules():
            if isinstance(m, ResnetBlock2D):
                yield name, m

    def _emb_act(self, timesteps_f32, dtype):
        """SiLU(time_embedding(time_proj(t))) [n, 512]: every consumer (ResnetBlock2D) applies SiLU first, so the
        activation rides in the second GEMM's epilogue."""
        cfg = self.config
        te = ops.timestep_embedding(timesteps_f32, cfg.block_out_channels[0], cfg.flip_sin_to_cos, cfg.freq_shift, dtype)
        h = ops.linear(te, self.time_embedding.linear_1.weight, self.time_embedding.linear_1.bias, act="silu")
        return ops.linear(h, self.time_embedding.linear_2.weight, self.time_embedding.linear_2.bias, act="silu")

    def precompute_time_tables(self, timesteps, step_ptr):
        """Tabulate time_emb_proj(SiLU(emb(t))) for all denoise steps: [steps, Cout] per resnet.  In the loop the
        conv epilogue reads row *step_ptr, so a captured step has no host-side timestep dependence."""
        dtype = self.conv_in.weight.dtype
        emb = self._emb_act(timesteps.float().contiguous(), dtype)
        self._time_tables = ({name: m.time_proj(emb) for name, m in self._resnets()}, step_ptr)

    def clear_time_tables(self):
        self._time_tables = None

    # ---- forward ----
    def forward(self, sample, timestep=None, encoder_hidden_states=None, class_labels=None, timestep_cond=None,
                attention_mask=None, cross_attention_kwargs=None, encoder_attention_mask=None, return_dict=True,
                encoder_hidden_states_1=None, encoder_attention_mask_1=None):
        """Reference signature (modeling_audioldm2.py:663-676); sample NCHW [B,C,H,W] -> (noise_pred NCHW,)."""
        B, Cc, H, W = sample.shape
        x = sample.permute(0, 2, 3, 1).reshape(B, H * W, Cc).contiguous()
        out = self.forward_nhwc(x, H, W, timestep, encoder_hidden_states, encoder_hidden_states_1,
                                encoder_attention_mask, encoder_attention_mask_1)
        out = out.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        if not return_dict:
            return (out,)
        return UNet2DConditionOutput(sample=out)

    def forward_nhwc(self, *args, **kwargs):
        prev = _CFG_EXPAND[0]
        try:
            return self._forward_nhwc(*args, **kwargs)
        finally:
            _CFG_EXPAND[0] = prev

    def _forward_nhwc(self, x, H, W, timestep, ehs, ehs1=None, emask=None, emask1=None, batch_repeat=1):
        """x [Bsrc, H*W, Cin] token-major; the effective batch is Bsrc*batch_repeat (CFG duplication is done by the
        first convolution's gather instead of a cat).  timestep None -> use the precomputed tables."""
        cfg = self.config
        dtype = self.conv_in.weight.dtype
        Bs = x.shape[0]
        B = Bs * batch_repeat
        # CFG duplication deferred to the first conditioned attention (cfg_expand): inference only, and only when a condition of
        # the full batch exists to trigger it (APAD_CFG_SHARED_PREFIX=0: duplicate in conv_in's gather, the A/B reference)
        # (table mode only: with explicit per-sample timesteps the two halves of the batch may sit at different steps, and the prefix
        #  would read the time projection of the first half for both)
        share = (batch_repeat > 1 and CFG_SHARED_PREFIX and not AG.on(x) and ehs is not None and ehs.dim() == 3 and ehs.shape[0] == B
                 and (ehs1 is None or ehs1.shape[0] == B) and timestep is None)
        _CFG_EXPAND[0] = share
        # mask (1 keep / 0 drop) -> additive bias [B,1,L]  (:741-747).  Built in fp32, the type apad_attention takes its
        # key bias in: the reference's cast to the hidden dtype would only add one bf16 -> fp32 conversion launch in
        # front of every masked attention site (44 per captured step)
        if emask is not None:
            emask = ((1 - emask.float()) * -10000.0).unsqueeze(1)
        if emask1 is not None:
            emask1 = ((1 - emask1.float()) * -10000.0).unsqueeze(1)
        if ehs1 is None:
            ehs1, emask1 = ehs, emask

        if timestep is None:
            if self._time_tables is None:
                raise RuntimeError("forward_nhwc(timestep=None) needs precompute_time_tables()")
            tables, step_ptr = self._time_tables
            tp = lambda name, m: (tables[name], 1 << 40)
        else:
            t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
            t = t.reshape(-1).to(device=x.device, dtype=torch.float32)
            t = t.expand(B).contiguous() if t.numel() == 1 else t.contiguous()
            emb = self._emb_act(t, dtype)
            step_ptr = None
            tp = lambda name, m: (m.time_proj(emb), None)

        B_of = lambda t_: t_.shape[0]

        def resnet(name, m, x, H, W, skip=None):
            tab, rpg = tp(name, m)
            return m(x, B_of(x), H, W, tab, rpg if rpg is not None else H * W, step_ptr, skip=skip)

        wp = self._pk_in.get(self.conv_in.weight, lambda w: w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())
        if share:
            x, _, _ = ops.conv3x3(x, wp, self.conv_in.bias, Bs, H, W)
        else:
            x, _, _ = ops.conv3x3(x, wp, self.conv_in.bias, B, H, W, src_batch_mod=(Bs if batch_repeat > 1 else 0))
        skips = [(x, H, W)]
        nb = len(cfg.block_out_channels)
        n_up = nb - 1
        fwd_up = (H % (2 ** n_up) != 0) or (W % (2 ** n_up) != 0)

        def down(i, x, H, W, skips, ehs, emask, ehs1, emask1):
            blk = self.down_blocks[i]
            for layer, rn in enumerate(blk.resnets):
                x = resnet(f"down_blocks.{i}.resnets.{layer}", rn, x, H, W)
                if blk.has_cross_attention:
                    x = blk._attn_stack(layer, x, ehs, emask, ehs1, emask1)
                skips.append((x, H, W))
            if blk.downsamplers is not None:
                x, H, W = blk.downsamplers[0](x, B_of(x), H, W)
                skips.append((x, H, W))
            return x, H, W

        def mid(x, H, W, ehs, emask, ehs1, emask1):
            mb = self.mid_block
            x = resnet("mid_block.resnets.0", mb.resnets[0], x, H, W)
            x = mb._attn_stack(0, x, ehs, emask, ehs1, emask1)
            return resnet("mid_block.resnets.1", mb.resnets[1], x, H, W)

        def up(i, x, H, W, skips, ehs, emask, ehs1, emask1):
            blk = self.up_blocks[i]
            n = len(blk.resnets)
            res = skips[-n:]
            del skips[-n:]
            final = i == nb - 1
            up_size = skips[-1][1:] if (not final and fwd_up) else None
            for layer, rn in enumerate(blk.resnets):
                s, _, _ = res.pop()
                if (NO_CAT and x.dtype in ops.FUSED_DTYPES and not AG.on(x, s) and x.shape[-1] % 64 == 0
                        and x.shape[0] >= s.shape[0] and rn.conv_shortcut is not None):
                    # (16-bit inference: both tensors go to the resnet as they are)
                    x = resnet(f"up_blocks.{i}.resnets.{layer}", rn, x.contiguous(), H, W, skip=s.contiguous())
                else:
                    if s.shape[0] != x.shape[0]:  # a skip of the shared CFG prefix (cfg_expand): one row set for both halves
                        s = s.repeat(x.shape[0] // s.shape[0], 1, 1)
                    x = torch.cat([x, s], dim=-1)  # channel concat of NHWC rows (data movement only)
                    x = resnet(f"up_blocks.{i}.resnets.{layer}", rn, x, H, W)
                if blk.has_cross_attention:
                    x = blk._attn_stack(layer, x, ehs, emask, ehs1, emask1)
            if blk.upsamplers is not None:
                x, H, W = blk.upsamplers[0](x, B_of(x), H, W, up_size)
            return x, H, W

        cond = (ehs, emask, ehs1, emask1)
        nsp = len(self.low_res_streams) if self.low_res_streams is not None else 0
        # (only while a hipGraph is being captured -- that is what the fork / join is for, and the captured allocations come from the graph's
        #  private pool; an EAGER forward on side streams was measured wrong in the fp32 mode at batch 32, round 5: it stays on one stream)
        split = (nsp > 1 and timestep is None and B % nsp == 0 and not AG.on(x) and nb >= 2 and x.is_cuda
                 and torch.cuda.is_current_stream_capturing())
        k0 = max(1, nb - self.low_res_levels) if split else nb  # first down block of the two-stream section
        for i in range(k0):
            x, H, W = down(i, x, H, W, skips, *cond)
        if split and x.shape[0] != B:  # (no conditioned attention above the section: expand here)
            x = x.repeat(B // x.shape[0], 1, 1)
        if split:
            # The lowest-resolution section (last down block(s), mid block, first up block(s): 64-token sequences) under-fills
            # the chip and is latency-bound kernel after kernel; its two batch halves are independent, so they run on two
            # streams (forked / joined inside the captured graph) and each other's launch tails overlap.
            cur = torch.cuda.current_stream()
            hb = B // nsp
            outs = [None] * nsp
            shared = list(skips)  # skips of the higher levels, consumed by the first up block
            for h, st in enumerate(self.low_res_streams):
                sl = slice(h * hb, (h + 1) * hb)
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    c_h = tuple(None if c is None else c[sl] for c in cond)
                    # (skips of the shared CFG prefix hold one row set for both halves of the batch: row b <-> b % rows)
                    part = lambda t_: t_[sl] if t_.shape[0] == B else (
                        t_[(h * hb) % t_.shape[0]:(h * hb) % t_.shape[0] + hb] if hb <= t_.shape[0] else t_)
                    sk = [(part(t_), hh, ww) for (t_, hh, ww) in shared]
                    xh, Hh, Wh = x[sl], H, W
                    for i in range(k0, nb):
                        xh, Hh, Wh = down(i, xh, Hh, Wh, sk, *c_h)
                    xh = mid(xh, Hh, Wh, *c_h)
                    for i in range(nb - k0):
                        xh, Hh, Wh = up(i, xh, Hh, Wh, sk, *c_h)
                    outs[h] = (xh, Hh, Wh, len(sk))
            for st in self.low_res_streams:
                cur.wait_stream(st)
            x = torch.cat([o_[0] for o_ in outs], dim=0)
            H, W = outs[0][1], outs[0][2]
            del skips[outs[0][3]:]
            first_up = nb - k0
        else:
            x = mid(x, H, W, *cond)
            first_up = 0
        for i in range(first_up, nb):
            x, H, W = up(i, x, H, W, skips, *cond)

        if x.shape[0] != B:  # no conditioned attention anywhere: the halves never differed
            x = x.repeat(B // x.shape[0], 1, 1)
        if AG.on(x):
            x = AG.group_norm(x, self.conv_norm_out.weight, self.conv_norm_out.bias, cfg.norm_num_groups,
                              self.conv_norm_out.eps, True)
            x, _, _ = AG.conv3x3(x, self.conv_out.weight, self.conv_out.bias, B, H, W)
            return x
        x = ops.group_norm(x, self.conv_norm_out.weight, self.conv_norm_out.bias, cfg.norm_num_groups,
                           self.conv_norm_out.eps, silu=True)
        wp = self._pk_out.get(self.conv_out.weight, lambda w: w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())
        x, _, _ = ops.conv3x3(x, wp, self.conv_out.bias, B, H, W)
        return x


@dataclass
class UNet2DConditionOutput:
    sample: torch.Tensor = None

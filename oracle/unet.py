"""Oracle: AudioLDM2UNet2DConditionModel.forward and its three cross-attention block types.

Control flow follows /root/reference/pipeline/modeling_audioldm2.py
  * forward                          :663-873   (mask->bias :741-747, time :750-773, down :803-819,
                                                 mid :822-832, up :835-862, out :865-868)
  * CrossAttnDownBlock2D.forward     :1076-1166 (per-layer routing idx<=1 -> encoder_hidden_states,
                                                 idx>1 -> encoder_hidden_states_1 :1140-1149)
  * UNetMidBlock2DCrossAttn.forward  :1255-1337
  * CrossAttnUpBlock2D.forward       :1422-1514
Block arithmetic: oracle/blocks.py (diffusers 0.21.2, PARITY UNPINNED).  TEST INFRASTRUCTURE ONLY.

``cfg`` is a plain dict:
  in_channels, out_channels, block_out_channels, layers_per_block, transformer_layers_per_block,
  cross_attention_dim (tuple per transformer slot, e.g. (None, 768, 1024, None)), heads, norm_num_groups,
  down_block_types / up_block_types ("DownBlock2D"/"CrossAttnDownBlock2D", "CrossAttnUpBlock2D"/"UpBlock2D").
"""
import torch
import torch.nn.functional as F

from . import blocks as B

# AudioLDM2-large geometry as inferred in SURVEY 8a-5 from copied_cross_attention/ names + shapes.
AUDIOLDM2_LARGE = dict(
    in_channels=8, out_channels=8, block_out_channels=(128, 256, 384, 640), layers_per_block=2,
    transformer_layers_per_block=2, cross_attention_dim=(None, 768, 1024, None), heads=8,
    norm_num_groups=32, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
)


def _route(cfg, idx, ehs, emask, ehs1, emask1):
    cad = cfg["cross_attention_dim"][idx]
    if cad is not None and idx <= 1:
        return ehs, emask
    if cad is not None and idx > 1:
        return ehs1, emask1
    return None, None


def _attn_stack(sd, cfg, p, layer, x, ehs, emask, ehs1, emask1, procs):
    n_per = len(cfg["cross_attention_dim"])
    for idx in range(n_per):
        e, m = _route(cfg, idx, ehs, emask, ehs1, emask1)
        x = B.transformer_2d(sd, f"{p}attentions.{layer * n_per + idx}.", x, e, m, cfg["heads"],
                             cfg["transformer_layers_per_block"], procs, cfg["norm_num_groups"])
    return x


def unet_forward(sd, cfg, sample, timestep, encoder_hidden_states, encoder_hidden_states_1=None,
                 encoder_attention_mask=None, encoder_attention_mask_1=None, procs=None):
    """procs: {"<attn path>.processor": dict(scale=, num_tokens=)} for the IP-adapted attn2 sites; every other
    site runs AttnProcessor2_0.  Masks are (1 keep / 0 discard) [B,L] as the pipeline passes them."""
    procs = procs or {}
    g = cfg["norm_num_groups"]
    nb = len(cfg["block_out_channels"])
    n_up = sum(1 for i in range(nb) if i != nb - 1)  # every up block but the last upsamples (:461-466)
    forward_upsample_size = any(s % (2 ** n_up) != 0 for s in sample.shape[-2:])

    emask = emask1 = None
    if encoder_attention_mask is not None:
        emask = ((1 - encoder_attention_mask.to(sample.dtype)) * -10000.0).unsqueeze(1)
    if encoder_attention_mask_1 is not None:
        emask1 = ((1 - encoder_attention_mask_1.to(sample.dtype)) * -10000.0).unsqueeze(1)
    ehs, ehs1 = encoder_hidden_states, encoder_hidden_states_1
    if ehs1 is None:  # :1091-1096
        ehs1, emask1 = ehs, emask

    t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
    t = t.reshape(-1).expand(sample.shape[0]) if t.numel() == 1 else t
    t_emb = B.timestep_embedding(t, cfg["block_out_channels"][0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    emb = B.time_embedding(sd, t_emb.to(sample.dtype))

    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips = [x]
    for i, typ in enumerate(cfg["down_block_types"]):
        p = f"down_blocks.{i}."
        for layer in range(cfg["layers_per_block"]):
            x = B.resnet_block(sd, f"{p}resnets.{layer}.", x, emb, g)
            if typ == "CrossAttnDownBlock2D":
                x = _attn_stack(sd, cfg, p, layer, x, ehs, emask, ehs1, emask1, procs)
            skips.append(x)
        if i != nb - 1:
            x = B.downsample(sd, f"{p}downsamplers.0.", x)
            skips.append(x)

    p = "mid_block."
    x = B.resnet_block(sd, p + "resnets.0.", x, emb, g)
    x = _attn_stack(sd, cfg, p, 0, x, ehs, emask, ehs1, emask1, procs)
    x = B.resnet_block(sd, p + "resnets.1.", x, emb, g)

    for i, typ in enumerate(cfg["up_block_types"]):
        p = f"up_blocks.{i}."
        n_layers = cfg["layers_per_block"] + 1
        res = skips[-n_layers:]
        skips = skips[:-n_layers]
        final = i == nb - 1
        up_size = skips[-1].shape[2:] if (not final and forward_upsample_size) else None
        for layer in range(n_layers):
            x = torch.cat([x, res.pop()], dim=1)
            x = B.resnet_block(sd, f"{p}resnets.{layer}.", x, emb, g)
            if typ == "CrossAttnUpBlock2D":
                x = _attn_stack(sd, cfg, p, layer, x, ehs, emask, ehs1, emask1, procs)
        if not final:
            x = B.upsample(sd, f"{p}upsamplers.0.", x, up_size)

    x = F.group_norm(x, g, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], 1e-5)
    return F.conv2d(F.silu(x), sd["conv_out.weight"], sd["conv_out.bias"], padding=1)

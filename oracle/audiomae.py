"""Oracle: AudioMAE ViT-B/16 encoder (no mask, no average) + the (avg+max)/2 pooling, and the audio-condition
assembly of the pipeline.

Follows
  * /root/reference/audio_encoder/models_mae.py  PatchEmbed_org :19-44, forward_encoder_no_random_mask_no_average
    :548-570, ViT-B factory :689-701 (depth 12, 12 heads, mlp 4x, LayerNorm eps 1e-6, qkv_bias)
  * /root/reference/audio_encoder/AudioMAE.py    pool :148-182, forward :190-212
  * /root/reference/pipeline/pipeline_audioldm2.py :928-956 (condition assembly)
The transformer block is timm's ``Block`` (PARITY UNPINNED: timm is not vendored / not installed); restated
as pre-LN MHSA (fused qkv Linear with bias, softmax(q k^T / sqrt(64)) v, proj) + MLP (fc1, exact-erf GELU,
fc2), cross-checked against transformers' ViTLayer in tests.  ``pos_embed`` is taken from the state dict (the
real checkpoint loads it).  pool is PINNED against torch.nn.AvgPool2d/MaxPool2d.
TEST INFRASTRUCTURE ONLY.
"""
import torch
import torch.nn.functional as F

from .attention import sdpa


def patch_embed(sd, mel):
    """mel [B,1,1024,128] -> [B,512,768]; token = 8*t_patch + f_patch (Conv2d k=s=16, flatten(2).transpose)."""
    x = F.conv2d(mel, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16)
    return x.flatten(2).transpose(1, 2)


def vit_block(sd, p, x, heads=12, eps=1e-6):
    b, n, c = x.shape
    h = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(b, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    o = sdpa(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(b, n, c)
    x = x + F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = F.layer_norm(x, (c,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


def encoder(sd, mel, depth=12, heads=12):
    """forward_encoder_no_random_mask_no_average (:548-570) -> [B,513,768]."""
    x = patch_embed(sd, mel) + sd["pos_embed"][:, 1:, :]
    cls = (sd["cls_token"] + sd["pos_embed"][:, :1, :]).expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x], dim=1)
    for i in range(depth):
        x = vit_block(sd, f"blocks.{i}.", x, heads)
    return F.layer_norm(x, (x.shape[-1],), sd["norm.weight"], sd["norm.bias"], 1e-6)


def pool(rep, time_pool, freq_pool):
    """AudioMAEConditionCTPoolRand.pool (:148-182): drop CLS, [B,768,64,8], (avg + max)/2, -> [B,La,768]."""
    r = rep[:, 1:, :].transpose(1, 2)
    b, c, _ = r.shape
    r = r.reshape(b, c, 64, 8)
    k = (time_pool, freq_pool)
    pooled = (F.avg_pool2d(r, k, k) + F.max_pool2d(r, k, k)) / 2
    return pooled.flatten(2).transpose(1, 2)


def audio_condition(sd, mel, time_pool, freq_pool, depth=12, heads=12):
    """AudioMAEConditionCTPoolRand.forward (:190-212): mel [B,1024,128] -> tokens [B,La,768]."""
    return pool(encoder(sd, mel.unsqueeze(1), depth, heads), time_pool, freq_pool)


def assemble_condition(generated_prompt_embeds, audio_tokens, uncond_audio_tokens):
    """pipeline_audioldm2.py:934-949: repeat the single audio prompt over the batch, text tokens first, audio
    after; unconditional half first.  generated_prompt_embeds [2B,8,768] = cat([negative, positive])."""
    num = generated_prompt_embeds.shape[0] // 2
    a = audio_tokens.repeat(num, 1, 1)
    u = uncond_audio_tokens.repeat(num, 1, 1)
    neg, pos = generated_prompt_embeds.chunk(2)
    return torch.cat([torch.cat([neg, u], dim=1), torch.cat([pos, a], dim=1)], dim=0)

"""AudioMAE ViT-B/16 audio encoder + (avg+max)/2 token pooling, MI355X-native.

Mirrors ``AudioMAEConditionCTPoolRand`` (/root/reference/audio_encoder/AudioMAE.py:104-212) and the encoder entry
``forward_encoder_no_random_mask_no_average`` (/root/reference/audio_encoder/models_mae.py:548-570).  Parameter names
follow the AudioMAE checkpoint (``patch_embed.proj``, ``cls_token``, ``pos_embed``, ``blocks.N.{norm1,attn.qkv,
attn.proj,norm2,mlp.fc1,mlp.fc2}``, ``norm``) so ``pretrained.pth['model']`` loads with strict=False.  Unlike the
reference (which rebuilds the model and re-reads the checkpoint on every pipeline call, pipeline_audioldm2.py:926)
the module is built once; the all-zero "unconditional" mel result is cached per pooling setting.
torch.nn modules are parameter containers only; compute goes through the C ABI.
"""
import torch
import torch.nn as nn

from . import ops


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(nn.Module):
    """timm vision_transformer.Block: pre-LN MHSA + MLP(GELU erf), LayerNorm eps 1e-6, qkv_bias."""

    def __init__(self, dim, heads, mlp_ratio=4):
        super().__init__()
        self.heads = heads
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, dim * mlp_ratio)
        self._vt = None

    def forward(self, x):
        B, N, Cc = x.shape
        h = ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        w, b = self.attn.qkv.weight, self.attn.qkv.bias
        qk = ops.linear(h, w[: 2 * Cc], b[: 2 * Cc])                       # [B,N,2C]: q | k
        Lpad = ops.round_up(N, 32)
        if self._vt is None or self._vt.shape != (B, self.heads, Cc // self.heads, Lpad) or self._vt.dtype != x.dtype:
            self._vt = torch.zeros(B, self.heads, Cc // self.heads, Lpad, dtype=x.dtype, device=x.device)
        ops.linear_vt(h, w[2 * Cc:], B, N, self.heads, self._vt, bias=b[2 * Cc:])
        o = ops.attention(qk[..., :Cc], qk[..., Cc:], self._vt, N, self.heads)
        x = ops.linear(o, self.attn.proj.weight, self.attn.proj.bias, residual=x)
        h = ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        h = ops.linear(h, self.mlp.fc1.weight, self.mlp.fc1.bias, act="gelu")
        return ops.linear(h, self.mlp.fc2.weight, self.mlp.fc2.bias, residual=x)


class _PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(1, dim, 16, stride=16)


class AudioMAEEncoder(nn.Module):
    """MaskedAutoencoderViT encoder half, mae_vit_base_patch16(in_chans=1, audio_exp=True, img_size=(1024,128))."""

    def __init__(self, dim=768, depth=12, heads=12, img_size=(1024, 128)):
        super().__init__()
        self.img_size = img_size
        n_tok = (img_size[0] // 16) * (img_size[1] // 16)
        self.patch_embed = _PatchEmbed(dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_tok + 1, dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, mel):
        """mel fp32 [B, 1024, 128] -> [B, 513, 768] (model dtype)."""
        dtype = self.norm.weight.dtype
        B = mel.shape[0]
        dim = self.norm.weight.shape[0]
        w = self.patch_embed.proj.weight.detach().reshape(dim, 256)
        n_tok = self.pos_embed.shape[1] - 1
        x = torch.empty(B, n_tok + 1, dim, dtype=dtype, device=mel.device)
        pos = self.pos_embed.detach()[0]
        # ONE implicit-GEMM launch for the whole batch (the mel patches are gathered while staging; + pos_embed rides in the
        # epilogue, row m of the batch reads pos row m % n_tok); the tokens then move behind the CLS slot (data movement)
        tok = torch.empty(B, n_tok, dim, dtype=dtype, device=mel.device)
        ops.gemm(mel, w, M=B * n_tok, N=dim, K=256, lda=0, out=tok, ldo=dim, bias=self.patch_embed.proj.bias,
                 residual=pos[1:], ldr=dim, residual_row_mod=n_tok, a_mode=ops.L.A_PATCH16,
                 conv=(mel.shape[1], mel.shape[2], 1, mel.shape[1] // 16, mel.shape[2] // 16, 1, 0, 0, 0))
        x[:, 1:, :] = tok
        x[:, 0, :] = (self.cls_token.detach()[0, 0] + pos[0])  # one 768-vector (parameter-only, data movement)
        for blk in self.blocks:
            x = blk(x)
        return ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)


class Vanilla_AudioMAE(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.model = AudioMAEEncoder(**kw)

    def forward(self, x, mask_ratio=0.0, no_mask=False, no_average=False):
        if not (no_mask and no_average):
            raise RuntimeError("only the no_mask / no_average encoder path is on the AP-adapter hot path")
        return self.model(x.squeeze(1) if x.dim() == 4 else x)


class AudioMAEConditionCTPoolRand(nn.Module):
    """forward(mel[B,1024,128], time_pool, freq_pool) -> [tokens [B,La,768], ones [B,La]] (AudioMAE.py:190-212)."""

    def __init__(self, time_pooling_factors=(1, 2, 4, 8), freq_pooling_factors=(1, 2, 4, 8), eval_time_pooling=8,
                 eval_freq_pooling=8, mask_ratio=0.0, regularization=False, no_audiomae_mask=True,
                 no_audiomae_average=True, **encoder_kw):
        super().__init__()
        if regularization:
            raise NotImplementedError("regularization=True is never used by the reference drivers")
        self.eval_time_pooling, self.eval_freq_pooling = eval_time_pooling, eval_freq_pooling
        self.audiomae = Vanilla_AudioMAE(**encoder_kw)
        for p in self.parameters():
            p.requires_grad = False

    def pool(self, representation, time_pool=None, freq_pool=None, out_dtype=None):
        assert representation.size(-1) == 768
        return ops.audiomae_pool(representation, time_pool, freq_pool, out_dtype)

    def forward(self, batch, time_pool=None, freq_pool=None):
        assert batch.size(-2) == 1024 and batch.size(-1) == 128
        device = self.audiomae.model.norm.weight.device
        mel = batch.to(device=device, dtype=torch.float32).contiguous()
        rep = self.audiomae(mel, no_mask=True, no_average=True)
        tokens = self.pool(rep, time_pool, freq_pool)
        return [tokens, torch.ones(tokens.shape[0], tokens.shape[1], device=device)]

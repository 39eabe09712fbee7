"""Prompt encoding of the pipeline (SURVEY f-4; /root/reference/pipeline/pipeline_audioldm2.py:231-270 ``generate_language_model``,
:272-580 ``encode_prompt``), MI355X-native: the CLAP text branch, the T5 encoder, ``AudioLDM2ProjectionModel``
(/root/reference/pipeline/modeling_audioldm2.py:82-145) and the GPT-2 hidden-state generation loop.

The three encoders are third-party transformers models in the reference (``ClapModel.get_text_features``, ``T5EncoderModel``,
``GPT2Model``); the classes here keep the transformers parameter names, so their state dicts load with ``load_state_dict``, and
run the arithmetic through the C ABI in the fp32 precision mode (they run once per prompt; 16-bit checkpoints are up-cast at
load): ``apad_gather_rows`` (embeddings), ``apad_gemm`` with the ReLU / gelu_new / gated-gelu_new epilogues, ``apad_layernorm`` /
``apad_rmsnorm`` (T5LayerNorm, F.normalize), and attention as ``apad_attention`` where the bias is a key mask (CLAP) or, where
it is a full matrix (T5's relative-position bias, GPT-2's causal mask), per (sample, head) ``apad_gemm`` Q.K^T ->
``apad_softmax_rows`` (+ fp32 bias) -> ``apad_gemm`` P.V.  Tokenizers are host-side vocabulary look-ups with no files offline:
the entry points take token ids.  No PyTorch compute fallback: CPU tensors raise.

Sequences are padded to a multiple of 8 tokens (masked keys) so that every score matrix meets the GEMM's vector width.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import ops

NEG = float("-inf")


def _need_gpu_f32(mod, what):
    p = next(mod.parameters())
    if not p.is_cuda:
        raise RuntimeError(f"{what}: expected a GPU module; the HIP path has no CPU fallback")
    if p.dtype != torch.float32:
        raise RuntimeError(f"{what}: the prompt encoders run in the fp32 precision mode; call .float() on the module")


def _pad_len(L):
    return ops.round_up(max(L, 8), 8)


def _pad_tokens(x, mask, Lp):
    """x [B, L, C], mask [B, L] (1 = token) -> zero / masked padding up to Lp tokens"""
    B, L, C = x.shape
    if L == Lp:
        return x.contiguous(), mask
    xp = x.new_zeros(B, Lp, C)
    xp[:, :L] = x
    mp = mask.new_zeros(B, Lp)
    mp[:, :L] = mask
    return xp, mp


def _matrix_attention(q, k, vt, heads, bias, scale):
    """softmax(scale * q k^T + bias) v per (sample, head): q, k [B, L, H*d]; vt [B, H, d, Lpad]; bias fp32 [B, H, L, L]"""
    B, L, C = q.shape
    d = C // heads
    o = torch.empty_like(q)
    q2, k2, o2 = q.view(B * L, C), k.view(B * L, C), o.view(B * L, C)
    s = torch.empty(L, L, dtype=q.dtype, device=q.device)
    for b in range(B):
        rows = slice(b * L, (b + 1) * L)
        for h in range(heads):
            cols = slice(h * d, (h + 1) * d)
            ops.gemm(q2[rows, cols], k2[rows, cols], M=L, N=L, K=d, lda=C, ldw=C, out=s, ldo=L)
            ops.softmax_rows(s, scale, out=s, bias=bias[b, h])
            ops.gemm(s, vt[b, h], M=L, N=d, K=L, lda=L, ldw=vt.shape[-1], out=o2[rows, cols], ldo=C)
    return o


def _vt(x, w, bias, B, L, heads):
    vt = torch.zeros(B, heads, w.shape[0] // heads, ops.round_up(L, 32), dtype=x.dtype, device=x.device)
    ops.linear_vt(x, w, B, L, heads, vt, bias=bias)
    return vt


# ------------------------------------------------------------------------------------------------------------------------------
# CLAP text branch (transformers ClapTextModelWithProjection / ClapModel.get_text_features)
# ------------------------------------------------------------------------------------------------------------------------------
@dataclass
class ClapTextConfig:
    """defaults = transformers.ClapTextConfig (laion/clap-htsat-unfused text tower, the one cvssp/audioldm2 ships)"""
    vocab_size: int = 50265
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 514
    type_vocab_size: int = 1
    layer_norm_eps: float = 1e-12
    pad_token_id: int = 1
    projection_dim: int = 512
    model_type: str = "clap"


class _BertSelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)


class _BertSelfOutput(nn.Module):
    def __init__(self, cin, c, eps):
        super().__init__()
        self.dense = nn.Linear(cin, c)
        self.LayerNorm = nn.LayerNorm(c, eps=eps)


class _BertAttention(nn.Module):
    def __init__(self, c, eps):
        super().__init__()
        self.self = _BertSelfAttention(c)
        self.output = _BertSelfOutput(c, c, eps)


class _BertIntermediate(nn.Module):
    def __init__(self, c, ci):
        super().__init__()
        self.dense = nn.Linear(c, ci)


class ClapTextLayer(nn.Module):
    """post-LN BERT layer"""

    def __init__(self, cfg):
        super().__init__()
        c = cfg.hidden_size
        self.heads = cfg.num_attention_heads
        self.attention = _BertAttention(c, cfg.layer_norm_eps)
        self.intermediate = _BertIntermediate(c, cfg.intermediate_size)
        self.output = _BertSelfOutput(cfg.intermediate_size, c, cfg.layer_norm_eps)

    def forward(self, x, key_bias, B, L):
        a, H = self.attention, self.heads
        q = ops.linear(x, a.self.query.weight, a.self.query.bias)
        k = ops.linear(x, a.self.key.weight, a.self.key.bias)
        vt = _vt(x, a.self.value.weight, a.self.value.bias, B, L, H)
        d = q.shape[-1] // H
        if d in (32, 48, 64, 80):  # apad_attention's head dims: the padding mask is a per-key bias
            ctx = ops.attention(q, k, vt, L, H, key_bias=key_bias)
        else:
            ctx = _matrix_attention(q, k, vt, H, key_bias[:, None, None, :].expand(B, H, L, L).contiguous(), 1.0 / math.sqrt(d))
        ln = a.output.LayerNorm
        x = ops.layer_norm(ops.linear(ctx, a.output.dense.weight, a.output.dense.bias, residual=x), ln.weight, ln.bias, ln.eps)
        h = ops.linear(x, self.intermediate.dense.weight, self.intermediate.dense.bias, act="gelu")
        ln = self.output.LayerNorm
        return ops.layer_norm(ops.linear(h, self.output.dense.weight, self.output.dense.bias, residual=x), ln.weight, ln.bias, ln.eps)


class _ClapEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        c = cfg.hidden_size
        self.word_embeddings = nn.Embedding(cfg.vocab_size, c, padding_idx=cfg.pad_token_id)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, c, padding_idx=cfg.pad_token_id)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, c)
        self.LayerNorm = nn.LayerNorm(c, eps=cfg.layer_norm_eps)
        self.padding_idx = cfg.pad_token_id


class _Encoder(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.layer = nn.ModuleList(layers)


class _Pooler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c, c)


class ClapTextModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _ClapEmbeddings(cfg)
        self.encoder = _Encoder([ClapTextLayer(cfg) for _ in range(cfg.num_hidden_layers)])
        self.pooler = _Pooler(cfg.hidden_size)


class _ClapProjection(nn.Module):
    def __init__(self, c, p):
        super().__init__()
        self.linear1, self.linear2 = nn.Linear(c, p), nn.Linear(p, p)


class ClapTextModelWithProjection(nn.Module):
    """``get_text_features(input_ids, attention_mask)`` of transformers' ClapModel (its ``text_model.*`` / ``text_projection.*``
    parameters; load a ClapModel state dict with strict=False)"""

    def __init__(self, config: ClapTextConfig = None):
        super().__init__()
        cfg = self.config = config or ClapTextConfig()
        self.text_model = ClapTextModel(cfg)
        self.text_projection = _ClapProjection(cfg.hidden_size, cfg.projection_dim)

    @torch.no_grad()
    def get_text_features(self, input_ids, attention_mask=None):
        """[B, L] token ids -> L2-normalised text features [B, projection_dim]"""
        _need_gpu_f32(self, "ClapTextModelWithProjection")
        if not input_ids.is_cuda:
            raise RuntimeError("ClapTextModelWithProjection: expected GPU token ids; the HIP path has no CPU fallback")
        e = self.text_model.embeddings
        B, L0 = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        Lp = _pad_len(L0)
        ids = input_ids.new_full((B, Lp), e.padding_idx)
        ids[:, :L0] = input_ids
        mask = attention_mask.new_zeros(B, Lp)
        mask[:, :L0] = attention_mask
        # RoBERTa position ids (index arithmetic): padding_idx + running count of non-padding tokens, padding stays at padding_idx
        ne = ids.ne(e.padding_idx).long()
        pos = torch.cumsum(ne, dim=1) * ne + e.padding_idx
        x = ops.mix3(ops.embedding(e.word_embeddings.weight, ids), ops.embedding(e.position_embeddings.weight, pos),
                     ops.embedding(e.token_type_embeddings.weight, torch.zeros_like(ids)), 1.0)
        x = ops.layer_norm(x, e.LayerNorm.weight, e.LayerNorm.bias, e.LayerNorm.eps)
        key_bias = torch.zeros(B, Lp, dtype=torch.float32, device=x.device).masked_fill_(mask == 0, -1e30)  # finite: apad_attention adds it to live scores
        for layer in self.text_model.encoder.layer:
            x = layer(x, key_bias, B, Lp)
        first = x[:, 0].contiguous()
        pooled = ops.linear(first, self.text_model.pooler.dense.weight, self.text_model.pooler.dense.bias, act="tanh")
        p = self.text_projection
        feat = ops.linear(ops.linear(pooled, p.linear1.weight, p.linear1.bias, act="relu"), p.linear2.weight, p.linear2.bias)
        return ops.l2_normalize(feat)


# ------------------------------------------------------------------------------------------------------------------------------
# T5 encoder (transformers T5EncoderModel, gated-gelu feed-forward: google/flan-t5-large in cvssp/audioldm2)
# ------------------------------------------------------------------------------------------------------------------------------
@dataclass
class T5Config:
    vocab_size: int = 32128
    d_model: int = 1024
    d_kv: int = 64
    d_ff: int = 2816
    num_layers: int = 24
    num_heads: int = 16
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6
    model_type: str = "t5"


class _T5Norm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))


class _T5Attention(nn.Module):
    def __init__(self, cfg, has_bias):
        super().__init__()
        inner = cfg.num_heads * cfg.d_kv
        self.q, self.k, self.v = (nn.Linear(cfg.d_model, inner, bias=False) for _ in range(3))
        self.o = nn.Linear(inner, cfg.d_model, bias=False)
        if has_bias:
            self.relative_attention_bias = nn.Embedding(cfg.relative_attention_num_buckets, cfg.num_heads)


class _T5LayerSelfAttention(nn.Module):
    def __init__(self, cfg, has_bias):
        super().__init__()
        self.SelfAttention = _T5Attention(cfg, has_bias)
        self.layer_norm = _T5Norm(cfg.d_model)


class _T5Dense(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.wi_0 = nn.Linear(cfg.d_model, cfg.d_ff, bias=False)  # gate (gelu_new)
        self.wi_1 = nn.Linear(cfg.d_model, cfg.d_ff, bias=False)  # value
        self.wo = nn.Linear(cfg.d_ff, cfg.d_model, bias=False)


class _T5LayerFF(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.DenseReluDense = _T5Dense(cfg)
        self.layer_norm = _T5Norm(cfg.d_model)


class T5Block(nn.Module):
    def __init__(self, cfg, has_bias):
        super().__init__()
        self.layer = nn.ModuleList([_T5LayerSelfAttention(cfg, has_bias), _T5LayerFF(cfg)])
        self.heads, self.eps = cfg.num_heads, cfg.layer_norm_epsilon
        self._gate_key, self._gate = None, None

    def _value_gate(self):
        d = self.layer[1].DenseReluDense
        key = tuple((id(p), p.data_ptr(), p._version) for p in (d.wi_0.weight, d.wi_1.weight))
        if key != self._gate_key:  # rows value | gate, the layout of the gated epilogue
            self._gate, self._gate_key = torch.cat([d.wi_1.weight.detach(), d.wi_0.weight.detach()], 0).contiguous(), key
        return self._gate

    def forward(self, x, bias, B, L):
        sa, H = self.layer[0], self.heads
        a = sa.SelfAttention
        h = ops.rms_norm(x, sa.layer_norm.weight, self.eps)
        q, k = ops.linear(h, a.q.weight), ops.linear(h, a.k.weight)
        vt = _vt(h, a.v.weight, None, B, L, H)
        x = ops.linear(_matrix_attention(q, k, vt, H, bias, 1.0), a.o.weight, residual=x)  # T5 does not scale the scores
        ff = self.layer[1]
        h = ops.rms_norm(x, ff.layer_norm.weight, self.eps)
        return ops.linear(ops.linear(h, self._value_gate(), act="geglu_tanh"), ff.DenseReluDense.wo.weight, residual=x)


class _T5Stack(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.d_model)
        self.block = nn.ModuleList([T5Block(cfg, i == 0) for i in range(cfg.num_layers)])
        self.final_layer_norm = _T5Norm(cfg.d_model)


def t5_relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """transformers T5Attention._relative_position_bucket, bidirectional (host-side index arithmetic, same float32 steps)"""
    num_buckets //= 2
    buckets = (relative_position > 0).to(torch.long) * num_buckets
    rp = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rp, large)


class T5EncoderModel(nn.Module):
    def __init__(self, config: T5Config = None):
        super().__init__()
        cfg = self.config = config or T5Config()
        self.shared = nn.Embedding(cfg.vocab_size, cfg.d_model)
        self.encoder = _T5Stack(cfg)
        # transformers ties the two tables (T5EncoderModel._tied_weights_keys) and an HF checkpoint stores only ``shared.weight``:
        # ONE parameter under both names, so a state dict with either key (or both) fills the table the forward reads
        self.encoder.embed_tokens.weight = self.shared.weight

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        a, b = prefix + "shared.weight", prefix + "encoder.embed_tokens.weight"
        if a in state_dict and b not in state_dict:
            state_dict[b] = state_dict[a]
        elif b in state_dict and a not in state_dict:
            state_dict[a] = state_dict[b]
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None):
        """[B, L] token ids -> (last_hidden_state [B, L, d_model],)"""
        _need_gpu_f32(self, "T5EncoderModel")
        if not input_ids.is_cuda:
            raise RuntimeError("T5EncoderModel: expected GPU token ids; the HIP path has no CPU fallback")
        cfg, enc = self.config, self.encoder
        B, L0 = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        Lp = _pad_len(L0)
        ids = input_ids.new_zeros(B, Lp)
        ids[:, :L0] = input_ids
        mask = attention_mask.new_zeros(B, Lp)
        mask[:, :L0] = attention_mask
        x = ops.embedding(enc.embed_tokens.weight, ids)
        # relative-position bias of block 0, shared by every block: bucket indices on the host, the table look-up on the device
        pos = torch.arange(Lp)
        buckets = t5_relative_position_bucket(pos[None, :] - pos[:, None], cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)
        table = enc.block[0].layer[0].SelfAttention.relative_attention_bias.weight
        pb = ops.embedding(table, buckets.reshape(-1).to(x.device)).view(Lp, Lp, cfg.num_heads).permute(2, 0, 1)  # [H, L, L]
        bias = pb[None].expand(B, -1, -1, -1).masked_fill((mask == 0)[:, None, None, :], NEG).contiguous()
        for blk in enc.block:
            x = blk(x, bias, B, Lp)
        x = ops.rms_norm(x, enc.final_layer_norm.weight, cfg.layer_norm_epsilon)
        return (x[:, :L0].contiguous(),)


# ------------------------------------------------------------------------------------------------------------------------------
# GPT-2 (transformers GPT2Model driven with inputs_embeds, as generate_language_model does)
# ------------------------------------------------------------------------------------------------------------------------------
@dataclass
class GPT2Config:
    n_positions: int = 1024
    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    layer_norm_epsilon: float = 1e-5
    vocab_size: int = 50257
    max_new_tokens: int = 8
    model_type: str = "gpt2"


class Conv1D(nn.Module):
    """transformers' Conv1D: a Linear whose weight is stored [in, out]"""

    def __init__(self, nf, nx):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(nx, nf).normal_(std=0.02))
        self.bias = nn.Parameter(torch.zeros(nf))
        self._key, self._wt = None, None

    def wt(self):
        key = (id(self.weight), self.weight.data_ptr(), self.weight._version)
        if key != self._key:
            self._wt, self._key = self.weight.detach().t().contiguous(), key
        return self._wt


class _GPT2Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c_attn, self.c_proj = Conv1D(3 * c, c), Conv1D(c, c)


class _GPT2MLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.c_fc, self.c_proj = Conv1D(4 * c, c), Conv1D(c, 4 * c)


class GPT2Block(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        c, eps = cfg.n_embd, cfg.layer_norm_epsilon
        self.heads = cfg.n_head
        self.ln_1, self.attn, self.ln_2, self.mlp = nn.LayerNorm(c, eps=eps), _GPT2Attention(c), nn.LayerNorm(c, eps=eps), _GPT2MLP(c)

    def forward(self, x, bias, B, L):
        c, H = x.shape[-1], self.heads
        h = ops.layer_norm(x, self.ln_1.weight, self.ln_1.bias, self.ln_1.eps)
        w, b = self.attn.c_attn.wt(), self.attn.c_attn.bias
        q, k = ops.linear(h, w[:c], b[:c]), ops.linear(h, w[c:2 * c], b[c:2 * c])
        vt = _vt(h, w[2 * c:], b[2 * c:], B, L, H)
        o = _matrix_attention(q, k, vt, H, bias, 1.0 / math.sqrt(c // H))
        x = ops.linear(o, self.attn.c_proj.wt(), self.attn.c_proj.bias, residual=x)
        h = ops.layer_norm(x, self.ln_2.weight, self.ln_2.bias, self.ln_2.eps)
        h = ops.linear(h, self.mlp.c_fc.wt(), self.mlp.c_fc.bias, act="gelu_tanh")
        return ops.linear(h, self.mlp.c_proj.wt(), self.mlp.c_proj.bias, residual=x)


class GPT2Model(nn.Module):
    def __init__(self, config: GPT2Config = None):
        super().__init__()
        cfg = self.config = config or GPT2Config()
        self.wte = nn.Embedding(cfg.vocab_size, cfg.n_embd)
        self.wpe = nn.Embedding(cfg.n_positions, cfg.n_embd)
        self.h = nn.ModuleList([GPT2Block(cfg) for _ in range(cfg.n_layer)])
        self.ln_f = nn.LayerNorm(cfg.n_embd, eps=cfg.layer_norm_epsilon)

    @torch.no_grad()
    def forward(self, inputs_embeds, attention_mask=None):
        """inputs_embeds [B, L, n_embd] -> last_hidden_state [B, L, n_embd] (causal; attention_mask [B, L] masks keys)"""
        _need_gpu_f32(self, "GPT2Model")
        if not inputs_embeds.is_cuda:
            raise RuntimeError("GPT2Model: expected GPU inputs; the HIP path has no CPU fallback")
        B, L0, c = inputs_embeds.shape
        if attention_mask is None:
            attention_mask = torch.ones(B, L0, dtype=torch.long, device=inputs_embeds.device)
        Lp = _pad_len(L0)
        x, mask = _pad_tokens(inputs_embeds.float(), attention_mask.to(inputs_embeds.device), Lp)
        pos = ops.embedding(self.wpe.weight, torch.arange(Lp, device=x.device).expand(B, Lp).contiguous())
        x = ops.mix3(x, pos, torch.zeros_like(x), 1.0)
        causal = torch.ones(Lp, Lp, dtype=torch.bool, device=x.device).tril()
        allowed = causal[None, None] & (mask != 0)[:, None, None, :]
        bias = torch.zeros(B, 1, Lp, Lp, dtype=torch.float32, device=x.device).masked_fill_(~allowed, NEG)
        bias = bias.expand(B, self.config.n_head, Lp, Lp).contiguous()
        for blk in self.h:
            x = blk(x, bias, B, Lp)
        x = ops.layer_norm(x, self.ln_f.weight, self.ln_f.bias, self.ln_f.eps)
        return x[:, :L0].contiguous()


# ------------------------------------------------------------------------------------------------------------------------------
# AudioLDM2ProjectionModel + generation loop + encode_prompt
# ------------------------------------------------------------------------------------------------------------------------------
def add_special_tokens(hidden_states, attention_mask, sos_token, eos_token):
    """modeling_audioldm2.py:45-60: learned SOS / EOS vectors either side of the sequence (layout only)"""
    B = hidden_states.shape[0]
    if attention_mask is not None:
        ones = attention_mask.new_ones((B, 1))
        attention_mask = torch.cat([ones, attention_mask, ones], dim=-1)
    sos = sos_token.expand(B, 1, -1)
    eos = eos_token.expand(B, 1, -1)
    return torch.cat([sos, hidden_states, eos], dim=1), attention_mask


class AudioLDM2ProjectionModel(nn.Module):
    """modeling_audioldm2.py:82-145 (parameter names kept: projection, projection_1, sos_embed, eos_embed, sos_embed_1, eos_embed_1)"""

    def __init__(self, text_encoder_dim=512, text_encoder_1_dim=1024, langauge_model_dim=768):
        super().__init__()
        self.projection = nn.Linear(text_encoder_dim, langauge_model_dim)
        self.projection_1 = nn.Linear(text_encoder_1_dim, langauge_model_dim)
        self.sos_embed = nn.Parameter(torch.ones(langauge_model_dim))
        self.eos_embed = nn.Parameter(torch.ones(langauge_model_dim))
        self.sos_embed_1 = nn.Parameter(torch.ones(langauge_model_dim))
        self.eos_embed_1 = nn.Parameter(torch.ones(langauge_model_dim))

    @torch.no_grad()
    def forward(self, hidden_states=None, hidden_states_1=None, attention_mask=None, attention_mask_1=None):
        _need_gpu_f32(self, "AudioLDM2ProjectionModel")
        hs = ops.linear(hidden_states.float().contiguous(), self.projection.weight, self.projection.bias)
        hs, attention_mask = add_special_tokens(hs, attention_mask, self.sos_embed, self.eos_embed)
        hs1 = ops.linear(hidden_states_1.float().contiguous(), self.projection_1.weight, self.projection_1.bias)
        hs1, attention_mask_1 = add_special_tokens(hs1, attention_mask_1, self.sos_embed_1, self.eos_embed_1)
        hs = torch.cat([hs, hs1], dim=1)
        if attention_mask is None and attention_mask_1 is not None:
            attention_mask = attention_mask_1.new_ones(hs.shape[0], hs.shape[1] - attention_mask_1.shape[1])
        elif attention_mask is not None and attention_mask_1 is None:
            attention_mask_1 = attention_mask.new_ones(hs1.shape[:2])
        if attention_mask is not None and attention_mask_1 is not None:
            attention_mask = torch.cat([attention_mask, attention_mask_1], dim=-1)
        return hs, attention_mask


@torch.no_grad()
def generate_language_model(language_model, inputs_embeds, attention_mask=None, max_new_tokens=8):
    """pipeline_audioldm2.py:231-270: hidden-state auto-regression -- each step appends the LAST hidden state of the language model
    to its own input and extends the attention mask by one (no cache: the sequence is re-run, as the reference does when
    ``use_cache`` is unset); returns the ``max_new_tokens`` generated vectors."""
    max_new_tokens = max_new_tokens if max_new_tokens is not None else language_model.config.max_new_tokens
    for _ in range(max_new_tokens):
        hidden = language_model(inputs_embeds, attention_mask=attention_mask)
        inputs_embeds = torch.cat([inputs_embeds, hidden[:, -1:, :]], dim=1)
        if attention_mask is not None:
            attention_mask = torch.cat([attention_mask, attention_mask.new_ones((attention_mask.shape[0], 1))], dim=-1)
    return inputs_embeds[:, -max_new_tokens:, :]


class PromptEncoder(nn.Module):
    """The text side of ``AudioLDM2Pipeline.encode_prompt`` (:272-580) from token ids: CLAP text feature -> one "token",
    T5 hidden states, projection + special tokens, 8 generated GPT-2 vectors.  ``encode`` returns what ``encode_prompt`` returns
    for one half of the CFG batch: (prompt_embeds = T5 states, attention_mask, generated_prompt_embeds)."""

    def __init__(self, text_encoder: ClapTextModelWithProjection, text_encoder_2: T5EncoderModel, projection_model: AudioLDM2ProjectionModel,
                 language_model: GPT2Model):
        super().__init__()
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self.projection_model, self.language_model = projection_model, language_model

    @torch.no_grad()
    def encode(self, clap_input_ids, clap_attention_mask, t5_input_ids, t5_attention_mask, max_new_tokens=None):
        B = clap_input_ids.shape[0]
        clap = self.text_encoder.get_text_features(clap_input_ids, attention_mask=clap_attention_mask)[:, None, :]  # (:401-406)
        clap_mask = t5_attention_mask.new_ones((B, 1))
        t5 = self.text_encoder_2(t5_input_ids, attention_mask=t5_attention_mask)[0]
        projected, projected_mask = self.projection_model(hidden_states=clap, hidden_states_1=t5, attention_mask=clap_mask,
                                                          attention_mask_1=t5_attention_mask)
        generated = generate_language_model(self.language_model, projected, attention_mask=projected_mask, max_new_tokens=max_new_tokens)
        return t5, t5_attention_mask, generated

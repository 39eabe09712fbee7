"""Oracle: the two attention processors the reference wires into the UNet.

Follows /root/reference/APadapter/ap_adapter/attention_processor.py
  * AttnProcessor2_0.__call__   :214-294
  * IPAttnProcessor2_0.__call__ :347-470
restated as pure functions over explicit weights (no ``attn`` object).  Softmax scale is SDPA's default
1/sqrt(head_dim) (:274, :429, :443).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import math
import torch
import torch.nn.functional as F


def sdpa(q, k, v, bias=None):
    """softmax(q k^T / sqrt(d) + bias) v on [B,h,N,d] x [B,h,L,d]; L == 0 yields zeros (reference :443 with
    an empty audio segment)."""
    if k.shape[-2] == 0:
        return torch.zeros_like(q)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if bias is not None:
        s = s + bias
    return torch.matmul(torch.softmax(s, dim=-1), v)


def _heads(x, h):
    b, n, c = x.shape
    return x.view(b, n, h, c // h).transpose(1, 2)


def _merge(x):
    b, h, n, d = x.shape
    return x.transpose(1, 2).reshape(b, n, h * d)


def prepare_attention_mask(mask, target_length, batch_size, heads):
    """diffusers Attention.prepare_attention_mask (out_dim=3) as used at reference :245-249 / :378-382:
    pad (never taken on this path), repeat_interleave over heads, view to [B,h,1,L]."""
    if mask is None:
        return None
    if mask.shape[-1] != target_length:
        mask = F.pad(mask, (0, target_length), value=0.0)
    if mask.shape[0] < batch_size * heads:
        mask = mask.repeat_interleave(heads, dim=0)
    return mask.view(batch_size, heads, -1, mask.shape[-1])


def _tokens_of(hs):
    """the 4-D entry of both processors (:232-236, :363-367): [B, C, H, W] -> [B, H*W, C]; returns (tokens, shape or None)"""
    if hs.dim() == 4:
        b, c, h, w = hs.shape
        return hs.view(b, c, h * w).transpose(1, 2), (b, c, h, w)
    return hs, None


def _image_of(out, shape4):
    """... and back (:290-291, :461-462): transpose(-1, -2).reshape(B, C, H, W)"""
    return out if shape4 is None else out.transpose(-1, -2).reshape(*shape4)


def attn_processor_2_0(hs, ehs, wq, wk, wv, wo, bo, heads, attention_mask=None):
    """AttnProcessor2_0 (:214-294).  hs [B,N,C] (or [B,C,H,W]); ehs [B,L,X] or None (self-attention); attention_mask is the
    additive bias [B,1,L] the UNet builds (modeling_audioldm2.py:741-747) or None."""
    hs, shape4 = _tokens_of(hs)
    b = hs.shape[0]
    src = hs if ehs is None else ehs
    bias = prepare_attention_mask(attention_mask, src.shape[1], b, heads)
    q = _heads(F.linear(hs, wq), heads)
    k = _heads(F.linear(src, wk), heads)
    v = _heads(F.linear(src, wv), heads)
    o = _merge(sdpa(q, k, v, bias))
    return _image_of(F.linear(o, wo, bo), shape4)


def ip_attn_processor_2_0(hs, ehs, wq, wk, wv, wo, bo, wk_ip, wv_ip, heads, num_tokens, scale,
                          attention_mask=None):
    """IPAttnProcessor2_0 (:347-470): text branch over ehs[:, :num_tokens] with the frozen to_k/to_v (:400-431),
    audio branch over ehs[:, num_tokens:] with to_k_ip/to_v_ip (:435-445), blend text + scale*audio (:454),
    to_out[0] with bias (:457).  The masked branch keeps only the first q_len mask columns (:424-428)."""
    hs, shape4 = _tokens_of(hs)
    b, n, _ = hs.shape
    if ehs.dim() < 3:
        ehs = ehs.unsqueeze(0)
    bias = prepare_attention_mask(attention_mask, ehs.shape[1], b, heads)
    q = _heads(F.linear(hs, wq), heads)
    txt, aud = ehs[:, :num_tokens, :], ehs[:, num_tokens:, :]
    k = _heads(F.linear(txt, wk), heads)
    v = _heads(F.linear(txt, wv), heads)
    if bias is not None:
        bias = bias.split(bias.shape[2], dim=3)[0]
    o_t = _merge(sdpa(q, k, v, bias))
    k_a = _heads(F.linear(aud, wk_ip), heads)
    v_a = _heads(F.linear(aud, wv_ip), heads)
    o_a = _merge(sdpa(q, k_a, v_a, None))
    return _image_of(F.linear(o_t + scale * o_a, wo, bo), shape4)

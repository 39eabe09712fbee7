"""The plug-in operators: MI355X-native counterparts of the reference's attention processors.

Same class names, constructor arguments, attributes and call signature as
/root/reference/APadapter/ap_adapter/attention_processor.py (``AttnProcessor2_0`` :199-294,
``IPAttnProcessor2_0`` :297-470) so that the reference wiring loop (inference.py:22-59) works unchanged:

    proc = IPAttnProcessor2_0(hidden_size=C, name=name, cross_attention_dim=768, scale=ap_scale, num_tokens=8)
    proc.to_k_ip.weight = torch.nn.Parameter(state_dict[name + ".to_k_ip.weight"].half())   # re-assignment is honoured
    unet.set_attn_processor({...})

The arithmetic runs in libapadapter_hip.so (apad_gemm + apad_attention); there is no PyTorch fallback.
Weights are read through the attributes at call time.  The timestep-invariant K/V projections can be hoisted
out of the denoise loop with ``kv_cache_enabled`` (the pipeline switches it on and clears it per call); with it
off every call recomputes them exactly like the reference.
"""
import torch
import torch.nn as nn

from . import autograd as AG
from . import ops

_vt_pool = {}

# Route the C = 256 / 8-head cross-attention sub-layers (<= 64 keys per segment) through the single-launch
# apad_fused_cross_attention kernel (LN + to_q + decoupled attention + to_out + residual).  APAD_FUSED_XATTN=0 selects the
# three-kernel chain it replaces (A/B measurements, tests).
import os as _os
USE_FUSED_XATTN = _os.environ.get("APAD_FUSED_XATTN", "1") == "1"
# Self-attention behind the row-panel projection: scale the to_q rows by log2(e) / sqrt(d) once (cached with the stacked weight) so
# that apad_attention takes q as the base-2 exponent operand (q_prescaled)
PRESCALE_Q = True


def _fused_xattn_ok(attn, hidden_states, residual, ln, L1, L2=0, masked=False):
    C_ = hidden_states.shape[-1]
    return (USE_FUSED_XATTN and residual is hidden_states and ln is not None and C_ == ops.XATTN_C and attn.heads == ops.XATTN_HEADS
            and tuple(attn.to_q.weight.shape) == (C_, C_) and ops.xattn_lengths_ok(L1, L2, masked)
            and hidden_states.is_contiguous() and hidden_states.dtype in ops.FUSED_DTYPES)


USE_XATTN_ROWS = True  # the 384-wide level's single-launch route (follows USE_FUSED_XATTN; module attribute only)


def _xrows_ok(attn, hidden_states, residual, ln, L1, L2=0):
    """the 384-wide cross-attention sub-layers with <= 64 keys per segment: apad_cross_attention_rows (masked or not)"""
    C_ = hidden_states.shape[-1]
    return (USE_FUSED_XATTN and USE_XATTN_ROWS and residual is hidden_states and ln is not None and ops.xrows_ok(C_, attn.heads, L1, L2)
            and tuple(attn.to_q.weight.shape) == (C_, C_) and hidden_states.is_contiguous() and hidden_states.dtype in ops.FUSED_DTYPES)


def _xrows_weights(attn):
    """fragment-packed to_q / to_out[0] of apad_cross_attention_rows, cached like _xattn_weights"""
    key = _pkey(attn.to_q.weight, attn.to_out[0].weight)
    if getattr(attn, "_xrows_key", None) != key:
        attn._xrows_w = (ops.xrows_pack_weight(attn.to_q.weight), ops.xrows_pack_weight(attn.to_out[0].weight))
        attn._xrows_key = key
    return attn._xrows_w


def _xattn_weights(attn, ln):
    """(fragment-packed to_q with the block's LayerNorm folded in, its fold vectors, fragment-packed to_out[0]) of the fused kernel, cached on
    the Attention module and re-packed when a parameter (the LayerNorm's included) is re-assigned, moved, cast or updated in place"""
    key = (_pkey(attn.to_q.weight, attn.to_out[0].weight, ln[0], ln[1]), float(ln[2]))
    if getattr(attn, "_xattn_key", None) != key:
        wq_p, q_fold = ops.xattn_pack_weight(attn.to_q.weight.detach(), ln)
        attn._xattn_w = (wq_p, q_fold, ops.xattn_pack_weight(attn.to_out[0].weight.detach()))
        attn._xattn_key = key
    return attn._xattn_w


def _hs_route(attn, hidden_states, residual, ln):
    """the 64-token level's two-launch route (csrc/hsattn.hip): [B, <= 64, 640], 8 heads, a bias-free square to_q; entered with or without
    the block's LayerNorm / residual (the reference call has neither)"""
    return (ops.hs_ok(hidden_states, attn.heads, attn.to_q.weight.shape[0]) and attn.to_q.bias is None and attn.to_out[0].weight.shape[0] == ops.HS_C
            and (residual is None or (residual.shape == hidden_states.shape and residual.is_contiguous())))


def _rows_kv_route(attn, hidden_states, residual, ln, L1, L2=0):
    """the site's cross-attention runs on a row-tile kernel that reads fragment-packed key / value sets (ops.rows_pack_kv): the 384-wide level's
    one-launch kernel or the 64-token level's head-sliced pair"""
    return ops.ROWS_KV_PACKED and (_xrows_ok(attn, hidden_states, residual, ln, L1, L2)
                                   or (_hs_route(attn, hidden_states, residual, ln) and ops.hs_cross_lengths_ok(L1, L2)))


def _rows_kv(pk, B, Lk, attn):
    return ops.RowsKV(pk, B, Lk, attn.heads, attn.to_out[0].weight.shape[0] // attn.heads)


def _hs_weights(attn, ln, self_attention):
    """(packed projection weights, their fp32 bias, packed to_out[0]) of apad_hs_attention / apad_hs_out, cached on the Attention module and
    re-packed when a parameter (the LayerNorm's included: it is folded into the projection) is re-assigned, moved, cast or updated in place"""
    ps = (attn.to_q.weight, attn.to_out[0].weight) + ((attn.to_k.weight, attn.to_v.weight) if self_attention else ()) + (tuple(ln[:2]) if ln is not None else ())
    key = (_pkey(*ps), self_attention, None if ln is None else float(ln[2]))
    if getattr(attn, "_hs_key", None) != key:
        if self_attention:  # (the to_q rows carry log2(e) / sqrt(d): q is the softmax's base-2 exponent operand as projected)
            d = attn.to_q.weight.shape[0] // attn.heads
            pk, bb = ops.hs_pack_qkv(attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, ln=ln, q_scale=ops.LOG2E / d ** 0.5)
        else:
            pk, bb = ops.hs_pack_rows(attn.to_q.weight, ln=ln)
        attn._hs_w = (pk, bb, ops.hs_pack_rows(attn.to_out[0].weight)[0])
        attn._hs_key = key
    return attn._hs_w


def _hs_sublayer(attn, hidden_states, residual, ln, **kv):
    """LayerNorm? -> projections -> attention (apad_hs_attention), then to_out + bias (+ residual) (apad_hs_out)"""
    self_attention = "k1" not in kv
    pk, bb, wo = _hs_weights(attn, ln, self_attention)
    o = ops.hs_attention(hidden_states, pk, bb, self_attention=self_attention, normalize=ln is not None, ln_eps=(ln[2] if ln is not None else 0.0),
                         q_prescaled=self_attention, **kv)
    return ops.hs_out(o, wo, attn.to_out[0].bias, residual, rowstat=True)


def vt_buffer(slot, B, heads, d, Lk, dtype, device):
    """Zero-padded V^T scratch [B, heads, d, round_up(Lk,32)].  The pad columns are never written (apad_gemm
    APAD_OUT_VT stores l < Lk only), so buffers are shared by shape across attention sites.
    The pool is keyed by (slot, shape, dtype, device, stream) and is never evicted on purpose: a captured hipGraph holds the raw
    addresses of the buffers its kernels were recorded with, so freeing an entry behind a live graph would hand its memory to someone
    else.  Its size is bounded by the distinct attention geometries of the loaded models (a dozen entries, < 100 MB at batch 64);
    ``clear_vt_pool()`` drops it when no captured graph is alive (e.g. between pipelines in one process)."""
    Lpad = ops.round_up(Lk, 32)
    # per stream: the denoise step may run the two CFG halves concurrently on two streams
    key = (slot, B, heads, d, Lpad, dtype, device, torch.cuda.current_stream().cuda_stream)
    buf = _vt_pool.get(key)
    if buf is None:
        buf = torch.zeros(B, heads, d, Lpad, dtype=dtype, device=device)
        _vt_pool[key] = buf
    return buf


def clear_vt_pool():
    """Drop the V^T scratch pool.  Only when no captured hipGraph that used it is still going to be replayed."""
    _vt_pool.clear()


def _pkey(*params):
    """identity + storage + version of parameters: changes when a weight is re-assigned (inference.py:56-57 re-assigns
    ``to_k_ip.weight`` / ``to_v_ip.weight``), moved, cast, or updated in place"""
    return tuple((id(p), p.data_ptr(), p._version, p.dtype) for p in params)


class _Hoist:
    """One hoisted (timestep-invariant) result: K / V^T / packed fragments of one (Attention site, condition tensor), or a
    mask's fp32 bias.  ``sig`` = what the result was computed from (condition version, weight identities / versions); when it
    no longer matches, ``make`` recomputes IN PLACE -- same buffers -- so that a captured hipGraph which reads them stays
    valid across pipeline calls (``refresh`` is what the pipeline runs before replaying a cached graph on new conditions)."""
    __slots__ = ("sig_fn", "make", "sig", "out", "owner")

    def __init__(self, sig_fn, make):
        self.sig_fn, self.make = sig_fn, make
        self.sig, self.out = sig_fn(), make()
        self.owner = HOIST_OWNER[0]  # who asked for it (a cached hipGraph, an eager pipeline call): dropped with its owner

    def get(self):
        sig = self.sig_fn()
        if sig != self.sig:
            new = self.make()
            same = len(new) == len(self.out) and all(
                (a is None and b is None) or (torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.dtype == b.dtype)
                or (not torch.is_tensor(a) and a == b) for a, b in zip(self.out, new))
            if same:
                for a, b in zip(self.out, new):
                    if torch.is_tensor(a):
                        a.copy_(b)
            else:
                self.out = new
            self.sig = sig
        return self.out


HOIST_OWNER = [None]  # set by the pipeline around a call (see AudioLDM2Pipeline.denoise)


def _loose_key(attn, t):
    """which (site, condition BUFFER) a hoisted result belongs to; the content is tracked by _Hoist.sig"""
    return (id(attn), t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype)


def _key_bias(attention_mask, B, Lk):
    """The UNet hands processors an additive bias [B,1,L] (modeling_audioldm2.py:741-747); apad_attention takes it
    as fp32 [B,L]."""
    if attention_mask is None:
        return None
    m = attention_mask
    if m.shape[-1] != Lk:
        raise ValueError(f"attention_mask has {m.shape[-1]} key positions, expected {Lk}")
    if m.numel() != B * Lk:
        raise ValueError("attention_mask must broadcast as [batch, 1, keys] (per-head masks are not on this path)")
    return m.reshape(B, Lk).float().contiguous()


def _as_tokens(hidden_states, residual):
    """the 4-D entry of the reference's processors (attention_processor.py:232-236, :363-367): [B, C, H, W] -> [B, H*W, C] (a transposed copy:
    the kernels want channel-contiguous rows); returns (tokens, residual in the same layout, the 4-D shape or None)"""
    if hidden_states.ndim != 4:
        return hidden_states, residual, None
    Bc, Cc, H, W = hidden_states.shape
    tok = hidden_states.reshape(Bc, Cc, H * W).transpose(1, 2).contiguous()
    if residual is hidden_states:
        residual = tok
    elif residual is not None:
        residual = residual.reshape(Bc, Cc, H * W).transpose(1, 2).contiguous()
    return tok, residual, (Bc, Cc, H, W)


def _as_image(out, shape4):
    """... and back (:290-291, :461-462)"""
    return out if shape4 is None else out.transpose(-1, -2).reshape(*shape4)


class AttnProcessor2_0(nn.Module):
    """Plain scaled-dot-product attention (reference :199-294).  Accepts dummy hidden_size / cross_attention_dim
    like the reference so it can live in AttnProcsLayers."""

    fuses_residual = True

    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()
        self.kv_cache_enabled = False
        self._kv_cache = None

    def clear_kv_cache(self):
        self._kv_cache = None

    def refresh_kv_cache(self):
        """recompute, in place, every hoisted result whose inputs changed (new condition content, re-assigned / stepped
        weights)"""
        for e in (self._kv_cache or {}).values():
            e.get()

    def drop_kv_owner(self, owner):
        if self._kv_cache:
            self._kv_cache = {k: e for k, e in self._kv_cache.items() if e.owner != owner}

    def _hoisted(self, key, sig_fn, make):
        if not self.kv_cache_enabled:
            return make()
        if self._kv_cache is None:
            self._kv_cache = {}
        e = self._kv_cache.get(key)
        if e is None:
            e = self._kv_cache[key] = _Hoist(sig_fn, make)
            return e.out
        return e.get()

    def _project_kv(self, attn, src, slot):
        B, Lk, _ = src.shape
        C_ = attn.to_k.weight.shape[0]
        k = ops.linear(src, attn.to_k.weight)
        if slot is None:  # persistent (cached) values own their buffer
            vt = torch.zeros(B, attn.heads, C_ // attn.heads, ops.round_up(Lk, 32), dtype=src.dtype, device=src.device)
        else:
            vt = vt_buffer(slot, B, attn.heads, C_ // attn.heads, Lk, src.dtype, src.device)
        ops.linear_vt(src, attn.to_v.weight, B, Lk, attn.heads, vt)
        return k, vt

    def _qkv_weight(self, attn, prescale=False):
        """to_q | to_k | to_v stacked for the one-launch projection.  prescale: the to_q rows carry log2(e) / sqrt(d) (scaled in
        fp32, rounded to the storage type once), so the projection writes q as the base-2 exponent operand of the softmax --
        rounded ONCE, like the reference's q -- and the attention kernel spends no instruction per score on the scale."""
        ps = (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight)
        key = tuple((id(p), p.data_ptr(), p._version, p.dtype, p.device) for p in ps) + (bool(prescale),)
        # cached on the Attention module (a processor instance may be shared by many sites)
        if getattr(attn, "_qkv_key", None) != key:
            wq = attn.to_q.weight.detach()
            if prescale:
                d = wq.shape[0] // attn.heads
                wq = (wq.float() * (ops.LOG2E / d ** 0.5)).to(wq.dtype)
            attn._qkv_w = torch.cat([wq, attn.to_k.weight.detach(), attn.to_v.weight.detach()], dim=0).contiguous()
            attn._qkv_key = key
        return attn._qkv_w

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 _residual=None, _ln=None):
        """``_residual`` (added after to_out) and ``_ln`` = (gamma, beta, eps) (LayerNorm applied to hidden_states
        first) are private fusion hooks used by this package's BasicTransformerBlock; without them the call is the
        reference's."""
        if attn.spatial_norm is not None or attn.group_norm is not None or attn.norm_cross:
            raise NotImplementedError("spatial_norm / group_norm / norm_cross are not on the AudioLDM2 path")
        if hidden_states.ndim == 4:
            tok, res, shape4 = _as_tokens(hidden_states, _residual)
            return _as_image(self(attn, tok, encoder_hidden_states, attention_mask, temb, _residual=res, _ln=_ln), shape4)
        if hidden_states.ndim != 3:
            raise ValueError("hidden_states must be [batch, tokens, channels] or [batch, channels, height, width]")
        B, N, C_ = hidden_states.shape
        heads = attn.heads
        if AG.on(hidden_states, encoder_hidden_states):
            return self._call_train(attn, hidden_states, encoder_hidden_states, attention_mask, _residual, _ln)
        prescaled = False
        if attn.residual_connection or attn.rescale_output_factor != 1.0:
            raise NotImplementedError("residual_connection / rescale_output_factor are not on the AudioLDM2 path")
        hs_route = _hs_route(attn, hidden_states, _residual, _ln)
        # (a masked self-attention -- the reference applies attention_mask to attn1 too, attention_processor.py:245-249 -- takes the generic
        #  q|k|v + apad_attention(key_bias) route below: the two fused self-attention kernels carry no key bias)
        if encoder_hidden_states is None and attention_mask is None and hs_route:
            return _hs_sublayer(attn, hidden_states, _residual, _ln)
        if (encoder_hidden_states is None and attention_mask is None and _ln is not None and ops.sattn_ok(hidden_states, heads) and attn.to_q.bias is None
                and tuple(attn.to_q.weight.shape) == (C_, C_)):
            # the two large levels: LayerNorm + q | k | v + attention in ONE launch (workgroup = (sample, head), K / V^T in LDS), then to_out
            key = (_pkey(attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, _ln[0], _ln[1]), float(_ln[2]))
            if getattr(attn, "_sattn_key", None) != key:
                attn._sattn_w = ops.sattn_pack(attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, _ln, heads)
                attn._sattn_key = key
            o = ops.self_attention_fused(hidden_states, attn._sattn_w[0], attn._sattn_w[1], heads, _ln[2])
            return ops.fused_linear(o, attn.to_out[0].weight, attn.to_out[0].bias, residual=_residual, rowstat=True)
        if encoder_hidden_states is None:
            Lk = N
            if ops.rp_ok(hidden_states) and attn.to_q.weight.shape[0] == C_:
                # LayerNorm + q|k|v in ONE launch: x is read once, V lands per-head transposed
                q = torch.empty(B, N, C_, dtype=hidden_states.dtype, device=hidden_states.device)
                k = torch.empty_like(q)
                vt = vt_buffer("self", B, heads, C_ // heads, N, hidden_states.dtype, hidden_states.device)
                prescaled = PRESCALE_Q and attn.to_q.bias is None
                ops.rowpanel(hidden_states, self._qkv_weight(attn, prescaled), [(q, None, C_, "row"), (k, None, C_, "row"), (vt, None, C_, "vt")],
                             ln=_ln, vt_geom=(heads, C_ // heads, N, vt.shape[-1]))
            elif attn.to_q.weight.shape[0] == C_ and C_ % 128 == 0:  # (fp32 mode: every width whose C % 128 == 0)
                # widths outside the row-panel envelope (the 640-wide level): q|k|v in ONE tiled launch, the LayerNorm folded into
                # it when the GEMM that produced hidden_states left its row statistics (ops.LN_FOLD), else a LayerNorm launch first
                qkv_w = self._qkv_weight(attn)
                fold = _ln is not None and ops.ln_foldable(hidden_states, qkv_w)
                hs = hidden_states if (_ln is None or fold) else ops.layer_norm(hidden_states, *_ln)
                q = torch.empty(B, N, C_, dtype=hidden_states.dtype, device=hidden_states.device)
                k = torch.empty_like(q)
                vt = vt_buffer("self", B, heads, C_ // heads, N, hidden_states.dtype, hidden_states.device)
                ops.linear_qkv(hs, qkv_w, B, N, heads, q, k, vt, ln=_ln if fold else None)
            else:
                hs = hidden_states if _ln is None else ops.layer_norm(hidden_states, *_ln)
                q = ops.linear(hs, attn.to_q.weight)
                k, vt = self._project_kv(attn, hs, "self")
        else:
            q = None  # projected below, unless the single-launch kernel takes the whole sub-layer
            ehs = encoder_hidden_states
            if ehs.dim() < 3:
                ehs = ehs.unsqueeze(0)
            Lk = ehs.shape[1]
            # hoisted K/V belong to ONE (Attention site, condition buffer) -- a processor instance may be shared by every site
            # (set_attn_processor(proc)) -- and are valid for one condition content and one pair of to_k / to_v weights
            # (re-assigned or stepped weights, an in-place update of the condition -> recomputed, in place)
            fused = _fused_xattn_ok(attn, hidden_states, _residual, _ln, Lk)
            rows = not fused and ehs.shape[0] == B and _rows_kv_route(attn, hidden_states, _residual, _ln, Lk)
            persistent = self.kv_cache_enabled

            def make(attn=attn, ehs=ehs, fused=fused, rows=rows, persistent=persistent):
                k_, vt_ = self._project_kv(attn, ehs, None if persistent else "cross")
                # packed with the projection: the weight-stationary kernel's layout, or the row-tile kernels' fragment sets
                return (k_, vt_, ops.xattn_pack_kv(k_, vt_, ehs.shape[1]) if fused else (ops.rows_pack_kv(k_, vt_).data if rows else None))

            k, vt, pk = self._hoisted(_loose_key(attn, ehs), lambda attn=attn, ehs=ehs: (ehs._version, _pkey(attn.to_k.weight, attn.to_v.weight)), make)
        if attention_mask is not None:
            # the mask -> fp32 bias conversion is timestep-invariant too: hoisted with the K/V (two tiny torch kernels
            # per masked site per step otherwise)
            (bias,) = self._hoisted(("bias",) + _loose_key(None, attention_mask) + (Lk,), lambda m=attention_mask: m._version,
                                    lambda m=attention_mask: (_key_bias(m, B, Lk),))
        else:
            bias = None
        if encoder_hidden_states is not None and fused and pk is not None:
            wq_p, q_fold, wo_p = _xattn_weights(attn, _ln)
            return ops.fused_cross_attention(hidden_states, wq_p, wo_p, attn.to_out[0].bias, pk, Lk, heads, ln=_ln, key_bias=bias, q_fold=q_fold)
        if encoder_hidden_states is not None and rows and pk is not None:
            k, vt = _rows_kv(pk, B, Lk, attn), None
        if encoder_hidden_states is not None and _xrows_ok(attn, hidden_states, _residual, _ln, Lk) and k.shape[0] == B:
            wq_p, wo_p = _xrows_weights(attn)
            return ops.cross_attention_rows(hidden_states, wq_p, wo_p, attn.to_out[0].bias, k, vt, heads, ln=_ln, key_bias=bias)
        if encoder_hidden_states is not None and hs_route and ops.hs_cross_lengths_ok(Lk) and k.shape[0] == B:
            return _hs_sublayer(attn, hidden_states, _residual, _ln, k1=k, vt1=vt, key_bias=bias)
        if q is None:
            q = ops.fused_linear(hidden_states, attn.to_q.weight, ln=_ln)
        o = ops.attention(q, k, vt, Lk, heads, key_bias=bias, q_prescaled=prescaled)
        return ops.fused_linear(o, attn.to_out[0].weight, attn.to_out[0].bias, residual=_residual, rowstat=True)


def _train_common(attn):
    if attn.residual_connection or attn.rescale_output_factor != 1.0:
        raise NotImplementedError("residual_connection / rescale_output_factor are not on the AudioLDM2 path")


def _attn_call_train(self, attn, hidden_states, encoder_hidden_states, attention_mask, _residual, _ln):
    """Training-mode AttnProcessor2_0 (gradients flow to hidden_states; autograd.py): the un-fused chain LayerNorm ->
    to_q / to_k / to_v -> attention -> to_out (+ residual), every node a HIP forward with a HIP backward."""
    _train_common(attn)
    B, N, _ = hidden_states.shape
    if _ln is not None and _residual is hidden_states:
        hs, _residual = AG.layer_norm_res(hidden_states, *_ln)  # (the residual gradient joins the LayerNorm backward launch)
    else:
        hs = hidden_states if _ln is None else AG.layer_norm(hidden_states, *_ln)
    src = hs if encoder_hidden_states is None else (encoder_hidden_states if encoder_hidden_states.dim() == 3
                                                    else encoder_hidden_states.unsqueeze(0))
    if encoder_hidden_states is None:  # self-attention: one input-gradient GEMM for the three projections
        q, k, v, vt = AG.qkv(hs, attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.heads)
    else:
        q = AG.linear(hs, attn.to_q.weight)
        k = AG.linear(src, attn.to_k.weight)
        v = AG.linear(src, attn.to_v.weight)
        vt = None
    o = AG.attention(q, k, v, attn.heads, _key_bias(attention_mask, B, src.shape[1]), vt=vt)
    return AG.linear(o, attn.to_out[0].weight, attn.to_out[0].bias, residual=_residual)


AttnProcessor2_0._call_train = _attn_call_train


class IPAttnProcessor2_0(nn.Module):
    """Decoupled cross-attention (reference :297-470): text branch over the first ``num_tokens`` tokens with the frozen
    ``attn.to_k/to_v``, audio branch over the remaining tokens with the trainable ``to_k_ip/to_v_ip``, blended
    ``text + scale * audio`` inside one fused kernel."""

    fuses_residual = True

    def __init__(self, hidden_size, name, cross_attention_dim=None, num_tokens=4, scale=1.0, do_copy=False,
                 copy_dir=None):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.num_tokens = num_tokens
        self.scale = scale
        self.name = name
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.kv_cache_enabled = False
        self._kv_cache = None
        if do_copy:
            # reference :328-344 warm-starts from copied_cross_attention/{name}_{k,v}.bin
            from .wiring import load_copied_cross_attention
            load_copied_cross_attention(self, copy_dir)

    def clear_kv_cache(self):
        self._kv_cache = None

    refresh_kv_cache = AttnProcessor2_0.refresh_kv_cache
    drop_kv_owner = AttnProcessor2_0.drop_kv_owner
    _hoisted = AttnProcessor2_0._hoisted

    def _project(self, attn, ehs):
        B = ehs.shape[0]
        nt = self.num_tokens
        txt = ehs[:, :nt, :]
        aud = ehs[:, nt:, :]
        Lt, La = txt.shape[1], aud.shape[1]
        C_ = attn.to_k.weight.shape[0]
        d = C_ // attn.heads
        persistent = self.kv_cache_enabled
        # ragged views -> dense rows for the GEMM A operand (plumbing copies of <= 520 x 768 tokens)
        txt = txt.contiguous()
        k_t = ops.linear(txt, attn.to_k.weight)
        vt_t = (torch.zeros(B, attn.heads, d, ops.round_up(Lt, 32), dtype=ehs.dtype, device=ehs.device) if persistent
                else vt_buffer("ip_txt", B, attn.heads, d, Lt, ehs.dtype, ehs.device))
        ops.linear_vt(txt, attn.to_v.weight, B, Lt, attn.heads, vt_t)
        k_a = vt_a = None
        if La > 0:
            aud = aud.contiguous()
            k_a = ops.linear(aud, self.to_k_ip.weight)
            vt_a = (torch.zeros(B, attn.heads, d, ops.round_up(La, 32), dtype=ehs.dtype, device=ehs.device)
                    if persistent else vt_buffer("ip_aud", B, attn.heads, d, La, ehs.dtype, ehs.device))
            ops.linear_vt(aud, self.to_v_ip.weight, B, La, attn.heads, vt_a)
        return k_t, vt_t, Lt, k_a, vt_a, La, None, None

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0,
                 _residual=None, _ln=None):
        if scale != 1.0:
            # the reference dereferences an undefined ``logger`` here (:356-357) -> NameError; reject loudly instead
            raise ValueError("`scale` of IPAttnProcessor2_0 is set through the `scale` attribute, not the call kwarg")
        if attn.spatial_norm is not None or attn.group_norm is not None or attn.norm_cross:
            raise NotImplementedError("spatial_norm / group_norm / norm_cross are not on the AudioLDM2 path")
        if hidden_states.ndim == 4:
            tok, res, shape4 = _as_tokens(hidden_states, _residual)
            return _as_image(self(attn, tok, encoder_hidden_states, attention_mask, temb, _residual=res, _ln=_ln), shape4)
        if hidden_states.ndim != 3:
            raise ValueError("hidden_states must be [batch, tokens, channels] or [batch, channels, height, width]")
        if encoder_hidden_states is None:
            raise ValueError("IPAttnProcessor2_0 needs encoder_hidden_states = [text tokens | audio tokens]")
        ehs = encoder_hidden_states
        if ehs.dim() < 3:
            ehs = ehs.unsqueeze(0)
        B, N, _ = hidden_states.shape
        if AG.on(hidden_states, ehs, self.to_k_ip.weight, self.to_v_ip.weight):
            return self._call_train(attn, hidden_states, ehs, attention_mask, _residual, _ln)
        # see AttnProcessor2_0: one entry per (site, condition buffer), valid for one condition content and one set of the four
        # projection weights, so a re-assigned to_k_ip / to_v_ip (inference.py:56-57) or an optimizer step is never served stale K/V
        Lt0 = min(self.num_tokens, ehs.shape[1])
        fused = _fused_xattn_ok(attn, hidden_states, _residual, _ln, Lt0, ehs.shape[1] - Lt0, attention_mask is not None)  # (no activation captured below)
        rows = not fused and ehs.shape[0] == B and _rows_kv_route(attn, hidden_states, _residual, _ln, Lt0, ehs.shape[1] - Lt0)

        def make(attn=attn, ehs=ehs, fused=fused, rows=rows):
            kv_ = self._project(attn, ehs)
            k_t_, vt_t_, Lt_, k_a_, vt_a_, La_ = kv_[:6]
            if fused:  # packed once, with the hoisted projection
                kv_ = kv_[:6] + (ops.xattn_pack_kv(k_t_, vt_t_, Lt_), ops.xattn_pack_kv(k_a_, vt_a_, La_) if La_ > 0 else None)
            elif rows:  # ... or into the fragment sets the row-tile kernels of the 384- / 640-wide levels read
                kv_ = kv_[:6] + (ops.rows_pack_kv(k_t_, vt_t_).data, ops.rows_pack_kv(k_a_, vt_a_).data if La_ > 0 else None)
            return kv_

        sig = lambda attn=attn, ehs=ehs: (ehs._version, self.num_tokens,
                                          _pkey(attn.to_k.weight, attn.to_v.weight, self.to_k_ip.weight, self.to_v_ip.weight))
        k_t, vt_t, Lt, k_a, vt_a, La, pk_t, pk_a = self._hoisted(_loose_key(attn, ehs), sig, make)
        bias = None
        if attention_mask is not None:
            # reference :424-428 keeps only mask column 0 (split by the singleton query dim) and broadcasts it
            # over the text keys
            m = attention_mask.reshape(B, -1)[:, :1].float()
            bias = m.expand(B, Lt).contiguous()
        if pk_t is not None and _fused_xattn_ok(attn, hidden_states, _residual, _ln, Lt, La, bias is not None):
            wq_p, q_fold, wo_p = _xattn_weights(attn, _ln)
            return ops.fused_cross_attention(hidden_states, wq_p, wo_p, attn.to_out[0].bias, pk_t, Lt, attn.heads, ln=_ln,
                                             key_bias=bias, kv2_packed=pk_a, L2=La, scale2=self.scale, q_fold=q_fold)
        if rows and pk_t is not None:
            k_t, vt_t = _rows_kv(pk_t, B, Lt, attn), None
            if La > 0:
                k_a, vt_a = _rows_kv(pk_a, B, La, attn), None
        if _xrows_ok(attn, hidden_states, _residual, _ln, Lt, La) and k_t.shape[0] == B:
            wq_p, wo_p = _xrows_weights(attn)
            return ops.cross_attention_rows(hidden_states, wq_p, wo_p, attn.to_out[0].bias, k_t, vt_t, attn.heads, ln=_ln, key_bias=bias,
                                            k2=k_a, vt2=vt_a, scale2=self.scale)
        if attn.residual_connection or attn.rescale_output_factor != 1.0:
            raise NotImplementedError("residual_connection / rescale_output_factor are not on the AudioLDM2 path")
        if _hs_route(attn, hidden_states, _residual, _ln) and ops.hs_cross_lengths_ok(Lt, La) and k_t.shape[0] == B:
            return _hs_sublayer(attn, hidden_states, _residual, _ln, k1=k_t, vt1=vt_t, key_bias=bias, k2=k_a, vt2=vt_a, scale2=self.scale)
        q = ops.fused_linear(hidden_states, attn.to_q.weight, ln=_ln)
        o = ops.attention(q, k_t, vt_t, Lt, attn.heads, key_bias=bias, k2=k_a, vt2=vt_a, L2=La, scale2=self.scale)
        return ops.fused_linear(o, attn.to_out[0].weight, attn.to_out[0].bias, residual=_residual, rowstat=True)


    def _call_train(self, attn, hidden_states, ehs, attention_mask, _residual, _ln):
        """Training mode (reference :347-470 under autograd; train_apadapter_v2.py:941-957): gradients w.r.t.
        hidden_states, to_k_ip.weight and to_v_ip.weight (and the condition tokens if they require grad)."""
        _train_common(attn)
        B = hidden_states.shape[0]
        nt = self.num_tokens
        txt, aud = ehs[:, :nt, :].contiguous(), ehs[:, nt:, :].contiguous()
        if _ln is not None and _residual is hidden_states:
            hs, _residual = AG.layer_norm_res(hidden_states, *_ln)  # (the residual gradient joins the LayerNorm backward launch)
        else:
            hs = hidden_states if _ln is None else AG.layer_norm(hidden_states, *_ln)
        q = AG.linear(hs, attn.to_q.weight)
        k_t, v_t = AG.linear(txt, attn.to_k.weight), AG.linear(txt, attn.to_v.weight)
        bias = None
        if attention_mask is not None:
            bias = attention_mask.reshape(B, -1)[:, :1].float().expand(B, txt.shape[1]).contiguous()
        if aud.shape[1] == 0:  # no audio tokens: the text branch alone (reference :435-445 on an empty slice contributes 0)
            o = AG.attention(q, k_t, v_t, attn.heads, bias)
        else:
            k_a, v_a = AG.linear(aud, self.to_k_ip.weight), AG.linear(aud, self.to_v_ip.weight)
            o = AG.ip_attention(q, k_t, v_t, k_a, v_a, attn.heads, bias, float(self.scale))
        return AG.linear(o, attn.to_out[0].weight, attn.to_out[0].bias, residual=_residual)

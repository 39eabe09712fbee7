"""Autograd plumbing for the adapter's training step (SURVEY a-11, reference train_apadapter_v2.py:941-979).

The reference back-propagates through the whole frozen UNet with torch autograd to reach the 64 trainable tensors
``to_k_ip.weight`` / ``to_v_ip.weight`` (attention_processor.py:324-325).  Here torch.autograd is only the tape:
every ``Function`` below runs its forward AND its backward in libapadapter_hip.so (``ops``), never in torch
kernels.  Gradient accumulation at forks (residual branches, U-Net skips) is done by the autograd engine.

The un-fused training forward is taken only where a gradient is actually needed (``on(...)``): upstream of the
first adapted attention nothing requires grad, so those layers keep running the fused inference kernels.
Frozen weights need no weight gradient; their transposed / flipped copies for the dgrad GEMMs are cached.
"""
import weakref

import os as _os

import torch

from . import ops


def on(*tensors):
    """True when autograd must record through an op with these inputs."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


_wcache = {}


def _psig(*ps):
    """identity + storage + version + type + place of companion weights (the signature of a stacked copy, not its key)"""
    return tuple((id(p), p.data_ptr(), p._version, p.dtype, p.device) for p in ps)


def _cached(w, tag, make, extra=()):
    """per-weight derived tensor (transposed / flipped copy), rebuilt when the weight is re-assigned or updated.  Only FROZEN
    weights are cached: the optimizer kernel updates trainable ones through raw pointers, which no version counter sees."""
    if w.requires_grad:
        return make(w.detach())
    key = (id(w), tag)  # ONE entry per (weight, kind): a changed companion (``extra``) overwrites it instead of piling up keys
    sig = (w.data_ptr(), w._version, w.dtype, w.device) + tuple(extra)
    hit = _wcache.get(key)
    # the entry must belong to THIS tensor object: ids (and allocator addresses) are recycled once a model is freed, and a recycled
    # id with an equal signature would otherwise serve another model's derived weight
    if hit is None or hit[0] != sig or hit[2]() is not w:
        if len(_wcache) > 4096:  # entries of freed models
            for k in [k for k, v in _wcache.items() if v[2]() is None]:
                del _wcache[k]
        hit = (sig, make(w.detach()), weakref.ref(w))
        _wcache[key] = hit
    return hit[1]


def _wt(w):
    """W [N, K] -> W^T [K, N] contiguous: dx = dy . W is apad_gemm(a = dy, w = W^T)"""
    return _cached(w, "T", lambda t: t.reshape(t.shape[0], -1).t().contiguous())


def _conv_dgrad_w(w):
    """conv weight [Cout, Cin, 3, 3] -> [Cin, 9*Cout] in (2-ky, 2-kx, cout) order: the stride-1 convolution of dy with
    it is the input gradient"""
    return _cached(w, "dgrad", lambda t: t.flip(2, 3).permute(1, 2, 3, 0).reshape(t.shape[1], -1).contiguous())


def _conv_fwd_w(w):
    return _cached(w, "packed", lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], -1).contiguous())


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Linear(torch.autograd.Function):
    """y = x W^T (+ b) (+ residual)"""

    @staticmethod
    def forward(ctx, x, w, b, residual):
        ctx.save_for_backward(x if w.requires_grad else None, w)
        ctx.has_res = residual is not None
        # a trainer may attach an fp32 accumulator to a trainable weight (AdapterTrainer: a view of its flat gradient buffer
        # and the loss scale): the weight gradient is then formed and accumulated in fp32, as the reference's fp32 adapter
        # gradients are, instead of being rounded to the storage type once per micro-batch
        ctx.sink = getattr(w, "_apad_grad_sink", None)
        w2 = w.reshape(w.shape[0], -1)
        return ops.linear(_c(x), w2, b, residual=None if residual is None else _c(residual))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear(dy, _wt(w))
        if ctx.needs_input_grad[1]:
            if ctx.sink is not None:
                acc, inv_scale = ctx.sink
                if inv_scale == 1.0 and acc.is_contiguous() and acc.dim() == 2:  # accumulated in the GEMM's epilogue
                    ops.weight_grad(dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1]), fp32=True, acc=acc)
                else:
                    acc.add_(ops.weight_grad(dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1]), fp32=True).reshape(acc.shape),
                             alpha=inv_scale)
            else:
                dw = ops.weight_grad(dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])).reshape(w.shape)
        return dx, dw, None, (dy if ctx.has_res and ctx.needs_input_grad[3] else None)


def linear(x, w, b=None, residual=None):
    return _Linear.apply(x, w, b, residual)


_BWD_PACKED = True


def _packed_grads(dq, dk, dv):
    """[dq | dk | dv] as one [.., 3C] tensor: the buffer apad_attention_bwd wrote them into as column blocks (ops.attention_bwd(packed=True))
    when the three are such views, their concatenation otherwise"""
    base = dq._base
    C_ = dq.shape[-1]
    if (base is not None and dk._base is base and dv._base is base and base.is_contiguous() and base.shape[-1] == 3 * C_
            and base.shape[:-1] == dq.shape[:-1] and dq.stride() == dk.stride() == dv.stride() == base.stride()
            and dq.storage_offset() == base.storage_offset() and dk.storage_offset() == base.storage_offset() + C_
            and dv.storage_offset() == base.storage_offset() + 2 * C_):
        return base
    return torch.cat([_c(dq), _c(dk), _c(dv)], dim=-1)


class _QKV(torch.autograd.Function):
    """q, k, v = x Wq^T, x Wk^T, x Wv^T with FROZEN weights (the UNet's self-attention): three forward GEMMs, but ONE input-gradient
    GEMM -- dx = [dq | dk | dv] . [Wq; Wk; Wv] -- instead of three GEMMs and two accumulation kernels (the training step at batch 4 is
    bound by its launch count)"""

    @staticmethod
    def forward(ctx, x, wq, wk, wv):
        x = _c(x)
        ctx.save_for_backward(wq, wk, wv)
        return ops.linear(x, wq), ops.linear(x, wk), ops.linear(x, wv)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        wq, wk, wv = ctx.saved_tensors
        wst = _cached(wq, "qkvT", lambda t: torch.cat([t, wk.detach(), wv.detach()], 0).t().contiguous(), extra=_psig(wk, wv))  # [C, 3C]
        return ops.linear(_packed_grads(dq, dk, dv), wst), None, None, None


class _QKVT(_QKV):
    """_QKV with ONE forward launch: q, k, v row-major and v^T (what the forward attention kernel reads) from apad_gemm's q | k | v^T
    output mode with its optional row-major v (four launches -- three projections and a transpose -- otherwise; the step at batch 4 is
    bound by its launch count).  v^T lives in a scratch buffer shared by shape: it is consumed by the attention launch that follows."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, heads):
        from .processors import vt_buffer
        x = _c(x)
        B, N, Cc = x.shape
        ctx.save_for_backward(wq, wk, wv)
        w = _cached(wq, "qkv", lambda t: torch.cat([t, wk.detach(), wv.detach()], 0).contiguous(), extra=_psig(wk, wv))
        q, k, v = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        vt = vt_buffer("train_self", B, heads, Cc // heads, N, x.dtype, x.device)
        ops.linear_qkv(x, w, B, N, heads, q, k, vt, v=v)
        ctx.mark_non_differentiable(vt)
        return q, k, v, vt

    @staticmethod
    def backward(ctx, dq, dk, dv, _dvt):
        return _QKV.backward(ctx, dq, dk, dv) + (None,)


_QKV_FUSED = True


def qkv(x, wq, wk, wv, heads=None):
    """-> (q, k, v, vt or None)"""
    if wq.requires_grad or wk.requires_grad or wv.requires_grad:
        return linear(x, wq), linear(x, wk), linear(x, wv), None
    C_ = x.shape[-1]
    if (_QKV_FUSED and heads is not None and x.dtype in ops.FUSED_DTYPES and x.dim() == 3 and C_ % 128 == 0
            and tuple(wq.shape) == tuple(wk.shape) == tuple(wv.shape) == (C_, C_)):
        return _QKVT.apply(x, wq, wk, wv, heads)
    return _QKV.apply(x, wq, wk, wv) + (None,)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = _c(x)
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        return ops.layer_norm(x, g, b, eps)

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        return ops.layer_norm_bwd(x, g, _c(dy), ctx.eps), None, None, None


def layer_norm(x, g, b, eps):
    return _LayerNorm.apply(x, g, b, eps)


class _LayerNormRes(torch.autograd.Function):
    """(LayerNorm(x), x): the pre-norm sub-layers feed x to the LayerNorm AND to the residual add behind the branch
    (BasicTransformerBlock: x = x + attn(norm(x))).  Handing the residual out of THIS node makes x single-use for the autograd
    engine, and the backward adds the residual gradient inside the LayerNorm-backward launch (384 accumulation launches per
    training step otherwise: the step at batch 4 is bound by its launch count)."""

    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = _c(x)
        ctx.save_for_backward(x, g)
        ctx.eps = eps
        return ops.layer_norm(x, g, b, eps), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, g = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None
        return ops.layer_norm_bwd(x, g, _c(dy), ctx.eps, dres=None if dres is None else _c(dres)), None, None, None


_LN_RES = True


def layer_norm_res(x, g, b, eps):
    """-> (LayerNorm(x), x as the residual operand of the sub-layer's closing Linear)"""
    if not _LN_RES:
        return _LayerNorm.apply(x, g, b, eps), x
    return _LayerNormRes.apply(x, g, b, eps)


class _GroupNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, groups, eps, silu):
        x = _c(x)
        ctx.save_for_backward(x, g, b)
        ctx.cfg = (groups, eps, silu)
        return ops.group_norm(x, g, b, groups, eps, silu=silu)

    @staticmethod
    def backward(ctx, dy):
        x, g, b = ctx.saved_tensors
        groups, eps, silu = ctx.cfg
        return ops.group_norm_bwd(x, g, b, _c(dy), groups, eps, silu), None, None, None, None, None


def group_norm(x, g, b, groups, eps, silu):
    return _GroupNorm.apply(x, g, b, groups, eps, silu)


class _Conv3x3(torch.autograd.Function):
    """NHWC 3x3 convolution, padding 1; stride 1 / 2, optional nearest-upsampled source, fused residual and
    per-sample bias (time-embedding projection, no gradient)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, rowgroup_bias, geom):
        B, H, W, stride, up, rows_per_group = geom
        ctx.save_for_backward(w)
        ctx.geom = geom
        ctx.has_res = residual is not None
        out, Ho, Wo = ops.conv3x3(_c(x), _conv_fwd_w(w), b, B, H, W, stride=stride, up=up,
                                  residual=None if residual is None else _c(residual), rowgroup_bias=rowgroup_bias,
                                  rows_per_group=rows_per_group)
        ctx.out_hw = (Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        B, H, W, stride, up, _ = ctx.geom
        Ho, Wo = ctx.out_hw
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            wd = _conv_dgrad_w(w)
            if stride == 2:
                z = ops.zero_stuff2(dy, B, H, W, Ho, Wo)
                dx, _, _ = ops.conv3x3(z, wd, None, B, H, W)
            elif up is not None:
                dup, _, _ = ops.conv3x3(dy, wd, None, B, up[0], up[1])
                dx = ops.upsample_nearest_bwd(dup, B, H, W, up[0], up[1])
            else:
                dx, _, _ = ops.conv3x3(dy, wd, None, B, H, W)
        return dx, None, None, (dy if ctx.has_res and ctx.needs_input_grad[3] else None), None, None


def conv3x3(x, w, b, B, H, W, stride=1, up=None, residual=None, rowgroup_bias=None, rows_per_group=0):
    """returns (out [B, Ho*Wo, Cout], Ho, Wo)"""
    Hs, Ws = up if up is not None else (H, W)
    Ho, Wo = (Hs + 2 - 3) // stride + 1, (Ws + 2 - 3) // stride + 1
    out = _Conv3x3.apply(x, w, b, residual, rowgroup_bias, (B, H, W, stride, up, rows_per_group))
    return out, Ho, Wo


class _Geglu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proj):
        proj = _c(proj)
        ctx.save_for_backward(proj)
        return ops.geglu(proj)

    @staticmethod
    def backward(ctx, dh):
        (proj,) = ctx.saved_tensors
        return ops.geglu_bwd(proj, _c(dh))


def geglu(proj):
    return _Geglu.apply(proj)


class _Attention(torch.autograd.Function):
    """o = softmax(q k^T / sqrt(d) + bias) v, all [B, tokens, C] row-major"""

    @staticmethod
    def forward(ctx, q, k, v, heads, key_bias, vt=None):
        q, k, v = _c(q), _c(k), _c(v)
        if vt is None:
            vt = ops.head_transpose(v, heads)
        o, lse = ops.attention_lse(q, k, vt, k.shape[1], heads, key_bias=key_bias)
        ctx.save_for_backward(q, k, v, o, lse, key_bias)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, key_bias = ctx.saved_tensors
        need_kv = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        # (self-attention whose q, k, v all take gradients: dq | dk | dv as column blocks of one buffer, see _QKV.backward)
        packed = _BWD_PACKED and all(ctx.needs_input_grad[:3]) and q.shape == k.shape
        dq, dk, dv = ops.attention_bwd(q, k, v, o, _c(do), lse, ctx.heads, key_bias=key_bias, need_dkv=need_kv, packed=packed)
        return dq, dk, dv, None, None, None


def attention(q, k, v, heads, key_bias=None, vt=None):
    """vt: v per-head transposed [B, heads, d, Lpad], when the projection already produced it (qkv)"""
    return _Attention.apply(q, k, v, heads, key_bias, vt)


class _IPAttention(torch.autograd.Function):
    """Decoupled cross-attention (attention_processor.py:429-454): o = A(q, k_t, v_t, bias) + scale * A(q, k_a, v_a).
    The blended output comes from the same dual-segment launch inference uses; the two single-segment launches only
    provide each branch's output and log-sum-exp for the backward."""

    @staticmethod
    def forward(ctx, q, k_t, v_t, k_a, v_a, heads, key_bias, scale):
        q, k_t, v_t, k_a, v_a = _c(q), _c(k_t), _c(v_t), _c(k_a), _c(v_a)
        vt_t, vt_a = ops.head_transpose(v_t, heads), ops.head_transpose(v_a, heads)
        o_t, lse_t = ops.attention_lse(q, k_t, vt_t, k_t.shape[1], heads, key_bias=key_bias)
        o_a, lse_a = ops.attention_lse(q, k_a, vt_a, k_a.shape[1], heads)
        o = ops.attention(q, k_t, vt_t, k_t.shape[1], heads, key_bias=key_bias, k2=k_a, vt2=vt_a, L2=k_a.shape[1], scale2=scale)
        ctx.save_for_backward(q, k_t, v_t, k_a, v_a, o_t, o_a, lse_t, lse_a, key_bias)
        ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k_t, v_t, k_a, v_a, o_t, o_a, lse_t, lse_a, key_bias = ctx.saved_tensors
        do = _c(do)
        need_t = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        need_a = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        dq, dk_t, dv_t = ops.attention_bwd(q, k_t, v_t, o_t, do, lse_t, ctx.heads, key_bias=key_bias, need_dkv=need_t)
        _, dk_a, dv_a = ops.attention_bwd(q, k_a, v_a, o_a, do, lse_a, ctx.heads, dout_scale=ctx.scale, need_dkv=need_a, dq=dq)
        return dq, dk_t, dv_t, dk_a, dv_a, None, None, None


def ip_attention(q, k_t, v_t, k_a, v_a, heads, key_bias, scale):
    return _IPAttention.apply(q, k_t, v_t, k_a, v_a, heads, key_bias, scale)


class _MSE(torch.autograd.Function):
    """F.mse_loss(pred.float(), target.float(), reduction="mean") (train_apadapter_v2.py:954)"""

    @staticmethod
    def forward(ctx, pred, target, grad_scale):
        loss, dpred = ops.mse_loss_grad(_c(pred), _c(target.float()), grad_scale)  # loss itself is unscaled
        ctx.save_for_backward(dpred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        # g is 1 for loss.backward(); applied as a device-side scalar multiply of a [B, 8, 250, 16] tensor (no host
        # sync: the step must stay capturable in a hipGraph)
        return dpred * g.to(dpred.dtype), None, None


def mse_loss(pred, target, grad_scale=1.0):
    """grad_scale: static loss scale -- the backward delivers grad_scale * d loss / d pred (applied in fp32 inside the kernel,
    before the rounding to the storage type); the returned loss is unscaled"""
    return _MSE.apply(pred, target, float(grad_scale))

"""Tensor-level wrappers over the C ABI (include/apadapter_hip.h).

PyTorch is used only as plumbing here: device memory (torch.empty), the current HIP stream, dtype tags.  All
arithmetic happens in libapadapter_hip.so.  Nothing in this module falls back to torch ops: a missing library
or a CPU tensor raises.
"""
import ctypes as C
import math

import torch

from . import _lib as L

_DT = {torch.bfloat16: L.BF16, torch.float16: L.F16, torch.float32: L.F32}
_EPI = {None: L.EPI_NONE, "none": L.EPI_NONE, "silu": L.EPI_SILU, "gelu": L.EPI_GELU, "geglu": L.EPI_GEGLU, "tanh": L.EPI_TANH,
        "relu": L.EPI_RELU, "gelu_tanh": L.EPI_GELU_TANH, "geglu_tanh": L.EPI_GEGLU_TANH}  # the last three: fp32 mode only


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _req(t, name, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor; the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: dtype {t.dtype} != {dtype}")
    if t.stride(-1) != 1:
        raise RuntimeError(f"{name}: last dim must be contiguous")
    return t


def round_up(x, m):
    return (x + m - 1) // m * m


def set_gemm_ring(mode):
    """process-wide kernel form of apad_gemm's latency-bound plain launches (apad_set_gemm_ring): 0 tiled, 1 / 2 LDS-DMA ring for
    under-filled grids / every launch below 16000 rows, -1 the APAD_GEMM_RING environment variable.  Returns the previous setting."""
    return int(L.lib().apad_set_gemm_ring(int(mode)))


def gemm(a, w, *, M, N, K, lda, out, ldo, bias=None, residual=None, ldr=0, act=None, rowgroup_bias=None, ld_rg=0,
         rows_per_group=0, step_ptr=None, a_mode=L.A_PLAIN, out_mode=L.OUT_ROWMAJOR, conv=None, vt=None, ldw=None,
         residual_row_mod=0, asym_pad=False, rowstat_out=None, ln_fold=None, a2=None, lda2=0, k_split=0, a_row_mod=0, a2_row_mod=0,
         w_halo=None, workspace=None):
    """Raw descriptor call; the typed helpers below are what the model code uses.  rowstat_out: fp32 [M, N/64, 2] side output
    (row statistics of the stored rows); ln_fold = (rowstat_in [M, T, 2], colsum, bias_fp32, eps): LayerNorm by algebra."""
    d = L.GemmDesc()
    if rowstat_out is not None:
        d.rowstat_out = rowstat_out.data_ptr()
    if ln_fold is not None:
        rs, cs, bb, eps = ln_fold
        d.rowstat_in, d.ln_colsum, d.ln_bias, d.rowstat_in_tiles, d.ln_eps = rs.data_ptr(), cs.data_ptr(), bb.data_ptr(), rs.shape[-2], float(eps)
    d.a, d.w, d.out = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.bias, d.residual, d.rowgroup_bias, d.step_ptr = _ptr(bias), _ptr(residual), _ptr(rowgroup_bias), _ptr(step_ptr)
    d.M, d.N, d.K, d.lda, d.ldw, d.ldo, d.ldr, d.ld_rg = M, N, K, lda, (K if ldw is None else ldw), ldo, ldr, ld_rg
    d.rows_per_group = rows_per_group
    d.residual_row_mod = residual_row_mod
    d.a_mode, d.epilogue, d.out_mode, d.dtype = a_mode, _EPI[act], out_mode, _DT[w.dtype]
    if conv is not None:
        (d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.stride, d.Hup, d.Wup, d.src_batch_mod) = conv
    if vt is not None:
        d.heads, d.head_dim, d.L, d.Lpad = vt
    d.conv_asym_pad = 1 if asym_pad else 0
    if a2 is not None:
        d.a2, d.lda2, d.k_split = a2.data_ptr(), lda2, k_split
    d.a_row_mod, d.a2_row_mod = a_row_mod, a2_row_mod
    d.w_halo = _ptr(w_halo)
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    L.check(L.lib().apad_gemm(C.byref(d), _stream()), "apad_gemm")
    return out


import os as _os
import weakref as _weakref

# LayerNorm folded into the Linear behind it where no fused row-panel kernel covers the width (the 640-wide level): the producing
# GEMM emits row statistics, the consuming GEMM applies (x - mean) * rstd * gamma + beta by algebra
LN_FOLD = True  # (module attribute only: no environment switch since round 5)
_fold_cache = {}


def rowstat_of(x):
    """the row statistics [M, C/64, 2] the producing GEMM left for x (an attribute of THAT tensor object: views and copies do not
    carry it, and nothing on this path writes into x after its producer), or None"""
    return getattr(x, "_apad_rowstat", None)


def _ln_folded(w, bias, gamma, beta):
    """(w * gamma in the storage type, its fp32 row sums, fp32 beta . W^T + bias) of a Linear behind a LayerNorm; cached per weight,
    rebuilt when any of the four parameters is re-assigned or updated"""
    ps = (w, bias, gamma, beta)
    sig = tuple(None if p is None else (id(p), p.data_ptr(), p._version, p.dtype, p.device) for p in ps)
    hit = _fold_cache.get(id(w))
    if hit is None or hit[0] != sig or hit[2]() is not w:
        if len(_fold_cache) > 1024:
            for k in [k for k, v in _fold_cache.items() if v[2]() is None]:
                del _fold_cache[k]
        wf = w.detach().float()
        wg = (wf * gamma.detach().float()).to(w.dtype).contiguous()
        cs = wg.float().sum(1).contiguous()
        bb = wf @ beta.detach().float()
        if bias is not None:
            bb = bb + bias.detach().float()
        hit = (sig, (wg, cs, bb.contiguous()), _weakref.ref(w))
        _fold_cache[id(w)] = hit
    return hit[1]


def linear(x, w, bias=None, residual=None, act=None, out=None, rowgroup_bias=None, rows_per_group=0, step_ptr=None,
           residual_row_mod=0, rowstat=False, ln=None):
    """x [..., K] (row stride = K) @ w[N(or 2N for geglu), K]^T -> [..., N].
    rowstat: also emit the row statistics of the result (for a folded LayerNorm in the NEXT Linear; retrieved with rowstat_of).
    ln = (gamma, beta, eps): LayerNorm(x) first -- only valid when rowstat_of(x) exists (the caller checks ln_foldable)."""
    _req(x, "linear.x", w.dtype)
    _req(w, "linear.w")
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0] // 2 if act in ("geglu", "geglu_tanh") else w.shape[0]
    x2 = x.reshape(M, K)
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=w.dtype, device=x.device)
    r2 = None if residual is None else _req(residual, "linear.residual", w.dtype).reshape(-1, N)
    fold = rs_out = None
    if ln is not None:
        rs = rowstat_of(x)
        if rs is None:
            raise RuntimeError("linear(ln=...): no row statistics for x (use fused_linear, which falls back to apad_layernorm)")
        w, cs, bb = _ln_folded(w, bias, ln[0], ln[1])
        bias, fold = None, (rs, cs, bb, ln[2])
    if rowstat and LN_FOLD and N % 64 == 0 and N not in RP_K and act in (None, "none") and w.dtype in FUSED_DTYPES:
        rs_out = torch.empty(M, N // 64, 2, dtype=torch.float32, device=x.device)
    gemm(x2, w, M=M, N=N, K=K, lda=x2.stride(0), out=out, ldo=N, bias=bias, residual=r2, ldr=N, act=act,
         rowgroup_bias=rowgroup_bias, ld_rg=(rowgroup_bias.stride(0) if rowgroup_bias is not None else 0),
         rows_per_group=rows_per_group, step_ptr=step_ptr, ldw=w.stride(0), residual_row_mod=residual_row_mod,
         rowstat_out=rs_out, ln_fold=fold)
    if rs_out is not None:
        out._apad_rowstat = rs_out
    elif hasattr(out, "_apad_rowstat"):  # (a reused ``out=`` buffer must not keep the statistics of what it held before)
        del out._apad_rowstat
    return out


def linear2(xa, xb, w, bias=None, out=None):
    """Linear over the channel concatenation [xa | xb] without materialising it: xa [Ba, T, Ca], xb [Bb, T, Cb] (Ca % 64 == 0),
    w [N, Ca + Cb] -> [B, T, N] with B = max(Ba, Bb); the smaller batch is read modulo (a skip tensor of the CFG-shared prefix holds
    one row set for both halves of the batch).  The 1x1 shortcut convolution of an up-block resnet (modeling_audioldm2.py:1488)."""
    _req(xa, "linear2.xa", w.dtype)
    _req(xb, "linear2.xb", w.dtype)
    Ba, T, Ca = xa.shape
    Bb, Tb, Cb = xb.shape
    B = max(Ba, Bb)
    if Tb != T or B % Ba or B % Bb or Ca % 64 or w.shape[1] != Ca + Cb or not (xa.is_contiguous() and xb.is_contiguous()):
        raise ValueError(f"linear2: xa {tuple(xa.shape)}, xb {tuple(xb.shape)}, w {tuple(w.shape)}")
    N = w.shape[0]
    if out is None:
        out = torch.empty(B, T, N, dtype=w.dtype, device=xa.device)
    gemm(xa, w, M=B * T, N=N, K=Ca + Cb, lda=Ca, out=out, ldo=N, bias=bias, ldw=w.stride(0), a2=xb, lda2=Cb, k_split=Ca,
         a_row_mod=(Ba * T if Ba != B else 0), a2_row_mod=(Bb * T if Bb != B else 0))
    return out


def ln_foldable(x, w):
    """a LayerNorm in front of this Linear can be folded: 16-bit, statistics available, a width the row-panel kernel does not cover"""
    return LN_FOLD and x.dtype in FUSED_DTYPES and not rp_ok(x) and rowstat_of(x) is not None and w.shape[1] == x.shape[-1]


RP_K = (256, 384)  # reduction dims the row-panel kernel covers
CGEMM_LINEAR = False  # (measured in round 3: no gain on the HBM-bound to_out / proj launches; module attribute only)
CGEMM_MIN_M = 16000  # the row threshold of csrc/cgemm.hip
FUSED_DTYPES = (torch.bfloat16, torch.float16)  # the fused kernels (row-panel, feed-forward, cross-attention) are 16-bit only;
# the fp32 precision mode (exact-f32 MFMA, csrc/f32_ops.hip) runs the un-fused apad_layernorm / apad_gemm / apad_attention chain


def rp_ok(x, K=None):
    """row-panel envelope: reduction dim in RP_K and a 16-bit storage type"""
    return (x.shape[-1] if K is None else K) in RP_K and x.dtype in FUSED_DTYPES


def rowpanel(x, w, segs, ln=None, residual=None, act=None, vt_geom=None):
    """Fused projection: x [M,K] (K in RP_K) @ w^T with up to 3 output column segments.
    segs: list of (out_tensor, bias_or_None, n_cols, "row" | "vt").  ln = (gamma, beta, eps) applies LayerNorm to x
    first.  vt_geom = (heads, head_dim, L, Lpad) for "vt" segments.  Returns the list of outputs."""
    _req(x, "rowpanel.x", w.dtype)
    K = x.shape[-1]
    M = x.numel() // K
    d = L.RpDesc()
    d.x, d.w = x.data_ptr(), w.data_ptr()
    if ln is not None:
        d.ln_gamma, d.ln_beta, d.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
    d.residual = _ptr(residual)
    d.M, d.lda, d.ldw, d.ldr = M, x.reshape(M, K).stride(0), w.stride(0), (residual.shape[-1] if residual is not None else 0)
    d.K, d.epilogue, d.dtype, d.n_segments = K, _EPI[act], _DT[w.dtype], len(segs)
    if vt_geom is not None:
        d.heads, d.head_dim, d.L, d.Lpad = vt_geom
    for i, (out, bias, n_cols, mode) in enumerate(segs):
        d.seg[i].out, d.seg[i].bias = out.data_ptr(), _ptr(bias)
        d.seg[i].ldo = out.shape[-1] if mode == "row" else 0
        d.seg[i].n_cols, d.seg[i].mode = n_cols, (L.OUT_ROWMAJOR if mode == "row" else L.OUT_VT)
    L.check(L.lib().apad_rowpanel_gemm(C.byref(d), _stream()), "apad_rowpanel_gemm")
    return [s[0] for s in segs]


def fused_linear(x, w, bias=None, ln=None, residual=None, act=None, out=None, rowstat=False):
    """LayerNorm? -> Linear -> activation? (+ residual).  One row-panel launch when K is in its envelope; otherwise apad_gemm with
    the LayerNorm folded in when the kernel that produced x left its row statistics (rowstat=True on that call), else
    apad_layernorm + apad_gemm."""
    K = x.shape[-1]
    N = w.shape[0] // 2 if act == "geglu" else w.shape[0]
    if (CGEMM_LINEAR and ln is None and act is None and x.dtype in FUSED_DTYPES and N % 128 == 0 and K % 64 == 0
            and x.numel() // K >= CGEMM_MIN_M):
        # plain projections of the large levels (to_out + residual, proj_in / proj_out at >= 16000 rows): apad_gemm's big-tile
        # LDS-DMA kernel (csrc/cgemm.hip) instead of the weight-stationary row-panel kernel
        return linear(x, w, bias, residual=residual, out=out)
    if rp_ok(x, K) and N % 64 == 0:
        if out is None:
            out = torch.empty(*x.shape[:-1], N, dtype=w.dtype, device=x.device)
        rowpanel(x, w, [(out, bias, N, "row")], ln=ln, residual=residual, act=act)
        return out
    if ln is not None and ln_foldable(x, w):
        return linear(x, w, bias, residual=residual, act=act, out=out, rowstat=rowstat, ln=ln)
    if ln is not None:
        x = layer_norm(x, ln[0], ln[1], ln[2])
    return linear(x, w, bias, residual=residual, act=act, out=out, rowstat=rowstat)


XATTN_C, XATTN_HEADS, XATTN_MAXL = 256, 8, 64  # envelope of apad_fused_cross_attention
XATTN_MAXL2 = 128  # ... plus the adapter's 8 text + 128 audio keys (pooling 2: the timbre / accompaniment presets), unmasked
XATTN_LONG2 = 512  # ... plus 8 text + 64 n <= 512 audio keys on the chunked form (pooling 1 and the mixed poolings of the sweep), unmasked


def xattn_lengths_ok(L1, L2=0, masked=False):
    """key counts apad_fused_cross_attention has a kernel for: <= 64 per segment, or exactly 8 + 128, or 8 + 64 n <= 512 (both without a key bias)"""
    return L1 <= XATTN_MAXL and (L2 <= XATTN_MAXL or (L1 == 8 and not masked and (L2 == XATTN_MAXL2 or (XATTN_MAXL2 < L2 <= XATTN_LONG2 and L2 % 64 == 0))))


def xattn_pack_weight(w, ln=None):
    """[256, 256] nn.Linear weight -> the fragment-major packing apad_fused_cross_attention keeps in registers (128 KB).
    ln = (gamma, beta, eps) -- to_q behind a LayerNorm: returns (packing of W' = round(W * gamma), q_fold fp32 [2, 256] = row sums of W', W . beta):
    the kernel applies the LayerNorm by algebra on the q accumulators (apad_xattn_desc::q_fold)"""
    _req(w, "xattn_pack_weight.w")
    if ln is not None:
        wf = w.detach().float()
        wg = (wf * ln[0].detach().float()).to(w.dtype).contiguous()
        q_fold = torch.stack([wg.float().sum(1), wf @ ln[1].detach().float()], 0).contiguous()
        return xattn_pack_weight(wg), q_fold
    if tuple(w.shape) != (XATTN_C, XATTN_C) or w.dtype not in FUSED_DTYPES:
        raise ValueError(f"xattn_pack_weight: weight {tuple(w.shape)} {w.dtype} outside the kernel envelope")
    out = torch.empty(XATTN_C * XATTN_C, dtype=w.dtype, device=w.device)
    L.check(L.lib().apad_xattn_pack_weight(w.data_ptr(), out.data_ptr(), w.stride(0), _DT[w.dtype], _stream()), "apad_xattn_pack_weight")
    return out


def xattn_pack_kv(k, vt, Lk):
    """k [B, Lk, 256], vt [B, 8, 32, Lpad] (per-head transposed values) -> fragment-major packing of one key segment"""
    _req(k, "xattn_pack_kv.k")
    _req(vt, "xattn_pack_kv.vt", k.dtype)
    B = k.shape[0]
    if k.shape[-1] != XATTN_C or Lk > XATTN_LONG2 or tuple(vt.shape[:3]) != (B, XATTN_HEADS, XATTN_C // XATTN_HEADS) or not vt.is_contiguous():
        raise ValueError(f"xattn_pack_kv: k {tuple(k.shape)}, vt {tuple(vt.shape)}, Lk={Lk} outside the kernel envelope")
    nbytes = L.lib().apad_xattn_packed_kv_bytes(B, Lk)
    out = torch.empty(nbytes // k.element_size(), dtype=k.dtype, device=k.device)
    L.check(L.lib().apad_xattn_pack_kv(k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, Lk, vt.shape[-1], k.stride(0), k.stride(1),
                                       vt.stride(0), _DT[k.dtype], _stream()), "apad_xattn_pack_kv")
    return out


def fused_cross_attention(x, wq_packed, wo_packed, bo, kv1_packed, L1, heads, ln=None, key_bias=None, kv2_packed=None, L2=0,
                          scale2=0.0, out=None, q_fold=None):
    """out = x + to_out(A(q, k1, v1, bias) [+ scale2 * A(q, k2, v2)]) + bo with q = to_q(LayerNorm(x)): the whole
    cross-attention sub-layer in one launch.  x [B, N, C]; weights from xattn_pack_weight, K/V from xattn_pack_kv.  With ln: wq_packed and
    q_fold = xattn_pack_weight(to_q.weight, ln) (the LayerNorm is folded into the projection)."""
    _req(x, "fused_cross_attention.x", wq_packed.dtype)
    B, N, Cc = x.shape
    if Cc != XATTN_C or heads != XATTN_HEADS or not xattn_lengths_ok(L1, L2, key_bias is not None) or x.dtype not in FUSED_DTYPES:
        raise ValueError(f"fused_cross_attention: C={Cc} heads={heads} L1={L1} L2={L2} outside the kernel envelope")
    if not x.is_contiguous():
        raise ValueError("fused_cross_attention.x: must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    d = L.XattnDesc()
    d.x, d.wq_packed, d.wo_packed, d.bo, d.kv1_packed, d.out = (x.data_ptr(), wq_packed.data_ptr(), wo_packed.data_ptr(), _ptr(bo),
                                                                 kv1_packed.data_ptr(), out.data_ptr())
    if ln is not None:
        if q_fold is None or q_fold.dtype != torch.float32 or tuple(q_fold.shape) != (2, Cc) or not q_fold.is_contiguous():
            raise ValueError("fused_cross_attention(ln=...): needs q_fold = the second result of xattn_pack_weight(to_q.weight, ln)")
        d.ln_gamma, d.ln_beta, d.ln_eps, d.q_fold = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2]), q_fold.data_ptr()
    d.key_bias = _ptr(key_bias)
    d.B, d.N, d.C, d.heads, d.L1 = B, N, Cc, heads, L1
    if L2 > 0:
        d.kv2_packed, d.L2 = kv2_packed.data_ptr(), L2
    d.dtype, d.softmax_scale, d.scale2 = _DT[x.dtype], 1.0 / math.sqrt(Cc // heads), float(scale2)
    L.check(L.lib().apad_fused_cross_attention(C.byref(d), _stream()), "apad_fused_cross_attention")
    return out


# envelope of apad_cross_attention_rows: 8 heads, C = 384, <= 64 keys per segment (...
XROWS_C, XROWS_MAXL = (384,), 64
XROWS_MAXL2 = 512                 # ... <= 512 in the second segment beside <= 32 in the first: the adapter's 8 text + 128 audio keys; 64-key chunks with a running maximum above 128)


def xrows_ok(C_, heads, L1, L2=0):
    return (C_ in XROWS_C and heads == XATTN_HEADS and 1 <= L1 <= XROWS_MAXL
            and (0 <= L2 <= XROWS_MAXL or (L2 <= XROWS_MAXL2 and L1 <= 32)))


def xrows_pack_weight(w):
    """[C, C] projection weight -> the fragment-major packing apad_cross_attention_rows reads: one contiguous KB per MFMA operand fragment
    (row tile of 32 output features x k-step of 16): w.view(C/32, 32, C/16, 2, 8) -> [row tile][k-step][half][row][8]"""
    N, K = w.shape
    if N % 32 or K % 16:
        raise ValueError(f"xrows_pack_weight: {tuple(w.shape)}")
    return w.detach().reshape(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


ROWS_KV_PACKED = True  # the processors hand the row-tile cross-attention kernels fragment-packed key / value sets (False: the (k, vt) pair; module attribute only)


class RowsKV:
    """One key / value set of a cross-attention site in apad_rows_pack_kv's layout: every MFMA operand fragment of every (sample, head) one contiguous
    KB (``rows_pack_kv``).  ``cross_attention_rows`` / ``hs_attention`` take it in place of the (k, vt) pair of ops.attention's layout."""
    __slots__ = ("data", "B", "L", "heads", "head_dim")

    def __init__(self, data, B, L, heads, head_dim):
        self.data, self.B, self.L, self.heads, self.head_dim = data, B, L, heads, head_dim

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def shape(self):  # (batch, keys, channels): what the routers ask of a key tensor
        return (self.B, self.L, self.heads * self.head_dim)


def rows_pack_kv(k, vt):
    """k [B, L, C] row-major, vt [B, heads, d, Lpad] (ops.linear / ops.linear_vt of the hoisted projections) -> RowsKV: the fragment-major packing the
    row-tile cross-attention kernels of the 384- and 640-wide levels read with whole-line loads (a fragment load from the row-major pair touches 32 - 64
    cache lines, 8 from the packing).  Done once per site with the hoisted projections."""
    _req(k, "rows_pack_kv.k")
    _req(vt, "rows_pack_kv.vt", k.dtype)
    B, Lk, Cc = k.shape
    heads, hd = vt.shape[1], vt.shape[2]
    if k.dtype not in FUSED_DTYPES or not k.is_contiguous() or not vt.is_contiguous() or vt.shape[0] != B or heads * hd != Cc or hd % 16 or Lk < 1 \
            or vt.shape[3] < Lk or vt.shape[3] % 32:
        raise ValueError(f"rows_pack_kv: k {tuple(k.shape)} {k.dtype}, vt {tuple(vt.shape)} outside the packing's envelope")
    nbytes = L.lib().apad_rows_packed_kv_bytes(B, heads, hd, Lk)
    out = torch.empty(nbytes // k.element_size(), dtype=k.dtype, device=k.device)
    L.check(L.lib().apad_rows_pack_kv(k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, heads, hd, Lk, vt.shape[3], _DT[k.dtype], _stream()), "apad_rows_pack_kv")
    return RowsKV(out, B, Lk, heads, hd)


def _rows_kv_args(k, vt, B, heads, hd, dtype, what):
    """(k pointer, vt pointer or 0, L, Lpad) of one segment given as a RowsKV (vt = None) or as the (k, vt) pair of ops.attention's layout"""
    if isinstance(k, RowsKV):
        if vt is not None or (k.B, k.heads, k.head_dim) != (B, heads, hd) or k.dtype != dtype:
            raise ValueError(f"{what}: a RowsKV stands for the (k, vt) pair and must have the batch / heads / type of x")
        return k.data.data_ptr(), 0, k.L, 0
    for t in (k, vt):
        if t is None or not t.is_contiguous() or t.dtype != dtype:
            raise ValueError(f"{what}: must be contiguous {dtype}")
    if k.shape[0] != B or tuple(vt.shape[:3]) != (B, heads, hd):
        raise ValueError(f"{what}: key / value sets must have the batch of x")
    return k.data_ptr(), vt.data_ptr(), k.shape[1], vt.shape[-1]


def cross_attention_rows(x, wq_packed, wo_packed, bo, k1, vt1, heads, ln=None, key_bias=None, k2=None, vt2=None, scale2=0.0, out=None):
    """The fused cross-attention sub-layer at the 384-wide level: out = x + to_out(A(q, k1, v1, bias) [+ scale2 * A(q, k2, v2)]) + bo,
    q = to_q(LayerNorm(x)), one launch (csrc/attention.hip, xattn_rows_kernel).  x [B, N, C]; k* [B, L, C] row-major, vt* [B, heads, d,
    Lpad] as ops.attention takes them, or k* = a RowsKV (rows_pack_kv) and vt* = None; weights from xrows_pack_weight."""
    _req(x, "cross_attention_rows.x", wq_packed.dtype)
    B, N, Cc = x.shape
    L1, L2 = k1.shape[1], (0 if k2 is None else k2.shape[1])
    if not xrows_ok(Cc, heads, L1, L2) or x.dtype not in FUSED_DTYPES:
        raise ValueError(f"cross_attention_rows: C={Cc} heads={heads} L1={L1} L2={L2} {x.dtype} outside the kernel envelope")
    if not x.is_contiguous():
        raise ValueError(f"cross_attention_rows.x: must be contiguous {x.dtype}")
    if out is None:
        out = torch.empty_like(x)
    d = L.XrowsDesc()
    d.x, d.wq_packed, d.wo_packed, d.bo, d.out = x.data_ptr(), wq_packed.data_ptr(), wo_packed.data_ptr(), _ptr(bo), out.data_ptr()
    d.k1, d.vt1, d.L1, d.Lpad1 = _rows_kv_args(k1, vt1, B, heads, Cc // heads, x.dtype, "cross_attention_rows.k1 / vt1")
    if ln is not None:
        d.ln_gamma, d.ln_beta, d.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
    d.key_bias = _ptr(key_bias)
    d.B, d.N, d.C, d.heads = B, N, Cc, heads
    if L2 > 0:
        d.k2, d.vt2, d.L2, d.Lpad2 = _rows_kv_args(k2, vt2, B, heads, Cc // heads, x.dtype, "cross_attention_rows.k2 / vt2")
    d.dtype, d.softmax_scale, d.scale2 = _DT[x.dtype], 1.0 / math.sqrt(Cc // heads), float(scale2)
    L.check(L.lib().apad_cross_attention_rows(C.byref(d), _stream()), "apad_cross_attention_rows")
    return out


# ---- LayerNorm + q|k|v + self-attention in one launch for the two large levels (csrc/attention.hip, sattn_fused_kernel) ----
SATTN_FUSED = _os.environ.get("APAD_SATTN_FUSED", "1") == "1"  # A/B switch (read once): 0 = row-panel LN + q|k|v launch, then apad_attention
SATTN_ENVELOPE = {256: (513, 1024), 384: (129, 256)}  # C -> token counts routed to it (shorter sequences keep the two-launch route)


def sattn_ok(x, heads):
    """[B, N, C] 16-bit contiguous rows inside apad_self_attention_fused's routed envelope"""
    if not (SATTN_FUSED and x.dim() == 3 and x.dtype in FUSED_DTYPES and x.is_contiguous() and heads == 8):
        return False
    lo_hi = SATTN_ENVELOPE.get(x.shape[-1])
    return lo_hi is not None and lo_hi[0] <= x.shape[1] <= lo_hi[1]


def sattn_pack(wq, wk, wv, ln, heads):
    """to_q / to_k / to_v [C, C] behind LayerNorm ln = (gamma, beta, eps) -> (per-head fragment packing [H][T3 = ceil(3 d / 32) tiles][C/16][64][8] with
    gamma and the softmax scale log2(e) / sqrt(d) folded in, fp32 [H][2][T3 * 32]: row sums of the ROUNDED packed weights, then W . beta) -- see
    apad_self_attention_fused in include/apadapter_hip.h"""
    Cc = wq.shape[0]
    d = Cc // heads
    T3 = (3 * d + 31) // 32
    qs = LOG2E / math.sqrt(d)
    gamma, beta = ln[0].detach().float(), ln[1].detach().float()
    wf = torch.stack([wq.detach().float() * qs, wk.detach().float(), wv.detach().float()], 0)  # [3, C, C]
    bb = wf @ beta                                                                               # [3, C]
    wg = (wf * gamma).to(wq.dtype)
    cs = wg.float().sum(-1)
    # per head: its 3 d rows [q | k | v] packed densely, zero-padded to T3 row tiles
    per_head = lambda t: t.reshape(3, heads, d, *t.shape[2:]).transpose(0, 1).reshape(heads, 3 * d, *t.shape[2:])
    pad = T3 * 32 - 3 * d
    wh = torch.nn.functional.pad(per_head(wg), (0, 0, 0, pad))                                   # [H, T3 * 32, C]
    w = wh.reshape(heads, T3, 32, Cc // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)
    csbb = torch.stack([torch.nn.functional.pad(per_head(cs), (0, pad)), torch.nn.functional.pad(per_head(bb), (0, pad))], 1).contiguous().reshape(-1)
    return w, csbb


def self_attention_fused(x, w_packed, csbb, heads, ln_eps, out=None):
    """O [B, N, C] = the heads' softmax(q k^T / sqrt(d)) v with q | k | v = Linear(LayerNorm(x)), one launch (apad_self_attention_fused; weights
    from sattn_pack)"""
    _req(x, "self_attention_fused.x", w_packed.dtype)
    B, N, Cc = x.shape
    if x.dtype not in FUSED_DTYPES or not x.is_contiguous() or csbb.dtype != torch.float32:
        raise ValueError("self_attention_fused: x must be contiguous 16-bit, colsum / bias fp32")
    if out is None:
        out = torch.empty_like(x)
    L.check(L.lib().apad_self_attention_fused(x.data_ptr(), w_packed.data_ptr(), csbb.data_ptr(), out.data_ptr(), B, N, Cc, heads, float(ln_eps),
                                              _DT[x.dtype], _stream()), "apad_self_attention_fused")
    return out


# ---- the 64-token level's attention sub-layers (csrc/hsattn.hip): head-sliced LayerNorm + projections + attention, then to_out + residual ----
HS_C, HS_HEADS, HS_MAXN = 640, 8, 64
HS_ATTN = _os.environ.get("APAD_HS_ATTN", "1") == "1"  # A/B switch (read once): 0 = the LN-folded q|k|v GEMM -> attention -> to_out chain


def hs_ok(x, heads, n_q_rows):
    """envelope of apad_hs_attention / apad_hs_out: [B, <= 64, 640], 8 heads, 16-bit, a square to_q"""
    return (HS_ATTN and x.dim() == 3 and x.shape[-1] == HS_C and x.shape[1] <= HS_MAXN and heads == HS_HEADS and n_q_rows == HS_C
            and x.dtype in FUSED_DTYPES and x.is_contiguous())


HS_FF2 = True  # its second Linear through apad_hs_ff2 (False: the tiled GEMM); module attribute only
HS_FF = True  # the feed-forward of that level through apad_hs_geglu (+ apad_hs_ff2); follows HS_ATTN (unet.FeedForward); module attribute only


def hs_rows_ok(x):
    """[B, <= 64, 640] contiguous 16-bit rows: what the row-tile kernels of csrc/hsattn.hip take"""
    return x.dim() == 3 and x.shape[-1] == HS_C and x.shape[1] <= HS_MAXN and x.dtype in FUSED_DTYPES and x.is_contiguous()


def hs_cross_lengths_ok(L1, L2=0):
    return 1 <= L1 <= 64 and (0 <= L2 <= 64 or (L2 <= XROWS_MAXL2 and L1 <= 32))


def _hs_frag(w):
    """[..., 160 rows (5 tiles x 32), 640] -> [..., tile, k-step, half, row, 8]: one contiguous KB per MFMA operand fragment"""
    lead = w.shape[:-2]
    n = len(lead)
    w = w.reshape(*lead, 5, 32, 40, 2, 8)
    return w.permute(*range(n), n, n + 2, n + 3, n + 1, n + 4)


def hs_pack_rows(w, ln=None, bias=None, scale=1.0):
    """[640, 640] projection weight (to_q of a cross-attention, to_out[0]) -> (fragment packing per 160-row quarter, fp32 bias [640] or None).
    ln = (gamma, beta, eps): the LayerNorm in front of the projection is folded in -- W * gamma (rounded once) and W . beta + bias as an
    fp32 bias, the algebra of ``_ln_folded``; ``scale`` multiplies both (the softmax scale of a pre-scaled q)."""
    wf = w.detach().float()
    bb = None if bias is None else bias.detach().float()
    if ln is not None:
        bb = wf @ ln[1].detach().float() + (0.0 if bb is None else bb)
        wf = wf * ln[0].detach().float()
    if scale != 1.0:
        wf = wf * scale
        bb = None if bb is None else bb * scale
    pk = _hs_frag(wf.to(w.dtype).reshape(4, 160, HS_C)).contiguous().reshape(-1)
    return pk, (None if bb is None else bb.contiguous())


def hs_pack_qkv(wq, wk, wv, ln=None, q_scale=1.0):
    """to_q / to_k / to_v [640, 640] of a self-attention -> ([4 head pairs][15 row tiles: 5 q, 5 k, 5 v][40][64][8] packing, fp32 bias
    [4][480] in the same row order or None); the LayerNorm folded in as in hs_pack_rows, the to_q rows scaled by ``q_scale``"""
    ws = [w.detach().float() for w in (wq, wk, wv)]
    bb = None
    if ln is not None:
        beta, gamma = ln[1].detach().float(), ln[0].detach().float()
        bb = [w @ beta for w in ws]
        ws = [w * gamma for w in ws]
    if q_scale != 1.0:
        ws[0] = ws[0] * q_scale
        if bb is not None:
            bb[0] = bb[0] * q_scale
    w = torch.stack(ws, 0).to(wq.dtype).reshape(3, 4, 160, HS_C).permute(1, 0, 2, 3)  # [pair][which][160][640]
    pk = _hs_frag(w).contiguous().reshape(-1)
    if bb is not None:
        bb = torch.stack(bb, 0).reshape(3, 4, 160).permute(1, 0, 2).contiguous().reshape(-1)
    return pk, bb


def hs_pack_geglu(w1, b1, ln=None):
    """GEGLU projection [5120, 640] (+ bias [5120]) -> ([4 hidden quarters][20 tiles][value, gate][40][64][8] packing, fp32 bias [4][20][2][32]);
    ln = (gamma, beta, eps) folded in as in hs_pack_rows"""
    wf = w1.detach().float()
    bb = torch.zeros(wf.shape[0], device=wf.device) if b1 is None else b1.detach().float()
    if ln is not None:
        bb = wf @ ln[1].detach().float() + bb
        wf = wf * ln[0].detach().float()
    w = wf.to(w1.dtype).reshape(2, 4, 20, 32, 40, 2, 8).permute(1, 2, 0, 4, 5, 3, 6)  # [q][t][which][ks][half][row][8]
    return w.contiguous().reshape(-1), bb.reshape(2, 4, 20, 32).permute(1, 2, 0, 3).contiguous().reshape(-1)


def hs_geglu(x, w_packed, w_bias, normalize=True, ln_eps=1e-5, out=None):
    """H [B, N, 2560] = value * gelu(gate), [value | gate] = Linear(LayerNorm(x)) of the 64-token level's feed-forward in one launch (apad_hs_geglu;
    weights from hs_pack_geglu)"""
    _req(x, "hs_geglu.x", w_packed.dtype)
    B, N, Cc = x.shape
    if Cc != HS_C or N > HS_MAXN or x.dtype not in FUSED_DTYPES or not x.is_contiguous():
        raise ValueError(f"hs_geglu: x {tuple(x.shape)} {x.dtype} outside the kernel envelope")
    if out is None:
        out = torch.empty(B, N, 4 * Cc, dtype=x.dtype, device=x.device)
    L.check(L.lib().apad_hs_geglu(x.data_ptr(), w_packed.data_ptr(), _ptr(w_bias), out.data_ptr(), B, N, Cc, int(bool(normalize)), float(ln_eps),
                                  _DT[x.dtype], _stream()), "apad_hs_geglu")
    return out


def hs_pack_ff2(w2):
    """FF2 weight [640, 2560] -> fragment packing per 160-row output quarter: [4][5 tiles][160 k-steps][half][row][8]"""
    return w2.detach().reshape(4, 5, 32, 160, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)


def hs_ff2(h, w2_packed, bias, residual, rowstat=False, out=None):
    """out = (residual +) (h @ W2^T + bias) for h [B, <= 64, 2560] -> [B, N, 640] (apad_hs_ff2; W2 from hs_pack_ff2); rowstat as hs_out"""
    _req(h, "hs_ff2.h", w2_packed.dtype)
    B, N, K = h.shape
    if K != 4 * HS_C or N > HS_MAXN or h.dtype not in FUSED_DTYPES or not h.is_contiguous():
        raise ValueError(f"hs_ff2: h {tuple(h.shape)} outside the kernel envelope")
    if residual is not None:
        _req(residual, "hs_ff2.residual", h.dtype)
        if not residual.is_contiguous() or tuple(residual.shape) != (B, N, HS_C):
            raise ValueError(f"hs_ff2: residual {tuple(residual.shape)} must be contiguous [B, N, 640]")
    if out is None:
        out = torch.empty(B, N, HS_C, dtype=h.dtype, device=h.device)
    rs = torch.empty(B * N, 20, 2, dtype=torch.float32, device=h.device) if (rowstat and LN_FOLD) else None
    d = L.HsOutDesc()
    d.o, d.w_packed, d.bias, d.residual, d.out, d.rowstat_out = h.data_ptr(), w2_packed.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), _ptr(rs)
    d.B, d.N, d.C, d.dtype = B, N, HS_C, _DT[h.dtype]
    L.check(L.lib().apad_hs_ff2(C.byref(d), _stream()), "apad_hs_ff2")
    if rs is not None:
        out._apad_rowstat = rs
    elif hasattr(out, "_apad_rowstat"):
        del out._apad_rowstat
    return out


def hs_attention(x, w_packed, w_bias, *, self_attention, normalize=True, ln_eps=1e-5, k1=None, vt1=None, key_bias=None, k2=None, vt2=None,
                 scale2=0.0, q_prescaled=False, out=None):
    """O = heads' softmax attention of the 64-token level in ONE launch (apad_hs_attention): rows of x normalised (the LayerNorm's affine
    part lives in w_packed / w_bias: hs_pack_qkv / hs_pack_rows), q|k|v (self) or q (cross: k*, vt* = the hoisted sets in ops.attention's
    layout, or k* = a RowsKV and vt* = None) projected per head pair, both heads attended, O [B, N, 640] written.  to_out + residual: ``hs_out``."""
    _req(x, "hs_attention.x", w_packed.dtype)
    B, N, Cc = x.shape
    if Cc != HS_C or N > HS_MAXN or x.dtype not in FUSED_DTYPES or not x.is_contiguous():
        raise ValueError(f"hs_attention: x {tuple(x.shape)} {x.dtype} outside the kernel envelope")
    d = L.HsAttnDesc()
    d.x, d.w_packed, d.w_bias = x.data_ptr(), w_packed.data_ptr(), _ptr(w_bias)
    if w_bias is not None and w_bias.dtype != torch.float32:
        raise ValueError("hs_attention.w_bias: must be fp32")
    if not self_attention:
        L1, L2 = k1.shape[1], (0 if k2 is None else k2.shape[1])
        if not hs_cross_lengths_ok(L1, L2):
            raise ValueError(f"hs_attention: segment lengths {L1} / {L2} outside the kernel envelope")
        d.k1, d.vt1, d.L1, d.Lpad1 = _rows_kv_args(k1, vt1, B, HS_HEADS, 80, x.dtype, "hs_attention.k1 / vt1")
        d.key_bias = _ptr(key_bias)
        if L2 > 0:
            d.k2, d.vt2, d.L2, d.Lpad2 = _rows_kv_args(k2, vt2, B, HS_HEADS, 80, x.dtype, "hs_attention.k2 / vt2")
    if out is None:
        out = torch.empty_like(x)
    d.out = out.data_ptr()
    d.B, d.N, d.C, d.heads = B, N, Cc, HS_HEADS
    d.self_attention, d.q_prescaled, d.dtype, d.normalize = int(bool(self_attention)), int(bool(q_prescaled)), _DT[x.dtype], int(bool(normalize))
    d.ln_eps, d.softmax_scale, d.scale2 = float(ln_eps), 1.0 / math.sqrt(80.0), float(scale2)
    L.check(L.lib().apad_hs_attention(C.byref(d), _stream()), "apad_hs_attention")
    return out


def hs_out(o, wo_packed, bias, residual, rowstat=False, out=None):
    """out = (residual +) (o @ Wo^T + bias) for [B, <= 64, 640] (apad_hs_out; Wo from hs_pack_rows).  rowstat: the per-32-column row statistics a
    folded LayerNorm in the NEXT Linear reads (retrieved with rowstat_of)."""
    _req(o, "hs_out.o", wo_packed.dtype)
    B, N, Cc = o.shape
    if Cc != HS_C or N > HS_MAXN or o.dtype not in FUSED_DTYPES or not o.is_contiguous():
        raise ValueError(f"hs_out: o {tuple(o.shape)} outside the kernel envelope")
    if residual is not None:
        _req(residual, "hs_out.residual", o.dtype)
        if not residual.is_contiguous() or residual.shape != o.shape:
            raise ValueError(f"hs_out: residual {tuple(residual.shape)} must be contiguous with the shape of o")
    if out is None:
        out = torch.empty_like(o)
    rs = torch.empty(B * N, 20, 2, dtype=torch.float32, device=o.device) if (rowstat and LN_FOLD) else None
    d = L.HsOutDesc()
    d.o, d.w_packed, d.bias, d.residual, d.out, d.rowstat_out = o.data_ptr(), wo_packed.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(), _ptr(rs)
    d.B, d.N, d.C, d.dtype = B, N, Cc, _DT[o.dtype]
    L.check(L.lib().apad_hs_out(C.byref(d), _stream()), "apad_hs_out")
    if rs is not None:
        out._apad_rowstat = rs
    elif hasattr(out, "_apad_rowstat"):
        del out._apad_rowstat
    return out


MLP_C = (256,)  # envelope of apad_geglu_mlp
# the 64-token register-block form from packed weights (csrc/mlp3.hip): routed from MLP_PACKED_MIN_M rows -- 256-token workgroups, one per CU, so a
# launch needs most of a chip's worth of them (the CFG-shared prefix's 32 000-row launches keep the 128-token kernel)
MLP_PACKED = _os.environ.get("APAD_MLP_PACKED", "1") == "1"  # A/B switch (read once)
MLP_PACKED_MIN_M = 48000


def mlp_pack(w1, b1, w2):
    """GEGLU.proj [8C, C] (+ bias [8C]) and net[2] [C, 4C] -> (the packed stream, the fp32 bias table) of apad_geglu_mlp_packed"""
    _req(w1, "mlp_pack.w1")
    _req(w2, "mlp_pack.w2", w1.dtype)
    Cc = w2.shape[0]
    if Cc not in MLP_C or tuple(w1.shape) != (8 * Cc, Cc) or tuple(w2.shape) != (Cc, 4 * Cc) or w1.dtype not in FUSED_DTYPES or not (w1.is_contiguous() and w2.is_contiguous()):
        raise ValueError(f"mlp_pack: w1 {tuple(w1.shape)}, w2 {tuple(w2.shape)} {w1.dtype} outside the kernel envelope")
    if b1 is not None and (b1.dtype != w1.dtype or not b1.is_contiguous()):
        raise ValueError("mlp_pack.b1: must be contiguous and of the weights' dtype")
    wp = torch.empty(L.lib().apad_mlp_packed_bytes(Cc) // w1.element_size(), dtype=w1.dtype, device=w1.device)
    bp = torch.empty(L.lib().apad_mlp_packed_bias_floats(Cc), dtype=torch.float32, device=w1.device)
    L.check(L.lib().apad_mlp_pack(w1.data_ptr(), _ptr(b1), w2.data_ptr(), wp.data_ptr(), bp.data_ptr(), Cc, _DT[w1.dtype], _stream()), "apad_mlp_pack")
    return wp, bp


def geglu_mlp_packed(x, w_packed, b1_packed, b2, ln=None, out=None):
    """geglu_mlp from mlp_pack's weights: the same operator on the 64-token register-block kernel"""
    _req(x, "geglu_mlp_packed.x", w_packed.dtype)
    Cc = x.shape[-1]
    if Cc not in MLP_C or x.dtype not in FUSED_DTYPES or not x.is_contiguous() or b1_packed.dtype != torch.float32:
        raise ValueError(f"geglu_mlp_packed: x {tuple(x.shape)} {x.dtype} outside the kernel envelope")
    if out is None:
        out = torch.empty_like(x)
    d = L.MlpDesc()
    d.x, d.b2, d.out = x.data_ptr(), _ptr(b2), out.data_ptr()
    if ln is not None:
        d.ln_gamma, d.ln_beta, d.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
    d.M, d.C, d.dtype = x.numel() // Cc, Cc, _DT[x.dtype]
    L.check(L.lib().apad_geglu_mlp_packed(C.byref(d), w_packed.data_ptr(), b1_packed.data_ptr(), _stream()), "apad_geglu_mlp_packed")
    return out


def geglu_mlp(x, w1, b1, w2, b2, ln=None, out=None):
    """out = x + W2 . (value * gelu(gate)) + b2 with [value|gate] = W1 . LayerNorm(x) + b1 -- the whole FeedForward of a
    BasicTransformerBlock plus its residual in one launch (C in MLP_C)."""
    _req(x, "geglu_mlp.x", w1.dtype)
    Cc = x.shape[-1]
    if Cc not in MLP_C or w1.shape != (8 * Cc, Cc) or w2.shape != (Cc, 4 * Cc) or x.dtype not in FUSED_DTYPES:
        raise ValueError(f"geglu_mlp: C={Cc}, w1 {tuple(w1.shape)}, w2 {tuple(w2.shape)} outside the kernel envelope")
    if not (x.is_contiguous() and w1.is_contiguous() and w2.is_contiguous()):
        raise ValueError("geglu_mlp: operands must be contiguous")
    if out is None:
        out = torch.empty_like(x)
    d = L.MlpDesc()
    d.x, d.w1, d.b1, d.w2, d.b2, d.out = x.data_ptr(), w1.data_ptr(), _ptr(b1), w2.data_ptr(), _ptr(b2), out.data_ptr()
    if ln is not None:
        d.ln_gamma, d.ln_beta, d.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])
    d.M, d.C, d.dtype = x.numel() // Cc, Cc, _DT[x.dtype]
    L.check(L.lib().apad_geglu_mlp(C.byref(d), _stream()), "apad_geglu_mlp")
    return out


# LayerNorm + GEGLU projection of the 384-wide level's feed-forward on the 64-token register-block kernel (csrc/geglu3.hip): routed from
# GEGLU_PACKED_MIN_M rows (256-token tiles x 4 hidden quarters: 16 128 rows = 252 workgroups); follows MLP_PACKED; bit-equal to the row-panel launch
GEGLU_PACKED_C = (384,)
GEGLU_PACKED_MIN_M = 12000


def geglu_pack(w1, b1):
    """GEGLU.proj [8C, C] (+ bias [8C]) -> (the packed stream, the fp32 bias table) of apad_layernorm_geglu_packed"""
    _req(w1, "geglu_pack.w1")
    Cc = w1.shape[1]
    if Cc not in GEGLU_PACKED_C or w1.shape[0] != 8 * Cc or w1.dtype not in FUSED_DTYPES or not w1.is_contiguous():
        raise ValueError(f"geglu_pack: w1 {tuple(w1.shape)} {w1.dtype} outside the kernel envelope")
    if b1 is not None and (b1.dtype != w1.dtype or not b1.is_contiguous()):
        raise ValueError("geglu_pack.b1: must be contiguous and of the weights' dtype")
    wp = torch.empty(L.lib().apad_geglu_packed_bytes(Cc) // w1.element_size(), dtype=w1.dtype, device=w1.device)
    bp = torch.empty(L.lib().apad_geglu_packed_bias_floats(Cc), dtype=torch.float32, device=w1.device)
    L.check(L.lib().apad_geglu_pack(w1.data_ptr(), _ptr(b1), wp.data_ptr(), bp.data_ptr(), Cc, _DT[w1.dtype], _stream()), "apad_geglu_pack")
    return wp, bp


def layernorm_geglu_packed(x, w_packed, b1_packed, ln=None, out=None):
    """H [..., 4C] = value * gelu(gate), [value | gate] = Linear(LayerNorm(x)) from geglu_pack's weights (one launch)"""
    _req(x, "layernorm_geglu_packed.x", w_packed.dtype)
    Cc = x.shape[-1]
    if Cc not in GEGLU_PACKED_C or x.dtype not in FUSED_DTYPES or not x.is_contiguous() or b1_packed.dtype != torch.float32:
        raise ValueError(f"layernorm_geglu_packed: x {tuple(x.shape)} {x.dtype} outside the kernel envelope")
    if out is None:
        out = torch.empty(*x.shape[:-1], 4 * Cc, dtype=x.dtype, device=x.device)
    g, b, eps = (ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2])) if ln is not None else (None, None, 0.0)
    L.check(L.lib().apad_layernorm_geglu_packed(x.data_ptr(), g, b, w_packed.data_ptr(), b1_packed.data_ptr(), out.data_ptr(), x.numel() // Cc, Cc, eps,
                                                _DT[x.dtype], _stream()), "apad_layernorm_geglu_packed")
    return out


def linear_vt(x, w, B, Lk, heads, out_vt, bias=None):
    """Values projection stored per-head transposed: x [B*Lk, K] @ w[C,K]^T -> out_vt [B, heads, d, Lpad]
    (zero padded by the caller; only l < Lk is written)."""
    _req(x, "linear_vt.x", w.dtype)
    K = x.shape[-1]
    Cc = w.shape[0]
    x2 = x.reshape(B * Lk, K)
    gemm(x2, w, M=B * Lk, N=Cc, K=K, lda=x2.stride(0), out=out_vt, ldo=0, out_mode=L.OUT_VT, bias=bias,
         vt=(heads, Cc // heads, Lk, out_vt.shape[-1]), ldw=w.stride(0))
    return out_vt


def linear_qkv(x, w_qkv, B, Lk, heads, q, k, vt, bias=None, ln=None, v=None):
    """Fused q|k|v projection on the tiled kernel (any K): x [B*Lk, K] @ w_qkv[3C, K]^T -> q, k [B*Lk, C] row-major and
    vt [B, heads, d, Lpad] per-head transposed, one launch.  ln = (gamma, beta, eps): LayerNorm(x) folded in (needs rowstat_of(x)).
    v: optional [B*Lk, C] row-major copy of the values (same rounded numbers as vt)."""
    _req(x, "linear_qkv.x", w_qkv.dtype)
    fold = None
    if ln is not None:
        rs = rowstat_of(x)
        w_qkv, cs, bb = _ln_folded(w_qkv, bias, ln[0], ln[1])
        bias, fold = None, (rs, cs, bb, ln[2])
    K = x.shape[-1]
    C3 = w_qkv.shape[0]
    Cc = C3 // 3
    x2 = x.reshape(B * Lk, K)
    d = L.GemmDesc()
    d.a, d.w, d.out, d.out2, d.out3 = x2.data_ptr(), w_qkv.data_ptr(), q.data_ptr(), k.data_ptr(), vt.data_ptr()
    d.out4 = _ptr(v)  # v row-major [B*Lk, C] as well (the training step's backward reads it)
    d.bias = _ptr(bias)
    d.M, d.N, d.K, d.lda, d.ldw, d.ldo = B * Lk, C3, K, x2.stride(0), w_qkv.stride(0), Cc
    d.a_mode, d.epilogue, d.out_mode, d.dtype = L.A_PLAIN, L.EPI_NONE, L.OUT_QKV, _DT[w_qkv.dtype]
    d.heads, d.head_dim, d.L, d.Lpad = heads, Cc // heads, Lk, vt.shape[-1]
    if fold is not None:
        rs, cs, bb, eps = fold
        d.rowstat_in, d.ln_colsum, d.ln_bias, d.rowstat_in_tiles, d.ln_eps = rs.data_ptr(), cs.data_ptr(), bb.data_ptr(), rs.shape[-2], float(eps)
    L.check(L.lib().apad_gemm(C.byref(d), _stream()), "apad_gemm(qkv)")
    return q, k, vt


HCONV = True  # route eligible 3x3 convolutions through the halo-resident kernel (csrc/hconv.hip); a module attribute a test may flip
_halo_cache = {}


def conv_halo_eligible(Cin, Cout, Wout, stride, dtype, src_batch_mod=0, asym_pad=False):
    """the LAYER-level part of apad_hconv_try's envelope (never the row count): what decides whether the packed form is built"""
    wide = Cout % 128 == 0 and Wout in (2, 4, 8, 16)
    narrow = Cout in (8, 16) and Cin in (64, 128) and Wout in (4, 8, 16)  # conv_out: stationary weights (the narrow form)
    return HCONV and stride == 1 and not asym_pad and src_batch_mod == 0 and dtype in FUSED_DTYPES and Cin % 64 == 0 and (wide or narrow)


def conv_halo_weight(w_packed):
    """w_packed [Cout, 9 * Cin] -> apad_conv_halo_pack's form (same element count), cached per weight tensor version"""
    key = id(w_packed)
    sig = (w_packed.data_ptr(), w_packed._version, w_packed.dtype, w_packed.device, tuple(w_packed.shape))
    hit = _halo_cache.get(key)
    if hit is None or hit[0] != sig or hit[2]() is not w_packed:
        if len(_halo_cache) > 512:
            for k in [k for k, v in _halo_cache.items() if v[2]() is None]:
                del _halo_cache[k]
        wc = w_packed.detach().contiguous()
        Cout, K = wc.shape
        out = torch.empty(L.lib().apad_conv_halo_packed_bytes(Cout, K // 9) // 2, dtype=wc.dtype, device=wc.device)
        L.check(L.lib().apad_conv_halo_pack(wc.data_ptr(), out.data_ptr(), Cout, K // 9, _DT[wc.dtype], _stream()), "apad_conv_halo_pack")
        hit = (sig, out, _weakref.ref(w_packed))
        _halo_cache[key] = hit
    return hit[1]


def conv3x3(x, w_packed, bias, B, Hin, Win, stride=1, up=None, residual=None, rowgroup_bias=None, rows_per_group=0,
            step_ptr=None, src_batch_mod=0, out=None, asym_pad=False):
    """NHWC implicit-GEMM 3x3 convolution, padding 1.  x [Bsrc, Hin*Win, Cin]; w_packed [Cout, 9*Cin] in
    (ky, kx, cin) order; up=(Hup, Wup) applies a nearest-neighbour upsample to the source first.  asym_pad: the zero row /
    column only at the bottom / right (F.pad(x, (0,1,0,1)) + an un-padded conv: the VAE encoder's down-sampler).
    Returns ([B, Hout*Wout, Cout], Hout, Wout)."""
    _req(x, "conv3x3.x", w_packed.dtype)
    Cin = x.shape[-1]
    Cout = w_packed.shape[0]
    Hs, Ws = (up if up is not None else (Hin, Win))
    Hout = (Hs + (1 if asym_pad else 2) - 3) // stride + 1
    Wout = (Ws + (1 if asym_pad else 2) - 3) // stride + 1
    M = B * Hout * Wout
    if out is None:
        out = torch.empty(B, Hout * Wout, Cout, dtype=x.dtype, device=x.device)
    wh = ws = None
    if conv_halo_eligible(Cin, Cout, Wout, stride, w_packed.dtype, src_batch_mod, asym_pad):
        wh = conv_halo_weight(w_packed)
        nws = L.lib().apad_conv_halo_workspace_bytes(M, Cout, Cin, Wout)  # (the K-sliced layers of the 64-token level: fp32 slabs)
        if nws:
            ws = torch.empty(nws // 4, dtype=torch.float32, device=x.device)
    gemm(x, w_packed, M=M, N=Cout, K=9 * Cin, lda=0, out=out, ldo=Cout, bias=bias, w_halo=wh, workspace=ws,
         residual=residual, ldr=Cout, rowgroup_bias=rowgroup_bias,
         ld_rg=(rowgroup_bias.stride(0) if rowgroup_bias is not None else 0), rows_per_group=rows_per_group,
         step_ptr=step_ptr, a_mode=L.A_CONV3X3, asym_pad=asym_pad,
         conv=(Hin, Win, Cin, Hout, Wout, stride, (up[0] if up else 0), (up[1] if up else 0), src_batch_mod))
    return out, Hout, Wout


def conv1d(x, w_packed, bias, taps, dilation=1, transposed_stride=0, pad=None, pre_slope=None, residual=None, act=None, out=None):
    """HiFi-GAN convolutions over channels-last x [B, T, Cin] as ONE implicit GEMM: nn.Conv1d(k = taps, dilation, "same" padding
    (k*d - d)/2) or, with ``transposed_stride`` = s, nn.ConvTranspose1d(k = taps, stride s, padding (k - s)/2) -> [B, T*s, Cout].
    w_packed [Cout, taps*Cin] in (tap, cin) order.  pre_slope: leaky_relu(x, pre_slope) applied while the input is gathered (the
    vocoder's pre-activations); act: None | "tanh"; residual [B, T_out, Cout] added after."""
    _req(x, "conv1d.x", w_packed.dtype)
    B, T, Cin = x.shape
    Cout = w_packed.shape[0]
    s_ = int(transposed_stride)
    if pad is None:
        pad = (taps - s_) // 2 if s_ else (taps * dilation - dilation) // 2
    Tout = (T - 1) * s_ - 2 * pad + taps if s_ else T + 2 * pad - dilation * (taps - 1)
    if out is None:
        out = torch.empty(B, Tout, Cout, dtype=x.dtype, device=x.device)
    d = L.GemmDesc()
    d.a, d.w, d.out, d.bias, d.residual = x.data_ptr(), w_packed.data_ptr(), out.data_ptr(), _ptr(bias), _ptr(residual)
    d.M, d.N, d.K, d.lda, d.ldw, d.ldo, d.ldr = B * Tout, Cout, taps * Cin, 0, w_packed.stride(0), Cout, Cout
    d.a_mode, d.epilogue, d.out_mode, d.dtype = L.A_CONV1D, _EPI[act], L.OUT_ROWMAJOR, _DT[w_packed.dtype]
    d.Hin, d.Hout, d.Cin, d.stride = T, Tout, Cin, max(s_, 1)
    d.taps, d.dilation, d.pad, d.transposed = taps, dilation, pad, 1 if s_ else 0
    if pre_slope is not None:
        d.a_pre_act, d.a_pre_slope = 1, float(pre_slope)
    d.rows_per_group = 1
    L.check(L.lib().apad_gemm(C.byref(d), _stream()), "apad_gemm(conv1d)")
    return out


def mix3(a, b, c, scale, out=None):
    """(a + b + c) * scale, element-wise"""
    _req(a, "mix3.a")
    if out is None:
        out = torch.empty_like(a)
    L.check(L.lib().apad_mix3(a.data_ptr(), b.data_ptr(), c.data_ptr(), out.data_ptr(), a.numel(), float(scale), _DT[a.dtype], _stream()),
            "apad_mix3")
    return out


def softmax_rows(x, scale=1.0, out=None, bias=None):
    """softmax(scale * x + bias) over the last dim of x [..., N] (row-contiguous), fp32 statistics; bias: fp32, same shape"""
    _req(x, "softmax_rows.x")
    N = x.shape[-1]
    x2 = x.reshape(-1, N)
    if out is None:
        out = torch.empty_like(x2)
    b2 = None
    if bias is not None:
        b2 = _req(bias, "softmax_rows.bias", torch.float32).reshape(-1, N)
        if b2.shape[0] != x2.shape[0]:
            raise ValueError(f"softmax_rows: bias {tuple(bias.shape)} does not match x {tuple(x.shape)}")
    L.check(L.lib().apad_softmax_rows(x2.data_ptr(), _ptr(b2), out.data_ptr(), x2.shape[0], N, x2.stride(0), b2.stride(0) if b2 is not None else 0,
                                      out.stride(0), float(scale), _DT[x.dtype], _stream()), "apad_softmax_rows")
    return out.view(x.shape)


def rms_norm(x, gamma, eps):
    """T5LayerNorm: x * rsqrt(mean(x^2) + eps) * gamma over the last dim"""
    _req(x, "rms_norm.x", gamma.dtype)
    Cc = x.shape[-1]
    x2 = x.reshape(-1, Cc)
    out = torch.empty_like(x2)
    L.check(L.lib().apad_rmsnorm(x2.data_ptr(), gamma.data_ptr(), out.data_ptr(), x2.shape[0], Cc, x2.stride(0), Cc, float(eps), 0, _DT[x.dtype],
                                 _stream()), "apad_rmsnorm")
    return out.view(x.shape)


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, dim=-1)"""
    _req(x, "l2_normalize.x")
    Cc = x.shape[-1]
    x2 = x.reshape(-1, Cc)
    out = torch.empty_like(x2)
    L.check(L.lib().apad_rmsnorm(x2.data_ptr(), None, out.data_ptr(), x2.shape[0], Cc, x2.stride(0), Cc, float(eps), 1, _DT[x.dtype], _stream()),
            "apad_rmsnorm")
    return out.view(x.shape)


def embedding(table, ids):
    """nn.Embedding lookup on the device: table [rows, C], ids int64 [...] -> [..., C]"""
    _req(table, "embedding.table")
    if ids.dtype != torch.int64 or not ids.is_cuda:
        raise RuntimeError("embedding.ids: expected an int64 GPU tensor")
    ids = ids.contiguous()
    out = torch.empty(*ids.shape, table.shape[1], dtype=table.dtype, device=table.device)
    L.check(L.lib().apad_gather_rows(table.data_ptr(), ids.data_ptr(), out.data_ptr(), ids.numel(), table.shape[0], table.shape[1],
                                     _DT[table.dtype], _stream()), "apad_gather_rows")
    return out


def gaussian_sample(moments, noise, scale=1.0):
    """moments [rows, 2L] = (mean | logvar), noise [rows, L] -> (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale"""
    _req(moments, "gaussian_sample.moments")
    _req(noise, "gaussian_sample.noise", moments.dtype)
    rows, L2 = moments.shape
    if not (moments.is_contiguous() and noise.is_contiguous() and tuple(noise.shape) == (rows, L2 // 2)):
        raise ValueError(f"gaussian_sample: moments {tuple(moments.shape)} / noise {tuple(noise.shape)} must be contiguous [rows, 2L] / [rows, L]")
    out = torch.empty_like(noise)
    L.check(L.lib().apad_gaussian_sample(moments.data_ptr(), noise.data_ptr(), out.data_ptr(), rows, L2 // 2, float(scale), _DT[moments.dtype],
                                         _stream()), "apad_gaussian_sample")
    return out


def patch_embed(mel, w, bias, dtype):
    """mel fp32 [B, H, W] -> tokens [B, (H/16)*(W/16), 768]; w [768, 256] (= Conv2d weight [768,1,16,16] flattened)."""
    _req(mel, "patch_embed.mel", torch.float32)
    B, H, W = mel.shape
    n_tok = (H // 16) * (W // 16)
    out = torch.empty(B, n_tok, w.shape[0], dtype=dtype, device=mel.device)
    gemm(mel, w, M=B * n_tok, N=w.shape[0], K=256, lda=0, out=out, ldo=w.shape[0], bias=bias, a_mode=L.A_PATCH16,
         conv=(H, W, 1, H // 16, W // 16, 1, 0, 0, 0))
    return out


LOG2E = 1.4426950408889634


def attention(q, k, vt, Lk, heads, key_bias=None, k2=None, vt2=None, L2=0, scale2=0.0, kv_batch_div=1,
              kv2_batch_div=1, out=None, q_prescaled=False):
    """q [B,N,C]; k [Bk,Lk,C]; vt [Bk,heads,d,Lpad]; optional second (audio) segment k2/vt2 with its own softmax,
    blended as seg1 + scale2*seg2.  key_bias: fp32 [B,Lk] additive.  q_prescaled: q was projected with to_q rows already
    multiplied by log2(e) / sqrt(d) (processors._qkv_weight), i.e. it is the base-2 exponent operand."""
    _req(q, "attention.q")
    _req(k, "attention.k", q.dtype)
    _req(vt, "attention.vt", q.dtype)
    B, N, Cc = q.shape
    d_head = Cc // heads
    if out is None:
        out = torch.empty(B, N, Cc, dtype=q.dtype, device=q.device)
    d = L.AttnDesc()
    d.q, d.k, d.vt, d.out = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    d.key_bias = _ptr(key_bias)
    d.q_stride_b, d.q_stride_n = q.stride(0), q.stride(1)
    d.k_stride_b, d.k_stride_l, d.vt_stride_b = k.stride(0), k.stride(1), vt.stride(0)
    d.o_stride_b, d.o_stride_n = out.stride(0), out.stride(1)
    d.B, d.N, d.H, d.D, d.L, d.Lpad = B, N, heads, d_head, Lk, vt.shape[-1]
    d.kv_batch_div, d.kv2_batch_div = kv_batch_div, kv2_batch_div
    d.dtype = _DT[q.dtype]
    d.softmax_scale = 1.0 / math.sqrt(d_head)
    d.scale2 = float(scale2)
    d.q_prescaled = 1 if q_prescaled else 0
    if L2 > 0:
        d.k2, d.vt2 = k2.data_ptr(), vt2.data_ptr()
        d.k2_stride_b, d.k2_stride_l, d.vt2_stride_b = k2.stride(0), k2.stride(1), vt2.stride(0)
        d.L2, d.Lpad2 = L2, vt2.shape[-1]
    L.check(L.lib().apad_attention(C.byref(d), _stream()), "apad_attention")
    return out


def layer_norm(x, gamma, beta, eps, out=None):
    _req(x, "layer_norm.x", gamma.dtype)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    if out is None:
        out = torch.empty_like(x)
    L.check(L.lib().apad_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), M, Cc, Cc, Cc,
                                   eps, _DT[x.dtype], _stream()), "apad_layernorm")
    return out


_gn_ws = {}


def _gn_workspace(dev, B, HW, groups):
    key = (dev, B, HW, groups, torch.cuda.current_stream().cuda_stream)
    ws = _gn_ws.get(key)
    if ws is None:
        ws = torch.empty(L.lib().apad_groupnorm_workspace_bytes(B, HW, groups) // 4, dtype=torch.float32, device=dev)
        _gn_ws[key] = ws
    return ws


def group_norm(x, gamma, beta, groups, eps, silu=False, out=None):
    """x [B, HW, C] (NHWC)."""
    _req(x, "group_norm.x", gamma.dtype)
    B, HW, Cc = x.shape
    ws = _gn_workspace(x.device, B, HW, groups)
    if out is None:
        out = torch.empty_like(x)
    L.check(L.lib().apad_groupnorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ws.data_ptr(), B, HW,
                                   Cc, groups, eps, 1 if silu else 0, _DT[x.dtype], _stream()), "apad_groupnorm")
    return out


def group_norm2(xa, xb, gamma, beta, groups, eps, silu=False):
    """GroupNorm (+ SiLU) of torch.cat([xa, xb], -1) without the concatenation: xa [Ba, HW, Ca], xb [Bb, HW, Cb] -> [B, HW, Ca + Cb]
    with B = max(Ba, Bb), the smaller batch read modulo (see linear2).  16-bit storage types."""
    _req(xa, "group_norm2.xa", gamma.dtype)
    _req(xb, "group_norm2.xb", gamma.dtype)
    Ba, HW, Ca = xa.shape
    Bb, HWb, Cb = xb.shape
    B = max(Ba, Bb)
    if HWb != HW or B % Ba or B % Bb or not (xa.is_contiguous() and xb.is_contiguous()) or xa.dtype not in FUSED_DTYPES:
        raise ValueError(f"group_norm2: xa {tuple(xa.shape)}, xb {tuple(xb.shape)} {xa.dtype}")
    ws = _gn_workspace(xa.device, B, HW, groups)
    out = torch.empty(B, HW, Ca + Cb, dtype=xa.dtype, device=xa.device)
    L.check(L.lib().apad_groupnorm2(xa.data_ptr(), xb.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ws.data_ptr(), B, Ba, Bb,
                                    HW, Ca, Cb, groups, eps, 1 if silu else 0, _DT[xa.dtype], _stream()), "apad_groupnorm2")
    return out


def audiomae_pool(rep, tp, fp, out_dtype=None):
    _req(rep, "audiomae_pool.rep")
    B = rep.shape[0]
    out_dtype = out_dtype or rep.dtype
    out = torch.empty(B, (64 // tp) * (8 // fp), 768, dtype=out_dtype, device=rep.device)
    L.check(L.lib().apad_audiomae_pool(rep.data_ptr(), out.data_ptr(), B, tp, fp, _DT[rep.dtype], _DT[out_dtype],
                                       _stream()), "apad_audiomae_pool")
    return out


def timestep_embedding(t, dim, flip_sin_to_cos, freq_shift, dtype):
    _req(t, "timestep_embedding.t", torch.float32)
    out = torch.empty(t.numel(), dim, dtype=dtype, device=t.device)
    L.check(L.lib().apad_timestep_embedding(t.data_ptr(), out.data_ptr(), t.numel(), dim, 1 if flip_sin_to_cos else 0,
                                            float(freq_shift), _DT[dtype], _stream()), "apad_timestep_embedding")
    return out


def cfg_ddim_step(eps2, latents, unet_in, coef, step_ptr, guidance_scale, eps_out=None):
    """eps2 [2B, n...]; latents fp32 [B, n...] (in place); unet_in [B, n...] model dtype."""
    _req(latents, "cfg_ddim_step.latents", torch.float32)
    B = latents.shape[0]
    n = latents.numel() // B
    L.check(L.lib().apad_cfg_ddim_step(eps2.data_ptr(), latents.data_ptr(), unet_in.data_ptr(), _ptr(eps_out),
                                       coef.data_ptr(), _ptr(step_ptr), float(guidance_scale), B, n, _DT[eps2.dtype],
                                       _stream()), "apad_cfg_ddim_step")


def step_advance(step_ptr):
    L.check(L.lib().apad_step_advance(step_ptr.data_ptr(), _stream()), "apad_step_advance")


# ---------------------------------------------------------------------------------------------------------------------
# training step (SURVEY a-11): raw wrappers of the backward / optimizer entry points; autograd.py composes them
# ---------------------------------------------------------------------------------------------------------------------
def attention_lse(q, k, vt, Lk, heads, key_bias=None):
    """single-segment apad_attention that also returns the base-2 log-sum-exp [B, heads, round_up(N, 32)] fp32"""
    _req(q, "attention_lse.q")
    B, N, Cc = q.shape
    out = torch.empty_like(q)
    lse = torch.empty(B, heads, round_up(N, 32), dtype=torch.float32, device=q.device)  # (the kernels write the pad entries: 0)
    d = L.AttnDesc()
    d.q, d.k, d.vt, d.out, d.key_bias, d.lse = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), _ptr(key_bias), lse.data_ptr()
    d.q_stride_b, d.q_stride_n = q.stride(0), q.stride(1)
    d.k_stride_b, d.k_stride_l, d.vt_stride_b = k.stride(0), k.stride(1), vt.stride(0)
    d.o_stride_b, d.o_stride_n = out.stride(0), out.stride(1)
    d.B, d.N, d.H, d.D = B, N, heads, Cc // heads
    d.L, d.Lpad, d.L2, d.Lpad2 = Lk, vt.shape[-1], 0, 0
    d.kv_batch_div, d.kv2_batch_div, d.dtype = 1, 1, _DT[q.dtype]
    d.softmax_scale, d.scale2 = 1.0 / math.sqrt(Cc // heads), 0.0
    L.check(L.lib().apad_attention(C.byref(d), _stream()), "apad_attention")
    return out, lse


def head_transpose(x, heads, pad=None):
    """x [B, N, C] -> [B, heads, C/heads, pad] (zero padded, pad % 32 == 0)"""
    _req(x, "head_transpose.x")
    B, N, Cc = x.shape
    pad = pad or round_up(N, 32)
    xt = torch.empty(B, heads, Cc // heads, pad, dtype=x.dtype, device=x.device)
    L.check(L.lib().apad_head_transpose(x.data_ptr(), xt.data_ptr(), B, N, heads, Cc // heads, pad, _DT[x.dtype], _stream()),
            "apad_head_transpose")
    return xt


def head_transpose_many(xs, heads, pad):
    """several [B, N, C] tensors of ONE shape -> their [B, heads, C/heads, pad] transposes in ONE launch (up to three per launch)"""
    B, N, Cc = xs[0].shape
    outs = [torch.empty(B, heads, Cc // heads, pad, dtype=x.dtype, device=x.device) for x in xs]
    for i in range(0, len(xs), 3):
        a = [(_req(x, "head_transpose.x").data_ptr(), y.data_ptr()) for x, y in zip(xs[i:i + 3], outs[i:i + 3])]
        a += [(None, None)] * (3 - len(a))
        L.check(L.lib().apad_head_transpose3(a[0][0], a[0][1], a[1][0], a[1][1], a[2][0], a[2][1], B, N, heads, Cc // heads, pad,
                                             _DT[xs[0].dtype], _stream()), "apad_head_transpose3")
    return outs


def attention_bwd(q, k, v, out, dout, lse, heads, key_bias=None, dout_scale=1.0, need_dkv=True, dq=None, packed=False):
    """gradients of one softmax segment; dq given -> accumulated into.  Returns (dq, dk, dv).
    packed (self-attention, 16-bit): the three gradients are the column blocks of ONE [B, N, 3C] buffer (views of it are returned), so
    the input-gradient GEMM of the q|k|v projection reads them without a torch.cat."""
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout")):
        _req(t, "attention_bwd." + n, q.dtype)
        if not t.is_contiguous():
            raise RuntimeError(f"attention_bwd.{n}: must be contiguous")
    B, N, Cc = q.shape
    Lk = k.shape[1]
    Npad, Lpad = round_up(N, 32), round_up(Lk, 32)
    d = L.AttnBwdDesc()
    f32 = q.dtype == torch.float32  # fp32 training mode: plain-FMA kernels on the row-major operands, no transposed copies
    same = need_dkv and not f32 and N == Lk  # self-attention: q, k and dO transposed by ONE launch
    if same:
        kt, qt, dot = head_transpose_many([k, q, dout], heads, Lpad)
    else:
        kt = None if f32 else head_transpose(k, heads, Lpad)
    keep = [kt]
    d.q, d.k, d.v, d.kt, d.out, d.dout, d.lse = (q.data_ptr(), k.data_ptr(), v.data_ptr(), _ptr(kt), out.data_ptr(),
                                                 dout.data_ptr(), lse.data_ptr())
    d.key_bias = _ptr(key_bias)
    # (16-bit kernels write the pad entries of delta themselves; the fp32 kernels never read them)
    delta = torch.empty(B, heads, Npad, dtype=torch.float32, device=q.device)
    d.delta = delta.data_ptr()
    acc = dq is not None
    packed = packed and need_dkv and not f32 and dq is None and N == Lk
    dk = dv = None
    if packed:
        buf = torch.empty(B, N, 3 * Cc, dtype=q.dtype, device=q.device)
        dq, dk, dv = buf[..., :Cc], buf[..., Cc:2 * Cc], buf[..., 2 * Cc:]
        d.ld_grad = 3 * Cc
    elif dq is None:
        dq = torch.empty_like(q)
    d.dq = dq.data_ptr()
    if need_dkv:
        if not packed:
            dk, dv = torch.empty_like(k), torch.empty_like(v)
        d.dk, d.dv = dk.data_ptr(), dv.data_ptr()
        if not f32:
            if not same:
                qt, dot = head_transpose_many([q, dout], heads, Npad)
            keep += [qt, dot]
            d.qt, d.doutt = qt.data_ptr(), dot.data_ptr()
    d.B, d.N, d.H, d.D, d.L, d.Npad, d.Lpad, d.dtype = B, N, heads, Cc // heads, Lk, Npad, Lpad, _DT[q.dtype]
    d.softmax_scale, d.dout_scale, d.accumulate_dq = 1.0 / math.sqrt(Cc // heads), float(dout_scale), 1 if acc else 0
    L.check(L.lib().apad_attention_bwd(C.byref(d), _stream()), "apad_attention_bwd")
    return dq, dk, dv


def layer_norm_bwd(x, gamma, dy, eps, dres=None):
    """dres: the gradient that reaches x past the LayerNorm (a pre-norm sub-layer's residual connection), added in the same launch"""
    _req(x, "layer_norm_bwd.x", gamma.dtype)
    Cc = x.shape[-1]
    dx = torch.empty_like(x)
    if dres is not None:
        _req(dres, "layer_norm_bwd.dres", x.dtype)
        if not dres.is_contiguous() or dres.shape != x.shape:
            raise ValueError("layer_norm_bwd.dres: must be contiguous and shaped like x")
    L.check(L.lib().apad_layernorm_bwd_add(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), _ptr(dres), dx.data_ptr(), x.numel() // Cc, Cc,
                                           eps, _DT[x.dtype], _stream()), "apad_layernorm_bwd")
    return dx


def group_norm_bwd(x, gamma, beta, dy, groups, eps, silu):
    _req(x, "group_norm_bwd.x", gamma.dtype)
    B, HW, Cc = x.shape
    dx = torch.empty_like(x)
    L.check(L.lib().apad_groupnorm_bwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dy.data_ptr(), dx.data_ptr(), B, HW,
                                       Cc, groups, eps, 1 if silu else 0, _DT[x.dtype], _stream()), "apad_groupnorm_bwd")
    return dx


def geglu(proj):
    _req(proj, "geglu.proj")
    N = proj.shape[-1] // 2
    h = torch.empty(*proj.shape[:-1], N, dtype=proj.dtype, device=proj.device)
    L.check(L.lib().apad_geglu(proj.data_ptr(), h.data_ptr(), proj.numel() // (2 * N), N, _DT[proj.dtype], _stream()), "apad_geglu")
    return h


def geglu_bwd(proj, dh):
    N = proj.shape[-1] // 2
    dproj = torch.empty_like(proj)
    L.check(L.lib().apad_geglu_bwd(proj.data_ptr(), dh.data_ptr(), dproj.data_ptr(), proj.numel() // (2 * N), N,
                                   _DT[proj.dtype], _stream()), "apad_geglu_bwd")
    return dproj


def upsample_nearest_bwd(dup, B, H, W, Hup, Wup):
    Cc = dup.shape[-1]
    dx = torch.empty(B, H * W, Cc, dtype=dup.dtype, device=dup.device)
    L.check(L.lib().apad_upsample_nearest_bwd(dup.data_ptr(), dx.data_ptr(), B, H, W, Hup, Wup, Cc, _DT[dup.dtype], _stream()),
            "apad_upsample_nearest_bwd")
    return dx


def zero_stuff2(dy, B, H, W, Ho, Wo):
    Cc = dy.shape[-1]
    z = torch.empty(B, H * W, Cc, dtype=dy.dtype, device=dy.device)
    L.check(L.lib().apad_zero_stuff2(dy.data_ptr(), z.data_ptr(), B, H, W, Ho, Wo, Cc, _DT[dy.dtype], _stream()), "apad_zero_stuff2")
    return z


def transpose_pad(x, Mpad):
    """x [M, C] -> [C, Mpad] zero padded"""
    M, Cc = x.shape
    xt = torch.empty(Cc, Mpad, dtype=x.dtype, device=x.device)
    L.check(L.lib().apad_transpose_pad(x.data_ptr(), xt.data_ptr(), M, Cc, Mpad, _DT[x.dtype], _stream()), "apad_transpose_pad")
    return xt


def weight_grad(dy, x, fp32=False, acc=None):
    """dW [N, K] = dy[M, N]^T . x[M, K] (the adapter's to_k_ip / to_v_ip gradient), reduction over the M token rows.
    fp32: the product of the (storage-type) operands is formed on the exact-f32 MFMA path and returned in fp32 -- the reference
    computes the adapter gradients in fp32, and a per-micro-batch rounding of dW to bf16 would sit in front of the fp32
    accumulation otherwise.  acc (fp32 [N, K], with fp32=True): dW is ADDED to it in the GEMM's epilogue (residual = out = acc) and
    None is returned -- two launches per adapter tensor (transposes + widening, GEMM + accumulation) instead of six."""
    M, N = dy.shape
    K = x.shape[-1]
    Mpad = round_up(M, 64)
    dy, x = dy.contiguous(), x.contiguous()
    if fp32 and dy.dtype in FUSED_DTYPES:
        dyt = torch.empty(N, Mpad, dtype=torch.float32, device=dy.device)
        xt = torch.empty(K, Mpad, dtype=torch.float32, device=dy.device)
        L.check(L.lib().apad_transpose_pad2(dy.data_ptr(), dyt.data_ptr(), N, x.data_ptr(), xt.data_ptr(), K, M, Mpad, _DT[dy.dtype], 1, _stream()),
                "apad_transpose_pad2")
    else:
        dyt, xt = transpose_pad(dy, Mpad), transpose_pad(x, Mpad)
        if fp32:
            dyt, xt = dyt.float(), xt.float()  # widening copies (exact)
    if acc is not None:
        if not (fp32 and acc.dtype == torch.float32 and acc.is_contiguous() and tuple(acc.shape) == (N, K)):
            raise ValueError("weight_grad(acc=...): needs fp32=True and a contiguous fp32 [N, K] accumulator")
        gemm(dyt, xt, M=N, N=K, K=Mpad, lda=Mpad, out=acc, ldo=K, ldw=Mpad, residual=acc, ldr=K)
        return None
    out = torch.empty(N, K, dtype=dyt.dtype, device=dy.device)
    gemm(dyt, xt, M=N, N=K, K=Mpad, lda=Mpad, out=out, ldo=K, ldw=Mpad)
    return out


def _reduce_ws(dev):
    return torch.empty(L.lib().apad_reduce_workspace_bytes() // 4, dtype=torch.float32, device=dev)


def mse_loss_grad(pred, target, grad_scale=1.0):
    """(loss fp32 scalar tensor, grad_scale * dpred in pred.dtype) of F.mse_loss(pred.float(), target.float())"""
    _req(pred, "mse_loss_grad.pred")
    _req(target, "mse_loss_grad.target", torch.float32)
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred)
    ws = _reduce_ws(pred.device)
    L.check(L.lib().apad_mse_loss_grad(pred.data_ptr(), target.data_ptr(), dpred.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                                       pred.numel(), float(grad_scale), _DT[pred.dtype], _stream()), "apad_mse_loss_grad")
    return loss, dpred


def step_advance_if_finite(step_ptr, grad_norm_t):
    L.check(L.lib().apad_step_advance_if_finite(step_ptr.data_ptr(), _ptr(grad_norm_t), _stream()), "apad_step_advance_if_finite")


def grad_norm(grad, out=None, ws=None):
    _req(grad, "grad_norm.grad", torch.float32)
    out = out if out is not None else torch.empty(1, dtype=torch.float32, device=grad.device)
    ws = ws if ws is not None else _reduce_ws(grad.device)
    L.check(L.lib().apad_grad_norm(grad.data_ptr(), out.data_ptr(), ws.data_ptr(), grad.numel(), _stream()), "apad_grad_norm")
    return out


def adamw_step(param, work, grad, exp_avg, exp_avg_sq, grad_norm_t, step_t, lr, betas, eps, weight_decay, max_grad_norm):
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, "adamw_step." + n, torch.float32)
    L.check(L.lib().apad_adamw_step(param.data_ptr(), _ptr(work), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                    _ptr(grad_norm_t), step_t.data_ptr(), param.numel(), lr, betas[0], betas[1], eps,
                                    weight_decay, max_grad_norm if max_grad_norm else 0.0,
                                    _DT[work.dtype] if work is not None else L.BF16, _stream()), "apad_adamw_step")

"""-m gpu: every C-ABI kernel against the oracle / plain fp32 torch on identical (storage-rounded) operands.
Tolerance: max-abs error <= TOL[dtype] * max|ref| (bf16 2e-2, f16 3e-3): outputs are rounded once to the storage
type (2^-8 / 2^-11 relative) after fp32 accumulation."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import TOL, q, rel_err

pytestmark = pytest.mark.gpu
DTYPES16 = [torch.bfloat16, torch.float16]     # the fused kernels (row-panel, feed-forward, cross-attention)
DTYPES = DTYPES16 + [torch.float32]            # float32: the fp32 precision mode (csrc/f32_ops.hip)


def R(*shape, seed=0, std=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * std


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 256), (1000, 384, 768), (77, 8, 72), (513, 640, 2560)])
def test_gemm_plain_bias_residual(dev, dtype, M, N, K):
    from ap_adapter_amd import ops
    x, w, b, r = q(R(M, K, seed=1), dtype), q(R(N, K, seed=2, std=0.05), dtype), q(R(N, seed=3), dtype), q(R(M, N, seed=4), dtype)
    ref = F.linear(x, w, b)
    out = ops.linear(x.to(dev, dtype), w.to(dev, dtype), b.to(dev, dtype))
    assert rel_err(out, ref) < TOL[dtype]
    out = ops.linear(x.to(dev, dtype), w.to(dev, dtype), None, residual=r.to(dev, dtype))
    assert rel_err(out, F.linear(x, w) + r) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", ["silu", "gelu", "geglu"])
def test_gemm_activations(dev, dtype, act):
    from ap_adapter_amd import ops
    M, K, N = 260, 256, 192
    rows = 2 * N if act == "geglu" else N
    x, w, b = q(R(M, K, seed=5), dtype), q(R(rows, K, seed=6, std=0.08), dtype), q(R(rows, seed=7, std=0.5), dtype)
    y = F.linear(x, w, b)
    if act == "silu":
        ref = F.silu(y)
    elif act == "gelu":
        ref = F.gelu(y)
    else:
        a, g = y.chunk(2, dim=-1)
        ref = a * F.gelu(g)
    out = ops.linear(x.to(dev, dtype), w.to(dev, dtype), b.to(dev, dtype), act=act)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_rowgroup_bias_and_step(dev, dtype):
    from ap_adapter_amd import ops
    B, HW, K, N = 3, 50, 64, 128
    x, w = q(R(B * HW, K, seed=8), dtype), q(R(N, K, seed=9, std=0.1), dtype)
    rg = q(R(B, N, seed=10), dtype)
    ref = (F.linear(x, w).view(B, HW, N) + rg[:, None, :]).view(B * HW, N)
    out = ops.linear(x.to(dev, dtype), w.to(dev, dtype), rowgroup_bias=rg.to(dev, dtype), rows_per_group=HW)
    assert rel_err(out, ref) < TOL[dtype]
    # table mode: every row uses table[*step_ptr]
    step = torch.tensor([2], dtype=torch.int32, device=dev)
    out = ops.linear(x.to(dev, dtype), w.to(dev, dtype), rowgroup_bias=rg.to(dev, dtype), rows_per_group=1 << 40, step_ptr=step)
    assert rel_err(out, F.linear(x, w) + rg[2][None, :]) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,L,heads,d", [(2, 100, 8, 32), (1, 513, 12, 64), (2, 8, 8, 80), (3, 40, 8, 48)])
def test_gemm_vt_output(dev, dtype, B, L, heads, d):
    from ap_adapter_amd import ops
    C_, K = heads * d, 128
    x, w, b = q(R(B * L, K, seed=11), dtype), q(R(C_, K, seed=12, std=0.1), dtype), q(R(C_, seed=13), dtype)
    Lpad = ops.round_up(L, 32)
    vt = torch.zeros(B, heads, d, Lpad, dtype=dtype, device=dev)
    ops.linear_vt(x.to(dev, dtype), w.to(dev, dtype), B, L, heads, vt, bias=b.to(dev, dtype))
    ref = F.linear(x, w, b).view(B, L, heads, d).permute(0, 2, 3, 1)
    assert rel_err(vt[..., :L], ref) < TOL[dtype]
    assert float(vt[..., L:].float().abs().max()) == 0.0 if Lpad > L else True


def _conv_ref(x_nchw, w, b, stride=1, up=None):
    if up is not None:
        x_nchw = F.interpolate(x_nchw, size=up, mode="nearest")
    return F.conv2d(x_nchw, w, b, stride=stride, padding=1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [
    (2, 10, 16, 8, 128, 1, None), (2, 25, 16, 128, 128, 2, None), (1, 63, 4, 384, 384, 2, None),
    (2, 32, 2, 640, 640, 1, (63, 4)), (2, 13, 8, 64, 96, 1, (26, 16)), (1, 250, 16, 128, 8, 1, None)])
def test_conv3x3_implicit_gemm(dev, dtype, B, H, W, Cin, Cout, stride, up):
    from ap_adapter_amd import ops
    x = q(R(B, Cin, H, W, seed=14), dtype)
    w = q(R(Cout, Cin, 3, 3, seed=15, std=0.05), dtype)
    b = q(R(Cout, seed=16), dtype)
    ref = _conv_ref(x, w, b, stride, up)
    xn = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous().to(dev, dtype)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev, dtype)
    out, Ho, Wo = ops.conv3x3(xn, wp, b.to(dev, dtype), B, H, W, stride=stride, up=up)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    out = out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(9, 125, 30, 64, 128), (3, 250, 45, 128, 256), (5, 111, 60, 192, 128)])
def test_conv3x3_big_tile_kernel(dev, dtype, B, H, W, Cin, Cout):
    """csrc/cgemm.hip (256x128 tile, LDS-DMA operands, zero padding through the buffer range check): >= 32768 output pixels, ragged
    last tile, every border, + bias + table-mode time embedding (row *step_ptr) + residual -- the resnet call"""
    from ap_adapter_amd import ops
    assert B * H * W >= 16000 and (B * H * W) % 256 != 0
    x = q(R(B, Cin, H, W, seed=14), dtype)
    w = q(R(Cout, Cin, 3, 3, seed=15, std=0.05), dtype)
    b = q(R(Cout, seed=16), dtype)
    t = q(R(4, Cout, seed=17), dtype)
    r = q(R(B, Cout, H, W, seed=18), dtype)
    ref = q(_conv_ref(x, w, b) + t[2][None, :, None, None], dtype) + r
    nhwc = lambda a: a.permute(0, 2, 3, 1).reshape(B, H * W, -1).contiguous().to(dev, dtype)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev, dtype)
    step = torch.tensor([2], dtype=torch.int32, device=dev)
    out, Ho, Wo = ops.conv3x3(nhwc(x), wp, b.to(dev, dtype), B, H, W, residual=nhwc(r), rowgroup_bias=t.to(dev, dtype),
                              rows_per_group=1 << 40, step_ptr=step)
    out = out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,H,W,Cin,Cout,up,temb", [
    (3, 250, 16, 128, 128, None, "table"),    # the 4000-pixel level: 256 x 128 tiles, a sample boundary inside a tile, ragged last tile
    (5, 125, 8, 256, 256, None, "table"),     # the 1000-pixel level: 256 x 256 tiles
    (7, 63, 4, 384, 384, None, "sample"),     # the 252-pixel level (N = 384: three 128-wide tiles), up to two sample boundaries per tile
    (2, 125, 8, 256, 256, (250, 16), None),   # Upsample2D: the nearest-x2 source folded into the halo fill
    (3, 63, 4, 384, 256, (125, 8), None),     # ... with an odd target size
    (9, 32, 2, 640, 384, (63, 4), None),
    (1, 5, 4, 64, 128, None, "sample"),       # a tile holding many tiny samples' rows ... here ONE sample smaller than a tile
    (40, 5, 4, 64, 128, None, "table"),       # 13 separator rows per tile
    (2, 40, 16, 640, 256, None, "table"),     # ten 64-channel chunks
    (5, 63, 4, 128, 128, None, "table"),      # 256 x 128 tiles on a 4-wide image
    (7, 32, 2, 640, 640, None, "table"),      # the 64-token level: three K slices of whole chunks, fp32 slabs summed in slice order
    (5, 32, 2, 384, 640, None, "sample"),     # ... two chunks per slice, per-sample time-embedding rows in the slice sum
    (3, 16, 2, 1280, 128, (32, 2), None),     # ... up-sampled source, 20 chunks in slices of 6 / 7 / 7
    (2, 32, 2, 128, 128, None, "table"),      # a 2-wide image with fewer than three chunks: no slices
])
def test_conv3x3_halo_kernel(dev, dtype, B, H, W, Cin, Cout, up, temb):
    """csrc/hconv.hip: the halo-resident 3x3 convolution (input rows of a tile resident in LDS, nine taps as nine shifts of the same tile,
    packed weight stream) against fp32 torch: borders, sample boundaries inside tiles, ragged last tile, bias + time-embedding row (table
    and per-sample form) + residual, nearest-up-sampled sources; the route is asserted"""
    from ap_adapter_amd import ops, _lib as L
    x = q(R(B, Cin, H, W, seed=14), dtype)
    w = q(R(Cout, Cin, 3, 3, seed=15, std=0.04), dtype)
    b = q(R(Cout, seed=16), dtype)
    Ho, Wo = up if up is not None else (H, W)
    t = q(R(B if temb == "sample" else 4, Cout, seed=17), dtype)
    r = q(R(B, Cout, Ho, Wo, seed=18), dtype)
    ref = _conv_ref(x, w, b, 1, up)
    if temb == "table":
        ref = ref + t[3][None, :, None, None]
    elif temb == "sample":
        ref = ref + t[:, :, None, None]
    ref = q(ref, dtype) + r
    nhwc = lambda a: a.permute(0, 2, 3, 1).reshape(a.shape[0], a.shape[2] * a.shape[3], -1).contiguous().to(dev, dtype)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev, dtype)
    step = torch.tensor([3 if temb == "table" else 0], dtype=torch.int32, device=dev)
    kw = {}
    if temb is not None:
        kw = dict(rowgroup_bias=t.to(dev, dtype), rows_per_group=(1 << 40) if temb == "table" else Ho * Wo, step_ptr=step)
    n0 = L.lib().apad_hconv_launch_count()
    out, Ho2, Wo2 = ops.conv3x3(nhwc(x), wp, b.to(dev, dtype), B, H, W, up=up, residual=nhwc(r), **kw)
    assert L.lib().apad_hconv_launch_count() == n0 + 1, "not on the halo kernel"
    assert (Ho2, Wo2) == (Ho, Wo)
    assert rel_err(out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref) < TOL[dtype]
    # the same pixels inside a larger batch: bit-equal (one kernel, one summation order, whatever the row count)
    if up is None and B >= 3:
        out1, _, _ = ops.conv3x3(nhwc(x)[1:2].contiguous(), wp, b.to(dev, dtype), 1, H, W, residual=nhwc(r)[1:2].contiguous(),
                                 **(dict(kw, rowgroup_bias=t[1:2].to(dev, dtype)) if temb == "sample" else kw))
        assert torch.equal(out1[0], out[1])


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,H,W,Cin,Cout,up", [
    (3, 250, 16, 128, 8, None),     # conv_out: two 64-channel chunks, a sample boundary inside a tile, ragged last tile, persistent workgroups (47 tiles)
    (70, 250, 16, 128, 8, None),    # ... more tiles than CUs: the chunk stream crosses tiles (1094 tiles)
    (5, 63, 4, 64, 16, None),       # one chunk per tile (the request two TILES ahead), 16 output channels
    (2, 125, 8, 128, 16, None),
    (2, 63, 4, 128, 8, (125, 8)),   # up-sampled source
])
def test_conv3x3_halo_kernel_narrow_form(dev, dtype, B, H, W, Cin, Cout, up):
    """csrc/hconv.hip, narrow form (N <= 16: weights stationary in registers as 16-row MFMA fragments, the pixels stream through three halo
    buffers): against fp32 torch, route asserted, a sample bit-equal inside and outside a batch"""
    from ap_adapter_amd import ops, _lib as L
    x = q(R(B, Cin, H, W, seed=24), dtype)
    w = q(R(Cout, Cin, 3, 3, seed=25, std=0.04), dtype)
    b = q(R(Cout, seed=26), dtype)
    ref = _conv_ref(x, w, b, 1, up)
    Ho, Wo = up if up is not None else (H, W)
    nhwc = lambda a: a.permute(0, 2, 3, 1).reshape(a.shape[0], a.shape[2] * a.shape[3], -1).contiguous().to(dev, dtype)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev, dtype)
    n0 = L.lib().apad_hconv_launch_count()
    out, Ho2, Wo2 = ops.conv3x3(nhwc(x), wp, b.to(dev, dtype), B, H, W, up=up)
    assert L.lib().apad_hconv_launch_count() == n0 + 1, "not on the halo kernel"
    assert (Ho2, Wo2) == (Ho, Wo)
    assert rel_err(out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref) < TOL[dtype]
    if up is None and B >= 3:
        out1, _, _ = ops.conv3x3(nhwc(x)[1:2].contiguous(), wp, b.to(dev, dtype), 1, H, W)
        assert torch.equal(out1[0], out[1])


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 32, 2, 640, 640), (5, 32, 2, 1280, 640), (3, 63, 4, 384, 384), (1, 9, 7, 256, 64), (7, 13, 5, 320, 192)])
def test_conv3x3_small_tile_ring_kernel(dev, dtype, B, H, W, Cin, Cout):
    """csrc/cgemm.hip, small-tile form (64 x 64 tile, four-stage LDS-DMA ring): the long-reduction 3x3 convolutions below 16000 output
    pixels (the 64-token level; K = 9 Cin >= 2304), ragged last tile, 1 .. 180 k-tiles, + bias + time-embedding table row + residual;
    same k order as the tiled kernel"""
    from ap_adapter_amd import ops
    x = q(R(B, Cin, H, W, seed=14), dtype)
    w = q(R(Cout, Cin, 3, 3, seed=15, std=0.03), dtype)
    b = q(R(Cout, seed=16), dtype)
    t = q(R(4, Cout, seed=17), dtype)
    r = q(R(B, Cout, H, W, seed=18), dtype)
    ref = q(_conv_ref(x, w, b) + t[1][None, :, None, None], dtype) + r
    nhwc = lambda a: a.permute(0, 2, 3, 1).reshape(B, H * W, -1).contiguous().to(dev, dtype)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev, dtype)
    step = torch.tensor([1], dtype=torch.int32, device=dev)
    out, Ho, Wo = ops.conv3x3(nhwc(x), wp, b.to(dev, dtype), B, H, W, residual=nhwc(r), rowgroup_bias=t.to(dev, dtype),
                              rows_per_group=1 << 40, step_ptr=step)
    assert rel_err(out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,up", [
    (9, 250, 30, 64, 128, 2, None), (5, 125, 8, 128, 256, 1, (250, 16)), (64, 32, 2, 384, 256, 1, (63, 4)), (3, 63, 4, 384, 384, 2, None),
    (4, 32, 2, 640, 640, 1, (63, 4)), (2, 125, 9, 256, 128, 2, None)])
def test_conv3x3_general_forms_on_the_dma_kernels(dev, dtype, B, H, W, Cin, Cout, stride, up):
    """stride-2 (Downsample2D) and nearest-upsampled-source (Upsample2D, odd target sizes) 3x3 convolutions on the LDS-DMA kernels
    (big-tile from 16000 output pixels, small-tile ring below for K >= 2304): per-row source offsets per tap, borders, ragged tiles"""
    from ap_adapter_amd import ops
    x = q(R(B, Cin, H, W, seed=14), dtype)
    w = q(R(Cout, Cin, 3, 3, seed=15, std=0.04), dtype)
    b = q(R(Cout, seed=16), dtype)
    ref = _conv_ref(x, w, b, stride, up)
    xn = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous().to(dev, dtype)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev, dtype)
    out, Ho, Wo = ops.conv3x3(xn, wp, b.to(dev, dtype), B, H, W, stride=stride, up=up)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    assert rel_err(out.reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("M,N,K", [(40001, 128, 192), (33000, 384, 1024), (32768, 256, 64)])
def test_gemm_big_tile_kernel(dev, dtype, M, N, K):
    """the same kernel on a plain A operand (1, 3, 16 k-tiles: prologue-only, one round of the three LDS stages, many)"""
    from ap_adapter_amd import ops
    x, w, b, r = q(R(M, K, seed=1), dtype), q(R(N, K, seed=2, std=0.05), dtype), q(R(N, seed=3), dtype), q(R(M, N, seed=4), dtype)
    out = ops.linear(x.to(dev, dtype), w.to(dev, dtype), b.to(dev, dtype), residual=r.to(dev, dtype))
    assert rel_err(out, q(F.linear(x, w, b), dtype) + r) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
def test_gemm_ring_form_equals_the_tiled_kernel_bit_for_bit(dev, dtype):
    """apad_set_gemm_ring(2): the LDS-DMA ring form of the 64 x 64 tile (the training step's launches) against the tiled kernel on the
    same descriptors -- plain / bias / residual / row-modulo residual, GEGLU, the K-group shapes (N >= 640, K >= 384), row statistics
    out and a folded LayerNorm in, the q | k | v^T output mode, two A sources; ragged M, 2 .. 40 k-tiles."""
    from ap_adapter_amd import ops

    def both(fn):
        outs = []
        for mode in (0, 2):
            old = ops.set_gemm_ring(mode)
            try:
                outs.append(fn())
            finally:
                ops.set_gemm_ring(old)
        return outs

    def same(a, b):
        a, b = (a, b) if isinstance(a, (list, tuple)) else ([a], [b])
        return all(torch.equal(x, y) for x, y in zip(a, b))

    D = lambda *s, seed, std=1.0: q(R(*s, seed=seed, std=std), dtype).to(dev, dtype)
    for i, (M, N, K) in enumerate([(4000, 256, 256), (1008, 640, 640), (250, 1280, 1280), (2048, 640, 2560), (333, 64, 128), (4000, 384, 1152)]):
        x, w, b, r = D(M, K, seed=10 + i), D(N, K, seed=20 + i, std=0.05), D(N, seed=30 + i), D(M, N, seed=40 + i)
        a, c = both(lambda: ops.linear(x, w, b, residual=r))
        assert torch.equal(a, c), (M, N, K)
        assert rel_err(a.cpu(), q(F.linear(x.cpu().float(), w.cpu().float(), b.cpu().float()), dtype) + r.cpu().float()) < TOL[dtype]
        a, c = both(lambda: ops.linear(x, w))
        assert torch.equal(a, c), (M, N, K)
        if M % 4 == 0:
            a, c = both(lambda: ops.linear(x, w, b, residual=r[: M // 4], residual_row_mod=M // 4))
            assert torch.equal(a, c), (M, N, K, "row-modulo residual")
    # GEGLU (the un-fused feed-forward of the 640 / 1280 levels)
    x, w, b = D(1008, 640, seed=50), D(2 * 2560, 640, seed=51, std=0.05), D(2 * 2560, seed=52)
    a, c = both(lambda: ops.linear(x, w, b, act="geglu"))
    assert torch.equal(a, c)
    # row statistics out, then LayerNorm folded into the consumer
    x, w1, b1, r = D(1000, 640, seed=53), D(640, 640, seed=54, std=0.05), D(640, seed=55), D(1000, 640, seed=56)
    g, be, w2, b2 = D(640, seed=57), D(640, seed=58), D(1280, 640, seed=59, std=0.05), D(1280, seed=60)

    def chain():
        y = ops.linear(x, w1, b1, residual=r, rowstat=True)
        outs = [y, ops.rowstat_of(y)]
        if ops.ln_foldable(y, w2):
            outs.append(ops.linear(y, w2, b2, ln=(g, be, 1e-5)))
        return outs

    a, c = both(chain)
    assert len(a) == 3 and same(a, c)
    # q | k | v^T (self-attention projections of the 640 level)
    B, Lk, heads, C = 3, 250, 8, 640
    x, wq = D(B, Lk, C, seed=61), D(3 * C, C, seed=62, std=0.05)
    Lp = ops.round_up(Lk, 8)

    def qkv():
        qo, ko = torch.empty(B, Lk, C, dtype=dtype, device=dev), torch.empty(B, Lk, C, dtype=dtype, device=dev)
        vt = torch.zeros(B, heads, C // heads, Lp, dtype=dtype, device=dev)
        vo = torch.empty(B, Lk, C, dtype=dtype, device=dev)
        ops.linear_qkv(x, wq, B, Lk, heads, qo, ko, vt, v=vo)  # (+ the optional row-major v of the training step)
        return [qo, ko, vt, vo]

    a, c = both(qkv)
    assert same(a, c)
    for qo, ko, vt, vo in (a, c):  # row-major v = the same rounded numbers as v^T
        assert torch.equal(vo.view(B, Lk, heads, C // heads).permute(0, 2, 3, 1), vt[..., :Lk])
        assert rel_err(vo.cpu(), F.linear(x.cpu().float(), wq.cpu().float()[2 * C:])) < TOL[dtype]
    # two A sources (the up-block shortcut over [hidden | skip]), the skip read modulo
    xa, xb, w, b = D(4, 250, 640, seed=63), D(2, 250, 320, seed=64), D(640, 960, seed=65, std=0.05), D(640, seed=66)
    a, c = both(lambda: ops.linear2(xa, xb, w, b))
    assert torch.equal(a, c)


def test_conv3x3_cfg_duplication_and_temb(dev):
    from ap_adapter_amd import ops
    dtype = torch.bfloat16
    B, H, W, Cin, Cout = 2, 12, 16, 8, 128
    x, w = q(R(B, Cin, H, W, seed=17), dtype), q(R(Cout, Cin, 3, 3, seed=18, std=0.1), dtype)
    t = q(R(2 * B, Cout, seed=19), dtype)
    ref = F.conv2d(torch.cat([x, x]), w, None, padding=1) + t[:, :, None, None]
    xn = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous().to(dev, dtype)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(dev, dtype)
    out, _, _ = ops.conv3x3(xn, wp, None, 2 * B, H, W, src_batch_mod=B, rowgroup_bias=t.to(dev, dtype), rows_per_group=H * W)
    out = out.reshape(2 * B, H, W, Cout).permute(0, 3, 1, 2)
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_patch_embed(dev, dtype):
    from ap_adapter_amd import ops
    mel = R(2, 1024, 128, seed=20, std=0.5)
    w, b = q(R(768, 1, 16, 16, seed=21, std=0.05), dtype), q(R(768, seed=22, std=0.1), dtype)
    ref = F.conv2d(q(mel, dtype).unsqueeze(1), w, b, stride=16).flatten(2).transpose(1, 2)
    out = ops.patch_embed(mel.to(dev), w.reshape(768, 256).to(dev, dtype), b.to(dev, dtype), dtype)
    assert rel_err(out, ref) < TOL[dtype]


def _attn_ref(qh, kh, vh, bias=None):
    s = qh @ kh.transpose(-1, -2) / math.sqrt(qh.shape[-1])
    if bias is not None:
        s = s + bias[:, None, None, :]
    return torch.softmax(s, -1) @ vh


def _heads(x, h):
    b, n, c = x.shape
    return x.view(b, n, h, c // h).transpose(1, 2)


def _vt(v, heads, dev, dtype):
    from ap_adapter_amd import ops
    B, L, C_ = v.shape
    buf = torch.zeros(B, heads, C_ // heads, ops.round_up(L, 32), dtype=dtype, device=dev)
    buf[..., :L] = _heads(v, heads).transpose(-1, -2).to(dev, dtype)
    return buf


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,L,heads,d,masked", [
    (2, 100, 100, 8, 32, False), (1, 1000, 1000, 8, 32, False), (2, 252, 252, 8, 48, False), (2, 64, 64, 8, 80, False),
    (2, 64, 16, 8, 80, True), (1, 513, 513, 12, 64, False), (2, 130, 33, 4, 64, True), (1, 40, 7, 2, 16, False),
    (1, 33, 70, 2, 128, False), (1, 64, 45, 2, 96, False),
    # the two-query-tile kernel of long self-attention (N >= 512, L >= 256, d = 32): ragged query and key tails, key sharing across a CFG pair
    (3, 1000, 1000, 8, 32, False), (2, 600, 300, 4, 32, False), (1, 777, 1029, 2, 32, False)])
def test_attention_single_segment(dev, dtype, B, N, L, heads, d, masked):
    from ap_adapter_amd import ops
    C_ = heads * d
    qq, kk, vv = q(R(B, N, C_, seed=23), dtype), q(R(B, L, C_, seed=24), dtype), q(R(B, L, C_, seed=25), dtype)
    bias = None
    if masked:
        bias = torch.zeros(B, L)
        bias[1::2, -4:] = -10000.0
        bias[0, 1] = -3.0
    ref = _attn_ref(_heads(qq, heads), _heads(kk, heads), _heads(vv, heads), bias).transpose(1, 2).reshape(B, N, C_)
    out = ops.attention(qq.to(dev, dtype), kk.to(dev, dtype), _vt(vv, heads, dev, dtype), L, heads,
                        key_bias=None if bias is None else bias.to(dev))
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
def test_attention_two_query_tile_kernel_lse_and_equality(dev, dtype):
    """long self-attention runs two query tiles per wave: the output, and the log-sum-exp the backward re-uses"""
    from ap_adapter_amd import ops
    B, N, heads, d = 2, 1000, 8, 32
    C_ = heads * d
    qq, kk, vv = q(R(B, N, C_, seed=26), dtype), q(R(B, N, C_, seed=27), dtype), q(R(B, N, C_, seed=28), dtype)
    qd, kd, vt = qq.to(dev, dtype), kk.to(dev, dtype), _vt(vv, heads, dev, dtype)
    out = ops.attention(qd, kd, vt, N, heads)
    o2, lse = ops.attention_lse(qd, kd, vt, N, heads)  # the training forward: same kernel, plus log2-sum-exp2 per query
    qh, kh = _heads(qq, heads), _heads(kk, heads)
    sc = (qh @ kh.transpose(-1, -2)) / math.sqrt(d)
    ref = (torch.softmax(sc, -1) @ _heads(vv, heads)).transpose(1, 2).reshape(B, N, C_)
    assert rel_err(out, ref) < TOL[dtype]
    assert torch.equal(o2, out)
    ref_lse = torch.logsumexp(sc, -1) * math.log2(math.e)
    assert rel_err(lse.view(B, heads, -1)[..., :N], ref_lse) < 1e-3



def _prescaled(qq, d, dtype):
    """q as the processors hand it over with q_prescaled: multiplied by log2(e) / sqrt(d) BEFORE the one rounding to the storage
    type (there through the scaled to_q rows); the reference of such a call is softmax over base-2 exponents q' . k"""
    return q(qq * (math.log2(math.e) / math.sqrt(d)), dtype)


def _attn_ref_exp2(qh, kh, vh):
    s = (qh @ kh.transpose(-1, -2)) * math.log(2.0)  # 2^x = e^(x ln 2)
    return torch.softmax(s, -1) @ vh, s


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,N,L,heads,d", [(2, 1000, 1000, 8, 32), (1, 777, 1029, 2, 32), (2, 600, 300, 4, 32), (3, 512, 256, 8, 32),
                                           (2, 252, 252, 8, 48), (2, 100, 100, 8, 32)])
def test_attention_prescaled_q(dev, dtype, B, N, L, heads, d):
    """q_prescaled (the self-attention route behind the row-panel projection): scores are base-2 exponents as they come out of
    the MFMA.  The long d = 32 shapes run the two-query-tile kernel's direct form (running max in the score MFMA's C operand, the
    tile's probability sum as the range check); the others the generic kernels with scale 1.  Output and log-sum-exp."""
    from ap_adapter_amd import ops
    C_ = heads * d
    qp = _prescaled(R(B, N, C_, seed=43), d, dtype)
    kk, vv = q(R(B, L, C_, seed=44), dtype), q(R(B, L, C_, seed=45), dtype)
    ref, sc = _attn_ref_exp2(_heads(qp, heads), _heads(kk, heads), _heads(vv, heads))
    ref = ref.transpose(1, 2).reshape(B, N, C_)
    qd, kd, vt = qp.to(dev, dtype), kk.to(dev, dtype), _vt(vv, heads, dev, dtype)
    out = ops.attention(qd, kd, vt, L, heads, q_prescaled=True)
    assert rel_err(out, ref) < TOL[dtype]
    # the same scores through the classic form (scale applied in the kernel) differ by one rounding of q only
    q0 = q(R(B, N, C_, seed=43), dtype)
    base = ops.attention(q0.to(dev, dtype), kd, vt, L, heads)
    assert rel_err(out, base.float().cpu()) < 2.5 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("spike_tile,gain", [(0, 6.0), (7, 6.0), (7, 30.0), (15, 12.0), (3, -8.0)])
def test_attention_direct_form_range_check_branches(dev, dtype, spike_tile, gain):
    """the direct form skips the per-tile maximum; a key whose score jumps far above the running max must trip the range check
    (classic update for that tile), one that stays inside the window must not change the result either.  One key row is set to
    gain x a query, in a chosen 64-key tile: +6 stays inside the bf16 window (2^30) for most rows, +30 / +12 leave it (and leave
    f16's 2^12 window), a negative spike exercises the far-below side.  Full-tensor fp32 reference."""
    from ap_adapter_amd import ops
    B, N, heads, d = 1, 1000, 2, 32
    L, C_ = N, heads * d
    qp = _prescaled(R(B, N, C_, seed=46) * 2.0, d, dtype)
    kk, vv = R(B, L, C_, seed=47), q(R(B, L, C_, seed=48), dtype)
    kk[:, spike_tile * 64 + 17] = (qp[:, 333] * gain)
    kk[:, min(spike_tile * 64 + 70, L - 1)] = (qp[:, 900] * gain * 0.5)
    kk = q(kk, dtype)
    ref, sc = _attn_ref_exp2(_heads(qp, heads), _heads(kk, heads), _heads(vv, heads))
    ref = ref.transpose(1, 2).reshape(B, N, C_)
    if gain > 0:
        assert float(sc.max()) / math.log(2.0) > (35.0 if gain >= 12 else 12.0)  # the spike really is outside / inside the window
    out = ops.attention(qp.to(dev, dtype), kk.to(dev, dtype), _vt(vv, heads, dev, dtype), L, heads, q_prescaled=True)
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,heads,d,Lt,La,scale", [
    (2, 100, 8, 32, 8, 32, 0.55), (1, 1000, 8, 32, 8, 512, 0.5), (2, 252, 8, 48, 8, 128, 0.5), (2, 64, 8, 80, 8, 8, 1.0),
    (2, 64, 8, 80, 8, 512, 0.5), (2, 100, 8, 32, 8, 0, 0.5), (2, 100, 8, 32, 8, 32, 0.0), (2, 70, 4, 64, 4, 33, 0.7)])
def test_attention_decoupled(dev, dtype, B, N, heads, d, Lt, La, scale):
    """two independently normalised key segments blended text + scale * audio (attention_processor.py:429-454)"""
    from ap_adapter_amd import ops
    C_ = heads * d
    qq = q(R(B, N, C_, seed=26), dtype)
    kt, vt_ = q(R(B, Lt, C_, seed=27), dtype), q(R(B, Lt, C_, seed=28), dtype)
    ka, va = q(R(B, max(La, 1), C_, seed=29), dtype)[:, :La], q(R(B, max(La, 1), C_, seed=30), dtype)[:, :La]
    qh = _heads(qq, heads)
    ref = _attn_ref(qh, _heads(kt, heads), _heads(vt_, heads))
    if La > 0:
        ref = ref + scale * _attn_ref(qh, _heads(ka, heads), _heads(va, heads))
    ref = ref.transpose(1, 2).reshape(B, N, C_)
    kw = {}
    if La > 0:
        kw = dict(k2=ka.contiguous().to(dev, dtype), vt2=_vt(va, heads, dev, dtype), L2=La, scale2=scale)
    out = ops.attention(qq.to(dev, dtype), kt.to(dev, dtype), _vt(vt_, heads, dev, dtype), Lt, heads, **kw)
    assert rel_err(out, ref) < TOL[dtype]


def test_attention_shared_kv_batch_div(dev):
    """kv_batch_div: one K/V set shared by a group of consecutive samples (one audio prompt per CFG half)"""
    from ap_adapter_amd import ops
    dtype, B, N, heads, d, L = torch.bfloat16, 4, 64, 8, 32, 40
    C_ = heads * d
    qq, kk, vv = q(R(B, N, C_, seed=31), dtype), q(R(2, L, C_, seed=32), dtype), q(R(2, L, C_, seed=33), dtype)
    kf, vf = kk.repeat_interleave(2, 0), vv.repeat_interleave(2, 0)
    ref = _attn_ref(_heads(qq, heads), _heads(kf, heads), _heads(vf, heads)).transpose(1, 2).reshape(B, N, C_)
    out = ops.attention(qq.to(dev, dtype), kk.to(dev, dtype), _vt(vv, heads, dev, dtype), L, heads, kv_batch_div=2)
    assert rel_err(out, ref) < TOL[dtype]


def test_attention_online_softmax_rescale_forced(dev):
    """a late key tile carries a much larger score than the early ones, forcing the running-max rescale branch"""
    from ap_adapter_amd import ops
    dtype, B, N, heads, d, L = torch.bfloat16, 1, 64, 2, 32, 200
    C_ = heads * d
    qq, kk, vv = q(R(B, N, C_, seed=34), dtype), q(R(B, L, C_, seed=35), dtype), q(R(B, L, C_, seed=36), dtype)
    kk[:, 150] = qq[:, 5] * 4.0  # spikes q.k for query 5 (and others) in tile 4
    kk = q(kk, dtype)
    ref = _attn_ref(_heads(qq, heads), _heads(kk, heads), _heads(vv, heads)).transpose(1, 2).reshape(B, N, C_)
    out = ops.attention(qq.to(dev, dtype), kk.to(dev, dtype), _vt(vv, heads, dev, dtype), L, heads)
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C_", [(1000, 256), (252, 384), (130, 640), (513, 768), (7, 1280)])
def test_layernorm(dev, dtype, M, C_):
    from ap_adapter_amd import ops
    x, g, b = q(R(M, C_, seed=37) * 2 + 0.5, dtype), q(1 + 0.1 * R(C_, seed=38), dtype), q(0.1 * R(C_, seed=39), dtype)
    for eps in (1e-5, 1e-6):
        ref = F.layer_norm(x, (C_,), g, b, eps)
        out = ops.layer_norm(x.to(dev, dtype), g.to(dev, dtype), b.to(dev, dtype), eps)
        assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,C_,silu", [(2, 4000, 128, True), (3, 1000, 256, False), (2, 252, 384, True),
                                          (2, 64, 1280, True), (1, 1000, 640, True), (2, 64, 896, False)])
def test_groupnorm(dev, dtype, B, HW, C_, silu):
    from ap_adapter_amd import ops
    x = q(R(B, HW, C_, seed=40) * 1.5 + 0.3, dtype)
    g, b = q(1 + 0.1 * R(C_, seed=41), dtype), q(0.1 * R(C_, seed=42), dtype)
    for eps in (1e-5, 1e-6):
        ref = F.group_norm(x.transpose(1, 2), 32, g, b, eps)
        ref = (F.silu(ref) if silu else ref).transpose(1, 2)
        out = ops.group_norm(x.to(dev, dtype), g.to(dev, dtype), b.to(dev, dtype), 32, eps, silu=silu)
        assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,Ba,Bb,HW,Ca,Cb,silu", [(4, 4, 4, 4000, 128, 128, True), (4, 4, 2, 1000, 384, 256, True), (6, 3, 6, 252, 640, 384, False),
                                                    (4, 4, 2, 64, 640, 640, True), (2, 2, 1, 1000, 256, 128, True)])
def test_groupnorm_two_sources(dev, dtype, B, Ba, Bb, HW, Ca, Cb, silu):
    """GroupNorm (+ SiLU) of the channel concatenation of two tensors that is never materialised, the smaller batch read modulo
    (two-pass and one-pass kernels; groups that straddle the two sources: 640 = 384 + 256 has 20-channel groups) -- against fp32
    torch on the concatenation and BIT-equal to the single-source kernel on torch.cat"""
    from ap_adapter_amd import ops
    xa, xb = q(R(Ba, HW, Ca, seed=91) + 0.3, dtype), q(R(Bb, HW, Cb, seed=92, std=2.0), dtype)
    C_ = Ca + Cb
    g, b = q(R(C_, seed=93) * 0.2 + 1.0, dtype), q(R(C_, seed=94) * 0.2, dtype)
    cat = torch.cat([xa.repeat(B // Ba, 1, 1), xb.repeat(B // Bb, 1, 1)], -1)
    ref = F.group_norm(cat.permute(0, 2, 1), 32, g, b, 1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(q(ref, dtype))
    D = lambda t: t.to(dev, dtype)
    out = ops.group_norm2(D(xa), D(xb), D(g), D(b), 32, 1e-5, silu=silu)
    assert rel_err(out, ref) < TOL[dtype]
    assert torch.equal(out, ops.group_norm(D(cat), D(g), D(b), 32, 1e-5, silu=silu))


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("Ba,Bb,T,Ca,Cb,N", [(4, 4, 1000, 256, 128, 256), (4, 2, 252, 640, 384, 384), (2, 4, 64, 640, 640, 640), (64, 32, 1000, 256, 128, 256),
                                               (8, 4, 4000, 128, 128, 128)])
def test_linear_two_sources(dev, dtype, Ba, Bb, T, Ca, Cb, N):
    """the up-block resnets' 1x1 shortcut over [hidden | skip] without the concatenation (tiled kernel below 16000 rows, big-tile
    LDS-DMA kernel from there on; K-group shapes): against fp32 torch and BIT-equal to the Linear of torch.cat"""
    from ap_adapter_amd import ops
    B = max(Ba, Bb)
    xa, xb = q(R(Ba, T, Ca, seed=95), dtype), q(R(Bb, T, Cb, seed=96), dtype)
    w, b = q(R(N, Ca + Cb, seed=97, std=0.05), dtype), q(R(N, seed=98), dtype)
    cat = torch.cat([xa.repeat(B // Ba, 1, 1), xb.repeat(B // Bb, 1, 1)], -1)
    ref = F.linear(cat, w, b)
    D = lambda t: t.to(dev, dtype)
    out = ops.linear2(D(xa), D(xb), D(w), D(b))
    assert rel_err(out, ref) < TOL[dtype]
    assert torch.equal(out, ops.linear(D(cat), D(w), D(b)))


@pytest.mark.parametrize("tp,fp", [(1, 1), (2, 2), (4, 4), (8, 8), (8, 1), (2, 8)])
def test_audiomae_pool(dev, tp, fp):
    from ap_adapter_amd import ops
    from oracle.audiomae import pool
    dtype = torch.bfloat16
    rep = q(R(2, 513, 768, seed=43), dtype)
    ref = pool(rep, tp, fp)
    out = ops.audiomae_pool(rep.to(dev, dtype), tp, fp, out_dtype=torch.float32)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 1e-6
    out = ops.audiomae_pool(rep.to(dev, dtype), tp, fp)
    assert rel_err(out, ref) < TOL[dtype]
    rep32 = R(2, 513, 768, seed=43)
    out = ops.audiomae_pool(rep32.to(dev), tp, fp)
    assert out.dtype == torch.float32 and rel_err(out, pool(rep32, tp, fp)) < 1e-6


def test_timestep_embedding(dev):
    from ap_adapter_amd import ops
    from oracle.blocks import timestep_embedding
    t = torch.tensor([996.0, 991.0, 501.0, 1.0, 0.0])
    ref = timestep_embedding(t, 128, True, 0.0)
    out = ops.timestep_embedding(t.to(dev), 128, True, 0.0, torch.bfloat16)
    assert rel_err(out, ref) < TOL[torch.bfloat16]
    out = ops.timestep_embedding(t.to(dev), 128, False, 1.0, torch.float16)
    assert rel_err(out, timestep_embedding(t, 128, False, 1.0)) < TOL[torch.float16]
    # fp32 mode: the frequency exponent is formed like torch's and exp is correctly rounded; torch's vectorised exp is
    # within 1 ulp, and 1 ulp of a frequency near 1 is 6e-5 in the sin / cos argument at t ~ 1000
    out = ops.timestep_embedding(t.to(dev), 128, True, 0.0, torch.float32)
    assert rel_err(out, ref) < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cfg_ddim_step_matches_oracle(dev, dtype):
    from ap_adapter_amd import ops
    from ap_adapter_amd.scheduler import DDIMScheduler
    from oracle import ddim
    B, n, steps, gs = 3, 4000 * 8, 10, 7.5
    s = DDIMScheduler()
    s.set_timesteps(steps)
    coef = s.coef_table().to(dev)
    lat = R(B, n, seed=44)
    lat_d = lat.clone().to(dev)
    unet_in = torch.empty(B, n, dtype=dtype, device=dev)
    eps_out = torch.empty(B, n, dtype=torch.float32, device=dev)
    step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
    acp = ddim.alphas_cumprod()
    for i, t in enumerate(ddim.timesteps(steps)):
        eps2 = q(R(2 * B, n, seed=100 + i) * 0.5, dtype)
        e = q(ddim.cfg_combine(eps2, gs), dtype)
        lat = ddim.ddim_step(e, t, lat, steps, acp)
        ops.cfg_ddim_step(eps2.to(dev, dtype), lat_d, unet_in, coef, step_ptr, gs, eps_out)
        ops.step_advance(step_ptr)
        assert rel_err(eps_out, e) < 1e-6
        assert rel_err(lat_d, lat) < 1e-5
        assert rel_err(unet_in, lat) < TOL[dtype]
    assert int(step_ptr.item()) == steps


def test_errors_are_python_exceptions(dev):
    from ap_adapter_amd import ops
    x = torch.zeros(4, 12, dtype=torch.bfloat16, device=dev)  # K % 8 != 0
    w = torch.zeros(8, 12, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.linear(x, w)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ops.linear(torch.zeros(4, 16, dtype=torch.bfloat16), w)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K", [(1000, 256), (333, 384), (128, 256), (4000, 384), (33000, 256)])  # >= 32768 rows: 8-wave workgroups
@pytest.mark.parametrize("ln", [False, True])
def test_rowpanel_plain_bias_act_residual(dev, dtype, M, K, ln):
    from ap_adapter_amd import ops
    N = K
    x = q(R(M, K, seed=50) * 1.5 + 0.2, dtype)
    w, b, r = q(R(N, K, seed=51, std=0.05), dtype), q(R(N, seed=52, std=0.3), dtype), q(R(M, N, seed=53), dtype)
    g, be = q(1 + 0.1 * R(K, seed=54), dtype), q(0.1 * R(K, seed=55), dtype)
    xin = q(F.layer_norm(x, (K,), g, be, 1e-5), dtype) if ln else x
    lnp = (g.to(dev, dtype), be.to(dev, dtype), 1e-5) if ln else None
    for act, fn in ((None, lambda t: t), ("silu", F.silu), ("gelu", F.gelu)):
        ref = fn(F.linear(xin, w, b))
        out = ops.fused_linear(x.to(dev, dtype), w.to(dev, dtype), b.to(dev, dtype), ln=lnp, act=act)
        assert rel_err(out, ref) < TOL[dtype]
    ref = F.linear(xin, w, b) + r
    out = ops.fused_linear(x.to(dev, dtype), w.to(dev, dtype), b.to(dev, dtype), ln=lnp, residual=r.to(dev, dtype))
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K", [(63990, 256), (100003, 256), (16128, 384), (70001, 384), (17, 256)])
def test_to_out_residual_launch_lean_tile_loop(dev, dtype, M, K):
    """to_out / proj_out of the two large levels: Linear + bias + residual through the weight-stationary kernel's residual form (x panel and the
    residual rows of all the wave's tiles requested up front, two tiles per step, no segment / activation logic in the loop).  Ragged last panel, more than
    one panel per wave (100003 rows > 2048 waves x 32; 70001 rows at K = 384: 3 column slices x 85 row groups), and IN PLACE on the residual buffer as the
    transformer blocks call it."""
    from ap_adapter_amd import ops
    x = q(R(M, K, seed=150), dtype)
    w, b, r = q(R(K, K, seed=151, std=0.05), dtype), q(R(K, seed=152, std=0.3), dtype), q(R(M, K, seed=153), dtype)
    ref = F.linear(x, w, b) + r
    xd, wd, bd = x.to(dev, dtype), w.to(dev, dtype), b.to(dev, dtype)
    out = ops.fused_linear(xd, wd, bd, residual=r.to(dev, dtype))
    assert rel_err(out, ref) < TOL[dtype]
    buf = r.to(dev, dtype).clone()
    out2 = ops.fused_linear(xd, wd, bd, residual=buf, out=buf)
    assert out2.data_ptr() == buf.data_ptr() and torch.equal(out2, out)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K", [(1000, 256), (300, 384), (33000, 256), (32900, 384)])
@pytest.mark.parametrize("ln", [False, True])
def test_rowpanel_geglu(dev, dtype, M, K, ln):
    from ap_adapter_amd import ops
    N = 4 * K
    x = q(R(M, K, seed=56), dtype)
    w, b = q(R(2 * N, K, seed=57, std=0.08), dtype), q(R(2 * N, seed=58, std=0.5), dtype)
    g, be = q(1 + 0.1 * R(K, seed=59), dtype), q(0.1 * R(K, seed=60), dtype)
    xin = q(F.layer_norm(x, (K,), g, be, 1e-5), dtype) if ln else x
    a, gate = F.linear(xin, w, b).chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    out = ops.fused_linear(x.to(dev, dtype), w.to(dev, dtype), b.to(dev, dtype),
                           ln=(g.to(dev, dtype), be.to(dev, dtype), 1e-5) if ln else None, act="geglu")
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("M,C", [(1000, 256), (300, 256), (33000, 256), (37, 256), (16500, 256), (32000, 256)])
@pytest.mark.parametrize("ln", [False, True])
def test_geglu_mlp(dev, dtype, M, C, ln):
    """norm3 + GEGLU + FeedForward.net[2] + residual in one launch (the 128-token-workgroup kernel from unpacked weights; ragged last panels)"""
    from ap_adapter_amd import ops
    x = q(R(M, C, seed=156), dtype)
    w1, b1 = q(R(8 * C, C, seed=157, std=0.08), dtype), q(R(8 * C, seed=158, std=0.5), dtype)
    w2, b2 = q(R(C, 4 * C, seed=159, std=0.04), dtype), q(R(C, seed=160, std=0.5), dtype)
    g, be = q(1 + 0.1 * R(C, seed=161), dtype), q(0.1 * R(C, seed=162), dtype)
    xin = q(F.layer_norm(x, (C,), g, be, 1e-5), dtype) if ln else x
    a, gate = F.linear(xin, w1, b1).chunk(2, dim=-1)
    ref = x + F.linear(a * F.gelu(gate), w2, b2)
    lnp = (g.to(dev, dtype), be.to(dev, dtype), 1e-5) if ln else None
    xd = x.to(dev, dtype)
    out = ops.geglu_mlp(xd, w1.to(dev, dtype), b1.to(dev, dtype), w2.to(dev, dtype), b2.to(dev, dtype), ln=lnp)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[dtype]
    # in place (out aliases x) gives the same bits
    out2 = ops.geglu_mlp(xd, w1.to(dev, dtype), b1.to(dev, dtype), w2.to(dev, dtype), b2.to(dev, dtype), ln=lnp, out=xd)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("M", [1000, 37, 256, 257, 8000])
@pytest.mark.parametrize("ln,bias", [(True, True), (False, True), (True, False)])
def test_geglu_mlp_packed(dev, dtype, M, ln, bias):
    """the same operator on the 64-token register-block kernel from packed weights (apad_mlp_pack + apad_geglu_mlp_packed): against fp32 torch on
    the storage-rounded operands, and BIT-EQUAL to the 128- / 64-token kernels -- the same MFMA chain per 16-unit chunk (b1 as the accumulators'
    initial value in all of them), the same GELU arithmetic and rounding points; ragged / partial workgroups"""
    from ap_adapter_amd import ops
    C = 256
    x = q(R(M, C, seed=256), dtype)
    w1, b1 = q(R(8 * C, C, seed=257, std=0.08), dtype), (q(R(8 * C, seed=258, std=0.5), dtype) if bias else None)
    w2, b2 = q(R(C, 4 * C, seed=259, std=0.04), dtype), (q(R(C, seed=260, std=0.5), dtype) if bias else None)
    g, be = q(1 + 0.1 * R(C, seed=261), dtype), q(0.1 * R(C, seed=262), dtype)
    xin = q(F.layer_norm(x, (C,), g, be, 1e-5), dtype) if ln else x
    a, gate = F.linear(xin, w1, b1).chunk(2, dim=-1)
    ref = x + F.linear(a * F.gelu(gate), w2, b2)
    lnp = (g.to(dev, dtype), be.to(dev, dtype), 1e-5) if ln else None
    dv = lambda t: None if t is None else t.to(dev, dtype)
    xd = dv(x)
    wp, bp = ops.mlp_pack(dv(w1), dv(b1), dv(w2))
    out = ops.geglu_mlp_packed(xd, wp, bp, dv(b2), ln=lnp)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL[dtype]
    # bit-equal to the 128- / 64-token kernels of the small launches (a row's result must not depend on which kernel its batch size selects)
    assert torch.equal(out, ops.geglu_mlp(xd, dv(w1), dv(b1), dv(w2), dv(b2), ln=lnp))
    # in place (out aliases x) gives the same bits
    assert torch.equal(ops.geglu_mlp_packed(xd, wp, bp, dv(b2), ln=lnp, out=xd), out)


def test_geglu_mlp_packed_at_the_product_launch_size(dev):
    """apad_geglu_mlp_packed at the size the denoise step launches it (63 993 rows: 250 workgroups, a buffer-addressed ragged tail) DIRECTLY
    against fp32 torch on the storage-rounded operands (reference evaluated in fp32 on the device: 67 GFLOP)"""
    from ap_adapter_amd import ops
    dtype, C, M = torch.bfloat16, 256, 63993
    x = q(R(M, C, seed=556), dtype)
    w1, b1 = q(R(8 * C, C, seed=557, std=0.08), dtype), q(R(8 * C, seed=558, std=0.5), dtype)
    w2, b2 = q(R(C, 4 * C, seed=559, std=0.04), dtype), q(R(C, seed=560, std=0.5), dtype)
    g, be = q(1 + 0.1 * R(C, seed=561), dtype), q(0.1 * R(C, seed=562), dtype)
    f = lambda t: t.to(dev, torch.float32)
    xin = F.layer_norm(f(x), (C,), f(g), f(be), 1e-5).to(dtype).float()
    a, gate = F.linear(xin, f(w1), f(b1)).chunk(2, dim=-1)
    ref = (f(x) + F.linear(a * F.gelu(gate), f(w2), f(b2))).cpu()
    dv = lambda t: t.to(dev, dtype)
    wp, bp = ops.mlp_pack(dv(w1), dv(b1), dv(w2))
    out = ops.geglu_mlp_packed(dv(x), wp, bp, dv(b2), ln=(dv(g), dv(be), 1e-5))
    assert rel_err(out, ref) < TOL[dtype]
    # every row block on its own, the last (ragged) workgroups included
    err = (out.float().cpu() - ref).abs().amax(dim=1)
    assert float(err[-300:].max()) < TOL[dtype] * float(ref.abs().max()) and float(err[:300].max()) < TOL[dtype] * float(ref.abs().max())


def test_layernorm_geglu_packed_at_the_product_launch_size_ragged(dev):
    """apad_layernorm_geglu_packed at 16 100 rows (the 252-token level's launch size with a ragged last tile) directly against fp32 torch"""
    from ap_adapter_amd import ops
    dtype, C, M = torch.bfloat16, 384, 16100
    x = q(R(M, C, seed=656), dtype)
    w1, b1 = q(R(8 * C, C, seed=657, std=0.06), dtype), q(R(8 * C, seed=658, std=0.5), dtype)
    g, be = q(1 + 0.1 * R(C, seed=661), dtype), q(0.1 * R(C, seed=662), dtype)
    f = lambda t: t.to(dev, torch.float32)
    xin = F.layer_norm(f(x), (C,), f(g), f(be), 1e-5).to(dtype).float()
    a, gate = F.linear(xin, f(w1), f(b1)).chunk(2, dim=-1)
    ref = (a * F.gelu(gate)).cpu()
    dv = lambda t: t.to(dev, dtype)
    wp, bp = ops.geglu_pack(dv(w1), dv(b1))
    out = ops.layernorm_geglu_packed(dv(x), wp, bp, ln=(dv(g), dv(be), 1e-5))
    assert out.shape == ref.shape and rel_err(out, ref) < TOL[dtype]
    err = (out.float().cpu() - ref).abs().amax(dim=1)
    assert float(err[-300:].max()) < TOL[dtype] * float(ref.abs().max())


def test_geglu_mlp_packed_route_and_repack(dev, monkeypatch):
    """FeedForward takes the packed kernel from ops.MLP_PACKED_MIN_M rows, packs once, and re-packs when a parameter is updated in place"""
    from ap_adapter_amd import ops
    from ap_adapter_amd.unet import FeedForward
    from ap_adapter_amd.synthetic import init_synthetic_
    dtype = torch.bfloat16
    ff = FeedForward(256)
    init_synthetic_(ff, 5, w_std=0.05, bias_std=0.05)
    ff = ff.to(dev, dtype).requires_grad_(False)
    ln = (torch.ones(256, device=dev, dtype=dtype), torch.zeros(256, device=dev, dtype=dtype), 1e-5)
    x = torch.randn(600, 256, device=dev).to(dtype)
    calls = []
    real = ops.geglu_mlp_packed
    monkeypatch.setattr(ops, "geglu_mlp_packed", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    a = ff(x, ln)
    assert not calls  # below the row threshold: the 128-token kernel
    monkeypatch.setattr(ops, "MLP_PACKED_MIN_M", 512)
    b = ff(x, ln)
    assert len(calls) == 1 and torch.equal(b, a)
    wp0 = ff._mlp3_w[0]
    assert ff(x, ln) is not None and ff._mlp3_w[0] is wp0  # packed once
    with torch.no_grad():
        ff.net[2].weight.mul_(0.5)
    c = ff(x, ln)
    assert ff._mlp3_w[0] is not wp0 and not torch.equal(b, c)
    monkeypatch.setattr(ops, "MLP_PACKED", False)
    assert torch.equal(c, ff(x, ln))


def test_geglu_mlp_outside_envelope_is_an_error(dev):
    from ap_adapter_amd import ops
    x = torch.zeros(64, 640, device=dev, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.geglu_mlp(x, torch.zeros(5120, 640, device=dev, dtype=torch.bfloat16), None,
                      torch.zeros(640, 2560, device=dev, dtype=torch.bfloat16), None)
    with pytest.raises(ValueError):  # fp32 storage: the un-fused exact-f32 chain's job
        ops.geglu_mlp(torch.zeros(64, 384, device=dev), torch.zeros(3072, 384, device=dev), None, torch.zeros(384, 1536, device=dev), None)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,L,K,heads", [(2, 1000, 256, 8), (3, 252, 384, 8), (2, 100, 256, 4), (1, 513, 384, 12), (2, 63, 256, 8),
                                         (34, 1000, 256, 8)])  # 34000 rows: the 8-wave workgroup variant, sample boundaries inside a workgroup
def test_rowpanel_fused_qkv_with_vt(dev, dtype, B, L, K, heads):
    """LayerNorm + q|k|v in one launch; V per-head transposed (incl. token counts that are not multiples of 4)"""
    from ap_adapter_amd import ops
    M, d = B * L, K // heads
    x = q(R(M, K, seed=61), dtype)
    w = q(R(3 * K, K, seed=62, std=0.06), dtype)
    g, be = q(1 + 0.1 * R(K, seed=63), dtype), q(0.1 * R(K, seed=64), dtype)
    xin = q(F.layer_norm(x, (K,), g, be, 1e-5), dtype)
    y = F.linear(xin, w)
    qo = torch.empty(M, K, dtype=dtype, device=dev)
    ko = torch.empty(M, K, dtype=dtype, device=dev)
    Lpad = ops.round_up(L, 32)
    vt = torch.zeros(B, heads, d, Lpad, dtype=dtype, device=dev)
    ops.rowpanel(x.to(dev, dtype), w.to(dev, dtype), [(qo, None, K, "row"), (ko, None, K, "row"), (vt, None, K, "vt")],
                 ln=(g.to(dev, dtype), be.to(dev, dtype), 1e-5), vt_geom=(heads, d, L, Lpad))
    assert rel_err(qo, y[:, :K]) < TOL[dtype]
    assert rel_err(ko, y[:, K:2 * K]) < TOL[dtype]
    ref_vt = y[:, 2 * K:].view(B, L, heads, d).permute(0, 2, 3, 1)
    assert rel_err(vt[..., :L], ref_vt) < TOL[dtype]
    if Lpad > L:
        assert float(vt[..., L:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,L,K,heads", [(2, 64, 640, 8), (3, 100, 128, 4), (64, 64, 640, 8)])
def test_tiled_fused_qkv(dev, dtype, B, L, K, heads):
    from ap_adapter_amd import ops
    M, d = B * L, K // heads
    x, w = q(R(M, K, seed=70), dtype), q(R(3 * K, K, seed=71, std=0.05), dtype)
    y = F.linear(x, w)
    qo = torch.empty(M, K, dtype=dtype, device=dev)
    ko = torch.empty(M, K, dtype=dtype, device=dev)
    Lpad = ops.round_up(L, 32)
    vt = torch.zeros(B, heads, d, Lpad, dtype=dtype, device=dev)
    ops.linear_qkv(x.to(dev, dtype), w.to(dev, dtype), B, L, heads, qo, ko, vt)
    assert rel_err(qo, y[:, :K]) < TOL[dtype]
    assert rel_err(ko, y[:, K:2 * K]) < TOL[dtype]
    assert rel_err(vt[..., :L], y[:, 2 * K:].view(B, L, heads, d).permute(0, 2, 3, 1)) < TOL[dtype]


def test_xattn_packings(dev):
    """the fragment-major packings the fused kernel reads: weights [slice][kk][lane][8], K / V^T in MFMA C-layout key order"""
    from ap_adapter_amd import ops
    dtype = torch.bfloat16
    w = q(R(256, 256, seed=300), dtype)
    pw = ops.xattn_pack_weight(w.to(dev, dtype)).float().cpu().view(8, 16, 2, 32, 8)  # [s][kk][half][l31][e]
    ref = w.view(8, 32, 16, 2, 8).permute(0, 2, 3, 1, 4)                               # W[s*32+l31][kk*16+half*8+e]
    assert torch.equal(pw, ref)
    B, L, H, D = 3, 40, 8, 32
    k, v = q(R(B, L, 256, seed=301), dtype), q(R(B, L, 256, seed=302), dtype)
    pk = ops.xattn_pack_kv(k.to(dev, dtype), _vt(v, H, dev, dtype), L).float().cpu().view(B, H, 8, 2, 32, 8)  # [b][h][frag][half][l31][e]
    perm = lambda half, e: 4 * half + e if e < 4 else 8 + 4 * half + e - 4
    kp = torch.zeros(B, 64, 256)
    kp[:, :L] = k
    vp = torch.zeros(B, 64, 256)
    vp[:, :L] = v
    for b, h, half, l31, e in [(0, 0, 0, 0, 0), (1, 3, 1, 17, 5), (2, 7, 0, 31, 7), (1, 5, 1, 9, 2)]:
        for sub_, kk in [(0, 0), (0, 1), (1, 0), (1, 1)]:
            assert pk[b, h, sub_ * 2 + kk, half, l31, e] == kp[b, sub_ * 32 + l31, h * D + kk * 16 + perm(half, e)]
        for st in range(3):  # keys < 48 (L = 40 -> 3 steps of 16)
            assert pk[b, h, 4 + st, half, l31, e] == vp[b, st * 16 + perm(half, e), h * D + l31]


def _xattn_ref(x, g, be, wq, wo, bo, ehs_t, wk, wv, heads, bias=None, ehs_a=None, wk_ip=None, wv_ip=None, scale=0.0):
    """fp32 restatement of LayerNorm -> to_q -> (decoupled) attention -> to_out + residual (attention_processor.py:347-470)"""
    B, N, C = x.shape
    d = C // heads
    hs = F.layer_norm(x, (C,), g, be, 1e-5)
    sp = lambda t: t.reshape(B, t.shape[1], heads, d).transpose(1, 2)
    qq = sp(F.linear(hs, wq))
    m = None if bias is None else bias[:, None, None, :]
    o = F.scaled_dot_product_attention(qq, sp(F.linear(ehs_t, wk)), sp(F.linear(ehs_t, wv)), attn_mask=m)
    if ehs_a is not None:
        o = o + scale * F.scaled_dot_product_attention(qq, sp(F.linear(ehs_a, wk_ip)), sp(F.linear(ehs_a, wv_ip)))
    o = o.transpose(1, 2).reshape(B, N, C)
    return x + F.linear(o, wo, bo)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,N,Lt,La,masked", [(2, 1000, 8, 32, False), (3, 100, 8, 8, False), (2, 250, 16, 0, True), (1, 33, 8, 64, False),
                                               (5, 64, 40, 0, False), (2, 130, 8, 33, False), (9, 1000, 8, 32, False), (2, 31, 40, 50, True),
                                               (2, 1000, 8, 128, False), (3, 77, 8, 128, False), (9, 250, 8, 128, False),
                                               (2, 1000, 8, 512, False), (3, 300, 8, 256, False), (2, 77, 8, 192, False), (9, 250, 8, 512, False)])
def test_fused_cross_attention(dev, dtype, B, N, Lt, La, masked):
    """LayerNorm + to_q + (decoupled) attention + to_out + residual in one launch, incl. panels that end inside a sample,
    a tail workgroup with idle waves, the masked T5 form, a 2-sub-tile segment, the 8 + 128-key form of the timbre /
    accompaniment presets (q of all four panels projected first, one panel attended at a time) and the chunked form of longer audio segments
    (192 / 256 / 512 keys: 64-key chunks with a running max / sum; tiles inside one sample and tiles that cross a sample boundary)"""
    from ap_adapter_amd import ops
    C, H = 256, 8
    x = q(R(B, N, C, seed=201), dtype)
    g, be = q(1 + 0.1 * R(C, seed=202), dtype), q(0.1 * R(C, seed=203), dtype)
    wq, wo, bo = q(R(C, C, seed=204, std=0.06), dtype), q(R(C, C, seed=205, std=0.06), dtype), q(R(C, seed=206, std=0.3), dtype)
    wk, wv = q(R(C, 768, seed=207, std=0.04), dtype), q(R(C, 768, seed=208, std=0.04), dtype)
    wki, wvi = q(R(C, 768, seed=209, std=0.04), dtype), q(R(C, 768, seed=210, std=0.04), dtype)
    et = q(R(B, Lt, 768, seed=211), dtype)
    ea = q(R(B, La, 768, seed=212), dtype) if La else None
    if La > 128:
        # the chunked form carries a running maximum / sum across its 64-key chunks and rescales the accumulator on every chunk: audio tokens late
        # in the segment that dominate the scores (a few rows scaled up so that later chunks raise some queries' maxima by a large factor) exercise
        # a rescale that is not ~1; the other samples keep ordinary data
        ea[0, 70:74] *= 6.0
        ea[0, La - 30:La - 27] *= 12.0
        ea = q(ea, dtype)
    bias = None
    if masked:
        bias = torch.zeros(B, Lt)
        bias[1::2, -4:] = -10000.0
    ref = _xattn_ref(x, g, be, wq, wo, bo, et, wk, wv, H, bias, ea, wki, wvi, 0.55)
    D = lambda t: t.to(dev, dtype)
    k1 = ops.linear(D(et), D(wk))
    v1t = torch.zeros(B, H, C // H, ops.round_up(Lt, 32), device=dev, dtype=dtype)
    ops.linear_vt(D(et), D(wv), B, Lt, H, v1t)
    k2 = v2t = None
    if La:
        k2 = ops.linear(D(ea), D(wki))
        v2t = torch.zeros(B, H, C // H, ops.round_up(La, 32), device=dev, dtype=dtype)
        ops.linear_vt(D(ea), D(wvi), B, La, H, v2t)
    (wq_p, q_fold), wo_p = ops.xattn_pack_weight(D(wq), (D(g), D(be), 1e-5)), ops.xattn_pack_weight(D(wo))
    pk1 = ops.xattn_pack_kv(k1, v1t, Lt)
    pk2 = ops.xattn_pack_kv(k2, v2t, La) if La else None
    out = ops.fused_cross_attention(D(x), wq_p, wo_p, D(bo), pk1, Lt, H, ln=(D(g), D(be), 1e-5),
                                    key_bias=None if bias is None else bias.to(dev), kv2_packed=pk2, L2=La, scale2=0.55, q_fold=q_fold)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 1.5 * TOL[dtype]
    # and against the three-kernel chain it replaces
    qd = ops.fused_linear(D(x), D(wq), ln=(D(g), D(be), 1e-5))
    od = ops.attention(qd, k1, v1t, Lt, H, key_bias=None if bias is None else bias.to(dev), k2=k2, vt2=v2t, L2=La, scale2=0.55)
    chain = ops.fused_linear(od, D(wo), D(bo), residual=D(x))
    assert rel_err(out, chain.float().cpu()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("shift,outlier", [(50.0, 1.0), (0.0, 40.0), (-30.0, 25.0)])
def test_fused_cross_attention_rows_with_a_large_mean(dev, dtype, shift, outlier):
    """the one-launch attn2 kernel applies its LayerNorm by algebra on the RAW rows (q = rstd (W' x) - mean rstd rowsum(W') + W beta): residual
    streams whose rows have |mean| >> std and a few outlier channels -- where that form cancels in fp32 -- against fp32 torch and the chain"""
    from ap_adapter_amd import ops
    B, N, Lt, La, C, H = 2, 300, 8, 32, 256, 8
    x = R(B, N, C, seed=401) + shift
    x[:, :, 7] *= outlier
    x[:, :, 130] -= 3.0 * outlier
    x = q(x, dtype)
    g, be = q(1 + 0.1 * R(C, seed=402), dtype), q(0.1 * R(C, seed=403), dtype)
    wq, wo, bo = q(R(C, C, seed=404, std=0.06), dtype), q(R(C, C, seed=405, std=0.06), dtype), q(R(C, seed=406, std=0.3), dtype)
    wk, wv = q(R(C, 768, seed=407, std=0.04), dtype), q(R(C, 768, seed=408, std=0.04), dtype)
    wki, wvi = q(R(C, 768, seed=409, std=0.04), dtype), q(R(C, 768, seed=410, std=0.04), dtype)
    et, ea = q(R(B, Lt, 768, seed=411), dtype), q(R(B, La, 768, seed=412), dtype)
    ref = _xattn_ref(x, g, be, wq, wo, bo, et, wk, wv, H, None, ea, wki, wvi, 0.55)
    D = lambda t: t.to(dev, dtype)
    k1, k2 = ops.linear(D(et), D(wk)), ops.linear(D(ea), D(wki))
    v1t = torch.zeros(B, H, C // H, ops.round_up(Lt, 32), device=dev, dtype=dtype)
    v2t = torch.zeros(B, H, C // H, ops.round_up(La, 32), device=dev, dtype=dtype)
    ops.linear_vt(D(et), D(wv), B, Lt, H, v1t)
    ops.linear_vt(D(ea), D(wvi), B, La, H, v2t)
    (wq_p, q_fold), wo_p = ops.xattn_pack_weight(D(wq), (D(g), D(be), 1e-5)), ops.xattn_pack_weight(D(wo))
    out = ops.fused_cross_attention(D(x), wq_p, wo_p, D(bo), ops.xattn_pack_kv(k1, v1t, Lt), Lt, H, ln=(D(g), D(be), 1e-5),
                                    kv2_packed=ops.xattn_pack_kv(k2, v2t, La), L2=La, scale2=0.55, q_fold=q_fold)
    # the residual dominates the output (|x| up to shift + outliers): the attention contribution is what has to be right
    delta, dref = out.float().cpu() - x.float(), ref - x.float()
    assert float((delta - dref).abs().max() / dref.abs().max()) < 3 * TOL[dtype] + float(x.abs().max()) * 2 ** -8 / float(dref.abs().max())
    qd = ops.fused_linear(D(x), D(wq), ln=(D(g), D(be), 1e-5))
    od = ops.attention(qd, k1, v1t, Lt, H, k2=k2, vt2=v2t, L2=La, scale2=0.55)
    chain = ops.fused_linear(od, D(wo), D(bo), residual=D(x))
    dchain = chain.float().cpu() - x.float()
    print(f"shift {shift} outlier {outlier}: fused vs fp32 {float((delta - dref).abs().max() / dref.abs().max()):.3e}, chain vs fp32 "
          f"{float((dchain - dref).abs().max() / dref.abs().max()):.3e}")
    assert float((delta - dref).abs().max()) < 2.0 * float((dchain - dref).abs().max()) + 1e-6


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,N,Lt,La,masked,C", [
    (2, 252, 8, 32, False, 384), (3, 100, 8, 8, False, 384), (2, 252, 16, 0, True, 384), (1, 33, 8, 64, False, 384), (5, 64, 40, 0,
    False, 384), (2, 130, 8, 33, False, 384), (9, 252, 8, 32, True, 384), (2, 31, 64, 50, True, 384), (64, 252, 8, 32, False, 384),
    (2, 65, 1, 1, False, 384), (2, 252, 8, 128, False, 384), (3, 100, 8, 70, True, 384), (64, 252, 8, 128, False, 384), (2, 60, 32,
    97, False, 384), (2, 252, 8, 256, False, 384), (3, 100, 8, 129, False, 384), (2, 70, 8, 200, True, 384), (64, 252, 8, 512, False, 384),
    (2, 33, 32, 320, False, 384)])
def test_cross_attention_rows(dev, dtype, B, N, Lt, La, masked, C, monkeypatch):
    """the 384- and 640-wide levels' single-launch form (apad_cross_attention_rows: 64- / 32-token row tiles in LDS through LayerNorm, to_q,
    attention, to_out, residual): ragged last tiles, one- and two-segment forms, the masked forms of both, the 8 + 128-key form of the timbre /
    accompaniment presets (second segment's fragments requested as they are used), 129 .. 512 audio keys in 64-key chunks with a running maximum /
    sum (pooling 1 and the mixed poolings, ragged last chunk), full CFG batch; against fp32 torch on
    storage-rounded operands and against the three-kernel chain it replaces"""
    from ap_adapter_amd import ops
    H = 8
    x = q(R(B, N, C, seed=301), dtype)
    g, be = q(1 + 0.1 * R(C, seed=302), dtype), q(0.1 * R(C, seed=303), dtype)
    wq, wo, bo = q(R(C, C, seed=304, std=0.05), dtype), q(R(C, C, seed=305, std=0.05), dtype), q(R(C, seed=306, std=0.3), dtype)
    wk, wv = q(R(C, 768, seed=307, std=0.04), dtype), q(R(C, 768, seed=308, std=0.04), dtype)
    wki, wvi = q(R(C, 768, seed=309, std=0.04), dtype), q(R(C, 768, seed=310, std=0.04), dtype)
    et = q(R(B, Lt, 768, seed=311), dtype)
    ea = q(R(B, La, 768, seed=312), dtype) if La else None
    bias = None
    if masked:
        bias = torch.zeros(B, Lt)
        bias[1::2, -(Lt // 2):] = -10000.0
    ref = _xattn_ref(x, g, be, wq, wo, bo, et, wk, wv, H, bias, ea, wki, wvi, 0.55)
    D = lambda t: t.to(dev, dtype)
    k1 = ops.linear(D(et), D(wk))
    v1t = torch.zeros(B, H, C // H, ops.round_up(Lt, 32), device=dev, dtype=dtype)
    ops.linear_vt(D(et), D(wv), B, Lt, H, v1t)
    k2 = v2t = None
    if La:
        k2 = ops.linear(D(ea), D(wki))
        v2t = torch.zeros(B, H, C // H, ops.round_up(La, 32), device=dev, dtype=dtype)
        ops.linear_vt(D(ea), D(wvi), B, La, H, v2t)
    wq_p, wo_p = ops.xrows_pack_weight(D(wq)), ops.xrows_pack_weight(D(wo))
    bd = None if bias is None else bias.to(dev)
    xd = D(x)
    out = ops.cross_attention_rows(xd, wq_p, wo_p, D(bo), k1, v1t, H, ln=(D(g), D(be), 1e-5), key_bias=bd, k2=k2, vt2=v2t, scale2=0.55)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 1.5 * TOL[dtype]
    # the same launch over the fragment-packed key / value sets (what the processors hand it): the same fragments, bit-equal
    p1, p2 = ops.rows_pack_kv(k1, v1t), (ops.rows_pack_kv(k2, v2t) if La else None)
    assert torch.equal(ops.cross_attention_rows(xd, wq_p, wo_p, D(bo), p1, None, H, ln=(D(g), D(be), 1e-5), key_bias=bd, k2=p2, scale2=0.55), out)
    if La:  # ... and mixed: one segment packed, the other not
        assert torch.equal(ops.cross_attention_rows(xd, wq_p, wo_p, D(bo), k1, v1t, H, ln=(D(g), D(be), 1e-5), key_bias=bd, k2=p2, scale2=0.55), out)
    qd = ops.fused_linear(xd, D(wq), ln=(D(g), D(be), 1e-5))
    od = ops.attention(qd, k1, v1t, Lt, H, key_bias=bd, k2=k2, vt2=v2t, L2=La, scale2=0.55)
    chain = ops.fused_linear(od, D(wo), D(bo), residual=xd)
    assert rel_err(out, chain.float().cpu()) < TOL[dtype]
    # in place (out aliases x), and without the bias of to_out
    x2 = xd.clone()
    ops.cross_attention_rows(x2, wq_p, wo_p, D(bo), k1, v1t, H, ln=(D(g), D(be), 1e-5), key_bias=bd, k2=k2, vt2=v2t, scale2=0.55, out=x2)
    assert torch.equal(x2, out)
    nb = ops.cross_attention_rows(xd, wq_p, wo_p, None, k1, v1t, H, ln=(D(g), D(be), 1e-5), key_bias=bd, k2=k2, vt2=v2t, scale2=0.55)
    assert rel_err(nb, ref - bo) < 1.5 * TOL[dtype]


def test_cross_attention_rows_outside_envelope(dev, monkeypatch):
    from ap_adapter_amd import ops
    assert ops.XROWS_C == (384,)
    bf = torch.bfloat16
    assert ops.xrows_ok(384, 8, 8, 32) and ops.xrows_ok(384, 8, 64, 64) and ops.xrows_ok(384, 8, 16)
    assert ops.xrows_ok(384, 8, 8, 128) and ops.xrows_ok(384, 8, 32, 128) and not ops.xrows_ok(384, 8, 40, 128) and ops.xrows_ok(384, 8, 8, 129) and ops.xrows_ok(384, 8, 8, 512) and not ops.xrows_ok(384, 8, 8, 513)
    assert not ops.xrows_ok(640, 8, 8, 32) and not ops.xrows_ok(1280, 8, 8, 32) and not ops.xrows_ok(256, 8, 8, 32)
    assert not ops.xrows_ok(384, 4, 8, 32) and not ops.xrows_ok(384, 8, 65)
    x = torch.zeros(1, 64, 512, device=dev, dtype=bf)
    w = ops.xrows_pack_weight(torch.zeros(512, 512, device=dev, dtype=bf))
    with pytest.raises(ValueError):
        ops.cross_attention_rows(x, w, w, None, torch.zeros(1, 8, 512, device=dev, dtype=bf), torch.zeros(1, 8, 64, 32, device=dev, dtype=bf), 8)
    x = torch.zeros(1, 64, 384, device=dev, dtype=bf)
    w = ops.xrows_pack_weight(torch.zeros(384, 384, device=dev, dtype=bf))
    with pytest.raises(ValueError):  # more than 512 audio keys: the chain's job
        ops.cross_attention_rows(x, w, w, None, torch.zeros(1, 8, 384, device=dev, dtype=bf), torch.zeros(1, 8, 48, 32, device=dev, dtype=bf), 8,
                                 k2=torch.zeros(1, 576, 384, device=dev, dtype=bf), vt2=torch.zeros(1, 8, 48, 576, device=dev, dtype=bf))
    # the packing: fragment (row tile rt, k-step ks) = one contiguous KB, lane = (k half, row)
    wt = torch.arange(384 * 384, dtype=torch.float32).reshape(384, 384).to(dev)
    pk = ops.xrows_pack_weight(wt).reshape(12, 24, 2, 32, 8)
    assert torch.equal(pk[5, 7, 1, 9], wt[5 * 32 + 9, 7 * 16 + 8: 7 * 16 + 16])


@pytest.mark.parametrize("H,hd,Lk", [(8, 48, 8), (8, 48, 40), (8, 80, 128), (8, 80, 200), (2, 32, 33)])
def test_rows_pack_kv_layout(dev, H, hd, Lk):
    """apad_rows_pack_kv against the layout include/apadapter_hip.h states, element by element (bf16 values are exact small integers)"""
    from ap_adapter_amd import ops
    B, Cc, Lpad = 3, H * hd, ops.round_up(Lk, 32) + 32  # (a vt buffer wider than the packing needs)
    k = (torch.arange(B * Lk * Cc, dtype=torch.float32) % 251).reshape(B, Lk, Cc).to(dev, torch.bfloat16)
    vt = torch.zeros(B, H, hd, Lpad, device=dev, dtype=torch.bfloat16)
    vt[..., :Lk] = ((torch.arange(B * H * hd * Lk, dtype=torch.float32) * 7) % 253).reshape(B, H, hd, Lk).to(dev, torch.bfloat16)
    pk = ops.rows_pack_kv(k, vt)
    NU, KC, DTT = (Lk + 31) // 32, hd // 16, (hd + 31) // 32
    assert pk.data.numel() * 2 == L_bytes(ops, B, H, hd, Lk) == B * H * NU * (KC + 2 * DTT) * 1024 and pk.shape == (B, Lk, Cc)
    d = pk.data.reshape(B, H, NU * (KC + 2 * DTT), 64, 8).cpu().float()
    kc, vc = k.cpu().float(), vt.cpu().float()
    lane = torch.arange(64)
    l31, half = lane % 32, lane // 32
    for b, h in ((0, 0), (B - 1, H - 1)):
        for u in range(NU):
            for cc in range(KC):
                key = torch.clamp(u * 32 + l31, max=Lk - 1)
                exp = torch.stack([kc[b, key, h * hd + cc * 16 + half * 8 + e] for e in range(8)], 1)
                assert torch.equal(d[b, h, u * KC + cc], exp)
        for st in range(2 * NU):
            for dt in range(DTT):
                dd = dt * 32 + l31
                cols = [st * 16 + 4 * half + e for e in range(4)] + [st * 16 + 4 * half + 8 + e for e in range(4)]
                exp = torch.stack([torch.where(dd < hd, vc[b, h, torch.clamp(dd, max=hd - 1), c], torch.zeros(64)) for c in cols], 1)
                assert torch.equal(d[b, h, NU * KC + st * DTT + dt], exp)
    with pytest.raises(ValueError):
        ops.rows_pack_kv(k, vt[..., :Lpad - 16])


def L_bytes(ops, B, H, hd, Lk):
    from ap_adapter_amd import _lib
    return _lib.lib().apad_rows_packed_kv_bytes(B, H, hd, Lk)


def test_fused_cross_attention_outside_envelope(dev):
    from ap_adapter_amd import ops
    x = torch.zeros(1, 64, 384, device=dev, dtype=torch.bfloat16)
    w = torch.zeros(384, 384, device=dev, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.xattn_pack_weight(w)
    with pytest.raises(ValueError):
        ops.xattn_pack_kv(torch.zeros(1, 8, 384, device=dev, dtype=torch.bfloat16), torch.zeros(1, 8, 48, 32, device=dev, dtype=torch.bfloat16), 8)
    with pytest.raises(ValueError):
        ops.fused_cross_attention(x, w, w, None, w, 8, 8)
    assert ops.xattn_lengths_ok(8, 128) and ops.xattn_lengths_ok(64, 64, True) and ops.xattn_lengths_ok(16, 0, True)
    assert not ops.xattn_lengths_ok(8, 128, True) and not ops.xattn_lengths_ok(8, 96) and not ops.xattn_lengths_ok(16, 128) and not ops.xattn_lengths_ok(8, 576)
    assert ops.xattn_lengths_ok(8, 512) and ops.xattn_lengths_ok(8, 256) and not ops.xattn_lengths_ok(8, 512, True) and not ops.xattn_lengths_ok(8, 200)


# ---- LayerNorm + q|k|v + self-attention in one launch (the two large levels) ----
@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,N,C", [(2, 1000, 256), (9, 1000, 256), (1, 1024, 256), (2, 513, 256), (3, 700, 256), (2, 33, 256), (2, 252, 384), (9, 252, 384),
                                   (1, 256, 384), (3, 130, 384), (2, 200, 384), (2, 970, 256), (2, 900, 256), (1, 897, 256), (2, 896, 256), (2, 450, 256), (3, 385, 256)])
def test_self_attention_fused(dev, dtype, B, N, C):
    """workgroup = (sample, head): K / V^T of the whole sample projected into LDS tiles, Q kept in registers, LayerNorm by algebra on the
    accumulators; full and ragged last panels / key tiles, odd batches (XCD-padded grid); against fp32 torch on storage-rounded operands and
    against the two-launch route (row-panel LN + q|k|v, then apad_attention); rows far from zero mean (the algebra subtracts mean * colsum);
    897 .. 1024 tokens at C = 256 take the projection with LDS-staged token fragments (odd panel counts: the zero padding the staging windows
    overwrote is restored), 896 and fewer the gathered one"""
    from ap_adapter_amd import ops
    H = 8
    x = q(R(B, N, C, seed=461) + 1.5, dtype)
    g, be = q(1 + 0.1 * R(C, seed=462), dtype), q(0.1 * R(C, seed=463), dtype)
    wq, wk, wv = (q(R(C, C, seed=464 + i, std=0.06), dtype) for i in range(3))
    hs = F.layer_norm(x, (C,), g, be, 1e-5)
    sp = lambda t: t.reshape(B, N, H, C // H).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(F.linear(hs, wq)), sp(F.linear(hs, wk)), sp(F.linear(hs, wv))).transpose(1, 2).reshape(B, N, C)
    D = lambda t: t.to(dev, dtype)
    xd, ln = D(x), (D(g), D(be), 1e-5)
    pk, csbb = ops.sattn_pack(D(wq), D(wk), D(wv), ln, H)
    out = ops.self_attention_fused(xd, pk, csbb, H, 1e-5)
    assert out.shape == ref.shape and rel_err(out, ref) < 1.5 * TOL[dtype]
    hn = ops.layer_norm(xd, *ln)
    qd, kd = ops.linear(hn, D(wq)), ops.linear(hn, D(wk))
    vt = torch.zeros(B, H, C // H, ops.round_up(N, 32), device=dev, dtype=dtype)
    ops.linear_vt(hn, D(wv), B, N, H, vt)
    chain = ops.attention(qd, kd, vt, N, H)
    assert rel_err(out, chain.float().cpu()) < TOL[dtype]
    i = B // 2  # a sample's rows do not depend on its batch
    assert torch.equal(ops.self_attention_fused(xd[i:i + 1].contiguous(), pk, csbb, H, 1e-5)[0], out[i])


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("N,C,shift,outlier", [(1000, 256, 8.0, 1.0), (1000, 256, -30.0, 1.0), (1000, 256, 0.0, 25.0), (252, 384, 30.0, 1.0), (252, 384, -8.0, 20.0)])
def test_self_attention_fused_rows_with_a_large_mean(dev, dtype, N, C, shift, outlier):
    """apad_self_attention_fused takes its row statistics as UN-shifted fp32 sums (sum x, sum x^2 on the dot-product instruction) and applies the
    LayerNorm by algebra on the accumulators: rows with |mean| >> std and outlier channels -- where E[x^2] - mean^2 and (W' x) - mean colsum(W')
    cancel -- against fp32 torch on the same stored rows and against the two-launch route (LayerNorm in registers, two-pass statistics)"""
    from ap_adapter_amd import ops
    B, H = 2, 8
    x = R(B, N, C, seed=471) + shift
    x[:, :, 5] *= outlier
    x[:, :, C // 2 + 3] -= 2.0 * outlier
    x = q(x, dtype)
    g, be = q(1 + 0.1 * R(C, seed=472), dtype), q(0.1 * R(C, seed=473), dtype)
    wq, wk, wv = (q(R(C, C, seed=474 + i, std=0.06), dtype) for i in range(3))
    hs = F.layer_norm(x, (C,), g, be, 1e-5)
    sp = lambda t: t.reshape(B, N, H, C // H).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(F.linear(hs, wq)), sp(F.linear(hs, wk)), sp(F.linear(hs, wv))).transpose(1, 2).reshape(B, N, C)
    D = lambda t: t.to(dev, dtype)
    xd, ln = D(x), (D(g), D(be), 1e-5)
    pk, csbb = ops.sattn_pack(D(wq), D(wk), D(wv), ln, H)
    out = ops.self_attention_fused(xd, pk, csbb, H, 1e-5).float().cpu()
    hn = ops.layer_norm(xd, *ln)
    vt = torch.zeros(B, H, C // H, ops.round_up(N, 32), device=dev, dtype=dtype)
    ops.linear_vt(hn, D(wv), B, N, H, vt)
    chain = ops.attention(ops.linear(hn, D(wq)), ops.linear(hn, D(wk)), vt, N, H).float().cpu()
    e_f, e_c = float((out - ref).abs().max() / ref.abs().max()), float((chain - ref).abs().max() / ref.abs().max())
    print(f"N {N} C {C} shift {shift} outlier {outlier}: fused vs fp32 {e_f:.3e}, two-launch route vs fp32 {e_c:.3e}")
    assert torch.isfinite(out).all() and e_f < 3 * TOL[dtype] and e_f < 2.0 * e_c + 1e-3


# ---- the 64-token level's attention sub-layers: apad_hs_attention (head-sliced) + apad_hs_out ----
def _hs_self_ref(x, g, be, wq, wk, wv, wo, bo, heads, residual=True):
    B, N, C = x.shape
    hs = x if g is None else F.layer_norm(x, (C,), g, be, 1e-5)
    sp = lambda t: t.reshape(B, N, heads, C // heads).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(F.linear(hs, wq)), sp(F.linear(hs, wk)), sp(F.linear(hs, wv))).transpose(1, 2).reshape(B, N, C)
    y = F.linear(o, wo, bo)
    return x + y if residual else y


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,N", [(2, 64), (5, 64), (3, 16), (2, 40), (1, 33), (64, 64), (3, 1)])
def test_hs_self_attention_sublayer(dev, dtype, B, N):
    """LayerNorm -> q|k|v -> softmax attention -> to_out + bias + residual of the 64-token level in two launches (workgroup = (sample, head
    pair), then (sample, column quarter)): full and ragged samples (a shorter clip's 16 / 40 tokens, one token), odd batches, the CFG batch;
    against fp32 torch on storage-rounded operands and against the chain it replaces; a sample's rows must not depend on its batch;
    the row statistics hs_out leaves are those of the stored rows"""
    from ap_adapter_amd import ops
    C, H = 640, 8
    x = q(R(B, N, C, seed=401), dtype)
    g, be = q(1 + 0.1 * R(C, seed=402), dtype), q(0.1 * R(C, seed=403), dtype)
    wq, wk, wv = (q(R(C, C, seed=404 + i, std=0.05), dtype) for i in range(3))
    wo, bo = q(R(C, C, seed=407, std=0.05), dtype), q(R(C, seed=408, std=0.3), dtype)
    ref = _hs_self_ref(x, g, be, wq, wk, wv, wo, bo, H)
    D = lambda t: t.to(dev, dtype)
    xd, ln = D(x), (D(g), D(be), 1e-5)
    pk, bb = ops.hs_pack_qkv(D(wq), D(wk), D(wv), ln=ln, q_scale=ops.LOG2E / math.sqrt(C // H))
    wo_p, _ = ops.hs_pack_rows(D(wo))
    o = ops.hs_attention(xd, pk, bb, self_attention=True, ln_eps=1e-5, q_prescaled=True)
    out = ops.hs_out(o, wo_p, D(bo), xd, rowstat=True)
    assert out.shape == ref.shape and rel_err(out, ref) < 1.5 * TOL[dtype]
    rs = ops.rowstat_of(out)
    xs = out.float().cpu().reshape(B * N, C)
    assert tuple(rs.shape) == (B * N, 20, 2)
    assert rel_err(rs[..., 0].sum(1), xs.sum(1)) < 1e-5 and rel_err(rs[..., 1].sum(1), (xs * xs).sum(1)) < 1e-5
    # the chain: LayerNorm -> q | k | v^T -> apad_attention -> to_out + residual
    hs_n = ops.layer_norm(xd, *ln)
    qd, kd = ops.linear(hs_n, D(wq)), ops.linear(hs_n, D(wk))
    vt = torch.zeros(B, H, C // H, ops.round_up(N, 32), device=dev, dtype=dtype)
    ops.linear_vt(hs_n, D(wv), B, N, H, vt)
    chain = ops.linear(ops.attention(qd, kd, vt, N, H), D(wo), D(bo), residual=xd)
    assert rel_err(out, chain.float().cpu()) < TOL[dtype]
    # batch independence: sample i alone == row i of the batch, bit for bit
    i = B // 2
    o1 = ops.hs_out(ops.hs_attention(xd[i:i + 1].contiguous(), pk, bb, self_attention=True, ln_eps=1e-5, q_prescaled=True), wo_p, D(bo), xd[i:i + 1].contiguous())
    assert torch.equal(o1[0], out[i])
    # without LayerNorm / residual / bias (the reference's bare processor call)
    pk0, bb0 = ops.hs_pack_qkv(D(wq), D(wk), D(wv), q_scale=ops.LOG2E / math.sqrt(C // H))
    assert bb0 is None
    bare = ops.hs_out(ops.hs_attention(xd, pk0, None, self_attention=True, normalize=False, q_prescaled=True), wo_p, None, None)
    assert rel_err(bare, _hs_self_ref(x, None, None, wq, wk, wv, wo, None, H, residual=False)) < 1.5 * TOL[dtype]
    # in place over the residual
    x2 = xd.clone()
    ops.hs_out(o, wo_p, D(bo), x2, out=x2)
    assert torch.equal(x2, out)


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,N,Lt,La,masked", [(2, 64, 8, 32, False), (3, 64, 8, 8, False), (2, 64, 16, 0, True), (5, 16, 8, 64, False), (2, 64, 40, 0, True),
                                               (2, 40, 8, 33, False), (64, 64, 8, 32, False), (2, 64, 40, 50, True), (3, 64, 8, 128, False), (2, 64, 32, 100, False),
                                               (3, 64, 8, 256, False), (2, 40, 8, 129, False), (64, 64, 8, 512, False), (2, 64, 32, 200, True)])
def test_hs_cross_attention_sublayer(dev, dtype, B, N, Lt, La, masked):
    """the cross-attention form: q of the head pair projected in the launch, K / V^T the hoisted sets in apad_attention's layout -- the
    adapter's text + scale * audio pair at every pooling rate (8 / 32 / 64 / 128 audio keys in one tile, 129 .. 512 in 64-key chunks), the masked T5 segment (one and two
    key sub-tiles), ragged samples, the CFG batch"""
    from ap_adapter_amd import ops
    C, H = 640, 8
    x = q(R(B, N, C, seed=421), dtype)
    g, be = q(1 + 0.1 * R(C, seed=422), dtype), q(0.1 * R(C, seed=423), dtype)
    wq, wo, bo = q(R(C, C, seed=424, std=0.05), dtype), q(R(C, C, seed=425, std=0.05), dtype), q(R(C, seed=426, std=0.3), dtype)
    wk, wv = q(R(C, 768, seed=427, std=0.04), dtype), q(R(C, 768, seed=428, std=0.04), dtype)
    wki, wvi = q(R(C, 768, seed=429, std=0.04), dtype), q(R(C, 768, seed=430, std=0.04), dtype)
    et = q(R(B, Lt, 768, seed=431), dtype)
    ea = q(R(B, La, 768, seed=432), dtype) if La else None
    bias = None
    if masked:
        bias = torch.zeros(B, Lt)
        bias[1::2, -(Lt // 2):] = -10000.0
    ref = _xattn_ref(x, g, be, wq, wo, bo, et, wk, wv, H, bias, ea, wki, wvi, 0.55)
    D = lambda t: t.to(dev, dtype)
    k1 = ops.linear(D(et), D(wk))
    v1t = torch.zeros(B, H, C // H, ops.round_up(Lt, 32), device=dev, dtype=dtype)
    ops.linear_vt(D(et), D(wv), B, Lt, H, v1t)
    k2 = v2t = None
    if La:
        k2 = ops.linear(D(ea), D(wki))
        v2t = torch.zeros(B, H, C // H, ops.round_up(La, 32), device=dev, dtype=dtype)
        ops.linear_vt(D(ea), D(wvi), B, La, H, v2t)
    xd, ln = D(x), (D(g), D(be), 1e-5)
    bd = None if bias is None else bias.to(dev)
    wq_p, qb = ops.hs_pack_rows(D(wq), ln=ln)
    wo_p, _ = ops.hs_pack_rows(D(wo))
    o = ops.hs_attention(xd, wq_p, qb, self_attention=False, ln_eps=1e-5, k1=k1, vt1=v1t, key_bias=bd, k2=k2, vt2=v2t, scale2=0.55)
    out = ops.hs_out(o, wo_p, D(bo), xd)
    assert out.shape == ref.shape and rel_err(out, ref) < 1.5 * TOL[dtype]
    # the same launch over the fragment-packed key / value sets (what the processors hand it): bit-equal
    p1, p2 = ops.rows_pack_kv(k1, v1t), (ops.rows_pack_kv(k2, v2t) if La else None)
    assert torch.equal(ops.hs_attention(xd, wq_p, qb, self_attention=False, ln_eps=1e-5, k1=p1, key_bias=bd, k2=p2, scale2=0.55), o)
    qd = ops.linear(ops.layer_norm(xd, *ln), D(wq))
    chain = ops.linear(ops.attention(qd, k1, v1t, Lt, H, key_bias=bd, k2=k2, vt2=v2t, L2=La, scale2=0.55), D(wo), D(bo), residual=xd)
    assert rel_err(out, chain.float().cpu()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,N", [(2, 64), (5, 64), (3, 16), (2, 40), (64, 64), (1, 1)])
def test_hs_geglu(dev, dtype, B, N):
    """LayerNorm + GEGLU projection (640 -> 5120 -> 2560) of the 64-token level's feed-forward in one launch (workgroup = (sample, hidden
    quarter), three / two hidden tiles per wave back to back) against fp32 torch on storage-rounded operands and the chain's launch"""
    from ap_adapter_amd import ops
    C = 640
    x = q(R(B, N, C, seed=441) + 0.5, dtype)
    g, be = q(1 + 0.1 * R(C, seed=442), dtype), q(0.1 * R(C, seed=443), dtype)
    w1, b1 = q(R(8 * C, C, seed=444, std=0.05), dtype), q(R(8 * C, seed=445, std=0.2), dtype)
    y = F.linear(F.layer_norm(x, (C,), g, be, 1e-5), w1, b1)
    a, gt = y.chunk(2, dim=-1)
    ref = a * F.gelu(gt)
    D = lambda t: t.to(dev, dtype)
    xd, ln = D(x), (D(g), D(be), 1e-5)
    pk, bb = ops.hs_pack_geglu(D(w1), D(b1), ln=ln)
    h = ops.hs_geglu(xd, pk, bb, ln_eps=1e-5)
    assert h.shape == ref.shape and rel_err(h, ref) < 1.5 * TOL[dtype]
    chain = ops.linear(ops.layer_norm(xd, *ln), D(w1), D(b1), act="geglu")
    assert rel_err(h, chain.float().cpu()) < TOL[dtype]
    i = B // 2  # a sample's rows do not depend on its batch
    assert torch.equal(ops.hs_geglu(xd[i:i + 1].contiguous(), pk, bb, ln_eps=1e-5)[0], h[i])
    # the packing: quarter qq, hidden tile t, value / gate j
    wt = torch.arange(5120 * 640, dtype=torch.float32).reshape(5120, 640).to(dev)
    pw, pb = ops.hs_pack_geglu(wt, torch.arange(5120, dtype=torch.float32).to(dev))
    pw = pw.reshape(4, 20, 2, 40, 2, 32, 8)
    assert torch.equal(pw[3, 17, 1, 5, 1, 30], wt[2560 + 3 * 640 + 17 * 32 + 30, 5 * 16 + 8: 5 * 16 + 16])
    assert float(pb.reshape(4, 20, 2, 32)[2, 9, 1, 7]) == 2560 + 2 * 640 + 9 * 32 + 7


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("B,N", [(2, 64), (5, 64), (3, 16), (2, 40), (64, 64), (1, 1)])
def test_hs_ff2(dev, dtype, B, N):
    """out = x + (H @ W2^T + b2) with H [B, N, 2560] streamed through LDS in eight chunks and the K-quarters of a workgroup summed in a fixed order;
    against fp32 torch and the tiled GEMM; row statistics of the stored rows; batch independence; no residual / bias; in place"""
    from ap_adapter_amd import ops
    C = 640
    h = q(R(B, N, 4 * C, seed=451), dtype)
    x = q(R(B, N, C, seed=452), dtype)
    w2, b2 = q(R(C, 4 * C, seed=453, std=0.03), dtype), q(R(C, seed=454, std=0.3), dtype)
    ref = x + F.linear(h, w2, b2)
    D = lambda t: t.to(dev, dtype)
    hd, xd = D(h), D(x)
    wp = ops.hs_pack_ff2(D(w2))
    out = ops.hs_ff2(hd, wp, D(b2), xd, rowstat=True)
    assert out.shape == ref.shape and rel_err(out, ref) < TOL[dtype]
    rs = ops.rowstat_of(out)
    xs = out.float().cpu().reshape(B * N, C)
    assert tuple(rs.shape) == (B * N, 20, 2)
    assert rel_err(rs[..., 0].sum(1), xs.sum(1)) < 1e-5 and rel_err(rs[..., 1].sum(1), (xs * xs).sum(1)) < 1e-5
    chain = ops.linear(hd, D(w2), D(b2), residual=xd)
    assert rel_err(out, chain.float().cpu()) < TOL[dtype]
    i = B // 2
    assert torch.equal(ops.hs_ff2(hd[i:i + 1].contiguous(), wp, D(b2), xd[i:i + 1].contiguous())[0], out[i])
    bare = ops.hs_ff2(hd, wp, None, None)
    assert rel_err(bare, F.linear(h, w2)) < TOL[dtype]
    x2 = xd.clone()
    ops.hs_ff2(hd, wp, D(b2), x2, out=x2)
    assert torch.equal(x2, out)
    wt = torch.arange(640 * 2560, dtype=torch.float32).reshape(640, 2560).to(dev)
    pk = ops.hs_pack_ff2(wt).reshape(4, 5, 160, 2, 32, 8)
    assert torch.equal(pk[1, 4, 159, 1, 3], wt[160 + 4 * 32 + 3, 159 * 16 + 8: 159 * 16 + 16])


def test_hs_attention_outside_envelope(dev):
    from ap_adapter_amd import ops
    bf = torch.bfloat16
    assert ops.HS_ATTN
    x = torch.zeros(2, 64, 640, device=dev, dtype=bf)
    assert ops.hs_ok(x, 8, 640) and not ops.hs_ok(x, 4, 640) and not ops.hs_ok(x.float(), 8, 640) and not ops.hs_ok(torch.zeros(2, 65, 640, device=dev, dtype=bf), 8, 640)
    assert not ops.hs_ok(torch.zeros(2, 64, 384, device=dev, dtype=bf), 8, 384)
    assert ops.hs_cross_lengths_ok(8, 32) and ops.hs_cross_lengths_ok(64, 64) and ops.hs_cross_lengths_ok(8, 128) and ops.hs_cross_lengths_ok(8, 512) and not ops.hs_cross_lengths_ok(8, 513) and not ops.hs_cross_lengths_ok(40, 128)
    w = torch.zeros(640, 640, device=dev, dtype=bf)
    pk, _ = ops.hs_pack_rows(w)
    with pytest.raises(ValueError):
        ops.hs_attention(torch.zeros(2, 65, 640, device=dev, dtype=bf), pk, None, self_attention=True)
    with pytest.raises(ValueError):  # more than 512 audio keys: the chain's job
        ops.hs_attention(x, pk, None, self_attention=False, k1=torch.zeros(2, 8, 640, device=dev, dtype=bf), vt1=torch.zeros(2, 8, 80, 32, device=dev, dtype=bf),
                         k2=torch.zeros(2, 576, 640, device=dev, dtype=bf), vt2=torch.zeros(2, 8, 80, 576, device=dev, dtype=bf))
    # the packing: quarter p, row tile t, k-step ks = one contiguous KB, lane = (k half, row)
    wt = torch.arange(640 * 640, dtype=torch.float32).reshape(640, 640).to(dev)
    pk = ops.hs_pack_rows(wt)[0].reshape(4, 5, 40, 2, 32, 8)
    assert torch.equal(pk[2, 3, 7, 1, 9], wt[2 * 160 + 3 * 32 + 9, 7 * 16 + 8: 7 * 16 + 16])
    pq = ops.hs_pack_qkv(wt, wt + 1e6, wt + 2e6)[0].reshape(4, 3, 5, 40, 2, 32, 8)
    assert torch.equal(pq[1, 2, 4, 39, 0, 31], wt[160 + 4 * 32 + 31, 39 * 16: 39 * 16 + 8] + 2e6)


# ---- LayerNorm folded into the Linear behind it (the 640-wide level: no row-panel kernel covers K = 640) ----
@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("M,C_,mean_shift", [(2048, 640, 0.0), (300, 640, 3.0), (4096, 640, -8.0), (130, 768, 1.0)])
def test_layernorm_folded_into_gemm(dev, dtype, M, C_, mean_shift):
    """producer: a GEMM (+ bias + residual) that also emits the row statistics of what it stores; consumers: the three Linear
    flavours a BasicTransformerBlock puts behind a LayerNorm (plain to_q, the GEGLU projection, fused q|k|v) with the normalisation
    applied by algebra -- against fp32 LayerNorm + Linear on the stored rows.  mean_shift puts the rows far from zero mean (the
    algebra subtracts mean * colsum from the accumulator)."""
    from ap_adapter_amd import ops
    D = lambda t: t.to(dev, dtype)
    a, w0, b0 = q(R(M, 256, seed=201), dtype), q(R(C_, 256, seed=202, std=0.06), dtype), q(R(C_, seed=203, std=0.1) + mean_shift, dtype)
    res = q(R(M, C_, seed=204), dtype)
    x = ops.linear(D(a), D(w0), D(b0), residual=D(res), rowstat=True)                 # producer
    x_ref = q((F.linear(a, w0, b0)).to(dtype).float() + res, dtype)                     # (linear output rounded before the add)
    assert rel_err(x, x_ref) < TOL[dtype]
    rs = ops.rowstat_of(x)
    assert rs is not None and tuple(rs.shape) == (M, C_ // 64, 2)
    xs = x.float().cpu()                                                               # statistics are those of the STORED rows
    assert rel_err(rs[..., 0].sum(1), xs.sum(1)) < 1e-5 and rel_err(rs[..., 1].sum(1), (xs * xs).sum(1)) < 1e-5
    g, be = q(1 + 0.2 * R(C_, seed=205), dtype), q(0.2 * R(C_, seed=206), dtype)
    ln = (D(g), D(be), 1e-5)
    xn = F.layer_norm(xs, (C_,), g, be, 1e-5)
    # plain Linear (+ bias + residual), folded
    w1, b1 = q(R(C_, C_, seed=207, std=0.05), dtype), q(R(C_, seed=208, std=0.1), dtype)
    out = ops.fused_linear(x, D(w1), D(b1), ln=ln, residual=x)
    assert rel_err(out, (F.linear(xn, w1, b1)).to(dtype).float() + xs) < TOL[dtype]
    unf = ops.linear(ops.layer_norm(x, *ln), D(w1), D(b1), residual=x)                 # the un-folded chain it replaces
    assert rel_err(out, unf.float().cpu()) < 1.5 * TOL[dtype]
    # GEGLU projection, folded
    w2, b2 = q(R(2 * 512, C_, seed=209, std=0.05), dtype), q(R(2 * 512, seed=210, std=0.2), dtype)
    h = ops.fused_linear(x, D(w2), D(b2), ln=ln, act="geglu")
    va, ga = F.linear(xn, w2, b2).chunk(2, dim=-1)
    assert rel_err(h, va * F.gelu(ga)) < TOL[dtype]
    # fused q|k|v, folded (C % 128 == 0 shapes only)
    if C_ % 128 == 0 and M % 4 == 0:
        heads, B = 8, 4
        Lk = M // B
        wq = q(R(3 * C_, C_, seed=211, std=0.05), dtype)
        qo = torch.empty(B, Lk, C_, device=dev, dtype=dtype)
        ko = torch.empty_like(qo)
        vt = torch.zeros(B, heads, C_ // heads, ops.round_up(Lk, 32), device=dev, dtype=dtype)
        ops.linear_qkv(x, D(wq), B, Lk, heads, qo, ko, vt, ln=ln)
        qr, kr, vr = F.linear(xn, wq).chunk(3, dim=-1)
        assert rel_err(qo.view(M, C_), qr) < TOL[dtype] and rel_err(ko.view(M, C_), kr) < TOL[dtype]
        vref = vr.view(B, Lk, heads, C_ // heads).permute(0, 2, 3, 1)
        assert rel_err(vt[..., :Lk], vref) < TOL[dtype]


def test_layernorm_fold_needs_the_producers_statistics(dev):
    """without row statistics on x (a tensor from anywhere else) fused_linear runs apad_layernorm + apad_gemm; a view of a
    statistics-carrying tensor does not inherit them"""
    from ap_adapter_amd import ops
    dtype, M, C_ = torch.bfloat16, 256, 640
    D = lambda t: t.to(dev, dtype)
    x = D(q(R(M, C_, seed=220), dtype))
    w, g, be = D(q(R(C_, C_, seed=221, std=0.05), dtype)), D(q(1 + 0.1 * R(C_, seed=222), dtype)), D(q(0.1 * R(C_, seed=223), dtype))
    assert ops.rowstat_of(x) is None and not ops.ln_foldable(x, w)
    out = ops.fused_linear(x, w, None, ln=(g, be, 1e-5))
    ref = F.linear(F.layer_norm(x.float().cpu(), (C_,), g.float().cpu(), be.float().cpu(), 1e-5), w.float().cpu())
    assert rel_err(out, ref) < TOL[dtype]
    y = ops.linear(x, w, None, rowstat=True)
    assert ops.rowstat_of(y) is not None and ops.rowstat_of(y[:128]) is None and ops.rowstat_of(y.clone()) is None


def test_kernel_form_environment_switches_are_bit_equal(dev, tmp_path):
    """the two C-ABI-level environment switches (read once per process, so flipped in a child process): APAD_CGEMM=0 (the tiled kernel
    instead of the big-tile LDS-DMA kernel) and APAD_GEMM_RING=2 (the LDS-DMA ring form of the 64 x 64 tile) select other kernel FORMS of
    apad_gemm with the same accumulation order -- the results are the same bits"""
    import subprocess
    import sys
    script = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from ap_adapter_amd import ops
torch.manual_seed(0)
dev = torch.device("cuda:0"); dt = torch.bfloat16
x = torch.randn(8, 1000, 256, device=dev).to(dt); w = (torch.randn(256, 9 * 256, device=dev) * 0.02).to(dt); b = torch.randn(256, device=dev).to(dt)
y, _, _ = ops.conv3x3(x.repeat(3, 1, 1), w, b, 24, 125, 8)          # 24 000 output pixels: the big-tile kernel's envelope
a = torch.randn(1000, 640, device=dev).to(dt); w2 = (torch.randn(640, 640, device=dev) * 0.03).to(dt)
z = ops.linear(a, w2, b.repeat(3)[:640])                            # a latency-bound 64 x 64-tile launch: the ring form's envelope
torch.save({"y": y.cpu(), "z": z.cpu()}, sys.argv[2])
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for i, extra in enumerate(({}, {"APAD_CGEMM": "0", "APAD_GEMM_RING": "2"})):
        env = dict(os.environ, **extra)
        for k in ("APAD_CGEMM", "APAD_GEMM_RING"):
            if k not in extra:
                env.pop(k, None)
        path = str(tmp_path / f"o{i}.pt")
        subprocess.run([sys.executable, "-c", script, root, path], check=True, env=env, timeout=600)
        outs.append(torch.load(path))
    assert torch.equal(outs[0]["y"], outs[1]["y"]) and torch.equal(outs[0]["z"], outs[1]["z"])


@pytest.mark.parametrize("dtype", DTYPES16)
@pytest.mark.parametrize("M", [252, 37, 256, 257, 2100, 16128])
@pytest.mark.parametrize("ln,bias", [(True, True), (False, True), (True, False)])
def test_layernorm_geglu_packed(dev, dtype, M, ln, bias):
    """LayerNorm + GEGLU projection at C = 384 on the 64-token register-block kernel from packed weights (apad_geglu_pack +
    apad_layernorm_geglu_packed): against fp32 torch on the storage-rounded operands, and BIT-EQUAL to the row-panel launch smaller batches take
    (b1 as the accumulators' initial value in both, the same MFMA chain, the same GELU arithmetic and rounding); ragged / partial tiles, the
    XCD-padded grid (tiles beyond the last are skipped)"""
    from ap_adapter_amd import ops
    C = 384
    x = q(R(M, C, seed=356), dtype)
    w1, b1 = q(R(8 * C, C, seed=357, std=0.06), dtype), (q(R(8 * C, seed=358, std=0.5), dtype) if bias else None)
    g, be = q(1 + 0.1 * R(C, seed=361), dtype), q(0.1 * R(C, seed=362), dtype)
    xin = q(F.layer_norm(x, (C,), g, be, 1e-5), dtype) if ln else x
    a, gate = F.linear(xin, w1, b1).chunk(2, dim=-1)
    ref = a * F.gelu(gate)
    lnp = (g.to(dev, dtype), be.to(dev, dtype), 1e-5) if ln else None
    dv = lambda t: None if t is None else t.to(dev, dtype)
    wp, bp = ops.geglu_pack(dv(w1), dv(b1))
    out = ops.layernorm_geglu_packed(dv(x), wp, bp, ln=lnp)
    assert out.shape == ref.shape and rel_err(out, ref) < TOL[dtype]
    assert torch.equal(out, ops.fused_linear(dv(x), dv(w1), dv(b1), ln=lnp, act="geglu"))


def test_layernorm_geglu_packed_route_and_repack(dev, monkeypatch):
    """FeedForward at C = 384 takes the packed GEGLU kernel from ops.GEGLU_PACKED_MIN_M rows, packs once, re-packs on an in-place update"""
    from ap_adapter_amd import ops
    from ap_adapter_amd.unet import FeedForward
    from ap_adapter_amd.synthetic import init_synthetic_
    dtype = torch.bfloat16
    ff = FeedForward(384)
    init_synthetic_(ff, 5, w_std=0.05, bias_std=0.05)
    ff = ff.to(dev, dtype).requires_grad_(False)
    ln = (torch.ones(384, device=dev, dtype=dtype), torch.zeros(384, device=dev, dtype=dtype), 1e-5)
    x = torch.randn(600, 384, device=dev).to(dtype)
    calls = []
    real = ops.layernorm_geglu_packed
    monkeypatch.setattr(ops, "layernorm_geglu_packed", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    a = ff(x, ln)
    assert not calls
    monkeypatch.setattr(ops, "GEGLU_PACKED_MIN_M", 512)
    b = ff(x, ln)
    assert len(calls) == 1 and torch.equal(b, a)
    wp0 = ff._geglu3_w[0]
    assert ff(x, ln) is not None and ff._geglu3_w[0] is wp0
    with torch.no_grad():
        ff.net[0].proj.weight.mul_(0.5)
    c = ff(x, ln)
    assert ff._geglu3_w[0] is not wp0 and not torch.equal(b, c)
    monkeypatch.setattr(ops, "MLP_PACKED", False)
    assert torch.equal(c, ff(x, ln))

"""-m gpu: the training step (SURVEY a-11).  Every backward kernel against fp32 torch autograd of the same op on
identical storage-rounded operands; then the adapter gradients, the loss and the optimizer step of a small UNet against
the oracle chain (oracle/train.py = torch autograd through oracle/unet.py).
Tolerance: gradients are rounded to the storage type once per layer, so the kernel-level bound is TOL[dtype] on
max-abs error relative to max|ref|; the end-to-end bound (hundreds of rounded layers) is stated per test."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from util import TOL, q, rel_err

pytestmark = pytest.mark.gpu
DTYPES = [torch.bfloat16, torch.float16, torch.float32]  # float32: the reference's default training precision (train.sh)


def R(*shape, seed=0, std=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * std


def _leaf(t):
    return t.clone().requires_grad_(True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C", [(1000, 256), (77, 384), (5, 640)])
def test_layer_norm_bwd(dev, dtype, M, C):
    from ap_adapter_amd import ops
    x, g, b, dy = q(R(M, C, seed=1), dtype), q(1 + 0.1 * R(C, seed=2), dtype), q(0.1 * R(C, seed=3), dtype), q(R(M, C, seed=4), dtype)
    xl = _leaf(x)
    F.layer_norm(xl, (C,), g, b, 1e-5).backward(dy)
    out = ops.layer_norm_bwd(x.to(dev, dtype), g.to(dev, dtype), dy.to(dev, dtype), 1e-5)
    assert rel_err(out, xl.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,HW,C,G", [(2, 416, 128, 32), (3, 104, 192, 16), (1, 63 * 4, 1280, 32)])
@pytest.mark.parametrize("silu", [False, True])
def test_group_norm_bwd(dev, dtype, B, HW, C, G, silu):
    from ap_adapter_amd import ops
    x, g, b = q(R(B, HW, C, seed=5), dtype), q(1 + 0.1 * R(C, seed=6), dtype), q(0.1 * R(C, seed=7), dtype)
    dy = q(R(B, HW, C, seed=8), dtype)
    xl = _leaf(x)
    y = F.group_norm(xl.transpose(1, 2), G, g, b, 1e-5)
    if silu:
        y = F.silu(y)
    y.transpose(1, 2).backward(dy)
    out = ops.group_norm_bwd(x.to(dev, dtype), g.to(dev, dtype), b.to(dev, dtype), dy.to(dev, dtype), G, 1e-5, silu)
    assert rel_err(out, xl.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_geglu_fwd_bwd(dev, dtype):
    from ap_adapter_amd import ops
    M, N = 333, 512
    proj, dh = q(R(M, 2 * N, seed=9), dtype), q(R(M, N, seed=10), dtype)
    pl = _leaf(proj)
    a, g = pl.chunk(2, dim=-1)
    h = a * F.gelu(g)
    h.backward(dh)
    assert rel_err(ops.geglu(proj.to(dev, dtype)), h.detach()) < TOL[dtype]
    assert rel_err(ops.geglu_bwd(proj.to(dev, dtype), dh.to(dev, dtype)), pl.grad) < TOL[dtype]


def _sdpa(qq, k, v, heads, bias=None):
    B, N, C = qq.shape
    sp = lambda t: t.reshape(B, t.shape[1], heads, C // heads).transpose(1, 2)
    m = None if bias is None else bias[:, None, None, :]
    o = F.scaled_dot_product_attention(sp(qq), sp(k), sp(v), attn_mask=m)
    return o.transpose(1, 2).reshape(B, N, C)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,L,heads,d,bias", [(2, 100, 100, 4, 32, False), (1, 252, 252, 8, 48, False), (2, 64, 64, 8, 80, False),
                                                (2, 130, 16, 4, 16, True), (1, 1000, 40, 8, 32, False), (2, 33, 7, 2, 64, True)])
def test_attention_bwd(dev, dtype, B, N, L, heads, d, bias):
    """dq, dk, dv of one softmax segment (ragged tiles, every head dim of the UNet, additive key bias)"""
    from ap_adapter_amd import autograd as AG
    C = heads * d
    qq, k, v = q(R(B, N, C, seed=11), dtype), q(R(B, L, C, seed=12), dtype), q(R(B, L, C, seed=13), dtype)
    do = q(R(B, N, C, seed=14), dtype)
    kb = None
    if bias:
        kb = torch.zeros(B, L)
        kb[:, -3:] = -10000.0
    ql, kl, vl = _leaf(qq), _leaf(k), _leaf(v)
    ref = _sdpa(ql, kl, vl, heads, kb)
    ref.backward(do)
    qd, kd, vd = (t.to(dev, dtype).requires_grad_(True) for t in (qq, k, v))
    out = AG.attention(qd, kd, vd, heads, None if kb is None else kb.to(dev))
    out.backward(do.to(dev, dtype))
    assert rel_err(out, ref.detach()) < TOL[dtype]
    for got, want, name in ((qd.grad, ql.grad, "dq"), (kd.grad, kl.grad, "dk"), (vd.grad, vl.grad, "dv")):
        assert rel_err(got, want) < 1.5 * TOL[dtype], name


@pytest.mark.parametrize("dtype", DTYPES)
def test_ip_attention_bwd(dev, dtype):
    """decoupled cross-attention (attention_processor.py:429-454): gradients of q and of the audio K/V; text K/V frozen"""
    from ap_adapter_amd import autograd as AG
    B, N, heads, d, Lt, La, s = 2, 252, 8, 48, 8, 32, 0.55
    C = heads * d
    qq, kt, vt = q(R(B, N, C, seed=21), dtype), q(R(B, Lt, C, seed=22), dtype), q(R(B, Lt, C, seed=23), dtype)
    ka, va, do = q(R(B, La, C, seed=24), dtype), q(R(B, La, C, seed=25), dtype), q(R(B, N, C, seed=26), dtype)
    ql, kal, val = _leaf(qq), _leaf(ka), _leaf(va)
    ref = _sdpa(ql, kt, vt, heads) + s * _sdpa(ql, kal, val, heads)
    ref.backward(do)
    qd, kad, vad = (t.to(dev, dtype).requires_grad_(True) for t in (qq, ka, va))
    out = AG.ip_attention(qd, kt.to(dev, dtype), vt.to(dev, dtype), kad, vad, heads, None, s)
    out.backward(do.to(dev, dtype))
    assert rel_err(out, ref.detach()) < TOL[dtype]
    for got, want, name in ((qd.grad, ql.grad, "dq"), (kad.grad, kal.grad, "dk_ip"), (vad.grad, val.grad, "dv_ip")):
        assert rel_err(got, want) < 1.5 * TOL[dtype], name


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_bwd_and_weight_grad(dev, dtype):
    """dx through a frozen weight, dW of a trainable one (to_k_ip: [C, 768], reduction over B*La token rows)"""
    from ap_adapter_amd import autograd as AG
    M, K, N = 4 * 32, 768, 384
    x, w, b, dy = q(R(M, K, seed=31), dtype), q(R(N, K, seed=32, std=0.05), dtype), q(R(N, seed=33), dtype), q(R(M, N, seed=34), dtype)
    xl, wl = _leaf(x), _leaf(w)
    F.linear(xl, wl, b).backward(dy)
    xd, wd = x.to(dev, dtype).requires_grad_(True), w.to(dev, dtype).requires_grad_(True)
    AG.linear(xd, wd, b.to(dev, dtype)).backward(dy.to(dev, dtype))
    assert rel_err(xd.grad, xl.grad) < TOL[dtype]
    assert rel_err(wd.grad, wl.grad) < TOL[dtype]
    # residual branch receives dy unchanged
    r = torch.zeros(M, N).to(dev, dtype).requires_grad_(True)
    AG.linear(x.to(dev, dtype), w.to(dev, dtype), None, residual=r).backward(dy.to(dev, dtype))
    assert torch.equal(r.grad.cpu().float(), dy)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["s1", "s2", "s2_odd", "up2", "up_size"])
def test_conv3x3_bwd(dev, dtype, case):
    from ap_adapter_amd import autograd as AG
    B, Cin, Cout = 2, 64, 96
    H, W, stride, up = {"s1": (26, 16, 1, None), "s2": (26, 16, 2, None), "s2_odd": (25, 7, 2, None), "up2": (13, 8, 1, (26, 16)),
                        "up_size": (13, 4, 1, (25, 8))}[case]
    x, w, b = q(R(B, Cin, H, W, seed=41), dtype), q(R(Cout, Cin, 3, 3, seed=42, std=0.05), dtype), q(R(Cout, seed=43), dtype)
    xl = _leaf(x)
    src = xl if up is None else F.interpolate(xl, size=up, mode="nearest")
    ref = F.conv2d(src, w, b, stride=stride, padding=1)
    dy = q(R(*ref.shape, seed=44), dtype)
    ref.backward(dy)
    xd = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).to(dev, dtype).requires_grad_(True)
    out, Ho, Wo = AG.conv3x3(xd, w.to(dev, dtype), b.to(dev, dtype), B, H, W, stride=stride, up=up)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    out.backward(dy.permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout).to(dev, dtype))
    want = xl.grad.permute(0, 2, 3, 1).reshape(B, H * W, Cin)
    assert rel_err(out, ref.detach().permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout)) < TOL[dtype]
    assert rel_err(xd.grad, want) < TOL[dtype]


def test_mse_loss_grad(dev):
    from ap_adapter_amd import autograd as AG
    pred = q(R(4, 8, 250, 16, seed=51), torch.bfloat16)
    tgt = R(4, 8, 250, 16, seed=52)
    pl = _leaf(pred)
    ref = F.mse_loss(pl.float(), tgt.float(), reduction="mean")
    ref.backward()
    pd = pred.to(dev, torch.bfloat16).requires_grad_(True)
    loss = AG.mse_loss(pd, tgt.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * float(ref.detach())
    assert rel_err(pd.grad, pl.grad) < TOL[torch.bfloat16]


def test_mse_loss_scaling_keeps_f16_gradients(dev):
    """at full geometry d loss / d pred = 2 (p - t) / (B*8*250*16) is ~1e-5: subnormal in f16.  The static loss
    scale is applied in fp32 inside the kernel, before the rounding; unscaled f16 loses most of the gradient's precision"""
    from ap_adapter_amd import autograd as AG
    pred = q(R(4, 8, 250, 16, seed=51), torch.float16)
    tgt = R(4, 8, 250, 16, seed=52)
    exact = 2.0 * (pred - tgt) / pred.numel()
    pd = pred.to(dev, torch.float16).requires_grad_(True)
    AG.mse_loss(pd, tgt.to(dev), 65536.0).backward()
    scaled = pd.grad.float().cpu() / 65536.0
    assert rel_err(scaled, exact) < TOL[torch.float16]
    pd2 = pred.to(dev, torch.float16).requires_grad_(True)
    AG.mse_loss(pd2, tgt.to(dev)).backward()
    # what the scale is for: every gradient here is an f16 subnormal (spacing 6e-8, i.e. ~0.5 % of a 1e-5 value); max-normalised error
    # hides that, the mean element-wise relative error shows it
    sel = exact.abs() > 1e-6
    mean_rel = lambda g: float(((g.float().cpu() - exact).abs() / exact.abs())[sel].mean())
    assert mean_rel(pd2.grad) > 5 * mean_rel(scaled)


def test_trainer_f16_loss_scale_and_overflow_skip(dev):
    """AdapterTrainer in f16: gradients reach the fp32 accumulator unscaled; a step whose gradient norm is not finite (forced
    here by an absurd scale) changes neither the weights, nor the moments, nor the bias-correction step index"""
    import ap_adapter_amd as A
    u, cfg, sd, procs = _small_unet(dev, torch.float16)
    tr = A.AdapterTrainer(u, lr=1e-3)
    assert tr.loss_scale == 65536.0
    lat, noise, t, ehs, ehs1, m1 = _batch(2, 8, torch.float16)
    D = lambda x: x.to(dev)
    tr.micro_step(D(lat), D(t), D(ehs), D(ehs1), D(m1), D(noise))
    g1 = tr.grad.clone()
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    u2, *_ = _small_unet(dev, torch.float16)
    tr2 = A.AdapterTrainer(u2, lr=1e-3, loss_scale=1.0)  # the same gradient without scaling, up to f16 rounding of the chain
    tr2.micro_step(D(lat), D(t), D(ehs), D(ehs1), D(m1), D(noise))
    assert 1 - float(F.cosine_similarity(g1.double(), tr2.grad.double(), dim=0)) < 1e-2
    tr.optimizer_step()
    assert int(tr.step_t.item()) == 1
    before = tr.master.clone()
    tr.grad.fill_(float("inf"))
    tr._micro = 1
    tr.optimizer_step()
    assert torch.equal(tr.master, before) and int(tr.step_t.item()) == 1 and torch.isfinite(tr.exp_avg).all()


def test_trainer_lr_schedule_and_backing_off_loss_scale(dev):
    """the reference's --lr_scheduler / --lr_warmup_steps (train_apadapter_v2.py:809-815) and the GradScaler behind accelerate's fp16 mode
    (:958): warm-up starts at lr 0 (the first update changes nothing), the scale halves behind an overflowed step WITHOUT a host sync in the
    step (read one step late from pinned memory), a captured micro-step re-captures with the new scale and still accumulates the same
    unscaled gradient"""
    import ap_adapter_amd as A
    u, cfg, sd, procs = _small_unet(dev, torch.float16)
    tr = A.AdapterTrainer(u, lr=1e-3, lr_scheduler="constant_with_warmup", lr_warmup_steps=2, gradient_accumulation_steps=1)
    assert tr.dynamic_loss_scale and tr.loss_scale == 65536.0 and tr.lr == 0.0
    lat, noise, t, ehs, ehs1, m1 = _batch(2, 8, torch.float16)
    args = tuple(x.to(dev) for x in (lat, t, ehs, ehs1, m1, noise))
    replay = tr.capture_micro_step(2, lat.shape[2], lat.shape[3], ehs.shape[1], ehs1.shape[1])
    def step():  # (the flag copy behind a step has landed by the next one in a real run; make that certain here)
        tr.optimizer_step()
        torch.cuda.synchronize()

    replay(*args)
    g_ref = tr.grad.clone()
    before = tr.master.clone()
    step()                                                # lr 0: AdamW (decay included) moves nothing
    assert torch.equal(tr.master, before) and tr.lr == 5e-4
    replay(*args)
    step()
    assert not torch.equal(tr.master, before) and tr.lr == 1e-3
    tr.grad.fill_(float("inf"))                           # an overflowed step ...
    tr._micro = 1
    step()
    assert tr.loss_scale == 65536.0                       # ... is noticed behind the NEXT boundary
    replay(*args)
    step()
    assert tr.loss_scale == 32768.0 and tr.skipped_steps == 1 and tr.global_step == 4
    tr.grad.zero_()
    tr._micro = 0
    replay(*args)                                         # re-captured with the halved scale ...
    g_new = tr.grad.clone()
    tr.grad.zero_()
    tr.micro_step(*args)                                  # ... = the eager micro-step at the same weights and scale, bit for bit
    assert torch.equal(tr.grad, g_new) and torch.isfinite(g_new).all() and float(g_new.abs().max()) > 0
    assert 0.3 < float(g_new.norm() / g_ref.norm()) < 3   # (unscaled: the same magnitude as under the old scale)
    tr.grad.zero_()
    tr._micro = 0
    sd_ = tr.state_dict()
    # (the schedule does not advance on the skipped update -- accelerate's prepared scheduler returns early while
    #  optimizer.step_was_skipped --: 4 boundaries, 1 of them skipped)
    assert sd_["loss_scale"] == 32768.0 and sd_["scheduler_step"] == 3
    # growth: scale_growth_interval clean boundaries double it
    u2, *_ = _small_unet(dev, torch.float16)
    tr2 = A.AdapterTrainer(u2, lr=1e-4, scale_growth_interval=2)
    for _ in range(4):
        tr2.micro_step(*args)
        tr2.optimizer_step()
        torch.cuda.synchronize()
    assert tr2.loss_scale == 131072.0 and tr2.skipped_steps == 0
    # bf16 runs unscaled and static
    u3, *_ = _small_unet(dev, torch.bfloat16)
    assert not A.AdapterTrainer(u3).dynamic_loss_scale


def test_grad_norm_and_adamw_match_the_oracle(dev):
    """clip coefficient + AdamW over a flat buffer, 3 steps, against oracle/train.py (itself pinned to torch.optim.AdamW)"""
    from ap_adapter_amd import ops
    from oracle import train as OT
    n = 100003
    p0, lr, wd, mx = R(n, seed=61, std=0.05), 1e-3, 1e-2, 1.0
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    P, M_, V = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    work = torch.empty(n, dtype=torch.bfloat16, device=dev)
    step_t = torch.zeros(1, dtype=torch.int32, device=dev)
    for step in (1, 2, 3):
        g = R(n, seed=70 + step, std=0.02 * step)
        coef, total = OT.clip_coef([g], mx)
        p, m, v = OT.adamw_update(p, g * coef, m, v, step, lr, weight_decay=wd)
        G = g.to(dev)
        ops.step_advance(step_t)
        gn = ops.grad_norm(G)
        assert abs(float(gn) - float(total)) < 1e-5 * float(total)
        ops.adamw_step(P, work, G, M_, V, gn, step_t, lr, (0.9, 0.999), 1e-8, wd, mx)
        assert rel_err(P, p) < 1e-6
        assert torch.equal(work.cpu(), P.cpu().to(torch.bfloat16))
    assert rel_err(M_, m) < 1e-5 and rel_err(V, v) < 1e-4  # v carries the squared fp32 clip coefficient


def _small_unet(dev, dtype, seed=100):
    import ap_adapter_amd as A
    from ap_adapter_amd.synthetic import init_synthetic_
    cfg = A.UNetConfig(block_out_channels=(64, 128, 192, 256), attention_head_dim=4, norm_num_groups=16)
    u = A.AudioLDM2UNet2DConditionModel(cfg)
    A.install_ap_adapter(u, None, scale=0.5)
    init_synthetic_(u, seed, w_std=0.05, bias_std=0.02, norm_jitter=0.1)
    u = u.to(dtype)
    sd = {k: v.detach().float().cpu() for k, v in u.state_dict().items()}
    procs = {n: dict(scale=p.scale, num_tokens=p.num_tokens) for n, p in u.attn_processors.items() if hasattr(p, "to_k_ip")}
    return u.to(dev), cfg, sd, procs


def _batch(B, La, dtype):
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(B, 8, 26, 16, generator=g)
    noise = torch.randn(B, 8, 26, 16, generator=g)
    t = torch.tensor([17, 503, 998, 250][:B])
    ehs = torch.randn(B, 8 + La, 768, generator=g).to(dtype).float()
    ehs1 = torch.randn(B, 16, 1024, generator=g).to(dtype).float()
    m1 = torch.ones(B, 16)
    m1[1::2, -4:] = 0
    return lat, noise, t, ehs, ehs1, m1


# (bf16: a rounding-noise bound through ~200 rounded layers; its realisation moves with the kernels' summation orders -- worst tensor
#  0.078 with three input-gradient GEMMs for q / k / v, 0.084 with the single stacked one; the exactness claim is the fp32 row)
# per-tensor relative L2 bound on top of the max-abs one (round 6; measured 5.1e-2 / 6.2e-3 / 8.3e-6): the max-abs bound of a 16-bit mode is set by
# the single worst entry of a rounding-noise field, the L2 norm is what the optimizer step feels
L2TOL = {torch.bfloat16: 7e-2, torch.float16: 9e-3, torch.float32: 3e-5}


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-1), (torch.float16, 2e-2), (torch.float32, 2e-4)])
def test_adapter_gradients_small_unet_vs_oracle(dev, dtype, tol):
    """loss and d loss / d to_{k,v}_ip.weight of all 32 adapted sites after a backward through the whole frozen UNet
    (per-sample timesteps, masked T5 stream), against torch autograd through the fp32 oracle.  Bound: max-abs error
    relative to the largest gradient entry of each tensor, `tol`; the flat gradient's direction within 1 - cos < 2e-3."""
    import ap_adapter_amd as A
    from oracle import train as OT
    u, cfg, sd, procs = _small_unet(dev, dtype)
    B = 3
    lat, noise, t, ehs, ehs1, m1 = _batch(B, 32, dtype)
    noisy = q(OT.add_noise(lat, noise, t), dtype)
    ref_loss, ref_grads, _ = OT.loss_and_grads(sd, cfg.geometry_dict(), procs, noisy, t, ehs, ehs1, m1, noise)
    tr = A.AdapterTrainer(u, lr=1e-3)
    loss = tr.micro_step(noisy.to(dev), t.to(dev), ehs.to(dev), ehs1.to(dev), m1.to(dev), noise.to(dev))
    assert abs(float(loss) - float(ref_loss)) < (1e-5 if dtype == torch.float32 else 2e-2) * float(ref_loss)
    names = [n for n, p in u.attn_processors.items() if hasattr(p, "to_k_ip")]
    assert len(names) == 32 and len(tr.params) == 64
    flat_ref = []
    worst_l2 = 0.0
    for i, n in enumerate(names):
        for j, which in enumerate(("to_k_ip", "to_v_ip")):
            p = tr.params[2 * i + j]
            got = tr.grad[tr.offsets[2 * i + j]: tr.offsets[2 * i + j] + p.numel()].view(p.shape)
            want = ref_grads[f"{n}.{which}.weight"]
            assert float(want.abs().max()) > 0
            assert rel_err(got, want) < tol, (n, which)
            l2 = float((got.float().cpu() - want).norm() / want.norm())
            worst_l2 = max(worst_l2, l2)
            flat_ref.append(want.reshape(-1))
    print(f"worst per-tensor relative L2 gradient error ({dtype}): {worst_l2:.3e}")
    assert worst_l2 < L2TOL[dtype]
    flat_ref = torch.cat(flat_ref)
    cos = F.cosine_similarity(tr.grad.cpu().double(), flat_ref.double(), dim=0)
    assert 1 - float(cos) < (1e-8 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_trainer_steps_track_the_oracle(dev, dtype):
    """two optimizer steps with gradient accumulation 2: parameters after each step against the oracle's
    loss_and_grads -> mean over micro-batches -> clip -> AdamW (fp32 = the reference's default precision: updates within 1e-3)"""
    import ap_adapter_amd as A
    from oracle import train as OT
    u, cfg, sd, procs = _small_unet(dev, dtype)
    lr, wd = 1e-2, 1e-2
    tr = A.AdapterTrainer(u, lr=lr, weight_decay=wd, gradient_accumulation_steps=2)
    keys = OT.adapter_keys(sd)
    names = [n for n, p in u.attn_processors.items() if hasattr(p, "to_k_ip")]
    order = [f"{n}.{w}.weight" for n in names for w in ("to_k_ip", "to_v_ip")]
    assert sorted(order) == keys
    m = {k: torch.zeros_like(sd[k]) for k in keys}
    v = {k: torch.zeros_like(sd[k]) for k in keys}
    before = tr.master.clone()
    for step in (1, 2):
        acc = {k: torch.zeros_like(sd[k]) for k in keys}
        for micro in range(2):
            lat, noise, t, ehs, ehs1, m1 = _batch(2, 8, dtype)
            lat = lat + step + micro  # different data per micro-batch
            noisy = q(OT.add_noise(lat, noise, t), dtype)
            # the oracle forward sees the storage-rounded working weights, like the kernels
            sdq = {k: (q(val, dtype) if k in keys else val) for k, val in sd.items()}
            _, g, _ = OT.loss_and_grads(sdq, cfg.geometry_dict(), procs, noisy, t, ehs, ehs1, m1, noise)
            for k in keys:
                acc[k] += g[k] / 2
            tr.micro_step(noisy.to(dev), t.to(dev), ehs.to(dev), ehs1.to(dev), m1.to(dev), noise.to(dev))
        tr.optimizer_step()
        coef, _ = OT.clip_coef([acc[k] for k in keys], 1.0)
        for k in keys:
            sd[k], m[k], v[k] = OT.adamw_update(sd[k], acc[k] * coef, m[k], v[k], step, lr, weight_decay=wd)
        want = torch.cat([sd[k].reshape(-1) for k in order])
        # AdamW moves every weight by ~lr per step whatever the gradient scale: compare the UPDATE, not the weight
        upd_ref, upd = want - before.cpu(), tr.master.cpu() - before.cpu()
        assert float((upd - upd_ref).abs().mean() / upd_ref.abs().mean()) < (1e-3 if dtype == torch.float32 else 0.15)
        assert 1 - float(F.cosine_similarity(upd, upd_ref, dim=0)) < (1e-6 if dtype == torch.float32 else 2e-2)
    assert tr.global_step == 2 and int(tr.step_t.item()) == 2
    # the nn.Parameters the forward reads are views of the updated working copy
    p0 = tr.params[0]
    assert torch.equal(p0.detach().reshape(-1), tr.work[: p0.numel()])
    assert torch.equal(tr.work.float().cpu(), tr.master.cpu().to(dtype).float())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-2)])
def test_accumulation_as_batch_equals_sequential_micro_batches(dev, dtype, tol):
    """gradient_accumulation_steps = 2 run the reference's way (two micro-batches of 2, :892 `with accelerator.accumulate`) against ONE
    pass over their concatenation with micro_batches=2: same accumulated gradient (fp32: to rounding of the summation order; bf16: to
    the storage rounding of two different kernel forms), same mean loss, same optimizer step, same counters."""
    import ap_adapter_amd as A
    u1, _, _, _ = _small_unet(dev, dtype)
    u2, _, _, _ = _small_unet(dev, dtype)
    lat, noise, t, ehs, ehs1, m1 = (x.to(dev) for x in _batch(4, 8, dtype))
    tr1 = A.AdapterTrainer(u1, lr=1e-2, gradient_accumulation_steps=2)
    tr2 = A.AdapterTrainer(u2, lr=1e-2, gradient_accumulation_steps=2)
    noisy = A.add_noise(lat, noise, t, tr1.alphas_cumprod)
    l1 = [tr1.micro_step(noisy[i:i + 2], t[i:i + 2], ehs[i:i + 2], ehs1[i:i + 2], m1[i:i + 2], noise[i:i + 2]) for i in (0, 2)]
    l2 = tr2.micro_step(noisy, t, ehs, ehs1, m1, noise, micro_batches=2)
    assert tr1._micro == tr2._micro == 2
    assert abs(float(l2) - 0.5 * float(l1[0] + l1[1])) < (1e-6 if dtype == torch.float32 else 1e-2) * float(l2)
    assert float(tr1.grad.abs().max()) > 0
    assert rel_err(tr2.grad, tr1.grad) < tol
    assert 1 - float(F.cosine_similarity(tr1.grad.double(), tr2.grad.double(), dim=0)) < (1e-9 if dtype == torch.float32 else 2e-3)
    tr1.optimizer_step()
    tr2.optimizer_step()
    assert tr1.global_step == tr2.global_step == 1 and tr1._micro == tr2._micro == 0
    if dtype == torch.float32:
        assert float((tr1.master - tr2.master).abs().mean()) < 1e-3 * 1e-2  # (the first AdamW step moves every weight by ~lr: compare on that scale)
    with pytest.raises(ValueError):
        tr2.micro_step(noisy[:3], t[:3], ehs[:3], ehs1[:3], m1[:3], noise[:3], micro_batches=2)
    # the captured form counts the same way
    replay = tr2.capture_micro_step(4, 26, 16, 16, 16, micro_batches=2)
    replay(noisy, t, ehs, ehs1, m1, noise)
    assert tr2._micro == 2


def test_collate_and_checkpoint(dev, tmp_path):
    """f-3: per-batch pooling choice + condition dropout in front of AudioMAE (text tokens first, :471), and a checkpoint
    that carries the fp32 master weights under the reference key scheme plus the optimizer state"""
    import random
    import ap_adapter_amd as A
    from ap_adapter_amd import training as T
    from ap_adapter_amd.synthetic import init_synthetic_
    dtype = torch.float16
    mae = A.AudioMAEConditionCTPoolRand(depth=1)
    init_synthetic_(mae, 7, w_std=0.03, bias_std=0.02, norm_jitter=0.1)
    mae = mae.to(dev, dtype)
    enc = lambda texts: (torch.zeros(len(texts), 16, 1024, device=dev, dtype=dtype), torch.ones(len(texts), 16, device=dev),
                         torch.full((len(texts), 8, 768), 2.0, device=dev, dtype=dtype))
    col = T.CollateFunction(mae, enc, rng=random.Random(3), device=dev)
    ex = [{"text": f"a recording of a {i}", "fbank": torch.randn(1024, 128, device=dev) * 0.5} for i in range(3)]
    b = col(ex)
    La = (64 // b["pooling_rate"]) * (8 // b["pooling_rate"])
    assert b["pooling_rate"] in T.POOL_LIST and b["generated_prompt_embeds"].shape == (3, 8 + La, 768)
    assert bool((b["generated_prompt_embeds"][:, :8] == 2.0).all())

    u, cfg, sd, procs = _small_unet(dev, torch.bfloat16)
    tr = A.AdapterTrainer(u, lr=1e-3)
    tr.master.add_(1e-4)                                   # something the bf16 working copy cannot represent
    tr.global_step = 7
    path = T.save_checkpoint(tr, str(tmp_path), checkpoints_total_limit=2)
    assert os.path.basename(path) == "checkpoint-7"
    ck = A.load_adapter(os.path.join(path, "pytorch_model.bin"))
    names = [n for n, p in u.attn_processors.items() if hasattr(p, "to_k_ip")]
    assert sorted(ck) == sorted(f"{n}.{w}.weight" for n in names for w in ("to_k_ip", "to_v_ip"))
    k0 = f"{names[0]}.to_k_ip.weight"
    assert ck[k0].dtype == torch.float32 and torch.equal(ck[k0].reshape(-1), tr.master[: ck[k0].numel()].cpu())
    st = torch.load(os.path.join(path, "optimizer.bin"), weights_only=True)
    tr2 = A.AdapterTrainer(_small_unet(dev, torch.bfloat16, seed=5)[0])
    tr2.load_state_dict(st)
    assert torch.equal(tr2.master, tr.master) and tr2.global_step == 7


def test_graph_captured_micro_step_equals_eager(dev):
    """the whole forward + loss + backward of a micro-batch replayed as one hipGraph: same loss and same accumulated
    gradient, bit for bit, as the eager tape on the same data"""
    import ap_adapter_amd as A
    from oracle import train as OT
    dtype = torch.bfloat16
    u, cfg, sd, procs = _small_unet(dev, dtype)
    tr = A.AdapterTrainer(u, lr=1e-3)
    lat, noise, t, ehs, ehs1, m1 = _batch(2, 8, dtype)
    noisy = OT.add_noise(lat, noise, t)
    args = (noisy.to(dev), t.to(dev), ehs.to(dev, dtype), ehs1.to(dev, dtype), m1.to(dev), noise.to(dev))
    loss_e = float(tr.micro_step(*args))
    grad_e = tr.grad.clone()
    tr.grad.zero_(); tr._micro = 0
    replay = tr.capture_micro_step(2, 26, 16, ehs.shape[1], ehs1.shape[1])
    assert float(tr.grad.abs().max()) == 0 and tr._micro == 0          # capturing leaves the trainer untouched
    loss_g = float(replay(*args))
    assert loss_g == loss_e and torch.equal(tr.grad, grad_e) and tr._micro == 1
    # new data through the same graph
    args2 = ((noisy + 0.5).to(dev),) + args[1:]
    tr.grad.zero_()
    l2 = float(replay(*args2))
    g2 = tr.grad.clone()
    tr.grad.zero_()
    assert float(tr.micro_step(*args2)) == l2 and torch.equal(tr.grad, g2)


@pytest.mark.parametrize("dtype,frames,B", [(torch.bfloat16, 250, 1), (torch.float32, 64, 1), (torch.bfloat16, 64, 4)])
def test_full_geometry_adapter_gradients_vs_oracle(dev, dtype, frames, B):
    """BASELINE config 5 geometry (AudioLDM2-large, 718 M frozen parameters, 64 trainable tensors = 21 626 880 elements),
    one 10 s sample at a random t: every adapter gradient after a backward through the whole UNet vs torch autograd
    through the fp32 oracle.  Bound (bf16): the flat gradient's direction within 1 - cos < 5e-3 and its norm within 3 %;
    per tensor, max-abs error below 0.15 of that tensor's largest entry (the smallest gradients sit deep in the stack).
    fp32 (the reference's default training precision, train.sh): 1 - cos < 1e-7, norm within 1e-4, every tensor within 1e-3 of its max
    (on a 2.56 s sample, ``frames`` = 64: same network, a quarter of the oracle's autograd work; the 10 s fp32 run measured cos 1.000000,
    worst tensor 2.1e-5).  B = 4: the cfg-5 micro-batch -- four samples at FOUR DIFFERENT timesteps (train_apadapter_v2.py:900-905 draws one
    per sample), 2.56 s each: the per-sample time projection and the batch-mean loss at full geometry."""
    import ap_adapter_amd as A
    from ap_adapter_amd.synthetic import init_synthetic_, synthetic_inputs
    from oracle import train as OT
    from util import full_unet
    u = full_unet(0.55)
    u = u.to(dtype)
    sd = {k: v.detach().float() for k, v in u.state_dict().items()}
    procs = {n: dict(scale=p.scale, num_tokens=p.num_tokens) for n, p in u.attn_processors.items() if hasattr(p, "to_k_ip")}
    inp = synthetic_inputs(B, 32)
    pipe = A.AudioLDM2Pipeline(u)
    ehs = pipe.assemble_condition(inp["generated_prompt_embeds"], inp["audio_tokens"], inp["uncond_audio_tokens"], dtype)[B:]
    ehs1 = inp["prompt_embeds"].to(dtype)[B:]
    m1 = inp["attention_mask"].float()[B:]
    g = torch.Generator().manual_seed(9)
    noise = torch.randn(B, 8, 250, 16, generator=g)[:, :, :frames].contiguous()
    t = torch.tensor([437, 12, 880, 651][:B])
    noisy = q(OT.add_noise(inp["latents"].float()[:, :, :frames].contiguous(), noise, t), dtype)
    ref_loss, ref_grads, _ = OT.loss_and_grads(sd, u.config.geometry_dict(), procs, noisy, t, ehs.float(), ehs1.float(), m1, noise)
    u = u.to(dev)
    tr = A.AdapterTrainer(u)
    loss = tr.micro_step(noisy.to(dev), t.to(dev), ehs.to(dev), ehs1.to(dev), m1.to(dev), noise.to(dev))
    assert abs(float(loss) - float(ref_loss)) < (1e-5 if dtype == torch.float32 else 2e-2) * float(ref_loss)
    names = [n for n, p in u.attn_processors.items() if hasattr(p, "to_k_ip")]
    flat_ref = torch.cat([ref_grads[f"{n}.{w}.weight"].reshape(-1) for n in names for w in ("to_k_ip", "to_v_ip")])
    got = tr.grad.cpu()
    assert got.numel() == flat_ref.numel() == 21626880
    cos = float(F.cosine_similarity(got.double(), flat_ref.double(), dim=0))
    nr = float(got.double().norm() / flat_ref.double().norm())
    worst = 0.0
    for i, (p, off) in enumerate(zip(tr.params, tr.offsets)):
        a, b = got[off:off + p.numel()], flat_ref[off:off + p.numel()]
        worst = max(worst, float((a - b).abs().max() / b.abs().max()))
    print(f"\\n[full-geometry adapter gradients, {dtype}] loss {float(loss):.5f} vs {float(ref_loss):.5f}; cos {cos:.6f}; norm ratio {nr:.4f}; "
          f"worst per-tensor rel-max {worst:.3e}")
    if dtype == torch.float32:
        assert 1 - cos < 1e-7 and abs(nr - 1) < 1e-4 and worst < 1e-3
    else:
        assert 1 - cos < 5e-3 and abs(nr - 1) < 3e-2 and worst < 0.15

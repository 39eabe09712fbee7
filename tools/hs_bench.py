"""The 64-token level's attention sub-layers, isolated (20 launches per hipGraph): the two-launch head-sliced route (apad_hs_attention +
apad_hs_out) against the chain it replaces, at the CFG batch (64 samples) and at one stream's half of it.   python tools/hs_bench.py"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops

dev, dt = torch.device("cuda:0"), torch.bfloat16
C, H, N = 640, 8, 64


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
for B in (64, 32):
    x = R(B, N, C)
    ln = (1 + 0.1 * R(C), 0.1 * R(C), 1e-5)
    wq, wk, wv, wo, bo = R(C, C, std=0.03), R(C, C, std=0.03), R(C, C, std=0.03), R(C, C, std=0.03), R(C, std=0.1)
    pk, bb = ops.hs_pack_qkv(wq, wk, wv, ln=ln, q_scale=ops.LOG2E / math.sqrt(80))
    wo_p, _ = ops.hs_pack_rows(wo)
    o = torch.empty_like(x); out = torch.empty_like(x)
    ta = timeit(lambda: ops.hs_attention(x, pk, bb, self_attention=True, ln_eps=1e-5, q_prescaled=True, out=o))
    tb = timeit(lambda: ops.hs_out(o, wo_p, bo, x, rowstat=True, out=out))
    tab = timeit(lambda: ops.hs_out(ops.hs_attention(x, pk, bb, self_attention=True, ln_eps=1e-5, q_prescaled=True, out=o), wo_p, bo, x, rowstat=True, out=out))
    # the chain: LN-folded q|k|v GEMM (row statistics from the producer), attention, to_out + residual
    xp = ops.linear(R(B, N, C), wo, bo, residual=x, rowstat=True)
    wqkv = torch.cat([wq, wk, wv], 0).contiguous()
    q_, k_ = torch.empty_like(x), torch.empty_like(x)
    vt = torch.zeros(B, H, 80, 64, device=dev, dtype=dt)
    def chain():
        ops.linear_qkv(xp, wqkv, B, N, H, q_, k_, vt, ln=ln)
        oo = ops.attention(q_, k_, vt, N, H)
        return ops.linear(oo, wo, bo, residual=xp, rowstat=True)
    tc = timeit(chain)
    fl = 2.0 * B * N * C * C * 4 + 4.0 * B * H * N * N * 80
    print(f"self  B={B}: hs_attention {ta:6.1f} us  hs_out {tb:6.1f} us  both {tab:6.1f} us ({fl / tab / 1e6:6.0f} TF/s)   chain {tc:6.1f} us", flush=True)
    for Lt, La in ((8, 32), (16, 0), (8, 128)):
        k1 = R(B, Lt, C); v1 = torch.zeros(B, H, 80, ops.round_up(Lt, 32), device=dev, dtype=dt); v1[..., :Lt] = R(B, H, 80, Lt)
        k2 = v2 = None
        if La:
            k2 = R(B, La, C); v2 = torch.zeros(B, H, 80, ops.round_up(La, 32), device=dev, dtype=dt); v2[..., :La] = R(B, H, 80, La)
        wq_p, qb = ops.hs_pack_rows(wq, ln=ln)
        tx = timeit(lambda: ops.hs_attention(x, wq_p, qb, self_attention=False, ln_eps=1e-5, k1=k1, vt1=v1, k2=k2, vt2=v2, scale2=0.55, out=o))
        txb = timeit(lambda: ops.hs_out(ops.hs_attention(x, wq_p, qb, self_attention=False, ln_eps=1e-5, k1=k1, vt1=v1, k2=k2, vt2=v2, scale2=0.55, out=o), wo_p, bo, x, rowstat=True, out=out))
        def xchain():
            qq = ops.fused_linear(xp, wq, ln=ln)
            oo = ops.attention(qq, k1, v1, Lt, H, k2=k2, vt2=v2, L2=La, scale2=0.55)
            return ops.linear(oo, wo, bo, residual=xp, rowstat=True)
        tcx = timeit(xchain)
        print(f"cross B={B} keys {Lt}+{La}: hs_attention {tx:6.1f} us  both {txb:6.1f} us   chain {tcx:6.1f} us", flush=True)
    # feed-forward: LN + GEGLU (hs_geglu) / FF2, against the chain's two launches
    w1, b1, w2, b2 = R(8 * C, C, std=0.03), R(8 * C, std=0.1), R(C, 4 * C, std=0.02), R(C, std=0.1)
    pg, bg = ops.hs_pack_geglu(w1, b1, ln=ln)
    hh = torch.empty(B, N, 4 * C, device=dev, dtype=dt)
    tg = timeit(lambda: ops.hs_geglu(x, pg, bg, ln_eps=1e-5, out=hh))
    tgc = timeit(lambda: ops.fused_linear(xp, w1, b1, ln=ln, act="geglu", out=hh))
    tf2 = timeit(lambda: ops.linear(hh, w2, b2, residual=x, rowstat=True, out=out))
    w2p = ops.hs_pack_ff2(w2)
    tf2h = timeit(lambda: ops.hs_ff2(hh, w2p, b2, x, rowstat=True, out=out))
    tffh = timeit(lambda: ops.hs_ff2(ops.hs_geglu(x, pg, bg, ln_eps=1e-5, out=hh), w2p, b2, x, rowstat=True, out=out))
    print(f"ff    B={B}: hs_ff2 {tf2h:6.1f} us ({2.0 * B * N * C * 4 * C / tf2h / 1e6:6.0f} TF/s)   hs_geglu + hs_ff2 {tffh:6.1f} us", flush=True)
    print(f"ff    B={B}: hs_geglu {tg:6.1f} us ({2.0 * B * N * C * 8 * C / tg / 1e6:6.0f} TF/s)   chain LN-folded GEGLU GEMM {tgc:6.1f} us   FF2 GEMM {tf2:6.1f} us", flush=True)

# ---- LayerNorm + q|k|v + self-attention in one launch at the two large levels, against the two launches it replaces ----
for (N2, C2, B) in ((1000, 256, 64), (1000, 256, 32), (252, 384, 64)):
    x = R(B, N2, C2)
    ln = (1 + 0.1 * R(C2), 0.1 * R(C2), 1e-5)
    wq, wk, wv = R(C2, C2, std=0.03), R(C2, C2, std=0.03), R(C2, C2, std=0.03)
    pk, csbb = ops.sattn_pack(wq, wk, wv, ln, H)
    o = torch.empty_like(x)
    tf = timeit(lambda: ops.self_attention_fused(x, pk, csbb, H, 1e-5, out=o))
    wqkv = torch.cat([(wq.float() * ops.LOG2E / math.sqrt(C2 // H)).to(dt), wk, wv], 0).contiguous()
    q_, k_ = torch.empty_like(x), torch.empty_like(x)
    vt = torch.zeros(B, H, C2 // H, ops.round_up(N2, 32), device=dev, dtype=dt)
    t1 = timeit(lambda: ops.rowpanel(x, wqkv, [(q_, None, C2, "row"), (k_, None, C2, "row"), (vt, None, C2, "vt")], ln=ln, vt_geom=(H, C2 // H, N2, vt.shape[-1])))
    t2 = timeit(lambda: ops.attention(q_, k_, vt, N2, H, q_prescaled=True, out=o))
    fl = 2.0 * B * N2 * C2 * 3 * C2 + 4.0 * B * N2 * N2 * C2
    print(f"sattn N={N2} C={C2} B={B}: fused {tf:6.1f} us ({fl / tf / 1e6:6.0f} TF/s)   row-panel LN+q|k|v {t1:6.1f} us + attention {t2:6.1f} us = {t1 + t2:6.1f} us", flush=True)

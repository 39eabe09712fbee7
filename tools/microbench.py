"""Per-kernel timing at the real AudioLDM2-large shapes (batch 32 -> 64 sample-forwards).  HIP events, random data.
usage: python tools/microbench.py [filter]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ap_adapter_amd as A
from ap_adapter_amd import ops

dev = torch.device("cuda:0")
dt = torch.bfloat16
flt = sys.argv[1] if len(sys.argv) > 1 else ""
B2 = int(os.environ.get("B2", "64"))


def timeit(fn, iters=20):
    """average kernel time per call: `iters` calls captured into one hipGraph (no host launch overhead), replayed 5x"""
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
    return e0.elapsed_time(e1) / (5 * iters)


def report(name, ms, flops, bytes_):
    print(f"{name:58s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.1f} TF/s  {bytes_/ms/1e6:8.1f} GB/s", flush=True)


def R(*s, std=1.0):
    return (torch.randn(*s, device=dev) * std).to(dt)


LEVELS = [(1000, 256), (252, 384), (64, 640)]
for N, C in LEVELS:
    M = B2 * N
    x = R(M, C)
    if "gemm" in flt or not flt:
        for name, n_out, act in (("proj CxC", C, None), ("geglu Cx8C", 8 * C, "geglu")):
            w = R(n_out, C, std=0.02); b = R(n_out, std=0.02)
            n_eff = n_out // 2 if act else n_out
            out = torch.empty(M, n_eff, device=dev, dtype=dt)
            ms = timeit(lambda: ops.linear(x, w, b, act=act, out=out))
            report(f"gemm {name} M={M} K={C}", ms, 2.0 * M * C * n_out, 2.0 * (M * C + n_out * C + M * n_eff))
        wp = R(C, C, std=0.02); bp = R(C, std=0.02); outp = torch.empty(M, C, device=dev, dtype=dt)
        ms = timeit(lambda: ops.linear(x, wp, bp, residual=x, out=outp))
        report(f"gemm proj CxC+res (tiled) M={M} K={C}", ms, 2.0 * M * C * C, 2.0 * (3 * M * C + C * C))
        h = R(M, 4 * C); w2 = R(C, 4 * C, std=0.02); b2 = R(C, std=0.02); out = torch.empty(M, C, device=dev, dtype=dt)
        ms = timeit(lambda: ops.linear(h, w2, b2, residual=x, out=out))
        report(f"gemm ff2 4CxC+res M={M} K={4*C}", ms, 2.0 * M * 4 * C * C, 2.0 * (M * 4 * C + 4 * C * C + 2 * M * C))
        vt = torch.zeros(B2, 8, C // 8, ops.round_up(N, 32), device=dev, dtype=dt)
        wv = R(C, C, std=0.02)
        ms = timeit(lambda: ops.linear_vt(x, wv, B2, N, 8, vt))
        report(f"gemm V^T out M={M} K={C}", ms, 2.0 * M * C * C, 2.0 * (2 * M * C + C * C))
    if ("rp" in flt or not flt) and C in ops.RP_K:
        g = R(C); be = R(C)
        wqkv = R(3 * C, C, std=0.02)
        qo = torch.empty(M, C, device=dev, dtype=dt); ko = torch.empty(M, C, device=dev, dtype=dt)
        vt = torch.zeros(B2, 8, C // 8, ops.round_up(N, 32), device=dev, dtype=dt)
        ms = timeit(lambda: ops.rowpanel(x, wqkv, [(qo, None, C, "row"), (ko, None, C, "row"), (vt, None, C, "vt")],
                                         ln=(g, be, 1e-5), vt_geom=(8, C // 8, N, vt.shape[-1])))
        report(f"rp LN+qkv M={M} K={C}", ms, 2.0 * M * C * 3 * C, 2.0 * (4 * M * C + 3 * C * C))
        w = R(C, C, std=0.02); b = R(C, std=0.02); out = torch.empty(M, C, device=dev, dtype=dt)
        ms = timeit(lambda: ops.fused_linear(x, w, b, residual=x, out=out))
        report(f"rp proj+bias+res M={M} K={C}", ms, 2.0 * M * C * C, 2.0 * (3 * M * C + C * C))
        ms = timeit(lambda: ops.fused_linear(x, w, None, ln=(g, be, 1e-5), out=out))
        report(f"rp LN+q M={M} K={C}", ms, 2.0 * M * C * C, 2.0 * (2 * M * C + C * C))
        wg = R(8 * C, C, std=0.02); bg = R(8 * C, std=0.02); og = torch.empty(M, 4 * C, device=dev, dtype=dt)
        ms = timeit(lambda: ops.fused_linear(x, wg, bg, ln=(g, be, 1e-5), act="geglu", out=og))
        report(f"rp LN+geglu M={M} K={C}", ms, 2.0 * M * C * 8 * C, 2.0 * (M * C + 8 * C * C + 4 * M * C))
    if "attn" in flt or not flt:
        q = R(B2, N, C); k = R(B2, N, C); v = R(B2, 8, C // 8, ops.round_up(N, 32))
        out = torch.empty(B2, N, C, device=dev, dtype=dt)
        ms = timeit(lambda: ops.attention(q, k, v, N, 8, out=out))
        report(f"attn self N={N} C={C} d={C//8}", ms, 4.0 * B2 * N * N * C, 2.0 * 4 * B2 * N * C)
        for La in (32, 512):
            kt = R(B2, 8, C); vt_ = R(B2, 8, C // 8, 32); ka = R(B2, La, C); va = R(B2, 8, C // 8, ops.round_up(La, 32))
            ms = timeit(lambda: ops.attention(q, kt, vt_, 8, 8, k2=ka, vt2=va, L2=La, scale2=0.5, out=out))
            report(f"attn decoupled N={N} C={C} Lt=8 La={La}", ms, 4.0 * B2 * N * (8 + La) * C, 2.0 * (2 * B2 * N * C + 2 * B2 * (8 + La) * C))
    if "norm" in flt or not flt:
        g = R(C); b = R(C); out = torch.empty(M, C, device=dev, dtype=dt)
        ms = timeit(lambda: ops.layer_norm(x, g, b, 1e-5, out=out))
        report(f"layernorm M={M} C={C}", ms, 0, 4.0 * M * C)
        x3 = x.view(B2, N, C); out3 = out.view(B2, N, C)
        ms = timeit(lambda: ops.group_norm(x3, g, b, 32, 1e-5, silu=True, out=out3))
        report(f"groupnorm+silu B={B2} HW={N} C={C}", ms, 0, 6.0 * M * C)
if "conv" in flt or not flt:
    for (H, W, Cin, Cout, stride) in [(250, 16, 128, 128, 1), (125, 8, 256, 256, 1), (125, 8, 512, 256, 1), (63, 4, 384, 384, 1),
                                      (32, 2, 640, 640, 1), (32, 2, 1280, 640, 1), (250, 16, 256, 128, 1), (250, 16, 128, 128, 2)]:
        x = R(B2, H * W, Cin); w = R(Cout, 9 * Cin, std=0.02); b = R(Cout, std=0.02)
        ms = timeit(lambda: ops.conv3x3(x, w, b, B2, H, W, stride=stride))
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        report(f"conv3x3 {H}x{W} {Cin}->{Cout} s{stride}", ms, 2.0 * B2 * Ho * Wo * 9 * Cin * Cout, 2.0 * (B2 * H * W * Cin + B2 * Ho * Wo * Cout))

if "mlp" in flt or not flt:
    for N, C in LEVELS[:1]:
        M = B2 * N
        x = R(M, C)
        w1 = R(8 * C, C, std=0.02); b1 = R(8 * C, std=0.02); w2 = R(C, 4 * C, std=0.02); b2 = R(C, std=0.02)
        g = R(C); be = R(C); out = torch.empty(M, C, device=dev, dtype=dt)
        hbuf = torch.empty(M, 4 * C, device=dev, dtype=dt)
        ms = timeit(lambda: ops.geglu_mlp(x, w1, b1, w2, b2, ln=(g, be, 1e-5), out=out))
        report(f"fused mlp LN+GEGLU+FF2+res M={M} C={C}", ms, 2.0 * M * 12 * C * C, 2.0 * (2 * M * C + 12 * C * C))

        def two():
            ops.fused_linear(x, w1, b1, ln=(g, be, 1e-5), act="geglu", out=hbuf)
            ops.linear(hbuf, w2, b2, residual=x, out=out)
        ms = timeit(two)
        report(f"  two-kernel LN+GEGLU ; FF2+res M={M} C={C}", ms, 2.0 * M * 12 * C * C, 2.0 * (2 * M * C + 12 * C * C + 2 * M * 4 * C))

if "xattn" in flt or not flt:
    N, C, H, Lt, La = 1000, 256, 8, 8, 32
    x = R(B2, N, C); g = R(C); be = R(C); wq = R(C, C, std=0.02); wo = R(C, C, std=0.02); bo = R(C, std=0.02)
    k1 = R(B2, Lt, C, std=0.3); k2 = R(B2, La, C, std=0.3)
    v1t = torch.zeros(B2, H, C // H, 32, device=dev, dtype=dt); v1t[..., :Lt].normal_(0, 0.3)
    v2t = torch.zeros(B2, H, C // H, 32, device=dev, dtype=dt); v2t[..., :La].normal_(0, 0.3)
    out = torch.empty_like(x)
    fl = 2.0 * 2 * B2 * N * C * C + 4.0 * B2 * N * (Lt + La) * C
    by = 2.0 * (2 * B2 * N * C + 2 * C * C + 2 * B2 * (Lt + La) * C)
    (wq_p, q_fold), wo_p = ops.xattn_pack_weight(wq, (g, be, 1e-5)), ops.xattn_pack_weight(wo)
    pk1, pk2 = ops.xattn_pack_kv(k1, v1t, Lt), ops.xattn_pack_kv(k2, v2t, La)
    ms = timeit(lambda: ops.fused_cross_attention(x, wq_p, wo_p, bo, pk1, Lt, H, ln=(g, be, 1e-5), kv2_packed=pk2, L2=La, scale2=0.55, out=out, q_fold=q_fold))
    report(f"fused cross-attention sub-layer N={N} C={C} Lt=8 La=32", ms, fl, by)

    def three():
        qd = ops.fused_linear(x, wq, ln=(g, be, 1e-5))
        od = ops.attention(qd, k1, v1t, Lt, H, k2=k2, vt2=v2t, L2=La, scale2=0.55)
        ops.fused_linear(od, wo, bo, residual=x, out=out)
    ms = timeit(three)
    report(f"  three kernels: LN+q ; decoupled attn ; to_out+res", ms, fl, by + 2.0 * 4 * B2 * N * C)

"""Run ONE op at its real shape a few times (for rocprofv3 --pmc passes).  usage: python tools/one_op.py <op> [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
op = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B2 = 64
R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
if op == "attn_self":
    N, C = 1000, 256
    # q as the model hands it over: already multiplied by log2(e) / sqrt(d) (the direct-form kernel of the 1000-token level)
    q = R(B2, N, C, std=1.4427 / 32 ** 0.5); k = R(B2, N, C); v = R(B2, 8, C // 8, ops.round_up(N, 32)); out = torch.empty_like(q)
    fn = lambda: ops.attention(q, k, v, N, 8, out=out, q_prescaled=True)
elif op == "sattn_fused":
    N, C = 1000, 256
    x = R(B2, N, C); g = R(C); be = R(C); wq, wk, wv = R(C, C, std=0.05), R(C, C, std=0.05), R(C, C, std=0.05); out = torch.empty_like(x)
    wp, cs = ops.sattn_pack(wq, wk, wv, (g, be, 1e-5), 8)
    fn = lambda: ops.self_attention_fused(x, wp, cs, 8, 1e-5, out=out)
elif op == "attn_ip":
    N, C, La = 1000, 256, 32
    q = R(B2, N, C); kt = R(B2, 8, C); vt_ = R(B2, 8, C // 8, 32); ka = R(B2, La, C); va = R(B2, 8, C // 8, ops.round_up(La, 32)); out = torch.empty_like(q)
    fn = lambda: ops.attention(q, kt, vt_, 8, 8, k2=ka, vt2=va, L2=La, scale2=0.5, out=out)
elif op == "rp_geglu":
    M, C = B2 * 1000, 256
    x = R(M, C); g = R(C); be = R(C); wg = R(8 * C, C, std=0.02); bg = R(8 * C, std=0.02); og = torch.empty(M, 4 * C, device=dev, dtype=dt)
    fn = lambda: ops.fused_linear(x, wg, bg, ln=(g, be, 1e-5), act="geglu", out=og)
elif op == "rp_qkv":
    M, C, N = B2 * 1000, 256, 1000
    x = R(M, C); g = R(C); be = R(C); w = R(3 * C, C, std=0.02)
    qo = torch.empty(M, C, device=dev, dtype=dt); ko = torch.empty(M, C, device=dev, dtype=dt)
    vt = torch.zeros(B2, 8, C // 8, ops.round_up(N, 32), device=dev, dtype=dt)
    fn = lambda: ops.rowpanel(x, w, [(qo, None, C, "row"), (ko, None, C, "row"), (vt, None, C, "vt")], ln=(g, be, 1e-5), vt_geom=(8, C // 8, N, vt.shape[-1]))
elif op == "gemm_ff2":
    M, C = B2 * 1000, 256
    h = R(M, 4 * C); w2 = R(C, 4 * C, std=0.02); b2 = R(C, std=0.02); x = R(M, C); out = torch.empty(M, C, device=dev, dtype=dt)
    fn = lambda: ops.linear(h, w2, b2, residual=x, out=out)
elif op == "mlp":
    M, C = B2 * 1000, 256
    x = R(M, C); g = R(C); be = R(C); w1 = R(8 * C, C, std=0.02); b1 = R(8 * C, std=0.02)
    w2 = R(C, 4 * C, std=0.02); b2 = R(C, std=0.02); out = torch.empty(M, C, device=dev, dtype=dt)
    fn = lambda: ops.geglu_mlp(x, w1, b1, w2, b2, ln=(g, be, 1e-5), out=out)
elif op == "mlp3":
    M, C = B2 * 1000, 256
    x = R(M, C); g = R(C); be = R(C); w1 = R(8 * C, C, std=0.02); b1 = R(8 * C, std=0.02)
    w2 = R(C, 4 * C, std=0.02); b2 = R(C, std=0.02); out = torch.empty(M, C, device=dev, dtype=dt)
    wp, bp = ops.mlp_pack(w1, b1, w2)
    fn = lambda: ops.geglu_mlp_packed(x, wp, bp, b2, ln=(g, be, 1e-5), out=out)
elif op == "xattn":
    N, C, H, Lt, La = 1000, 256, 8, 8, int(os.environ.get("LA", "32"))
    x, g, be, wq, wo, bo = R(B2, N, C), R(C), R(C), R(C, C, std=0.02), R(C, C, std=0.02), R(C, std=0.02)
    k1, k2 = R(B2, Lt, C, std=0.3), R(B2, La, C, std=0.3)
    v1t = torch.zeros(B2, H, 32, 32, device=dev, dtype=dt); v1t[..., :Lt].normal_(0, 0.3)
    v2t = torch.zeros(B2, H, 32, ops.round_up(La, 32), device=dev, dtype=dt); v2t[..., :La].normal_(0, 0.3)
    (wq_p, q_fold), wo_p = ops.xattn_pack_weight(wq, (g, be, 1e-5)), ops.xattn_pack_weight(wo)
    pk1, pk2 = ops.xattn_pack_kv(k1, v1t, Lt), ops.xattn_pack_kv(k2, v2t, La)
    out = torch.empty_like(x)
    fn = lambda: ops.fused_cross_attention(x, wq_p, wo_p, bo, pk1, Lt, H, ln=(g, be, 1e-5), kv2_packed=pk2, L2=La, scale2=0.55, out=out, q_fold=q_fold)
elif op in ("mlp384", "mlp384_chain"):
    M, C = B2 * 252, 384
    x = R(M, C); g = R(C); be = R(C); w1 = R(8 * C, C, std=0.02); b1 = R(8 * C, std=0.02)
    w2 = R(C, 4 * C, std=0.02); b2 = R(C, std=0.02); out = torch.empty(M, C, device=dev, dtype=dt)
    if op == "mlp384":
        fn = lambda: ops.geglu_mlp(x, w1, b1, w2, b2, ln=(g, be, 1e-5), out=out)
    else:
        def fn():
            h = ops.fused_linear(x, w1, b1, ln=(g, be, 1e-5), act="geglu")
            return ops.linear(h, w2, b2, residual=x, out=out)
elif op in ("geglu3", "rp_geglu384"):
    M, C = B2 * 252, 384
    x = R(M, C); g = R(C); be = R(C); w1 = R(8 * C, C, std=0.02); b1 = R(8 * C, std=0.02); og = torch.empty(M, 4 * C, device=dev, dtype=dt)
    if op == "geglu3":
        wp, bp = ops.geglu_pack(w1, b1)
        fn = lambda: ops.layernorm_geglu_packed(x, wp, bp, ln=(g, be, 1e-5), out=og)
    else:
        fn = lambda: ops.fused_linear(x, w1, b1, ln=(g, be, 1e-5), act="geglu", out=og)
elif op in ("xrows", "xrows_chain"):
    C = int(os.environ.get("XC", "384"))
    N, H, Lt, La = (252 if C == 384 else 64), 8, 8, int(os.environ.get("LA", "32"))
    B2 = int(os.environ.get("XB", B2))
    x, g, be, wq, wo, bo = R(B2, N, C), R(C), R(C), R(C, C, std=0.02), R(C, C, std=0.02), R(C, std=0.02)
    k1, k2 = R(B2, Lt, C, std=0.3), R(B2, La, C, std=0.3)
    v1t = torch.zeros(B2, H, C // H, 32, device=dev, dtype=dt); v1t[..., :Lt].normal_(0, 0.3)
    v2t = torch.zeros(B2, H, C // H, ops.round_up(La, 32), device=dev, dtype=dt); v2t[..., :La].normal_(0, 0.3)
    wq_p, wo_p = ops.xrows_pack_weight(wq), ops.xrows_pack_weight(wo)
    out = torch.empty_like(x)
    if op == "xrows":
        fn = lambda: ops.cross_attention_rows(x, wq_p, wo_p, bo, k1, v1t, H, ln=(g, be, 1e-5), k2=k2, vt2=v2t, scale2=0.55, out=out)
    else:
        def fn():
            qd = ops.fused_linear(x, wq, ln=(g, be, 1e-5))
            od = ops.attention(qd, k1, v1t, Lt, H, k2=k2, vt2=v2t, L2=La, scale2=0.55)
            return ops.fused_linear(od, wo, bo, residual=x, out=out)
elif op in ("hconv256", "hconv128", "hconv640"):  # the halo-resident 3x3 convolution (csrc/hconv.hip) at the 1000- / 4000- / 64-pixel level
    H_, W_, Ci, Co = {"hconv256": (125, 8, 256, 256), "hconv128": (250, 16, 128, 128), "hconv640": (32, 2, 640, 640)}[op]
    x = R(B2, H_ * W_, Ci, std=0.5); w = R(Co, 9 * Ci, std=0.02); b = R(Co, std=0.1); r = R(B2, H_ * W_, Co)
    out = torch.empty(B2, H_ * W_, Co, device=dev, dtype=dt)
    fn = lambda: ops.conv3x3(x, w, b, B2, H_, W_, residual=r, out=out)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
if os.environ.get("TIME"):  # event timing of the op alone (isolated; the in-step figure is what counts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{op} {os.environ.get('APAD_LIB_PATH', 'product')}: {e0.elapsed_time(e1) / n * 1000:.1f} us")


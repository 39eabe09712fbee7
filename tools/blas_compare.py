"""How do the vendor BLAS libraries (through torch) do on the plain GEMM shapes of the step?  (context for DESIGN.md)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ap_adapter_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16


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


for M, K, N in [(4096, 640, 640), (4096, 2560, 640), (16128, 384, 384), (16128, 1536, 384), (64000, 256, 256), (64000, 1024, 256), (64000, 256, 2048)]:
    x = (torch.randn(M, K, device=dev)).to(dt); w = (torch.randn(N, K, device=dev) * 0.02).to(dt); b = torch.zeros(N, device=dev, dtype=dt)
    r = torch.randn(M, N, device=dev).to(dt); out = torch.empty(M, N, device=dev, dtype=dt)
    t_ours = timeit(lambda: ops.linear(x, w, b, residual=r, out=out))
    t_lib = timeit(lambda: torch.addmm(r, x, w.t(), out=out))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} K={K:5d} N={N:5d}  ours {t_ours:7.1f} us ({fl/t_ours/1e6:6.0f} TF/s)   torch.addmm {t_lib:7.1f} us ({fl/t_lib/1e6:6.0f} TF/s)")

# attention context: torch's fused SDPA on the roofline kernel's problem (64 samples x 8 heads x 1000 x 1000, d = 32)
for (B, H, N, d) in [(64, 8, 1000, 32), (64, 8, 252, 48), (64, 8, 64, 80)]:
    q = (torch.randn(B, N, H * d, device=dev) * 0.32).to(dt); k = (torch.randn(B, N, H * d, device=dev) * 0.32).to(dt)
    v = (torch.randn(B, N, H * d, device=dev) * 0.32).to(dt)
    vt = ops.head_transpose(v, H) if hasattr(ops, "head_transpose") else None
    out = torch.empty_like(q)
    t_ours = timeit(lambda: ops.attention(q, k, vt, N, H, out=out))
    q4, k4, v4 = (t.reshape(B, N, H, d).transpose(1, 2) for t in (q, k, v))
    try:
        t_lib = timeit(lambda: F.scaled_dot_product_attention(q4, k4, v4))
    except Exception as e:
        t_lib = float("nan"); print("sdpa failed:", e)
    fl = 4.0 * B * H * N * N * d
    print(f"attention B={B} H={H} N={N} d={d}: ours {t_ours:7.1f} us ({fl/t_ours/1e6:6.0f} TF/s)   torch SDPA {t_lib:7.1f} us ({fl/t_lib/1e6:6.0f} TF/s)")

"""to_out / proj_out launches of the two large levels (weight-stationary row-panel kernel + residual), hipGraph-timed, with a check against fp32 torch.
    python tools/ws_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops

dev, dt = torch.device("cuda:0"), torch.bfloat16


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
for (M, C) in ((64000, 256), (32000, 256), (16128, 384), (8064, 384), (63990, 256)):
    o, x = R(M, C), R(M, C)
    w, b = R(C, C, std=0.05), R(C, std=0.1)
    out = torch.empty_like(x)
    t = timeit(lambda: ops.fused_linear(o, w, b, residual=x, out=out))
    ref = o.float() @ w.float().t() + b.float() + x.float()
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    print(f"to_out M={M} C={C}: {t:6.1f} us  ({(3 * M * C * 2 + C * C * 2) / t / 1e6:5.2f} TB/s algorithmic)  rel err {err:.2e}", flush=True)

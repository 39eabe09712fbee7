import sys, os, math
sys.path.insert(0, "/root/repo")
import torch
from ap_adapter_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
B2, N, C, H = 64, 1000, 256, 8
def t_eager(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def run(name, q, k, vt):
    out = torch.empty_like(q)
    print(f"{name:40s} {t_eager(lambda: ops.attention(q, k, vt, N, H, out=out)):7.1f} us")
for std in (1.0, 0.32, 0.05):
    q = (torch.randn(B2, N, C, device=dev) * std).to(dt); k = (torch.randn(B2, N, C, device=dev) * std).to(dt)
    vt = torch.zeros(B2, H, C // H, 1024, device=dev, dtype=dt); vt[..., :N].normal_(0, std)
    run(f"randn std {std}", q, k, vt)
# in-model style: LayerNorm-ed activations through N(0, 0.02^2) q|k|v weights
x = torch.randn(B2, N, C, device=dev).to(dt)
g = torch.ones(C, device=dev, dtype=dt); be = torch.zeros(C, device=dev, dtype=dt)
w = (torch.randn(3 * C, C, device=dev) * 0.02).to(dt)
q = torch.empty(B2, N, C, device=dev, dtype=dt); k = torch.empty_like(q)
vt = torch.zeros(B2, H, C // H, 1024, device=dev, dtype=dt)
ops.rowpanel(x.reshape(-1, C), w, [(q, None, C, "row"), (k, None, C, "row"), (vt, None, C, "vt")], ln=(g, be, 1e-5), vt_geom=(H, C // H, N, 1024))
run("LN + q|k|v projection of randn", q, k, vt)
# correlated tokens (smooth along the sequence, like feature maps)
base = torch.randn(B2, 1, C, device=dev) + 0.1 * torch.randn(B2, N, C, device=dev)
x = base.to(dt)
ops.rowpanel(x.reshape(-1, C), w, [(q, None, C, "row"), (k, None, C, "row"), (vt, None, C, "vt")], ln=(g, be, 1e-5), vt_geom=(H, C // H, N, 1024))
run("LN + q|k|v of strongly correlated tokens", q, k, vt)

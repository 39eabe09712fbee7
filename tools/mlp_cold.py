"""The fused feed-forward with COLD weights: every call uses the next of NSETS weight sets (300 MB in all: nothing stays in L2 / the Infinity Cache
between two uses of a set), as in the denoise / training step, where each of the 40 feed-forwards has its own 3 MB.  usage: python tools/mlp_cold.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops
dev, dt, C, NSETS = torch.device("cuda:0"), torch.bfloat16, 256, 100
R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
sets = []
for i in range(NSETS):
    w1, b1, w2, b2 = R(8 * C, C, std=0.06), R(8 * C, std=0.3), R(C, 4 * C, std=0.04), R(C, std=0.3)
    sets.append((w1, b1, w2, b2) + ops.mlp_pack(w1, b1, w2))
ln = (1 + 0.1 * R(C), 0.1 * R(C), 1e-5)


def timeit(fn, iters=NSETS):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for i in range(iters):
            fn(i)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3


for M in [int(a) for a in sys.argv[1:]] or [64000, 32000, 16000, 4000]:
    x = R(M, C); o = torch.empty_like(x)
    t_old = timeit(lambda i: ops.geglu_mlp(x, *sets[i % NSETS][:4], ln=ln, out=o))
    t_new = timeit(lambda i: ops.geglu_mlp_packed(x, sets[i % NSETS][4], sets[i % NSETS][5], sets[i % NSETS][3], ln=ln, out=o))
    print(f"M={M} cold weights: geglu_mlp {t_old:7.1f} us   packed {t_new:7.1f} us   lib={os.environ.get('APAD_LIB_PATH', 'product')}", flush=True)

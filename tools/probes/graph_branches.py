"""Do parallel branches of a captured hipGraph run concurrently on this ROCm?  Two independent chains of latency-bound launches
(M = 2048, K = N = 640 GEMMs: ~9 us each, 320 workgroups) captured (a) back to back on one stream, (b) forked onto two streams;
and the same with 4 chains / 4 streams.  If (b) ~ (a) / 2 the branches overlap; if (b) ~ (a) they are serialised."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ap_adapter_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
NCH, LEN = 4, 40
xs = [(torch.randn(2048, 640, device=dev) * 0.5).to(dt) for _ in range(NCH)]
w = (torch.randn(640, 640, device=dev) * 0.04).to(dt)
bufs = [[torch.empty(2048, 640, device=dev, dtype=dt) for _ in range(2)] for _ in range(NCH)]
def chain(c):
    a = xs[c]
    for i in range(LEN):
        a = ops.linear(a, w, out=bufs[c][i & 1])
streams = [torch.cuda.Stream() for _ in range(NCH)]
def run(nchains, forked):
    cur = torch.cuda.current_stream()
    if not forked:
        for c in range(nchains): chain(c)
        return
    for c in range(nchains):
        streams[c].wait_stream(cur)
        with torch.cuda.stream(streams[c]): chain(c)
    for c in range(nchains): cur.wait_stream(streams[c])
for nchains in (1, 2, 4):
    for forked in (False, True):
        run(nchains, forked); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run(nchains, forked)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{nchains} chain(s) x {LEN} launches, {'forked onto streams' if forked else 'one stream'}: {ms * 1e3:8.1f} us  ({ms * 1e3 / (nchains * LEN):.2f} us per launch)")

"""Probe: can HIP events bracket ONE kernel INSIDE a captured hipGraph (torch.cuda.Event(external=True) -> event-record nodes)?
Prints the per-launch time of a known kernel measured three ways: external events inside a graph, plain events around eager
launches, and graph-of-20 average."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
x = torch.randn(64, 1000, 256, device=dev).bfloat16()
y = torch.empty_like(x)
def k():
    torch.mul(x, 2.0, out=y)
k(); torch.cuda.synchronize()
# eager events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    e0.record(); k(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
print("eager event pair around one launch: median %.2f us min %.2f" % (sorted(ts)[10], min(ts)))
ts = []
for _ in range(20):
    e0.record(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
print("eager empty event pair: median %.2f us" % sorted(ts)[10])
try:
    evs = [(torch.cuda.Event(enable_timing=True, external=True), torch.cuda.Event(enable_timing=True, external=True)) for _ in range(8)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        k()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for a, b in evs:
            k(); k()
            a.record(); k(); b.record()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("external events inside a graph: us per launch", ["%.2f" % (a.elapsed_time(b) * 1e3) for a, b in evs])
    evs2 = [(torch.cuda.Event(enable_timing=True, external=True), torch.cuda.Event(enable_timing=True, external=True)) for _ in range(4)]
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for a, b in evs2:
            k()
            a.record(); b.record()
    g2.replay(); torch.cuda.synchronize()
    print("external empty pair inside a graph: us", ["%.2f" % (a.elapsed_time(b) * 1e3) for a, b in evs2])
except Exception as e:
    print("external events in graph: FAILED", repr(e))

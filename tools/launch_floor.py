"""Launch floor inside a hipGraph: N dependent launches of a trivial kernel captured on one stream, replayed; time per
launch = the per-kernel cost the captured denoise step pays on top of its kernels' own work (2279 launches per step)."""
import sys
import torch

sys.path.insert(0, ".")
from ap_adapter_amd import ops  # noqa: E402  (apad kernels: step_advance is a 1-thread kernel)

dev = torch.device("cuda:0")
n = 2000
x = torch.zeros(64, device=dev)
ptr = torch.zeros(1, dtype=torch.int32, device=dev)


def timed(fn, label):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1) / (5 * n) * 1e3:7.2f} us per launch (graph of {n})")


timed(lambda: ops.step_advance(ptr), "apad_step_advance (1 thread)")
timed(lambda: x.add_(1.0), "torch add_ on 64 floats")
big = torch.zeros(64, 1000, 256, device=dev, dtype=torch.bfloat16)
g_ = torch.ones(256, device=dev, dtype=torch.bfloat16)
b_ = torch.zeros(256, device=dev, dtype=torch.bfloat16)
small = torch.zeros(64, 64, 640, device=dev, dtype=torch.bfloat16)
g2, b2 = torch.ones(640, device=dev, dtype=torch.bfloat16), torch.zeros(640, device=dev, dtype=torch.bfloat16)
timed(lambda: ops.layer_norm(small, g2, b2, 1e-5), "layer_norm 4096 x 640")
timed(lambda: ops.layer_norm(big, g_, b_, 1e-5), "layer_norm 64000 x 256")

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
def timeit(fn, iters=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
for C in (256, 384):
    w = R(C, C, std=0.02); g = R(C); be = R(C)
    for M in (256, 2048, 8192, 16128, 32768, 64000, 128000):
        x = R(M, C); out = torch.empty(M, C, device=dev, dtype=dt)
        t = timeit(lambda: ops.fused_linear(x, w, None, ln=(g, be, 1e-5), out=out))
        print(f"ws LN+q K={C} M={M:7d}: {t:7.1f} us")
# empty-ish kernel launch floor
z = torch.zeros(1, dtype=torch.int32, device=dev)
print("step_advance launch: %.1f us" % timeit(lambda: ops.step_advance(z)))

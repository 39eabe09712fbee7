"""The fused feed-forward at the 1000-token level (M = 64 000 rows, C = 256): the 128-token kernel (apad_geglu_mlp) against the 64-token register-block
kernel from packed weights (apad_geglu_mlp_packed), hipGraph-timed; with APAD_LIB_PATH=exp/lib_<tag>.so an ablation build of csrc/mlp3.hip
(tools/ab_build.sh <tag> mlp3.hip -DM3_ABL=<bits>).   usage: python tools/mlp_bench.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops

dev, dt, C = torch.device("cuda:0"), torch.bfloat16, 256


def timeit(fn, iters=20, reps=5):
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
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3


R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
w1, b1, w2, b2 = R(8 * C, C, std=0.06), R(8 * C, std=0.3), R(C, 4 * C, std=0.04), R(C, std=0.3)
ln = (1 + 0.1 * R(C), 0.1 * R(C), 1e-5)
wp, bp = ops.mlp_pack(w1, b1, w2)
if __name__ == "__main__":
  for M in [int(a) for a in sys.argv[1:]] or [64000, 32000]:
      x = R(M, C)
      o1, o2 = torch.empty_like(x), torch.empty_like(x)
      t_old = timeit(lambda: ops.geglu_mlp(x, w1, b1, w2, b2, ln=ln, out=o1))
      t_new = timeit(lambda: ops.geglu_mlp_packed(x, wp, bp, b2, ln=ln, out=o2))
      gf = 2.0 * M * C * 12 * C / 1e9
      diff = float((o1.float() - o2.float()).abs().max() / o1.float().abs().max())
      print(f"M={M}: geglu_mlp {t_old:7.1f} us ({gf / t_old * 1e3:6.0f} TF/s)   packed {t_new:7.1f} us ({gf / t_new * 1e3:6.0f} TF/s)   rel diff {diff:.2e}"
            f"   lib={os.environ.get('APAD_LIB_PATH', 'product')}", flush=True)

"""Phase timeline of geglu3_kernel (probe build with s_memtime stamps):
    tools/ab_build.sh g3tr geglu3.hip -DG3_TRACE=0     then     APAD_LIB_PATH=exp/lib_g3tr.so python tools/g3_trace.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ap_adapter_amd as A
from ap_adapter_amd import ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
M, Cc = 64 * 252, 384
R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
x, g, be, w1, b1 = R(M, Cc), R(Cc), R(Cc), R(8 * Cc, Cc, std=0.02), R(8 * Cc, std=0.02)
og = torch.empty(M, 4 * Cc, device=dev, dtype=dt)
wp, bp = ops.geglu_pack(w1, b1)
for _ in range(5):
    ops.layernorm_geglu_packed(x, wp, bp, ln=(g, be, 1e-5), out=og)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (1024 * 40))()
lib = A.lib()
lib.apad_g3_trace_read.argtypes = [C.c_void_p, C.c_int]
assert lib.apad_g3_trace_read(buf, 1024 * 40 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 40).astype(np.int64)
t = t[t[:, 0] > 0]
names = {1: "x panels requested", 2: "bias -> LDS, 5 stages requested", 3: "LayerNorm (x landed)", 4: "AGPR pin + setup", 5: "__syncthreads (DMA drained)", 6: "-> loop"}
print(f"{t.shape[0]} workgroups; mean s_memtime ticks between stamps")
for i in range(1, 6):
    d = t[:, i] - t[:, i - 1]
    print(f"  {names[i]:36s} {d.mean():9.0f}  (min {d.min():7d} max {d.max():7d})")
it = np.concatenate([t[:, 7:32] - t[:, 6:31], (t[:, 36] - t[:, 31])[:, None]], axis=1)
print("  iterations:", " ".join(f"{v:.0f}" for v in it.mean(axis=0)))
print(f"  loop total {(t[:, 36] - t[:, 6]).mean():.0f}; last iterations -> vmcnt(0) {(t[:, 37] - t[:, 36]).mean():.0f}; kernel {(t[:, 37] - t[:, 0]).mean():.0f}")
wc = t[:, 38:40]; w0 = wc[:, 0].min()
print(f"  kernel span {(wc[:, 1].max() - w0) / 100:.1f} us; workgroup starts after the first: median {np.median(wc[:, 0] - w0) / 100:.2f} max {(wc[:, 0].max() - w0) / 100:.2f} us; "
      f"workgroup duration mean {((wc[:, 1] - wc[:, 0]) / 100).mean():.2f} min {((wc[:, 1] - wc[:, 0]) / 100).min():.2f} max {((wc[:, 1] - wc[:, 0]) / 100).max():.2f} us")

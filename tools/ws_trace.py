"""Timeline of the weight-stationary row-panel kernel on a to_out launch (probe build with wall-clock stamps):
    tools/ab_build.sh wstr0 wsgemm.hip -DWS_TRACE=0
    APAD_LIB_PATH=exp/lib_wstr0.so python tools/ws_trace.py [M] [C]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ap_adapter_amd as A
from ap_adapter_amd import ops

dev, dt = torch.device("cuda:0"), torch.bfloat16
M, Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 64000, int(sys.argv[2]) if len(sys.argv) > 2 else 256
R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
o, x, w, b = R(M, Cc), R(M, Cc), R(Cc, Cc, std=0.05), R(Cc, std=0.1)
out = torch.empty_like(x)
for _ in range(5):
    ops.fused_linear(o, w, b, residual=x, out=out)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (1024 * 16))()
lib = A.lib()
lib.apad_ws_trace_read.argtypes = [C.c_void_p, C.c_int]
assert lib.apad_ws_trace_read(buf, 1024 * 16 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 12] > 0)]
w0 = t[:, 0].min()
names = ["start", "weight slice staged (loads + LDS stores)", "bias + barrier"] + [f"tile {i} starts" for i in range(9)] + ["panel done"]
print(f"M={M} C={Cc}: {t.shape[0]} workgroups with a panel for the traced wave; kernel span {(t[:, 12].max() - w0) / 100:.1f} us; "
      f"workgroup start median {np.median(t[:, 0] - w0) / 100:.2f} max {(t[:, 0].max() - w0) / 100:.2f} us")
idx = [i for i in range(13) if (t[:, i] > 0).all()]
for a, b_ in zip(idx[:-1], idx[1:]):
    d = (t[:, b_] - t[:, a]) / 100.0
    print(f"   -> {names[b_]:42s} {d.mean():7.2f} us   (min {d.min():6.2f} max {d.max():6.2f})")
print(f"   total {((t[:, 12] - t[:, 0]) / 100.0).mean():7.2f} us")

"""s_memtime stamps of the halo convolution kernel (probe build: tools/ab_build.sh hctrace hconv.hip -DHC_TRACE=1; run with
APAD_LIB_PATH=exp/lib_hctrace.so): per-tile phases of a few workgroups, in thousands of SHADER cycles (s_memtime counts shader clocks: about 1.6 per ns on real operands).
usage: python tools/hconv_trace.py [B H W Cin Cout]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops, _lib as L

dev = torch.device("cuda:0")
dt = torch.bfloat16
B, H, W, Cin, Cout = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (64, 250, 16, 128, 128)
x = (torch.randn(B, H * W, Cin, device=dev) * 0.5).to(dt)
w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(dt)
b = (torch.randn(Cout, device=dev) * 0.1).to(dt)
out, _, _ = ops.conv3x3(x, w, b, B, H, W)
h = L.lib()
tr = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
h.apad_hconv_set_trace.argtypes = [C.c_void_p]
h.apad_hconv_set_trace(tr.data_ptr())
for _ in range(3):
    tr.zero_()
    ops.conv3x3(x, w, b, B, H, W, out=out)
    torch.cuda.synchronize()
t = tr.view(256, 32).cpu()
names = ["start", "setup+requests"]
for wg in (0, 1, 100, 255):
    r = t[wg]
    base = int(r[0])
    vals = [(int(v) - base) / 1000.0 for v in r[:19] if int(v) != 0]
    print(f"wg {wg:3d}: " + " ".join(f"{v:6.2f}" for v in vals))
for wg in (0, 100):
    r = t[wg]
    st = [int(v) for v in r[19:31]]
    if all(st):
        print(f"wg {wg:3d} stage starts of the first tile's second chunk, deltas in cycles: " + " ".join(str(b - a) for a, b in zip(st[:-1], st[1:])))
print("columns: start, first requests out | per tile: loop start, loop end, barrier, next tile's setup + requests out, epilogue issued   (k cycles)")

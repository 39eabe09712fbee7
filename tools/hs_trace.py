"""Phase timeline of the head-sliced sub-layer kernels (probe build with s_memtime stamps):
    tools/ab_build.sh hstr0 hsattn.hip -DHS_TRACE=0     (the traced wave: 0 .. 7)
    APAD_LIB_PATH=exp/lib_hstr0.so python tools/hs_trace.py"""
import ctypes as C, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ap_adapter_amd as A
from ap_adapter_amd import ops

dev, dt = torch.device("cuda:0"), torch.bfloat16
B, N, Cc, H = int(os.environ.get("B2", "64")), 64, 640, 8
R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
x = R(B, N, Cc)
ln = (1 + 0.1 * R(Cc), 0.1 * R(Cc), 1e-5)
wq, wk, wv, wo, bo = R(Cc, Cc, std=0.03), R(Cc, Cc, std=0.03), R(Cc, Cc, std=0.03), R(Cc, Cc, std=0.03), R(Cc, std=0.1)
pk, bb = ops.hs_pack_qkv(wq, wk, wv, ln=ln, q_scale=ops.LOG2E / math.sqrt(80))
wo_p, _ = ops.hs_pack_rows(wo)
o, out = torch.empty_like(x), torch.empty_like(x)
for _ in range(5):
    ops.hs_attention(x, pk, bb, self_attention=True, ln_eps=1e-5, q_prescaled=True, out=o)
    ops.hs_out(o, wo_p, bo, x, rowstat=True, out=out)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 1024 * 16))()
lib = A.lib()
lib.apad_hs_trace_read.argtypes = [C.c_void_p, C.c_int]
assert lib.apad_hs_trace_read(buf, 2 * 1024 * 16 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 1024, 16).astype(np.int64)[:, :B * 4]
names = [["start", "weights requested", "rows loaded + normalised + stored", "barrier 1", "projection loop", "q/k/v -> LDS", "barrier 2", "attention", "-", "O stored"],
         ["start", "weights + residual requested", "O rows -> LDS", "barrier", "projection loop", "-", "-", "-", "-", "epilogue + stores"]]
for k, kn in enumerate(("hs_attn_kernel", "hs_out_kernel")):
    tt = t[k]
    idx = [i for i in range(10) if names[k][i] != "-" and (tt[:, i] > 0).all()]
    print(f"{kn}: mean cycles (s_memtime ticks) between stamps of the traced wave, {tt.shape[0]} workgroups")
    for a, b_ in zip(idx[:-1], idx[1:]):
        d = tt[:, b_] - tt[:, a]
        print(f"   -> {names[k][b_]:36s} {d.mean():9.0f}   (min {d.min():7d} max {d.max():7d})")
    print(f"   total {(tt[:, idx[-1]] - tt[:, idx[0]]).mean():9.0f}")
    wc = tt[:, 14:16]
    w0 = wc[:, 0].min()
    print(f"   kernel span {(wc[:, 1].max() - w0) / 100:.1f} us; workgroup starts after the first: median {np.median(wc[:, 0] - w0) / 100:.2f} max {(wc[:, 0].max() - w0) / 100:.2f} us; "
          f"workgroup duration mean {((wc[:, 1] - wc[:, 0]) / 100).mean():.2f} us")

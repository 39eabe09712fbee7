"""Phase timeline of the fused LN + q|k|v + self-attention kernel (probe build with s_memtime stamps):
    tools/ab_build.sh sftr0 attention.hip -DSF_TRACE=0     (the traced wave: 0 .. 7)
    APAD_LIB_PATH=exp/lib_sftr0.so python tools/sf_trace.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ap_adapter_amd as A
from ap_adapter_amd import ops

dev, dt = torch.device("cuda:0"), torch.bfloat16
B, N, Cc, H = int(os.environ.get("B2", "64")), int(os.environ.get("N", "1000")), int(os.environ.get("C", "256")), 8
R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
x = R(B, N, Cc)
ln = (1 + 0.1 * R(Cc), 0.1 * R(Cc), 1e-5)
wp, cs = ops.sattn_pack(R(Cc, Cc, std=0.05), R(Cc, Cc, std=0.05), R(Cc, Cc, std=0.05), ln, H)
out = torch.empty_like(x)
for _ in range(5):
    ops.self_attention_fused(x, wp, cs, H, 1e-5, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.self_attention_fused(x, wp, cs, H, 1e-5, out=out)
e1.record()
torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1) * 100:.1f} us")
buf = (C.c_ulonglong * (1024 * 16))()
lib = A.lib()
lib.apad_sf_trace_read.argtypes = [C.c_void_p, C.c_int]
assert lib.apad_sf_trace_read(buf, 1024 * 16 * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)[:min(B * H, 1024)]
names = ["start", "init (csbb, clears) + barrier", "projection 1", "epilogue 1 (fold, K / V^T -> LDS)", "projection 2", "epilogue 2", "barrier", "key loop 1 + O store",
         "key loop 2 + O store", "end"]
idx = [i for i in range(10) if (t[:, i] > 0).all()]
print(f"mean s_memtime ticks (100 MHz: 100 ticks = 1 us) between stamps of the traced wave, {t.shape[0]} workgroups")
for a, b_ in zip(idx[:-1], idx[1:]):
    d = t[:, b_] - t[:, a]
    print(f"   -> {names[b_]:36s} {d.mean() / 100:8.2f} us   (min {d.min() / 100:6.2f} max {d.max() / 100:6.2f})")
print(f"   total {(t[:, idx[-1]] - t[:, idx[0]]).mean() / 100:8.2f} us")
wc = t[:, 14:16]
w0 = wc[:, 0].min()
st = (wc[:, 0] - w0) / 100
print(f"   kernel span {(wc[:, 1].max() - w0) / 100:.1f} us; workgroup starts: first wave of workgroups (start < 5 us) {int((st < 5).sum())}, median start of the rest "
      f"{np.median(st[st >= 5]) if (st >= 5).any() else 0:.1f} us; workgroup duration mean {((wc[:, 1] - wc[:, 0]) / 100).mean():.2f} us")
first = st < 5
dur = (wc[:, 1] - wc[:, 0]) / 100
if first.any() and (~first).any():
    print(f"   first-round workgroups: duration mean {dur[first].mean():.1f} us (end {((wc[first, 1] - w0) / 100).mean():.1f}); later ones: start mean {st[~first].mean():.1f}, "
          f"duration mean {dur[~first].mean():.1f} us")


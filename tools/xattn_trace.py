"""Phase timeline of the fused cross-attention kernel (probe build with s_memtime stamps).
    make -C ap-adapter_amd/csrc trace      # -> exp/libtrace.so (never shipped as the product library)
    APAD_LIB_PATH=exp/libtrace.so python tools/xattn_trace.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ap_adapter_amd as A
from ap_adapter_amd import ops

dev, dt = torch.device("cuda:0"), torch.bfloat16
B2, N, Cc, H, Lt, La = int(os.environ.get("B2", "64")), 1000, 256, 8, 8, int(os.environ.get("LA", "32"))
R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
x, g, be, wq, wo, bo = R(B2, N, Cc), R(Cc), R(Cc), R(Cc, Cc, std=0.02), R(Cc, Cc, std=0.02), R(Cc, std=0.02)
k1, k2 = R(B2, Lt, Cc, std=0.3), R(B2, La, Cc, std=0.3)
v1t = torch.zeros(B2, H, 32, 32, device=dev, dtype=dt); v1t[..., :Lt].normal_(0, 0.3)
v2t = torch.zeros(B2, H, 32, ops.round_up(La, 32), device=dev, dtype=dt); v2t[..., :La].normal_(0, 0.3)
(wq_p, q_fold), wo_p = ops.xattn_pack_weight(wq, (g, be, 1e-5)), ops.xattn_pack_weight(wo)
pk1, pk2 = ops.xattn_pack_kv(k1, v1t, Lt), ops.xattn_pack_kv(k2, v2t, La)
out = torch.empty_like(x)
for _ in range(3):
    ops.fused_cross_attention(x, wq_p, wo_p, bo, pk1, Lt, H, ln=(g, be, 1e-5), kv2_packed=pk2, L2=La, scale2=0.55, out=out, q_fold=q_fold)
torch.cuda.synchronize()
ntiles = (B2 * ((N + 31) // 32) + 3) // 4
buf = (C.c_ulonglong * (ntiles * 32))()
lib = A.lib()
lib.apad_xattn_trace_read.argtypes = [C.c_void_p, C.c_int]
assert lib.apad_xattn_trace_read(buf, ntiles * 32) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(ntiles, 2, 16).astype(np.int64)
names = ["start", "LN done", "barrier1", "q-proj x4", "attn p0,1", "attn p2,3", "(unused)", "wo loaded", "barrier2", "phase3 done", "barrier3", "end"]
t0 = t[:, :, 0].min()
for w in (0, 1):
    d = np.diff(t[:, w, :12], axis=1)
    print(f"wave {'0' if w == 0 else '7'}: mean cycles per segment (s_memtime ticks)")
    for i, n in enumerate(names[1:]):
        print(f"   -> {n:12s} {d[:, i].mean():9.0f}   (min {d[:, i].min():7d} max {d[:, i].max():7d})")
    print(f"   total {(t[:, w, 11] - t[:, w, 0]).mean():9.0f}")
wc = t[:, 0, 12:14]  # wall_clock64 (100 MHz) at tile start / end
w0 = wc[:, 0].min()
order = np.argsort(wc[:, 0])
print(f"kernel span {(wc[:, 1].max() - w0) / 100:.1f} us; tile starts (us after the first): median {np.median(wc[:, 0] - w0) / 100:.1f}, "
      f"p25 {np.percentile(wc[:, 0] - w0, 25) / 100:.1f}, p75 {np.percentile(wc[:, 0] - w0, 75) / 100:.1f}, max {(wc[:, 0].max() - w0) / 100:.1f}")
dur = (wc[:, 1] - wc[:, 0]) / 100
first = (wc[:, 0] - w0) < 300
print(f"tile duration: first round ({first.sum()} tiles) mean {dur[first].mean():.1f} us, later ({(~first).sum()}) mean {dur[~first].mean():.1f} us")
print("XCC ids of tiles 0..15:", t[:16, 0, 14].tolist())

"""Stress of apad_layernorm_geglu_packed: guard rows around the output, many shapes, repeated launches, bit-equality with the row-panel launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
C = 384
g = torch.Generator(device="cpu").manual_seed(1)
R = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(dt).to(dev)
w1, b1 = R(8 * C, C, std=0.05), R(8 * C, std=0.3)
ln = (1 + 0.1 * R(C), 0.1 * R(C), 1e-5)
wp, bp = ops.geglu_pack(w1, b1)
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for M in (1, 31, 32, 33, 255, 256, 257, 2047, 2048, 4097, 16128, 16129, 20000):
        x = R(M, C)
        buf = torch.full((M + 64, 4 * C), 7.0, device=dev, dtype=dt)
        out = buf[32:32 + M]
        for _ in range(10):
            ops.layernorm_geglu_packed(x, wp, bp, ln=ln, out=out)
        torch.cuda.synchronize()
        ref = ops.fused_linear(x, w1, b1, ln=ln, act="geglu")
        okg = bool((buf[:32] == 7.0).all() and (buf[32 + M:] == 7.0).all())
        eq = torch.equal(out, ref)
        if not (okg and eq):
            bad += 1
            print(f"M={M}: guards {'ok' if okg else 'CLOBBERED'}, bit-equal {eq}, max diff {(out.float() - ref.float()).abs().max().item():.3g}")
print("stress done, failures:", bad)

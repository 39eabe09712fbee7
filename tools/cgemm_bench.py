"""Big-tile LDS-DMA kernel (csrc/cgemm.hip) vs the 128x128 tiled kernel on the compute-bound launches of the step: correctness against
fp32 torch and hipGraph-timed launches.  Runs itself twice (APAD_CGEMM=0 / 1, the knob is read once per process).
usage: python tools/cgemm_bench.py"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("CGEMM_CHILD") is None:
    for mode in ("0", "1"):
        subprocess.run([sys.executable, __file__], env=dict(os.environ, CGEMM_CHILD="1", APAD_CGEMM=mode), check=False)
    sys.exit(0)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from ap_adapter_amd import ops  # noqa: E402
from bench import time_kernel_graphed  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
print("== APAD_CGEMM =", os.environ.get("APAD_CGEMM"))
torch.manual_seed(0)
CHECK = os.environ.get("CGEMM_NOCHECK") is None
# correctness on small-but-eligible shapes (M >= 32768) incl. ragged M and borders
for B, H, W, Cin, Cout in [(9, 125, 30, 64, 128), (3, 250, 45, 128, 256)] if CHECK else []:
    x = (torch.randn(B, H * W, Cin, device=dev) * 0.5).to(dt)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).to(dt)
    b = (torch.randn(Cout, device=dev) * 0.1).to(dt)
    r = torch.randn(B, H * W, Cout, device=dev).to(dt)
    out, _, _ = ops.conv3x3(x, w.reshape(Cout, -1), b, B, H, W, residual=r)
    ref = F.conv2d(x.float().view(B, H, W, Cin).permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1)
    ref = ref.to(dt).float() + r.float().view(B, H, W, Cout).permute(0, 3, 1, 2)
    got = out.float().view(B, H, W, Cout).permute(0, 3, 1, 2)
    print(f"conv check B={B} {H}x{W} {Cin}->{Cout} (M={B*H*W}): rel err {float((got - ref).abs().max() / ref.abs().max()):.2e}")
for M, K, N in [(40000, 192, 128), (33000, 1024, 384)] if CHECK else []:
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    b = (torch.randn(N, device=dev) * 0.1).to(dt)
    out = ops.linear(x, w, b)
    ref = x.float() @ w.float().t() + b.float()
    print(f"gemm check M={M} K={K} N={N}: rel err {float((out.float() - ref).abs().max() / ref.abs().max()):.2e}")

for B, H, W, Cin, Cout, what in [(64, 250, 16, 128, 128, "4000-px level"), (64, 250, 16, 256, 128, "4000-px up"), (64, 250, 16, 384, 128, "4000-px up (384 in)"),
                                 (64, 125, 8, 128, 256, "1000-px down"), (64, 125, 8, 256, 256, "1000-px level"), (64, 125, 8, 512, 256, "1000-px up"),
                                 (64, 125, 8, 640, 256, "1000-px up (640 in)"), (64, 63, 4, 384, 384, "252-px level (M = 16128: below the default threshold)"),
                                 (64, 63, 4, 768, 384, "252-px up (768 in)"), (64, 63, 4, 1024, 384, "252-px up (1024 in)"),
                                 (64, 250, 16, 128, 8, "conv_out"), (64, 32, 2, 384, 640, "64-px down (384 in)"), (64, 32, 2, 640, 640, "64-px level"), (64, 32, 2, 1280, 640, "64-px up (1280 in)")]:
    x = (torch.randn(B, H * W, Cin, device=dev) * 0.5).to(dt)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).to(dt)
    if os.environ.get("CGEMM_ZERO"):  # all-zero operands: what the data-dependent power draw costs (MFMA clocks)
        x.zero_(), w.zero_()
    b = (torch.randn(Cout, device=dev) * 0.1).to(dt)
    out, Ho, Wo = ops.conv3x3(x, w.reshape(Cout, -1), b, B, H, W)
    ms = time_kernel_graphed(lambda: ops.conv3x3(x, w.reshape(Cout, -1), b, B, H, W, out=out))
    print(f"conv B={B:3d} {H}x{W} {Cin:4d}->{Cout:4d}  {ms * 1e3:7.2f} us  {2.0 * B * H * W * 9 * Cin * Cout / (ms * 1e-3) / 1e12:7.1f} TF/s   {what}")
for M, K, N, what in [(64000, 1024, 256, "1000-token FF2"), (64000, 256, 256, "1000-token to_out"), (16128, 1536, 384, "252-token FF2 (below threshold)"), (65536, 4096, 4096, "large square-ish")]:
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    ops.linear(x, w, out=out)
    ms = time_kernel_graphed(lambda: ops.linear(x, w, out=out))
    print(f"gemm M={M:6d} K={K:5d} N={N:5d}  {ms * 1e3:7.2f} us  {2.0 * M * K * N / (ms * 1e-3) / 1e12:7.1f} TF/s   {what}")

"""Latency-bound GEMM shapes of the 640- / 384-wide UNet levels through apad_gemm (hipGraph-timed, 20 launches per replay), with a
correctness check against fp32 torch.  (The in-workgroup split-K factor is fixed by (N, K): gemm.hip.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ap_adapter_amd import ops  # noqa: E402
from bench import time_kernel_graphed  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
SHAPES = [  # (M, K, N, residual, what)
    (2048, 640, 640, True, "640 level, one stream half: to_out / to_q"),
    (4096, 640, 640, True, "640 level, whole batch: to_out / to_q"),
    (2048, 2560, 640, True, "640 level half: FF2"),
    (4096, 2560, 640, True, "640 level: FF2"),
    (4096, 640, 1920, False, "640 level: q|k|v (row-major here)"),
    (4096, 1280, 640, False, "640 level: 1x1 shortcut"),
    (16128, 384, 384, True, "384 level: to_out"),
    (16128, 1536, 384, True, "384 level: FF2"),
    (64000, 1024, 256, True, "256 level: FF2 (128-tile)"),
]
for M, K, N, res, what in SHAPES:
    x = (torch.randn(M, K, device=dev) * 0.5).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    b = (torch.randn(N, device=dev) * 0.1).to(dt)
    r = torch.randn(M, N, device=dev).to(dt) if res else None
    out = torch.empty(M, N, device=dev, dtype=dt)
    ops.linear(x, w, b, residual=r, out=out)
    ref = x.float() @ w.float().t() + b.float()
    if res:
        ref = ref.to(dt).float() + r.float()
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    ms = time_kernel_graphed(lambda: ops.linear(x, w, b, residual=r, out=out))
    print(f"M={M:6d} K={K:5d} N={N:5d}  {ms * 1e3:7.2f} us  {2.0 * M * K * N / (ms * 1e-3) / 1e12:7.1f} TF/s  rel err {err:.2e}   {what}")

# 3x3 convolutions of the low-resolution resnets (implicit GEMM, K = 9 Cin); APAD_CONV_KG=0|2|4
import torch.nn.functional as F  # noqa: E402
print("APAD_CONV_KG =", os.environ.get("APAD_CONV_KG", "(default)"))
for B, H, W, Cin, Cout, what in [(32, 32, 2, 640, 640, "640 level, one stream half"), (64, 32, 2, 640, 640, "640 level"),
                                 (32, 32, 2, 1280, 640, "640 level up-block, half"), (64, 32, 2, 1280, 640, "640 level up-block"),
                                 (64, 63, 4, 384, 384, "384 level"), (64, 63, 4, 768, 384, "384 level up-block"),
                                 (64, 125, 8, 256, 256, "256 level (128-tile)")]:
    x = (torch.randn(B, H * W, Cin, device=dev) * 0.5).to(dt)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).to(dt)
    b = (torch.randn(Cout, device=dev) * 0.1).to(dt)
    out, Ho, Wo = ops.conv3x3(x, w.reshape(Cout, -1), b, B, H, W)
    ref = F.conv2d(x.float().view(B, H, W, Cin).permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1)
    err = float((out.float().view(B, H, W, Cout).permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max())
    ms = time_kernel_graphed(lambda: ops.conv3x3(x, w.reshape(Cout, -1), b, B, H, W, out=out))
    print(f"conv B={B:3d} {H}x{W} {Cin:4d}->{Cout:4d}  {ms * 1e3:7.2f} us  {2.0 * B * H * W * 9 * Cin * Cout / (ms * 1e-3) / 1e12:7.1f} TF/s  rel err {err:.2e}   {what}")

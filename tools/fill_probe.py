"""How the 640-wide level's GEMMs scale with the number of workgroups: if the time barely moves from 320 to 1280
workgroups, the launches are latency-bound per workgroup and a split-K schedule (more, shorter workgroups) would pay."""
import sys
import torch

sys.path.insert(0, ".")
from ap_adapter_amd import ops  # noqa: E402
from bench import time_kernel_graphed  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
for B in (32, 64, 128, 256):
    x = torch.randn(B, 64, 640, device=dev).to(dt)
    w = (torch.randn(640, 9 * 640, device=dev) * 0.02).to(dt)
    b = torch.zeros(640, device=dev, dtype=dt)
    ms = time_kernel_graphed(lambda: ops.conv3x3(x, w, b, B, 32, 2))
    fl = 2.0 * B * 64 * 640 * 9 * 640
    print(f"conv 32x2 640->640 B={B:4d} (M={B * 64:6d}, {B * 64 // 64 * 10:5d} wgs)  {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TF/s")
for K in (640, 2560):
    for B in (32, 64, 128, 256):
        M = B * 64
        x = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(640, K, device=dev) * 0.02).to(dt)
        r = torch.randn(M, 640, device=dev).to(dt)
        ms = time_kernel_graphed(lambda: ops.linear(x, w, None, residual=r))
        print(f"gemm N=640 K={K:5d} M={M:6d} ({M // 64 * 10:5d} wgs)  {ms * 1e3:7.1f} us  {2.0 * M * 640 * K / ms / 1e9:7.1f} TF/s")

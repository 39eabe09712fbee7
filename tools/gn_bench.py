"""GroupNorm(32) + SiLU launch times at the UNet's shapes (20 launches inside a hipGraph; APAD_LIB_PATH selects a variant library).
usage: python tools/gn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops
from bench import time_kernel_graphed
dev = torch.device("cuda:0"); dt = torch.bfloat16
for B, HW, C in [(64, 4000, 128), (32, 4000, 128), (64, 4000, 256), (64, 1000, 256), (64, 1000, 512), (64, 252, 384), (64, 64, 640)]:
    x = torch.randn(B, HW, C, device=dev).to(dt)
    g, b = torch.randn(C, device=dev).to(dt), torch.randn(C, device=dev).to(dt)
    ms = time_kernel_graphed(lambda: ops.group_norm(x, g, b, 32, 1e-5, silu=True))
    print(f"group_norm B={B} HW={HW} C={C}: {ms*1e3:7.1f} us  {2*x.numel()*2/ms/1e9:6.2f} TB/s (read + write)")

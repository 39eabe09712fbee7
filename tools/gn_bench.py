"""GroupNorm(32) launch times at the UNet's shapes, timed as 20 launches inside a hipGraph (APAD_GN_ONEPASS_MAXHW selects
the one-pass kernel's envelope: 0 = always the two-pass schedule)."""
import os
import sys
import torch

sys.path.insert(0, ".")
from ap_adapter_amd import ops  # noqa: E402
from bench import time_kernel_graphed  # noqa: E402

dev = torch.device("cuda:0")
print("APAD_GN_ONEPASS_MAXHW =", os.environ.get("APAD_GN_ONEPASS_MAXHW", "(default)"))
for B, HW, C in [(32, 64, 640), (32, 64, 1280), (64, 252, 384), (64, 252, 1024), (64, 1000, 256), (64, 1000, 512), (64, 1000, 768)]:
    x = torch.randn(B, HW, C, device=dev).to(torch.bfloat16)
    g, b = torch.ones(C, device=dev, dtype=torch.bfloat16), torch.zeros(C, device=dev, dtype=torch.bfloat16)
    out = torch.empty_like(x)
    for silu in (False, True):
        ms = time_kernel_graphed(lambda: ops.group_norm(x, g, b, 32, 1e-5, silu=silu, out=out))
        print(f"B={B:3d} HW={HW:5d} C={C:5d} silu={int(silu)}  {ms * 1e3:7.1f} us   {2 * x.numel() * 2 / ms / 1e6:7.0f} GB/s")

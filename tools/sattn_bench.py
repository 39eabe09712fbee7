"""apad_self_attention_fused at the two routed levels (B' = 64), hipGraph-timed; with APAD_LIB_PATH an ablation build of attention.hip
(tools/ab_build.sh <tag> attention.hip -DSF_ABL=<bits>).   usage: python tools/sattn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ap_adapter_amd import ops
from mlp_bench import timeit  # noqa: E402  (same directory)
dev, dt = torch.device("cuda:0"), torch.bfloat16
R = lambda *s, std=1.0: (torch.randn(*s, device=dev) * std).to(dt)
for N, C in ((1000, 256), (252, 384)):
    x = R(64, N, C); g = 1 + 0.1 * R(C); be = 0.1 * R(C)
    wq, wk, wv = R(C, C, std=0.05), R(C, C, std=0.05), R(C, C, std=0.05)
    pk, cs = ops.sattn_pack(wq, wk, wv, (g, be, 1e-5), 8)
    out = torch.empty_like(x)
    us = timeit(lambda: ops.self_attention_fused(x, pk, cs, 8, 1e-5, out=out))
    fl = (4.0 * N * N * C + 6.0 * N * C * C) * 64
    print(f"sattn_fused N={N} C={C}: {us:7.1f} us  {fl / us / 1e6:6.0f} TF/s   lib={os.environ.get('APAD_LIB_PATH', 'product')}", flush=True)

"""Self-attention of the 1000-token level (64 x 8 heads x 1000 x 1000, d = 32) through apad_attention: classic two-tile kernel vs the
direct form (q pre-scaled).  Isolated hipGraph timing on operands produced the way the model produces them (LayerNorm-ed
activations through N(0, 0.02^2) q|k|v weights).  APAD_ATTN_DIRECT=0|1 is read once per process; run it twice for the A/B."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ap_adapter_amd import ops  # noqa: E402
from bench import time_kernel_graphed  # noqa: E402

dev, dtype = torch.device("cuda:0"), torch.bfloat16
B2, N, C, heads = 64, 1000, 256, 8
x = torch.randn(B2, N, C, device=dev).to(dtype)
g_, b_ = torch.ones(C, device=dev, dtype=dtype), torch.zeros(C, device=dev, dtype=dtype)
w = (torch.randn(3 * C, C, device=dev) * 0.02)
for pre in (False, True):
    ww = w.clone()
    if pre:
        ww[:C] *= math.log2(math.e) / math.sqrt(C // heads)
    ww = ww.to(dtype)
    q = torch.empty(B2, N, C, device=dev, dtype=dtype)
    k = torch.empty_like(q)
    vt = torch.zeros(B2, heads, C // heads, ops.round_up(N, 32), device=dev, dtype=dtype)
    ops.rowpanel(x.reshape(-1, C), ww, [(q, None, C, "row"), (k, None, C, "row"), (vt, None, C, "vt")], ln=(g_, b_, 1e-5),
                 vt_geom=(heads, C // heads, N, vt.shape[-1]))
    out = torch.empty_like(q)
    ms = time_kernel_graphed(lambda: ops.attention(q, k, vt, N, heads, out=out, q_prescaled=pre))
    print(f"APAD_ATTN_DIRECT={os.environ.get('APAD_ATTN_DIRECT', '1')} q_prescaled={pre}: {ms * 1e3:.1f} us  "
          f"{4.0 * N * N * C * B2 / (ms * 1e-3) / 1e12:.0f} TF/s  finite={bool(torch.isfinite(out).all())}")

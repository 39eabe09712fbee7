"""Case table + deterministic input generator shared by make_golden.py (reference run, build container)
and the parity tests (oracle / HIP path, anywhere).  Inputs come from numpy's frozen RandomState stream,
so the committed fixture only needs the expected outputs."""
import numpy as np
import torch

W_STD = 0.05


def _case(name, kind, B, N, C, heads, L, X, seed, scale=1.0, num_tokens=8, masked=False, bf16=False, ehs2d=False):
    return dict(name=name, kind=kind, B=B, N=N, C=C, heads=heads, L=L, X=X, seed=seed, scale=scale,
                num_tokens=num_tokens, masked=masked, bf16=bf16, ehs2d=ehs2d)


# kind "ip": IPAttnProcessor2_0 with L = num_tokens + La keys of width X=768.
# kind "self": AttnProcessor2_0, encoder_hidden_states=None.  kind "cross": AttnProcessor2_0 over L keys.
CASES = [
    # AudioLDM2-large geometries: (N,C) = (64,640), (252,384), (1000,256); heads 8
    _case("ip_640_La0", "ip", 2, 64, 640, 8, 8 + 0, 768, 11, scale=0.5),
    _case("ip_640_La8", "ip", 2, 64, 640, 8, 8 + 8, 768, 12, scale=0.5, bf16=True),
    _case("ip_640_La32", "ip", 2, 64, 640, 8, 8 + 32, 768, 13, scale=0.55),
    _case("ip_640_La128", "ip", 2, 64, 640, 8, 8 + 128, 768, 14, scale=0.5),
    _case("ip_640_La512", "ip", 2, 64, 640, 8, 8 + 512, 768, 15, scale=1.0),
    _case("ip_384_La32", "ip", 2, 252, 384, 8, 8 + 32, 768, 16, scale=0.55, bf16=True),
    _case("ip_384_La128", "ip", 2, 252, 384, 8, 8 + 128, 768, 17, scale=0.5),
    _case("ip_256_La8", "ip", 2, 100, 256, 8, 8 + 8, 768, 18, scale=0.5),
    _case("ip_256_La32", "ip", 2, 100, 256, 8, 8 + 32, 768, 19, scale=0.55, bf16=True),
    _case("ip_256_La512", "ip", 2, 100, 256, 8, 8 + 512, 768, 20, scale=0.5),
    _case("ip_256_full_La32", "ip", 1, 1000, 256, 8, 8 + 32, 768, 21, scale=0.55),
    # scale = 0 must reduce to the plain processor on the first 8 tokens
    _case("ip_256_scale0", "ip", 2, 100, 256, 8, 8 + 32, 768, 22, scale=0.0),
    # masked IP call: only mask column 0 survives (attention_processor.py:424-428)
    _case("ip_256_masked", "ip", 2, 100, 256, 8, 8 + 32, 768, 23, scale=0.5, masked=True),
    # un-batched condition: a 2-D encoder_hidden_states [L, 768] is unsqueezed to batch 1 (attention_processor.py:371-372)
    _case("ip_256_ehs2d", "ip", 1, 100, 256, 8, 8 + 32, 768, 25, scale=0.5, ehs2d=True),
    # other head widths the kernels must handle (d = 64, 16 heads)
    _case("ip_256_h4", "ip", 2, 100, 256, 4, 8 + 32, 768, 24, scale=0.5),
    # plain processor: self-attention and masked T5 cross-attention
    _case("self_256", "self", 2, 100, 256, 8, 100, 256, 31, bf16=True),
    _case("self_384", "self", 2, 252, 384, 8, 252, 384, 32),
    _case("self_640", "self", 2, 64, 640, 8, 64, 640, 33),
    _case("t5_640_masked", "cross", 2, 64, 640, 8, 16, 1024, 34, masked=True, bf16=True),
    _case("t5_256_masked", "cross", 2, 100, 256, 8, 16, 1024, 35, masked=True),
    _case("t5_384_nomask", "cross", 2, 252, 384, 8, 16, 1024, 36),
]
CASE_BY_NAME = {c["name"]: c for c in CASES}


def _randn(rs, *shape, std=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * std).astype(np.float32))


def make_inputs(case):
    """All tensors fp32 on CPU.  mask_bias is the additive bias [B,1,L] the UNet hands to the processor
    ((1-m)*-10000, modeling_audioldm2.py:741-747): last 4 key positions of odd batch rows are masked."""
    rs = np.random.RandomState(case["seed"])
    B, N, C, L, X = case["B"], case["N"], case["C"], case["L"], case["X"]
    t = {}
    t["hs"] = _randn(rs, B, N, C)
    t["ehs"] = None if case["kind"] == "self" else _randn(rs, B, L, X)
    if case.get("ehs2d"):
        t["ehs"] = t["ehs"][0]  # [L, X]
    t["wq"] = _randn(rs, C, C, std=W_STD)
    t["wk"] = _randn(rs, C, X, std=W_STD)
    t["wv"] = _randn(rs, C, X, std=W_STD)
    t["wo"] = _randn(rs, C, C, std=W_STD)
    t["bo"] = _randn(rs, C, std=0.1)
    if case["kind"] == "ip":
        t["wk_ip"] = _randn(rs, C, X, std=W_STD)
        t["wv_ip"] = _randn(rs, C, X, std=W_STD)
    t["mask_bias"] = None
    if case["masked"]:
        m = torch.ones(B, L)
        m[1::2, -4:] = 0
        t["mask_bias"] = ((1 - m) * -10000.0).unsqueeze(1)
    return t

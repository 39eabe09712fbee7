"""Generate golden vectors for the attention processors by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports /root/reference/APadapter/ap_adapter/attention_processor.py (torch-only imports), drives
``IPAttnProcessor2_0`` / ``AttnProcessor2_0`` with a minimal stand-in for diffusers' ``Attention`` module
(the attributes the processors read: attention_processor.py:227-292, :359-468) on CPU, and records the
outputs.  Inputs and weights are NOT stored: they are regenerated from ``numpy.random.RandomState(seed)``
(a frozen stream) by ``tests/golden/cases.py``, which both this script and the tests import.  Only data
(case table + expected outputs) is committed; no reference source travels.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from cases import CASES, BLOCK_CASES, R4_BLOCK_CASES, R4_CASES, R5_BLOCK_CASES, R5_CASES, R6_BLOCK_CASES, LN_EPS, make_inputs  # noqa: E402
from APadapter.ap_adapter.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0  # noqa: E402
from safetensors.torch import save_file  # noqa: E402


class StubAttention(torch.nn.Module):
    """What diffusers==0.21.2 ``Attention`` exposes to a processor on this path (SURVEY 8b 'Call')."""

    def __init__(self, C, X, heads):
        super().__init__()
        self.heads = heads
        self.to_q = torch.nn.Linear(C, C, bias=False)
        self.to_k = torch.nn.Linear(X, C, bias=False)
        self.to_v = torch.nn.Linear(X, C, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C, bias=True), torch.nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        if attention_mask is None:
            return None
        if attention_mask.shape[-1] != target_length:
            attention_mask = torch.nn.functional.pad(attention_mask, (0, target_length), value=0.0)
        if attention_mask.shape[0] < batch_size * self.heads:
            attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
        return attention_mask


def run_case(case, dtype):
    t = make_inputs(case)
    C, X, heads = case["C"], case["X"], case["heads"]
    attn = StubAttention(C, X, heads)
    with torch.no_grad():
        attn.to_q.weight.copy_(t["wq"])
        attn.to_k.weight.copy_(t["wk"])
        attn.to_v.weight.copy_(t["wv"])
        attn.to_out[0].weight.copy_(t["wo"])
        attn.to_out[0].bias.copy_(t["bo"])
    attn = attn.to(dtype)
    hs = t["hs"].to(dtype)
    ehs = None if t["ehs"] is None else t["ehs"].to(dtype)
    mask = None if t["mask_bias"] is None else t["mask_bias"].to(dtype)
    if case["kind"] == "ip":
        proc = IPAttnProcessor2_0(hidden_size=C, name="golden", cross_attention_dim=X,
                                  num_tokens=case["num_tokens"], scale=case["scale"])
        proc.to_k_ip.weight = torch.nn.Parameter(t["wk_ip"].clone())
        proc.to_v_ip.weight = torch.nn.Parameter(t["wv_ip"].clone())
        proc = proc.to(dtype)
    else:
        proc = AttnProcessor2_0()
    with torch.no_grad():
        if case.get("block"):
            # diffusers 0.21.2 BasicTransformerBlock.forward, the attn2 sub-layer: norm2 -> attn2 -> + hidden_states
            ln = torch.nn.LayerNorm(C, eps=LN_EPS)
            ln.weight.copy_(t["ln_g"])
            ln.bias.copy_(t["ln_b"])
            ln = ln.to(dtype)
            return proc(attn, ln(hs), encoder_hidden_states=ehs, attention_mask=mask) + hs
        out = proc(attn, hs, encoder_hidden_states=ehs, attention_mask=mask)
    return out


def main_r4(cases=None, fname="attn_r4.safetensors", flag="--r4"):
    """one round's additions only (attn_r4 / attn_r5.safetensors); attn_processors / attn_blocks stay byte-identical"""
    cases = R4_BLOCK_CASES + R4_CASES if cases is None else cases
    tensors = {}
    for case in cases:
        tensors[case["name"] + ".fp32"] = run_case(case, torch.float32).contiguous()
        if case.get("bf16"):
            tensors[case["name"] + ".bf16"] = run_case(case, torch.bfloat16).contiguous()
        print(case["name"], tuple(tensors[case["name"] + ".fp32"].shape))
    meta = {"generator": "tests/golden/make_golden.py " + flag, "torch": torch.__version__,
            "reference": "fundwotsai2001/AP-adapter @ 2024-10-22, APadapter/ap_adapter/attention_processor.py",
            "cases": json.dumps([c["name"] for c in cases])}
    save_file(tensors, os.path.join(HERE, fname), metadata=meta)
    print("wrote", fname, os.path.getsize(os.path.join(HERE, fname)), "bytes")


def main():
    tensors = {}
    for case in CASES:
        tensors[case["name"] + ".fp32"] = run_case(case, torch.float32).contiguous()
        if case.get("bf16"):
            tensors[case["name"] + ".bf16"] = run_case(case, torch.bfloat16).contiguous()
        print(case["name"], tuple(tensors[case["name"] + ".fp32"].shape))
    meta = {"generator": "tests/golden/make_golden.py", "torch": torch.__version__,
            "reference": "fundwotsai2001/AP-adapter @ 2024-10-22, APadapter/ap_adapter/attention_processor.py",
            "cases": json.dumps([c["name"] for c in CASES])}
    save_file(tensors, os.path.join(HERE, "attn_processors.safetensors"), metadata=meta)
    sz = os.path.getsize(os.path.join(HERE, "attn_processors.safetensors"))
    print("wrote attn_processors.safetensors", sz, "bytes")
    blocks = {}
    for case in BLOCK_CASES:
        blocks[case["name"] + ".fp32"] = run_case(case, torch.float32).contiguous()
        if case.get("bf16"):
            blocks[case["name"] + ".bf16"] = run_case(case, torch.bfloat16).contiguous()
        print(case["name"], tuple(blocks[case["name"] + ".fp32"].shape))
    meta["cases"] = json.dumps([c["name"] for c in BLOCK_CASES])
    save_file(blocks, os.path.join(HERE, "attn_blocks.safetensors"), metadata=meta)
    print("wrote attn_blocks.safetensors", os.path.getsize(os.path.join(HERE, "attn_blocks.safetensors")), "bytes")


if __name__ == "__main__":
    if "--r4" in sys.argv:
        main_r4()
        sys.exit(0)
    if "--r6" in sys.argv:
        main_r4(R6_BLOCK_CASES, "attn_r6.safetensors", "--r6")
        sys.exit(0)
    if "--r5" in sys.argv:
        main_r4(R5_BLOCK_CASES + R5_CASES, "attn_r5.safetensors", "--r5")
        sys.exit(0)
    main()


def write_copied_cross_attention_index():
    """File names + tensor shapes of the reference's copied_cross_attention/ directory (no weights): pins which 32 attn2
    sites take the adapter and their widths (SURVEY 2 row 'copied_cross_attention/')."""
    d = "/root/reference/copied_cross_attention"
    shapes = {n: list(torch.load(os.path.join(d, n), weights_only=True, map_location="cpu").shape) for n in sorted(os.listdir(d))}
    with open(os.path.join(HERE, "copied_cross_attention_index.json"), "w") as f:
        json.dump({"source": "ls /root/reference/copied_cross_attention (file names + tensor shapes only; no weights)",
                   "files": shapes}, f, indent=0)


if __name__ == "__main__":
    write_copied_cross_attention_index()


def make_task_presets(out_path=None):
    """tests/golden/task_presets.json: the reference's own ``get_config(task)`` outputs (/root/reference/config.py), imported and
    run in the build container -- what ap_adapter_amd.config.get_config is pinned against."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("refcfg", "/root/reference/config.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out_path = out_path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "task_presets.json")
    with open(out_path, "w") as f:
        json.dump({t: m.get_config(t) for t in ("timbre_transfer", "style_transfer", "accompaniment_generation", "test")}, f,
                  indent=1, sort_keys=True)


if __name__ == "__main__":
    make_task_presets()

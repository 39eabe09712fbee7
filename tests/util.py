import torch


def rel_err(out, ref):
    """max-abs error relative to the reference's max magnitude"""
    ref = ref.detach().float()
    return float((out.detach().float().cpu() - ref.detach().cpu()).abs().max() / ref.abs().max().clamp_min(1e-12))


def q(t, dtype):
    """round a fp32 CPU tensor to the storage dtype and back (so oracle and HIP path see the same operand bits)"""
    return t.to(dtype).float()


TOL = {torch.bfloat16: 2e-2, torch.float16: 3e-3}

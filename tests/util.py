import torch
import torch.utils._python_dispatch
import torch.utils._pytree


def rel_err(out, ref):
    """max-abs error relative to the reference's max magnitude"""
    ref = ref.detach().float()
    return float((out.detach().float().cpu() - ref.detach().cpu()).abs().max() / ref.abs().max().clamp_min(1e-12))


class PerOpRounding(torch.utils._python_dispatch.TorchDispatchMode):
    """Runs fp32 torch code the way a 16-bit PyTorch pipeline runs it: every aten op computes in fp32 and its fp32
    outputs are rounded to ``dtype`` (what torch's half / bfloat16 kernels do: fp32 accumulate inside the op, one
    rounding per op output).  Under this mode the fp32 oracle chain becomes the SAME-PRECISION reference of the HIP
    path: 'the reference pipeline at this storage type'."""

    def __init__(self, dtype):
        super().__init__()
        self.dtype = dtype

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        rnd = lambda t: t.to(self.dtype).float() if isinstance(t, torch.Tensor) and t.dtype == torch.float32 else t
        return torch.utils._pytree.tree_map(rnd, out)


def q(t, dtype):
    """round a fp32 CPU tensor to the storage dtype and back (so oracle and HIP path see the same operand bits)"""
    return t.to(dtype).float()


# float32 = the fp32 precision mode (exact-f32 MFMA): only the summation order differs from the oracle
TOL = {torch.bfloat16: 2e-2, torch.float16: 3e-3, torch.float32: 1e-5}


_FULL_UNET = []


def full_unet(scale):
    """a fresh fp32 CPU copy of the AudioLDM2-large UNet (718 M parameters) with the adapter installed and the synthetic init the
    full-geometry tests share (seed 100, bias_std 0.01): built once per process, deep-copied per test (construction + init is ~15 s)"""
    import copy
    import ap_adapter_amd as A
    from ap_adapter_amd.synthetic import init_synthetic_
    if not _FULL_UNET:
        u = A.AudioLDM2UNet2DConditionModel()
        A.install_ap_adapter(u, None, scale=0.55)
        init_synthetic_(u, 100, bias_std=0.01)
        _FULL_UNET.append(u)
    u = copy.deepcopy(_FULL_UNET[0])
    for p in u.attn_processors.values():
        if hasattr(p, "to_k_ip"):
            p.scale = scale
    return u

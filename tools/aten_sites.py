"""Which lines of the package launch the small ATen kernels of the step (copies, fills, cats, elementwise)?
One eager denoise step of the bench workload under a TorchDispatchMode; every ATen call that produces device work is counted by
(op, innermost ap-adapter_amd frame) with the bytes of its output.

    python tools/aten_sites.py [--batch 32] [--top 50]
"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SKIP = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.slice", "aten.select", "aten.transpose", "aten.permute", "aten.t.",
        "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.as_strided", "aten.detach", "aten.alias", "aten.split", "aten.chunk",
        "aten.unbind", "aten.empty", "aten.sym_", "aten.unflatten", "aten.flatten", "aten.narrow", "aten.lift_fresh", "aten.is_",
        "aten._local_scalar_dense", "aten.new_empty", "aten.empty_like", "aten.movedim")


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()
        self.bytes = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "ap-adapter_amd" in fr.filename or "ap_adapter_amd" in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            o = out[0] if isinstance(out, (tuple, list)) and out else out
            nb = o.numel() * o.element_size() if isinstance(o, torch.Tensor) else 0
            self.n[(name, site)] += 1
            self.bytes[(name, site)] += nb
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--la", type=int, default=32)
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    import ap_adapter_amd as A
    from ap_adapter_amd import ops
    from ap_adapter_amd.synthetic import init_synthetic_, synthetic_inputs
    dev, dtype, B = torch.device("cuda", 0), torch.bfloat16, args.batch
    with torch.device(dev):
        unet = A.AudioLDM2UNet2DConditionModel()
        A.install_ap_adapter(unet, None, scale=0.55)
    init_synthetic_(unet, 100, on_device=True)
    unet = unet.to(dev, dtype)
    inp = synthetic_inputs(B, args.la, seed=0)
    pipe = A.AudioLDM2Pipeline(unet)
    ge = pipe.assemble_condition(inp["generated_prompt_embeds"].to(dev), inp["audio_tokens"].to(dev), inp["uncond_audio_tokens"].to(dev), dtype)
    pe, am = inp["prompt_embeds"].to(dev, dtype), inp["attention_mask"].to(dev)
    H, W, Cc = 250, 16, 8
    sched = pipe.scheduler
    sched.set_timesteps(200)
    coef = sched.coef_table().to(dev)
    step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
    lat = inp["latents"].to(dev).float().permute(0, 2, 3, 1).reshape(B, H * W, Cc).contiguous()
    unet_in = lat.to(dtype)
    unet.set_kv_cache(True)
    unet.precompute_time_tables(sched.timesteps.to(dev), step_ptr)
    unet.low_res_streams = tuple(torch.cuda.Stream() for _ in range(2))
    unet.low_res_levels = 1

    def step():
        eps2 = unet.forward_nhwc(unet_in, H, W, None, ge, pe, None, am, batch_repeat=2)
        ops.cfg_ddim_step(eps2, lat, unet_in, coef, step_ptr, 9.5)
        ops.step_advance(step_ptr)

    with torch.no_grad():
        step()
        step()
        torch.cuda.synchronize()
        with Sites() as s:
            step()
        torch.cuda.synchronize()
    tot = sum(s.n.values())
    print(f"# {tot} ATen calls with device work in one eager step (batch {B}); top {args.top} by count")
    for (name, site), n in s.n.most_common(args.top):
        print(f"{n:5d}  {s.bytes[(name, site)] / n / 1e6:9.3f} MB/call  {name:38s} {site}")


if __name__ == "__main__":
    main()

"""Launch census of ONE denoise step at the bench geometry (batch 32, La 32): C-ABI calls by entry point and torch device ops by
name, counted on an eager (un-captured) step after the K/V hoist.  usage: python tools/step_census.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import ap_adapter_amd as A
from ap_adapter_amd import _lib as L, ops
from ap_adapter_amd.synthetic import init_synthetic_, synthetic_inputs

dev = torch.device("cuda:0")
dtype = torch.bfloat16
B, La = int(os.environ.get("B", "32")), 32
with torch.device(dev):
    unet = A.AudioLDM2UNet2DConditionModel()
    A.install_ap_adapter(unet, None, scale=0.55)
init_synthetic_(unet, 100, on_device=True)
unet = unet.to(dev, dtype)
inp = synthetic_inputs(B, La, seed=0)
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
if os.environ.get("LOW_RES", "1") != "0":  # bench.py's default: the two batch halves of the lowest-resolution level on two streams
    unet.low_res_streams = tuple(torch.cuda.Stream() for _ in range(2))
    unet.low_res_levels = 1


def step():
    eps2 = unet.forward_nhwc(unet_in, H, W, None, ge, pe, None, am, batch_repeat=2)
    ops.cfg_ddim_step(eps2, lat, unet_in, coef, step_ptr, 9.5)
    ops.step_advance(step_ptr)


abi = collections.Counter()
h = L.lib()


class Proxy:
    def __getattr__(self, name):
        f = getattr(h, name)
        if not name.startswith("apad_") or name in ("apad_last_error", "apad_groupnorm_workspace_bytes", "apad_xattn_packed_kv_bytes"):
            return f

        def g(*a):
            abi[name] += 1
            return f(*a)
        return g


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        base = name.split(".")[1] if name.startswith("aten.") else name
        if base not in ("view", "reshape", "detach", "alias", "select", "slice", "split", "chunk", "permute", "transpose", "unsqueeze", "squeeze",
                        "expand", "as_strided", "empty", "empty_like", "empty_strided", "unbind", "_unsafe_view", "t", "sym_size", "sym_stride",
                        "is_contiguous", "stride", "split_with_sizes", "lift_fresh", "_reshape_alias"):
            self.n[name] += 1
        return out


with torch.no_grad():
    step()  # hoists K/V, packs weights
    torch.cuda.synchronize()
    L._lib = Proxy()
    with Census() as c:
        step()
    torch.cuda.synchronize()
print("C-ABI calls per step: %d" % sum(abi.values()))
for k, v in abi.most_common():
    print(f"  {v:5d}  {k}")
print("torch device ops per step: %d" % sum(c.n.values()))
for k, v in c.n.most_common():
    print(f"  {v:5d}  {k}")

"""Adapter wiring and checkpoint I/O -- the drop-in boundary of SURVEY 8b.

``install_ap_adapter`` reproduces the reference wiring loop (/root/reference/inference.py:22-59, twin
train_apadapter_v2.py:616-664): exactly one processor per Attention, ``IPAttnProcessor2_0`` on every ``attn2`` whose
cross-attention width is 768 (the GPT-2 + audio token stream), ``AttnProcessor2_0`` everywhere else, weights assigned
from checkpoint keys ``"<attn path>.processor.to_{k,v}_ip.weight"``.
"""
import os
from typing import Dict, Optional

import torch

from .processors import AttnProcessor2_0, IPAttnProcessor2_0


def ip_layer_names(unet):
    """Names (``...attn2.processor``) of the attention sites that take the adapter, in ``attn_processors`` order.
    Equivalent to the reference's ``cross[layer_num % 8] == 768`` table (inference.py:16,37-38)."""
    names = []
    for name, module in unet.named_modules():
        if name.endswith("attn2") and hasattr(module, "to_k") and module.to_k.in_features == 768 \
                and module.to_k.in_features != module.to_q.in_features:
            names.append(name + ".processor")
    order = list(unet.attn_processors.keys())
    return sorted(names, key=order.index)


def build_processors(unet, scale=1.0, num_tokens=8, do_copy=False, copy_dir=None):
    """The dict the reference builds at inference.py:22-49 (same keys, same classes)."""
    ip = set(ip_layer_names(unet))
    procs = {}
    for name in unet.attn_processors.keys():
        if name in ip:
            block = unet.get_submodule(name[: -len(".processor")])
            procs[name] = IPAttnProcessor2_0(hidden_size=block.to_q.in_features, name=name, cross_attention_dim=768,
                                             scale=scale, num_tokens=num_tokens, do_copy=do_copy, copy_dir=copy_dir)
        else:
            procs[name] = AttnProcessor2_0()
    return procs


def install_ap_adapter(unet, state_dict: Optional[Dict[str, torch.Tensor]] = None, scale=1.0, num_tokens=8,
                       do_copy=False, copy_dir=None):
    """Build the processors, move them to the UNet's device/dtype (inference.py:47), assign checkpoint weights
    (inference.py:51-57) and register them (``unet.set_attn_processor``, inference.py:59).  Returns the dict."""
    w = unet.conv_in.weight
    procs = build_processors(unet, scale, num_tokens, do_copy, copy_dir)
    for name, p in procs.items():
        if isinstance(p, IPAttnProcessor2_0):
            p.to(device=w.device, dtype=w.dtype)
            if state_dict is not None:
                p.to_k_ip.weight = torch.nn.Parameter(state_dict[name + ".to_k_ip.weight"].to(device=w.device, dtype=w.dtype))
                p.to_v_ip.weight = torch.nn.Parameter(state_dict[name + ".to_v_ip.weight"].to(device=w.device, dtype=w.dtype))
    registered = dict(procs)
    unet.set_attn_processor(procs)  # pops the dict, like the reference
    return registered


def adapter_state_dict(unet) -> Dict[str, torch.Tensor]:
    """fp32 CPU tensors under the reference checkpoint key scheme (SURVEY 5 'Checkpoint / resume')."""
    sd = {}
    for name, p in unet.attn_processors.items():
        if hasattr(p, "to_k_ip"):
            sd[name + ".to_k_ip.weight"] = p.to_k_ip.weight.detach().float().cpu()
            sd[name + ".to_v_ip.weight"] = p.to_v_ip.weight.detach().float().cpu()
    return sd


def save_adapter(unet, path):
    torch.save(adapter_state_dict(unet), path)


def load_adapter(path, map_location="cpu"):
    return torch.load(path, map_location=map_location, weights_only=True)


def load_copied_cross_attention(proc, copy_dir=None):
    """``do_copy`` warm start (attention_processor.py:328-344): ``<copy_dir>/<name>_k.bin`` / ``_v.bin`` hold the frozen
    to_k/to_v of the same layer as pickled fp16 tensors; loaded as fp32 trainable parameters."""
    copy_dir = copy_dir or os.environ.get("APADAPTER_COPIED_CROSS_ATTENTION", "copied_cross_attention")
    for which, lin in (("k", proc.to_k_ip), ("v", proc.to_v_ip)):
        w = torch.load(os.path.join(copy_dir, f"{proc.name}_{which}.bin"), map_location="cpu", weights_only=True)
        lin.weight = torch.nn.Parameter(w.detach().to(torch.float32))
        lin.weight.requires_grad = True


def copy_frozen_kv(unet):
    """Same warm start without files: copy each adapted layer's frozen attn.to_k/to_v into to_k_ip/to_v_ip (what
    copy_weight.py:44-63 extracts offline)."""
    for name, p in unet.attn_processors.items():
        if hasattr(p, "to_k_ip"):
            attn = unet.get_submodule(name[: -len(".processor")])
            p.to_k_ip.weight = torch.nn.Parameter(attn.to_k.weight.detach().clone())
            p.to_v_ip.weight = torch.nn.Parameter(attn.to_v.weight.detach().clone())

"""Data-parallel plumbing: one process per GPU, independent clips sharded across ranks with no collective
in the denoise loop; RCCL (torch.distributed backend "nccl" on ROCm) only for the one-off weight broadcast, the final
latent gather, and -- in training -- ONE flat all-reduce of the 64 adapter gradients per optimizer step
(the reference gets this implicitly from accelerate/DDP, train_apadapter_v2.py:831-833, :958)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_clips(n_clips, rank, world):
    """clip i -> rank i mod world (seeds are per clip, so results do not depend on the rank count)."""
    return list(range(rank, n_clips, world))


def broadcast_module(module, src=0):
    """Rank ``src``'s weights to every rank as ONE flat buffer per dtype."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    by_dtype = {}
    for p in module.parameters():
        by_dtype.setdefault(p.dtype, []).append(p)
    for dtype, ps in by_dtype.items():
        flat = torch.cat([p.detach().reshape(-1) for p in ps])
        dist.broadcast(flat, src)
        off = 0
        for p in ps:
            p.data.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()


def gather_latents(latents, dst=0):
    """all_gather of the per-rank latents [b,8,250,16]; returns the rank-major concatenation on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return latents
    out = [torch.empty_like(latents) for _ in range(dist.get_world_size())]
    dist.all_gather(out, latents.contiguous())
    return torch.cat(out, dim=0)


def adapter_parameters(unet):
    ps = []
    for _, p in unet.attn_processors.items():
        if hasattr(p, "to_k_ip"):
            ps += [p.to_k_ip.weight, p.to_v_ip.weight]
    return ps


def allreduce_adapter_grads(params, average=True):
    """One all-reduce over a flat fp32 buffer of every adapter gradient (21 626 880 elements = 86.5 MB for
    AudioLDM2-large) instead of DDP's bucketed sequence; clip-norm afterwards runs on identical data on every rank."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def average_flat_gradient_(flat, micro_batches=1):
    """The training step's ONE collective: sum the flat fp32 gradient buffer of the 64 adapter tensors over
    ranks, then divide by world * micro_batches (DDP's mean over ranks x accelerate's loss / accumulation steps).
    In place; returns the divisor."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    denom = float(world * max(int(micro_batches), 1))
    if denom != 1.0:
        flat.mul_(1.0 / denom)
    return denom

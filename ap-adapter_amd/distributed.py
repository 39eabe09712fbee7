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


def launched_by_torchrun():
    """True inside a rank process (torch.distributed.run / torchrun exports RANK and WORLD_SIZE)"""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, script, argv, extra_env=None, timeout=None):
    """``python script --gpus n ...`` typed without a launcher: start the n rank processes of ONE node ourselves -- the same command the
    round driver uses (python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P script
    argv...), one rank per GPU.  The reference gets its ranks from ``accelerate launch`` (train_apadapter_v2.py:831-833); rank 0's
    stdout (the JSON line) passes through.  Returns the launcher's exit code."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


def timed_steps(run_steps, steps, world, sync, device=None):
    """The bench contract's timed region: barrier + sync, EXACTLY ``steps`` steps, sync + barrier.  Returns (MAX over ranks of the
    seconds between the two barriers -- what the job took --, [each rank's own seconds up to its sync, before the closing barrier]).
    ``sync`` = torch.cuda.synchronize on a GPU rank (a no-op for CPU stubs)."""
    import time
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run_steps(steps)
    sync()
    own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        mine = torch.tensor([dt, own], dtype=torch.float64, device=device)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        mx = mine[:1].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        return float(mx.item()), [float(t[1].item()) for t in every]
    return dt, [own]


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

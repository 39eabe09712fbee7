"""CPU, world_size 2 over gloo: the data-parallel plumbing (clip sharding, weight broadcast, one flat adapter-gradient
all-reduce, latent gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import ap_adapter_amd as A
    from ap_adapter_amd import distributed as D
    from ap_adapter_amd.synthetic import init_synthetic_
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    cfg = A.UNetConfig(block_out_channels=(32, 64, 96, 128), attention_head_dim=4, transformer_layers_per_block=1)
    u = A.AudioLDM2UNet2DConditionModel(cfg)
    A.install_ap_adapter(u, None, scale=0.5)
    init_synthetic_(u, seed=100 + rank)              # ranks start DIFFERENT ...
    D.broadcast_module(u, src=0)                     # ... and end equal to rank 0
    ref = A.AudioLDM2UNet2DConditionModel(cfg)
    A.install_ap_adapter(ref, None, scale=0.5)
    init_synthetic_(ref, seed=100)
    same = all(torch.equal(a, b) for a, b in zip(u.state_dict().values(), ref.state_dict().values()))
    params = D.adapter_parameters(u)
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    D.allreduce_adapter_grads(params)                # mean over ranks: (1+2)/2 * (i+1)
    ok_grads = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate(params))
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)   # the trainer's flat fp32 gradient buffer
    denom = D.average_flat_gradient_(flat, micro_batches=4)   # (1+2) * i / (2 ranks * 4 micro-batches)
    ok_grads = ok_grads and denom == 8.0 and torch.allclose(flat, torch.arange(10, dtype=torch.float32) * 3 / 8)
    clips = D.shard_clips(7, rank, world)
    lat = torch.full((len(clips[:3]), 2), float(rank))
    gathered = D.gather_latents(lat)
    q.put((rank, same, ok_grads, clips, gathered.tolist(), len(params)))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, same0, g0, c0, gat0, n0), (r1, same1, g1, c1, gat1, n1) = res
    assert same0 and same1 and g0 and g1
    assert c0 == [0, 2, 4, 6] and c1 == [1, 3, 5] and sorted(c0 + c1) == list(range(7))
    assert gat0 == gat1 == [[0.0, 0.0]] * 3 + [[1.0, 1.0]] * 3
    assert n0 == n1 == 32


def _sharded_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import ap_adapter_amd as A
    from ap_adapter_amd import distributed as D, sharded as S
    D.init_from_env("gloo")
    cfg = A.get_config("timbre_transfer")
    La = A.config.audio_tokens(cfg)
    clips = S.list_clips([f"a{i}.wav" for i in range(3)], cfg, 7)
    enc = lambda path, tp, fp: (torch.full((La, 768), float(int(path[1]))), torch.zeros(La, 768))
    den = lambda lat, gen, t5, mask, gs: lat * 2 + gen[lat.shape[0]:, 8, 0].reshape(-1, 1, 1, 1) + t5[lat.shape[0]:, 0, 0].reshape(-1, 1, 1, 1)
    local = S.run_sharded(clips, cfg, enc, den, batch=2, rank=rank, world=world, latent_shape=(8, 4, 16))
    allc = S.gather_clips(local, len(clips), rank, world)
    q.put((rank, sorted(local), torch.stack(allc)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_job_world2_equals_world1():
    """cfg 4: clip i -> rank i mod world, per-clip seeds, one gather at the end: the gathered latents of a 2-rank job
    are, clip for clip, those of the 1-rank job"""
    import ap_adapter_amd as A
    from ap_adapter_amd import sharded as S
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5]
    assert torch.equal(res[0][2], res[1][2])
    cfg = A.get_config("timbre_transfer")
    La = A.config.audio_tokens(cfg)
    clips = S.list_clips([f"a{i}.wav" for i in range(3)], cfg, 7)
    enc = lambda path, tp, fp: (torch.full((La, 768), float(int(path[1]))), torch.zeros(La, 768))
    den = lambda lat, gen, t5, mask, gs: lat * 2 + gen[lat.shape[0]:, 8, 0].reshape(-1, 1, 1, 1) + t5[lat.shape[0]:, 0, 0].reshape(-1, 1, 1, 1)
    one = S.gather_clips(S.run_sharded(clips, cfg, enc, den, batch=3, latent_shape=(8, 4, 16)), len(clips))
    assert torch.equal(torch.stack(one), res[0][2])


def test_single_process_is_a_noop():
    from ap_adapter_amd import distributed as D
    assert D.shard_clips(5, 0, 1) == [0, 1, 2, 3, 4]
    t = torch.ones(2, 3)
    assert D.gather_latents(t) is t
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3)
    D.allreduce_adapter_grads([p])
    assert torch.equal(p.grad, torch.ones(3))
    f = torch.full((4,), 6.0)
    assert D.average_flat_gradient_(f, micro_batches=3) == 3.0 and torch.allclose(f, torch.full((4,), 2.0))


def test_bench_gpus_n_starts_its_own_ranks_and_reports_the_max(tmp_path):
    """`python bench.py --gpus 2` typed WITHOUT a launcher (the form the round driver uses for --gpus 1) must start 2 ranks itself
    (train_apadapter_v2.py:831-833: the reference gets its ranks from accelerate).  Driven here over gloo with the stubbed step
    (rank r sleeps (1 + r/2) x 20 ms per step): both ranks join, n_gpus = 2, the step time is the MAX over ranks (rank 1's 30 ms),
    the value is the whole-job aggregate."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-step-ms", "20", "--steps", "5", "--warmup", "1",
                        "--batch", "32"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and len(line["rank_ms_per_step"]) == 2
    r0, r1 = line["rank_ms_per_step"]
    assert 19.0 < r0 < 27.0 and 29.0 < r1 < 40.0, line
    assert line["ms_per_step"] >= r1 - 0.5 and line["ms_per_step"] < 45.0  # the MAX, not rank 0's own time
    assert abs(line["value"] - 32 * 2 / (200 * line["ms_per_step"] * 1e-3)) < 1e-2 * line["value"]

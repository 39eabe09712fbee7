"""BASELINE cfg 4: a set of clips (audio prompt x text prompt) denoised data-parallel over the GPUs of one node -- the
reference's driver loop (/root/reference/inference.py:62-81: for every positive prompt, ``pipeline(audio_file=..., time_pooling,
freq_pooling, prompt, negative_prompt, guidance_scale, ...)``) turned inside out: one process per GPU, clip i -> rank
i mod world, every rank runs ``wav -> front-end -> AudioMAE -> per-clip condition -> CFG + DDIM`` over batches of its
own clips, no collective inside the loop, ONE gather of the latents at the end.  Latents are seeded per CLIP, so the
result of a clip does not depend on the number of ranks or on which other clips share its batch.

The text side (CLAP / T5 / GPT-2, SURVEY f-4: third-party encoders without weights offline) enters as precomputed
embeddings -- the reference pipeline accepts them too (pipeline_audioldm2.py:748-775); ``synthetic_text_embeddings`` stands
in for them, seeded by the prompt string.
"""
import hashlib
import os

import torch
import torch.distributed as dist

from .config import audio_tokens, get_config
from .distributed import shard_clips


def list_clips(audio_files, cfg, n_clips=None):
    """the job: clip c = (audio file c mod F, positive prompt (c div F) mod P), the driver loop's (prompt, file) pairs flattened;
    ``n_clips`` cycles / truncates the list (cfg 4: 256 clips over the 50 evaluation wavs)"""
    prompts = [p[0] if isinstance(p, (list, tuple)) else p for p in cfg["positive_text_prompt"]]
    n = n_clips if n_clips is not None else len(audio_files) * len(prompts)
    return [dict(index=c, audio=audio_files[c % len(audio_files)], prompt=prompts[(c // len(audio_files)) % len(prompts)],
                 negative_prompt=cfg["negative_text_prompt"][0]) for c in range(n)]


def _seed_of(text, salt):
    return int.from_bytes(hashlib.sha256((salt + "\0" + text).encode()).digest()[:4], "little")


def synthetic_text_embeddings(prompt, t5_len=16):
    """stand-in for encode_prompt (pipeline_audioldm2.py:272-580) of ONE prompt: GPT-2 generated embeds [8, 768], T5 embeds
    [t5_len, 1024], T5 mask [t5_len]; a function of the prompt string only"""
    g = torch.Generator().manual_seed(_seed_of(prompt, "ap-adapter"))
    gen = torch.randn(8, 768, generator=g)
    t5 = torch.randn(t5_len, 1024, generator=g)
    mask = torch.ones(t5_len, dtype=torch.long)
    mask[t5_len - (len(prompt) % 5):] = 0  # prompts of different length mask different tails (at least 12 tokens stay)
    return gen, t5, mask


def clip_latents(index, shape, seed=0):
    """initial latents of clip ``index``: a CPU generator seeded by the clip, so every rank count gives the same bits"""
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed * 1000003 + index))


def run_sharded(clips, cfg, encode_audio, denoise, batch, rank=0, world=1, latent_shape=(8, 250, 16), seed=0, device="cpu",
                text_embeddings=synthetic_text_embeddings):
    """Denoise this rank's share of ``clips`` in batches of ``batch``; returns {clip index: latents} for the local clips.

    encode_audio(path, time_pooling, freq_pooling) -> (tokens [La, 768], uncond tokens [La, 768])   (front-end + AudioMAE)
    denoise(latents [b, ...], gen [2b, 8 + La, 768], t5 [2b, L, 1024], mask [2b, L], guidance_scale) -> latents [b, ...]
    Condition layout per batch (pipeline_audioldm2.py:934-956): unconditional half first; text tokens first, audio after."""
    mine = [clips[i] for i in shard_clips(len(clips), rank, world)]
    out = {}
    audio_cache = {}
    for b0 in range(0, len(mine), batch):
        chunk = mine[b0:b0 + batch]
        pad = batch - len(chunk)  # the last batch repeats its last clip: ONE captured graph geometry per job
        chunk_p = chunk + [chunk[-1]] * pad
        pos, neg, aud, unc, t5p, t5n, mp, mn, lat = [], [], [], [], [], [], [], [], []
        for c in chunk_p:
            if c["audio"] not in audio_cache:
                audio_cache[c["audio"]] = encode_audio(c["audio"], cfg["time_pooling"], cfg["freq_pooling"])
            a, u = audio_cache[c["audio"]]
            gp, tp_, mp_ = text_embeddings(c["prompt"])
            gn, tn_, mn_ = text_embeddings(c["negative_prompt"])
            pos.append(gp); neg.append(gn); aud.append(a); unc.append(u)
            t5p.append(tp_); t5n.append(tn_); mp.append(mp_); mn.append(mn_)
            lat.append(clip_latents(c["index"], latent_shape, seed))
        dv = lambda ts: torch.stack([t.to(device) for t in ts])
        gen = torch.cat([torch.cat([dv(neg), dv(unc).to(dv(neg).dtype)], 1), torch.cat([dv(pos), dv(aud).to(dv(pos).dtype)], 1)], 0)
        t5 = torch.cat([dv(t5n), dv(t5p)], 0)
        mask = torch.cat([dv(mn), dv(mp)], 0)
        res = denoise(dv(lat), gen, t5, mask, cfg["guidance_scale"])
        for j, c in enumerate(chunk):
            out[c["index"]] = res[j].detach()
    return out


def gather_clips(local, n_clips, rank=0, world=1):
    """every rank's {clip index: latents} -> the full list in clip order on every rank (one all_gather of padded stacks:
    rank r holds clips r, r + world, ...)"""
    if world == 1 or not dist.is_initialized():
        return [local[i] for i in range(n_clips)]
    per = (n_clips + world - 1) // world
    sample = next(iter(local.values())) if local else None
    shape = [torch.zeros(8, dtype=torch.int64)]
    if sample is not None:
        shape[0][:sample.dim()] = torch.tensor(sample.shape)
    # RCCL moves device tensors only: under "nccl" everything exchanged lives on this rank's GPU, also on a rank that holds no clip
    dev = sample.device if sample is not None else (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
    shape_t = shape[0].to(dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(shape_t, op=dist.ReduceOp.MAX)  # ranks without clips learn the latent shape
    dims = [int(x) for x in shape_t.tolist() if x > 0]
    dtype = sample.dtype if sample is not None else torch.float32
    mine = torch.zeros(per, *dims, dtype=dtype, device=dev)
    for j, i in enumerate(shard_clips(n_clips, rank, world)):
        mine[j] = local[i]
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [parts[i % world][i // world] for i in range(n_clips)]

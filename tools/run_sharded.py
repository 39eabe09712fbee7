"""BASELINE cfg 4 end to end: the clips of a directory of wavs (default: seeded synthetic 10 s tones, the evaluation audio is
not redistributable) x the task's prompts, sharded over the GPUs of the node, each rank running
wav -> Kaldi fbank -> AudioMAE (fp32) -> per-clip condition -> 200-step CFG + DDIM (hipGraph captured once per rank), latents
gathered in clip order on every rank; rank 0 writes them.

    python tools/run_sharded.py --task style_transfer --clips 256 --batch 32 --steps 200 [--audio-dir DIR] [--out latents.pt]
    python tools/run_sharded.py --gpus 8 --clips 256 ...        (starts its 8 ranks itself; equivalent to the line below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_sharded.py --clips 256 ...
"""
import argparse
import glob
import json
import math
import os
import struct
import sys
import tempfile
import time
import wave

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


EVAL_RATE_MIX = ((44100, 45), (48000, 5), (16000, 2))  # sample rates of the reference's eval_audio_* wavs: 45 + 5 + 2 files


def write_synthetic_wavs(out_dir, n=8, seconds=10.0, sr=16000, rate_mix=None):
    """seeded two-partial tones with a slow envelope, 16-bit PCM: stand-ins for eval_audio_in_domain/*.wav.  rate_mix = ((rate, weight), ...):
    file i takes the sample rate of its slot in the weighted cycle (EVAL_RATE_MIX: the evaluation set's 44.1 k / 48 k / 16 k proportions),
    so the front-end's resampler is on the path as it is for the real set"""
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    # (each rate's slots spread evenly over the cycle: 16 files of EVAL_RATE_MIX already hold all three rates)
    cycle = [r for _, r in sorted(((k + 0.5) / w_, r) for r, w_ in (rate_mix or ((sr, 1),)) for k in range(w_))]
    for i in range(n):
        sr = cycle[i % len(cycle)] if rate_mix else sr
        g = torch.Generator().manual_seed(i)
        f0 = 110.0 * 2 ** (float(torch.rand(1, generator=g)) * 3)
        t = torch.arange(int(seconds * sr)) / sr
        x = 0.4 * torch.sin(2 * math.pi * f0 * t) + 0.2 * torch.sin(2 * math.pi * 2.01 * f0 * t + 0.3)
        x = x * (0.6 + 0.4 * torch.sin(2 * math.pi * 0.5 * t)) + 0.01 * torch.randn(t.shape, generator=g)
        p = os.path.join(out_dir, f"tone_{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
            w.writeframes(struct.pack("<%dh" % x.numel(), *(x.clamp(-1, 1) * 32767).to(torch.int16).tolist()))
        paths.append(p)
    return paths


def write_wav16(path, x, sr=16000):
    """mono float waveform in [-1, 1] -> 16-bit PCM (the reference writes float32 through scipy.io.wavfile, inference.py:79-81)"""
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((x.clamp(-1, 1) * 32767).to(torch.int16).cpu().numpy().tobytes())


def build_decoder(dev, dtype, small=False, seed=300):
    """latents -> mel -> waveform stages (f-4): AutoencoderKL + SpeechT5HifiGan on the HIP path, random-init weights of the real shapes"""
    import ap_adapter_amd as A
    torch.manual_seed(seed)
    vae = A.AutoencoderKL(A.VaeConfig(block_out_channels=(32, 64, 64), layers_per_block=1, norm_num_groups=8) if small else A.VaeConfig())
    voc = A.SpeechT5HifiGan(A.HifiGanConfig(upsample_initial_channel=256) if small else A.HifiGanConfig())
    return vae.to(dev, dtype).requires_grad_(False), voc.to(dev, dtype).requires_grad_(False)


def build_job(dev, dtype, task_cfg, small=False, seed=100, adapter_ckpt=None):
    """UNet + adapter + AudioMAE (fp32, the reference's type) on ``dev``; random-init weights of the real shapes unless an adapter
    checkpoint (reference key scheme) is given"""
    import ap_adapter_amd as A
    from ap_adapter_amd.synthetic import init_synthetic_
    cfg = A.UNetConfig(block_out_channels=(64, 128, 192, 256), attention_head_dim=4, norm_num_groups=16) if small else A.UNetConfig()
    with torch.device(dev):
        unet = A.AudioLDM2UNet2DConditionModel(cfg)
        sd = A.load_adapter(adapter_ckpt) if adapter_ckpt else None
        A.install_ap_adapter(unet, sd, scale=task_cfg["ap_scale"], num_tokens=8)
        mae = A.AudioMAEConditionCTPoolRand(**({"depth": 2} if small else {}))
    init_synthetic_(unet, seed, on_device=True)
    init_synthetic_(mae, seed + 1, w_std=0.02, on_device=True)
    if sd is not None:  # the checkpoint's adapter weights survive the synthetic init of the frozen part
        for n, p in unet.attn_processors.items():
            if hasattr(p, "to_k_ip"):
                p.to_k_ip.weight = torch.nn.Parameter(sd[n + ".to_k_ip.weight"].to(dev))
                p.to_v_ip.weight = torch.nn.Parameter(sd[n + ".to_v_ip.weight"].to(dev))
    unet = unet.to(dev, dtype).requires_grad_(False)
    return A.AudioLDM2Pipeline(unet, audiomae=mae.to(dev, torch.float32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="style_transfer")
    ap.add_argument("--audio-dir", default=None)
    ap.add_argument("--clips", type=int, default=None, help="number of clips (default: files x prompts)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--small", action="store_true", help="small UNet / 2-block AudioMAE (smoke runs)")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--adapter-ckpt", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--wav-dir", default=None, help="decode every clip (VAE + vocoder on the HIP path) and write 16 kHz wavs named as "
                                                     "inference.py:79 names them")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 without a launcher: this script starts its N ranks itself (one per GPU)")
    args = ap.parse_args()

    from ap_adapter_amd import distributed as D
    if args.gpus > 1 and not D.launched_by_torchrun():
        sys.exit(D.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    import ap_adapter_amd as A
    from ap_adapter_amd import sharded as S
    from ap_adapter_amd.frontend import load_mel
    rank, world, local = A.distributed.init_from_env()
    torch.cuda.set_device(local)
    dev, dtype = torch.device("cuda", local), torch.bfloat16
    cfg = A.get_config(args.task)
    if args.audio_dir:
        files = sorted(glob.glob(os.path.join(args.audio_dir, "*.wav")))
    else:
        files = write_synthetic_wavs(os.path.join(tempfile.gettempdir(), "apad_synth_wavs"), seconds=args.seconds)
    clips = S.list_clips(files, cfg, args.clips)
    pipe = build_job(dev, dtype, cfg, small=args.small, adapter_ckpt=args.adapter_ckpt)
    H = int(args.seconds / pipe.vocoder_upsample_factor) // pipe.vae_scale_factor  # latent frames: 10 s -> 250 (pipeline_audioldm2.py:872-880)

    def encode_audio(path, tp, fp):
        tok, unc = pipe.encode_audio(load_mel(path, device=dev), tp, fp)
        return tok[0], unc[0]

    def denoise(lat, gen, t5, mask, gs):
        return pipe.denoise(lat, gen, t5, mask, args.steps, gs)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    local_out = S.run_sharded(clips, cfg, encode_audio, denoise, args.batch, rank, world, latent_shape=(8, H, 16), device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_wavs = 0
    if args.wav_dir:  # each rank decodes and writes its own clips (no collective): latents / scaling_factor -> mel -> waveform
        os.makedirs(args.wav_dir, exist_ok=True)
        vae, voc = build_decoder(dev, dtype, small=args.small)
        pipe.vae, pipe.vocoder = vae, voc
        for idx in sorted(local_out):
            mel = vae.decode(local_out[idx][None].to(dev, dtype) / vae.config.scaling_factor).sample
            wav = pipe.mel_spectrogram_to_waveform(mel)[0, : int(args.seconds * 16000)]
            c = clips[idx]
            name = f"{c['prompt']}_{idx}_ip{cfg['ap_scale']}_t{cfg['time_pooling']}_f{cfg['freq_pooling']}.wav".replace("/", "_")
            write_wav16(os.path.join(args.wav_dir, name), wav)
            n_wavs += 1
    allc = S.gather_clips(local_out, len(clips), rank, world)
    if rank == 0:
        finite = all(bool(torch.isfinite(x).all()) for x in allc)
        print(json.dumps({"task": args.task, "clips": len(clips), "world": world, "batch": args.batch, "steps": args.steps,
                          "La": A.config.audio_tokens(cfg), "seconds_rank0": round(dt, 2), "clips_per_s": round(len(clips) / dt, 4),
                          "graph_captures": pipe.graph_captures, "graph_hits": pipe.graph_hits, "finite": finite,
                          "wavs_written_rank0": n_wavs}))
        if args.out:
            torch.save({"latents": torch.stack([x.cpu() for x in allc]), "clips": clips}, args.out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

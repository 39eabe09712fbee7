"""Seeded synthetic weights and inputs: no checkpoint of AudioLDM2 / AudioMAE / the adapter is
available offline, so parity tests and the benchmark run on random-init weights of the real shapes.  Everything is
generated on CPU with a seeded torch.Generator and then moved, so every device sees identical bits."""
import torch
import torch.nn as nn


def init_synthetic_(module: nn.Module, seed=100, w_std=0.02, bias_std=0.0, norm_jitter=0.0, on_device=False):
    """Linear/Conv weights ~ N(0, w_std^2); biases ~ N(0, bias_std^2) (0 for the benchmark config); norm scales
    1 (+ N(0, norm_jitter^2)), norm shifts N(0, norm_jitter^2).  ``on_device``: draw on each tensor's own device with a
    device generator (the benchmark: 718 M parameters in a second instead of a minute; the bits then depend on the device,
    so parity tests keep the CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    dev_gens = {}

    def rn(t, std):
        if std == 0.0:
            t.zero_()
        elif on_device and t.device.type != "cpu":
            dg = dev_gens.get(t.device)
            if dg is None:
                dg = dev_gens[t.device] = torch.Generator(device=t.device).manual_seed(seed)
            t.copy_(torch.randn(t.shape, generator=dg, dtype=torch.float32, device=t.device) * std)
        else:
            t.copy_(torch.randn(t.shape, generator=g, dtype=torch.float32) * std)

    # thousands of small CPU draws / scalings / copies: with the intra-op pool of a many-core host (128 threads on the GPU boxes) every one of
    # them pays the pool's wake-up -- 10.7 s for the tests' small UNet against 0.4 s on one thread, bit-identical values (the CPU generator's
    # stream does not depend on the thread count)
    n_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        _fill(module, rn, w_std, bias_std, norm_jitter)
    finally:
        torch.set_num_threads(n_threads)
    return module


def _fill(module, rn, w_std, bias_std, norm_jitter):
    with torch.no_grad():
        for _, m in module.named_modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                rn(m.weight, w_std)
                if m.bias is not None:
                    rn(m.bias, bias_std)
            elif isinstance(m, (nn.LayerNorm, nn.GroupNorm)):
                rn(m.weight, norm_jitter)
                m.weight.add_(1.0)
                rn(m.bias, norm_jitter)
        for name, p in module.named_parameters():
            if name.endswith("cls_token") or name.endswith("pos_embed"):
                rn(p, w_std)


def synthetic_inputs(batch, La, t5_len=16, seed=0, latent_hw=(250, 16), channels=8):
    """latents N(0,1) [B,8,250,16] (seed), GPT-2 embeds N(0,1) [2B,8,768] (seed+1), audio tokens N(0,1) [1,La,768] x2
    (seed+2), T5 embeds N(0,1) [2B,t5_len,1024] with the last 4 positions of odd rows masked (seed+3),
    mel N(0,0.5) [1,1024,128] (seed+4)."""
    def gen(s):
        return torch.Generator().manual_seed(s)
    H, W = latent_hw
    out = {}
    out["latents"] = torch.randn(batch, channels, H, W, generator=gen(seed))
    out["generated_prompt_embeds"] = torch.randn(2 * batch, 8, 768, generator=gen(seed + 1))
    g2 = gen(seed + 2)
    out["audio_tokens"] = torch.randn(1, La, 768, generator=g2)
    out["uncond_audio_tokens"] = torch.randn(1, La, 768, generator=g2)
    out["prompt_embeds"] = torch.randn(2 * batch, t5_len, 1024, generator=gen(seed + 3))
    mask = torch.ones(2 * batch, t5_len, dtype=torch.long)
    mask[1::2, -4:] = 0
    out["attention_mask"] = mask
    out["mel"] = torch.randn(1, 1024, 128, generator=gen(seed + 4)) * 0.5
    return out

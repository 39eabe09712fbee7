"""Oracle: CFG + DDIM denoise loop of the reference pipeline.

Follows /root/reference/pipeline/pipeline_audioldm2.py
  * prepare_latents        :724-744  (randn * init_noise_sigma, sigma = 1 for DDIM)
  * set_timesteps          :983-984
  * denoise loop           :1003-1031 (cat x2 :1006, scale_model_input = identity :1007, UNet :1010-1017,
                                       CFG :1020-1022, scheduler.step :1025)
Scheduler arithmetic is diffusers==0.21.2 DDIMScheduler (PARITY UNPINNED, not vendored), with the
AudioLDM2 scheduler_config values: beta_start 0.0015, beta_end 0.0195, scaled_linear, 1000 train steps,
clip_sample False, set_alpha_to_one False, steps_offset 1, epsilon prediction, leading spacing, eta 0.
TEST INFRASTRUCTURE ONLY.
"""
import numpy as np
import torch

SCHED = dict(beta_start=0.0015, beta_end=0.0195, num_train_timesteps=1000, steps_offset=1,
             set_alpha_to_one=False)


def alphas_cumprod(cfg=SCHED):
    betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, cfg["num_train_timesteps"],
                           dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def timesteps(num_inference_steps, cfg=SCHED):
    """'leading' spacing: arange(n) * (T // n), reversed, + steps_offset."""
    ratio = cfg["num_train_timesteps"] // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
    return torch.from_numpy(ts + cfg["steps_offset"])


def ddim_step(noise_pred, t, sample, num_inference_steps, acp=None, cfg=SCHED):
    """DDIMScheduler.step, eta=0, epsilon prediction, no clipping/thresholding."""
    acp = alphas_cumprod(cfg) if acp is None else acp
    prev_t = int(t) - cfg["num_train_timesteps"] // num_inference_steps
    a_t = acp[int(t)]
    a_prev = acp[prev_t] if prev_t >= 0 else (torch.tensor(1.0) if cfg["set_alpha_to_one"] else acp[0])
    x0 = (sample - (1 - a_t) ** 0.5 * noise_pred) / a_t ** 0.5
    direction = (1 - a_prev) ** 0.5 * noise_pred
    return a_prev ** 0.5 * x0 + direction


def cfg_combine(noise_pred, guidance_scale):
    u, c = noise_pred.chunk(2)
    return u + guidance_scale * (c - u)


def denoise_loop(unet_fn, latents, num_inference_steps, guidance_scale, callback=None):
    """unet_fn(latent_model_input [2B,...], t) -> noise_pred [2B,...].  Returns final latents and the list of
    per-step guided noise predictions (the tensor the north-star tolerance is stated on)."""
    acp = alphas_cumprod()
    preds = []
    for t in timesteps(num_inference_steps):
        x = torch.cat([latents] * 2)
        eps = cfg_combine(unet_fn(x, t), guidance_scale)
        preds.append(eps)
        latents = ddim_step(eps, t, latents, num_inference_steps, acp)
        if callback is not None:
            callback(int(t), latents)
    return latents, preds

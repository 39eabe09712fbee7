"""Oracle for the adapter's training step (SURVEY a-11).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/train_apadapter_v2.py:
  * noising           :910            noise_scheduler.add_noise        [3P diffusers DDPMScheduler, restated]
  * forward + loss    :941-954        unet(...) ; F.mse_loss(model_pred.float(), target.float(), reduction="mean")
  * backward          :957            torch autograd through oracle/unet.py (plain torch fp32, so autograd IS the oracle)
  * clip + AdamW      :975-979, :763-769   torch.nn.utils.clip_grad_norm_(max_norm) ; torch.optim.AdamW
The optimizer restatement below is pinned against torch's own clip_grad_norm_ / AdamW in tests/test_oracle_train.py.
"""
import torch
import torch.nn.functional as F

from . import unet as OU


def alphas_cumprod(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(latents, noise, timesteps, acp=None):
    acp = alphas_cumprod() if acp is None else acp
    a = acp[timesteps].sqrt().reshape(-1, 1, 1, 1)
    s = (1 - acp[timesteps]).sqrt().reshape(-1, 1, 1, 1)
    return a * latents + s * noise


def adapter_keys(sd):
    return sorted(k for k in sd if k.endswith("to_k_ip.weight") or k.endswith("to_v_ip.weight"))


def loss_and_grads(sd, cfg, procs, noisy_latents, timesteps, ehs, ehs1, mask1, target):
    """fp32 loss and {key: grad} for every adapter tensor (the only trainable ones, :665-669)."""
    sd = dict(sd)
    keys = adapter_keys(sd)
    for k in keys:
        sd[k] = sd[k].detach().clone().requires_grad_(True)
    pred = OU.unet_forward(sd, cfg, noisy_latents, timesteps, ehs, ehs1, None, mask1, procs)
    loss = F.mse_loss(pred.float(), target.float(), reduction="mean")
    grads = torch.autograd.grad(loss, [sd[k] for k in keys])
    return loss.detach(), dict(zip(keys, grads)), pred.detach()


def clip_coef(grads, max_norm):
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    return torch.clamp(max_norm / (total + 1e-6), max=1.0), total


def adamw_update(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
    """one torch.optim.AdamW step (decoupled decay), returns (p, m, v)"""
    b1, b2 = betas
    p = p * (1 - lr * weight_decay)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p = p - (lr / bc1) * m / (v.sqrt() / (bc2 ** 0.5) + eps)
    return p, m, v

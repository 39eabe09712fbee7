"""Training step of the adapter (SURVEY a-11; reference train_apadapter_v2.py:892-979, module selection :665-669,
optimizer :763-769).

What the reference does per step, and where it runs here:
  noisy = add_noise(latents, noise, t)                 (:910)   DDPM forward process, input preparation
  eps_hat = unet(noisy, t, gen_embeds, t5, mask)       (:941)   HIP forward, autograd.py where a gradient is needed
  loss = mse(eps_hat.float(), noise.float())           (:954)   apad_mse_loss_grad (fp32)
  backward                                             (:957)   HIP backward kernels, torch.autograd as the tape
  [DDP all-reduce of the 64 adapter gradients]                  ONE flat fp32 all-reduce (RCCL), distributed plumbing
  clip_grad_norm_(1.0); AdamW; zero_grad               (:975-979) apad_grad_norm + apad_adamw_step on flat buffers

The UNet is frozen (`unet.requires_grad_(False)`, :604); the trainable tensors are the 64 to_k_ip / to_v_ip weights.
They live in ONE flat fp32 master buffer (as the reference trains them in fp32, :645) with a flat working copy in the
model dtype that the forward kernels read: every nn.Parameter is a view into the working copy, and the optimizer
kernel writes master + working copy in one pass.
"""
import torch
from . import autograd as AG
from . import ops
from .distributed import adapter_parameters, average_flat_gradient_
from .scheduler import DDIMScheduler


def add_noise(latents, noise, timesteps, alphas_cumprod):
    """diffusers DDPMScheduler.add_noise (call site train_apadapter_v2.py:910): sqrt(acp_t) x0 + sqrt(1 - acp_t) eps,
    per-sample t.  Input preparation in fp32."""
    acp = alphas_cumprod.to(latents.device)[timesteps.long()].float()
    a = acp.sqrt().reshape(-1, 1, 1, 1)
    s = (1.0 - acp).sqrt().reshape(-1, 1, 1, 1)
    return a * latents.float() + s * noise.float()


class AdapterTrainer:
    def __init__(self, unet, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0,
                 gradient_accumulation_steps=1):
        self.unet = unet
        unet.requires_grad_(False)  # :604 (processors are submodules of the UNet: re-enabled below)
        self.params = adapter_parameters(unet)
        if not self.params:
            raise ValueError("no IPAttnProcessor2_0 installed: call install_ap_adapter(unet, ...) first")
        dev, dtype = self.params[0].device, self.params[0].dtype
        n = sum(p.numel() for p in self.params)
        self.master = torch.empty(n, dtype=torch.float32, device=dev)
        self.work = torch.empty(n, dtype=dtype, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets = []
        off = 0
        for p in self.params:
            self.master[off:off + p.numel()].copy_(p.detach().reshape(-1).float())
            self.work[off:off + p.numel()].copy_(p.detach().reshape(-1))
            p.data = self.work[off:off + p.numel()].view(p.shape)  # the kernels now read the flat working copy
            p.requires_grad_(True)
            self.offsets.append(off)
            off += p.numel()
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.norm_t = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = ops._reduce_ws(dev)
        self.lr, self.betas, self.weight_decay, self.eps = lr, betas, weight_decay, eps
        self.max_grad_norm = max_grad_norm
        self.accum = gradient_accumulation_steps
        self._micro = 0
        self.global_step = 0
        self.alphas_cumprod = DDIMScheduler().alphas_cumprod  # same beta schedule as the DDPM training scheduler

    # ---- one micro-batch: forward + loss + backward, gradients accumulated in the flat fp32 buffer ----
    def micro_step(self, noisy_latents, timesteps, generated_prompt_embeds, prompt_embeds, attention_mask, target):
        """Arguments as the reference passes them to the UNet (:941-948); target = the noise (epsilon prediction, :949-950).
        Returns the fp32 loss (0-dim tensor, on device)."""
        dtype = self.work.dtype
        pred = self.unet(noisy_latents.to(dtype), timesteps, encoder_hidden_states=generated_prompt_embeds.to(dtype),
                         encoder_hidden_states_1=prompt_embeds.to(dtype), encoder_attention_mask_1=attention_mask,
                         return_dict=False)[0]
        loss = AG.mse_loss(pred, target)
        loss.backward()
        for p, off in zip(self.params, self.offsets):
            if p.grad is not None:
                self.grad[off:off + p.numel()].add_(p.grad.reshape(-1))  # fp32 accumulation across micro-batches
                p.grad = None
        self._micro += 1
        return loss.detach()

    # ---- optimizer step on the accumulation boundary ----
    def optimizer_step(self):
        average_flat_gradient_(self.grad, self._micro)  # the ONE collective of the step (86.5 MB fp32 for -large)
        ops.step_advance(self.step_t)
        gn = None
        if self.max_grad_norm and self.max_grad_norm > 0:
            gn = ops.grad_norm(self.grad, out=self.norm_t, ws=self._ws)
        ops.adamw_step(self.master, self.work, self.grad, self.exp_avg, self.exp_avg_sq, gn, self.step_t, self.lr,
                       self.betas, self.eps, self.weight_decay, self.max_grad_norm)
        self.grad.zero_()
        self._micro = 0
        self.global_step += 1

    def train_step(self, latents, noise, timesteps, generated_prompt_embeds, prompt_embeds, attention_mask):
        """noise the latents, run one micro-batch, and step the optimizer on the accumulation boundary (:892-979)."""
        noisy = add_noise(latents, noise, timesteps, self.alphas_cumprod)
        loss = self.micro_step(noisy, timesteps, generated_prompt_embeds, prompt_embeds, attention_mask, noise.float())
        if self._micro >= self.accum:
            self.optimizer_step()
        return loss

    # ---- checkpoint / resume of the trainable state (reference: accelerator.save_state, :988-1011) ----
    def state_dict(self):
        return {"master": self.master.cpu(), "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(),
                "step": int(self.step_t.item()), "global_step": self.global_step}

    def load_state_dict(self, sd):
        self.master.copy_(sd["master"])
        self.work.copy_(sd["master"].to(self.work.dtype))
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_t.fill_(sd["step"])
        self.global_step = sd["global_step"]

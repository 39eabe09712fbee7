"""DDIM scheduler state for the denoise loop (diffusers==0.21.2 DDIMScheduler semantics with the AudioLDM2
scheduler_config: scaled_linear betas 0.0015..0.0195, 1000 train steps, leading spacing, steps_offset 1,
set_alpha_to_one False, epsilon prediction, eta 0; call sites /root/reference/pipeline/pipeline_audioldm2.py:983-984,
:1007, :1025).  The per-step update itself runs in apad_cfg_ddim_step from a device-resident coefficient table, which
removes the reference's per-step host<->device sync inside ``scheduler.step``."""
import numpy as np
import torch


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195, steps_offset=1,
                 set_alpha_to_one=False):
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coef_table(self):
        """[steps, 2] fp32 (c_x, c_eps) with x_prev = c_x * x + c_eps * eps, i.e. DDIM eta=0:
        x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);  x_prev = sqrt(a_p) x0 + sqrt(1-a_p) eps.  Formed in float64."""
        acp = self.alphas_cumprod.double()
        ratio = self.num_train_timesteps // self.num_inference_steps
        rows = []
        for t in self.timesteps.tolist():
            a_t = acp[t]
            p = t - ratio
            a_p = acp[p] if p >= 0 else self.final_alpha_cumprod.double()
            c_x = (a_p / a_t).sqrt()
            c_e = (1 - a_p).sqrt() - (a_p / a_t).sqrt() * (1 - a_t).sqrt()
            rows.append([float(c_x), float(c_e)])
        return torch.tensor(rows, dtype=torch.float32)

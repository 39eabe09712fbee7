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
import os

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


# ---- learning-rate schedules (train_apadapter_v2.py:809-815 -> diffusers.optimization.get_scheduler, 0.21.2) ----
SCHEDULES = ("constant", "constant_with_warmup", "linear", "cosine", "cosine_with_restarts", "polynomial", "piecewise_constant")


def get_scheduler(name, num_warmup_steps=0, num_training_steps=None, num_cycles=1, power=1.0, step_rules=None, lr_init=None, lr_end=1e-7):
    """The multiplier ``f(step)`` of diffusers' ``get_scheduler(name, optimizer, ...)`` (a torch LambdaLR there; the optimizer
    kernel here takes the learning rate as an argument, so the schedule is a plain function of the scheduler step).  Same names,
    same formulas: constant, constant_with_warmup, linear, cosine (half cosine per ``num_cycles`` of 0.5), cosine_with_restarts
    (hard restarts), polynomial (down to lr_end / lr_init, needs ``lr_init``), piecewise_constant (``step_rules`` = "1:10,0.1:20,0.01")."""
    import math
    if name not in SCHEDULES:
        raise ValueError(f"unknown lr scheduler {name!r}: one of {SCHEDULES}")
    W, T = int(num_warmup_steps), num_training_steps
    if name in ("linear", "cosine", "cosine_with_restarts", "polynomial") and T is None:
        raise ValueError(f"{name} requires `num_training_steps`, please provide that argument.")
    if name == "constant":
        return lambda step: 1.0
    if name == "constant_with_warmup":
        return lambda step: float(step) / float(max(1.0, W)) if step < W else 1.0
    if name == "piecewise_constant":
        if not step_rules:
            raise ValueError('piecewise_constant requires `step_rules`, e.g. "1:10,0.1:20,0.01" (multiplier:until-step, ..., last multiplier)')
        rules = step_rules.split(",")
        # (diffusers walks the rule dict in sorted step order)
        table = sorted((int(r.split(":")[1]), float(r.split(":")[0])) for r in rules[:-1])
        last = float(rules[-1])

        def piecewise(step):
            for end, mult in table:
                if step < end:
                    return mult
            return last
        return piecewise
    warm = lambda step: float(step) / float(max(1, W))
    if name == "linear":
        return lambda step: warm(step) if step < W else max(0.0, float(T - step) / float(max(1, T - W)))
    if name == "cosine":
        nc = 0.5  # (diffusers' get_scheduler does not forward num_cycles to the plain cosine schedule: always half a cosine)

        def cosine(step):
            if step < W:
                return warm(step)
            prog = float(step - W) / float(max(1, T - W))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(nc) * 2.0 * prog)))
        return cosine
    if name == "cosine_with_restarts":
        def restarts(step):
            if step < W:
                return warm(step)
            prog = float(step - W) / float(max(1, T - W))
            if prog >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * prog) % 1.0))))
        return restarts
    if lr_init is None or not lr_init > lr_end:
        raise ValueError(f"polynomial needs lr_init > lr_end ({lr_end})")

    def poly(step):
        if step < W:
            return warm(step)
        if step > T:
            return lr_end / lr_init
        return ((lr_init - lr_end) * (1 - (step - W) / (T - W)) ** power + lr_end) / lr_init
    return poly


# kernel form of the training step's latency-bound GEMMs (ops.set_gemm_ring; measured 60.0 -> 50.4 ms per cfg-5 step): -1 = leave the
# process setting alone
TRAIN_GEMM_RING = 2


class AdapterTrainer:
    def __init__(self, unet, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0,
                 gradient_accumulation_steps=1, loss_scale=None, lr_scheduler="constant", lr_warmup_steps=0, max_train_steps=None,
                 lr_num_cycles=1, lr_power=1.0, scheduler_steps_per_update=1, dynamic_loss_scale=None, scale_growth_interval=2000,
                 lr_step_rules=None):
        """lr_scheduler / lr_warmup_steps / max_train_steps / lr_num_cycles: the reference's arguments (train_apadapter_v2.py:125-140); like
        it, warm-up, length and cycles are multiplied by gradient_accumulation_steps before they reach the schedule (:812-814), and the
        schedule advances ``scheduler_steps_per_update`` per optimizer step -- 1 here; accelerate's prepared scheduler advances once per
        PROCESS (pass the world size to reproduce a multi-GPU reference run's curve).  dynamic_loss_scale (default: on for f16 with the
        default scale): the GradScaler behind accelerate's fp16 mode -- halve on an overflowed step, double after
        ``scale_growth_interval`` clean ones.  lr_step_rules: the rule string of ``piecewise_constant``.
        Like accelerate's prepared scheduler (which returns early while ``optimizer.step_was_skipped``), the schedule does not advance on an
        optimizer step the device skipped for a non-finite gradient norm: the skip count is read back one step late (no host sync), so the
        position is corrected on the following boundary."""
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
        # f16 needs loss scaling (the reference's fp16 mode gets a GradScaler from accelerate): d loss / d pred is ~1e-5 at full
        # geometry, subnormal in f16.  Static scale 2^16 by default, applied in fp32 inside the loss kernel and removed when a
        # gradient enters the fp32 accumulator; an overflowed step (non-finite gradient norm) is skipped on the device.  bf16 has
        # the fp32 exponent range and runs unscaled.
        self.loss_scale = float(loss_scale) if loss_scale is not None else (65536.0 if dtype == torch.float16 else 1.0)
        self.offsets = []
        off = 0
        for p in self.params:
            self.master[off:off + p.numel()].copy_(p.detach().reshape(-1).float())
            self.work[off:off + p.numel()].copy_(p.detach().reshape(-1))
            p.data = self.work[off:off + p.numel()].view(p.shape)  # the kernels now read the flat working copy
            p.requires_grad_(True)
            # weight gradients are formed in fp32 and accumulated straight into the flat buffer (autograd._Linear)
            p._apad_grad_sink = (self.grad[off:off + p.numel()].view(p.shape), 1.0 / self.loss_scale)
            self.offsets.append(off)
            off += p.numel()
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)
        self.norm_t = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = ops._reduce_ws(dev)
        self.base_lr, self.betas, self.weight_decay, self.eps = lr, betas, weight_decay, eps
        acc = gradient_accumulation_steps
        self._lr_mult = get_scheduler(lr_scheduler, lr_warmup_steps * acc, None if max_train_steps is None else max_train_steps * acc,
                                      lr_num_cycles * acc, lr_power, step_rules=lr_step_rules, lr_init=lr)
        self._sched_step, self._sched_per_update = 0, int(scheduler_steps_per_update)
        self.dynamic_loss_scale = (dtype == torch.float16 and loss_scale is None) if dynamic_loss_scale is None else bool(dynamic_loss_scale)
        self.scale_growth_interval, self._good_steps, self._scale_epoch = int(scale_growth_interval), 0, 0
        self._applied_host = torch.zeros(1, dtype=torch.int32).pin_memory() if dev.type == "cuda" else torch.zeros(1, dtype=torch.int32)
        self._applied_evt, self._applied_expect, self._applied_seen = None, 0, 0
        self.max_grad_norm = max_grad_norm
        self.accum = gradient_accumulation_steps
        self._micro = 0
        self.global_step = 0
        self.alphas_cumprod = DDIMScheduler().alphas_cumprod  # same beta schedule as the DDPM training scheduler

    # ---- one micro-batch: forward + loss + backward, gradients accumulated in the flat fp32 buffer ----
    def micro_step(self, noisy_latents, timesteps, generated_prompt_embeds, prompt_embeds, attention_mask, target, micro_batches=1):
        """Arguments as the reference passes them to the UNet (:941-948); target = the noise (epsilon prediction, :949-950).
        Returns the fp32 loss (0-dim tensor, on device).
        micro_batches = k > 1: the batch is k micro-batches of the accumulation window CONCATENATED (accumulation as batch).  The
        reference runs them one after another and averages their gradients (accelerate divides each loss by k, :549 / :951); the mean
        loss over the concatenation is that same average, so one pass at k x the batch gives the same optimizer step at a fraction
        of the launches (a batch-4 pass is launch-bound on this chip).  The pass enters the accumulator with weight k and counts k
        micro-batches; the loss returned is the mean over all of them.  Needs equal shapes (one pooling rate across the window)."""
        dtype = self.work.dtype
        k = int(micro_batches)
        if k < 1 or noisy_latents.shape[0] % k:
            raise ValueError(f"micro_batches={micro_batches} does not divide the batch of {noisy_latents.shape[0]}")
        # the step's ~2600 small GEMM launches run on one stream: the LDS-DMA ring form of the 64 x 64 tile (gemm.hip; bit-equal)
        ring0 = ops.set_gemm_ring(TRAIN_GEMM_RING) if TRAIN_GEMM_RING >= 0 else None
        try:
            pred = self.unet(noisy_latents.to(dtype), timesteps, encoder_hidden_states=generated_prompt_embeds.to(dtype),
                             encoder_hidden_states_1=prompt_embeds.to(dtype), encoder_attention_mask_1=attention_mask,
                             return_dict=False)[0]
            loss = AG.mse_loss(pred, target, self.loss_scale * k)  # (d mean / d pred is k x smaller per sample than in one micro-batch)
            loss.backward()  # adapter weight gradients land in self.grad (fp32, unscaled) through the parameters' grad sinks
        finally:
            if ring0 is not None:
                ops.set_gemm_ring(ring0)
        for p, off in zip(self.params, self.offsets):
            if p.grad is not None:  # (a gradient that reached the parameter another way, e.g. a foreign layer)
                self.grad[off:off + p.numel()].add_(p.grad.reshape(-1).float(), alpha=1.0 / self.loss_scale)
                p.grad = None
        self._micro += k
        return loss.detach()

    # ---- the same micro-batch as ONE hipGraph: the un-fused training chain is ~7k launches, host-bound when eager ----
    def capture_micro_step(self, batch, height, width, n_gen_tokens, n_t5_tokens, micro_batches=1):
        """Capture forward + loss + backward + gradient accumulation for a fixed batch geometry and return
        ``replay(noisy_latents, timesteps, generated_prompt_embeds, prompt_embeds, attention_mask, target) -> loss``.
        The autograd tape is recorded once at capture; replays re-run its kernels on new data copied into static buffers.
        Trainable state is snapshotted around the warm-up runs, so capturing does not change the training trajectory."""
        dev, dtype = self.work.device, self.work.dtype
        C_in = self.unet.config.in_channels
        st = {"noisy": torch.zeros(batch, C_in, height, width, device=dev),
              "t": torch.zeros(batch, dtype=torch.int64, device=dev),
              "gen": torch.zeros(batch, n_gen_tokens, 768, dtype=dtype, device=dev),
              "pe": torch.zeros(batch, n_t5_tokens, 1024, dtype=dtype, device=dev),
              "mask": torch.ones(batch, n_t5_tokens, device=dev),
              "target": torch.zeros(batch, C_in, height, width, device=dev)}
        run = lambda: self.micro_step(st["noisy"], st["t"], st["gen"], st["pe"], st["mask"], st["target"], micro_batches)
        grad0, micro0 = self.grad.clone(), self._micro
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up: builds the cached transposed weights, primes the allocator
                run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.grad.copy_(grad0)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = run()
        self._micro = micro0
        cap = {"graph": graph, "loss": loss, "epoch": self._scale_epoch}

        def recapture():  # the loss scale changed (dynamic_loss_scale): it is a kernel argument of the captured launches (a multi-second stall,
            # paid on the first replay after the change wherever in the accumulation window that falls)
            g0, m0 = self.grad.clone(), self._micro
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                l_ = run()
            self.grad.copy_(g0)
            self._micro = m0
            cap.update(graph=g, loss=l_, epoch=self._scale_epoch)

        def replay(noisy_latents, timesteps, generated_prompt_embeds, prompt_embeds, attention_mask, target):
            if cap["epoch"] != self._scale_epoch:
                recapture()
            graph, loss = cap["graph"], cap["loss"]
            st["noisy"].copy_(noisy_latents)
            st["t"].copy_(timesteps)
            st["gen"].copy_(generated_prompt_embeds)
            st["pe"].copy_(prompt_embeds)
            st["mask"].copy_(attention_mask)
            st["target"].copy_(target)
            graph.replay()
            self._micro += int(micro_batches)
            return loss

        replay.graph = lambda: cap["graph"]  # (the CURRENT capture: a loss-scale change re-captures, and the old graph bakes the old scale in)
        return replay

    @property
    def lr(self):
        """the learning rate the NEXT optimizer step applies (``lr_scheduler.get_last_lr()[0]``, train_apadapter_v2.py:1025)"""
        return self.base_lr * self._lr_mult(self._sched_step)

    def _set_loss_scale(self, scale):
        self.loss_scale = float(scale)
        for p, off in zip(self.params, self.offsets):
            p._apad_grad_sink = (self.grad[off:off + p.numel()].view(p.shape), 1.0 / self.loss_scale)
        self._scale_epoch += 1  # (captured micro-steps hold the scale as a kernel argument: they re-capture on their next replay)

    def _update_loss_scale(self):
        """GradScaler.update() without a host sync (``dynamic_loss_scale`` only): the device counter of APPLIED updates is copied to pinned
        memory behind every optimizer step and read one step later, when the copy has long landed -- the scale backs off, and the LR schedule
        is rewound by the skipped update, one boundary after the overflow.  (Without ``dynamic_loss_scale`` a skipped update still advances
        the schedule, like the reference's unconditional ``lr_scheduler.step()``.)"""
        if self._applied_evt is not None and self._applied_evt.query():
            applied = int(self._applied_host[0])
            skipped = (self._applied_expect - applied) - self._applied_seen
            if skipped > 0:
                self._applied_seen += skipped
                self._sched_step = max(0, self._sched_step - skipped * self._sched_per_update)  # (a skipped update does not step the schedule)
                self._good_steps = 0
                self._set_loss_scale(max(self.loss_scale * 0.5, 1.0))
            else:
                self._good_steps += 1
                if self._good_steps >= self.scale_growth_interval:
                    self._good_steps = 0
                    self._set_loss_scale(self.loss_scale * 2.0)
            self._applied_evt = None

    # ---- optimizer step on the accumulation boundary ----
    def optimizer_step(self):
        if self.dynamic_loss_scale:
            self._update_loss_scale()
        average_flat_gradient_(self.grad, self._micro)  # the ONE collective of the step (86.5 MB fp32 for -large)
        gn = None
        if (self.max_grad_norm and self.max_grad_norm > 0) or self.loss_scale != 1.0:
            gn = ops.grad_norm(self.grad, out=self.norm_t, ws=self._ws)
        ops.step_advance_if_finite(self.step_t, gn)  # an overflowed (loss-scaled f16) step is skipped entirely, on the device
        ops.adamw_step(self.master, self.work, self.grad, self.exp_avg, self.exp_avg_sq, gn, self.step_t, self.lr,
                       self.betas, self.eps, self.weight_decay, self.max_grad_norm)
        self.grad.zero_()
        for p in self.params:  # the kernel wrote the working copy through raw pointers: tell version-keyed caches (K/V hoist)
            torch.autograd.graph.increment_version(p)
        self._micro = 0
        self.global_step += 1
        self._sched_step += self._sched_per_update  # lr_scheduler.step() (:978)
        if self.dynamic_loss_scale and self._applied_evt is None and self.step_t.is_cuda:
            self._applied_expect = self.global_step
            self._applied_host.copy_(self.step_t, non_blocking=True)
            self._applied_evt = torch.cuda.Event()
            self._applied_evt.record()

    @property
    def skipped_steps(self):
        """optimizer steps skipped on the device because the (loss-scaled) gradient norm was not finite: ``global_step`` counts every
        boundary like the reference's (train_apadapter_v2.py:961-963 advances on ``sync_gradients`` whether or not the GradScaler
        skipped), the device counter only the applied updates.  Reading it synchronises; poll it at logging cadence.  With
        ``dynamic_loss_scale`` the scale halves behind an overflowed step like accelerate's GradScaler (one step late, no sync)."""
        return self.global_step - int(self.step_t.item())

    def train_step(self, latents, noise, timesteps, generated_prompt_embeds, prompt_embeds, attention_mask, micro_batches=1):
        """noise the latents, run one micro-batch (or ``micro_batches`` concatenated ones, see micro_step), and step the optimizer on
        the accumulation boundary (:892-979)."""
        noisy = add_noise(latents, noise, timesteps, self.alphas_cumprod)
        loss = self.micro_step(noisy, timesteps, generated_prompt_embeds, prompt_embeds, attention_mask, noise.float(), micro_batches)
        if self._micro >= self.accum:
            self.optimizer_step()
        return loss

    def train_batch(self, batch, vae, generator=None, num_train_timesteps=1000):
        """One iteration of the reference's loop body (:892-979) from a CollateFunction batch: mel -> VAE posterior draw x
        scaling_factor (:895-897, vae.AutoencoderKL: encoder + apad_gaussian_sample on the HIP path), noise and ONE uniform timestep
        per sample (:900-905), then ``train_step``.  ``generator``: a device torch.Generator for the three draws."""
        dev = self.master.device
        mel = batch["mel"].to(dev)
        if mel.dim() == 3:
            mel = mel.unsqueeze(1)  # [B, T, 64] -> [B, 1, T, 64]
        latents = vae.encode(mel).latent_dist.sample(generator=generator, scale=vae.config.scaling_factor)
        noise = torch.randn(latents.shape, generator=generator, device=dev, dtype=latents.dtype)
        timesteps = torch.randint(0, num_train_timesteps, (latents.shape[0],), generator=generator, device=dev).long()
        pe = batch["prompt_embeds"]
        if pe.dim() == 4:
            pe = pe.squeeze(-3)  # :917
        return self.train_step(latents, noise, timesteps, batch["generated_prompt_embeds"].to(dev), pe.to(dev), batch["attention_mask"].to(dev))

    # ---- checkpoint / resume of the trainable state (reference: accelerator.save_state, :988-1011) ----
    def state_dict(self):
        """``scheduler_step`` is derived from the SYNCHRONISED count of applied updates (``step_t.item()`` is read here anyway): with
        ``dynamic_loss_scale`` a skipped update is rewound out of the schedule one boundary late (``_update_loss_scale``), so the running
        ``_sched_step`` may still count a skip that has not been read back; without it the schedule advances on every boundary like the
        reference's ``lr_scheduler.step()`` (train_apadapter_v2.py:978)."""
        applied = int(self.step_t.item())
        sched = applied * self._sched_per_update if self.dynamic_loss_scale else self._sched_step
        return {"master": self.master.cpu(), "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(),
                "step": applied, "global_step": self.global_step, "scheduler_step": sched, "loss_scale": self.loss_scale}

    def load_state_dict(self, sd):
        self.master.copy_(sd["master"])
        self.work.copy_(sd["master"].to(self.work.dtype))
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_t.fill_(sd["step"])
        self.global_step = sd["global_step"]
        self._sched_step = sd.get("scheduler_step", sd["global_step"] * self._sched_per_update)
        if "loss_scale" in sd and sd["loss_scale"] != self.loss_scale:
            self._set_loss_scale(sd["loss_scale"])
        self._applied_evt, self._applied_seen = None, sd["global_step"] - sd["step"]


# ---------------------------------------------------------------------------------------------------------------------
# data side of the trainer ("next" row f-3): condition dropout, per-batch pooling choice, checkpoint rotation
# ---------------------------------------------------------------------------------------------------------------------
POOL_LIST = (1, 2, 4, 8)  # train_apadapter_v2.py:443


def apply_condition_dropout(prompt_texts, mels, rng):
    """train_apadapter_v2.py:446-454, one uniform draw per example: < 0.05 drop the text, < 0.10 zero the audio
    condition, < 0.15 drop both (5 % each).  ``mels`` are the AudioMAE inputs [1024, 128].  Returns new lists."""
    texts, out = list(prompt_texts), list(mels)
    for i in range(len(texts)):
        r = rng.random()
        if r < 0.05:
            texts[i] = ""
        elif r < 0.1:
            out[i] = torch.zeros_like(out[i])
        elif r < 0.15:
            texts[i] = ""
            out[i] = torch.zeros_like(out[i])
    return texts, out


class CollateFunction:
    """The reference's CollateFunction (:424-479) on this package's kernels.  ``encode_prompt(list_of_texts)`` must return
    (prompt_embeds, attention_mask, generated_prompt_embeds) -- the CLAP / T5 / GPT-2 encoders are third-party models
    outside the hot path (SURVEY f-4); the audio condition (front-end + AudioMAE + pooling) runs here."""

    def __init__(self, audiomae, encode_prompt, rng=None, device=None):
        import random
        self.model = audiomae
        self.encode_prompt = encode_prompt
        self.rng = rng or random.Random()
        self.device = device

    def __call__(self, examples):
        from .frontend import load_mel
        texts = [e["text"] for e in examples]
        mel_spect = [e["fbank"] if "fbank" in e else load_mel(e["audio_path"], device=self.device)[0] for e in examples]
        pooling_rate = self.rng.choice(POOL_LIST)  # ONE rate per batch (:443-444)
        texts, mel_spect = apply_condition_dropout(texts, mel_spect, self.rng)
        with torch.no_grad():
            prompt_embeds, attention_mask, generated = self.encode_prompt(texts)
            loa = self.model(torch.stack(mel_spect), time_pool=pooling_rate, freq_pool=pooling_rate)[0]
            generated = torch.cat((generated.to(loa.device), loa.to(generated.dtype)), dim=1)  # text tokens first (:471)
        batch = {"prompt_embeds": prompt_embeds, "attention_mask": attention_mask, "generated_prompt_embeds": generated,
                 "pooling_rate": pooling_rate}
        if all("mel" in e for e in examples):
            batch["mel"] = torch.stack([e["mel"] for e in examples]).float()
        return batch


def rotate_checkpoints(output_dir, checkpoints_total_limit):
    """:990-1006 -- before saving, keep at most ``checkpoints_total_limit - 1`` ``checkpoint-<step>`` directories
    (oldest removed first).  Returns the removed names."""
    import os
    import shutil
    if checkpoints_total_limit is None or not os.path.isdir(output_dir):
        return []
    cps = sorted((d for d in os.listdir(output_dir) if d.startswith("checkpoint")), key=lambda x: int(x.split("-")[1]))
    removed = []
    if len(cps) >= checkpoints_total_limit:
        removed = cps[0:len(cps) - checkpoints_total_limit + 1]
        for d in removed:
            shutil.rmtree(os.path.join(output_dir, d))
    return removed


def save_checkpoint(trainer, output_dir, checkpoints_total_limit=None):
    """:1008-1010 ``accelerator.save_state`` equivalent for the trainable state: adapter weights under the reference key
    scheme (fp32, from the master buffer) + optimizer state, in ``checkpoint-<global_step>``."""
    import os
    from .wiring import adapter_state_dict
    rotate_checkpoints(output_dir, checkpoints_total_limit)
    path = os.path.join(output_dir, f"checkpoint-{trainer.global_step}")
    os.makedirs(path, exist_ok=True)
    sd = adapter_state_dict(trainer.unet)
    # fp32 master values, not the rounded working copy
    names = [n for n, pr in trainer.unet.attn_processors.items() if hasattr(pr, "to_k_ip")]
    for i, n in enumerate(names):
        for j, which in enumerate(("to_k_ip", "to_v_ip")):
            p, o = trainer.params[2 * i + j], trainer.offsets[2 * i + j]
            sd[f"{n}.{which}.weight"] = trainer.master[o:o + p.numel()].view(p.shape).cpu().clone()
    torch.save(sd, os.path.join(path, "pytorch_model.bin"))
    torch.save(trainer.state_dict(), os.path.join(path, "optimizer.bin"))
    return path

"""Hot-path half of ``AudioLDM2Pipeline`` (/root/reference/pipeline/pipeline_audioldm2.py:748-1061): audio-condition
assembly (:919-956), latent preparation (:724-744), the CFG + DDIM denoise loop (:983-1031), ``output_type="latent"``
exit (:1036-1040).  The keyword surface of ``__call__`` is the reference's; the one-off [3P] stages either side of the
loop (text encoders, VAE, vocoder -- SURVEY 8 out of scope) are not rebuilt: drive it with the precomputed-embedding
arguments the reference already accepts.

MI355X-first structure of the loop:
  * K/V of all 64 cross-attention sites are projected once per call (timestep-invariant), not once per step
  * the time-embedding MLP and every resnet's time_emb_proj are tabulated for all steps before the loop
  * DDIM coefficients live in a device table; CFG combine + DDIM update are one kernel; the step index is a device
    counter -> zero host synchronisation inside the loop
  * the whole step (UNet on the duplicated batch, CFG, DDIM, counter) is captured once as a hipGraph and replayed
"""
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from .scheduler import DDIMScheduler


@dataclass
class AudioPipelineOutput:
    audios: torch.Tensor = None


class AudioLDM2Pipeline:
    vae_scale_factor = 4           # AutoencoderKL of AudioLDM2: 2 ** (len(block_out_channels) - 1)
    vocoder_model_in_dim = 64      # mel bins
    vocoder_upsample_factor = 0.01  # prod(upsample_rates) / sampling_rate = 160 / 16000

    def __init__(self, unet, scheduler: Optional[DDIMScheduler] = None, audiomae=None):
        self.unet = unet
        self.scheduler = scheduler or DDIMScheduler()
        self.audiomae = audiomae
        self._uncond_cache = {}
        self._graph = None
        self._graph_key = None
        self.last_noise_pred = None

    # ---- pieces ----
    def prepare_latents(self, batch_size, num_channels_latents, height, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor,
                 self.vocoder_model_in_dim // self.vae_scale_factor)
        if latents is None:
            # randn_tensor: generated on the generator's device (CPU generators keep seeds device-independent)
            gdev = generator.device if generator is not None else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32).to(device)
        else:
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def encode_audio(self, mel, time_pooling, freq_pooling):
        """AudioMAE over the prompt mel and over zeros_like(mel) (:928-929); the zero-mel result depends only on the
        weights and the pooling setting and is cached."""
        if self.audiomae is None:
            raise RuntimeError("this pipeline was built without an AudioMAE encoder")
        mel = mel.reshape(-1, 1024, 128)
        tokens = self.audiomae(mel, time_pool=time_pooling, freq_pool=freq_pooling)[0]
        key = (time_pooling, freq_pooling)
        if key not in self._uncond_cache:
            self._uncond_cache[key] = self.audiomae(torch.zeros_like(mel[:1]), time_pool=time_pooling,
                                                    freq_pool=freq_pooling)[0]
        return tokens, self._uncond_cache[key]

    @staticmethod
    def assemble_condition(generated_prompt_embeds, audio_tokens, uncond_audio_tokens, dtype):
        """:934-956 -- text tokens first, audio after; unconditional half first; cast to the UNet dtype."""
        num = generated_prompt_embeds.shape[0] // 2
        a = audio_tokens.to(dtype).repeat(num, 1, 1)
        u = uncond_audio_tokens.to(dtype).repeat(num, 1, 1)
        neg, pos = generated_prompt_embeds.to(dtype).chunk(2)
        return torch.cat([torch.cat([neg, u], dim=1), torch.cat([pos, a], dim=1)], dim=0).contiguous()

    # ---- the loop ----
    @torch.no_grad()
    def denoise(self, latents_nchw, generated_prompt_embeds, prompt_embeds, attention_mask, num_inference_steps,
                guidance_scale, use_graph=True, callback=None, callback_steps=1, keep_noise_pred=False):
        unet = self.unet
        dev = latents_nchw.device
        dtype = unet.conv_in.weight.dtype
        B, Cc, H, W = latents_nchw.shape
        if not guidance_scale > 1.0:
            raise NotImplementedError("the audio-conditioned path requires classifier-free guidance (:941 chunk(2))")
        sched = self.scheduler
        sched.set_timesteps(num_inference_steps)
        coef = sched.coef_table().to(dev)
        step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
        lat = latents_nchw.float().permute(0, 2, 3, 1).reshape(B, H * W, Cc).contiguous()  # fp32 master, NHWC
        unet_in = lat.to(dtype)
        gen = generated_prompt_embeds.to(dtype).contiguous()
        pe = prompt_embeds.to(dtype).contiguous()
        eps_out = torch.empty_like(lat) if keep_noise_pred else None

        unet.set_kv_cache(True)
        unet.precompute_time_tables(sched.timesteps.to(dev), step_ptr)
        if use_graph and unet.low_res_streams is None and B >= 8:
            # the 64-token section of the UNet is latency-bound at any batch: its two batch halves run on two streams
            # inside the captured step (measured -1.1 % per step at batch 32; no effect on the arithmetic of a sample)
            unet.low_res_streams = (torch.cuda.Stream(), torch.cuda.Stream())

        def step():
            eps2 = unet.forward_nhwc(unet_in, H, W, None, gen, pe, None, attention_mask, batch_repeat=2)
            ops.cfg_ddim_step(eps2, lat, unet_in, coef, step_ptr, guidance_scale, eps_out)
            ops.step_advance(step_ptr)

        try:
            if use_graph and callback is None:
                # warm-up run on a side stream (fills K/V caches and scratch buffers), then restore the state
                lat0, in0 = lat.clone(), unet_in.clone()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    step()
                torch.cuda.current_stream().wait_stream(s)
                lat.copy_(lat0)
                unet_in.copy_(in0)
                step_ptr.zero_()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    step()
                lat.copy_(lat0)
                unet_in.copy_(in0)
                step_ptr.zero_()
                self._graph = g
                for _ in range(num_inference_steps):
                    g.replay()
            else:
                for i in range(num_inference_steps):
                    step()
                    if callback is not None and i % callback_steps == 0:
                        callback(i, int(sched.timesteps[i]), lat.reshape(B, H, W, Cc).permute(0, 3, 1, 2))
        finally:
            unet.set_kv_cache(False)
            unet.clear_time_tables()
        self.last_noise_pred = None if eps_out is None else eps_out.reshape(B, H, W, Cc).permute(0, 3, 1, 2)
        return lat.reshape(B, H, W, Cc).permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def __call__(self, audio_file=None, audio_file2=None, time_pooling=8, freq_pooling=8, prompt=None,
                 audio_length_in_s=None, num_inference_steps=200, guidance_scale=7.5, negative_prompt=None,
                 num_waveforms_per_prompt=1, eta=0.0, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, generated_prompt_embeds=None, negative_generated_prompt_embeds=None,
                 attention_mask=None, negative_attention_mask=None, max_new_tokens=None, return_dict=True,
                 callback=None, callback_steps=1, cross_attention_kwargs=None, output_type="latent", mel=None,
                 use_graph=True):
        if prompt is not None or negative_prompt is not None:
            raise NotImplementedError(
                "text prompts need the CLAP/T5/GPT-2 encoders, which are outside the hot path; pass prompt_embeds, "
                "generated_prompt_embeds, attention_mask and their negative_* twins (the reference accepts them too)")
        if output_type != "latent":
            raise NotImplementedError("VAE decode + vocoder are outside the hot path; use output_type='latent'")
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not used by the reference drivers")
        for n, v in (("prompt_embeds", prompt_embeds), ("negative_prompt_embeds", negative_prompt_embeds),
                     ("generated_prompt_embeds", generated_prompt_embeds),
                     ("negative_generated_prompt_embeds", negative_generated_prompt_embeds),
                     ("attention_mask", attention_mask), ("negative_attention_mask", negative_attention_mask)):
            if v is None:
                raise ValueError(f"{n} is required when no text prompt is given")
        if audio_file is not None and mel is None:
            from .frontend import load_mel  # "next" row f-2
            mel = load_mel(audio_file)
        if audio_length_in_s is None:
            audio_length_in_s = 10.24
        height = int(audio_length_in_s / self.vocoder_upsample_factor)
        if height % self.vae_scale_factor != 0:
            height = -(-height // self.vae_scale_factor) * self.vae_scale_factor
        dev = self.unet.conv_in.weight.device
        dtype = self.unet.conv_in.weight.dtype
        batch_size = prompt_embeds.shape[0]
        rep = lambda t: t.to(dev).repeat_interleave(num_waveforms_per_prompt, dim=0)
        # encode_prompt with precomputed embeddings (:547-580): [negative; positive]
        pe = torch.cat([rep(negative_prompt_embeds), rep(prompt_embeds)])
        am = torch.cat([rep(negative_attention_mask), rep(attention_mask)])
        ge = torch.cat([rep(negative_generated_prompt_embeds), rep(generated_prompt_embeds)])
        if mel is not None:
            tokens, uncond = self.encode_audio(mel.to(dev), time_pooling, freq_pooling)
            ge = self.assemble_condition(ge, tokens, uncond, dtype)
        lat = self.prepare_latents(batch_size * num_waveforms_per_prompt, self.unet.config.in_channels, height, dtype,
                                   dev, generator, latents)
        out = self.denoise(lat, ge, pe, am, num_inference_steps, guidance_scale, use_graph=use_graph, callback=callback,
                           callback_steps=callback_steps)
        if not return_dict:
            return (out,)
        return AudioPipelineOutput(audios=out)

"""Hot-path half of ``AudioLDM2Pipeline`` (/root/reference/pipeline/pipeline_audioldm2.py:748-1061): audio-condition
assembly (:919-956), latent preparation (:724-744), the CFG + DDIM denoise loop (:983-1031), ``output_type="latent"``
exit (:1036-1040), and -- with ``vae=`` / ``vocoder=`` supplied -- the VAE decode and HiFi-GAN stages after it (:1036-1044,
SURVEY f-4, ``vae.py`` / ``vocoder.py``) and, with ``prompt_encoder=`` and the two tokenizers, ``encode_prompt`` for text prompts
(:272-580, ``text_encoders.py``).  The keyword surface of ``__call__`` is the reference's; without a prompt encoder, drive it with the
precomputed-embedding arguments the reference already accepts.

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

    def __init__(self, unet, scheduler: Optional[DDIMScheduler] = None, audiomae=None, vocoder=None, vae=None, prompt_encoder=None,
                 tokenizer=None, tokenizer_2=None):
        self.unet = unet
        self.vocoder = vocoder  # vocoder.SpeechT5HifiGan (HIP) -- mel -> waveform
        self.vae = vae          # vae.AutoencoderKL (HIP) -- latents -> mel
        self.prompt_encoder = prompt_encoder  # text_encoders.PromptEncoder (HIP): CLAP text + T5 + projection + GPT-2
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2  # the caller's CLAP (RoBERTa) / T5 tokenizers (host-side, vocab files)
        self.scheduler = scheduler or DDIMScheduler()
        self.audiomae = audiomae
        self._uncond_cache = {}
        self._graphs = {}          # captured denoise steps, keyed by geometry / steps / guidance (see denoise)
        self._side_stream = None   # ONE warm-up / capture stream per pipeline (per-stream scratch buffers are keyed on it)
        self.graph_captures = 0
        self.graph_hits = 0
        self.last_noise_pred = None

    # ---- pieces ----
    def mel_spectrogram_to_waveform(self, mel_spectrogram):
        """pipeline_audioldm2.py:583-590"""
        if self.vocoder is None:
            raise RuntimeError("this pipeline was built without a vocoder")
        if mel_spectrogram.dim() == 4:
            mel_spectrogram = mel_spectrogram.squeeze(1)
        return self.vocoder(mel_spectrogram).cpu().float()

    def _encode_text(self, texts, t5_max_length=None):
        """tokenise as encode_prompt does (:381-392 positive, :485-496 negative) and run the HIP prompt encoder on one CFG half"""
        if self.prompt_encoder is None or self.tokenizer is None or self.tokenizer_2 is None:
            raise NotImplementedError("text prompts need prompt_encoder=ap_adapter_amd.PromptEncoder(...) and the CLAP / T5 tokenizers "
                                      "(tokenizer=, tokenizer_2=); or pass prompt_embeds, generated_prompt_embeds, attention_mask and their "
                                      "negative_* twins (the reference accepts them too)")
        dev = next(self.prompt_encoder.parameters()).device
        # the first tokenizer is CLAP's RoBERTa tokenizer: always padded to its model_max_length; the second (T5) to the longest prompt,
        # or -- for the negative prompts -- to the positive prompts' length
        ct = self.tokenizer(texts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True, return_tensors="pt")
        if t5_max_length is None:
            tt = self.tokenizer_2(texts, padding=True, max_length=self.tokenizer_2.model_max_length, truncation=True, return_tensors="pt")
        else:
            tt = self.tokenizer_2(texts, padding="max_length", max_length=t5_max_length, truncation=True, return_tensors="pt")
        return self.prompt_encoder.encode(ct.input_ids.to(dev), ct.attention_mask.to(dev), tt.input_ids.to(dev), tt.attention_mask.to(dev),
                                          max_new_tokens=self._max_new_tokens)

    def encode_prompt(self, prompt, device, num_waveforms_per_prompt, do_classifier_free_guidance, negative_prompt=None, prompt_embeds=None,
                      negative_prompt_embeds=None, generated_prompt_embeds=None, negative_generated_prompt_embeds=None, attention_mask=None,
                      negative_attention_mask=None, max_new_tokens=None):
        """pipeline_audioldm2.py:272-580, same arguments and return value: (prompt_embeds = T5 states, attention_mask,
        generated_prompt_embeds = GPT-2 vectors), each repeated per waveform and -- under guidance -- stacked [negative; positive]."""
        self._max_new_tokens = max_new_tokens
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
            prompt = [prompt]
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        if prompt_embeds is None:
            prompt_embeds, attention_mask, generated_prompt_embeds = self._encode_text(prompt)
        if attention_mask is None:
            attention_mask = torch.ones(prompt_embeds.shape[:2], dtype=torch.long)
        rep = lambda t: t.to(device).repeat_interleave(num_waveforms_per_prompt, dim=0)  # == repeat(1, n, 1).view(b * n, ...) (:427-443)
        prompt_embeds, attention_mask, generated_prompt_embeds = rep(prompt_embeds), rep(attention_mask), rep(generated_prompt_embeds)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            if negative_prompt is None:
                uncond_tokens = [""] * batch_size
            elif isinstance(negative_prompt, str):
                # the reference takes [negative_prompt] here and only works at batch 1 (a longer prompt list trips over the
                # mismatched halves downstream); one negative prompt for the whole batch is the evident intent
                uncond_tokens = [negative_prompt] * batch_size
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt` has batch size {len(negative_prompt)}, but `prompt` has batch size {batch_size}")
            else:
                uncond_tokens = negative_prompt
            negative_prompt_embeds, negative_attention_mask, negative_generated_prompt_embeds = self._encode_text(
                uncond_tokens, t5_max_length=prompt_embeds.shape[1])
        if do_classifier_free_guidance:
            if negative_attention_mask is None:
                negative_attention_mask = torch.ones(negative_prompt_embeds.shape[:2], dtype=torch.long)
            prompt_embeds = torch.cat([rep(negative_prompt_embeds), prompt_embeds])
            attention_mask = torch.cat([rep(negative_attention_mask), attention_mask])
            generated_prompt_embeds = torch.cat([rep(negative_generated_prompt_embeds), generated_prompt_embeds])
        return prompt_embeds, attention_mask, generated_prompt_embeds

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
    MAX_CACHED_GRAPHS = 4  # each holds its activation pool (GBs at batch 32): least-recently-used entries are dropped

    def clear_graphs(self):
        """drop the cached hipGraphs (and the hoisted K/V they read)"""
        for key in list(self._graphs):
            self._evict(key)
        self.unet.set_kv_cache(False)

    def _evict(self, key):
        self._graphs.pop(key, None)
        self.unet.drop_kv_owner(key)

    def _weights_signature(self):
        """identity + version of everything a captured step bakes in: parameter storage (the kernels hold raw pointers, and
        re-laid-out copies are rebuilt -- as NEW tensors -- when a parameter changes) and each processor's ap_scale"""
        h = 0
        for p in self.unet.parameters():
            h = hash((h, p.data_ptr(), p._version))
        return (h, tuple(getattr(pr, "scale", None) for pr in self.unet.attn_processors.values()))

    @torch.no_grad()
    def denoise(self, latents_nchw, generated_prompt_embeds, prompt_embeds, attention_mask, num_inference_steps,
                guidance_scale, use_graph=True, callback=None, callback_steps=1, keep_noise_pred=False):
        """CFG + DDIM loop (:983-1031).  With ``use_graph`` the step is captured ONCE per (batch, token counts, steps, guidance,
        weights) and kept: later calls copy their latents / conditions into the graph's static buffers, refresh the hoisted
        K/V in place and replay -- no warm-up step, no re-capture (a sharded job runs many batches through one pipeline)."""
        unet = self.unet
        dev = latents_nchw.device
        dtype = unet.conv_in.weight.dtype
        B, Cc, H, W = latents_nchw.shape
        if not guidance_scale > 1.0:
            raise NotImplementedError("the audio-conditioned path requires classifier-free guidance (:941 chunk(2))")
        sched = self.scheduler
        sched.set_timesteps(num_inference_steps)
        graphed = use_graph and callback is None
        key = (B, Cc, H, W, tuple(generated_prompt_embeds.shape), tuple(prompt_embeds.shape),
               None if attention_mask is None else (tuple(attention_mask.shape), attention_mask.dtype), num_inference_steps,
               float(guidance_scale), dtype, bool(keep_noise_pred), str(dev))
        e = None
        if graphed:
            wsig = self._weights_signature()
            e = self._graphs.get(key)
            if e is not None and e["wsig"] != wsig:  # weights were re-assigned / trained / cast since the capture
                self._evict(key)
                e = None
        if e is not None:
            self._graphs[key] = self._graphs.pop(key)  # most recently used last
            e["lat"].copy_(latents_nchw.float().permute(0, 2, 3, 1).reshape(B, H * W, Cc))
            e["unet_in"].copy_(e["lat"])
            e["gen"].copy_(generated_prompt_embeds)
            e["pe"].copy_(prompt_embeds)
            if attention_mask is not None:
                e["mask"].copy_(attention_mask)
            e["step_ptr"].zero_()
            unet.set_kv_cache(True, clear=False)
            try:
                unet.refresh_kv_cache()  # hoisted K/V of the new conditions, recomputed into the buffers the graph reads
            finally:
                unet.set_kv_cache(False, clear=False)  # (see the end of the capture branch)
            for _ in range(num_inference_steps):
                e["graph"].replay()
            self.graph_hits += 1
        else:
            e = {"lat": latents_nchw.float().permute(0, 2, 3, 1).reshape(B, H * W, Cc).contiguous(),  # fp32 master, NHWC
                 "gen": generated_prompt_embeds.to(dtype).contiguous().clone(), "pe": prompt_embeds.to(dtype).contiguous().clone(),
                 "mask": None if attention_mask is None else attention_mask.clone(),
                 "coef": sched.coef_table().to(dev), "step_ptr": torch.zeros(1, dtype=torch.int32, device=dev)}
            e["unet_in"] = e["lat"].to(dtype).clone() if dtype == torch.float32 else e["lat"].to(dtype)
            e["eps_out"] = torch.empty_like(e["lat"]) if keep_noise_pred else None
            lat, unet_in, gen, pe, mask, coef, step_ptr, eps_out = (e[k] for k in ("lat", "unet_in", "gen", "pe", "mask", "coef", "step_ptr", "eps_out"))
            from . import processors as P_
            owner = key if graphed else ("eager", id(e))
            P_.HOIST_OWNER[0] = owner  # hoisted K/V created below belong to this call (graph: until the graph is evicted)
            unet.set_kv_cache(True, clear=False)
            unet.precompute_time_tables(sched.timesteps.to(dev), step_ptr)
            e["tables"] = unet._time_tables  # the captured step keeps reading these
            # (the 64-token section stays on the one captured stream: four sub-layer workgroups per sample fill the chip at the CFG batch
            #  -- same-box A/B 37.35 vs 37.49 ms with two half-batch streams -- and this is the configuration bench.py measures;
            #  ``unet.low_res_streams`` remains an opt-in attribute)

            def step():
                eps2 = unet.forward_nhwc(unet_in, H, W, None, gen, pe, None, mask, batch_repeat=2)
                ops.cfg_ddim_step(eps2, lat, unet_in, coef, step_ptr, guidance_scale, eps_out)
                ops.step_advance(step_ptr)

            try:
                if graphed:
                    # warm-up run on a (persistent) side stream: fills the hoisted K/V and the scratch buffers; then restore
                    lat0 = lat.clone()
                    if self._side_stream is None:
                        self._side_stream = torch.cuda.Stream()
                    s = self._side_stream
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        step()
                    torch.cuda.current_stream().wait_stream(s)
                    lat.copy_(lat0)
                    unet_in.copy_(lat0)
                    step_ptr.zero_()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s):
                        step()
                    lat.copy_(lat0)
                    unet_in.copy_(lat0)
                    step_ptr.zero_()
                    e["graph"], e["wsig"] = g, wsig
                    while len(self._graphs) >= self.MAX_CACHED_GRAPHS:
                        self._evict(next(iter(self._graphs)))
                    self._graphs[key] = e
                    self.graph_captures += 1
                    for _ in range(num_inference_steps):
                        g.replay()
                else:
                    for i in range(num_inference_steps):
                        step()
                        if callback is not None and i % callback_steps == 0:
                            callback(i, int(sched.timesteps[i]), lat.reshape(B, H, W, Cc).permute(0, 3, 1, 2))
            finally:
                P_.HOIST_OWNER[0] = None
                unet.clear_time_tables()  # (a captured step keeps reading its own tables: e["tables"])
                if key not in self._graphs:  # eager call, or a failed capture: nothing will read its hoisted K/V again
                    unet.drop_kv_owner(owner)
                # the hoist is a property of THIS call: outside it the processors recompute K/V per call like the reference's (a
                # cached graph keeps its hoisted buffers -- they are refreshed in place before each replay -- but a direct
                # unet(...) / a training step on the same UNet never enters the cache, so nothing accumulates there)
                unet.set_kv_cache(False, clear=False)
        eps_out = e["eps_out"]
        self.last_noise_pred = None if eps_out is None else eps_out.reshape(B, H, W, Cc).permute(0, 3, 1, 2).clone()
        return e["lat"].reshape(B, H, W, Cc).permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def __call__(self, audio_file=None, audio_file2=None, time_pooling=8, freq_pooling=8, prompt=None,
                 audio_length_in_s=None, num_inference_steps=200, guidance_scale=7.5, negative_prompt=None,
                 num_waveforms_per_prompt=1, eta=0.0, generator=None, latents=None, prompt_embeds=None,
                 negative_prompt_embeds=None, generated_prompt_embeds=None, negative_generated_prompt_embeds=None,
                 attention_mask=None, negative_attention_mask=None, max_new_tokens=None, return_dict=True,
                 callback=None, callback_steps=1, cross_attention_kwargs=None, output_type="np", mel=None,
                 use_graph=True):
        """Same keyword surface and defaults as the reference (pipeline_audioldm2.py:748-775, ``output_type="np"`` included): a
        pipeline built with ``vae=`` and ``vocoder=`` returns waveforms by default; ``output_type="latent"`` is the exit for a pipeline
        that holds the denoise path only."""
        if output_type != "latent" and (self.vae is None or self.vocoder is None):
            raise NotImplementedError("waveform output needs latents -> mel (vae=ap_adapter_amd.AutoencoderKL) and mel -> waveform "
                                      "(vocoder=ap_adapter_amd.SpeechT5HifiGan); or use output_type='latent'")
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not used by the reference drivers")
        if num_waveforms_per_prompt > 1 and prompt is not None and output_type != "latent":
            # pipeline_audioldm2.py:1048-1056 re-orders the candidates by CLAP text-audio similarity (score_waveforms); the CLAP audio
            # tower is outside this path, and returning them un-ranked would silently differ from the reference
            raise NotImplementedError("num_waveforms_per_prompt > 1 with text prompts needs the CLAP audio tower for score_waveforms "
                                      "(not on this path); use output_type='latent' or rank the waveforms yourself")
        if prompt is None:
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
        if prompt is not None:
            batch_size = 1 if isinstance(prompt, str) else len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        # encode_prompt (:272-580): text prompts through the HIP prompt encoder, or the precomputed embeddings; [negative; positive]
        pe, am, ge = self.encode_prompt(prompt, dev, num_waveforms_per_prompt, True, negative_prompt, prompt_embeds=prompt_embeds,
                                        negative_prompt_embeds=negative_prompt_embeds, generated_prompt_embeds=generated_prompt_embeds,
                                        negative_generated_prompt_embeds=negative_generated_prompt_embeds, attention_mask=attention_mask,
                                        negative_attention_mask=negative_attention_mask, max_new_tokens=max_new_tokens)
        if mel is not None:
            tokens, uncond = self.encode_audio(mel.to(dev), time_pooling, freq_pooling)
            ge = self.assemble_condition(ge, tokens, uncond, dtype)
        lat = self.prepare_latents(batch_size * num_waveforms_per_prompt, self.unet.config.in_channels, height, dtype,
                                   dev, generator, latents)
        out = self.denoise(lat, ge, pe, am, num_inference_steps, guidance_scale, use_graph=use_graph, callback=callback,
                           callback_steps=callback_steps)
        if output_type != "latent":  # :1036-1044
            scaling = getattr(getattr(self.vae, "config", None), "scaling_factor", 1.0)
            mel = self.vae.decode(out / scaling)
            mel = getattr(mel, "sample", mel)
            out = self.mel_spectrogram_to_waveform(mel)[:, : int(audio_length_in_s * 16000)]
            if output_type == "np":
                out = out.numpy()
        if not return_dict:
            return (out,)
        return AudioPipelineOutput(audios=out)

"""-m gpu: the mel VAE (diffusers AutoencoderKL restated: oracle/vae.py, parity unpinned -- diffusers is absent) on the HIP path:
decoder (latents -> mel, pipeline_audioldm2.py:1036-1038) and encoder + posterior draw (train_apadapter_v2.py:895-897), plus the two
kernels added for it (row softmax, asymmetric-padding stride-2 convolution, gaussian draw)."""
import math

import pytest
import torch
import torch.nn.functional as F

from util import TOL, q, rel_err

pytestmark = pytest.mark.gpu


def R(*shape, seed=0, std=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * std


def _cfg_dict(cfg):
    return dict(norm_num_groups=cfg.norm_num_groups, block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block)


def _build(cfg, dtype, dev, seed=0):
    """a seeded AutoencoderKL whose parameters are exactly representable in ``dtype`` (oracle and HIP path see the same bits)"""
    import ap_adapter_amd as A
    torch.manual_seed(seed)
    vae = A.AutoencoderKL(cfg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in vae.named_parameters():
            if p.dim() > 1:
                fan = p[0].numel()
                p.copy_(q(torch.randn(p.shape, generator=g) / math.sqrt(fan), dtype))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(q(1 + 0.1 * torch.randn(p.shape, generator=g), dtype))
            else:
                p.copy_(q(0.05 * torch.randn(p.shape, generator=g), dtype))
    sd = {k: v.detach().clone().float() for k, v in vae.state_dict().items()}
    return vae.to(dev, dtype), sd


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("M,N", [(37, 48), (4000, 4000), (5, 7), (300, 1000)])
def test_softmax_rows(dev, dtype, M, N):
    from ap_adapter_amd import ops
    x = q(R(M, N, seed=1) * 4, dtype)
    ref = torch.softmax(x * 0.37, dim=-1)
    out = ops.softmax_rows(x.to(dev, dtype), 0.37)
    assert rel_err(out, ref) < TOL[dtype]
    assert float((out.float().sum(-1) - 1).abs().max()) < (1e-5 if dtype == torch.float32 else 2e-2)
    xi = x.to(dev, dtype)
    ops.softmax_rows(xi, 0.37, out=xi)  # in place, as the VAE attention runs it
    assert torch.equal(xi, out)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 20, 8, 64, 64), (1, 21, 9, 32, 64), (2, 1000, 64, 128, 128)])
def test_conv3x3_asymmetric_padding_stride2(dev, dtype, B, H, W, Cin, Cout):
    """F.pad(x, (0,1,0,1)) + Conv2d(stride 2, padding 0): the VAE encoder's down-sampler (both gather paths: Cin % 64 == 0 and not)"""
    from ap_adapter_amd import ops
    x, w, b = q(R(B, Cin, H, W, seed=2), dtype), q(R(Cout, Cin, 3, 3, seed=3, std=0.05), dtype), q(R(Cout, seed=4, std=0.1), dtype)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    D = lambda t: t.to(dev, dtype)
    out, Ho, Wo = ops.conv3x3(D(x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous()), D(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()), D(b),
                              B, H, W, stride=2, asym_pad=True)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    assert rel_err(out.view(B, Ho, Wo, Cout).permute(0, 3, 1, 2), ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gaussian_sample(dev, dtype):
    from ap_adapter_amd import ops
    m = q(R(1000, 16, seed=5) * 3, dtype)
    m[0, 8:] = 50.0   # clamped to 20
    m[1, 8:] = -50.0  # clamped to -30
    n = q(R(1000, 8, seed=6), dtype)
    ref = (m[:, :8] + torch.exp(0.5 * m[:, 8:].clamp(-30, 20)) * n) * 0.41
    out = ops.gaussian_sample(m.to(dev, dtype), n.to(dev, dtype), 0.41)
    assert rel_err(out, ref) < TOL[dtype]


SMALL = dict(block_out_channels=(32, 64, 64), layers_per_block=1, norm_num_groups=8)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
def test_vae_decode_small(dev, dtype, tol):
    """whole decoder against the oracle on a reduced geometry; 16-bit tolerances: ~20 rounded layers deep"""
    import ap_adapter_amd as A
    from oracle import vae as O
    cfg = A.VaeConfig(**SMALL)
    vae, sd = _build(cfg, dtype, dev)
    z = q(R(2, 8, 12, 4, seed=7), dtype)
    ref = O.decode(sd, _cfg_dict(cfg), z)
    out = vae.decode(z.to(dev, dtype)).sample
    assert out.shape == ref.shape == (2, 1, 48, 16)
    assert rel_err(out, ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 6e-2)])
def test_vae_encode_small(dev, dtype, tol):
    import ap_adapter_amd as A
    from oracle import vae as O
    cfg = A.VaeConfig(**SMALL)
    vae, sd = _build(cfg, dtype, dev, seed=3)
    x = q(R(2, 1, 48, 16, seed=8), dtype)
    noise = q(R(2, 8, 12, 4, seed=9), dtype)
    mean, logvar = O.encode_moments(sd, _cfg_dict(cfg), x)
    dist = vae.encode(x.to(dev, dtype)).latent_dist
    assert dist.mean.shape == mean.shape == (2, 8, 12, 4)
    assert rel_err(dist.mean, mean) < tol
    ref = O.encode_sample(sd, _cfg_dict(cfg), x, noise) * cfg.scaling_factor
    out = dist.sample(noise=noise.to(dev, dtype), scale=cfg.scaling_factor)
    assert rel_err(out, ref) < tol
    # an un-seeded draw has the posterior's spread around the mean
    s = dist.sample(generator=torch.Generator(device=dev).manual_seed(1))
    assert s.shape == mean.shape and float((s.float().cpu() - mean).abs().max()) > 0


def test_vae_decode_audioldm2_geometry_fp32(dev):
    """the AudioLDM2 VAE geometry (8 x 250 x 16 latents -> 1 x 1000 x 64 mel; mid-block attention over 4000 pixels, d = 512) in the fp32
    mode against the oracle; then the 16-bit mode against the same oracle output"""
    import ap_adapter_amd as A
    from oracle import vae as O
    cfg = A.VaeConfig()
    vae, sd = _build(cfg, torch.bfloat16, dev, seed=5)  # bf16-representable weights, shared by both modes
    z = q(R(1, 8, 250, 16, seed=10), torch.bfloat16)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    ref = O.decode(sd, _cfg_dict(cfg), z)
    assert ref.shape == (1, 1, 1000, 64)
    out16 = vae.decode(z.to(dev, torch.bfloat16)).sample
    assert rel_err(out16, ref) < 8e-2
    out32 = vae.float().decode(z.to(dev)).sample
    assert rel_err(out32, ref) < 5e-5


def test_pipeline_latents_to_waveform(dev):
    """output_type='np' with vae= and vocoder=: latents / scaling_factor -> AutoencoderKL.decode -> SpeechT5HifiGan, cropped to the
    requested length (pipeline_audioldm2.py:1036-1044), equal to running the two oracles on the pipeline's own latents"""
    import ap_adapter_amd as A
    from ap_adapter_amd import synthetic
    from oracle import vae as OV, vocoder as OH
    dtype = torch.float32
    ucfg = A.UNetConfig(block_out_channels=(64, 128, 192, 256), attention_head_dim=4, norm_num_groups=16)
    unet = A.AudioLDM2UNet2DConditionModel(ucfg)
    A.install_ap_adapter(unet, None, scale=0.5)
    synthetic.init_synthetic_(unet, 100, w_std=0.05, bias_std=0.02, norm_jitter=0.1)
    unet = unet.to(dev, dtype)
    vcfg = A.VaeConfig(**SMALL)
    vae, vsd = _build(vcfg, dtype, dev, seed=11)
    hcfg = A.HifiGanConfig(upsample_initial_channel=256, upsample_rates=(5, 4, 2, 2, 2), upsample_kernel_sizes=(16, 16, 8, 4, 4))
    torch.manual_seed(2)
    voc = A.SpeechT5HifiGan(hcfg)
    hsd = {k: v.detach().clone().float() for k, v in voc.state_dict().items()}
    voc = voc.to(dev, dtype)
    pipe = A.AudioLDM2Pipeline(unet, audiomae=None, vocoder=voc, vae=vae)
    B, Lt = 1, 40
    e = dict(prompt_embeds=R(B, 16, 1024, seed=20), negative_prompt_embeds=R(B, 16, 1024, seed=21),
             generated_prompt_embeds=R(B, Lt, 768, seed=22), negative_generated_prompt_embeds=R(B, Lt, 768, seed=23),
             attention_mask=torch.ones(B, 16, dtype=torch.long), negative_attention_mask=torch.ones(B, 16, dtype=torch.long))
    lat0 = R(B, 8, 16, 16, seed=24)
    kw = dict(num_inference_steps=2, audio_length_in_s=0.64, latents=lat0, use_graph=False, **e)
    lat = pipe(output_type="latent", **kw).audios
    wav = pipe(output_type="np", **kw).audios
    assert wav.shape == (B, int(0.64 * 16000))
    mel = OV.decode(vsd, _cfg_dict(vcfg), lat.float().cpu() / vcfg.scaling_factor)
    hc = dict(upsample_rates=hcfg.upsample_rates, upsample_kernel_sizes=hcfg.upsample_kernel_sizes, resblock_kernel_sizes=hcfg.resblock_kernel_sizes,
              resblock_dilation_sizes=hcfg.resblock_dilation_sizes, leaky_relu_slope=hcfg.leaky_relu_slope, normalize_before=hcfg.normalize_before)
    ref = OH.hifigan(hsd, hc, mel.squeeze(1))[:, : wav.shape[1]]
    assert 1e-4 < float(ref.abs().max()) < 0.99  # an un-saturated waveform
    assert rel_err(torch.from_numpy(wav), ref) < 2e-4


def test_trainer_train_batch_encodes_mel_through_the_vae(dev):
    """train_apadapter_v2.py:892-958 from a collate batch: mel -> vae.encode(...).latent_dist.sample() * scaling_factor -> noise,
    per-sample timestep -> add_noise -> UNet -> MSE -> backward.  Re-played by hand with the same device generator: identical loss
    and gradient buffer, and the latents equal the oracle encoder's on the same noise."""
    import ap_adapter_amd as A
    from ap_adapter_amd import training as T
    from ap_adapter_amd.synthetic import init_synthetic_
    from oracle import vae as O
    dtype = torch.bfloat16
    ucfg = A.UNetConfig(block_out_channels=(64, 128, 192, 256), attention_head_dim=4, norm_num_groups=16)

    def make():
        u = A.AudioLDM2UNet2DConditionModel(ucfg)
        A.install_ap_adapter(u, None, scale=0.5)
        init_synthetic_(u, 100, w_std=0.05, bias_std=0.02, norm_jitter=0.1)
        return A.AdapterTrainer(u.to(dev, dtype), lr=1e-3, gradient_accumulation_steps=4)

    vcfg = A.VaeConfig(**SMALL)
    vae, vsd = _build(vcfg, dtype, dev, seed=13)
    B = 2
    batch = {"mel": q(R(B, 104, 64, seed=30), dtype), "prompt_embeds": q(R(B, 1, 16, 1024, seed=31), dtype),
             "generated_prompt_embeds": q(R(B, 16, 768, seed=32), dtype), "attention_mask": torch.ones(B, 16)}
    tr = make()
    loss = tr.train_batch(batch, vae, generator=torch.Generator(device=dev).manual_seed(7))
    # by hand, same generator stream
    g = torch.Generator(device=dev).manual_seed(7)
    mel = batch["mel"].to(dev, dtype).unsqueeze(1)
    n0 = torch.randn(B, 8, 26, 16, generator=g, device=dev, dtype=dtype)
    lat = vae.encode(mel).latent_dist.sample(noise=n0, scale=vcfg.scaling_factor)
    noise = torch.randn(lat.shape, generator=g, device=dev, dtype=dtype)
    t = torch.randint(0, 1000, (B,), generator=g, device=dev)
    tr2 = make()
    loss2 = tr2.train_step(lat, noise, t, batch["generated_prompt_embeds"].to(dev), batch["prompt_embeds"].squeeze(1).to(dev),
                           batch["attention_mask"].to(dev))
    assert float(loss) == float(loss2) and torch.equal(tr.grad, tr2.grad) and float(tr.grad.abs().max()) > 0
    ref = O.encode_sample(vsd, _cfg_dict(vcfg), batch["mel"].unsqueeze(1), n0.float().cpu()) * vcfg.scaling_factor
    assert lat.shape == ref.shape == (B, 8, 26, 16)
    assert rel_err(lat, ref) < 6e-2

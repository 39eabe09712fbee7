"""CPU: the VAE oracle (oracle/vae.py -- diffusers AutoencoderKL restated, parity unpinned: diffusers is not installed and the reference
holds no vector for it) against the host module's parameter scheme, and the host module's loud failure without a GPU."""
import pytest
import torch

import ap_adapter_amd as A
from oracle import vae as O

CFG = dict(norm_num_groups=8, block_out_channels=(32, 64, 64), layers_per_block=1)


def test_state_dict_scheme_is_diffusers_and_fully_consumed_by_the_oracle():
    vae = A.AutoencoderKL(A.VaeConfig(**CFG))
    sd = {k: v.detach().float() for k, v in vae.state_dict().items()}
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.0.resnets.0.norm1.weight", "encoder.down_blocks.0.downsamplers.0.conv.weight",
              "encoder.down_blocks.1.resnets.0.conv_shortcut.weight", "encoder.mid_block.attentions.0.group_norm.weight",
              "encoder.mid_block.attentions.0.to_q.bias", "encoder.mid_block.attentions.0.to_out.0.weight", "encoder.mid_block.resnets.1.conv2.bias",
              "encoder.conv_norm_out.weight", "encoder.conv_out.weight", "quant_conv.weight", "post_quant_conv.bias", "decoder.conv_in.weight",
              "decoder.up_blocks.0.resnets.1.conv1.weight", "decoder.up_blocks.0.upsamplers.0.conv.weight",
              "decoder.up_blocks.2.resnets.0.conv_shortcut.weight", "decoder.conv_norm_out.bias", "decoder.conv_out.bias"):
        assert k in sd, k
    assert "encoder.down_blocks.2.downsamplers.0.conv.weight" not in sd and "decoder.up_blocks.2.upsamplers.0.conv.weight" not in sd
    assert sd["encoder.conv_out.weight"].shape == (16, 64, 3, 3) and sd["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"].shape == (32, 64, 1, 1)

    class Spy(dict):
        def __init__(self, d):
            super().__init__(d)
            self.read = set()

        def __getitem__(self, k):
            self.read.add(k)
            return super().__getitem__(k)

    spy = Spy(sd)
    mel = O.decode(spy, CFG, torch.randn(2, 8, 6, 4))
    mean, logvar = O.encode_moments(spy, CFG, torch.randn(2, 1, 24, 16))
    assert mel.shape == (2, 1, 24, 16) and mean.shape == logvar.shape == (2, 8, 6, 4)
    assert spy.read == set(sd), sorted(set(sd) - spy.read)[:5]


def test_oracle_posterior_and_down_sampler_conventions():
    vae = A.AutoencoderKL(A.VaeConfig(**CFG))
    sd = {k: v.detach().float() for k, v in vae.state_dict().items()}
    x = torch.randn(1, 1, 24, 16)
    mean, logvar = O.encode_moments(sd, CFG, x)
    assert float(logvar.max()) <= 20 and float(logvar.min()) >= -30
    n = torch.randn_like(mean)
    assert torch.allclose(O.encode_sample(sd, CFG, x, n), mean + torch.exp(0.5 * logvar) * n)
    assert torch.equal(O.encode_sample(sd, CFG, x, torch.zeros_like(mean)), mean)
    # odd sizes: the bottom/right-only padding gives floor(H / 2) rows (symmetric padding would give ceil)
    m2, _ = O.encode_moments(sd, CFG, torch.randn(1, 1, 22, 18))
    assert m2.shape == (1, 8, 5, 4)


def test_host_module_has_no_cpu_path():
    vae = A.AutoencoderKL(A.VaeConfig(**CFG))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vae.decode(torch.randn(1, 8, 6, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vae.encode(torch.randn(1, 1, 24, 16))
    assert vae.config.scaling_factor == pytest.approx(0.4110932946205139)
    assert 2 ** (len(vae.config.block_out_channels) - 1) == A.AudioLDM2Pipeline.vae_scale_factor

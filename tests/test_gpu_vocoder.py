"""-m gpu: the HiFi-GAN vocoder on the HIP path (apad_gemm in its conv1d mode) against the transformers module's own output
(golden fixture) and against the oracle at the AudioLDM2 geometry."""
import pytest
import torch
import torch.nn.functional as F

from test_oracle_vocoder import load_gold
from util import TOL, q, rel_err

pytestmark = pytest.mark.gpu


def R(*shape, seed=0, std=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * std


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("B,T,Cin,Cout,k,dil", [(2, 50, 64, 128, 7, 1), (1, 333, 32, 32, 11, 5), (2, 100, 512, 512, 3, 3), (1, 40, 32, 8, 7, 1)])
def test_conv1d_implicit_gemm(dev, dtype, B, T, Cin, Cout, k, dil):
    """nn.Conv1d ("same" padding, dilation) with the pre-activation and residual of a HiFi-GAN residual block"""
    from ap_adapter_amd import ops
    x, w, b = q(R(B, Cin, T, seed=1), dtype), q(R(Cout, Cin, k, seed=2, std=0.05), dtype), q(R(Cout, seed=3, std=0.1), dtype)
    pad = (k * dil - dil) // 2
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=dil, padding=pad)
    res = q(R(B, Cout, T, seed=4), dtype)
    D = lambda t: t.to(dev, dtype)
    xl = D(x.transpose(1, 2).contiguous())
    wp = D(w.permute(0, 2, 1).reshape(Cout, -1).contiguous())
    out = ops.conv1d(xl, wp, D(b), k, dilation=dil, pre_slope=0.1)
    assert rel_err(out.transpose(1, 2), ref) < TOL[dtype]
    out = ops.conv1d(xl, wp, D(b), k, dilation=dil, pre_slope=0.1, residual=D(res.transpose(1, 2).contiguous()))
    assert rel_err(out.transpose(1, 2), ref + res) < TOL[dtype]
    out = ops.conv1d(xl, wp, D(b), k, dilation=dil, act="tanh")
    assert rel_err(out.transpose(1, 2), torch.tanh(F.conv1d(x, w, b, dilation=dil, padding=pad))) < TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,Cin,Cout,k,s", [(2, 37, 64, 32, 16, 5), (1, 100, 128, 64, 16, 4), (2, 61, 32, 16, 4, 2), (1, 50, 64, 32, 8, 2)])
def test_conv_transpose1d_implicit_gemm(dev, dtype, B, T, Cin, Cout, k, s):
    """nn.ConvTranspose1d(stride s, padding (k - s) // 2): the up-samplers, including k - s odd (output 5T + 1)"""
    from ap_adapter_amd import ops
    x, w, b = q(R(B, Cin, T, seed=5), dtype), q(R(Cin, Cout, k, seed=6, std=0.05), dtype), q(R(Cout, seed=7, std=0.1), dtype)
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=(k - s) // 2)
    D = lambda t: t.to(dev, dtype)
    out = ops.conv1d(D(x.transpose(1, 2).contiguous()), D(w.permute(1, 2, 0).reshape(Cout, -1).contiguous()), D(b), k, transposed_stride=s,
                     pre_slope=0.1)
    assert out.shape == (B, ref.shape[2], Cout)
    assert rel_err(out.transpose(1, 2), ref) < TOL[dtype]


def _module(cfg, sd, dev, dtype):
    from ap_adapter_amd.vocoder import HifiGanConfig, SpeechT5HifiGan
    m = SpeechT5HifiGan(HifiGanConfig(model_in_dim=cfg["model_in_dim"], upsample_initial_channel=cfg["upsample_initial_channel"],
                                      upsample_rates=tuple(cfg["upsample_rates"]), upsample_kernel_sizes=tuple(cfg["upsample_kernel_sizes"])))
    m.load_state_dict(sd)
    return m.to(dev, dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float16, 1e-2), (torch.bfloat16, 5e-2)])
def test_vocoder_vs_transformers_golden(dev, dtype, tol):
    """the whole vocoder against the output of transformers.SpeechT5HifiGan itself (committed fixture): fp32 mode to rounding;
    16-bit storage through ~40 stacked convolutions at the bounds below (relative to the largest sample)"""
    cfg, sd, x, y = load_gold()
    m = _module(cfg, sd, dev, dtype)
    out = m(x.to(dev))
    assert out.shape == y.shape and out.dtype == dtype
    assert rel_err(out, y) < tol
    one = m(x[0].to(dev))  # un-batched spectrogram -> 1-D waveform (transformers semantics)
    assert one.dim() == 1 and torch.equal(one, out[0])


def test_vocoder_audioldm2_geometry_fp32_vs_oracle(dev):
    """64 mel bins -> 1024 channels, rates 5-4-2-2-2 (x160), 0.5 s of audio, seeded weights, fp32 mode vs the pinned oracle"""
    from ap_adapter_amd.vocoder import SpeechT5HifiGan
    from ap_adapter_amd.synthetic import init_synthetic_
    from oracle import vocoder as OV
    m = SpeechT5HifiGan()
    init_synthetic_(m, 11, w_std=0.02, bias_std=0.01)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    c = m.config
    cfg = dict(upsample_rates=c.upsample_rates, upsample_kernel_sizes=c.upsample_kernel_sizes, resblock_kernel_sizes=c.resblock_kernel_sizes,
               resblock_dilation_sizes=c.resblock_dilation_sizes, leaky_relu_slope=c.leaky_relu_slope)
    x = R(1, 50, 64, seed=12)
    with torch.no_grad():
        ref = OV.hifigan(sd, cfg, x)
    out = m.to(dev)(x.to(dev))
    assert out.shape == ref.shape == (1, 50 * 160 + 32)  # the k - s = 11 of the first up-sampler adds one frame, x32 downstream
    e = rel_err(out, ref)
    print(f"\n[vocoder, AudioLDM2 geometry, fp32] rel err {e:.3e}, max|ref| {float(ref.abs().max()):.3e}")
    assert e < 2e-5
    bf = m.to(torch.bfloat16)(x.to(dev))
    assert rel_err(bf, ref) < 5e-2


def test_pipeline_mel_spectrogram_to_waveform(dev):
    """pipeline_audioldm2.py:583-590: 4-D mel [B, 1, T, 64] is squeezed, the vocoder runs, the waveform comes back as fp32 on the CPU"""
    import ap_adapter_amd as A
    cfg, sd, x, y = load_gold()
    m = _module(cfg, sd, dev, torch.float32)
    pipe = A.AudioLDM2Pipeline(unet=None, vocoder=m)
    w = pipe.mel_spectrogram_to_waveform(x.unsqueeze(1).to(dev))
    assert w.device.type == "cpu" and w.dtype == torch.float32 and rel_err(w, y) < 2e-5

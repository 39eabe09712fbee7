"""Oracle: the HiFi-GAN vocoder AudioLDM2Pipeline calls in mel_spectrogram_to_waveform
(/root/reference/pipeline/pipeline_audioldm2.py:583-590: ``self.vocoder(mel_spectrogram)``, a transformers SpeechT5HifiGan).

PINNED: transformers IS installed in the build / GPU image, so this restatement (plain torch.nn.functional on a flat state
dict) is asserted equal to ``transformers.SpeechT5HifiGan`` itself on seeded weights (tests/test_oracle_vocoder.py), and
tests/golden/vocoder_small.safetensors holds outputs of the transformers module (tests/golden/make_vocoder_golden.py).
TEST INFRASTRUCTURE ONLY.
"""
import torch
import torch.nn.functional as F


def hifigan(sd, cfg, spectrogram):
    """sd: state dict with the transformers key names; cfg: dict(upsample_rates, upsample_kernel_sizes, resblock_kernel_sizes,
    resblock_dilation_sizes, leaky_relu_slope, normalize_before).  spectrogram [B, T, mel] -> [B, T * prod(rates)]."""
    slope = cfg["leaky_relu_slope"]
    if cfg.get("normalize_before", False):
        spectrogram = (spectrogram - sd["mean"]) / sd["scale"]
    h = spectrogram.transpose(2, 1)
    h = F.conv1d(h, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (r, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        h = F.leaky_relu(h, slope)
        h = F.conv_transpose1d(h, sd[f"upsampler.{i}.weight"], sd[f"upsampler.{i}.bias"], stride=r, padding=(k - r) // 2)
        acc = None
        for j, (ks, dil) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            p = f"resblocks.{i * nk + j}."
            x = h
            for m, d in enumerate(dil):
                res = x
                x = F.conv1d(F.leaky_relu(x, slope), sd[p + f"convs1.{m}.weight"], sd[p + f"convs1.{m}.bias"], dilation=d,
                             padding=(ks * d - d) // 2)
                x = F.conv1d(F.leaky_relu(x, slope), sd[p + f"convs2.{m}.weight"], sd[p + f"convs2.{m}.bias"], padding=(ks - 1) // 2)
                x = x + res
            acc = x if acc is None else acc + x
        h = acc / nk
    h = F.leaky_relu(h)
    h = F.conv1d(h, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(h).squeeze(1)

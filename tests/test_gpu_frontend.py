"""-m gpu: audio front-end kernels (apad_resample_fir, apad_kaldi_fbank) against oracle/fbank.py on synthetic signals.
fp32 path; tolerance 2e-4 absolute on the normalised log-mel (the 512-point fp32 radix-2 FFT and the mel sums are ordered
differently from the oracle's float64 rfft; a log-mel unit is 1 / (2 * 4.569) = 0.109 normalised units) and 2e-5 on
resampled samples.  Mel bins whose energy sits at the log floor (eps) are compared like the others: both sides clamp."""
import math
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _signal(sr, seconds, seed):
    t = np.arange(int(sr * seconds)) / sr
    rs = np.random.RandomState(seed)
    x = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t + 1.0) + 0.05 * rs.randn(t.shape[0]) + 0.01
    return x.astype(np.float32)


@pytest.mark.parametrize("sr", [44100, 48000, 22050, 8000])
def test_resample_vs_oracle(dev, sr):
    from ap_adapter_amd import frontend as FE
    from oracle import fbank as OF
    x = _signal(sr, 0.7, 3)
    ref = OF.resample(x, sr, 16000)
    out = FE.resample(torch.from_numpy(x)[None].to(dev), sr, 16000)
    assert out.shape == (1, ref.shape[0])
    assert float(np.abs(out[0].cpu().numpy() - ref).max()) < 2e-5


@pytest.mark.parametrize("sr,seconds", [(16000, 3.0), (16000, 11.5), (44100, 5.0), (48000, 0.02)])
def test_fbank_feature_vs_oracle(dev, sr, seconds):
    """short clip (padding rows), long clip (crop to 1024), resampled input, clip shorter than one frame (all padding)"""
    from ap_adapter_amd import frontend as FE
    from oracle import fbank as OF
    x = np.stack([_signal(sr, seconds, 4), _signal(sr, seconds, 5)])      # stereo: channel 0 is used, the mean is global
    ref = OF.extract_kaldi_fbank_feature(x, sr)
    out = FE.extract_kaldi_fbank_feature(x, sr, torch.zeros(1024, 128), device=dev)
    assert out.shape == (1024, 128) and out.dtype == torch.float32
    assert float(np.abs(out.cpu().numpy() - ref).max()) < 2e-4


def test_load_mel_from_a_wav_file(dev, tmp_path):
    """pipeline entry: wav on disk -> [1, 1024, 128] (pipeline_audioldm2.py:919-925)"""
    from ap_adapter_amd import frontend as FE
    from oracle import fbank as OF
    x = _signal(22050, 2.0, 6)
    p = str(tmp_path / "clip.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(22050)
        w.writeframes((x * 32767).astype("<i2").tobytes())
    wav, sr = FE.load_wav(p)
    mel = FE.load_mel(p, device=dev)
    assert mel.shape == (1, 1024, 128)
    ref = OF.extract_kaldi_fbank_feature(wav, sr)
    assert float(np.abs(mel[0].cpu().numpy() - ref).max()) < 2e-4

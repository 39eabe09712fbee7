"""CPU: properties of the front-end oracle (oracle/fbank.py; PARITY UNPINNED, see its header) and of the host-side wav
reader / tables of ap-adapter_amd/frontend.py (no GPU work)."""
import math
import os
import struct
import wave

import numpy as np

from oracle import fbank as OF


def test_frame_count_padding_and_normalisation():
    x = (0.1 * np.random.RandomState(0).randn(16000 * 2)).astype(np.float32)
    fb = OF.extract_kaldi_fbank_feature(x[None], 16000)
    assert fb.shape == (1024, 128) and fb.dtype == np.float32
    n_frames = 1 + (32000 - 400) // 160
    pad = (0.0 - OF.NORM_MEAN) / (2 * OF.NORM_STD)          # zero-padded BEFORE normalisation -> +0.467, not 0
    assert np.allclose(fb[n_frames:], pad, atol=1e-6) and not np.allclose(fb[n_frames - 1], pad)
    long = (0.1 * np.random.RandomState(1).randn(16000 * 11)).astype(np.float32)
    assert OF.extract_kaldi_fbank_feature(long[None], 16000).shape == (1024, 128)      # cropped


def test_sine_lands_in_the_right_mel_bin():
    t = np.arange(16000) / 16000.0
    mel = lambda f: 1127.0 * math.log(1 + f / 700.0)
    for f0 in (440.0, 2000.0, 6000.0):
        fb = OF.kaldi_fbank((0.5 * np.sin(2 * np.pi * f0 * t)).astype(np.float32))
        lo, hi = mel(20.0), mel(8000.0)
        want = (mel(f0) - lo) / ((hi - lo) / 129) - 1           # bin whose triangle is centred on f0
        assert abs(int(fb[10].argmax()) - want) <= 1.0


def test_mel_banks_are_triangles_over_20_to_8000():
    mb = OF.mel_banks()
    assert mb.shape == (128, 257) and (mb >= 0).all() and (mb <= 1).all() and (mb[:, 256] == 0).all()
    assert mb[:, :1].sum() == 0                                  # 0 Hz is below low_freq
    centers = (mb * np.arange(257)).sum(1)[10:] / mb.sum(1)[10:]   # 128 bins over 256 FFT bins: low bins share a bin
    assert (np.diff(centers) >= 0).all() and (np.diff(centers[40:]) > 0).all()


def test_resample_kernel_is_unity_gain_and_keeps_a_tone():
    for sr in (44100, 48000, 22050, 8000):
        k, width, orig, new = OF.resample_kernel(sr, 16000)
        assert k.shape == (new, 2 * width + orig)
        assert np.allclose(k.sum(1), 1.0, atol=2e-3)            # DC gain of every phase
        t = np.arange(sr) / sr
        y = OF.resample((0.5 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32), sr, 16000)
        assert y.shape[0] == math.ceil(16000 * sr / sr) and abs(np.abs(y[200:-200]).max() - 0.5) < 5e-3
        ref = 0.5 * np.sin(2 * np.pi * 1000 * np.arange(y.shape[0]) / 16000.0)
        assert np.abs(y[200:-200] - ref[200:-200]).max() < 5e-3


def test_wav_reader_matches_the_pcm_it_wrote(tmp_path):
    from ap_adapter_amd.frontend import load_wav, _mel_banks, _resample_kernel
    x = (np.random.RandomState(2).rand(2, 1000) * 2 - 1).astype(np.float32)
    p16 = str(tmp_path / "a.wav")
    with wave.open(p16, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100)
        w.writeframes((np.clip(x.T, -1, 1) * 32767).astype("<i2").tobytes())
    y, sr = load_wav(p16)
    assert sr == 44100 and y.shape == (2, 1000) and np.abs(y - x).max() < 1.0 / 16000
    pf = str(tmp_path / "f.wav")
    body = x.T.astype("<f4").tobytes()
    with open(pf, "wb") as f:
        fmt = struct.pack("<HHIIHH", 3, 2, 16000, 16000 * 8, 8, 32)
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt
                + b"data" + struct.pack("<I", len(body)) + body)
    y, sr = load_wav(pf)
    assert sr == 16000 and np.array_equal(y, x)
    # the product's host tables are the oracle's formulas
    assert np.array_equal(_mel_banks(128), OF.mel_banks())
    assert np.array_equal(_resample_kernel(44100, 16000)[0], OF.resample_kernel(44100, 16000)[0])


def test_fbank_oracle_agrees_with_an_independent_kaldi_fbank_implementation():
    """transformers.audio_utils re-implements torchaudio.compliance.kaldi.fbank in numpy for its AST feature extractor
    (same 25 ms hanning / 10 ms / 128 kaldi-mel-bin / pre-emphasis 0.97 / DC-removal configuration AudioMAE inherited
    from AST).  torchaudio itself is not installable here, so this is the closest available pin of the restatement."""
    import warnings
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    rs = np.random.RandomState(0)
    x = (0.1 * rs.randn(32000) + 0.3 * np.sin(2 * np.pi * 440 * np.arange(32000) / 16000.0)).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mel = mel_filter_bank(num_frequency_bins=257, num_mel_filters=128, min_frequency=20, max_frequency=8000,
                              sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    assert np.abs(mel.T - OF.mel_banks()).max() < 1e-4
    ref = spectrogram(x, window_function(400, "hann", periodic=False), frame_length=400, hop_length=160, fft_length=512,
                      power=2.0, center=False, preemphasis=0.97, mel_filters=mel, log_mel="log",
                      mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    got = OF.kaldi_fbank(x)
    assert got.shape == ref.shape == (198, 128)
    assert np.abs(got - ref).max() < 2e-3           # log-mel units (values span ~[-16, 5])

"""Oracle for the audio front-end ("next" row f-2).  TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED**: the arithmetic lives in
torchaudio==2.0.2 (requirements.txt:12-13 of the reference; not installed, not vendored), restated here from its
published algorithm and anchored on the reference's call site

    /root/reference/audio_encoder/AudioMAE.py:356-394  extract_kaldi_fbank_feature(waveform, sampling_rate, log_mel_spec)
      :361-366  torchaudio.functional.resample(waveform, orig_freq, 16000)        (defaults: sinc_interp_hann, width 6, rolloff 0.99)
      :368      waveform - waveform.mean()
      :369-378  torchaudio.compliance.kaldi.fbank(htk_compat=True, sample_frequency=16000, use_energy=False,
                window_type="hanning", num_mel_bins=128, dither=0.0, frame_shift=10)
                (defaults: frame_length 25 ms, preemphasis 0.97, remove_dc_offset, round_to_power_of_two, snip_edges,
                 low_freq 20, high_freq 0 = Nyquist, use_power, use_log_fbank, channel 0)
      :380-391  zero-pad / crop to TARGET_LEN frames BEFORE normalisation
      :393      (fbank - norm_mean) / (norm_std * 2),  norm_mean -4.2677393, norm_std 4.5689974

Cross-check available offline: transformers.audio_utils carries an independent numpy re-implementation of kaldi.fbank
(for its AST feature extractor, the model AudioMAE inherited this front-end from); tests/test_frontend_oracle.py holds
this restatement to it (log-mel within 2e-3, mel banks within 1e-4).  The resampler has no such second source.

numpy, float32 where torch computes in float32.  The tables (window, mel banks, resampling kernel) are shared with the
product path by construction of the same formulas in ap-adapter_amd/frontend.py; this file recomputes them independently.
"""
import math

import numpy as np

NORM_MEAN = -4.2677393
NORM_STD = 4.5689974
EPS = np.float32(1.1920928955078125e-07)


def resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional._get_sinc_resample_kernel (sinc_interp_hann), float32 like waveform.dtype"""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    f32 = np.float32
    idx = (np.arange(-width, width + orig, dtype=f32) / f32(orig))[None, :]
    t = (np.arange(0, -new, -1, dtype=f32) / f32(new))[:, None] + idx
    t = t * f32(base_freq)
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width).astype(f32)
    window = np.cos(t * f32(math.pi) / f32(lowpass_filter_width) / f32(2)) ** 2
    t = t * f32(math.pi)
    scale = f32(base_freq / orig)
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.where(t == 0, f32(1.0), np.sin(t) / t).astype(f32)
    return (k * window * scale).astype(f32), width, orig, new


def resample(x, orig_freq, new_freq):
    """x [samples] float32 -> float32 [ceil(new * n / orig)]"""
    if int(orig_freq) == int(new_freq):
        return x.astype(np.float32)
    kern, width, orig, new = resample_kernel(orig_freq, new_freq)
    n = x.shape[0]
    xp = np.concatenate([np.zeros(width, np.float32), x.astype(np.float32), np.zeros(width + orig, np.float32)])
    nblk = (xp.shape[0] - kern.shape[1]) // orig + 1
    out = np.empty((nblk, new), np.float32)
    for b in range(nblk):
        out[b] = kern @ xp[b * orig: b * orig + kern.shape[1]]
    target = int(math.ceil(new * n / orig))
    return out.reshape(-1)[:target]


def mel_banks(num_bins=128, padded=512, sample_freq=16000.0, low=20.0, high=0.0):
    """torchaudio.compliance.kaldi.get_mel_banks without VTLN: [num_bins, padded/2 + 1] (last column zero)"""
    f32 = np.float32
    nfft = padded // 2
    nyq = 0.5 * sample_freq
    if high <= 0:
        high += nyq
    mel = lambda f: f32(1127.0) * np.log(f32(1.0) + np.asarray(f, f32) / f32(700.0))
    fft_bin_width = sample_freq / padded
    mlo, mhi = mel(low), mel(high)
    delta = (mhi - mlo) / f32(num_bins + 1)
    b = np.arange(num_bins, dtype=f32)[:, None]
    left, center, right = mlo + b * delta, mlo + (b + 1) * delta, mlo + (b + 2) * delta
    m = mel(f32(fft_bin_width) * np.arange(nfft, dtype=f32))[None, :]
    up = (m - left) / (center - left)
    down = (right - m) / (right - center)
    w = np.maximum(f32(0), np.minimum(up, down)).astype(f32)
    return np.concatenate([w, np.zeros((num_bins, 1), f32)], axis=1)


def kaldi_fbank(x, num_mel_bins=128):
    """x float32 [samples] at 16 kHz -> log-mel [frames, 128] float32 (the :369-378 call)"""
    f32 = np.float32
    win, shift, padded = 400, 160, 512
    n = x.shape[0]
    if n < win:
        return np.zeros((0, num_mel_bins), f32)
    m = 1 + (n - win) // shift
    frames = np.stack([x[i * shift: i * shift + win] for i in range(m)]).astype(f32)
    frames = frames - frames.mean(axis=1, keepdims=True, dtype=f32)
    prev = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)  # replicate pad on the left
    frames = frames - f32(0.97) * prev
    window = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win, dtype=np.float64) / (win - 1))).astype(f32)  # hann, periodic=False
    frames = frames * window
    frames = np.concatenate([frames, np.zeros((m, padded - win), f32)], axis=1)
    spec = np.abs(np.fft.rfft(frames.astype(np.float64), axis=1)).astype(f32) ** 2
    e = spec @ mel_banks(num_mel_bins).T
    return np.log(np.maximum(e, EPS)).astype(f32)


def extract_kaldi_fbank_feature(waveform, sampling_rate, target_len=1024):
    """waveform float32 [channels, samples] in [-1, 1] (torchaudio.load convention) -> [target_len, 128]"""
    w = np.asarray(waveform, np.float32)
    if w.ndim == 1:
        w = w[None]
    if int(sampling_rate) != 16000:
        w = np.stack([resample(c, sampling_rate, 16000) for c in w])
    w = w - w.mean(dtype=np.float32)
    fb = kaldi_fbank(w[0])
    p = target_len - fb.shape[0]
    if p > 0:
        fb = np.concatenate([fb, np.zeros((p, fb.shape[1]), np.float32)])
    elif p < 0:
        fb = fb[:target_len]
    return ((fb - np.float32(NORM_MEAN)) / np.float32(NORM_STD * 2)).astype(np.float32)

"""Audio front-end ("next" row f-2): wav -> 16 kHz -> Kaldi log-mel [1024, 128], the input of AudioMAE.

Mirrors /root/reference/audio_encoder/AudioMAE.py:356-394 ``extract_kaldi_fbank_feature(waveform, sampling_rate,
log_mel_spec)`` and its caller /root/reference/pipeline/pipeline_audioldm2.py:919-925 (``torchaudio.load`` ->
``extract_kaldi_fbank_feature`` -> ``unsqueeze(0)``).  torchaudio is not a dependency: the wav container is read with the
standard library, resampling and the filterbank run in libapadapter_hip.so (``apad_resample_fir``,
``apad_kaldi_fbank``); only the small constant tables (sinc kernel, hann window, FFT twiddles, mel banks) are built
on the host, once per configuration.
"""
import math
import struct
import wave

import numpy as np
import torch

from . import _lib as L
from . import ops

NORM_MEAN = -4.2677393   # AudioMAE.py:357-358
NORM_STD = 4.5689974
_tables = {}


def load_wav(path):
    """(waveform float32 [channels, samples] in [-1, 1), sample_rate): what ``torchaudio.load`` returns for PCM / float
    RIFF files (the only container the reference's data uses)."""
    try:
        with wave.open(path, "rb") as w:
            ch, sw, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
            raw = w.readframes(n)
        if sw == 2:
            a = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
        elif sw == 4:
            a = np.frombuffer(raw, "<i4").astype(np.float32) / 2147483648.0
        elif sw == 3:
            b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            a = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
        elif sw == 1:
            a = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
        else:
            raise ValueError(f"{path}: unsupported PCM sample width {sw}")
    except wave.Error:
        a, ch, sr = _read_float_wav(path)  # WAVE_FORMAT_IEEE_FLOAT is rejected by the stdlib reader
    return np.ascontiguousarray(a.reshape(-1, ch).T), sr


def _read_float_wav(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None or fmt[0] not in (3, 0xFFFE) or fmt[5] != 32:
        raise ValueError(f"{path}: unsupported wav encoding")
    return np.frombuffer(pcm, "<f4").astype(np.float32), fmt[1], fmt[2]


def _resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """sinc_interp_hann kernel of torchaudio.functional.resample, computed in float32 like the waveform"""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    f32 = np.float32
    idx = (np.arange(-width, width + orig, dtype=f32) / f32(orig))[None, :]
    t = ((np.arange(0, -new, -1, dtype=f32) / f32(new))[:, None] + idx) * f32(base)
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width).astype(f32)
    win = np.cos(t * f32(math.pi) / f32(lowpass_filter_width) / f32(2)) ** 2
    t = t * f32(math.pi)
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.where(t == 0, f32(1.0), np.sin(t) / t).astype(f32)
    return (k * win * f32(base / orig)).astype(f32), width, orig, new


def _mel_banks(num_bins, padded=512, sample_freq=16000.0, low=20.0, high=0.0):
    """kaldi.get_mel_banks (no VTLN) + the zero column for the Nyquist bin: [num_bins, 257] float32"""
    f32 = np.float32
    mel = lambda f: f32(1127.0) * np.log(f32(1.0) + np.asarray(f, f32) / f32(700.0))
    high = high + 0.5 * sample_freq if high <= 0 else high
    mlo, mhi = mel(low), mel(high)
    delta = (mhi - mlo) / f32(num_bins + 1)
    b = np.arange(num_bins, dtype=f32)[:, None]
    left, center, right = mlo + b * delta, mlo + (b + 1) * delta, mlo + (b + 2) * delta
    m = mel(f32(sample_freq / padded) * np.arange(padded // 2, dtype=f32))[None, :]
    w = np.maximum(f32(0), np.minimum((m - left) / (center - left), (right - m) / (right - center))).astype(f32)
    return np.concatenate([w, np.zeros((num_bins, 1), f32)], axis=1)


def _fbank_tables(dev, num_mel_bins):
    key = (str(dev), num_mel_bins)
    if key not in _tables:
        n = np.arange(400, dtype=np.float64)
        window = (0.5 - 0.5 * np.cos(2 * np.pi * n / 399)).astype(np.float32)  # torch.hann_window(400, periodic=False)
        k = np.arange(256, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * k / 512), -np.sin(2 * np.pi * k / 512)], axis=1).astype(np.float32)
        _tables[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (window, tw, _mel_banks(num_mel_bins)))
    return _tables[key]


def resample(waveform, orig_freq, new_freq):
    """waveform fp32 GPU [channels, samples] -> [channels, ceil(new * n / orig)]"""
    if int(orig_freq) == int(new_freq):
        return waveform
    key = ("rs", str(waveform.device), int(orig_freq), int(new_freq))
    if key not in _tables:
        k, width, orig, new = _resample_kernel(orig_freq, new_freq)
        _tables[key] = (torch.from_numpy(k).to(waveform.device), width, orig, new)
    kern, width, orig, new = _tables[key]
    ch, n = waveform.shape
    n_out = int(math.ceil(new * n / orig))
    out = torch.empty(ch, n_out, dtype=torch.float32, device=waveform.device)
    for c in range(ch):
        src = waveform[c].contiguous()
        L.check(L.lib().apad_resample_fir(src.data_ptr(), kern.data_ptr(), out[c].data_ptr(), n, n_out, orig, new, width,
                                          ops._stream()), "apad_resample_fir")
    return out


def extract_kaldi_fbank_feature(waveform, sampling_rate, log_mel_spec=None, device=None):
    """Reference signature (AudioMAE.py:356).  waveform: float [channels, samples] (numpy or tensor, torchaudio.load
    convention); ``log_mel_spec`` only supplies TARGET_LEN = log_mel_spec.size(0) (1024 when omitted).
    Returns fp32 [TARGET_LEN, 128] on the GPU."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    w = torch.as_tensor(np.asarray(waveform, dtype=np.float32) if not torch.is_tensor(waveform) else waveform)
    w = w.to(device=dev, dtype=torch.float32)
    if w.dim() == 1:
        w = w.unsqueeze(0)
    target = 1024 if log_mel_spec is None else int(log_mel_spec.shape[0])
    w16 = resample(w.contiguous(), sampling_rate, 16000)
    # :368 subtracts the global mean; the per-frame DC removal inside the filterbank absorbs it exactly (up to fp32
    # rounding), the scalar is passed through for fidelity
    dc = float(w16.mean())
    ch0 = w16[0].contiguous()  # kaldi.fbank: channel 0
    window, tw, mel = _fbank_tables(dev, 128)
    out = torch.empty(target, 128, dtype=torch.float32, device=dev)
    L.check(L.lib().apad_kaldi_fbank(ch0.data_ptr(), ch0.numel(), dc, window.data_ptr(), tw.data_ptr(), mel.data_ptr(),
                                     out.data_ptr(), target, 128, 0.97, NORM_MEAN, NORM_STD, ops._stream()), "apad_kaldi_fbank")
    return out


def load_mel(audio_file, device=None):
    """pipeline_audioldm2.py:919-925: wav file -> mel_spect_tensor [1, 1024, 128]"""
    waveform, sr = load_wav(audio_file)
    return extract_kaldi_fbank_feature(waveform, sr, device=device).unsqueeze(0)

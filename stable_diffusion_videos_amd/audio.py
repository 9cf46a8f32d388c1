"""Audio-driven interpolation schedule - ``get_timesteps_arr`` of the reference
(/root/reference/stable_diffusion_videos/utils.py:12-39), SURVEY.md section 8(f) rank 1.

The reference delegates every signal-processing step to ``librosa`` (not installed offline): ``load`` ->
``stft(n_fft=2048, hop=512)`` -> ``decompose.hpss(margin)`` -> ``istft`` -> ``feature.melspectrogram`` -> column max
-> min-max normalise -> cumsum -> ``np.interp`` to ``int(duration * fps)`` points -> blend with a linear ramp by
``smooth``.  This module restates those published librosa algorithms with numpy / scipy (librosa defaults: hann
window, centred frames with zero padding, 31-bin median filters, power-2 soft masks, 128 Slaney-normalised mel
bands).  **parity unpinned**: there is no librosa here to generate reference vectors, so tests check the
mathematical properties (STFT/iSTFT round trip, mask partition of unity, mel filter known answers, monotone T).
It is host-side, millisecond-scale work per clip and stays on the CPU.

Two places where librosa's own behaviour changed between releases - the reference does not pin librosa
(pyproject.toml), so both are explicit here instead of silently picked:
* ``STFT_PAD_MODE``: centred frames are padded with zeros since librosa 0.10 ("constant"), by reflection before;
* ``load`` resamples with ``soxr_hq`` since 0.10 (``kaiser_best`` before); neither library exists offline, this module
  uses scipy's polyphase ``resample_poly`` - a third band-limited resampler.  The schedule is a cumulative sum of a
  128-band envelope, so the choice moves T by far less than one frame (tests/test_audio.py bounds both effects).
"""
from __future__ import annotations

import numpy as np

N_FFT = 2048
HOP = N_FFT // 4
SR = 22050
STFT_PAD_MODE = "constant"      # librosa >= 0.10; "reflect" reproduces librosa < 0.10


def load_audio(path, sr: int = SR, mono: bool = True, offset: float = 0.0, duration=None):
    """``librosa.load``: float32 in [-1, 1], mono, resampled to ``sr``, cut to [offset, offset + duration)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    native_sr, y = wavfile.read(str(path))
    if y.dtype.kind == "i":
        y = y.astype(np.float32) / float(np.iinfo(y.dtype).max + 1)
    elif y.dtype.kind == "u":
        y = (y.astype(np.float32) - 128.0) / 128.0
    else:
        y = y.astype(np.float32)
    if y.ndim == 2:
        y = y.mean(axis=1) if mono else y.T
    start = int(round(offset * native_sr))
    stop = None if duration is None else start + int(round(duration * native_sr))
    y = y[..., start:stop]
    if native_sr != sr:
        g = np.gcd(int(sr), int(native_sr))
        y = resample_poly(y, sr // g, native_sr // g, axis=-1).astype(np.float32)
    return np.ascontiguousarray(y, dtype=np.float32), sr


def _hann(n: int) -> np.ndarray:
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float32)      # periodic hann


def stft(y: np.ndarray, n_fft: int = N_FFT, hop: int = HOP, pad_mode: str = None) -> np.ndarray:
    """Centred STFT (frames padded per ``pad_mode`` / ``STFT_PAD_MODE``), hann window -> complex64
    [1 + n_fft/2, 1 + len(y)//hop]."""
    ypad = np.pad(y.astype(np.float32), n_fft // 2, mode=pad_mode or STFT_PAD_MODE)
    n_frames = 1 + (len(ypad) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = ypad[idx] * _hann(n_fft)[None, :]
    return np.fft.rfft(frames, axis=1).T.astype(np.complex64)


def istft(D: np.ndarray, length: int, n_fft: int = N_FFT, hop: int = HOP) -> np.ndarray:
    """Inverse of ``stft`` by windowed overlap-add with squared-window normalisation."""
    win = _hann(n_fft)
    frames = np.fft.irfft(D.T, n=n_fft, axis=1).astype(np.float32) * win[None, :]
    n_frames = frames.shape[0]
    out = np.zeros(n_fft + hop * (n_frames - 1), dtype=np.float32)
    wsum = np.zeros_like(out)
    for i in range(n_frames):
        out[i * hop:i * hop + n_fft] += frames[i]
        wsum[i * hop:i * hop + n_fft] += win * win
    nz = wsum > np.finfo(np.float32).tiny
    out[nz] /= wsum[nz]
    out = out[n_fft // 2:]
    if len(out) < length:
        out = np.pad(out, (0, length - len(out)))
    return out[:length]


def softmask(X: np.ndarray, X_ref: np.ndarray, power: float = 2.0, split_zeros: bool = False) -> np.ndarray:
    Z = np.maximum(X, X_ref).astype(np.float32)
    bad = Z < np.finfo(np.float32).tiny
    Z[bad] = 1
    mask = (X / Z) ** power
    ref = (X_ref / Z) ** power
    good = ~bad
    mask[good] /= mask[good] + ref[good]
    mask[bad] = 0.5 if split_zeros else 0.0
    return mask


def hpss(D: np.ndarray, margin: float = 1.0, kernel_size: int = 31, power: float = 2.0):
    """Median-filtering harmonic / percussive separation (Fitzgerald 2010, Driedger 2014 margins)."""
    from scipy.ndimage import median_filter
    S = np.abs(D)
    harm = median_filter(S, size=(1, kernel_size), mode="reflect")     # smooth along time  -> harmonic
    perc = median_filter(S, size=(kernel_size, 1), mode="reflect")     # smooth along freq. -> percussive
    split = margin == 1
    mask_h = softmask(harm, perc * margin, power=power, split_zeros=split)
    mask_p = softmask(perc, harm * margin, power=power, split_zeros=split)
    return D * mask_h, D * mask_p


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    mel = f / (200.0 / 3)
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / (200.0 / 3) + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mel)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f = m * (200.0 / 3)
    min_log_mel, logstep = 1000.0 / (200.0 / 3), np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, 1000.0 * np.exp(logstep * (m - min_log_mel)), f)


def mel_filterbank(sr: int = SR, n_fft: int = N_FFT, n_mels: int = 128) -> np.ndarray:
    """Slaney-scale triangular filters with Slaney area normalisation (librosa.filters.mel defaults)."""
    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    fb = np.maximum(0.0, np.minimum(lower, upper))
    fb *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return fb.astype(np.float32)


def melspectrogram(y: np.ndarray, sr: int = SR) -> np.ndarray:
    return mel_filterbank(sr) @ (np.abs(stft(y)) ** 2)


def get_timesteps_arr(audio_filepath, offset, duration, fps=30, margin=1.0, smooth=0.0):
    """Interpolation weights T (length ``int(duration * fps)``) that advance with the percussive energy."""
    y, sr = load_audio(audio_filepath, offset=offset, duration=duration)
    _, D_perc = hpss(stft(y), margin=margin)
    y_perc = istft(D_perc, length=len(y))
    spec_max = np.amax(melspectrogram(y_perc, sr), axis=0)
    spec_norm = (spec_max - np.min(spec_max)) / np.ptp(spec_max)
    x_norm = np.linspace(0, spec_norm.shape[-1], spec_norm.shape[-1])
    y_norm = np.cumsum(spec_norm)
    y_norm /= y_norm[-1]
    x_resize = np.linspace(0, y_norm.shape[-1], int(duration * fps))
    T = np.interp(x_resize, x_norm, y_norm)
    return T * (1 - smooth) + np.linspace(0.0, 1.0, T.shape[0]) * smooth

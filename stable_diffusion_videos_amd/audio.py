"""Audio-driven interpolation schedule (reference: utils.py:12-39, built on librosa).  SURVEY.md section 8(f)
rank 1.  librosa is not installed in this image; when it is, the reference computation is used verbatim."""
from __future__ import annotations

import numpy as np


def load_audio(path, sr=22050, mono=True, offset=0.0, duration=None):
    try:
        import librosa
    except Exception as exc:  # pragma: no cover
        raise RuntimeError("audio-driven walks need librosa (reference utils.py:13), which is not installed here") from exc
    return librosa.load(path, sr=sr, mono=mono, offset=offset, duration=duration)


def get_timesteps_arr(audio_filepath, offset, duration, fps=30, margin=1.0, smooth=0.0):
    try:
        import librosa
    except Exception as exc:  # pragma: no cover
        raise RuntimeError("audio-driven walks need librosa (reference utils.py:12-39), which is not installed here; "
                           "pass audio_filepath=None") from exc
    y, sr = librosa.load(audio_filepath, offset=offset, duration=duration)
    D = librosa.stft(y, n_fft=2048, hop_length=2048 // 4, win_length=2048)
    _, D_percussive = librosa.decompose.hpss(D, margin=margin)
    y_percussive = librosa.istft(D_percussive, length=len(y))
    spec_max = np.amax(librosa.feature.melspectrogram(y=y_percussive, sr=sr), axis=0)
    spec_norm = (spec_max - np.min(spec_max)) / np.ptp(spec_max)
    x_norm = np.linspace(0, spec_norm.shape[-1], spec_norm.shape[-1])
    y_norm = np.cumsum(spec_norm)
    y_norm /= y_norm[-1]
    T = np.interp(np.linspace(0, y_norm.shape[-1], int(duration * fps)), x_norm, y_norm)
    return T * (1 - smooth) + np.linspace(0.0, 1.0, T.shape[0]) * smooth

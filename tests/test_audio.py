"""Audio -> interpolation schedule (reference utils.py:12-39).  librosa is not installable
here, so these are property / known-answer tests of the numpy-scipy restatement ("parity unpinned")."""
from pathlib import Path

import numpy as np
import pytest

from stable_diffusion_videos_amd import audio, get_timesteps_arr

WAV = Path(__file__).parent / "samples" / "choice.wav"     # the reference's own fixture (tests/samples/choice.wav)


def test_load_matches_wav_contract():
    y, sr = audio.load_audio(WAV)
    assert sr == 22050 and y.dtype == np.float32 and y.ndim == 1 and abs(len(y) / sr - 10.0) < 1e-3
    assert np.abs(y).max() <= 1.0
    seg, _ = audio.load_audio(WAV, offset=2.0, duration=1.5)
    assert len(seg) == int(round(1.5 * sr)) and np.array_equal(seg, y[2 * sr:2 * sr + len(seg)])


def test_stft_istft_round_trip_and_shapes():
    y, sr = audio.load_audio(WAV, offset=1.0, duration=2.0)
    D = audio.stft(y)
    assert D.shape == (1025, 1 + len(y) // 512) and D.dtype == np.complex64
    back = audio.istft(D, length=len(y))
    assert np.abs(back - y)[1024:-1024].max() < 1e-4
    # a pure tone lands in the right bin
    t = np.arange(sr) / sr
    tone = np.sin(2 * np.pi * 1000.0 * t).astype(np.float32)
    assert abs(int(np.abs(audio.stft(tone)).mean(axis=1).argmax()) - round(1000.0 * 2048 / sr)) <= 1


def test_hpss_masks_partition_and_separate():
    sr = 22050
    t = np.arange(2 * sr) / sr
    tone = 0.5 * np.sin(2 * np.pi * 440.0 * t)
    clicks = np.zeros_like(tone)
    clicks[::sr // 4] = 1.0
    D = audio.stft((tone + clicks).astype(np.float32))
    H, P = audio.hpss(D, margin=1.0)
    assert np.allclose(H + P, D, atol=1e-4)                 # margin 1: the two soft masks sum to one
    eh, ep = np.abs(H) ** 2, np.abs(P) ** 2
    tone_bin = round(440.0 * 2048 / sr)
    assert eh[tone_bin].sum() > 10 * ep[tone_bin].sum()     # the steady tone goes to the harmonic part
    click_frame = (sr // 4) // 512
    assert ep[600:, click_frame].sum() > eh[600:, click_frame].sum()   # the click goes to the percussive part
    H2, P2 = audio.hpss(D, margin=3.0)
    assert (np.abs(H2) + np.abs(P2) <= np.abs(D) + 1e-4).all()


def test_mel_filterbank_known_answers():
    fb = audio.mel_filterbank()
    assert fb.shape == (128, 1025) and (fb >= 0).all()
    assert abs(float(audio._hz_to_mel(1000.0)) - 15.0) < 1e-9 and abs(float(audio._mel_to_hz(15.0)) - 1000.0) < 1e-6
    assert abs(float(audio._hz_to_mel(6400.0)) - 42.0) < 1e-9           # log region: 27 mels per factor 6.4
    centers = fb.argmax(axis=1)
    assert (np.diff(centers) > 0).all()
    # slaney normalisation: each triangle integrates to ~1 over frequency (bin width sr/n_fft)
    area = fb.sum(axis=1) * (22050 / 2048)
    assert np.allclose(area[5:-5], 1.0, atol=0.15)


def test_get_timesteps_arr_properties():
    fps = 6
    for offset, duration in ((2, 2), (4, 1), (5, 3)):                   # the reference's audio test offsets
        T = get_timesteps_arr(str(WAV), offset=offset, duration=duration, fps=fps)
        assert T.shape == (int(duration * fps),) and T.dtype == np.float64
        assert (np.diff(T) >= -1e-12).all() and T[0] >= 0 and abs(T[-1] - 1.0) < 1e-9
    lin = get_timesteps_arr(str(WAV), offset=2, duration=2, fps=fps, smooth=1.0)
    assert np.allclose(lin, np.linspace(0, 1, 12))
    half = get_timesteps_arr(str(WAV), offset=2, duration=2, fps=fps, smooth=0.5)
    raw = get_timesteps_arr(str(WAV), offset=2, duration=2, fps=fps)
    assert np.allclose(half, 0.5 * raw + 0.5 * np.linspace(0, 1, 12))
    assert not np.allclose(raw, np.linspace(0, 1, 12), atol=0.02)      # audio really bends the schedule


def test_librosa_version_choices_move_T_by_less_than_a_frame(monkeypatch):
    """The reference does not pin librosa; its STFT padding changed ("reflect" < 0.10 <= "constant") and so did ``load``'s
    resampler.  Both effects on the schedule are bounded here: at 30 fps a frame is 1/n of the [0, 1] range."""
    fps, offset, duration = 30, 2, 2
    n = duration * fps
    base = get_timesteps_arr(str(WAV), offset=offset, duration=duration, fps=fps)
    monkeypatch.setattr(audio, "STFT_PAD_MODE", "reflect")
    refl = get_timesteps_arr(str(WAV), offset=offset, duration=duration, fps=fps)
    monkeypatch.setattr(audio, "STFT_PAD_MODE", "constant")
    assert np.abs(refl - base).max() < 0.5 / n
    # the fixture is native 22.05 kHz (``load`` does not resample it), so a 2x round trip - polyphase up, FFT-based down -
    # stands in for "a different band-limited resampler" in ``load``
    from scipy.signal import resample
    real_load = audio.load_audio

    def fft_load(path, sr=audio.SR, mono=True, offset=0.0, duration=None):
        y, native = real_load(path, sr=44100, mono=mono, offset=offset, duration=duration)
        return resample(y, int(round(len(y) * sr / native))).astype(np.float32), sr

    monkeypatch.setattr(audio, "load_audio", fft_load)
    other = get_timesteps_arr(str(WAV), offset=offset, duration=duration, fps=fps)
    assert np.abs(other - base).max() < 0.5 / n


def test_stft_and_mel_stage_match_the_transformers_restatement_of_librosa():
    """An independent pin for two of the unpinned stages: ``transformers.audio_utils`` (installed here, written to reproduce
    ``librosa.stft`` / ``librosa.filters.mel`` / ``librosa.feature.melspectrogram`` for the Whisper / CLAP feature extractors)
    must agree with the numpy restatement on the reference's own fixture - complex STFT, Slaney filter bank and the mel power
    spectrogram of ``utils.py:25`` (n_fft 2048, hop 512, periodic hann, centred).  HPSS and the resampler stay unpinned."""
    tau = pytest.importorskip("transformers.audio_utils")
    y, sr = audio.load_audio(WAV, offset=1.0, duration=3.0)
    fb = tau.mel_filter_bank(num_frequency_bins=1025, num_mel_filters=128, min_frequency=0.0, max_frequency=sr / 2,
                             sampling_rate=sr, norm="slaney", mel_scale="slaney")
    assert np.abs(fb.T - audio.mel_filterbank(sr)).max() < 1e-7 * np.abs(fb).max()
    assert abs(float(tau.hertz_to_mel(4000.0, "slaney")) - float(audio._hz_to_mel(4000.0))) < 1e-9
    win = tau.window_function(2048, "hann")
    for pad in ("reflect", "constant"):                      # the two librosa-version-dependent paddings (audio.STFT_PAD_MODE)
        S = tau.spectrogram(y, win, frame_length=2048, hop_length=512, fft_length=2048, power=None, center=True, pad_mode=pad)
        D = audio.stft(y, pad_mode=pad)
        assert S.shape == D.shape and np.abs(S - D).max() < 1e-6 * np.abs(S).max()
        mel = tau.spectrogram(y, win, frame_length=2048, hop_length=512, fft_length=2048, power=2.0, center=True, pad_mode=pad,
                              mel_filters=fb, mel_floor=0.0)
        mine = audio.mel_filterbank(sr) @ (np.abs(D) ** 2)
        assert np.abs(mel - mine).max() < 1e-6 * np.abs(mel).max()

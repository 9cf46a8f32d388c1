"""mp4 muxing (reference utils.py:69-128).  Without ffmpeg the built-in writer emits an ISO-BMFF
file with a Motion-JPEG track; the test re-parses the container and decodes the samples back."""
import io
import struct

import numpy as np
import pytest
import torch
from PIL import Image

from stable_diffusion_videos_amd import make_video_pyav


def parse_boxes(buf, start=0, end=None):
    end = len(buf) if end is None else end
    out, pos = [], start
    while pos < end:
        size, kind = struct.unpack(">I4s", buf[pos:pos + 8])
        out.append((kind.decode("latin1"), pos + 8, pos + size))
        pos += size
    return out


def find(buf, path):
    lo, hi = 0, len(buf)
    for name in path:
        skip = 8 if name == "stsd_entry" else 0
        boxes = parse_boxes(buf, lo, hi)
        kind, lo, hi = next(b for b in boxes if b[0] == name)
        lo += skip
    return lo, hi


def test_mjpeg_mp4_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    base = np.kron(rng.integers(0, 255, (8, 8, 3)), np.ones((8, 8, 1))).astype(np.uint8)      # 64x64 blocky image
    frames = [np.roll(base, 4 * k, axis=1) for k in range(5)]
    d = tmp_path / "clip"
    d.mkdir()
    for k, fr in enumerate(frames):
        Image.fromarray(fr).save(d / f"frame{k:06d}.png")
    out = make_video_pyav(d, fps=5, output_filepath=tmp_path / "clip" / "clip.mp4", glob_pattern="*.png")
    buf = open(out, "rb").read()
    top = parse_boxes(buf)
    assert [b[0] for b in top] == ["ftyp", "mdat", "moov"]
    stbl = ("moov", "trak", "mdia", "minf", "stbl")
    lo, hi = find(buf, stbl + ("stsz",))
    _, _, n = struct.unpack(">III", buf[lo:lo + 12])
    sizes = struct.unpack(f">{n}I", buf[lo + 12:lo + 12 + 4 * n])
    assert n == 5
    lo, hi = find(buf, stbl + ("stco",))
    offset = struct.unpack(">III", buf[lo:lo + 12])[2]
    lo, hi = find(buf, stbl + ("stts",))
    _, entries, count, delta = struct.unpack(">IIII", buf[lo:lo + 16])
    assert (entries, count, delta) == (1, 5, 18000)                     # 90 kHz timescale / 5 fps
    lo, hi = find(buf, ("moov", "trak", "tkhd"))
    w, h = struct.unpack(">II", buf[hi - 8:hi])
    assert (w >> 16, h >> 16) == (64, 64)
    pos = offset
    for k, sz in enumerate(sizes):                                       # every sample is a decodable JPEG of the frame
        im = np.asarray(Image.open(io.BytesIO(buf[pos:pos + sz])).convert("RGB")).astype(int)
        assert im.shape == (64, 64, 3) and np.abs(im - frames[k].astype(int)).mean() < 15
        pos += sz
    # tensor input form (T, C, H, W), as the reference accepts
    t = torch.from_numpy(np.stack(frames)).permute(0, 3, 1, 2)
    out2 = make_video_pyav(t, fps=5, output_filepath=tmp_path / "t.mp4")
    assert abs(len(open(out2, "rb").read()) - len(buf)) < 64


def traks(buf):
    lo, hi = find(buf, ("moov",))
    return [(a, b) for kind, a, b in parse_boxes(buf, lo, hi) if kind == "trak"]


def find_in(buf, lo, hi, path):
    for name in path:
        kind, lo, hi = next(b for b in parse_boxes(buf, lo, hi) if b[0] == name)
    return lo, hi


def test_mjpeg_mp4_carries_the_audio_window(tmp_path):
    """make_video_pyav(audio_filepath=..., audio_offset, audio_duration) (utils.py:69-128, called at
    stable_diffusion_pipeline.py:786-807): the dependency-free writer adds a 16-bit PCM track holding exactly the
    [offset, offset + duration) window of the file at ``sr`` - re-parsed here and compared sample by sample."""
    from pathlib import Path
    from stable_diffusion_videos_amd.audio import load_audio
    wav = Path(__file__).parent / "samples" / "choice.wav"
    frames = torch.randint(0, 255, (6, 3, 32, 48), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    sr, off, dur, fps = 44100, 1.5, 2.0, 3
    out = make_video_pyav(frames, audio_filepath=str(wav), fps=fps, audio_offset=off, audio_duration=dur, sr=sr,
                          output_filepath=tmp_path / "a.mp4")
    buf = open(out, "rb").read()
    tk = traks(buf)
    assert len(tk) == 2
    # movie header: 90 kHz timescale, duration = the longer track (6 frames at 3 fps = 2 s = the audio window)
    lo, hi = find(buf, ("moov", "mvhd"))
    _, _, _, timescale, duration = struct.unpack(">IIIII", buf[lo:lo + 20])
    assert timescale == 90000 and duration == 180000
    assert struct.unpack(">I", buf[hi - 4:hi])[0] == 3                    # next_track_ID
    # handler types
    kinds = []
    for a, b in tk:
        lo, hi = find_in(buf, a, b, ("mdia", "hdlr"))
        kinds.append(buf[lo + 8:lo + 12])
    assert kinds == [b"vide", b"soun"]
    a, b = tk[1]
    lo, hi = find_in(buf, a, b, ("mdia", "mdhd"))
    _, _, _, a_timescale, a_duration = struct.unpack(">IIIII", buf[lo:lo + 20])
    assert a_timescale == sr and a_duration == int(sr * dur)
    stbl = ("mdia", "minf", "stbl")
    lo, hi = find_in(buf, a, b, stbl + ("stsd",))
    entry = parse_boxes(buf, lo + 8, hi)[0]
    assert entry[0] == "ipcm"
    channels, bits = struct.unpack(">HH", buf[entry[1] + 16:entry[1] + 20])
    assert (channels, bits) == (1, 16)
    assert struct.unpack(">I", buf[entry[1] + 24:entry[1] + 28])[0] == sr << 16
    sub = {k: (x, y) for k, x, y in parse_boxes(buf, entry[1] + 28, entry[2])}
    assert set(sub) == {"pcmC", "chnl"} and buf[sub["pcmC"][0] + 4:sub["pcmC"][0] + 6] == bytes([1, 16])
    lo, hi = find_in(buf, a, b, stbl + ("stsz",))
    _, sample_size, count = struct.unpack(">III", buf[lo:lo + 12])
    assert (sample_size, count) == (2, int(sr * dur))
    lo, hi = find_in(buf, a, b, stbl + ("stco",))
    offset = struct.unpack(">III", buf[lo:lo + 12])[2]
    mdat = next(x for x in parse_boxes(buf) if x[0] == "mdat")
    assert mdat[1] < offset and offset + 2 * count == mdat[2]            # the PCM samples close the mdat box
    pcm = np.frombuffer(buf[offset:offset + 2 * count], dtype="<i2").astype(np.float64) / 32767.0
    ref, _ = load_audio(wav, sr=sr, mono=True, offset=off, duration=dur)
    assert ref.shape[0] == count and np.abs(pcm - np.clip(ref, -1, 1)).max() <= 1.0 / 32767.0
    assert np.abs(pcm).max() > 0.05                                      # (the window is not silence)
    # the video samples still start where the video track says
    va, vb = tk[0]
    lo, hi = find_in(buf, va, vb, stbl + ("stco",))
    v_off = struct.unpack(">III", buf[lo:lo + 12])[2]
    assert buf[v_off:v_off + 2] == b"\xff\xd8"


def test_h264_aac_path_when_pyav_is_present(tmp_path):
    """The reference's encoding (utils.py:69-128: ``torchvision.io.write_video(..., options={"crf": "10", "pix_fmt": "yuv420p"},
    audio_codec="aac")``) is taken whenever torchvision + pyav are importable; neither is in this image, so this test skips
    here and runs wherever they exist: the file must then carry an H.264 video track (sample entry ``avc1``) instead of the
    Motion-JPEG fallback's ``mp4v``."""
    pytest.importorskip("av")
    pytest.importorskip("torchvision")
    from PIL import Image
    d = tmp_path / "frames"
    d.mkdir()
    rng = np.random.RandomState(0)
    for k in range(6):
        Image.fromarray(rng.randint(0, 255, (64, 64, 3), dtype=np.uint8)).save(d / f"frame{k:06d}.png")
    out = make_video_pyav(d, fps=6, output_filepath=tmp_path / "h264.mp4", glob_pattern="*.png")
    data = open(out, "rb").read()
    assert b"avc1" in data and b"mp4v" not in data

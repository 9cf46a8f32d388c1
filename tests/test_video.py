"""mp4 muxing (reference utils.py:69-128).  Without ffmpeg the built-in writer emits an ISO-BMFF
file with a Motion-JPEG track; the test re-parses the container and decodes the samples back."""
import io
import struct

import numpy as np
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

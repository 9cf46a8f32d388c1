"""mp4 muxing of the generated frames - ``make_video_pyav`` of the reference
(/root/reference/stable_diffusion_videos/utils.py:69-128: torchvision.io.write_video, libx264 crf 10, optional
AAC audio track).  SURVEY.md section 8(f) rank 2.

torchvision / pyav / ffmpeg are not installed in this image.  When they are, the reference encoding is used.  When
they are not, a dependency-free writer produces a standards-conforming ISO-BMFF (.mp4) file with a Motion-JPEG
video track (sample entry ``mp4v``, MPEG-4 objectTypeIndication 0x6C = JPEG, quality 95) so that ``walk()`` returns
the same ``{name}.mp4`` paths as the reference; H.264 compression and the audio track need ffmpeg and are skipped
with a warning in that mode.  Frames are read once each (the reference's pairwise ``torch.cat`` is O(n^2), :91-93).
"""
from __future__ import annotations

import io
import logging
import struct
from pathlib import Path
from typing import List, Union

import numpy as np
import torch

logger = logging.getLogger("stable_diffusion_videos_amd")


def _box(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I4s", 8 + len(payload), kind) + payload


def _full(kind: bytes, version: int, flags: int, payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def _descr(tag: int, payload: bytes) -> bytes:
    assert len(payload) < 128
    return bytes([tag, len(payload)]) + payload


def write_mjpeg_mp4(jpegs: List[bytes], width: int, height: int, fps: float, path: Union[str, Path]) -> str:
    """Minimal ISO base media file: ftyp | mdat (the JPEG frames) | moov (one video track, constant frame rate)."""
    n = len(jpegs)
    timescale = 90000
    delta = int(round(timescale / float(fps)))
    duration = n * delta
    ftyp = _box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isomiso2mp41")
    mdat = _box(b"mdat", b"".join(jpegs))
    data_offset = len(ftyp) + 8
    matrix = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration) + struct.pack(">IH", 0x10000, 0x100) +
                 b"\0" * 10 + matrix + b"\0" * 24 + struct.pack(">I", 2))
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration) + b"\0" * 8 + struct.pack(">HHHH", 0, 0, 0, 0) +
                 matrix + struct.pack(">II", width << 16, height << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration) + struct.pack(">HH", 0x55C4, 0))
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide") + b"\0" * 12 + b"VideoHandler\0")
    vmhd = _full(b"vmhd", 0, 1, struct.pack(">HHHH", 0, 0, 0, 0))
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
    total = sum(len(j) for j in jpegs)
    dec_cfg = _descr(0x04, bytes([0x6C, 0x11]) + struct.pack(">I", max(len(j) for j in jpegs))[1:] +
                     struct.pack(">II", int(total * 8 * fps / max(n, 1)), int(total * 8 * fps / max(n, 1))))
    esds = _full(b"esds", 0, 0, _descr(0x03, struct.pack(">HB", 1, 0) + dec_cfg + _descr(0x06, b"\x02")))
    mp4v = _box(b"mp4v", b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HH", width, height) +
                struct.pack(">II", 0x480000, 0x480000) + struct.pack(">I", 0) + struct.pack(">H", 1) + b"\0" * 32 +
                struct.pack(">Hh", 0x18, -1) + esds)
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1) + mp4v)
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1))
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", len(j)) for j in jpegs))
    stco = _full(b"stco", 0, 0, struct.pack(">II", 1, data_offset))
    stbl = _box(b"stbl", stsd + stts + stsc + stsz + stco)
    minf = _box(b"minf", vmhd + dinf + stbl)
    trak = _box(b"trak", tkhd + _box(b"mdia", mdhd + hdlr + minf))
    moov = _box(b"moov", mvhd + trak)
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(ftyp + mdat + moov)
    return str(path)


def _frames_uint8(frames_or_frame_dir, glob_pattern: str) -> List[np.ndarray]:
    from PIL import Image
    if isinstance(frames_or_frame_dir, (str, Path)):
        paths = sorted(Path(frames_or_frame_dir).glob(glob_pattern))                      # same ordering as reference :90
        return [np.asarray(Image.open(p).convert("RGB")) for p in paths]
    t = frames_or_frame_dir
    if t.ndim == 4 and t.shape[1] in (1, 3):                                              # (T, C, H, W) like the reference
        t = t.permute(0, 2, 3, 1)
    return [np.ascontiguousarray(x) for x in t.cpu().numpy().astype(np.uint8)]


def make_video_pyav(frames_or_frame_dir: Union[str, Path, torch.Tensor] = "./images", audio_filepath=None, fps: int = 30,
                    audio_offset: int = 0, audio_duration: int = 2, sr: int = 22050,
                    output_filepath: Union[str, Path] = "output.mp4", glob_pattern: str = "*.png"):
    """Write the frames (a directory of images or a (T,C,H,W) uint8 tensor) to ``output_filepath`` and return it."""
    output_filepath = str(output_filepath)
    frames = _frames_uint8(frames_or_frame_dir, glob_pattern)
    if not frames:
        raise ValueError(f"no frames found for {frames_or_frame_dir} / {glob_pattern}")
    try:
        from torchvision.io import write_video
    except Exception:
        write_video = None
    if write_video is not None:                                                           # reference encoding
        stack = torch.from_numpy(np.stack(frames))
        if audio_filepath:
            from .audio import load_audio
            audio, _ = load_audio(audio_filepath, sr=sr, mono=True, offset=audio_offset, duration=audio_duration)
            write_video(output_filepath, stack, fps=fps, audio_array=torch.from_numpy(audio)[None], audio_fps=sr,
                        audio_codec="aac", options={"crf": "10", "pix_fmt": "yuv420p"})
        else:
            write_video(output_filepath, stack, fps=fps, options={"crf": "10", "pix_fmt": "yuv420p"})
        return output_filepath
    from PIL import Image
    if audio_filepath:
        logger.warning("no ffmpeg/pyav in this environment: writing a Motion-JPEG mp4 WITHOUT the audio track")
    jpegs = []
    for fr in frames:
        buf = io.BytesIO()
        Image.fromarray(fr).save(buf, format="JPEG", quality=95, subsampling=2)
        jpegs.append(buf.getvalue())
    h, w = frames[0].shape[:2]
    return write_mjpeg_mp4(jpegs, w, h, fps, output_filepath)

"""mp4 muxing of the generated frames (reference: utils.py:69-128, torchvision.io.write_video + libx264).
SURVEY.md section 8(f) rank 2.  torchvision / pyav / ffmpeg are not installed in this image; when they are, the
reference behaviour is reproduced, otherwise the call fails with an explicit message (frames stay on disk)."""
from __future__ import annotations

from pathlib import Path
from typing import Union

import numpy as np
import torch


def make_video_pyav(frames_or_frame_dir: Union[str, Path, torch.Tensor] = "./images", audio_filepath=None, fps: int = 30,
                    audio_offset: int = 0, audio_duration: int = 2, sr: int = 22050,
                    output_filepath: Union[str, Path] = "output.mp4", glob_pattern: str = "*.png"):
    try:
        from torchvision.io import write_video
    except Exception as exc:  # pragma: no cover - depends on the image
        raise RuntimeError("make_video_pyav needs torchvision + pyav/ffmpeg, which are not installed here; the frames "
                           "are on disk - call walk(..., make_video=False) or mux them with ffmpeg") from exc
    from PIL import Image
    output_filepath = str(output_filepath)
    if isinstance(frames_or_frame_dir, (str, Path)):
        paths = sorted(Path(frames_or_frame_dir).glob(glob_pattern))
        frames = torch.from_numpy(np.stack([np.asarray(Image.open(p).convert("RGB")) for p in paths]))  # (T,H,W,C), O(n)
    else:
        frames = frames_or_frame_dir.permute(0, 2, 3, 1) if frames_or_frame_dir.shape[1] in (1, 3) else frames_or_frame_dir
    if audio_filepath:
        from .audio import load_audio
        audio, _ = load_audio(audio_filepath, sr=sr, mono=True, offset=audio_offset, duration=audio_duration)
        write_video(output_filepath, frames, fps=fps, audio_array=torch.from_numpy(audio)[None], audio_fps=sr,
                    audio_codec="aac", options={"crf": "10", "pix_fmt": "yuv420p"})
    else:
        write_video(output_filepath, frames, fps=fps, options={"crf": "10", "pix_fmt": "yuv420p"})
    return output_filepath

"""Host-side helpers of the walk path with the reference's names
(/root/reference/stable_diffusion_videos/utils.py): ``slerp`` (:42-66), ``get_timesteps_arr`` (:12-39),
``make_video_pyav`` (:69-128), ``pad_along_axis`` (:131-136) - plus the asynchronous frame writer that
takes PNG encoding (~50 ms/frame/core, serial with the GPU in the reference at :550-554) off the
critical path."""
from __future__ import annotations

import os
from concurrent.futures import Future, ThreadPoolExecutor
from pathlib import Path
from typing import List, Optional, Union

import numpy as np
import torch

from . import hip


def slerp(t: float, v0, v1, DOT_THRESHOLD: float = 0.9995):
    """Spherical interpolation of two WHOLE tensors, same contract as the reference (utils.py:42-66):
    torch in -> torch out on the input device.  GPU tensors go through the HIP kernels
    (``sdv_slerp_stats`` + ``sdv_slerp_batch``) with no host round trip; fp32 arithmetic (the reference's
    numpy round trip cannot do bf16 at all - SURVEY.md fact 5), result cast back to the input dtype.
    numpy inputs are interpolated with numpy exactly like the reference does."""
    if not isinstance(v0, torch.Tensor):
        v0 = np.asarray(v0)
        v1 = np.asarray(v1)
        dot = np.sum(v0 * v1 / (np.linalg.norm(v0) * np.linalg.norm(v1)))
        if np.abs(dot) > DOT_THRESHOLD:
            return (1 - t) * v0 + t * v1
        th = np.arccos(dot)
        return np.sin(th - th * t) / np.sin(th) * v0 + np.sin(th * t) / np.sin(th) * v1
    if not v0.is_cuda:
        raise hip.SdvHipError("slerp: torch inputs must be GPU tensors (the HIP path has no CPU fallback); "
                              "pass numpy arrays for host-side interpolation")
    dtype = v0.dtype
    a = v0.detach().to(torch.float32).contiguous()
    b = v1.detach().to(torch.float32).contiguous()
    stats = hip.slerp_stats(a, b)
    T = torch.tensor([float(t)], dtype=torch.float32, device=a.device)
    out = hip.slerp_batch(a, b, stats, T, C_=1, HW=a.numel(), to_hwc=False, dot_threshold=DOT_THRESHOLD)
    return out.reshape(v0.shape).to(dtype)


def pad_along_axis(array: np.ndarray, pad_size: int, axis: int = 0) -> np.ndarray:
    if pad_size <= 0:
        return array
    widths = [(0, 0)] * array.ndim
    widths[axis] = (0, pad_size)
    return np.pad(array, pad_width=widths, mode="constant", constant_values=0)


# ------------------------------------------------------------------------------------------------
# asynchronous frame writer
# ------------------------------------------------------------------------------------------------
class FrameWriter:
    """Thread pool that PNG-encodes and writes frames while the GPU works on the next batch.  zlib
    releases the GIL, so threads scale across the host cores."""

    def __init__(self, workers: Optional[int] = None):
        # this rank's share of the host cores, minus the thread that drives the GPU (8 ranks x (cpu_count - 1) PNG threads on one
        # host would oversubscribe it eight-fold)
        from .parallel import host_threads_per_rank
        workers = workers or int(os.environ.get("SDV_WRITER_THREADS", max(2, host_threads_per_rank() - 1)))
        self.workers = workers
        self.pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="sdv-png")
        self.pending: List[Future] = []

    @staticmethod
    def _save(image, path: Path, upsampler):
        """Encode into ``<name>.part`` and rename: a killed run never leaves a truncated frame that ``resume`` (which
        only looks at complete ``frame%06d`` files) or the video mux would pick up."""
        from PIL import Image
        if upsampler is not None:
            image = upsampler(image)
        path = Path(path)
        fmt = Image.registered_extensions().get(path.suffix.lower())
        if fmt is None:
            image.save(path)                     # unknown extension: let PIL raise its own error
            return
        tmp = path.with_name(path.name + ".part")
        image.save(tmp, format=fmt)
        os.replace(tmp, path)

    def submit(self, image, path: Path, upsampler=None):
        self.pending.append(self.pool.submit(self._save, image, path, upsampler))

    def drain(self):
        pending, self.pending = self.pending, []
        for f in pending:
            f.result()

    def close(self):
        self.drain()
        self.pool.shutdown(wait=True)


def numpy_to_pil(images: np.ndarray):
    """uint8 NHWC (already rounded on the GPU) or float NHWC in [0,1] -> list of PIL images
    (diffusers ``numpy_to_pil``, used at stable_diffusion_pipeline.py:450)."""
    from PIL import Image
    if images.ndim == 3:
        images = images[None]
    if images.dtype != np.uint8:
        images = (images * 255).round().astype("uint8")
    return [Image.fromarray(im) for im in images]


# ------------------------------------------------------------------------------------------------
# audio -> T and video muxing: section-8(f) rows; their third-party dependencies are not installed here
# ------------------------------------------------------------------------------------------------
def get_timesteps_arr(audio_filepath, offset, duration, fps=30, margin=1.0, smooth=0.0):
    """Audio-driven interpolation schedule (utils.py:12-39)."""
    from .audio import get_timesteps_arr as _impl
    return _impl(audio_filepath, offset, duration, fps=fps, margin=margin, smooth=smooth)


def make_video_pyav(frames_or_frame_dir: Union[str, Path, torch.Tensor] = "./images", audio_filepath=None, fps: int = 30,
                    audio_offset: int = 0, audio_duration: int = 2, sr: int = 22050,
                    output_filepath: Union[str, Path] = "output.mp4", glob_pattern: str = "*.png"):
    """mp4 muxing (utils.py:69-128).  Needs torchvision/pyav + ffmpeg, which this image lacks; the frames
    on disk are the product of the hot path, so ``walk(make_video=False)`` is fully functional without it."""
    from .video import make_video_pyav as _impl
    return _impl(frames_or_frame_dir, audio_filepath=audio_filepath, fps=fps, audio_offset=audio_offset,
                 audio_duration=audio_duration, sr=sr, output_filepath=output_filepath, glob_pattern=glob_pattern)

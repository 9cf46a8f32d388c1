"""Real-ESRGAN x4 upsampling - host-side mirror of the reference's ``RealESRGANModel``
(/root/reference/stable_diffusion_videos/upsampling.py:13-99); SURVEY.md section 8(f) rank 3.

Same constructor / ``forward`` / ``from_pretrained`` / ``upsample_imagefolder`` surface; the network runs on the HIP
kernels (``esrgan.RRDBNetEngine``) instead of ``realesrgan.RealESRGANer`` + ``basicsr``.  Differences, all stated:
  * bf16 storage / fp32 accumulation where the reference defaults to fp16 (``half=not fp32``);
  * there are no checkpoints offline: ``from_pretrained`` loads ``RealESRGAN_x4plus.pth`` from a local path (or
    ``$SDV_ESRGAN_PATH``) when one exists and otherwise builds seeded synthetic weights of the same architecture
    (and says so in the log) - the arithmetic is shape-identical;
  * ``tile > 0`` (RealESRGANer.tile_process) is not needed with 288 GB of HBM and is rejected loudly;
  * ``outscale != 4`` resamples the x4 result with PIL's Lanczos filter (the reference uses cv2.INTER_LANCZOS4;
    OpenCV is not installed here).
There is no CPU fallback: the frames are moved to the GPU and the HIP library must be present.
"""
from __future__ import annotations

import logging
import os
from pathlib import Path
from typing import Union

import numpy as np
import torch

from .config import RRDBNetConfig
from .weights import load_rrdbnet, rrdbnet_shapes, synthetic_state_dict

logger = logging.getLogger("stable_diffusion_videos_amd")


class RealESRGANModel:
    def __init__(self, model_path=None, tile=0, tile_pad=10, pre_pad=0, fp32=False, device=None, seed: int = 0):
        if tile:
            raise NotImplementedError("RealESRGANModel(tile>0): tiled inference is not implemented on the HIP path "
                                      "(a 512x512 frame needs ~2 GB of the 288 GB HBM un-tiled)")
        if pre_pad:
            raise NotImplementedError("RealESRGANModel(pre_pad>0) is not implemented on the HIP path")
        self.cfg = RRDBNetConfig()
        self.scale = self.cfg.scale
        shapes = rrdbnet_shapes(self.cfg)
        if model_path is not None and Path(model_path).exists():
            self.state_dict_ = load_rrdbnet(model_path, shapes)
            self.synthetic = False
        else:
            logger.warning("Real-ESRGAN weights %s not found (no network here): using seeded synthetic RRDBNet weights",
                           model_path)
            self.state_dict_ = synthetic_state_dict(shapes, seed=seed)
            self.synthetic = True
        self.engine = None
        self.device = None
        if device is not None:
            self.to(device)

    # nn.Module-like surface the pipeline uses (stable_diffusion_pipeline.py:516)
    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("RealESRGANModel runs on the MI355X HIP path only (no CPU fallback); got device " + str(device))
        if self.engine is None or self.device != device:
            from .esrgan import RRDBNetEngine
            self.engine = RRDBNetEngine(self.cfg, self.state_dict_, device)
            self.device = device
        return self

    def upsample_u8(self, frames_u8: torch.Tensor) -> torch.Tensor:
        """GPU-resident fast path used by the walk: uint8 RGB NHWC [n, H, W, 3] -> uint8 [n, 4H, 4W, 3]."""
        if self.engine is None:
            raise RuntimeError("RealESRGANModel: call .to('cuda') first")
        return self.engine(frames_u8.to(self.device))[0]

    def forward(self, image, outscale=4, convert_to_pil=True):
        """Upsample an image array (RGB float in [0, 1], HxWx3) or an image path.  Returns a PIL image, or - with
        ``convert_to_pil=False`` - a BGR uint8 array like the reference (upsampling.py:30-54)."""
        from PIL import Image
        if isinstance(image, (str, Path)):
            rgb = np.asarray(Image.open(image).convert("RGB"))
        else:
            rgb = (np.asarray(image) * 255).round().astype("uint8")
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise ValueError(f"expected an HxWx3 image, got shape {rgb.shape}")
        out = self.upsample_u8(torch.from_numpy(np.ascontiguousarray(rgb))[None])[0].cpu().numpy()
        if outscale is not None and float(outscale) != float(self.scale):
            h, w = rgb.shape[:2]
            out = np.asarray(Image.fromarray(out).resize((int(w * outscale), int(h * outscale)), Image.LANCZOS))
        if convert_to_pil:
            return Image.fromarray(out)
        return out[:, :, ::-1]

    __call__ = forward

    @classmethod
    def from_pretrained(cls, model_name_or_path="nateraw/real-esrgan", **kwargs):
        """Local ``RealESRGAN_x4plus.pth`` file, a directory holding one, ``$SDV_ESRGAN_PATH``, or (offline, nothing
        found) synthetic weights.  The reference downloads from the hub (upsampling.py:74-77); there is no network."""
        cands = [Path(model_name_or_path), Path(model_name_or_path) / "RealESRGAN_x4plus.pth"]
        if os.environ.get("SDV_ESRGAN_PATH"):
            cands.insert(0, Path(os.environ["SDV_ESRGAN_PATH"]))
        file = next((p for p in cands if p.is_file()), None)
        return cls(file, **kwargs)

    def upsample_imagefolder(self, in_dir, out_dir, suffix="out", outfile_ext=".png", recursive=False, force=False):
        in_dir, out_dir = Path(in_dir), Path(out_dir)
        if not in_dir.exists():
            raise FileNotFoundError(f"Provided input directory {in_dir} does not exist")
        out_dir.mkdir(exist_ok=True, parents=True)
        paths = sorted(p for p in (in_dir.rglob("*") if recursive else in_dir.glob("*"))
                       if p.suffix.lower() in (".png", ".jpg", ".jpeg"))
        for i, image in enumerate(paths):
            out_filepath = out_dir / (str(image.relative_to(in_dir).with_suffix("")) + suffix + outfile_ext)
            if not force and out_filepath.exists():
                logger.info("[%d/%d] %s already exists, skipping. To avoid skipping, pass force=True.", i, len(paths), out_filepath)
                continue
            logger.info("[%d/%d] upscaling %s", i, len(paths), image)
            im = self(str(image))
            out_filepath.parent.mkdir(parents=True, exist_ok=True)
            im.save(out_filepath)


PipelineRealESRGAN = RealESRGANModel

"""CPU oracle (test infrastructure) for the Real-ESRGAN x4 upsampling step of the walk.

Reference call sites
    /root/reference/stable_diffusion_videos/upsampling.py:25-28   RRDBNet(3, 3, num_feat=64, num_block=23,
                                                                  num_grow_ch=32, scale=4) inside
                                                                  RealESRGANer(scale=4, tile=0, tile_pad=10,
                                                                  pre_pad=0, half=not fp32)
    /root/reference/stable_diffusion_videos/upsampling.py:30-54   forward(): RGB float [0,1] -> uint8 BGR ->
                                                                  upsampler.enhance(img, outscale=4) -> RGB PIL
    /root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:513-516, :552   the hook in the walk

The arithmetic lives in two third-party packages that are neither vendored under /root/reference nor
installed here: ``basicsr`` (``basicsr.archs.rrdbnet_arch.RRDBNet``) and ``realesrgan``
(``realesrgan.RealESRGANer``); ``pyproject.toml:20`` lists ``realesrgan`` without a version (basicsr comes in
as its dependency).  This module restates their published algorithms (ESRGAN, Wang et al. 2018;
Real-ESRGAN, Wang et al. 2021): **parity unpinned** - the reference holds no golden vectors for this step
(``tests/test_pipeline.py`` never sets ``upsample=True``).  Anchors: the published RealESRGAN_x4plus
parameter count (16,697,987) and state-dict key schema, both asserted in tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class RRDBNetConfig:
    num_in_ch: int = 3
    num_out_ch: int = 3
    num_feat: int = 64
    num_block: int = 23
    num_grow_ch: int = 32
    scale: int = 4


class ResidualDenseBlock(nn.Module):
    """Five 3x3 convs, each seeing the block input concatenated with every earlier growth output;
    LeakyReLU(0.2) after the first four; the block returns x5 * 0.2 + x."""

    def __init__(self, num_feat: int, num_grow_ch: int):
        super().__init__()
        self.conv1 = nn.Conv2d(num_feat, num_grow_ch, 3, 1, 1)
        self.conv2 = nn.Conv2d(num_feat + num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv3 = nn.Conv2d(num_feat + 2 * num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv4 = nn.Conv2d(num_feat + 3 * num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv5 = nn.Conv2d(num_feat + 4 * num_grow_ch, num_feat, 3, 1, 1)

    def forward(self, x):
        x1 = F.leaky_relu(self.conv1(x), 0.2)
        x2 = F.leaky_relu(self.conv2(torch.cat((x, x1), 1)), 0.2)
        x3 = F.leaky_relu(self.conv3(torch.cat((x, x1, x2), 1)), 0.2)
        x4 = F.leaky_relu(self.conv4(torch.cat((x, x1, x2, x3), 1)), 0.2)
        x5 = self.conv5(torch.cat((x, x1, x2, x3, x4), 1))
        return x5 * 0.2 + x


class RRDB(nn.Module):
    """Residual-in-residual dense block: three dense blocks, out * 0.2 + x."""

    def __init__(self, num_feat: int, num_grow_ch: int):
        super().__init__()
        self.rdb1 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb2 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb3 = ResidualDenseBlock(num_feat, num_grow_ch)

    def forward(self, x):
        out = self.rdb3(self.rdb2(self.rdb1(x)))
        return out * 0.2 + x


class RRDBNet(nn.Module):
    """scale-4 generator: conv_first -> 23 RRDB -> conv_body (+ trunk residual) -> 2 x (nearest-2x, conv,
    LeakyReLU) -> conv_hr, LeakyReLU -> conv_last.  (scale 2 / 1 variants pixel-unshuffle the input first; the
    reference only ever builds scale 4.)"""

    def __init__(self, cfg: RRDBNetConfig = RRDBNetConfig()):
        super().__init__()
        if cfg.scale != 4:
            raise ValueError("only the scale-4 generator is on the reference path (upsampling.py:25)")
        self.cfg = cfg
        nf = cfg.num_feat
        self.conv_first = nn.Conv2d(cfg.num_in_ch, nf, 3, 1, 1)
        self.body = nn.Sequential(*[RRDB(nf, cfg.num_grow_ch) for _ in range(cfg.num_block)])
        self.conv_body = nn.Conv2d(nf, nf, 3, 1, 1)
        self.conv_up1 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.conv_up2 = nn.Conv2d(nf, nf, 3, 1, 1)
        self.conv_hr = nn.Conv2d(nf, nf, 3, 1, 1)
        self.conv_last = nn.Conv2d(nf, cfg.num_out_ch, 3, 1, 1)

    def forward(self, x):
        feat = self.conv_first(x)
        feat = feat + self.conv_body(self.body(feat))
        feat = F.leaky_relu(self.conv_up1(F.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
        feat = F.leaky_relu(self.conv_up2(F.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
        return self.conv_last(F.leaky_relu(self.conv_hr(feat), 0.2))


@torch.no_grad()
def enhance_rgb_u8(model: RRDBNet, img_u8: np.ndarray) -> np.ndarray:
    """RealESRGANer.enhance for a 3-channel 8-bit image with outscale == netscale == 4, tile == 0, pre_pad == 0,
    expressed RGB -> RGB (the reference flips RGB -> BGR before enhance (upsampling.py:44), enhance flips BGR -> RGB
    for the network and back afterwards, and :49 flips to RGB again: the flips cancel).

        img / 255 -> CHW float -> model -> clamp(0, 1) -> (x * 255).round() -> uint8 HWC

    (no mod-padding at scale 4; fp32 here where the reference defaults to half precision)."""
    if img_u8.dtype != np.uint8 or img_u8.ndim != 3 or img_u8.shape[2] != 3:
        raise ValueError("expected an HxWx3 uint8 image")
    x = torch.from_numpy(np.ascontiguousarray(img_u8)).float().div(255.0).permute(2, 0, 1).unsqueeze(0)
    y = model(x).squeeze(0).float().clamp_(0, 1).permute(1, 2, 0).numpy()
    return (y * 255.0).round().astype(np.uint8)

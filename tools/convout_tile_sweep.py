#!/usr/bin/env python
"""Time the typed-output convolutions (igemm out_mode, N <= 4 padded to one MFMA tile) per 4-wave tile:
UNet conv_out 320 -> 4 @64^2 (fp32 out) and VAE conv_out 128 -> 3 @512^2 (uint8 out).  usage: convout_tile_sweep.py [frames]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
hip.load()
for name, n, H, Cin, Cout, mode in (("unet conv_out 320->4 @64", 2 * B, 64, 320, 4, 1), ("vae conv_out 128->3 @512", B // 4, 512, 128, 3, 2)):
    x = torch.randn((n * H * H, Cin), device=dev).to(torch.bfloat16)
    w = (torch.randn((Cout, 9 * Cin), device=dev) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev)
    of = torch.empty((n * H * H, Cout), dtype=torch.float32, device=dev) if mode == 1 else None
    ou = torch.empty((n * H * H, Cout), dtype=torch.uint8, device=dev) if mode == 2 else None
    line = f"{name:28s} n={n:4d}"
    for tile in (10, 11, 2, 3, 1, 4):
        try:
            for _ in range(2):
                hip.conv3x3(x, w, b, nimg=n, H=H, W=H, out_mode=mode, out_f32=of, out_u8=ou, tile=tile)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                hip.conv3x3(x, w, b, nimg=n, H=H, W=H, out_mode=mode, out_f32=of, out_u8=ou, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            line += f"  t{tile}: {e0.elapsed_time(e1) / 5:7.3f} ms"
        except hip.SdvHipError as e:
            line += f"  t{tile}: n/a"
    print(line)

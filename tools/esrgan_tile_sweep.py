#!/usr/bin/env python
"""Tile sweep over the RRDBNet conv shapes (n frames of 512^2, dense-block strides) - calibrates the narrow-N rows of the
cost model in sdv_gemm.hip."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402

n, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 4), 512, 512
M = n * H * W
buf = torch.randn((M, 192), device="cuda").to(torch.bfloat16)
dst = torch.zeros((M, 192), device="cuda", dtype=torch.bfloat16)
for kp, cout in [(64, 32), (128, 32), (192, 32), (192, 64), (64, 64)]:
    w = (torch.randn((cout, 9 * kp), device="cuda") * (9 * kp) ** -0.5).to(torch.bfloat16)
    b = torch.zeros(cout, device="cuda")
    line = f"Kp={kp:4d} Cout={cout:3d}:"
    for tile in (0, 1, 2, 3, 8, 10, 11):
        def run():
            hip.conv3x3(buf[:, :kp], w, b, nimg=n, H=H, W=W, out=dst[:, 64:64 + cout], epi=3, tile=tile)
        run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        line += f"  t{tile}: {2.0 * M * cout * 9 * kp / ms / 1e9:5.0f}"
    print(line + "   (TFLOP/s, unpadded)")

#!/usr/bin/env python
"""A handful of igemm launches on the UNet's shapes - the workload for `rocprofv3 --pmc ...` passes over single kernels.
usage: gemm_pmc.py [tile ...]   (default tile 6)"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402

tiles = [int(t) for t in sys.argv[1:]] or [6]
dev = torch.device("cuda")
hip.load()
nimg = 128


def mk(M, cin, cout, conv):
    x = (torch.randn((M, cin), device=dev) * 0.5).to(torch.bfloat16)
    k = 9 * cin if conv else cin
    w = (torch.randn((cout, k), device=dev) * k ** -0.5).to(torch.bfloat16)
    return x, w, torch.randn(cout, device=dev), torch.randn((M, cout), device=dev).to(torch.bfloat16), torch.empty((M, cout), dtype=torch.bfloat16, device=dev)


for t in tiles:
    x, w, b, r, o = mk(nimg * 4096, 320, 320, False)
    for _ in range(3):
        hip.linear(x, w, b, residual=r, out=o, tile=t)
    x, w, b, r, o = mk(nimg * 4096, 320, 320, True)
    for _ in range(3):
        hip.conv3x3(x, w, b, nimg=nimg, H=64, W=64, residual=r, out=o, tile=t)
    x, w, b, r, o = mk(nimg * 256, 2560, 1280, True)
    for _ in range(3):
        hip.conv3x3(x, w, b, nimg=nimg, H=16, W=16, residual=r, out=o, tile=t)
torch.cuda.synchronize()
print("done")

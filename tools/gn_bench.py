#!/usr/bin/env python
"""GroupNorm(+SiLU) stats / apply bandwidth on the UNet's and the VAE's shapes.  usage: gn_bench.py [nimg]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import EventProfiler  # noqa: E402
from stable_diffusion_videos_amd import hip  # noqa: E402
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda")
tot = {}
for HW, C in ((4096, 320), (4096, 640), (1024, 640), (1024, 1280), (256, 1280), (256, 2560), (64, 1280)):
    x = torch.randn((nimg * HW, C), device=dev).to(torch.bfloat16)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    hip.groupnorm(x, g, b, nimg=nimg, HW=HW, groups=32, eps=1e-5, silu=True)
    prof = EventProfiler()
    hip.LAUNCH_HOOK = prof
    for _ in range(5):
        hip.groupnorm(x, g, b, nimg=nimg, HW=HW, groups=32, eps=1e-5, silu=True)
    hip.LAUNCH_HOOK = None
    s = prof.summary()
    print(f"HW={HW:5d} C={C:5d}: " + "  ".join(f"{k} {v['ms'] / 5 * 1e3:7.1f} us {v['bytes'] / v['ms'] / 1e6:6.0f} GB/s" for k, v in s.items()))

#!/usr/bin/env python
"""FETCH_SIZE workload: a few shapes under the current env knobs.  usage: gemm_pmc2.py"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402
dev = torch.device("cuda")
hip.load()
nimg = 128


def mk(M, cin, cout, conv):
    x = (torch.randn((M, cin), device=dev) * 0.5).to(torch.bfloat16)
    k = 9 * cin if conv else cin
    w = (torch.randn((cout, k), device=dev) * k ** -0.5).to(torch.bfloat16)
    return x, w, torch.randn(cout, device=dev), torch.empty((M, cout), dtype=torch.bfloat16, device=dev)


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


x, w, b, o = mk(nimg * 256, 5120, 1280, False)
ms = t(lambda: hip.linear(x, w, b, out=o, tile=6)); print(f"gemm 5120->1280 @16: {2.0*x.shape[0]*5120*1280/ms/1e9:.0f} TF")
x, w, b, o = mk(nimg * 256, 1280, 1280, True)
ms = t(lambda: hip.conv3x3(x, w, b, nimg=nimg, H=16, W=16, out=o, tile=6)); print(f"conv 1280->1280 @16: {18.0*x.shape[0]*1280*1280/ms/1e9:.0f} TF")
x, w, b, o = mk(nimg * 4096, 320, 320, True)
ms = t(lambda: hip.conv3x3(x, w, b, nimg=nimg, H=64, W=64, out=o, tile=6)); print(f"conv 320->320 @64: {18.0*x.shape[0]*320*320/ms/1e9:.0f} TF")
x, w, b, o = mk(nimg * 4096, 320, 320, False)
ms = t(lambda: hip.linear(x, w, b, out=o, tile=6)); print(f"gemm 320->320 @64: {2.0*x.shape[0]*320*320/ms/1e9:.0f} TF")
x, w, b, o = mk(nimg * 4096, 1280, 320, False)
ms = t(lambda: hip.linear(x, w, b, out=o, tile=6)); print(f"gemm 1280->320 @64: {2.0*x.shape[0]*1280*320/ms/1e9:.0f} TF")

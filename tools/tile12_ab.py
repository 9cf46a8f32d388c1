#!/usr/bin/env python
"""Experiment: the 256 x 320 tile with FOUR waves (one per SIMD, 4 x 5 MFMA tiles each: 9 LDS fragments per 20 MFMAs instead of 7 per
10) against the shipped 8-wave tile, on long-K shapes where the epilogue is a small share.
(needs a library built from tools/experiments/sdv_gemm_four_wave_256x320.diff.txt with -DSDV_EXPERIMENT_TILE12)
usage: SDV_HIP_LIB=tools/ubench/libsdv_gemm_tile12.so python tools/tile12_ab.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402

BF16 = torch.bfloat16


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda")
    nimg = 256
    for label, H, cin, cout, conv in (("conv 1280->1280 @16", 16, 1280, 1280, True), ("conv 640->640 @32", 32, 640, 640, True),
                                      ("conv 320->320 @64", 64, 320, 320, True), ("gemm 2560->640 @32", 32, 2560, 640, False),
                                      ("gemm 640->640 @32", 32, 640, 640, False)):
        M = nimg * H * H
        kk = 9 * cin if conv else cin
        x = (torch.randn((M, cin), device=dev) * 0.5).to(BF16)
        w = (torch.randn((cout, kk), device=dev) * kk ** -0.5).to(BF16)
        bias = torch.randn(cout, device=dev)
        outs = {}
        for tile in (6, 12):
            out = torch.zeros((M, cout), dtype=BF16, device=dev)
            fn = (lambda: hip.conv3x3(x, w, bias, nimg=nimg, H=H, W=H, out=out, tile=tile)) if conv else (lambda: hip.linear(x, w, bias, out=out, tile=tile))
            t = timed(fn)
            outs[tile] = out
            print(f"{label:22s} tile {tile:2d}: {t:.3f} ms  {2.0 * M * cout * kk / t / 1e9:7.1f} TFLOP/s")
        print(f"{'':22s} identical: {bool(torch.equal(outs[6], outs[12]))}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Time every compiled igemm tile on the UNet's layer shapes (HIP events) - calibrates the tile cost model in
sdv_gemm_bf16.  Usage on the GPU box:  python tools/tile_sweep.py [nimg] > gpurun_out/tile_sweep.txt"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402

TILES = {1: "128x128", 2: "128x64", 3: "64x64", 6: "256x320", 7: "256x256", 8: "256x128", 9: "128x320"}


def bench(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda")
    hip.load()
    shapes = []   # (label, mode, H, Cin, Cout)   mode 0 = dense with M = nimg*H*H, K = Cin, N = Cout
    for H, C in ((64, 320), (32, 640), (16, 1280), (8, 1280)):
        shapes.append((f"conv {C}->{C} @{H}", 1, H, C, C))
        shapes.append((f"gemm {C}->{C} @{H}", 0, H, C, C))
        shapes.append((f"gemm {4*C}->{C} @{H}", 0, H, 4 * C, C))
        shapes.append((f"gemm {C}->{2*C} @{H}", 0, H, C, 2 * C))
    shapes.append(("conv 960->320 @64", 1, 64, 960, 320))
    shapes.append(("conv 2560->1280 @16", 1, 16, 2560, 1280))
    print(f"nimg={nimg}")
    for label, mode, H, cin, cout in shapes:
        M = nimg * H * H
        x = (torch.randn((M, cin), device=dev) * 0.5).to(torch.bfloat16)
        if mode == 0:
            w = (torch.randn((cout, cin), device=dev) * cin ** -0.5).to(torch.bfloat16)
            flops = 2.0 * M * cin * cout
        else:
            w = (torch.randn((cout, 9 * cin), device=dev) * (9 * cin) ** -0.5).to(torch.bfloat16)
            flops = 18.0 * M * cin * cout
        bias = torch.randn(cout, device=dev)
        res = torch.randn((M, cout), device=dev).to(torch.bfloat16)
        out = torch.empty((M, cout), dtype=torch.bfloat16, device=dev)
        row = []
        for t, name in TILES.items():
            if mode == 0:
                fn = lambda: hip.linear(x, w, bias, residual=res, out=out, tile=t)
            else:
                fn = lambda: hip.conv3x3(x, w, bias, nimg=nimg, H=H, W=H, residual=res, out=out, tile=t)
            ms = bench(fn)
            row.append((flops / ms / 1e9, name))
        auto = bench((lambda: hip.linear(x, w, bias, residual=res, out=out)) if mode == 0 else
                     (lambda: hip.conv3x3(x, w, bias, nimg=nimg, H=H, W=H, residual=res, out=out)))
        best = max(row)
        print(f"{label:24s} M={M:7d} " + " ".join(f"{n}:{tf:6.0f}" for tf, n in row) +
              f" | auto {flops / auto / 1e9:6.0f} best {best[1]} {best[0]:.0f}")


if __name__ == "__main__":
    main()

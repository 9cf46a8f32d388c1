#!/usr/bin/env python
"""The transposed V^T projections of the UNet's self-attention (column-side LayerNorm fold, M = channels, N = tokens of one image,
batch = samples) on the tiles that carry them: 9 (128 x 320), 7 (256 x 256), 14 (320 x 256, round 4; selectable, not a cost-model
candidate - DESIGN.md row 2b) and what the cost model picks.
usage: python tools/vt_ab.py [nimg] [rounds]"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402


def timed(fn, reps=3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda")
    hip.load()
    print(f"nimg={nimg} rounds={rounds}   ms median (TFLOP/s) per tile; 0 = cost model")
    for H, C in ((64, 320), (32, 640), (16, 1280)):
        L = H * H
        g = torch.Generator(device=dev).manual_seed(3)
        x = (torch.randn((nimg * L, C), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn((C, C), device=dev, generator=g) * C ** -0.5).to(torch.bfloat16)
        st = torch.stack([torch.randn(nimg * L, device=dev, generator=g) * 0.1, 1.0 + 0.1 * torch.rand(nimg * L, device=dev, generator=g)], 1).contiguous()
        sv, tv = torch.randn(C, device=dev, generator=g), torch.randn(C, device=dev, generator=g)
        vt = torch.empty((nimg, C, L), dtype=torch.bfloat16, device=dev)
        kw = dict(M=C, N=L, K=C, ldx=C, ldw=C, ldc=L, batch=nimg, sX=0, sW=L * C, sC=C * L, bias=tv, bias_mode=2, ln=(st, sv), ln_side=2)
        tiles = (0, 9, 7, 14)
        ref = None
        for t in tiles:
            hip.gemm(w, x, vt, tile=t, **kw)
            torch.cuda.synchronize()
            if ref is None:
                ref = vt.clone()
            else:
                assert torch.equal(ref, vt), (H, t)
        ms = {t: [] for t in tiles}
        for _ in range(rounds):
            for t in tiles:
                ms[t].append(timed(lambda t=t: hip.gemm(w, x, vt, tile=t, **kw)))
        flops = 2.0 * nimg * L * C * C
        # the same GEMM without the fold (per-row bias only): separates the K loop / store path from the fold's epilogue
        kw0 = {k: v for k, v in kw.items() if k not in ("ln", "ln_side")}
        ms0 = {t: [] for t in tiles}
        for _ in range(rounds):
            for t in tiles:
                ms0[t].append(timed(lambda t=t: hip.gemm(w, x, vt, tile=t, **kw0)))
        print(f"    plain {C}x{C} @{H}  " + "   ".join(f"tile {t:2d}: {statistics.median(ms0[t]):.3f} ms ({flops / statistics.median(ms0[t]) / 1e9:5.0f})" for t in tiles), flush=True)
        print(f"V^T {C}x{C} @{H}  " + "   ".join(f"tile {t:2d}: {statistics.median(ms[t]):.3f} ms ({flops / statistics.median(ms[t]) / 1e9:5.0f})" for t in tiles), flush=True)


if __name__ == "__main__":
    main()

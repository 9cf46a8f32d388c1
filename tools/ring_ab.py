#!/usr/bin/env python
"""A/B of the double-buffered 256 x 320 tile (6) against the persistent 4-slot ring (12) on the transformer-block GEMM
shapes of a 256-sample UNet forward (B = 128 frames with CFG): interleaved rounds in ONE process, median + min per arm
(guide rule 24).  usage: python tools/ring_ab.py [nimg] [rounds]"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402
from stable_diffusion_videos_amd.weights import geglu_interleave  # noqa: E402


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
    cases = []
    for H, C in ((64, 320), (32, 640), (16, 1280)):
        M = nimg * H * H
        cases += [(f"proj  {C}->{C} @{H} bias", M, C, C, "bias"), (f"out   {C}->{C} @{H} +res", M, C, C, "res"),
                  (f"qk    {C}->{2*C} @{H} bias", M, C, 2 * C, "bias"), (f"ff1   {C}->{8*C} @{H} geglu", M, C, 8 * C, "geglu"),
                  (f"ff2   {4*C}->{C} @{H} +res", M, 4 * C, C, "res")]
    tiles = (6, 12)
    print(f"nimg={nimg} rounds={rounds}   TFLOP/s median (min..max) per tile; GB/s = (X + out [+ res]) bytes / median time")
    for label, M, K, N, kind in cases:
        x = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn((N, K), device=dev) * K ** -0.5)
        bias = torch.randn(N, device=dev)
        nout = N // 2 if kind == "geglu" else N
        res = torch.randn((M, nout), device=dev).to(torch.bfloat16) if kind == "res" else None
        out = torch.empty((M, nout), dtype=torch.bfloat16, device=dev)
        if kind == "geglu":
            w, bias = geglu_interleave(w), geglu_interleave(bias)
        w = w.to(torch.bfloat16)
        fns = {t: (lambda t=t: hip.linear(x, w, bias, residual=res, out=out, epi=1 if kind == "geglu" else 0, tile=t)) for t in tiles}
        ref = None
        for t in tiles:
            fns[t]()
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                assert torch.equal(out, ref), f"{label}: tile {t} differs from tile {tiles[0]}"
        ms = {t: [] for t in tiles}
        for _ in range(rounds):
            for t in tiles:
                ms[t].append(timed(fns[t]))
        flops = 2.0 * M * N * K
        nbytes = 2.0 * M * (K + nout + (nout if res is not None else 0))
        row = []
        for t in tiles:
            med = statistics.median(ms[t])
            row.append(f"tile {t:2d}: {flops / med / 1e9:6.0f} ({flops / max(ms[t]) / 1e9:5.0f}..{flops / min(ms[t]) / 1e9:5.0f}) "
                       f"{nbytes / med / 1e6:5.0f} GB/s")
        gain = statistics.median(ms[tiles[0]]) / statistics.median(ms[tiles[1]])
        print(f"{label:28s} M={M:8d}  " + "   ".join(row) + f"   ring/dbuf = {gain:.3f}")
        del x, w, out, res


if __name__ == "__main__":
    main()

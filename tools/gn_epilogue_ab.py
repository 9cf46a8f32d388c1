#!/usr/bin/env python
"""What the GroupNorm-statistics epilogue (sdv_gemm_args.gn_out, the FEAT 4 variant of the 8-wave tiles) costs the launch that
carries it: the ResBlock conv shapes (bias and +residual) and the GroupNorm-feeding GEMMs of a 256-sample forward with and without
gn_out, interleaved rounds in ONE process, against the statistics pass (gn_stats) it replaces.
usage: [SDV_HIP_LIB=...] python tools/gn_epilogue_ab.py [nimg] [rounds]"""
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
    cases = [("conv 320->320 @64 bias", True, 64, 320, 320, False), ("conv 320->320 @64 +res", True, 64, 320, 320, True),
             ("conv 640->640 @32 bias", True, 32, 640, 640, False), ("conv 640->640 @32 +res", True, 32, 640, 640, True),
             ("conv 1280->1280 @16 bias", True, 16, 1280, 1280, False), ("conv 1280->1280 @16 +res", True, 16, 1280, 1280, True),
             ("conv 1280->1280 @8 +res", True, 8, 1280, 1280, True),
             ("proj_out 320->320 @64 +res", False, 64, 320, 320, True), ("proj_out 640->640 @32 +res", False, 32, 640, 640, True)]
    print(f"lib={hip._LIB_PATH} nimg={nimg} rounds={rounds}   ms median (min..max): plain | with gn_out | the gn_stats pass over the output")
    tot = [0.0, 0.0, 0.0]
    for label, conv, H, cin, cout, use_res in cases:
        M = nimg * H * H
        taps = 9 if conv else 1
        g = torch.Generator(device=dev).manual_seed(1)
        x = (torch.randn((M, cin), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn((cout, taps * cin), device=dev, generator=g) * (taps * cin) ** -0.5).to(torch.bfloat16)
        bias = torch.randn(cout, device=dev, generator=g)
        res = torch.randn((M, cout), device=dev, generator=g).to(torch.bfloat16) if use_res else None
        out = torch.empty((M, cout), dtype=torch.bfloat16, device=dev)
        gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)

        def run(gn):
            hip.gemm(x, w, out, M=M, N=cout, K=cin, ldx=cin, ldw=w.stride(0), ldc=cout, bias=bias, residual=res,
                     ldr=cout if use_res else 0, mode=1 if conv else 0, Hin=H, Win=H, Hout=H, Wout=H, tile=6, gn_hw=H * H if gn else 0)

        run(True)
        torch.cuda.synchronize()
        assert getattr(out, "_sdv_gn", None) is not None, "gn_out was not emitted"
        a = out.clone()
        run(False)
        torch.cuda.synchronize()
        assert torch.equal(a, out), label
        plain = out.clone()          # (carries no statistics: groupnorm runs its own pass)
        seen = []

        def stats_pass():
            hip.groupnorm(plain, gamma, beta, nimg=nimg, HW=H * H, groups=32, eps=1e-5, silu=True, out=a)

        hip.LAUNCH_HOOK = lambda kind, info, fn: (seen.append(kind), fn())
        stats_pass()
        hip.LAUNCH_HOOK = None
        torch.cuda.synchronize()
        # time of the statistics kernel alone = (stats + apply) - apply, measured as two separate loops
        ms = {"plain": [], "gn": [], "stats+apply": []}
        for _ in range(rounds):
            ms["plain"].append(timed(lambda: run(False)))
            ms["gn"].append(timed(lambda: run(True)))
            ms["stats+apply"].append(timed(stats_pass))
        med = {k: statistics.median(v) for k, v in ms.items()}
        tot[0] += med["plain"]
        tot[1] += med["gn"]
        tot[2] += med["stats+apply"]
        print(f"{label:28s} M={M:8d}  {med['plain']:7.3f} ({min(ms['plain']):.3f}..{max(ms['plain']):.3f}) | {med['gn']:7.3f} "
              f"({min(ms['gn']):.3f}..{max(ms['gn']):.3f})  gn/plain = {med['gn'] / med['plain']:.3f} (+{med['gn'] - med['plain']:.3f} ms) | "
              f"{'+'.join(seen)} {med['stats+apply']:.3f} ms", flush=True)
        del x, w, out, res, a, plain
    print(f"sums: plain {tot[0]:.3f} ms, with gn_out {tot[1]:.3f} ms (+{tot[1] - tot[0]:.3f}); gn_stats + gn_apply passes over the same outputs {tot[2]:.3f} ms")


if __name__ == "__main__":
    main()

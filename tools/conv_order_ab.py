#!/usr/bin/env python
"""A/B of the conv K-loop order (sdv_hip.h k_order): 0 = tap-major, 1 = channel-major, on the conv3x3 shapes of a 256-sample UNet
forward and of a 128-frame VAE decode.  Interleaved rounds in one process, median (min..max) TFLOP/s per arm.
usage: python tools/conv_order_ab.py [nimg] [rounds]          timing
       python tools/conv_order_ab.py pmc <order>              two launches per shape with ONE order (for rocprofv3 --pmc FETCH_SIZE)"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402


def timed(fn, reps=2):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


import os
ORDERS = tuple(int(x) for x in os.environ.get("SDV_AB_ORDERS", "0,1").split(","))      # e.g. SDV_AB_ORDERS=2,0: unfused vs fused


def main():
    pmc = len(sys.argv) > 1 and sys.argv[1] == "pmc"
    nimg = 256 if pmc or len(sys.argv) < 2 else int(sys.argv[1])
    rounds = 5 if pmc or len(sys.argv) < 3 else int(sys.argv[2])
    dev = torch.device("cuda")
    hip.load()
    # (label, images, H, C1, C2, Cout, residual)
    cases = [("unet 320->320 @64", nimg, 64, 320, 0, 320, True), ("unet 640+320->320 @64", nimg, 64, 640, 320, 320, False),
             ("unet 640->640 @32", nimg, 32, 640, 0, 640, True), ("unet 1280+640->640 @32", nimg, 32, 1280, 640, 640, False),
             ("unet 1280->1280 @16", nimg, 16, 1280, 0, 1280, True), ("unet 1280->1280 @8", nimg, 8, 1280, 0, 1280, True),
             ("vae 512->512 @128", nimg // 8, 128, 512, 0, 512, True), ("vae 256->256 @256", nimg // 16, 256, 256, 0, 256, True),
             ("vae 128->128 @512", nimg // 32, 512, 128, 0, 128, True)]
    if pmc:
        order = int(sys.argv[2])
        cases = cases[:1] + cases[2:3] + cases[6:7]
    print(f"nimg={nimg} rounds={rounds}")
    for label, n, H, c1, c2, cout, use_res in cases:
        M = n * H * H
        g = torch.Generator(device=dev).manual_seed(1)
        x = (torch.randn((M, c1), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        x2 = (torch.randn((M, c2), device=dev, generator=g) * 0.5).to(torch.bfloat16) if c2 else None
        w = (torch.randn((cout, 9 * (c1 + c2)), device=dev, generator=g) * (9 * (c1 + c2)) ** -0.5).to(torch.bfloat16)
        bias = torch.randn(cout, device=dev, generator=g)
        res = torch.randn((M, cout), device=dev, generator=g).to(torch.bfloat16) if use_res else None
        out = torch.empty((M, cout), dtype=torch.bfloat16, device=dev)

        def run(order):
            hip.gemm(x, w, out, M=M, N=cout, K=c1 + c2, ldx=c1, ldw=w.stride(0), ldc=cout, bias=bias, residual=res,
                     ldr=cout if use_res else 0, x2=x2, C1=c1 if c2 else 0, ldx2=c2, mode=1, Hin=H, Win=H, Hout=H, Wout=H, k_order=order)
        if pmc:
            run(order)
            run(order)
            torch.cuda.synchronize()
            print(label, "order", order, "done")
            continue
        A, B = ORDERS
        run(A)
        torch.cuda.synchronize()
        ref = out.float().clone()
        run(B)
        torch.cuda.synchronize()
        dev_rel = float((out.float() - ref).norm() / ref.norm())
        assert dev_rel < 3e-3, (label, dev_rel)        # same products, another summation order: bf16 rounding flips only
        ms = {A: [], B: []}
        for _ in range(rounds):
            for o in (A, B):
                ms[o].append(timed(lambda: run(o)))
        flops = 18.0 * M * (c1 + c2) * cout
        row = [f"order {o}: {flops / statistics.median(ms[o]) / 1e9:6.0f} ({flops / max(ms[o]) / 1e9:5.0f}..{flops / min(ms[o]) / 1e9:5.0f})"
               for o in (A, B)]
        print(f"{label:26s} M={M:8d}  " + "   ".join(row) +
              f"   order {B} / order {A} = {statistics.median(ms[A]) / statistics.median(ms[B]):.3f}   (rel-L2 between the two {dev_rel:.1e})")
        del x, w, out, res, ref


if __name__ == "__main__":
    main()

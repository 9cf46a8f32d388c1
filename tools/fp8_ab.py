#!/usr/bin/env python
"""A/B of the three operand forms of sdv_gemm_bf16 on the shapes BASELINE config 5 runs in fp8 - the ResBlock conv3x3 of a
256-sample UNet forward and the ff.net.2 GEMMs (K = 4C, e4m3 GEGLU output): bf16 (v_mfma_f32_32x32x16_bf16), fp8 form 1
(v_mfma_f32_32x32x16_fp8_fp8, bf16 rate, half the operand bytes) and fp8 form 2 (v_mfma_scale_f32_32x32x64_f8f6f4 with unit
block scales, twice the rate).  Interleaved rounds in one process, median (min..max) TFLOP/s per arm; the two fp8 forms are
checked against each other first (same products, another summation order).
usage: python tools/fp8_ab.py [nimg] [rounds]"""
import os
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


def q8(t):
    s = float(t.float().abs().max()) / 448.0
    return (t.float() / s).to(hip.FP8), s


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda")
    hip.load()
    # (label, conv?, images, H, C1, C2, Cout, residual)
    cases = [("conv 320->320 @64", True, nimg, 64, 320, 0, 320, True), ("conv 960->320 @64", True, nimg, 64, 960, 0, 320, False),
             ("conv 640->640 @32", True, nimg, 32, 640, 0, 640, True), ("conv 1920->640 @32", True, nimg, 32, 1920, 0, 640, False),
             ("conv 1280->1280 @16", True, nimg, 16, 1280, 0, 1280, True), ("conv 2560->1280 @16", True, nimg, 16, 2560, 0, 1280, False),
             ("conv 1280->1280 @8", True, nimg, 8, 1280, 0, 1280, True),
             ("ff2 1280->320 @64", False, nimg, 64, 1280, 0, 320, True), ("ff2 2560->640 @32", False, nimg, 32, 2560, 0, 640, True),
             ("ff2 5120->1280 @16", False, nimg, 16, 5120, 0, 1280, True)]
    print(f"nimg={nimg} rounds={rounds}   TFLOP/s median (min..max)")
    for label, conv, n, H, c1, c2, cout, use_res in cases:
        M = n * H * H
        taps = 9 if conv else 1
        g = torch.Generator(device=dev).manual_seed(1)
        x = (torch.randn((M, c1), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn((cout, taps * c1), device=dev, generator=g) * (taps * c1) ** -0.5).to(torch.bfloat16)
        x8, sx = q8(x)
        w8, sw = q8(w)
        bias = torch.randn(cout, device=dev, generator=g)
        res = torch.randn((M, cout), device=dev, generator=g).to(torch.bfloat16) if use_res else None
        out = torch.empty((M, cout), dtype=torch.bfloat16, device=dev)

        def run(form):
            hip.FP8_MX = 1 if form == 2 else 0
            xx, ww, al = (x, w, 1.0) if form == 0 else (x8, w8, sx * sw)
            hip.gemm(xx, ww, out, M=M, N=cout, K=c1, ldx=c1, ldw=ww.stride(0), ldc=cout, bias=bias, residual=res,
                     ldr=cout if use_res else 0, mode=1 if conv else 0, Hin=H, Win=H, Hout=H, Wout=H, alpha=al,
                     tile=int(os.environ.get("SDV_AB_TILE", "0")))
        run(1)
        torch.cuda.synchronize()
        ref = out.float().clone()
        run(2)
        torch.cuda.synchronize()
        dev_rel = float((out.float() - ref).norm() / ref.norm())
        assert dev_rel < 3e-3, (label, dev_rel)
        ms = {0: [], 1: [], 2: []}
        for _ in range(rounds):
            for f in (0, 1, 2):
                ms[f].append(timed(lambda: run(f)))
        flops = 2.0 * taps * M * c1 * cout
        tf = lambda f: flops / statistics.median(ms[f]) / 1e9
        row = [f"{nm}: {tf(f):6.0f} ({flops / max(ms[f]) / 1e9:5.0f}..{flops / min(ms[f]) / 1e9:5.0f})" for f, nm in ((0, "bf16"), (1, "fp8"), (2, "fp8-MX"))]
        print(f"{label:22s} M={M:8d}  " + "   ".join(row) + f"   MX/fp8 = {tf(2) / tf(1):.3f}  MX/bf16 = {tf(2) / tf(0):.3f}   (rel-L2 MX vs fp8 {dev_rel:.1e})")
        del x, w, x8, w8, out, res, ref
    hip.FP8_MX = 1


if __name__ == "__main__":
    main()

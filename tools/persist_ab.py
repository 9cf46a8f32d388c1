#!/usr/bin/env python
"""A/B the persistent tile walk of the 8-wave igemm tiles on the UNet's layer shapes (HIP events, same box, same process).
usage: persist_ab.py [nimg]      columns: one-workgroup-per-tile TFLOP/s | persistent TFLOP/s | ratio | bit-identical?"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402


def bench(fn, reps=10):
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
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda")
    lib = hip.load()
    shapes = []   # (label, mode, H, Cin, Cout, residual, geglu)
    for H, C in ((64, 320), (32, 640), (16, 1280), (8, 1280)):
        shapes.append((f"conv {C}->{C} @{H} +res", 1, H, C, C, True, False))
        shapes.append((f"gemm {C}->{C} @{H}", 0, H, C, C, False, False))
        shapes.append((f"gemm {C}->{C} @{H} +res", 0, H, C, C, True, False))
        shapes.append((f"gemm {C}->{2*C} @{H}", 0, H, C, 2 * C, False, False))
        shapes.append((f"gemm {C}->{8*C} @{H} geglu", 0, H, C, 8 * C, False, True))
        shapes.append((f"gemm {4*C}->{C} @{H} +res", 0, H, 4 * C, C, True, False))
    shapes.append(("conv 960->320 @64", 1, 64, 960, 320, False, False))
    shapes.append(("conv 2560->1280 @16", 1, 16, 2560, 1280, False, False))
    print(f"nimg={nimg}")
    tot = [0.0, 0.0]
    for label, mode, H, cin, cout, use_res, geglu in shapes:
        M = nimg * H * H
        x = (torch.randn((M, cin), device=dev) * 0.5).to(torch.bfloat16)
        kk = cin if mode == 0 else 9 * cin
        w = (torch.randn((cout, kk), device=dev) * kk ** -0.5).to(torch.bfloat16)
        flops = 2.0 * M * kk * cout
        bias = torch.randn(cout, device=dev)
        nout = cout // 2 if geglu else cout
        res = torch.randn((M, nout), device=dev).to(torch.bfloat16) if use_res else None
        outs = []
        ms = []
        for on in (0, 1):
            lib.sdv_gemm_set_persistent(on)
            out = torch.zeros((M, nout), dtype=torch.bfloat16, device=dev)
            if mode == 0:
                fn = lambda: hip.linear(x, w, bias, residual=res, out=out, epi=1 if geglu else 0)
            else:
                fn = lambda: hip.conv3x3(x, w, bias, nimg=nimg, H=H, W=H, residual=res, out=out)
            ms.append(bench(fn))
            outs.append(out)
        lib.sdv_gemm_set_persistent(1)
        same = bool(torch.equal(outs[0], outs[1]))
        tot[0] += ms[0]
        tot[1] += ms[1]
        print(f"{label:30s} M={M:8d}  {flops/ms[0]/1e9:7.0f}  {flops/ms[1]/1e9:7.0f}  x{ms[0]/ms[1]:.3f}  {'same' if same else 'DIFFERENT'}", flush=True)
    print(f"sum of times: {tot[0]:.2f} ms -> {tot[1]:.2f} ms")


if __name__ == "__main__":
    main()

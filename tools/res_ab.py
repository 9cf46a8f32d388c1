#!/usr/bin/env python
"""Timing of the +residual epilogue on the 256 x 320 tile: the ResBlock conv2 shapes and the transformer's out / ff.net.2 GEMMs of
a 256-sample forward, TFLOP/s median (min..max).  Meant to be run once per build of the library (SDV_HIP_LIB=<an experimental
build made with tools/ubench/build_gemm_timing.py notiming -D...>) on ONE box, back to back - how the growing residual look-ahead
experiment of round 3 was measured (profiles/round3_epilogue_whatif.txt; the experiment's source knob is gone again).
usage: [SDV_HIP_LIB=...] python tools/res_ab.py [nimg] [rounds]"""
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
    cases = [("conv 320->320 @64 +res", True, 64, 320, 320), ("conv 640->640 @32 +res", True, 32, 640, 640),
             ("conv 1280->1280 @16 +res", True, 16, 1280, 1280), ("conv 1280->1280 @8 +res", True, 8, 1280, 1280),
             ("out 320->320 @64 +res", False, 64, 320, 320), ("ff2 1280->320 @64 +res", False, 64, 1280, 320),
             ("out 640->640 @32 +res", False, 32, 640, 640), ("ff2 2560->640 @32 +res", False, 32, 2560, 640),
             ("out 1280->1280 @16 +res", False, 16, 1280, 1280), ("ff2 5120->1280 @16 +res", False, 16, 5120, 1280),
             ("proj 320->320 @64 bias", False, 64, 320, -320)]
    print(f"lib={hip._LIB_PATH} nimg={nimg} rounds={rounds}")
    tot = 0.0
    for label, conv, H, cin, cout in cases:
        use_res = cout > 0
        cout = abs(cout)
        M = nimg * H * H
        taps = 9 if conv else 1
        g = torch.Generator(device=dev).manual_seed(1)
        x = (torch.randn((M, cin), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn((cout, taps * cin), device=dev, generator=g) * (taps * cin) ** -0.5).to(torch.bfloat16)
        bias = torch.randn(cout, device=dev, generator=g)
        res = torch.randn((M, cout), device=dev, generator=g).to(torch.bfloat16) if use_res else None
        out = torch.empty((M, cout), dtype=torch.bfloat16, device=dev)

        def run():
            hip.gemm(x, w, out, M=M, N=cout, K=cin, ldx=cin, ldw=w.stride(0), ldc=cout, bias=bias, residual=res,
                     ldr=cout if use_res else 0, mode=1 if conv else 0, Hin=H, Win=H, Hout=H, Wout=H, tile=6)
        run()
        torch.cuda.synchronize()
        ref = (torch.nn.functional.conv2d(x.float().view(nimg, H, H, cin).permute(0, 3, 1, 2)[:2], w.float().view(cout, 3, 3, cin).permute(0, 3, 1, 2), bias, padding=1)
               .permute(0, 2, 3, 1).reshape(-1, cout) if conv else x[:2 * H * H].float() @ w.float().T + bias)
        if use_res:
            ref = ref + res[:2 * H * H].float()
        rel = float((out[:2 * H * H].float() - ref).norm() / ref.norm())
        assert rel < 4e-3, (label, rel)
        ms = [timed(run) for _ in range(rounds)]
        flops = 2.0 * taps * M * cin * cout
        tot += statistics.median(ms)
        print(f"{label:26s} M={M:8d}  {flops / statistics.median(ms) / 1e9:6.0f} ({flops / max(ms) / 1e9:5.0f}..{flops / min(ms) / 1e9:5.0f}) TFLOP/s   {statistics.median(ms):7.3f} ms   rel-L2 {rel:.1e}")
        del x, w, out, res
    print(f"sum of medians {tot:.3f} ms")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""ff.net.0 (GEGLU-in) at the three transformer levels of a 256-sample forward on the 256 x 320 tile: TFLOP/s median of `rounds`.
For A/B of library builds (SDV_HIP_LIB=tools/ubench/libsdv_<name>.so): run the arms alternately, A B A B, on one box.
usage: python tools/ff1_bench.py [nimg] [rounds]"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402
from stable_diffusion_videos_amd.weights import geglu_interleave  # noqa: E402


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    dev = torch.device("cuda")
    hip.load()
    row = []
    for H, C in ((64, 320), (32, 640), (16, 1280)):
        M, K, N = nimg * H * H, C, 8 * C
        x = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16)
        w = geglu_interleave(torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = geglu_interleave(torch.randn(N, device=dev))
        out = torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)
        ms = []
        for r in range(rounds + 1):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                hip.linear(x, w, bias, out=out, epi=1, tile=6)
            e.record()
            torch.cuda.synchronize()
            if r:
                ms.append(s.elapsed_time(e) / 3)
        row.append(f"ff1 {C}->{8*C} @{H}: {2.0 * M * N * K / statistics.median(ms) / 1e9:6.0f} TFLOP/s")
        del x, w, out
    import os
    print(f"[{os.environ.get('SDV_HIP_LIB', 'libsdv_hip.so')}] " + "   ".join(row))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Where does a short-K GEMM spend its time?  Sweep K at fixed M, N for the epilogue variants."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402


def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    dev = torch.device("cuda")
    M = 262144
    for N, tile, epi in ((320, 6, 0), (2560, 7, 1), (2560, 7, 0)):
        nout = N // 2 if epi == 1 else N
        out = torch.empty((M, nout), dtype=torch.bfloat16, device=dev)
        res = torch.randn((M, nout), device=dev).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        for K in (64, 128, 320, 640, 1280):
            x = torch.randn((M, K), device=dev).to(torch.bfloat16)
            w = torch.randn((N, K), device=dev).to(torch.bfloat16)
            t_plain = bench(lambda: hip.linear(x, w, None, out=out, epi=epi, tile=tile))
            t_bias = bench(lambda: hip.linear(x, w, bias, out=out, epi=epi, tile=tile))
            t_res = bench(lambda: hip.linear(x, w, bias, residual=res, out=out, epi=epi, tile=tile)) if epi != 1 else float("nan")
            bm = 256 if tile != 9 else 128
            bn = {6: 320, 7: 256, 9: 320, 12: 320, 13: 256}[tile]
            rounds = ((M // bm) * (N // bn) + 255) // 256
            print(f"N={N} tile={tile} epi={epi} K={K:5d}: plain {t_plain:8.1f} us  +bias {t_bias:8.1f}  +bias+res {t_res:8.1f}"
                  f" | per round {t_plain / rounds:6.2f} us  TF(plain) {2.0 * M * N * K / t_plain / 1e6:7.1f}")


if __name__ == "__main__":
    main()

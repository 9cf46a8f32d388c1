#!/usr/bin/env python
"""The residual-free C = 640 projections (proj_in, fused QKV, attn2.to_q of the 32 x 32 level) on the panel kernel's 10-slab form
(sdv_linear640_bf16) against the igemm launches they replace: agreement + timings at M = samples * 1024.
usage: linear640_ab.py [samples=256] [rounds=3]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402
from stable_diffusion_videos_amd.weights import ffn_fold_columns, ln_fold  # noqa: E402

BF16 = torch.bfloat16
K = 640


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    qs = hip.q_prescale(80)
    gamma, beta = 1.0 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    w = lambda: torch.randn(K, K, generator=g) * K ** -0.5
    bias = (torch.randn(K, generator=g) * 0.1).to(dev)
    wb = w().to(dev, BF16)
    wxb = ffn_fold_columns(torch.zeros(K, device=dev), bias)
    parts = [ln_fold(w(), gamma, beta, None, dev) for _ in range(3)]
    W3, s3, t3 = (torch.cat([p_[j] for p_ in parts]).contiguous() for j in range(3))
    wx3 = ffn_fold_columns(s3, t3)
    al3 = torch.tensor([qs, qs, 1.0, 1.0, 1.0, 1.0], device=dev)
    t3s = torch.cat([t3[:K] * qs, t3[K:]])
    wq, sq, tq = parts[0]
    wxq, alq = ffn_fold_columns(sq, tq), torch.tensor([qs, qs], device=dev)
    M = nimg * 1024
    x = (torch.randn(M, K, device=dev) * 1.5 + 0.7).to(BF16)
    xf = x.float()
    st = torch.stack([xf.mean(1), torch.rsqrt(xf.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    cases = {
        "bias+stats": (lambda: hip.linear320(x, wb, wxb, want_stats=True)[0], lambda: hip.linear(x, wb, bias, want_stats=True)[0], K),
        "qkv fold": (lambda: hip.linear320(x, W3, wx3, ln_stats=st, alpha=al3), lambda: hip.linear(x, W3, t3s, alpha=qs, alpha_cols=K, ln=(st, s3)), 3 * K),
        "q2 fold": (lambda: hip.linear320(x, wq, wxq, ln_stats=st, alpha=alq), lambda: hip.linear(x, wq, tq * qs, alpha=qs, ln=(st, sq)), K),
    }
    for name, (new, old, N) in cases.items():
        a, b, a2 = new(), old(), new()
        torch.cuda.synchronize()
        ulp = torch.exp2(torch.floor(torch.log2(torch.maximum(a.float().abs(), b.float().abs()).clamp_min(2.0 ** -6))) - 7)
        print(f"M={M} {name:10s}: differ {int((a != b).sum())}/{a.numel()}  max {float(((a.float() - b.float()).abs() / ulp).max()):.1f} ulp   "
              f"repeat identical {bool(torch.equal(a, a2))}   finite {bool(torch.isfinite(a.float()).all())}")

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        return sorted(ts)[len(ts) // 2]

    for _ in range(rounds):
        for name, (new, old, N) in cases.items():
            tn, to = timed(new), timed(old)
            gb, tf = 2.0 * M * (K + N) / 1e9, 2.0 * M * N * K / 1e9
            print(f"M={M} {name:10s}: panel {tn:.3f} ms ({gb / tn:.2f} TB/s, {tf / tn:.0f} TFLOP/s)   igemm {to:.3f} ms ({tf / to:.0f} TFLOP/s)   x{to / tn:.2f}")


if __name__ == "__main__":
    main()

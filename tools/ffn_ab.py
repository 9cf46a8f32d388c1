#!/usr/bin/env python
"""Fused GEGLU feed-forward (sdv_ffn_geglu_bf16, csrc/sdv_ffn.hip) against the two-launch form it replaces (ff.net.0 with the GEGLU
epilogue + LayerNorm fold, ff.net.2 with the residual) and against a float64 reference.   usage: ffn_ab.py [samples=256] [rounds=5]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402
from stable_diffusion_videos_amd.weights import ffn_fold_columns, ffn_w2_permute, geglu_interleave, ln_fold  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def make(C, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn(8 * C, C, generator=g) * C ** -0.5
    b1 = torch.randn(8 * C, generator=g) * 0.1
    w2 = torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5
    b2 = torch.randn(C, generator=g) * 0.1
    gamma = 1.0 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    w1p, s1, t1 = ln_fold(geglu_interleave(w1), gamma, beta, geglu_interleave(b1), dev)
    return dict(w1=w1, b1=b1, w2=w2, b2=b2, gamma=gamma, beta=beta, w1p=w1p, s1=s1, t1=t1, w1x=ffn_fold_columns(s1, t1),
                w2d=w2.to(dev, BF16).contiguous(), w2p=ffn_w2_permute(w2.to(BF16)).to(dev), b2d=b2.to(dev, F32))


def reference(x_bf16, P, eps=1e-5):
    """float64 on the bf16-rounded operands the kernels see (x, gamma o W1, W2), hidden activations NOT rounded"""
    x = x_bf16.double().cpu()
    mu, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
    n = (x - mu) / torch.sqrt(var + eps)
    C = x.shape[1]
    w1 = (P["w1"].double() * P["gamma"].double()[None]).to(BF16).double()     # what ln_fold rounds
    h = n @ w1.T / 1.0
    # LN(x) W^T + b with gamma folded: n_hat gamma W^T + (W beta + b)
    h = h + (P["w1"].double() @ P["beta"].double() + P["b1"].double())[None]
    v, gt = h[:, :4 * C], h[:, 4 * C:]
    hid = v * torch.nn.functional.gelu(gt)
    return x + hid @ P["w2"].to(BF16).double().T + P["b2"].double()[None]


def stats_of(x, eps=1e-5):
    xf = x.float()
    mu = xf.mean(1)
    rstd = torch.rsqrt(xf.var(1, unbiased=False) + eps)
    return torch.stack([mu, rstd], 1).contiguous()


def two_launch(x, st, P):
    g = hip.linear(x, P["w1p"], P["t1"], epi=1, ln=(st, P["s1"]))
    return hip.linear(g, P["w2d"], P["b2d"], residual=x)


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    C = 320
    P = make(C, dev)
    # ---- correctness: ragged M, rows with a common mode ----
    for M in (128, 1000, 128 * 300 + 17):
        g = torch.Generator().manual_seed(M)
        x = (torch.randn(M, C, generator=g) * 1.5 + torch.randn(M, 1, generator=g) * 3.0).to(dev, BF16)
        st = stats_of(x)
        ref = reference(x, P)
        a = hip.ffn_geglu(x, st, P["w1p"], P["w1x"], P["w2p"], P["b2d"])
        b = two_launch(x, st, P)
        torch.cuda.synchronize()
        ea = (a.double().cpu() - ref).abs()
        eb = (b.double().cpu() - ref).abs()
        a2 = hip.ffn_geglu(x, st, P["w1p"], P["w1x"], P["w2p"], P["b2d"])
        torch.cuda.synchronize()
        print(f"M={M:6d}  fused: max|d| {ea.max():.4f} rel-L2 {float((ea ** 2).sum().sqrt() / (ref ** 2).sum().sqrt()):.2e}   two-launch: max|d| {eb.max():.4f} "
              f"rel-L2 {float((eb ** 2).sum().sqrt() / (ref ** 2).sum().sqrt()):.2e}   fused vs two-launch: {int((a != b).sum())} of {a.numel()} differ, "
              f"max {float((a.float() - b.float()).abs().max()):.4f}   repeat identical: {bool(torch.equal(a, a2))}   finite: {bool(torch.isfinite(a.float()).all())}")
    # ---- timing ----
    M = nimg * 4096
    x = (torch.randn(M, C, device=dev) * 1.5).to(BF16)
    st = stats_of(x)
    out = torch.empty_like(x)
    flops = 2.0 * M * 12 * C * C

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

    for r in range(rounds):
        tf = timed(lambda: hip.ffn_geglu(x, st, P["w1p"], P["w1x"], P["w2p"], P["b2d"], out=out))
        tt = timed(lambda: two_launch(x, st, P))
        print(f"M={M}: fused {tf:.3f} ms ({flops / tf / 1e9:.0f} TFLOP/s)   two launches {tt:.3f} ms ({flops / tt / 1e9:.0f} TFLOP/s)   x{tt / tf:.2f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""The C = 320 projections on the panel kernel (sdv_linear320_bf16) against the igemm launches they replace and a float64 reference:
proj_in (bias, row statistics), the fused QKV projection (LayerNorm fold, alpha on the Q third), attn.to_out (residual, statistics),
attn2.to_q (fold, alpha).   usage: linear320_ab.py [samples=256] [rounds=3]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402
from stable_diffusion_videos_amd.weights import ffn_fold_columns, ln_fold  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
C = 320


def stats_of(x, eps=1e-5):
    xf = x.float()
    return torch.stack([xf.mean(1), torch.rsqrt(xf.var(1, unbiased=False) + eps)], 1).contiguous()


def forms(dev):
    g = torch.Generator().manual_seed(0)
    qs = hip.q_prescale(40)
    gamma, beta = 1.0 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w = lambda n: torch.randn(n, C, generator=g) * C ** -0.5
    b = lambda n: torch.randn(n, generator=g) * 0.1
    out = {}
    # proj_in / to_out: plain bias
    wp, bp = w(C), b(C)
    out["bias+stats"] = dict(w=wp.to(dev, BF16), bias=bp.to(dev), wx=ffn_fold_columns(torch.zeros(C), bp).to(dev), N=C)
    # fused QKV: fold, alpha on the first block
    parts = [ln_fold(w(C), gamma, beta, None, dev, scale=qs if i == 0 else 1.0) for i in range(3)]
    parts1 = [(p_[0], p_[1], p_[2] / (qs if i == 0 else 1.0)) for i, p_ in enumerate(parts)]
    W3, s3, t3 = (torch.cat([p_[j] for p_ in parts]).contiguous() for j in range(3))
    t3u = torch.cat([p_[2] for p_ in parts1]).contiguous()
    out["qkv fold"] = dict(w=W3, s=s3, t=t3, wx=ffn_fold_columns(s3, t3u), alpha=torch.tensor([qs, 1.0, 1.0], device=dev), N=3 * C, qs=qs)
    wq, sq, tq = ln_fold(w(C), gamma, beta, None, dev, scale=qs)
    out["q2 fold"] = dict(w=wq, s=sq, t=tq, wx=ffn_fold_columns(sq, tq / qs), alpha=torch.tensor([qs], device=dev), N=C, qs=qs)
    return out


def run_old(x, st, res, F, name):
    if name == "bias+stats":
        return hip.linear(x, F["w"], F["bias"], residual=res, want_stats=True)
    if name == "qkv fold":
        return hip.linear(x, F["w"], F["t"], alpha=F["qs"], alpha_cols=C, ln=(st, F["s"])), None
    return hip.linear(x, F["w"], F["t"], alpha=F["qs"], ln=(st, F["s"])), None


def run_new(x, st, res, F, name, vt=None, hw=0):
    if name == "bias+stats":
        return hip.linear320(x, F["w"], F["wx"], residual=res, want_stats=True)
    return hip.linear320(x, F["w"], F["wx"], ln_stats=st, alpha=F["alpha"], vt=vt, hw=hw), None


def reference(x, st, res, F, name):
    x64 = x.double()
    if name == "bias+stats":
        y = x64 @ F["w"].double().T + F["bias"].double()[None]
        return y + (res.double() if res is not None else 0.0)
    al = torch.ones(F["N"], dtype=torch.float64, device=x.device)
    al[:C] = F["qs"]
    t = F["t"].double()
    return (x64 @ F["w"].double().T - st[:, :1].double() * F["s"].double()[None]) * (st[:, 1:].double() * al[None]) + t[None]


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    FS = forms(dev)
    cases = [("bias+stats", False), ("bias+stats", True), ("qkv fold", False), ("q2 fold", False)]
    for M in (128, 1000, 128 * 300 + 17):
        g = torch.Generator().manual_seed(M)
        x = (torch.randn(M, C, generator=g) * 1.5 + torch.randn(M, 1, generator=g) * 3.0).to(dev, BF16)
        r = (torch.randn(M, C, generator=g) * 2.0).to(dev, BF16)
        st = stats_of(x)
        for name, use_r in cases:
            F = FS[name]
            ref = reference(x, st, r if use_r else None, F, name)
            a, sa = run_new(x, st, r if use_r else None, F, name)
            b, sb = run_old(x, st, r if use_r else None, F, name)
            a2, _ = run_new(x, st, r if use_r else None, F, name)
            torch.cuda.synchronize()
            ea, eb = (a.double() - ref).abs(), (b.double() - ref).abs()
            msg = (f"M={M:6d} {name:10s} res={int(use_r)}  new: max|d| {float(ea.max()):.4f} rel-L2 {float(ea.norm() / ref.norm()):.2e}   igemm: max|d| {float(eb.max()):.4f} "
                   f"rel-L2 {float(eb.norm() / ref.norm()):.2e}   differ {int((a != b).sum())}/{a.numel()}   repeat identical {bool(torch.equal(a, a2))}")
            if sa is not None:
                true = stats_of(a)
                msg += f"   stats: max|mean err| {float((sa[:, 0] - true[:, 0]).abs().max()):.2e} max rel rstd err {float(((sa[:, 1] - true[:, 1]) / true[:, 1]).abs().max()):.2e}"
            print(msg)
    # transposed V output: [Q | K] + V^T must be the row-major result, re-arranged
    for M, hw in ((1024, 256), (4096 * 3, 4096), (128 * 20, 128)):
        g = torch.Generator().manual_seed(M)
        x = (torch.randn(M, C, generator=g) * 1.5 + torch.randn(M, 1, generator=g) * 3.0).to(dev, BF16)
        st = stats_of(x)
        F = FS["qkv fold"]
        full, _ = run_new(x, st, None, F, "qkv fold")
        vt = torch.zeros((M // hw, C, hw + 64), dtype=BF16, device=dev)[:, :, :hw]
        qk, _ = run_new(x, st, None, F, "qkv fold", vt=vt, hw=hw)
        torch.cuda.synchronize()
        ok_qk = bool(torch.equal(qk, full[:, :2 * C]))
        ok_vt = bool(torch.equal(vt, full[:, 2 * C:].reshape(M // hw, hw, C).transpose(1, 2)))
        print(f"M={M} hw={hw}: [Q | K] identical {ok_qk}, V^T identical {ok_vt}")
    M = nimg * 4096
    x = (torch.randn(M, C, device=dev) * 1.5).to(BF16)
    r = (torch.randn(M, C, device=dev) * 2.0).to(BF16)
    st = stats_of(x)

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
        for name, use_r in cases:
            F = FS[name]
            tn = timed(lambda: run_new(x, st, r if use_r else None, F, name))
            to = timed(lambda: run_old(x, st, r if use_r else None, F, name))
            gb = 2.0 * M * (C + F["N"] + (C if use_r else 0)) / 1e9
            print(f"M={M} {name:10s} res={int(use_r)}: panel {tn:.3f} ms ({gb / tn:.2f} TB/s)   igemm {to:.3f} ms ({gb / to:.2f} TB/s)   x{to / tn:.2f}")
        vtb = torch.empty((nimg, C, 4096), dtype=BF16, device=dev)
        tv = timed(lambda: run_new(x, st, None, FS["qkv fold"], "qkv fold", vt=vtb, hw=4096))
        print(f"M={M} qkv fold with transposed V: panel {tv:.3f} ms")


if __name__ == "__main__":
    main()

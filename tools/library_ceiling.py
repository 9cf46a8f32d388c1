#!/usr/bin/env python
"""Calibration, not product: what AMD's own libraries reach ON THIS BOX on the shapes of a 256-sample UNet forward, next to
the shipped kernels - torch.matmul / F.linear (hipBLASLt, assembly kernels), F.conv2d channels_last (MIOpen),
F.scaled_dot_product_attention (the flash kernel torch ships for ROCm).  The product never calls any of these (there is no
library GEMM, conv or attention anywhere under stable_diffusion_videos_amd/); the numbers bound what "asm-level" would buy on
each shape and the kernel names say which macro-tile / staging form wins.  The library ops do LESS work than the shipped
launches they sit beside (no GEGLU, no LayerNorm fold, no residual unless noted), so a library number is an upper bound.

usage: python tools/library_ceiling.py [nimg] [rounds] [what=gemm,conv,attn]"""
import statistics
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402
from stable_diffusion_videos_amd.weights import geglu_interleave  # noqa: E402


def timed(fn, reps=3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def kernel_names(fn):
    """names of the device kernels one call launches (torch.profiler; best effort)"""
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            fn()
            torch.cuda.synchronize()
        names = []
        for ev in prof.events():
            if ev.device_type == torch.autograd.DeviceType.CUDA and ev.name not in names:
                names.append(ev.name)
        return names
    except Exception as ex:  # noqa: BLE001
        return [f"(profiler unavailable: {ex})"]


def ab(label, flops, fns, rounds, note=""):
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    ms = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            ms[k].append(timed(f))
    row = "   ".join(f"{k}: {flops / statistics.median(v) / 1e9:6.0f} ({flops / max(v) / 1e9:5.0f}..{flops / min(v) / 1e9:5.0f})" for k, v in ms.items())
    keys = list(fns)
    ratio = statistics.median(ms[keys[0]]) / statistics.median(ms[keys[-1]])
    print(f"{label:30s} {row}   {keys[-1]}/{keys[0]} = {ratio:.3f} {note}", flush=True)
    return ms


def gemms(nimg, rounds, dev):
    print("== dense GEMMs: TFLOP/s median (min..max); sdv = shipped launch (bias / +res / GEGLU as named), lib = F.linear + bias "
          "(hipBLASLt; '+res' adds the residual with addmm-style beta where torch fuses it, GEGLU rows are plain bias)")
    for H, C in ((64, 320), (32, 640), (16, 1280)):
        M = nimg * H * H
        for label, K, N, kind in ((f"proj  {C}->{C} @{H} bias", C, C, "bias"), (f"out   {C}->{C} @{H} +res", C, C, "res"),
                                  (f"qk    {C}->{2*C} @{H} bias", C, 2 * C, "bias"), (f"ff1   {C}->{8*C} @{H} geglu", C, 8 * C, "geglu"),
                                  (f"ff2   {4*C}->{C} @{H} +res", 4 * C, C, "res")):
            x = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16)
            w = torch.randn((N, K), device=dev) * K ** -0.5
            bias = torch.randn(N, device=dev)
            nout = N // 2 if kind == "geglu" else N
            res = torch.randn((M, nout), device=dev).to(torch.bfloat16) if kind == "res" else None
            out = torch.empty((M, nout), dtype=torch.bfloat16, device=dev)
            wl, bl = w.to(torch.bfloat16), bias.to(torch.bfloat16)
            ws, bs = (geglu_interleave(w), geglu_interleave(bias)) if kind == "geglu" else (w, bias)
            ws = ws.to(torch.bfloat16)
            lib_out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

            def lib():
                if res is not None:
                    torch.addmm(res, x, wl.t(), out=lib_out)      # beta * res + x @ W^T (no bias: one epilogue operand, as the asm kernels take)
                else:
                    torch.addmm(bl, x, wl.t(), out=lib_out)

            fns = {"sdv": lambda: hip.linear(x, ws, bs, residual=res, out=out, epi=1 if kind == "geglu" else 0), "lib": lib}
            ab(label + f" M={M}", 2.0 * M * N * K, fns, rounds)
            if H == 64:
                print("      lib kernels:", "; ".join(n[:150] for n in kernel_names(lib)), flush=True)
            del x, w, out, res, lib_out


def convs(nimg, rounds, dev):
    print("== conv3x3 stride 1 pad 1, NHWC bf16: sdv = shipped implicit GEMM (+bias), lib = F.conv2d on channels_last tensors (MIOpen)")
    for H, cin, cout in ((64, 320, 320), (32, 640, 640), (16, 1280, 1280), (8, 1280, 1280)):
        n = nimg
        x = (torch.randn((n, H, H, cin), device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn((cout, 3, 3, cin), device=dev) * (9 * cin) ** -0.5).to(torch.bfloat16)
        bias = torch.randn(cout, device=dev)
        x_cl = x.permute(0, 3, 1, 2)            # NCHW view of the NHWC storage = channels_last
        w_cl = w.permute(0, 3, 1, 2)
        bl = bias.to(torch.bfloat16)
        x2d, w2d = x.reshape(n * H * H, cin), w.reshape(cout, 9 * cin)
        out = torch.empty((n * H * H, cout), dtype=torch.bfloat16, device=dev)
        fns = {"sdv": lambda: hip.conv3x3(x2d, w2d, bias, nimg=n, H=H, W=H, out=out), "lib": lambda: F.conv2d(x_cl, w_cl, bl, padding=1)}
        try:
            ab(f"conv {cin}->{cout} @{H} n={n}", 2.0 * n * H * H * cout * 9 * cin, fns, rounds)
            if H == 64:
                print("      lib kernels:", "; ".join(nm[:150] for nm in kernel_names(fns["lib"])), flush=True)
        except Exception as ex:  # noqa: BLE001
            print(f"conv {cin}->{cout} @{H}: {type(ex).__name__}: {ex}", flush=True)
        del x, w, out


def attns(nimg, rounds, dev):
    print("== self-attention: sdv = shipped flash kernel (Q pre-scaled, V^T supplied), lib = F.scaled_dot_product_attention")
    for H, C, heads in ((64, 320, 8), (32, 640, 8), (16, 1280, 8)):
        L, dh = H * H, C // heads
        q = torch.randn((nimg, L, C), device=dev).to(torch.bfloat16)
        k = torch.randn((nimg, L, C), device=dev).to(torch.bfloat16)
        v = torch.randn((nimg, L, C), device=dev).to(torch.bfloat16)
        vt = v.permute(0, 2, 1).contiguous()                                   # [B, C = H * dh, L]
        out = torch.empty((nimg * L, C), dtype=torch.bfloat16, device=dev)
        q2, k2 = q.reshape(nimg * L, C), k.reshape(nimg * L, C)
        q4, k4, v4 = (t.reshape(nimg, L, heads, dh).permute(0, 2, 1, 3) for t in (q, k, v))
        fns = {"sdv": lambda: hip.attention(q2, k2, vt, out, B=nimg, H=heads, Lq=L, Lk=L, dh=dh, ldq=C, ldk=C, ldv=L, ldo=C, scale=dh ** -0.5),
               "lib": lambda: F.scaled_dot_product_attention(q4, k4, v4)}
        try:
            ab(f"attn L={L} dh={dh} B={nimg}", 4.0 * nimg * heads * L * L * dh, fns, rounds)
            print("      lib kernels:", "; ".join(nm[:150] for nm in kernel_names(fns["lib"])), flush=True)
        except Exception as ex:  # noqa: BLE001
            print(f"attn L={L} dh={dh}: {type(ex).__name__}: {ex}", flush=True)
        del q, k, v, vt, out


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    what = (sys.argv[3] if len(sys.argv) > 3 else "gemm,conv,attn").split(",")
    dev = torch.device("cuda")
    hip.load()
    print(f"nimg={nimg} rounds={rounds}  torch {torch.__version__}  {torch.cuda.get_device_name(0)}")
    if "gemm" in what:
        gemms(nimg, rounds, dev)
    if "attn" in what:
        attns(nimg, rounds, dev)
    if "conv" in what:
        convs(nimg, rounds, dev)


if __name__ == "__main__":
    main()

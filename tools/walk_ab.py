#!/usr/bin/env python
"""A/B of the persistent workgroups' TILE ORDER (sdv_gemm_set_walk): 0 = strided raster (workgroup b takes tiles b, b + 256, ...:
an X panel is shared only between the CUs that miss on it at the same moment), S = 1 / 2 / 4 = panel walk (a workgroup walks
tiles_n / S N tiles of ONE M panel back to back: from its second tile on the panel comes out of L2 / Infinity Cache).
Interleaved rounds in ONE process, median (min..max) per arm, results asserted bit-identical.

NEEDS a library built with the walk compiled in: python tools/ubench/build_variant.py walk -DSDV_PANEL_WALK=1, then
usage: SDV_HIP_LIB=tools/ubench/libsdv_walk.so python tools/walk_ab.py [nimg] [rounds] [unet]      ("unet": also whole eager UNet forwards per setting, HIP-event totals)
"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402
from stable_diffusion_videos_amd.weights import geglu_interleave  # noqa: E402

ARMS = (0, 1, 2, 4)


def timed(fn, reps=3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def shapes(nimg, dev):
    for H, C in ((64, 320), (32, 640), (16, 1280)):
        M = nimg * H * H
        for label, K, N, kind in ((f"qk    {C}->{2*C} @{H} bias", C, 2 * C, "bias"), (f"ff1   {C}->{8*C} @{H} geglu", C, 8 * C, "geglu"),
                                  (f"ff2   {4*C}->{C} @{H} +res", 4 * C, C, "res"), (f"out   {C}->{C} @{H} +res", C, C, "res")):
            x = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16)
            w = torch.randn((N, K), device=dev) * K ** -0.5
            bias = torch.randn(N, device=dev)
            nout = N // 2 if kind == "geglu" else N
            res = torch.randn((M, nout), device=dev).to(torch.bfloat16) if kind == "res" else None
            out = torch.empty((M, nout), dtype=torch.bfloat16, device=dev)
            if kind == "geglu":
                w, bias = geglu_interleave(w), geglu_interleave(bias)
            w = w.to(torch.bfloat16)
            fn = (lambda x=x, w=w, bias=bias, res=res, out=out, kind=kind:
                  hip.linear(x, w, bias, residual=res, out=out, epi=1 if kind == "geglu" else 0, tile=6))
            yield label, M, 2.0 * M * N * K, fn, out
            del x, w, out, res
    for label, H, C1, C2, Co in (("conv 640->640 @32", 32, 640, 0, 640), ("conv 1280->1280 @16", 16, 1280, 0, 1280),
                                 ("conv 1280->1280 @8", 8, 1280, 0, 1280), ("conv 1280+1280->1280 @16", 16, 1280, 1280, 1280),
                                 ("conv 640+640->640 @32", 32, 640, 640, 640), ("conv 320->320 @64", 64, 320, 0, 320)):
        M = nimg * H * H
        x = (torch.randn((M, C1), device=dev) * 0.5).to(torch.bfloat16)
        x2 = (torch.randn((M, C2), device=dev) * 0.5).to(torch.bfloat16) if C2 else None
        w = (torch.randn((Co, 9 * (C1 + C2)), device=dev) * (9 * (C1 + C2)) ** -0.5).to(torch.bfloat16)
        bias = torch.randn(Co, device=dev)
        res = torch.randn((M, Co), device=dev).to(torch.bfloat16)
        out = torch.empty((M, Co), dtype=torch.bfloat16, device=dev)
        fn = (lambda x=x, x2=x2, w=w, bias=bias, res=res, out=out, H=H:
              hip.conv3x3(x, w, bias, nimg=nimg, H=H, W=H, x2=x2, residual=res, out=out, tile=6))
        yield label, M, 18.0 * M * Co * (C1 + C2), fn, out
        del x, x2, w, out, res


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda")
    lib = hip.load()
    print(f"nimg={nimg} rounds={rounds}   TFLOP/s median (min..max) per walk setting; last column = best / strided")
    for label, M, flops, fn, out in shapes(nimg, dev):
        ref = None
        for a in ARMS:
            lib.sdv_gemm_set_walk(a)
            fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                assert torch.equal(out, ref), f"{label}: walk {a} differs from the strided order"
        ms = {a: [] for a in ARMS}
        for _ in range(rounds):
            for a in ARMS:
                lib.sdv_gemm_set_walk(a)
                ms[a].append(timed(fn))
        med = {a: statistics.median(ms[a]) for a in ARMS}
        row = "   ".join(f"walk {a}: {flops / med[a] / 1e9:6.0f} ({flops / max(ms[a]) / 1e9:5.0f}..{flops / min(ms[a]) / 1e9:5.0f})" for a in ARMS)
        best = min(ARMS, key=lambda a: med[a])
        print(f"{label:28s} M={M:8d}  {row}   best {best}: x{med[0] / med[best]:.3f}")
    lib.sdv_gemm_set_walk(0)
    if "unet" in sys.argv[3:]:
        from bench import EventProfiler
        from stable_diffusion_videos_amd import StableDiffusionWalkPipeline
        B = nimg // 2
        pipe = StableDiffusionWalkPipeline.from_pretrained("CompVis/stable-diffusion-v1-4", arch="sd14").to("cuda")
        emb = pipe.embed_text(["a cat"] * B)
        ctx = torch.cat([pipe._uncond_embeddings(None, B), emb.float()])
        pipe._schedule(50, 0.0)
        pipe.unet.prepare_context(ctx)
        pipe.unet.reserve(2 * B, 64, 64)
        x2 = torch.randn((2 * B * 64 * 64, 4), device="cuda").to(torch.bfloat16)
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        res = {a: [] for a in ARMS}
        for r in range(rounds + 1):
            for a in ARMS:
                lib.sdv_gemm_set_walk(a)
                prof = EventProfiler()
                hip.LAUNCH_HOOK = prof
                pipe.unet.forward(x2, 2 * B, 64, 64, step, cfg_shared=True)
                torch.cuda.synchronize()
                hip.LAUNCH_HOOK = None
                if r:
                    res[a].append({k: v["ms"] for k, v in prof.summary().items()})
        kinds = sorted({k for v in res.values() for d in v for k in d})
        print(f"UNet forward, {2 * B} samples, ms per forward (min over {rounds} rounds)")
        print(f"{'walk':6s} " + " ".join(f"{k:>12s}" for k in kinds) + f" {'total':>10s}")
        for a in ARMS:
            mins = {k: min(d.get(k, 0.0) for d in res[a]) for k in kinds}
            print(f"{a:<6d} " + " ".join(f"{mins[k]:12.3f}" for k in kinds) + f" {min(sum(d.values()) for d in res[a]):10.3f}")
        lib.sdv_gemm_set_walk(0)


if __name__ == "__main__":
    main()

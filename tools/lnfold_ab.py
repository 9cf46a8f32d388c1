#!/usr/bin/env python
"""A/B of the LayerNorm fold: two engines built in ONE process (SDV_LN_FOLD=1 / 0), interleaved forwards, HIP events per launch.
usage: lnfold_ab.py [B] [rounds]"""
import os
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import EventProfiler  # noqa: E402
from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pipes = {}
for name, v in (("fold", "1"), ("plain", "0")):
    os.environ["SDV_LN_FOLD"] = v
    pipes[name] = StableDiffusionWalkPipeline.from_pretrained("CompVis/stable-diffusion-v1-4", arch="sd14").to("cuda")
x2 = torch.randn((2 * B * 64 * 64, 4), device="cuda").to(torch.bfloat16)
step = torch.zeros(1, dtype=torch.int32, device="cuda")
res = {k: [] for k in pipes}
shapes = {}
for name, pipe in pipes.items():
    emb = pipe.embed_text(["a cat"] * B)
    ctx = torch.cat([pipe._uncond_embeddings(None, B), emb.float()])
    pipe._schedule(50, 0.0)
    pipe.unet.prepare_context(ctx)
    pipe.unet.reserve(2 * B, 64, 64)
for r in range(rounds + 1):
    for name, pipe in pipes.items():
        prof = EventProfiler()
        hip.LAUNCH_HOOK = prof
        eps = pipe.unet.forward(x2, 2 * B, 64, 64, step, cfg_shared=True)
        torch.cuda.synchronize()
        hip.LAUNCH_HOOK = None
        if r:
            res[name].append({k: v["ms"] for k, v in prof.summary().items()})
            shapes[name] = prof.by_shape()
kinds = sorted({k for v in res.values() for d in v for k in d})
print(f"{'config':8s} " + " ".join(f"{k:>12s}" for k in kinds) + f" {'total':>10s}")
for name in pipes:
    mins = {k: min(d.get(k, 0.0) for d in res[name]) for k in kinds}
    print(f"{name:8s} " + " ".join(f"{mins[k]:12.3f}" for k in kinds) + f" {min(sum(d.values()) for d in res[name]):10.3f}")
key = lambda row: tuple((k, row[k]) for k in row if k not in ("launches", "ms", "tflops", "gbps"))
bm = {key(r): r for r in shapes["plain"]}
print("--- fold vs plain, GEMM shapes")
for row in shapes["fold"]:
    b = bm.get(key(row))
    if row["kind"] == "gemm" and b and b["ms"] > 0.1:
        print(f"  {dict(key(row))}: {b['ms']:.3f} -> {row['ms']:.3f} ms")

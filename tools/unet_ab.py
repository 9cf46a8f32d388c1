#!/usr/bin/env python
"""A/B harness: ONE process, ONE pipeline build, several env-knob configurations of libsdv_hip.so, each timed with HIP
events around every launch of an eager SD-1.4 UNet forward (2B samples) - interleaved rounds, per-kind and per-shape ms.

usage: unet_ab.py B rounds "NAME=K1=V1,K2=V2" "NAME2=..."       (a config with no knobs: "base=")
"""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import EventProfiler  # noqa: E402
from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, hip  # noqa: E402


def main():
    B, rounds = int(sys.argv[1]), int(sys.argv[2])
    configs = []
    for spec in sys.argv[3:]:
        name, _, kv = spec.partition("=")
        configs.append((name, dict(x.split("=") for x in kv.split(",") if x)))
    knobs = sorted({k for _, c in configs for k in c})
    pipe = StableDiffusionWalkPipeline.from_pretrained("CompVis/stable-diffusion-v1-4", arch="sd14").to("cuda")
    emb = pipe.embed_text(["a cat"] * B)
    ctx = torch.cat([pipe._uncond_embeddings(None, B), emb.float()])
    pipe._schedule(50, 0.0)
    pipe.unet.prepare_context(ctx)
    pipe.unet.reserve(2 * B, 64, 64)
    x2 = torch.randn((2 * B * 64 * 64, 4), device="cuda").to(torch.bfloat16)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = {name: [] for name, _ in configs}
    shapes = {}
    for r in range(rounds + 1):
        for name, cfg in configs:
            for k in knobs:
                os.environ.pop(k, None)
            os.environ.update(cfg)
            prof = EventProfiler()
            hip.LAUNCH_HOOK = prof
            pipe.unet.forward(x2, 2 * B, 64, 64, step, cfg_shared=True)
            torch.cuda.synchronize()
            hip.LAUNCH_HOOK = None
            if r == 0:
                continue       # warm-up round
            s = prof.summary()
            res[name].append({k: v["ms"] for k, v in s.items()})
            shapes[name] = prof.by_shape()
    kinds = sorted({k for v in res.values() for d in v for k in d})
    print(f"B={B} rounds={rounds}  (ms per forward, min over rounds)")
    print(f"{'config':14s} " + " ".join(f"{k:>12s}" for k in kinds) + f" {'total':>10s}")
    for name, _ in configs:
        mins = {k: min(d.get(k, 0.0) for d in res[name]) for k in kinds}
        tot = min(sum(d.values()) for d in res[name])
        print(f"{name:14s} " + " ".join(f"{mins[k]:12.3f}" for k in kinds) + f" {tot:10.3f}")
    if os.environ.get("SDV_AB_SHAPES"):
        json.dump(shapes, open(os.environ["SDV_AB_SHAPES"], "w"), indent=1)
    # per-shape comparison of the first config against the others (last round)
    base = configs[0][0]
    key = lambda row: tuple((k, row[k]) for k in row if k not in ("launches", "ms", "tflops", "gbps"))
    bmap = {key(r): r for r in shapes[base]}
    for name, _ in configs[1:]:
        print(f"--- {name} vs {base}: shapes that moved by more than 3 %")
        for row in shapes[name]:
            b = bmap.get(key(row))
            if b and b["ms"] > 0.05 and abs(row["ms"] / b["ms"] - 1) > 0.03:
                print(f"  {dict(key(row))}: {b['ms']:.3f} -> {row['ms']:.3f} ms ({b['tflops']} -> {row['tflops']} TF)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Cache-blocked UNet forward (UNetEngine._segment): wall time of a whole eager SD-1.4 forward of 2B samples (two HIP events
around the forward, so inter-kernel gaps count) for several chunk sizes, interleaved rounds, ONE process.

usage: chunk_ab.py B rounds ROWS[@HW,HW..] ...   (ROWS = SDV_CHUNK_ROWS, tokens per chunk, 0 = whole batch; @ = SDV_CHUNK_LEVELS)
Writes the fastest setting to gpurun_out/best_chunk_mb.txt when that directory exists."""
import os
import statistics
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import StableDiffusionWalkPipeline  # noqa: E402


def main():
    B, rounds = int(sys.argv[1]), int(sys.argv[2])
    settings = sys.argv[3:]
    fp8 = os.environ.get("SDV_AB_FP8") == "1"
    pipe = StableDiffusionWalkPipeline.from_pretrained("CompVis/stable-diffusion-v1-4", arch="sd14", fp8=fp8).to("cuda")
    emb = pipe.embed_text(["a cat"] * B)
    ctx = torch.cat([pipe._uncond_embeddings(None, B), emb.float()])
    pipe._schedule(50, 0.0)
    pipe.unet.prepare_context(ctx)
    pipe.unet.reserve(2 * B, 64, 64)
    x2 = torch.randn((2 * B * 64 * 64, 4), device="cuda").to(torch.bfloat16)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    ms = {s: [] for s in settings}
    ref = None
    for r in range(rounds + 1):
        for s in settings:
            rows, _, levels = s.partition("@")
            os.environ["SDV_CHUNK_ROWS"] = rows
            os.environ.pop("SDV_CHUNK_LEVELS", None)
            if levels:
                os.environ["SDV_CHUNK_LEVELS"] = levels
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            eps = pipe.unet.forward(x2, 2 * B, 64, 64, step, cfg_shared=True)
            e1.record()
            torch.cuda.synchronize()
            if r == 0:
                if ref is None:
                    ref = eps.clone()
                else:
                    d = float((eps - ref).abs().max()) / float(ref.abs().max())
                    print(f"  chunk {s:>14s} vs {settings[0]}: max |d eps| / max |eps| = {d:.2e}, identical: {torch.equal(eps, ref)}")
                continue
            ms[s].append(e0.elapsed_time(e1))
        torch.cuda.empty_cache()
    print(f"B={B} ({2 * B} samples), eager forward wall ms (median, min..max over {rounds} rounds), peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    best = None
    for s in settings:
        med = statistics.median(ms[s])
        print(f"  SDV_CHUNK_ROWS={s:>14s}: {med:8.2f}  ({min(ms[s]):.2f} .. {max(ms[s]):.2f})")
        if best is None or med < best[1]:
            best = (s, med)
    print(f"fastest: SDV_CHUNK_ROWS={best[0]} ({best[1]:.2f} ms)")
    out = ROOT / "gpurun_out"
    if out.is_dir():
        (out / "best_chunk_mb.txt").write_text(best[0])


if __name__ == "__main__":
    main()

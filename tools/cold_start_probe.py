#!/usr/bin/env python
"""Where does the first call at a new batch size spend its time, and what do the reference's own batch sizes run at?
For B in argv (default 60 16 4 1): cold call (buffers + eager warm-up + capture + 50 replays + VAE), warm call, the split of the
graph build, and the per-shape launch table of one eager UNet forward (written to gpurun_out/shapes_b<B>.json).
usage: python tools/cold_start_probe.py [B ...]"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from stable_diffusion_videos_amd import StableDiffusionWalkPipeline  # noqa: E402


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [60, 16, 4, 1]
    dev = torch.device("cuda", 0)
    pipe = StableDiffusionWalkPipeline.from_pretrained("CompVis/stable-diffusion-v1-4", arch="sd14").to(dev)
    out_dir = ROOT / "gpurun_out"
    out_dir.mkdir(exist_ok=True)
    # load every kernel once (code-object load is per process, not per batch size)
    _, e, n = next(pipe.generate_inputs("a cat", "a dog", 42, 1337, (1, 4, 64, 64), np.linspace(0, 1, 2), 2))
    pipe(latents=n, text_embeddings=e, height=512, width=512, num_inference_steps=2, output_type="numpy_u8")
    torch.cuda.synchronize()
    for B in Bs:
        T = np.linspace(0.0, 1.0, 3 * B)
        gen = pipe.generate_inputs("a cat", "a dog", 42, 1337, (1, 4, 64, 64), T, B)

        def one():
            _, e, n = next(gen)
            t0 = time.perf_counter()
            pipe(latents=n, text_embeddings=e, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, eta=0.0,
                 output_type="numpy_u8")
            torch.cuda.synchronize()
            return time.perf_counter() - t0, e, n

        cold, _, _ = one()
        gb = dict(getattr(pipe, "last_graph_build", {}))
        warm, e, n = one()
        warm2, e, n = one()
        row = {"B": B, "cold_s": round(cold, 3), "warm_s": round(min(warm, warm2), 3), "warm_fps": round(B / min(warm, warm2), 3),
               "cold_fps": round(B / cold, 3), "graph_build": {k: round(v, 3) if isinstance(v, float) else v for k, v in gb.items()},
               "timings": pipe.last_timings}
        kp = bench.kernel_pass(pipe, e, n, 512, 50)
        shapes = kp.pop("_unet_shapes")
        row["unet_forward_event_ms"] = kp["unet_forward_event_ms"]
        row["unet_forward_wall_ms_eager"] = kp["unet_forward_wall_ms_eager"]
        row["buckets"] = {k: (v["ms"], v["launches"]) for k, v in kp.items() if isinstance(v, dict) and k.startswith("unet.")}
        (out_dir / f"shapes_b{B}.json").write_text(json.dumps(shapes, indent=1))
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""The reference README's first example (README.md:30-45 there) on the MI355X path.

    python examples/walk.py [--model-dir /path/to/diffusers/stable-diffusion-v1-4] [--frames 60] [--batch-size 64]

Without a local diffusers-layout checkpoint directory (there is no network here) the pipeline runs the same
architecture with seeded synthetic weights: the frames are noise-like, the arithmetic and the output layout are real.
"""
import argparse

from stable_diffusion_videos_amd import StableDiffusionWalkPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--model-dir", default="CompVis/stable-diffusion-v1-4")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--batch-size", type=int, default=64)      # 288 GB of HBM: fill it (the reference default is 1)
ap.add_argument("--upsample", action="store_true")
args = ap.parse_args()

pipeline = StableDiffusionWalkPipeline.from_pretrained(args.model_dir, safety_checker=None).to("cuda")
video_path = pipeline.walk(
    prompts=["a cat", "a dog"],
    seeds=[42, 1337],
    num_interpolation_steps=args.frames,
    height=512,
    width=512,
    output_dir="dreams",
    name="animals_test",
    guidance_scale=8.5,
    num_inference_steps=50,
    batch_size=args.batch_size,
    upsample=args.upsample,
)
print(video_path)

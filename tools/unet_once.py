#!/usr/bin/env python
"""One eager SD-1.4 UNet forward (+ one VAE decode) of a B-frame batch - the workload for rocprofv3 --pmc passes.
usage: unet_once.py [B] [fp8]      fp8: BASELINE config 5 (e4m3 ResBlock convs; the first forward calibrates the scales)"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FP8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"
pipe = StableDiffusionWalkPipeline.from_pretrained("CompVis/stable-diffusion-v1-4", arch="sd14", fp8=FP8).to("cuda")
emb = pipe.embed_text(["a cat"] * B)
ctx = torch.cat([pipe._uncond_embeddings(None, B), emb.float()])
pipe._schedule(50, 0.0)
pipe.unet.prepare_context(ctx)
x2 = torch.randn((2 * B * 64 * 64, 4), device="cuda").to(torch.bfloat16)
step = torch.zeros(1, dtype=torch.int32, device="cuda")
for _ in range(3 if FP8 else 2):
    eps = pipe.unet.forward(x2, 2 * B, 64, 64, step)
torch.cuda.synchronize()
u8, _ = pipe.vae.decode(torch.randn((B, 64, 64, 4), device="cuda") * 0.18215)
torch.cuda.synchronize()
print("done", float(eps.abs().mean()), u8.shape)

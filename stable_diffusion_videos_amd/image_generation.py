"""``generate_images`` - the still-image helper of the reference
(/root/reference/stable_diffusion_videos/image_generation.py:81-215), SURVEY.md section 8(f) rank 4.  It is a thin
loop over the same ``pipeline(text_embeddings=, latents=)`` call the walk makes, so it inherits the whole HIP path
(text encoder, UNet, VAE and - with ``upsample=True`` - the Real-ESRGAN generator, fed GPU-resident uint8 frames).

Out of scope, loudly: ``push_to_hub`` (upload_folder_chunked, :39-78, needs the network; control plane, not the path).
Noise comes from ``pipeline.init_noise`` (CPU generator by default) where the reference draws from a
device generator - the same documented deviation as in ``walk``.
"""
from __future__ import annotations

import json
import random
import time
from pathlib import Path

import torch

from .utils import numpy_to_pil


def _as_dict(cfg):
    return dict(cfg) if isinstance(cfg, dict) else dict(vars(cfg))


def generate_input_batches(pipeline, prompts, seeds, batch_size, height, width):
    """Reference :81-105.  Yields ``(batch_idx, embeds (b,77,D), noise (b,C,h/8,w/8))``."""
    if len(prompts) != len(seeds):
        raise ValueError("Number of prompts and seeds must be equal.")
    embeds_batch, noise_batch = [], []
    batch_idx = 0
    for i, (prompt, seed) in enumerate(zip(prompts, seeds)):
        embeds_batch.append(pipeline.embed_text(prompt))
        noise_batch.append(pipeline.init_noise(seed, (1, pipeline.unet.in_channels, height // 8, width // 8)))
        if len(embeds_batch) != batch_size and i + 1 != len(prompts):
            continue
        yield batch_idx, torch.cat(embeds_batch), torch.cat(noise_batch)
        batch_idx += 1
        embeds_batch, noise_batch = [], []


def generate_images(pipeline, prompt, batch_size=1, num_batches=1, seeds=None, num_inference_steps=50, guidance_scale=7.5,
                    output_dir="./images", image_file_ext=".jpg", upsample=False, height=512, width=512, eta=0.0,
                    push_to_hub=False, repo_id=None, private=False, create_pr=False, name=None):
    """Reference :108-215: ``batch_size * num_batches`` images of one prompt, one seed each, saved as
    ``{output_dir}/{name}/{seed}{ext}`` next to a ``prompt_config.json``; returns the file paths."""
    if push_to_hub:
        if repo_id is None:
            raise ValueError("Must provide repo_id if push_to_hub is True.")
        raise NotImplementedError("push_to_hub needs the Hugging Face Hub (no network here; outside the MI355X hot path)")
    name = name or time.strftime("%Y%m%d-%H%M%S")
    save_path = Path(output_dir) / name
    save_path.mkdir(exist_ok=False, parents=True)
    num_images = batch_size * num_batches
    seeds = seeds or [random.choice(range(0, 9999999)) for _ in range(num_images)]
    if len(seeds) != num_images:
        raise ValueError("Number of seeds must be equal to batch_size * num_batches.")
    if upsample:
        if getattr(pipeline, "upsampler", None) is None:
            from .upsampling import RealESRGANModel
            pipeline.upsampler = RealESRGANModel.from_pretrained("nateraw/real-esrgan")
        pipeline.upsampler.to(pipeline.device)
    from . import __version__
    cfg = dict(prompt=prompt, guidance_scale=guidance_scale, eta=eta, num_inference_steps=num_inference_steps,
               upsample=upsample, height=height, width=width, scheduler=_as_dict(pipeline.scheduler.config),
               tiled=pipeline.tiled, stable_diffusion_videos_amd_version=__version__,
               device_name=torch.cuda.get_device_name(0) if torch.cuda.is_available() else "unknown")
    (save_path / "prompt_config.json").write_text(json.dumps(cfg, indent=2, sort_keys=False, default=str))
    frame_index = 0
    frame_filepaths = []
    for batch_idx, embeds, noise in generate_input_batches(pipeline, [prompt] * num_images, seeds, batch_size, height, width):
        print(f"Generating batch {batch_idx}")
        outputs = pipeline(text_embeddings=embeds, latents=noise, num_inference_steps=num_inference_steps,
                           guidance_scale=guidance_scale, eta=eta, height=height, width=width,
                           output_type="pil" if not upsample else "u8_cuda")["images"]
        images = numpy_to_pil(pipeline.upsampler.upsample_u8(outputs).cpu().numpy()) if upsample else outputs
        for image in images:
            frame_filepath = save_path / f"{seeds[frame_index]}{image_file_ext}"
            image.save(frame_filepath)
            frame_filepaths.append(str(frame_filepath))
            frame_index += 1
    return frame_filepaths

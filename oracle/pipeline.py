"""CPU oracle (test infrastructure): the reference's per-batch control flow, restated.

Follows /root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:

* :308-358  batch size from ``text_embeddings``, unconditional embeddings repeated per frame,
            ``cat([uncond, cond])``
* :389-401  latents shape check, ``set_timesteps``, ``* init_noise_sigma``
* :412-430  the denoise loop: ``cat([latents]*2)`` -> ``scale_model_input`` -> UNet ->
            ``eps_u + g (eps_c - eps_u)`` -> ``scheduler.step(...).prev_sample``
* :432-438  ``1/0.18215 * latents`` -> ``vae.decode`` -> ``(x/2+0.5).clamp(0,1)`` -> NHWC float32
* :450      ``numpy_to_pil``: ``(img*255).round().astype(uint8)`` (diffusers DiffusionPipeline)
* :481-554  ``make_clip_frames`` (T = linspace, generate_inputs, per-batch call)

Modules (``oracle.models``) and scheduler (``oracle.scheduler``) are restatements of un-vendored
diffusers code: **parity unpinned** for everything below the interpolation step.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch

from . import interp
from .scheduler import DDIMScheduler


@torch.no_grad()
def denoise_and_decode(unet, vae, scheduler: DDIMScheduler, text_embeddings: torch.Tensor,
                       uncond_embeddings: torch.Tensor, latents: torch.Tensor, num_inference_steps: int = 50,
                       guidance_scale: float = 7.5, eta: float = 0.0,
                       callback: Optional[Callable] = None, return_latents: bool = False,
                       variance_noise: Optional[List[torch.Tensor]] = None):
    """One ``__call__`` of the reference with ``text_embeddings=`` and ``latents=`` supplied
    (that is how ``make_clip_frames`` :538-548 invokes it).  Returns float32 NHWC images in [0,1]."""
    batch = text_embeddings.shape[0]                                             # :308
    do_cfg = guidance_scale > 1.0                                                # :318
    if do_cfg:
        uncond = uncond_embeddings.repeat(batch, 1, 1)                           # :352-353
        ctx = torch.cat([uncond, text_embeddings])                               # :358
    else:
        ctx = text_embeddings
    scheduler.set_timesteps(num_inference_steps)                                 # :394
    latents = latents * scheduler.init_noise_sigma                               # :401
    for i, t in enumerate(scheduler.timesteps):                                  # :412
        x_in = torch.cat([latents] * 2) if do_cfg else latents                   # :414
        x_in = scheduler.scale_model_input(x_in, t)                              # :415
        eps = unet(x_in, t, ctx)                                                 # :418
        if do_cfg:
            eps_u, eps_c = eps.chunk(2)                                          # :422
            eps = eps_u + guidance_scale * (eps_c - eps_u)                       # :423
        vn = variance_noise[i] if variance_noise is not None else None
        latents = scheduler.step(eps, t, latents, eta=eta, variance_noise=vn)    # :426
        if callback is not None:
            callback(i, t, latents)                                              # :429-430
    if return_latents:
        return latents
    return decode_latents(vae, latents)


@torch.no_grad()
def decode_latents(vae, latents: torch.Tensor) -> np.ndarray:
    latents = 1 / 0.18215 * latents                                              # :432
    image = vae.decode(latents)                                                  # :433
    image = (image / 2 + 0.5).clamp(0, 1)                                        # :435
    return image.cpu().permute(0, 2, 3, 1).float().numpy()                       # :438


def numpy_to_uint8(images: np.ndarray) -> np.ndarray:
    """diffusers ``numpy_to_pil`` arithmetic (called at :450): round-half-even, uint8."""
    return (images * 255).round().astype("uint8")


@torch.no_grad()
def make_clip_frames(unet, vae, scheduler, embeds_a, embeds_b, uncond_embeddings, seed_a, seed_b,
                     num_interpolation_steps: int, height: int, width: int, batch_size: int = 1,
                     num_inference_steps: int = 50, guidance_scale: float = 7.5, eta: float = 0.0,
                     T: Optional[np.ndarray] = None, skip: int = 0, in_channels: int = 4) -> np.ndarray:
    """:481-554 without the file writes: returns uint8 frames (n, H, W, 3)."""
    T = T if T is not None else np.linspace(0.0, 1.0, num_interpolation_steps)   # :509
    if T.shape[0] != num_interpolation_steps:                                    # :510
        raise ValueError(f"Unexpected T shape, got {T.shape}, expected dim 0 to be {num_interpolation_steps}")
    shape = (1, in_channels, height // 8, width // 8)                            # :523
    lat_a = interp.init_noise(seed_a, shape, embeds_a.dtype)                     # :461
    lat_b = interp.init_noise(seed_b, shape, embeds_a.dtype)                     # :462
    frames = []
    for _, embeds_batch, noise_batch in interp.generate_inputs(embeds_a, embeds_b, lat_a, lat_b, T[skip:],
                                                               batch_size):
        imgs = denoise_and_decode(unet, vae, scheduler, embeds_batch, uncond_embeddings, noise_batch,
                                  num_inference_steps, guidance_scale, eta)
        frames.append(numpy_to_uint8(imgs))
    return np.concatenate(frames, axis=0)

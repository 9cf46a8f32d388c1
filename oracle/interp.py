"""CPU oracle (test infrastructure): interpolation between walk endpoints.

Restates, in numpy, what the reference does per interpolated frame:

* ``slerp``  - /root/reference/stable_diffusion_videos/utils.py:42-66: WHOLE-TENSOR spherical
  interpolation of the two latent-noise endpoints: ``dot = sum(v0*v1 / (|v0| |v1|))``; if
  ``|dot| > 0.9995`` fall back to lerp, else ``sin((1-t)th)/sin(th) * v0 + sin(t th)/sin(th) * v1``
  with ``th = arccos(dot)``.  Arithmetic runs in the dtype of the inputs (numpy), exactly like the
  reference (fp32 in -> fp32 out, fp16 in -> fp16 out).
* ``lerp``   - stable_diffusion_pipeline.py:467 ``torch.lerp(embeds_a, embeds_b, t)`` on the text
  embeddings (NOT slerp - SURVEY.md section 0 fact 4).
* ``generate_inputs`` - stable_diffusion_pipeline.py:457-479 batching of the per-frame tensors.

PINNED: ``tests/golden/slerp_*.npz`` were generated in the build container by executing the
reference's own ``slerp`` (AST-lifted from utils.py) - see ``tests/golden/make_golden.py`` - and
``tests/test_oracle.py`` checks this restatement against them bit-for-bit.
"""
from __future__ import annotations

from typing import Iterator, Tuple

import numpy as np
import torch

DOT_THRESHOLD = 0.9995


def slerp_np(t: float, v0: np.ndarray, v1: np.ndarray, dot_threshold: float = DOT_THRESHOLD) -> np.ndarray:
    """utils.py:51-61 on numpy arrays (dtype-preserving)."""
    cosine = np.sum(v0 * v1 / (np.linalg.norm(v0) * np.linalg.norm(v1)))       # :51
    if np.abs(cosine) > dot_threshold:                                           # :52
        return (1 - t) * v0 + t * v1                                             # :53
    theta = np.arccos(cosine)                                                    # :55
    sin_theta = np.sin(theta)                                                    # :56
    theta_t = theta * t                                                          # :57
    w0 = np.sin(theta - theta_t) / sin_theta                                     # :59
    w1 = np.sin(theta_t) / sin_theta                                             # :60
    return w0 * v0 + w1 * v1                                                     # :61


def slerp(t: float, v0, v1, dot_threshold: float = DOT_THRESHOLD):
    """utils.py:42-66 including the torch -> numpy -> torch round trip (:45-49, :63-64)."""
    if isinstance(v0, torch.Tensor):
        dev = v0.device
        out = slerp_np(t, v0.cpu().numpy(), v1.cpu().numpy(), dot_threshold)
        return torch.from_numpy(np.asarray(out)).to(dev)
    return slerp_np(t, v0, v1, dot_threshold)


def lerp(a: torch.Tensor, b: torch.Tensor, t) -> torch.Tensor:
    """stable_diffusion_pipeline.py:467."""
    return torch.lerp(a, b, float(t))


def init_noise(seed: int, noise_shape, dtype=torch.float32) -> torch.Tensor:
    """stable_diffusion_pipeline.py:822-838 with the generator on the CPU (the
    new path draws endpoint noise from a CPU generator so seeds mean the same thing everywhere)."""
    return torch.randn(noise_shape, generator=torch.Generator(device="cpu").manual_seed(int(seed)), dtype=dtype)


def generate_inputs(embeds_a: torch.Tensor, embeds_b: torch.Tensor, latents_a: torch.Tensor,
                    latents_b: torch.Tensor, T: np.ndarray, batch_size: int
                    ) -> Iterator[Tuple[int, torch.Tensor, torch.Tensor]]:
    """stable_diffusion_pipeline.py:464-479 (endpoints already embedded / drawn)."""
    batch_idx = 0
    embeds_batch, noise_batch = [], []
    for i, t in enumerate(T):
        embeds_batch.append(lerp(embeds_a, embeds_b, t))                          # :467
        noise_batch.append(slerp(float(t), latents_a, latents_b))                 # :468
        if len(embeds_batch) == batch_size or i + 1 == T.shape[0]:               # :472
            yield batch_idx, torch.cat(embeds_batch), torch.cat(noise_batch)      # :475
            batch_idx += 1
            embeds_batch, noise_batch = [], []

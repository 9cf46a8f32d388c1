"""CPU oracle (test infrastructure): restatement of diffusers ``DDIMScheduler`` as the reference
drives it - ``set_timesteps`` (stable_diffusion_pipeline.py:394), ``init_noise_sigma`` (:401),
``scale_model_input`` (:415) and ``step(eps, t, x, eta=eta).prev_sample`` (:426).

Config is the one the reference's ``__init__`` forces (:85-110: ``steps_offset=1``,
``clip_sample=False``) on the SD-v1 schedule (``beta_start=0.00085, beta_end=0.012,
beta_schedule="scaled_linear", num_train_timesteps=1000, set_alpha_to_one=False``).
diffusers is un-vendored/unpinned (``pyproject.toml:14``): **parity unpinned**.
"""
from __future__ import annotations

import numpy as np
import torch


class DDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
                 prediction_type="epsilon"):
        if beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.clip_sample = clip_sample
        self.prediction_type = prediction_type
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts) + self.steps_offset

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _variance(self, t, prev_t):
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, variance_noise=None):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.prediction_type == "epsilon":
            pred_x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
            eps = model_output
        elif self.prediction_type == "v_prediction":
            pred_x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
            eps = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
        else:
            raise NotImplementedError(self.prediction_type)
        if self.clip_sample:
            pred_x0 = pred_x0.clamp(-1, 1)
        std = eta * self._variance(t, prev_t) ** 0.5
        direction = (1 - a_p - std ** 2) ** 0.5 * eps
        prev = a_p ** 0.5 * pred_x0 + direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return prev

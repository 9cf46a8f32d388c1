"""DDIM scheduler with the diffusers call surface the reference touches
(``set_timesteps`` :394, ``timesteps`` :398, ``init_noise_sigma`` :401, ``scale_model_input`` :415,
``step(...).prev_sample`` :426 of stable_diffusion_pipeline.py) plus the per-step coefficient table the
fused ``sdv_cfg_ddim_step`` kernel consumes.

The configuration is the one the reference's constructor enforces (``steps_offset=1``,
``clip_sample=False``; stable_diffusion_pipeline.py:85-110) on the SD schedule
(scaled_linear 0.00085..0.012, 1000 train steps, set_alpha_to_one=False).
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor


class DDIMScheduler:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", clip_sample: bool = False, set_alpha_to_one: bool = False,
                 steps_offset: int = 1, prediction_type: str = "epsilon"):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"beta_schedule {beta_schedule}")
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not supported (the reference forces it off, :99-110)")
        self.betas = betas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, clip_sample=clip_sample,
                                      set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                      prediction_type=prediction_type)
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    # -- diffusers surface --------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps: int, device=None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train or num_inference_steps <= 0:
            raise ValueError(f"num_inference_steps={num_inference_steps} must be in [1, {n_train}]")
        self.num_inference_steps = num_inference_steps
        ratio = n_train // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts) + self.config.steps_offset

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _alphas(self, t: int):
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_p

    def step_coefficients(self, t: int, eta: float = 0.0):
        """x_prev = c_x * x + c_e * model_output + sigma * z, in float64."""
        a_t, a_p = self._alphas(int(t))
        b_t = 1.0 - a_t
        var = (1.0 - a_p) / (1.0 - a_t) * (1.0 - a_t / a_p)
        sigma = eta * var ** 0.5
        d = (1.0 - a_p - sigma ** 2) ** 0.5
        if self.config.prediction_type == "epsilon":
            c_x = (a_p / a_t) ** 0.5
            c_e = d - (a_p * b_t / a_t) ** 0.5
        elif self.config.prediction_type == "v_prediction":
            c_x = (a_p * a_t) ** 0.5 + d * b_t ** 0.5
            c_e = d * a_t ** 0.5 - (a_p * b_t) ** 0.5
        else:
            raise NotImplementedError(self.config.prediction_type)
        return c_x, c_e, sigma

    def coefficient_table(self, eta: float = 0.0) -> torch.Tensor:
        """[steps, 4] fp32 rows {c_x, c_e, sigma, 0} for the current timestep schedule."""
        rows = [list(self.step_coefficients(int(t), eta)) + [0.0] for t in self.timesteps]
        return torch.tensor(rows, dtype=torch.float32)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0, generator=None,
             variance_noise: Optional[torch.Tensor] = None, return_dict: bool = True):
        """API-compatible single step on GPU tensors (NCHW or any layout - purely elementwise)."""
        from . import hip
        c_x, c_e, sigma = self.step_coefficients(int(timestep), eta)
        dev = sample.device
        coefs = torch.tensor([[c_x, c_e, sigma, 0.0]], dtype=torch.float32, device=dev)
        lat = sample.detach().to(torch.float32).contiguous().clone()
        eps = model_output.detach().to(torch.float32).contiguous()
        noise = None
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(sample.shape, generator=generator, dtype=torch.float32,
                                             device=generator.device if generator is not None else "cpu")
            noise = variance_noise.to(dev, torch.float32).contiguous()
        scratch = torch.empty(lat.numel(), dtype=torch.bfloat16, device=dev)
        hip.cfg_ddim_step(eps, lat, scratch, coefs, None, noise, 1.0, False, lat.numel())
        prev = lat.to(sample.dtype)
        return SchedulerOutput(prev) if return_dict else (prev,)

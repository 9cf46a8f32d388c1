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


# ======================================================================================================================
# The other schedulers of the reference's constructor signature (stable_diffusion_pipeline.py:71-78), as COEFFICIENT TABLES for
# the fused kernel ``sdv_cfg_multistep_step`` (include/sdv_hip.h): PNDM / PLMS (the SD-v1 default scheduler), LMSDiscrete
# (examples/make_music_video.py:15), EulerDiscrete, EulerAncestralDiscrete, DPM-Solver++ 2M.  Each keeps the diffusers call
# surface the reference touches (``set_timesteps`` :394, ``timesteps`` :398, ``init_noise_sigma`` :401, ``scale_model_input``
# :415, ``step(...).prev_sample`` :426); ``fused_table()`` is what the pipeline's denoise graph consumes.  One row per UNet
# evaluation (PLMS needs num_inference_steps + 1 of them):
#     x' = a x_base + c (w0 m + w1 H[-1] + w2 H[-2] + w3 H[-3]) + s_noise z,    m = u x + v model_output,   x2 = bf16(s_in x')
# ======================================================================================================================
F_PUSH, F_SAVE, F_USE_SAVED = 1, 2, 4


def _sd_alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule) -> np.ndarray:
    if beta_schedule == "scaled_linear":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    elif beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    else:
        raise NotImplementedError(f"beta_schedule {beta_schedule}")
    return torch.cumprod(1.0 - betas, dim=0).double().numpy()       # the fp32 cumprod diffusers computes, then exact


class _TableScheduler:
    """Common part: config, the row builder and a CPU/GPU ``step()`` that replays the table with torch ops (API parity for
    callers that drive the scheduler themselves; the pipeline uses the fused kernel)."""
    stochastic = False
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "epsilon", steps_offset: int = 1, **extra):
        if prediction_type not in ("epsilon", "v_prediction"):
            raise NotImplementedError(prediction_type)
        self.alphas_cumprod_f64 = _sd_alphas_cumprod(num_train_timesteps, beta_start, beta_end, beta_schedule)
        self.alphas_cumprod = torch.from_numpy(self.alphas_cumprod_f64).float()
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, prediction_type=prediction_type, steps_offset=steps_offset,
                                      **extra)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1)
        self._rows = None
        self._reset_state()

    # -- table ------------------------------------------------------------------------------------
    @staticmethod
    def _row(a, c, w=(1.0,), u=0.0, v=1.0, s_in=1.0, s_noise=0.0, flags=F_PUSH, head=0):
        w = list(w) + [0.0] * (4 - len(w))
        return [a, c, *w, u, v, s_in, s_noise, float(flags), float(head), 0.0, 0.0, 0.0, 0.0]

    def _build_rows(self):          # -> list of 16-float rows, one per entry of self.timesteps
        raise NotImplementedError

    def fused_table(self) -> torch.Tensor:
        """[len(timesteps), 16] fp32 (computed in float64) for ``sdv_cfg_multistep_step``."""
        if self._rows is None:
            rows = self._build_rows()
            head = 0
            for r in rows:                       # ring position of the history at every evaluation (host-known)
                r[11] = float(head)
                if int(r[10]) & F_PUSH:
                    head += 1
            self._rows = rows
        return torch.tensor(self._rows, dtype=torch.float32)

    def first_input_scale(self) -> float:
        """scale_model_input of the FIRST evaluation (the later ones ride in the rows' s_in)."""
        return 1.0

    # -- diffusers surface ------------------------------------------------------------------------
    def _reset_state(self):
        self._i, self._hist, self._xsave, self._head = 0, [None] * 4, None, 0

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, generator=None,
             variance_noise: Optional[torch.Tensor] = None, return_dict: bool = True, **_):
        """One ``scheduler.step`` by replaying row ``i`` of the table (calls must come in schedule order, as the reference
        makes them :412-426)."""
        rows = self.fused_table()
        i = self._i
        if i >= rows.shape[0]:
            raise RuntimeError("step() called more often than set_timesteps() scheduled")
        if timestep is not None and abs(float(timestep) - float(self.timesteps[i])) > 1e-3:
            raise ValueError(f"step() number {i} was given timestep {float(timestep)}, the schedule has {float(self.timesteps[i])} "
                             "there: the table-driven schedulers must be stepped in schedule order (as the reference does, :412-426)")
        a, c, w0, w1, w2, w3, u, v, s_in, s_noise, flags, head = (float(x) for x in rows[i, :12].double())
        flags, head = int(flags), int(head)
        x = sample.to(torch.float32)
        m = u * x + v * model_output.to(torch.float32)
        comb = w0 * m
        for wk, back in ((w1, 1), (w2, 2), (w3, 3)):
            if wk != 0.0:
                comb = comb + wk * self._hist[(head - back) & 3]
        xb = self._xsave if flags & F_USE_SAVED else x
        if flags & F_SAVE:
            self._xsave = x
        prev = a * xb + c * comb
        if s_noise != 0.0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=torch.float32,
                                             device=generator.device if generator is not None else "cpu")
            prev = prev + s_noise * variance_noise.to(prev.device, torch.float32)
        if flags & F_PUSH:
            self._hist[head & 3] = m
        self._i += 1
        prev = prev.to(sample.dtype)
        return SchedulerOutput(prev) if return_dict else (prev,)


class PNDMScheduler(_TableScheduler):
    """diffusers ``PNDMScheduler`` with ``skip_prk_steps=True`` (what every SD-v1 / v2 checkpoint ships): pseudo linear
    multistep.  The warm-up evaluates the second timestep twice, so N inference steps are N + 1 UNet evaluations."""
    order = 4

    def __init__(self, *args, skip_prk_steps: bool = True, set_alpha_to_one: bool = False, **kw):
        if not skip_prk_steps:
            raise NotImplementedError("PNDMScheduler(skip_prk_steps=False): the Runge-Kutta warm-up is not implemented "
                                      "(no Stable Diffusion checkpoint uses it)")
        super().__init__(*args, skip_prk_steps=True, set_alpha_to_one=set_alpha_to_one, **kw)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod_f64[0])

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        if not 2 <= num_inference_steps <= T:
            raise ValueError(f"num_inference_steps={num_inference_steps} must be in [2, {T}]")
        self.num_inference_steps = num_inference_steps
        base = (np.arange(0, num_inference_steps) * (T // num_inference_steps)).round() + self.config.steps_offset
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        self._rows = None
        self._reset_state()

    def _transfer(self, t: int, prev_t: int):
        """x_prev = A x + C eps' (Liu et al. 2022 eq. 9 as diffusers' _get_prev_sample writes it); for v-prediction the
        combined output is converted with the BASE sample, which stays linear: A' = A + C sqrt(b_t), C' = C sqrt(a_t)."""
        ac = self.alphas_cumprod_f64
        a_t = ac[t]
        a_p = ac[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1.0 - a_t, 1.0 - a_p
        A = (a_p / a_t) ** 0.5
        C = -(a_p - a_t) / (a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5)
        if self.config.prediction_type == "v_prediction":
            return A + C * b_t ** 0.5, C * a_t ** 0.5
        return A, C

    def _build_rows(self):
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        rows, n_ets = [], 0
        for j, t in enumerate(int(x) for x in self.timesteps):
            if j == 1:       # the repeated evaluation: Heun-style corrector from the sample saved at j = 0, nothing is pushed
                A, C = self._transfer(t + ratio, t)
                rows.append(self._row(A, C, w=(0.5, 0.5), flags=F_USE_SAVED))
                continue
            n_ets = min(n_ets + 1, 4)
            A, C = self._transfer(t, t - ratio)
            w = {1: (1.0,), 2: (1.5, -0.5), 3: (23 / 12, -16 / 12, 5 / 12), 4: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}[n_ets]
            rows.append(self._row(A, C, w=w, flags=F_PUSH | (F_SAVE if j == 0 else 0)))
        return rows


class _SigmaScheduler(_TableScheduler):
    """k-diffusion sigma space (Karras et al. 2022): sigma = sqrt((1 - abar) / abar), float timesteps on a linspace, the model
    sees x / sqrt(sigma^2 + 1); the solver variable is d = (x - x0_pred) / sigma (= eps for an eps-model)."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        s = ((1.0 - self.alphas_cumprod_f64) / self.alphas_cumprod_f64) ** 0.5
        self.init_noise_sigma = float(np.float32(s.max()))
        self.set_timesteps(self.config.num_train_timesteps)

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        if not 1 <= num_inference_steps <= T:
            raise ValueError(f"num_inference_steps={num_inference_steps} must be in [1, {T}]")
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, T - 1, num_inference_steps, dtype=float)[::-1].copy()
        s = ((1.0 - self.alphas_cumprod_f64) / self.alphas_cumprod_f64) ** 0.5
        s = np.interp(ts, np.arange(0, T), s)
        self._sig = np.concatenate([s, [0.0]]).astype(np.float32).astype(np.float64)      # diffusers keeps them in fp32
        self.sigmas = torch.from_numpy(self._sig).float()
        self.timesteps = torch.from_numpy(ts)
        self._rows = None
        self._reset_state()

    def first_input_scale(self) -> float:
        return float(1.0 / (self._sig[0] ** 2 + 1.0) ** 0.5)

    def scale_model_input(self, sample, timestep):
        idx = (self.timesteps == float(timestep)).nonzero()
        sigma = float(self._sig[int(idx[0])])
        return sample / ((sigma ** 2 + 1.0) ** 0.5)

    def _uv(self, sigma: float):
        if self.config.prediction_type == "epsilon":
            return 0.0, 1.0                                       # d = eps
        return sigma / (sigma ** 2 + 1.0), 1.0 / (sigma ** 2 + 1.0) ** 0.5        # d for a v-model

    def _s_in(self, i: int) -> float:
        return float(1.0 / (self._sig[i + 1] ** 2 + 1.0) ** 0.5)


class EulerDiscreteScheduler(_SigmaScheduler):
    def _build_rows(self):
        rows = []
        for i in range(self.num_inference_steps):
            u, v = self._uv(self._sig[i])
            rows.append(self._row(1.0, self._sig[i + 1] - self._sig[i], u=u, v=v, s_in=self._s_in(i), flags=0))
        return rows


class EulerAncestralDiscreteScheduler(_SigmaScheduler):
    stochastic = True

    def _build_rows(self):
        rows = []
        for i in range(self.num_inference_steps):
            s_from, s_to = self._sig[i], self._sig[i + 1]
            s_up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
            s_down = (s_to ** 2 - s_up ** 2) ** 0.5
            u, v = self._uv(s_from)
            rows.append(self._row(1.0, s_down - s_from, u=u, v=v, s_in=self._s_in(i), s_noise=s_up, flags=0))
        return rows


class LMSDiscreteScheduler(_SigmaScheduler):
    """Linear multistep (Adams-Bashforth on the sigma grid, order 4): the weights are integrals of the Lagrange basis over
    [sigma_i, sigma_{i+1}] - polynomials, so they are integrated exactly (Gauss-Legendre) instead of diffusers'
    ``scipy.integrate.quad(..., epsrel=1e-4)``; the two agree to the quadrature's tolerance."""
    order = 4

    def _lms_coefficient(self, order: int, i: int, k: int) -> float:
        nodes, weights = np.polynomial.legendre.leggauss(4)       # exact for degree <= 7 (the basis has degree order - 1 <= 3)
        lo, hi = self._sig[i], self._sig[i + 1]
        tau = 0.5 * (hi - lo) * nodes + 0.5 * (hi + lo)
        val = np.ones_like(tau)
        for j in range(order):
            if j != k:
                val = val * (tau - self._sig[i - j]) / (self._sig[i - k] - self._sig[i - j])
        return float(0.5 * (hi - lo) * np.sum(weights * val))

    def _build_rows(self):
        rows = []
        for i in range(self.num_inference_steps):
            order = min(i + 1, self.order)
            u, v = self._uv(self._sig[i])
            rows.append(self._row(1.0, 1.0, w=[self._lms_coefficient(order, i, k) for k in range(order)], u=u, v=v,
                                  s_in=self._s_in(i), flags=F_PUSH))
        return rows


class DPMSolverMultistepScheduler(_TableScheduler):
    """DPM-Solver++ (2M, midpoint): data prediction x0 = (x - sigma eps) / alpha as the solver variable."""
    order = 2

    def __init__(self, *args, solver_order: int = 2, algorithm_type: str = "dpmsolver++", solver_type: str = "midpoint",
                 lower_order_final: bool = True, **kw):
        if solver_order not in (1, 2) or algorithm_type != "dpmsolver++" or solver_type != "midpoint":
            raise NotImplementedError("DPMSolverMultistepScheduler: dpmsolver++ / midpoint, solver_order 1 or 2")
        super().__init__(*args, solver_order=solver_order, algorithm_type=algorithm_type, solver_type=solver_type,
                         lower_order_final=lower_order_final, **kw)

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        if not 1 <= num_inference_steps <= T:
            raise ValueError(f"num_inference_steps={num_inference_steps} must be in [1, {T}]")
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, T - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)
        self._rows = None
        self._reset_state()

    def _build_rows(self):
        ac = self.alphas_cumprod_f64
        alpha, sigma = ac ** 0.5, (1.0 - ac) ** 0.5
        lam = np.log(alpha) - np.log(sigma)
        ts = [int(t) for t in self.timesteps]
        rows = []
        for i, s0 in enumerate(ts):
            t = 0 if i == len(ts) - 1 else ts[i + 1]
            h = lam[t] - lam[s0]
            a = sigma[t] / sigma[s0]
            c = -alpha[t] * (np.exp(-h) - 1.0)
            if self.config.prediction_type == "epsilon":
                u, v = 1.0 / alpha[s0], -sigma[s0] / alpha[s0]
            else:
                u, v = alpha[s0], -sigma[s0]
            lower_final = i == len(ts) - 1 and self.config.lower_order_final and len(ts) < 15
            if self.config.solver_order == 1 or i == 0 or lower_final:
                w = (1.0,)
            else:
                r0 = (lam[s0] - lam[ts[i - 1]]) / h
                w = (1.0 + 0.5 / r0, -0.5 / r0)
            rows.append(self._row(a, c, w=w, u=u, v=v, flags=F_PUSH))
        return rows


SCHEDULERS = {c.__name__: c for c in (DDIMScheduler, PNDMScheduler, LMSDiscreteScheduler, EulerDiscreteScheduler,
                                      EulerAncestralDiscreteScheduler, DPMSolverMultistepScheduler)}


# Config keys each class honours (everything its __init__ acts on).  A foreign scheduler's config / a checkpoint's
# scheduler_config.json is forwarded key by key, so that the unsupported-value guards of the constructors stay active
# (diffusers' PNDMScheduler defaults to skip_prk_steps=False, its DDIMScheduler to set_alpha_to_one=True - ADVICE r3: both used to
# be dropped and silently replaced by this module's defaults).
_COMMON_KEYS = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "prediction_type", "steps_offset")
_CONFIG_KEYS = {
    "DDIMScheduler": _COMMON_KEYS + ("set_alpha_to_one",),          # clip_sample: forced off, as the reference does (:99-110)
    "PNDMScheduler": _COMMON_KEYS + ("skip_prk_steps", "set_alpha_to_one"),
    "LMSDiscreteScheduler": _COMMON_KEYS,
    "EulerDiscreteScheduler": _COMMON_KEYS,
    "EulerAncestralDiscreteScheduler": _COMMON_KEYS,
    "DPMSolverMultistepScheduler": _COMMON_KEYS + ("solver_order", "algorithm_type", "solver_type", "lower_order_final"),
}
# recognised diffusers options that change the arithmetic and are NOT implemented: anything but the listed "off" value raises
_UNSUPPORTED_KEYS = {"trained_betas": (None,), "thresholding": (False, None), "use_karras_sigmas": (False, None),
                     "use_exponential_sigmas": (False, None), "use_beta_sigmas": (False, None), "use_lu_lambdas": (False, None),
                     "rescale_betas_zero_snr": (False, None), "euler_at_final": (False, None),
                     "variance_type": (None, "fixed_small"), "interpolation_type": (None, "linear"),
                     "final_sigmas_type": (None, "zero"), "use_flow_sigmas": (False, None),
                     "timestep_type": (None, "discrete")}
# timestep_spacing: each class implements the spacing its diffusers namesake defaults to
_SPACING = {"DDIMScheduler": "leading", "PNDMScheduler": "leading", "LMSDiscreteScheduler": "linspace",
            "EulerDiscreteScheduler": "linspace", "EulerAncestralDiscreteScheduler": "linspace",
            "DPMSolverMultistepScheduler": "linspace"}


def kwargs_from_config(class_name: str, cfg) -> dict:
    """Constructor keywords of ``SCHEDULERS[class_name]`` from a diffusers-style config (mapping or attribute object)."""
    has = (lambda k: k in cfg) if isinstance(cfg, dict) else (lambda k: hasattr(cfg, k))
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    for k, ok in _UNSUPPORTED_KEYS.items():
        if has(k) and get(k) not in ok:
            raise NotImplementedError(f"{class_name}: config option {k}={get(k)!r} is not implemented by the table-driven "
                                      f"schedulers (supported: {ok[0]!r})")
    if has("timestep_spacing") and get("timestep_spacing") not in (None, _SPACING[class_name]):
        raise NotImplementedError(f"{class_name}: timestep_spacing={get('timestep_spacing')!r} is not implemented "
                                  f"(this class spaces its timesteps {_SPACING[class_name]!r}, its diffusers default)")
    return {k: get(k) for k in _CONFIG_KEYS[class_name] if has(k)}


def adopt(scheduler):
    """A scheduler object of THIS module is returned as it is; a foreign one with a known class name (a diffusers scheduler
    handed to ``from_pretrained(scheduler=...)``, examples/make_music_video.py:15) is rebuilt here from its config - every key
    the class acts on is carried over, recognised-but-unsupported options raise."""
    if isinstance(scheduler, (DDIMScheduler, _TableScheduler)):
        return scheduler
    name = type(scheduler).__name__
    cls = SCHEDULERS.get(name)
    if cls is None:
        raise NotImplementedError(f"{name}: not one of the schedulers the reference accepts "
                                  f"(stable_diffusion_pipeline.py:71-78): {sorted(SCHEDULERS)}")
    return cls(**kwargs_from_config(name, getattr(scheduler, "config", None) or {}))

"""CPU tests of the schedulers besides DDIM (stable_diffusion_pipeline.py:71-78: PNDM / LMS / Euler / EulerAncestral /
DPM-Solver++), which the product implements as per-evaluation coefficient TABLES for one fused kernel
(stable_diffusion_videos_amd/scheduler.py, include/sdv_hip.h ``sdv_cfg_multistep_step``):

  * against the oracle's classic stateful restatements (oracle/scheduler.py) on the same model outputs - two independent
    forms of each algorithm;
  * against the samplers' defining property, which needs no diffusers: fed the TRUE noise of x_t = alpha_t x0 + sigma_t eps,
    a consistent solver lands on the marginal of the next noise level at every step;
  * the LMS weights against diffusers' own recipe (scipy.integrate.quad of the Lagrange basis)."""
import numpy as np
import pytest
import torch

from oracle import scheduler as O
from stable_diffusion_videos_amd import scheduler as P

NAMES = ["PNDMScheduler", "LMSDiscreteScheduler", "EulerDiscreteScheduler", "EulerAncestralDiscreteScheduler",
         "DPMSolverMultistepScheduler"]


@pytest.mark.parametrize("ptype", ["epsilon", "v_prediction"])
@pytest.mark.parametrize("N", [50, 10, 3])
@pytest.mark.parametrize("name", NAMES)
def test_table_driven_step_equals_the_classic_form(name, N, ptype):
    o, p = getattr(O, name)(prediction_type=ptype), getattr(P, name)(prediction_type=ptype)
    o.set_timesteps(N)
    p.set_timesteps(N)
    assert torch.equal(o.timesteps.double(), p.timesteps.double())
    assert len(p.timesteps) == (N + 1 if name == "PNDMScheduler" else N)      # PLMS evaluates the second timestep twice
    assert abs(float(o.init_noise_sigma) - float(p.init_noise_sigma)) < 1e-5
    tab = p.fused_table()
    assert tab.shape == (len(p.timesteps), 16) and tab.dtype == torch.float32 and bool(torch.isfinite(tab).all())
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64) * float(o.init_noise_sigma)
    xo, xp = x.clone(), x.float()
    for t in o.timesteps:
        e = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
        z = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
        assert torch.allclose(o.scale_model_input(xo, t).float(), p.scale_model_input(xo.float(), t), rtol=1e-5)
        xo = o.step(e, t, xo, variance_noise=z)
        xp = p.step(e.float(), t, xp, variance_noise=z.float()).prev_sample
        assert float((xo - xp.double()).abs().max()) <= 5e-6 * float(xo.abs().max()), (name, float(t))
    with pytest.raises(RuntimeError):
        p.step(e.float(), t, xp)                                          # more steps than scheduled


@pytest.mark.parametrize("name", ["PNDMScheduler", "LMSDiscreteScheduler", "EulerDiscreteScheduler", "DPMSolverMultistepScheduler"])
@pytest.mark.parametrize("ptype", ["epsilon", "v_prediction"])
def test_true_noise_walks_down_the_forward_marginals(name, ptype):
    if name == "PNDMScheduler" and ptype == "v_prediction":
        # diffusers' PLMS mixes the RAW model outputs of four timesteps and converts v -> eps afterwards with the current
        # timestep's (alpha, sigma): v is not the same quantity at different t, so even a perfect v-model leaves a residual
        # (0.078 here, in the oracle's classic form and in the table form alike - the equality test above covers both)
        pytest.skip("PLMS with v-prediction is not marginal-preserving by construction")
    p = getattr(P, name)(prediction_type=ptype)
    p.set_timesteps(50)
    g = torch.Generator().manual_seed(3)
    x0, eps = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    ac = torch.from_numpy(p.alphas_cumprod_f64).float()
    sigma_space = hasattr(p, "sigmas")

    def marginal(level):          # sigma space: x0 + sigma eps;  VP space: sqrt(abar) x0 + sqrt(1 - abar) eps
        return x0 + level * eps if sigma_space else level.sqrt() * x0 + (1 - level).sqrt() * eps

    def model_output(x, i):       # what a perfect eps- or v-model answers at evaluation i
        if ptype == "epsilon":
            return eps
        if sigma_space:           # the model sees x / sqrt(sigma^2 + 1) at abar = 1 / (sigma^2 + 1)
            a = 1.0 / (p.sigmas[i] ** 2 + 1.0)
        else:
            a = ac[int(p.timesteps[i])]
        return a.sqrt() * eps - (1 - a).sqrt() * x0
    x = marginal(p.sigmas[0] if sigma_space else ac[int(p.timesteps[0])])
    for i, t in enumerate(p.timesteps):
        x = p.step(model_output(x, i), t, x).prev_sample
    end = p.sigmas[-1] if sigma_space else ac[0]
    assert float((x - marginal(end)).abs().max()) < 2e-4


def test_lms_weights_equal_the_quadrature_of_the_lagrange_basis():
    from scipy import integrate
    p = P.LMSDiscreteScheduler()
    p.set_timesteps(50)
    sig = p.sigmas.double().numpy()
    for i in (0, 1, 2, 3, 10, 49):
        order = min(i + 1, 4)
        for k in range(order):
            def basis(tau):
                v = 1.0
                for j in range(order):
                    if j != k:
                        v *= (tau - sig[i - j]) / (sig[i - k] - sig[i - j])
                return v
            ref = integrate.quad(basis, sig[i], sig[i + 1], epsrel=1e-4)[0]
            assert abs(p._lms_coefficient(order, i, k) - ref) <= 1e-4 * max(abs(ref), 1e-3)
    rows = p.fused_table()
    assert abs(float(rows[10, 2:6].double().sum()) - (sig[11] - sig[10])) < 1e-5     # the weights integrate the constant 1


def test_foreign_scheduler_objects_are_adopted_by_class_name():
    from types import SimpleNamespace
    LMSDiscreteScheduler = type("LMSDiscreteScheduler", (), {})          # "diffusers.schedulers.LMSDiscreteScheduler"
    foreign = LMSDiscreteScheduler()
    foreign.config = {"beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear", "num_train_timesteps": 1000}
    mine = P.adopt(foreign)
    assert isinstance(mine, P.LMSDiscreteScheduler) and abs(mine.init_noise_sigma - 14.6146) < 1e-3
    assert P.adopt(mine) is mine
    with pytest.raises(NotImplementedError, match="not one of the schedulers"):
        P.adopt(SimpleNamespace(config=SimpleNamespace()))
    with pytest.raises(NotImplementedError):
        P.PNDMScheduler(skip_prk_steps=False)
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline
    pipe = StableDiffusionWalkPipeline.from_pretrained("tiny", scheduler=foreign)
    assert isinstance(pipe.scheduler, P.LMSDiscreteScheduler)


def test_checkpoint_directory_picks_its_own_scheduler(tmp_path):
    """diffusers' from_pretrained instantiates scheduler/scheduler_config.json's _class_name (SD-v1: PNDMScheduler)."""
    import json
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline as Pipe
    from stable_diffusion_videos_amd import config as cfgs
    from stable_diffusion_videos_amd import weights
    from safetensors.torch import save_file
    uc, vc = cfgs.tiny_unet(), cfgs.tiny_vae()
    for sub, shapes, c in (("unet", weights.unet_shapes(uc), uc), ("vae", weights.vae_decoder_shapes(vc), vc)):
        (tmp_path / sub).mkdir()
        save_file(weights.synthetic_state_dict(shapes, seed=0), str(tmp_path / sub / "diffusion_pytorch_model.safetensors"))
        (tmp_path / sub / "config.json").write_text(json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(c).items()}))
    (tmp_path / "scheduler").mkdir()
    (tmp_path / "scheduler" / "scheduler_config.json").write_text(json.dumps(
        {"_class_name": "PNDMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
         "num_train_timesteps": 1000, "skip_prk_steps": True, "steps_offset": 1, "set_alpha_to_one": False}))
    pipe = Pipe.from_pretrained(str(tmp_path))
    assert isinstance(pipe.scheduler, P.PNDMScheduler)
    pipe.scheduler.set_timesteps(50)
    assert pipe.scheduler.timesteps[:4].tolist() == [981, 961, 961, 941] and len(pipe.scheduler.timesteps) == 51


def test_adopt_forwards_every_config_key_and_keeps_the_guards_active():
    """adopt() used to rebuild a foreign scheduler from five keys only, so diffusers' own defaults -
    DDIMScheduler(set_alpha_to_one=True), PNDMScheduler(skip_prk_steps=False) - were silently replaced by this module's."""
    def foreign(name, **cfg):
        obj = type(name, (), {})()
        obj.config = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", **cfg)
        return obj

    d1 = P.adopt(foreign("DDIMScheduler", set_alpha_to_one=True, steps_offset=0, clip_sample=True))
    d0 = P.adopt(foreign("DDIMScheduler", set_alpha_to_one=False))
    assert float(d1.final_alpha_cumprod) == 1.0 and float(d0.final_alpha_cumprod) == float(d0.alphas_cumprod[0])
    assert d1.config.steps_offset == 0 and d1.config.clip_sample is False          # clip_sample is forced off (:99-110)
    d1.set_timesteps(50)
    d0.set_timesteps(50)
    assert not torch.allclose(d1.coefficient_table(0.0)[-1], d0.coefficient_table(0.0)[-1])       # the last step differs
    with pytest.raises(NotImplementedError, match="Runge-Kutta"):
        P.adopt(foreign("PNDMScheduler", skip_prk_steps=False))                      # diffusers' default PNDM
    assert P.adopt(foreign("PNDMScheduler", skip_prk_steps=True, set_alpha_to_one=True)).final_alpha_cumprod == 1.0
    dp = P.adopt(foreign("DPMSolverMultistepScheduler", solver_order=1, lower_order_final=False, prediction_type="v_prediction"))
    assert dp.config.solver_order == 1 and dp.config.lower_order_final is False and dp.config.prediction_type == "v_prediction"
    with pytest.raises(NotImplementedError):
        P.adopt(foreign("DPMSolverMultistepScheduler", algorithm_type="sde-dpmsolver++"))
    for bad in (dict(use_karras_sigmas=True), dict(thresholding=True), dict(trained_betas=[0.1, 0.2]),
                dict(rescale_betas_zero_snr=True), dict(timestep_spacing="trailing")):
        with pytest.raises(NotImplementedError, match="not implemented"):
            P.adopt(foreign("EulerDiscreteScheduler", **bad))
    assert isinstance(P.adopt(foreign("EulerDiscreteScheduler", use_karras_sigmas=False, timestep_spacing="linspace")),
                      P.EulerDiscreteScheduler)


def test_table_schedulers_refuse_out_of_order_steps():
    s = P.PNDMScheduler()
    s.set_timesteps(5)
    x = torch.zeros(1, 4, 2, 2)
    ts = [int(t) for t in s.timesteps]
    x = s.step(torch.zeros_like(x), ts[0], x).prev_sample
    x = s.step(torch.zeros_like(x), ts[1], x).prev_sample
    with pytest.raises(ValueError, match="schedule order"):
        s.step(torch.zeros_like(x), ts[4], x)

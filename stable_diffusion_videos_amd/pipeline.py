"""``StableDiffusionWalkPipeline`` - drop-in for the reference class of the same name
(/root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:38-858) on MI355X.

Same public methods, keyword arguments, error behaviour, on-disk layout and ``prompt_config.json``;
the arithmetic behind them runs in hand-written gfx950 HIP kernels (libsdv_hip.so) instead of
diffusers/ATen:

    walk            :556-807   orchestration, prompt_config.json (same keys, same order); ``resume`` regenerates the
                               MISSING-frame set - the reference's "continue after the last frame on disk" rule and its
                               :747-752 quirk (a clip with exactly one missing frame is skipped) are deliberately NOT
                               reproduced, see walk()
    make_clip_frames:481-554   T = linspace, batches, frame%06d.png
    generate_inputs :457-479   lerp(text embeddings) + slerp(noise) - ONE fused launch per batch, on device
    __call__        :191-455   CFG + 50-step DDIM loop (one hipGraph replay per step) + VAE decode + uint8
    embed_text      :809-820   tokenizer + CLIP text encoder
    init_noise      :822-838   seeded N(0,1) endpoints (CPU generator by default so seeds are portable)

Deliberate, documented differences:
  * endpoint noise is drawn from a CPU generator and uploaded (``noise_device="cpu"``) so the same seeds give
    the same walk on any device and can be compared with a CPU run;
  * slerp runs in fp32 on the GPU (the reference's numpy round trip raises on bf16, fact 5);
  * the unconditional embedding, time-embedding tables and cross-attention K/V are computed once per call
    instead of once per step; PNG encoding is asynchronous;
  * frames can be sharded across ranks (torch.distributed / RCCL) - see ``parallel.py``.
"""
from __future__ import annotations

import json
import logging
import math
import os
import time
from pathlib import Path
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import config as cfgs
from . import hip, parallel, weights
from .engine import UNetEngine, VAEDecoderEngine
from .scheduler import DDIMScheduler, adopt as adopt_scheduler
from .text import build_text_encoder, load_tokenizer
from .utils import FrameWriter, get_timesteps_arr, make_video_pyav, numpy_to_pil

logger = logging.getLogger("stable_diffusion_videos_amd")

F32 = torch.float32
BF16 = torch.bfloat16


class StableDiffusionPipelineOutput(dict):
    """``outputs["images"]`` / ``outputs.images`` / ``outputs.nsfw_content_detected`` (reference :455, :548)."""

    def __init__(self, images, nsfw_content_detected=None):
        super().__init__(images=images, nsfw_content_detected=nsfw_content_detected)
        self.images = images
        self.nsfw_content_detected = nsfw_content_detected


class _PendingModule:
    """Host-side (CPU) parameters of a network before ``pipeline.to("cuda")`` builds its HIP engine."""

    def __init__(self, kind: str, config, state_dict):
        self.kind, self.config, self.state_dict = kind, config, state_dict
        if kind == "unet":
            self.in_channels = config.in_channels


class _ForeignVAE:
    """Adapter for a ``vae=`` object that is not one of this package's engines (the reference forwards ``vae=`` to diffusers:
    examples/make_music_video.py:14 passes a fine-tuned AutoencoderKL).  Anything with ``decode(z)`` returning a tensor or an
    object with ``.sample`` (NCHW, roughly [-1, 1]) works; it runs wherever that object lives - the HIP decoder is bypassed,
    the image epilogue (:432-438 and numpy_to_pil's rounding) is done here."""

    def __init__(self, vae):
        self.inner = vae
        cfg = getattr(vae, "config", None)
        get = (lambda k, d: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d: getattr(cfg, k, d))
        self.config = SimpleNamespace(block_out_channels=tuple(get("block_out_channels", (128, 256, 512, 512))),
                                      scaling_factor=float(get("scaling_factor", 0.18215)))

    def decode(self, latents: torch.Tensor, want_float: bool = False):
        z = latents.permute(0, 3, 1, 2) / self.config.scaling_factor                           # NHWC fp32 -> NCHW, :432
        p = next(iter(self.inner.parameters()), None) if hasattr(self.inner, "parameters") else None
        if p is not None:
            z = z.to(p.device, p.dtype)
        out = self.inner.decode(z)
        img = out.sample if hasattr(out, "sample") else (out[0] if isinstance(out, (tuple, list)) else out)
        f32 = (img.float() / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).contiguous()             # :435-438
        u8 = (f32 * 255).round().to(torch.uint8)
        return u8.to(latents.device), (f32.to(latents.device) if want_float else None)


class StableDiffusionWalkPipeline:
    _optional_components = ["safety_checker", "feature_extractor"]

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, safety_checker=None, feature_extractor=None,
                 requires_safety_checker: bool = False, text_config=None):
        if safety_checker is not None and feature_extractor is None:
            raise ValueError("Make sure to define a feature extractor when loading StableDiffusionWalkPipeline if you "
                             "want to use the safety checker. If you do not want to use the safety checker, you can "
                             "pass `'safety_checker=None'` instead.")
        # a diffusers scheduler object of a known class (DDIM / PNDM / LMS / Euler / EulerA / DPM-Solver++) is rebuilt as the
        # table-driven scheduler of the same name from its config; anything else raises here, not in the middle of a walk
        scheduler = adopt_scheduler(scheduler)
        # the two config patches the reference applies to the scheduler (:85-110)
        if getattr(scheduler.config, "steps_offset", 1) != 1:
            scheduler.config.steps_offset = 1
        if getattr(scheduler.config, "clip_sample", False) is True:
            scheduler.config.clip_sample = False
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        self.safety_checker, self.feature_extractor = safety_checker, feature_extractor
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)      # :158
        self.tiled = False
        self.upsampler = None
        self.noise_device = "cpu"          # portable seeds
        self.embed_interp = "lerp"         # reference torch path lerps embeddings (:467); "slerp" = flax behaviour
        self.use_graphs = os.environ.get("SDV_NO_GRAPH", "0") != "1"
        self.cfg_shared_prefix = os.environ.get("SDV_NO_CFG_SHARED", "0") != "1"   # see UNetEngine.forward
        self._device = torch.device("cpu")
        self._graphs: Dict[tuple, dict] = {}
        self._graph_pool = None            # the private memory pool every captured step allocates from (see _capture)
        self.max_cached_graphs = 4         # LRU bound on captured denoise-step graphs (all share ONE private pool, which only shrinks when every graph is dropped)
        self._uncond_cache: Dict[str, torch.Tensor] = {}
        self._sched_cache: Dict[tuple, tuple] = {}
        self._writer: Optional[FrameWriter] = None
        self.last_timings: Dict[str, float] = {}

    # ------------------------------------------------------------------------------------------
    # construction
    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, *args, tiled: bool = False, torch_dtype=None,
                        revision=None, safety_checker=None, feature_extractor=None, vae=None, scheduler=None,
                        device=None, arch: Optional[str] = None, synthetic_seed: int = 0, fp8: bool = False, **kwargs):
        """Build the pipeline (reference :840-858).

        ``pretrained_model_name_or_path`` may be a LOCAL diffusers-layout directory (``unet/``, ``vae/``,
        ``text_encoder/``, ``tokenizer/``); there is no network, so hub ids resolve to ``$SDV_MODEL_DIR`` when
        that is set and otherwise to the same ARCHITECTURE with seeded synthetic weights (``arch`` in
        {"sd14", "sd21", "tiny"}; inferred from the name).  ``tiled=True`` makes every convolution circular
        (the reference monkey-patches nn.Conv2d; here it is a kernel flag).  ``fp8=True`` (or
        ``torch_dtype="fp8"``): the UNet's ResBlock convolutions run on the fp8 (e4m3) MFMA path with per-tensor scales
        calibrated on the first forward; everything else stays bf16 (BASELINE.json configs[4])."""
        name = str(pretrained_model_name_or_path or "")
        model_dir = None
        if name and Path(name).is_dir():
            model_dir = Path(name)
        elif os.environ.get("SDV_MODEL_DIR") and Path(os.environ["SDV_MODEL_DIR"]).is_dir():
            model_dir = Path(os.environ["SDV_MODEL_DIR"])
            logger.warning("from_pretrained(%r): not a local directory - loading $SDV_MODEL_DIR=%s instead", name, model_dir)
        synthetic = bool(kwargs.pop("synthetic", False)) or arch is not None or name.lower() == "tiny"
        if model_dir is None and not synthetic:
            # a hub id or a mistyped path must not silently turn into noise frames from random weights
            raise FileNotFoundError(
                f"from_pretrained({name!r}): no such local diffusers-layout directory (there is no network here, hub ids "
                "cannot be downloaded).  Point it at a model directory / set $SDV_MODEL_DIR, or ask explicitly for the "
                "architecture with seeded SYNTHETIC weights: arch='sd14' | 'sd21' | 'tiny' (or synthetic=True).")
        if arch is None:
            low = name.lower()
            arch = "tiny" if "tiny" in low else ("sd21" if ("stable-diffusion-2" in low or "sd21" in low) else "sd14")
        if model_dir is None:
            logger.warning("from_pretrained(%r): building the %s architecture with seeded SYNTHETIC weights and the hash "
                           "tokenizer (no checkpoint on disk) - outputs are for benchmarking / parity only", name, arch)
        if torch_dtype not in (None, torch.bfloat16, torch.float16, torch.float32, "fp8"):
            raise ValueError(f"unsupported torch_dtype {torch_dtype}")
        if torch_dtype in (torch.float16, torch.float32):
            # the reference passes torch_dtype through to diffusers (its own tests run float16, tests/test_pipeline.py:19-27);
            # the HIP engines have ONE storage format - bf16 activations / weights, fp32 accumulation, fp32 latents - so the
            # request cannot be honoured and must not be swallowed silently
            import warnings
            warnings.warn(f"from_pretrained(torch_dtype={torch_dtype}): the MI355X HIP engines store activations and weights as "
                          "bfloat16 (fp32 accumulation, fp32 latents and scheduler state); this pipeline runs in bfloat16, not "
                          f"{torch_dtype} - pipe.torch_dtype reports what actually runs", UserWarning, stacklevel=2)
        if model_dir is not None:
            ucfg = cfgs.unet_from_json(model_dir / "unet" / "config.json")
            vcfg = cfgs.vae_from_json(model_dir / "vae" / "config.json")
            tcfg = {1024: cfgs.sd21_text(), 768: cfgs.sd14_text()}.get(ucfg.cross_attention_dim) or cfgs.TextConfig(
                hidden_size=ucfg.cross_attention_dim, intermediate_size=4 * ucfg.cross_attention_dim, num_hidden_layers=2,
                num_attention_heads=max(1, ucfg.cross_attention_dim // 64))
            u_sd = weights.load_component(model_dir, "unet", weights.unet_shapes(ucfg))
            v_sd = weights.load_component(model_dir, "vae", weights.vae_decoder_shapes(vcfg))
        else:
            ucfg = {"sd14": cfgs.sd14_unet, "sd21": cfgs.sd21_unet, "tiny": cfgs.tiny_unet}[arch]()
            vcfg = {"sd14": cfgs.sd_vae, "sd21": cfgs.sd_vae, "tiny": cfgs.tiny_vae}[arch]()
            tcfg = {"sd14": cfgs.sd14_text, "sd21": cfgs.sd21_text, "tiny": cfgs.tiny_text}[arch]()
            rank, ws = parallel.world()
            if ws == 1 or rank == 0:
                u_sd = weights.synthetic_state_dict(weights.unet_shapes(ucfg), seed=synthetic_seed)
                v_sd = weights.synthetic_state_dict(weights.vae_decoder_shapes(vcfg), seed=synthetic_seed + 1)
            else:
                u_sd = v_sd = None       # arrives through the RCCL weight broadcast in .to(device)
        if scheduler is None:
            ptype = "v_prediction" if (arch == "sd21" and model_dir is None) else "epsilon"
            sched_cls = DDIMScheduler          # synthetic pipelines: the scheduler BASELINE.json names
            sched_kw = {}
            if model_dir is not None and (model_dir / "scheduler" / "scheduler_config.json").exists():
                # a checkpoint directory decides its own scheduler, as diffusers' from_pretrained does (SD-v1: PNDMScheduler)
                from .scheduler import SCHEDULERS, kwargs_from_config
                sc = json.loads((model_dir / "scheduler" / "scheduler_config.json").read_text())
                cname = sc.get("_class_name", "DDIMScheduler")
                if cname not in SCHEDULERS:
                    raise NotImplementedError(f"{model_dir}/scheduler: {cname} is not one of {sorted(SCHEDULERS)}")
                sched_cls = SCHEDULERS[cname]
                # every key the class acts on (set_alpha_to_one, skip_prk_steps, solver_order ...), unsupported options raise
                sched_kw = kwargs_from_config(cname, sc)
                ptype = sched_kw.pop("prediction_type", "epsilon")
            scheduler = sched_cls(prediction_type=ptype, **sched_kw)
        text_encoder = build_text_encoder(tcfg, model_dir, seed=synthetic_seed + 2)
        tokenizer = load_tokenizer(model_dir, tcfg)
        if vae is not None and not isinstance(vae, (_PendingModule, VAEDecoderEngine)):
            vae = _ForeignVAE(vae)
        pipe = cls(vae=vae or _PendingModule("vae", vcfg, v_sd), text_encoder=text_encoder, tokenizer=tokenizer,
                   unet=_PendingModule("unet", ucfg, u_sd), scheduler=scheduler, safety_checker=safety_checker,
                   feature_extractor=feature_extractor, requires_safety_checker=False, text_config=tcfg)
        pipe.tiled = tiled
        pipe.fp8 = bool(fp8) or torch_dtype == "fp8"    # BASELINE config 5: e4m3 operands in the UNet's ResBlock convs
        pipe.synthetic = model_dir is None
        pipe.torch_dtype = torch.bfloat16            # what the engines compute in (see the warning above)
        pipe.requested_torch_dtype = torch_dtype
        if device is not None:
            pipe.to(device)
        return pipe

    def to(self, device):
        """Move to the GPU: (optionally RCCL-broadcast and) re-layout the weights into HBM and build the
        HIP engines.  Only CUDA/HIP devices are accepted for the networks - there is no CPU path."""
        device = torch.device(device)
        if device.type != "cuda":
            if isinstance(self.unet, _PendingModule):
                self.text_encoder.to(device)
                self._device = device
                return self
            raise hip.SdvHipError("the HIP engines cannot be moved off the GPU")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(device)      # hip._stream() is the current stream of the CURRENT device
        hip.load()
        if isinstance(self.unet, _PendingModule):
            u, v = self.unet, self.vae
            u_shapes = weights.unet_shapes(u.config)
            u_sd = parallel.broadcast_state_dict(u.state_dict, u_shapes, device)
            self.unet = UNetEngine(u.config, u_sd, device, tiled=self.tiled, fp8=getattr(self, "fp8", False))
            del u_sd
            u.state_dict = None
            if isinstance(v, _PendingModule):
                v_shapes = weights.vae_decoder_shapes(v.config)
                v_sd = parallel.broadcast_state_dict(v.state_dict, v_shapes, device)
                self.vae = VAEDecoderEngine(v.config, v_sd, device, tiled=self.tiled)
                v.state_dict = None
        self.text_encoder.to(device)
        self._device = device
        self._uncond_cache.clear()
        return self

    @property
    def device(self) -> torch.device:
        return self._device

    def enable_attention_slicing(self, slice_size: Optional[Union[str, int]] = "auto"):
        """Accepted for API compatibility (reference :161-181).  The HIP attention kernel is flash-style and
        never materialises the score matrix, so there is nothing to slice."""
        self._attention_slice = slice_size

    def disable_attention_slicing(self):
        self.enable_attention_slicing(None)

    def progress_bar(self, iterable):
        return iterable

    @staticmethod
    def numpy_to_pil(images):
        return numpy_to_pil(images)

    # ------------------------------------------------------------------------------------------
    # text / noise endpoints
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def embed_text(self, text, negative_prompt=None):
        """Helper to embed some text (reference :809-820) -> fp32 (1, 77, D) on the pipeline device."""
        text_input = self.tokenizer(text, padding="max_length", max_length=self.tokenizer.model_max_length,
                                    truncation=True, return_tensors="pt")
        return self.text_encoder(text_input.input_ids.to(self.device))[0]

    def init_noise(self, seed, noise_shape, dtype=torch.float32):
        """Helper to initialize noise (reference :822-838)."""
        if self.noise_device == "cpu" or self.device.type != "cuda":
            g = torch.Generator(device="cpu").manual_seed(int(seed))
            return torch.randn(noise_shape, generator=g, dtype=dtype, device="cpu").to(self.device)
        g = torch.Generator(device=self.device).manual_seed(int(seed))
        return torch.randn(noise_shape, generator=g, dtype=dtype, device=self.device)

    def _uncond_embeddings(self, negative_prompt, batch_size: int) -> torch.Tensor:
        if negative_prompt is None:
            toks = [""]
        elif isinstance(negative_prompt, str):
            toks = [negative_prompt]
        elif batch_size != len(negative_prompt):
            raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                             f" has batch size {batch_size}. Please make sure that passed `negative_prompt` matches"
                             " the batch size of `prompt`.")
        else:
            toks = list(negative_prompt)
        key = "\x00".join(toks)
        if key not in self._uncond_cache:
            ids = self.tokenizer(toks, padding="max_length", max_length=self.tokenizer.model_max_length,
                                 truncation=True, return_tensors="pt").input_ids
            with torch.no_grad():
                self._uncond_cache[key] = self.text_encoder(ids.to(self.device))[0].float()
        u = self._uncond_cache[key]
        return u.repeat(batch_size, 1, 1) if u.shape[0] == 1 else u

    # ------------------------------------------------------------------------------------------
    # the denoise + decode call
    # ------------------------------------------------------------------------------------------
    def _schedule(self, num_inference_steps: int, eta: float):
        """set_timesteps (:394) + the device tables of this schedule.  DDIM (the scheduler BASELINE.json names): {c_x, c_e,
        sigma} per step for ``sdv_cfg_ddim_step``; every other scheduler of the reference's signature (:71-78): the 16-float
        rows of ``sdv_cfg_multistep_step`` (scheduler.py).  Returns (key, table, evaluations) - PLMS runs one UNet
        evaluation more than ``num_inference_steps``."""
        self.scheduler = adopt_scheduler(self.scheduler)      # (a scheduler assigned after construction)
        ddim = hasattr(self.scheduler, "coefficient_table")
        self.scheduler.set_timesteps(num_inference_steps)
        ts = tuple(float(t) for t in self.scheduler.timesteps)
        if not ddim:
            eta = 0.0        # "eta is only used with the DDIMScheduler, it will be ignored for others" (:236, :404-409)
        # (the whole config: two objects of one class with different betas / set_alpha_to_one / solver_order must not share a
        #  coefficient table - and the captured graphs that bake in its pointer)
        key = (type(self.scheduler).__name__, ts, float(eta), repr(sorted(vars(self.scheduler.config).items())))
        if key not in self._sched_cache:
            while len(self._sched_cache) >= 8:
                gone = next(iter(self._sched_cache))
                self._sched_cache.pop(gone)
                # captured steps bake in the pointers of that schedule's coefficient / time-embedding tables
                for gk in [k for k in self._graphs if k[0] == gone]:
                    dead = self._graphs.pop(gk)
                    dead["graph"] = dead["one_step"] = None
            coefs = (self.scheduler.coefficient_table(eta) if ddim else self.scheduler.fused_table()).to(self.device)
            self.unet.prepare_timesteps(ts)
            tables = [r.bias_table for r in self.unet.res]
            self._sched_cache[key] = (coefs, tables)
        coefs, tables = self._sched_cache[key]
        for r, t in zip(self.unet.res, tables):
            r.bias_table = t
        return key, coefs, len(ts)

    def enable_fp8_saturation_check(self, on: bool = True):
        """Debug aid of the fp8 path: count the activations the e4m3 conversion clamps at +-448 (a calibration that is too tight
        for the prompts actually run).  Call before the first frame is generated - captured step graphs bake the counter's
        address in; ``fp8_saturated()`` reads it."""
        if on:
            self._fp8_sat = torch.zeros(1, dtype=torch.int32, device=self.device)
            hip.set_fp8_saturation_counter(self._fp8_sat)
        else:
            hip.set_fp8_saturation_counter(None)
            self._fp8_sat = None
        # steps captured before the switch carry the OLD counter address (or none): replayed, they would count into a freed tensor
        # or read as "nothing clamped" - both directions drop them
        self._drop_graphs()

    def _drop_graphs(self):
        """Forget every captured step the way the LRU eviction does: the graph object goes, and so do the per-batch-size
        cross-attention buffers it held raw pointers into."""
        sizes = {k[1] for k in self._graphs}
        for ent in self._graphs.values():
            ent["graph"] = ent["one_step"] = None        # (the closure holds the entry: break the cycle, the buffers go now)
        self._graphs.clear()
        for nimg in sizes:
            self.unet.release(nimg)

    def fp8_saturated(self) -> int:
        """Clamped e4m3 conversions since ``enable_fp8_saturation_check()`` (host synchronising)."""
        sat = getattr(self, "_fp8_sat", None)
        if sat is None:
            raise hip.SdvHipError("fp8_saturated(): call enable_fp8_saturation_check() first")
        return int(sat.item())

    def _calibrate_fp8(self, h: int, w: int, coefs, nsteps: int, guidance: float, cfg: bool, cond: Optional[torch.Tensor] = None):
        """fp8 mode: fix the e4m3 activation scales of the UNet's ResBlock convs with ONE pilot denoise run, eager, before any
        graph is warmed up or captured: one frame of seeded N(0,1) latents (seed 0, CPU generator), all ``nsteps`` steps,
        scales = 2 x the running amax over BOTH halves of a real guidance pair - the unconditional half under the empty prompt,
        the conditional half under ``cond`` ([1, L, D]: the embedding of the walk's FIRST prompt, ``walk()`` sets
        ``_fp8_pilot_cond``; a bare ``__call__`` passes its own first row).  With trained weights the conditional branch is
        where the large activations are; rounds 2-3 calibrated under the unconditional context only.  The pilot
        does not depend on the batch, the rank or where a resumed walk starts - every rank embeds the same first prompt - so
        every rank and every re-run quantises identically (the scales used to come from the graph warm-up's
        zero-filled buffers).  ``enable_fp8_saturation_check()`` counts what the scales still clip afterwards."""
        C = self.unet.cfg.in_channels
        g = torch.Generator(device="cpu").manual_seed(0)
        lat = torch.randn((1, h, w, C), generator=g, dtype=F32).to(self.device) * self.scheduler.init_noise_sigma
        uncond = self._uncond_embeddings(None, 1)
        nimg = 2 if cfg else 1
        if cond is not None and tuple(cond.shape[1:]) == tuple(uncond.shape[1:]):
            cond = cond[:1].to(self.device, F32)
            pilot_ctx = torch.cat([uncond, cond]) if cfg else cond
        else:
            pilot_ctx = torch.cat([uncond] * nimg)
        self.unet.prepare_context(pilot_ctx)
        self.unet.reserve(nimg, h, w)
        x2 = torch.zeros((nimg * h * w, C), dtype=BF16, device=self.device)
        step = torch.zeros(1, dtype=torch.int32, device=self.device)
        s0 = self.scheduler.first_input_scale() if hasattr(self.scheduler, "first_input_scale") else 1.0
        hip.latents_to_unet_input(lat if s0 == 1.0 else lat * s0, x2, cfg, lat.numel())
        multistep = coefs.shape[1] == 16
        hist = torch.zeros((4,) + tuple(lat.shape), dtype=F32, device=self.device) if multistep else None
        xsave = torch.zeros_like(lat) if multistep else None
        before = self.unet.fp8_scales()
        self.unet.fp8_calibration(True)
        try:
            for _ in range(nsteps):
                eps = self.unet.forward(x2, nimg, h, w, step, cfg_shared=cfg and self.cfg_shared_prefix)
                if multistep:      # (a stochastic scheduler's noise term is left out of the pilot: it only widens the scales)
                    hip.cfg_multistep_step(eps, lat, x2, hist, xsave, coefs, step, None, guidance, cfg, lat.numel())
                else:
                    hip.cfg_ddim_step(eps, lat, x2, coefs, step, None, guidance, cfg, lat.numel())
                hip.step_counter_add(step, 1)
            torch.cuda.synchronize(self.device)
        except BaseException:
            # a pilot that died part-way (out of memory, a bad shape) must not leave half-widened scales behind and must not
            # mark the engine calibrated: the next call runs the whole pilot again
            self.unet.fp8_calibration(False, ok=False, restore=before)
            raise
        self.unet.fp8_calibration(False)
        # the pilot's 2-sample cross-attention K / V^T buffers and workspaces are only kept when a 2-sample step is what runs next
        if not any(k[1] == nimg for k in self._graphs):
            self.unet.release(nimg)

    def _graph_entry(self, key: tuple, nimg: int, B: int, h: int, w: int, cfg: bool, guidance: float, coefs, eta_noise):
        """Static buffers + a captured hipGraph of ONE denoise step (UNet forward + CFG/DDIM update + step++)."""
        if key in self._graphs:
            ent = self._graphs[key] = self._graphs.pop(key)    # most recently used last
            # an entry made while use_graphs was off (or whose capture failed) is captured behind its next eager step once graphs
            # are on again - otherwise it would run eagerly for ever
            if self.use_graphs and ent["graph"] is None and not ent.get("capture_failed"):
                ent["capture_pending"] = True
            return ent
        while len(self._graphs) >= self.max_cached_graphs:     # bound the static buffers + cross-attention K / V^T sets kept alive
            old_key = next(iter(self._graphs))
            ent = self._graphs.pop(old_key)
            ent["graph"] = ent["one_step"] = None
            del ent
            # the cross-attention K / V^T buffers of the text context are kept per batch size because captured graphs hold raw
            # pointers into them: once no cached graph runs at that batch size any more they go too (a resumed walk with many
            # short runs would otherwise leave one set per distinct batch size behind)
            if not any(k[1] == old_key[1] for k in self._graphs) and old_key[1] != nimg:
                self.unet.release(old_key[1])
        dev = self.device
        C = self.unet.cfg.in_channels
        ent = {
            "latents": torch.zeros((B, h, w, C), dtype=F32, device=dev),
            "x2": torch.zeros((nimg * h * w, C), dtype=BF16, device=dev),
            "step": torch.zeros(1, dtype=torch.int32, device=dev),
            "graph": None,
        }
        self.unet.reserve(nimg, h, w)
        n = B * h * w * C
        multistep = coefs.shape[1] == 16           # sdv_cfg_multistep_step rows (every scheduler but DDIM)
        if multistep:
            ent["hist"] = torch.zeros((4, B, h, w, C), dtype=F32, device=dev)      # ring of earlier model outputs
            ent["xsave"] = torch.zeros((B, h, w, C), dtype=F32, device=dev)        # PLMS: the sample of the first evaluation

        def one_step():
            eps = self.unet.forward(ent["x2"], nimg, h, w, ent["step"], cfg_shared=cfg and self.cfg_shared_prefix)
            if multistep:
                hip.cfg_multistep_step(eps, ent["latents"], ent["x2"], ent["hist"], ent["xsave"], coefs, ent["step"], eta_noise,
                                       guidance, cfg, n)
            else:
                hip.cfg_ddim_step(eps, ent["latents"], ent["x2"], coefs, ent["step"], eta_noise, guidance, cfg, n)
            hip.step_counter_add(ent["step"], 1)

        ent["one_step"] = one_step
        # The step is captured by ``_capture`` right AFTER its first eager execution - which is the first REAL denoise step of the
        # first call at this key (lazy allocations / attribute sets happen there), not a throw-away warm-up on zero-filled buffers.
        ent["capture_pending"] = bool(self.use_graphs)
        self._graphs[key] = ent
        return ent

    def _capture(self, ent: dict):
        """Capture ``ent``'s denoise step into a hipGraph (nothing executes: the static buffers keep the state the eager step
        left).  Not through the ``torch.cuda.graph`` context manager: its ``__enter__`` runs ``torch.cuda.empty_cache()``, which
        hands every cached block of the allocator back to the driver - 0.9 s for a 16-frame capture that followed a 60-frame call,
        1.7 s of the 2.15 s "cold start" the round-4 bench line showed (profiles/round5_cold_start_probe_before.txt); the capture
        itself is 14 ms.  All captured steps share ONE private memory pool: their intermediates are dead when a replay ends and
        two steps never run concurrently, so a new batch size costs the pool only what it needs beyond the largest so far."""
        t0 = time.perf_counter()
        dev = self.device
        # The shared pool lives as long as one graph captured into it does: once every captured step has been dropped (fp8 counter
        # switch, LRU eviction of the last one, a caller clearing ``_graphs``) PyTorch forgets the pool, and capturing into the stale
        # handle trips an internal assert of its caching allocator (seen when the collector had already freed the old graphs).
        if self._graph_pool is None or not any(e.get("graph") is not None for e in self._graphs.values()):
            self._graph_pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        # with a process group alive its watchdog thread may touch the runtime while this thread captures: only this
        # thread's calls (kernel launches through the C ABI) need to be capture-safe
        mode = "thread_local" if parallel.world()[1] > 1 else "global"
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        # No garbage collection while the stream captures: an unreachable step graph of an earlier call (its entry holds a closure
        # that holds the entry - a cycle only the collector frees) would be destroyed in the middle of the capture, and destroying
        # a graph / returning its blocks is not a capturable call (seen as a crash inside the first captured launch when this ran
        # late in a long test session; ``torch.cuda.graph`` runs a full ``gc.collect()`` up front for the same reason - switching
        # the collector off for the ~10 ms of the capture costs nothing).
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()
        ent["capture_pending"] = False          # whatever happens below, the capture is not retried on every following step
        err = None
        try:
            with torch.cuda.stream(side):
                g.capture_begin(pool=self._graph_pool, capture_error_mode=mode)
                try:
                    ent["one_step"]()
                except BaseException as e:      # keep the ORIGINAL error: capture_end() on a broken capture raises one of its own
                    err = e
                try:
                    g.capture_end()
                except BaseException as e:
                    if err is None:
                        err = e
        finally:
            if gc_was_on:
                gc.enable()
            torch.cuda.current_stream().wait_stream(side)
        if err is not None:
            ent["capture_failed"] = True        # this entry keeps running eagerly
            raise err
        ent["graph"] = g
        self.last_graph_build = {"capture_s": time.perf_counter() - t0}

    @torch.no_grad()
    def __call__(self, prompt: Optional[Union[str, List[str]]] = None, height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 negative_prompt: Optional[Union[str, List[str]]] = None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "pil", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: Optional[int] = 1,
                 text_embeddings: Optional[torch.Tensor] = None, **kwargs):
        """Same contract as the reference ``__call__`` (:191-455)."""
        if isinstance(self.unet, _PendingModule):
            raise hip.SdvHipError("call pipeline.to('cuda') first: the hot path only exists as HIP kernels")
        t_start = time.perf_counter()
        height = height or self.unet.config.sample_size * self.vae_scale_factor                 # :268
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        if height % 8 != 0 or width % 8 != 0:                                                      # :271
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (not isinstance(callback_steps, int) or callback_steps <= 0):  # :274
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type"
                             f" {type(callback_steps)}.")
        prompt_given = text_embeddings is None
        if text_embeddings is None:
            if isinstance(prompt, str):
                batch_size = 1
            elif isinstance(prompt, list):
                batch_size = len(prompt)
            else:
                raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")    # :288
            text_inputs = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length,
                                         return_tensors="pt")
            ids = text_inputs.input_ids
            if ids.shape[-1] > self.tokenizer.model_max_length:                                  # :299
                removed = self.tokenizer.batch_decode(ids[:, self.tokenizer.model_max_length:])
                print("The following part of your input was truncated because CLIP can only handle sequences up to"
                      f" {self.tokenizer.model_max_length} tokens: {removed}")
                ids = ids[:, : self.tokenizer.model_max_length]
            text_embeddings = self.text_encoder(ids.to(self.device))[0]
        else:
            batch_size = text_embeddings.shape[0]                                                 # :308
        text_embeddings = text_embeddings.to(self.device, F32)
        bs_embed, seq_len, _ = text_embeddings.shape
        text_embeddings = text_embeddings.repeat(1, num_images_per_prompt, 1).view(bs_embed * num_images_per_prompt,
                                                                                   seq_len, -1)
        do_cfg = guidance_scale > 1.0                                                             # :318
        if do_cfg:
            if negative_prompt is not None and prompt_given and type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")                                             # :325
            uncond = self._uncond_embeddings(negative_prompt, batch_size)
            if num_images_per_prompt != 1:
                uncond = uncond.repeat(1, num_images_per_prompt, 1).view(batch_size * num_images_per_prompt, seq_len, -1)
            ctx = torch.cat([uncond, text_embeddings])                                            # :358
        else:
            ctx = text_embeddings
        B = batch_size * num_images_per_prompt
        C = self.unet.in_channels
        h, w = height // 8, width // 8
        latents_shape = (B, C, h, w)                                                              # :365-370
        if latents is None:
            gdev = generator.device if generator is not None else torch.device("cpu")
            latents = torch.randn(latents_shape, generator=generator, device=gdev, dtype=F32)
        elif tuple(latents.shape) != latents_shape:
            raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {latents_shape}")   # :390
        latents = latents.to(self.device, F32)

        sched_key, coefs, nsteps = self._schedule(num_inference_steps, eta)                       # :394
        if not hasattr(self.scheduler, "coefficient_table"):
            eta = 0.0                                                                             # :404-409: DDIM only
        if getattr(self.unet, "fp8", False) and not self.unet.fp8_calibrated:
            pilot_cond = getattr(self, "_fp8_pilot_cond", None)
            self._calibrate_fp8(h, w, coefs, nsteps, float(guidance_scale), do_cfg,
                                cond=pilot_cond if pilot_cond is not None else text_embeddings[:1])
        # A ragged last batch (B frames where a graph for B' > B frames is already captured) is padded with copies of its
        # last frame and replays the big graph: a second capture would own a second multi-GB private pool for one call.
        # Only while the padding is at most a quarter of the big batch - 60 frames replayed as 128 would pay for 128.
        B_real = B
        if self.use_graphs and eta == 0 and callback is None and not getattr(self.scheduler, "stochastic", False):
            tail = (h, w, do_cfg, float(guidance_scale), False, int(ctx.shape[1]), self.cfg_shared_prefix, self.tiled)
            mult = 2 if do_cfg else 1
            bigger = [k[1] // mult for k in self._graphs if k[0] == sched_key and k[2:] == tail and k[1] // mult > B]
            if bigger and 4 * (min(bigger) - B) <= min(bigger) and \
                    (B * mult, ) + tail not in {k[1:] for k in self._graphs if k[0] == sched_key}:
                pad = min(bigger) - B
                latents = torch.cat([latents, latents[-1:].expand(pad, -1, -1, -1)])
                if do_cfg:
                    u, c = ctx[:B], ctx[B:]
                    ctx = torch.cat([u, u[-1:].expand(pad, -1, -1), c, c[-1:].expand(pad, -1, -1)])
                else:
                    ctx = torch.cat([ctx, ctx[-1:].expand(pad, -1, -1)])
                B += pad
        nimg = 2 * B if do_cfg else B
        eta_noise = None
        stochastic = (eta > 0 and hasattr(self.scheduler, "coefficient_table")) or getattr(self.scheduler, "stochastic", False)
        if stochastic:
            gdev = generator.device if generator is not None else torch.device("cpu")
            eta_noise = torch.randn((nsteps, B, h, w, C), generator=generator, device=gdev, dtype=F32).to(self.device)
        self.unet.prepare_context(ctx)
        # everything a captured step bakes in: pointers of the (nimg, Lc) cross-attention K/V buffers, the shared-prefix
        # structure, padding mode, guidance scale, schedule
        gkey = (sched_key, nimg, h, w, do_cfg, float(guidance_scale), stochastic, int(ctx.shape[1]), self.cfg_shared_prefix,
                self.tiled)
        ent = self._graph_entry(gkey, nimg, B, h, w, do_cfg, float(guidance_scale), coefs, eta_noise)
        if stochastic:
            ent.setdefault("noise", eta_noise)
            if ent["noise"] is not eta_noise:
                ent["noise"].copy_(eta_noise)

        lat_nhwc = hip.nchw_to_nhwc(latents * self.scheduler.init_noise_sigma)                    # :401
        ent["latents"].copy_(lat_nhwc)
        ent["step"].zero_()
        s0 = self.scheduler.first_input_scale() if hasattr(self.scheduler, "first_input_scale") else 1.0
        # :414-415: torch.cat([latents] * 2) -> scheduler.scale_model_input (sigma-space schedulers: x / sqrt(sigma_0^2 + 1))
        hip.latents_to_unet_input(ent["latents"] if s0 == 1.0 else ent["latents"] * s0, ent["x2"], do_cfg, ent["latents"].numel())
        t_prep = time.perf_counter()
        for i in range(nsteps):                                                                   # :412
            if ent["graph"] is not None:
                ent["graph"].replay()
            else:
                ent["one_step"]()
                if ent.get("capture_pending"):
                    self._capture(ent)
            if callback is not None and i % callback_steps == 0:                                  # :429
                callback(i, self.scheduler.timesteps[i].item(), hip.nhwc_to_nchw(ent["latents"]))
        if kwargs.get("return_latents", False):
            return hip.nhwc_to_nchw(ent["latents"])[:B_real]
        # "numpy_u8": rounded uint8 NHWC array, no PIL objects; "u8_cuda": the same array left in HBM (upsampler input)
        want_float = output_type not in ("pil", "numpy_u8", "u8_cuda")
        u8, f32 = self.vae.decode(ent["latents"], want_float=want_float)                         # :432-435
        if output_type == "u8_cuda":
            image = u8[:B_real]
        elif want_float:
            image = f32[:B_real].cpu().numpy()                                                    # :438
        else:
            image = u8[:B_real].cpu().numpy()
        t_done = time.perf_counter()
        self.last_timings = {"prepare_s": t_prep - t_start, "denoise_decode_s": t_done - t_prep, "frames": B_real}
        has_nsfw = None
        if self.safety_checker is not None:
            raise NotImplementedError("safety_checker is outside the hot path; pass safety_checker=None")
        if output_type == "pil":
            image = numpy_to_pil(image)                                                           # :450
        if not return_dict:
            return (image, has_nsfw)
        return StableDiffusionPipelineOutput(images=image, nsfw_content_detected=has_nsfw)

    # ------------------------------------------------------------------------------------------
    # interpolation + clip generation
    # ------------------------------------------------------------------------------------------
    def generate_inputs(self, prompt_a, prompt_b, seed_a, seed_b, noise_shape, T, batch_size):
        """Reference :457-479.  Yields ``(batch_idx, embeds (b,77,D), noise (b,C,h,w))``; every batch is ONE
        lerp launch + ONE slerp launch over all its frames (the whole-tensor dot/norms of utils.py:51 are
        per-clip constants and are reduced once)."""
        embeds_a = self.embed_text(prompt_a).float().contiguous()
        embeds_b = self.embed_text(prompt_b).float().contiguous()
        latents_a = self.init_noise(seed_a, noise_shape, F32).contiguous()
        latents_b = self.init_noise(seed_b, noise_shape, F32).contiguous()
        T = np.asarray(T)
        if embeds_a.is_cuda:
            stats = hip.slerp_stats(latents_a, latents_b)
            e_stats = hip.slerp_stats(embeds_a, embeds_b) if self.embed_interp == "slerp" else None
        batch_idx = 0
        _, Cc, hh, ww = noise_shape
        for s in range(0, T.shape[0], batch_size):
            tb = torch.tensor(np.asarray(T[s:s + batch_size], dtype=np.float64), dtype=F32, device=self.device)
            n = tb.numel()
            embeds = torch.empty((n,) + tuple(embeds_a.shape[1:]), dtype=F32, device=self.device)
            if self.embed_interp == "slerp":
                hip.slerp_batch(embeds_a, embeds_b, e_stats, tb, C_=1, HW=embeds_a.numel(), to_hwc=False, out=embeds)
            else:
                hip.lerp_batch(embeds_a, embeds_b, tb, out_f32=embeds)                          # :467
            noise = hip.slerp_batch(latents_a, latents_b, stats, tb, C_=1, HW=latents_a.numel(), to_hwc=False)  # :468
            yield batch_idx, embeds, noise.view(n, Cc, hh, ww)
            batch_idx += 1

    def make_clip_frames(self, prompt_a: str, prompt_b: str, seed_a: int, seed_b: int, num_interpolation_steps: int = 5,
                         save_path: Union[str, Path] = "outputs/", num_inference_steps: int = 50,
                         guidance_scale: float = 7.5, eta: float = 0.0, height: Optional[int] = None,
                         width: Optional[int] = None, upsample: bool = False, batch_size: int = 1,
                         image_file_ext: str = ".png", T: np.ndarray = None, skip: int = 0, negative_prompt: str = None,
                         step: Optional[Tuple[int, int]] = None, stop: Optional[int] = None,
                         frame_indices: Optional[List[int]] = None):
        """Reference :481-554.  Two extensions for frame sharding / hole-filling resume: ``stop`` (frames [skip, stop) of the
        clip are generated and written under their own indices) and ``frame_indices`` (an explicit, sorted list of the clip's
        frames to generate - the missing frames of a resumed clip are then batched TOGETHER, ``batch_size`` at a time, instead
        of one short batch per hole)."""
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        save_path = Path(save_path)
        save_path.mkdir(parents=True, exist_ok=True)
        T = T if T is not None else np.linspace(0.0, 1.0, num_interpolation_steps)             # :509
        if T.shape[0] != num_interpolation_steps:                                                # :510
            raise ValueError(f"Unexpected T shape, got {T.shape}, expected dim 0 to be {num_interpolation_steps}")
        if upsample:
            if getattr(self, "upsampler", None) is None:
                from .upsampling import RealESRGANModel
                self.upsampler = RealESRGANModel.from_pretrained("nateraw/real-esrgan")
            self.upsampler.to(self.device)
        stop = num_interpolation_steps if stop is None else stop
        indices = list(range(skip, stop)) if frame_indices is None else [int(k) for k in frame_indices]
        if any(k < 0 or k >= num_interpolation_steps for k in indices):
            raise ValueError(f"frame_indices outside [0, {num_interpolation_steps})")
        batch_generator = self.generate_inputs(prompt_a, prompt_b, seed_a, seed_b,
                                               (1, self.unet.in_channels, height // 8, width // 8), np.asarray(T)[indices],
                                               batch_size)
        num_batches = math.ceil(num_interpolation_steps / batch_size)
        log_prefix = "" if step is None else f"[{step[0]}/{step[1]}] "
        writer = self._writer or FrameWriter()
        pos = 0
        for batch_idx, embeds_batch, noise_batch in batch_generator:
            frame_index = indices[pos]
            if batch_size == 1:
                msg = f"Generating frame {frame_index}"
            else:
                msg = f"Generating frames {frame_index}-{indices[pos + embeds_batch.shape[0] - 1]}"
            logger.info(f"{log_prefix}[{batch_idx}/{num_batches}] {msg}")
            outputs = self(latents=noise_batch, text_embeddings=embeds_batch, height=height, width=width,
                           guidance_scale=guidance_scale, eta=eta, num_inference_steps=num_inference_steps,
                           output_type="pil" if not upsample else "u8_cuda", negative_prompt=negative_prompt)["images"]
            if upsample:
                # :552 - the reference upsamples frame by frame on the way to disk (float -> uint8 -> RealESRGANer); here the
                # uint8 frames never leave HBM before the x4 network has run on the whole batch
                outputs = numpy_to_pil(self.upsampler.upsample_u8(outputs).cpu().numpy())
            for image in outputs:
                frame_filepath = save_path / (f"frame%06d{image_file_ext}" % indices[pos])
                writer.submit(image, frame_filepath)                                             # :553
                pos += 1
        if self._writer is None:
            writer.close()

    # ``resume=True``: which frames of a clip are (re)generated.  "holes" (default) = every frame whose file is missing or empty;
    # "reference" = the reference's rule, bit for bit (:741-753): continue behind the LAST frame on disk, whatever lies before it,
    # and skip a clip whose last frame is num_step - 2 or later (its one-missing-frame quirk included) - only right for ONE sequential
    # writer, so it is meant for single-rank walks that must leave exactly the reference's files.  SDV_RESUME=reference selects it.
    resume_policy = os.environ.get("SDV_RESUME", "holes")

    @staticmethod
    def resume_todo(save_path, mp4_path, num_step: int, image_file_ext: str = ".png", policy: str = "holes"):
        """Frames of one clip a resumed walk has to generate (ascending list), or None when the clip is skipped."""
        if policy not in ("holes", "reference"):
            raise ValueError(f"resume policy must be 'holes' or 'reference', got {policy!r}")
        save_path = Path(save_path)
        if Path(mp4_path).exists():
            return None
        if policy == "reference":
            existing = sorted(save_path.glob(f"*{image_file_ext}"))
            if not existing:
                return list(range(num_step))
            skip = int(existing[-1].stem[-6:]) + 1
            return None if skip + 1 >= num_step else list(range(skip, num_step))
        have = set()
        for f in save_path.glob(f"frame*{image_file_ext}"):
            digits = f.stem[len("frame"):]
            if len(digits) == 6 and digits.isdigit() and f.stat().st_size > 0:
                have.add(int(digits))
        todo = [k for k in range(num_step) if k not in have]
        return todo or None

    def walk(self, prompts: Optional[List[str]] = None, seeds: Optional[List[int]] = None,
             num_interpolation_steps: Optional[Union[int, List[int]]] = 5, output_dir: Optional[str] = "./dreams",
             name: Optional[str] = None, image_file_ext: Optional[str] = ".png", fps: Optional[int] = 30,
             num_inference_steps: Optional[int] = 50, guidance_scale: Optional[float] = 7.5, eta: Optional[float] = 0.0,
             height: Optional[int] = None, width: Optional[int] = None, upsample: Optional[bool] = False,
             batch_size: Optional[int] = 1, resume: Optional[bool] = False, audio_filepath: str = None,
             audio_start_sec: Optional[Union[int, float]] = None, margin: Optional[float] = 1.0,
             smooth: Optional[float] = 0.0, negative_prompt: Optional[str] = None, make_video: Optional[bool] = True):
        """Generate the frames (and video) of a walk through ``prompts`` / ``seeds`` - reference :556-807.

        Output layout (identical to the reference docstring :648-666):
            {output_dir}/{name}/prompt_config.json
            {output_dir}/{name}/{name}_{i:06d}/frame{k:06d}{ext}   and   .../{name}_{i:06d}.mp4
            {output_dir}/{name}/{name}.mp4
        With torch.distributed initialised the frames are sharded over the ranks (``parallel.py``); rank 0
        writes the config and muxes the video.  Returns the video path, or None when ``make_video=False``."""
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        rank, world_size = parallel.world()
        output_path = Path(output_dir)
        # one name for all ranks: they reach this line seconds apart (weight broadcast, engine build), so a per-rank
        # timestamp would scatter the shards over several directories
        name = parallel.broadcast_object(name or time.strftime("%Y%m%d-%H%M%S"))
        save_path_root = output_path / name
        save_path_root.mkdir(parents=True, exist_ok=True)
        output_filepath = save_path_root / f"{name}.mp4"
        if not resume and isinstance(num_interpolation_steps, int):
            num_interpolation_steps = [num_interpolation_steps] * (len(prompts) - 1)            # :685-686
        if not resume:
            audio_start_sec = audio_start_sec or 0
        prompt_config_path = save_path_root / "prompt_config.json"
        if not resume:
            if rank == 0:
                prompt_config_path.write_text(json.dumps(dict(
                    prompts=prompts, seeds=seeds, num_interpolation_steps=num_interpolation_steps, fps=fps,
                    num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, eta=eta, upsample=upsample,
                    height=height, width=width, audio_filepath=audio_filepath, audio_start_sec=audio_start_sec,
                    negative_prompt=negative_prompt), indent=2, sort_keys=False))              # :694-714
        else:
            data = json.load(open(prompt_config_path))                                           # :716-729
            prompts = data["prompts"]
            seeds = data["seeds"]
            num_interpolation_steps = data["num_interpolation_steps"]
            fps = data["fps"]
            num_inference_steps = data["num_inference_steps"]
            guidance_scale = data["guidance_scale"]
            eta = data["eta"]
            upsample = data["upsample"]
            height = data["height"]
            width = data["width"]
            audio_filepath = data["audio_filepath"]
            audio_start_sec = data["audio_start_sec"]
            negative_prompt = data.get("negative_prompt", None)

        # pass 1: what has to be generated.  The reference resumes at "last frame on disk + 1" (:741-753), which is only
        # right for ONE sequential writer; here frames are sharded over ranks and written by a thread pool, so a killed run
        # leaves holes (rank 0 died at frame 20 of its block, rank 1 at frame 70 of its own).  Resume therefore regenerates
        # exactly the frames whose file is missing or empty (frames are renamed into place only when complete) - for the
        # reference's own on-disk states this is the same set, except that its :750 quirk (a clip with exactly one missing
        # frame is skipped and stays incomplete) is not reproduced.  `resume_policy = "reference"` (SDV_RESUME=reference) switches
        # to the reference's rule bit for bit: `resume_todo`.
        clips, all_clips = [], {}
        for i, (prompt_a, prompt_b, seed_a, seed_b, num_step) in enumerate(
                zip(prompts, prompts[1:], seeds, seeds[1:], num_interpolation_steps)):
            save_path = save_path_root / f"{name}_{i:06d}"
            step_output_filepath = save_path / f"{name}_{i:06d}.mp4"
            todo = list(range(num_step))
            all_clips[i] = dict(i=i, prompt_a=prompt_a, prompt_b=prompt_b, seed_a=seed_a, seed_b=seed_b, num_step=num_step,
                                todo=todo, save_path=save_path, mp4=step_output_filepath)
            if resume:
                todo = self.resume_todo(save_path, step_output_filepath, num_step, image_file_ext, self.resume_policy)
                if todo is None:
                    print(f"Skipping {save_path} because frames already exist")
                    continue
                if len(todo) < num_step:
                    print(f"Resuming {save_path.name}: {len(todo)} of {num_step} frames to generate (first {todo[0]})")
            all_clips[i]["todo"] = todo
            clips.append(all_clips[i])
        if world_size > 1:
            # every rank must shard the SAME work list: rank 0's view of the directory decides - which clips are still open
            # AND which of their frames are missing (another rank may list the directory a moment later and see more)
            work = parallel.broadcast_object([(c["i"], c["todo"]) for c in clips] if rank == 0 else None)
            clips = []
            for i, t in work:
                all_clips[i]["todo"] = list(t)
                clips.append(all_clips[i])
            parallel.barrier()      # every rank has looked at the directory before anyone writes new frames
        shares = parallel.partition_frame_list([c["todo"] for c in clips], world_size, rank)

        # fp8 mode: the pilot calibration's conditional half runs under the walk's FIRST prompt on every rank and on every resume
        # (prompts[0] of prompt_config.json, whatever frames are left to do), so all of them quantise identically
        if getattr(self, "fp8", False) and not getattr(self.unet, "fp8_calibrated", True):
            self._fp8_pilot_cond = self.embed_text(prompts[0]).float().contiguous()
        # pass 2: generate this rank's frames
        # this rank's runs of consecutive frames, coalesced per clip: the holes of a resumed clip fill whole batches together
        per_clip: Dict[int, List[int]] = {}
        for ci, first, stop in shares:
            per_clip.setdefault(ci, []).extend(range(first, stop))
        self._writer = FrameWriter()
        try:
            for ci, frames in per_clip.items():
                c = clips[ci]
                i, num_step = c["i"], c["num_step"]
                audio_offset = audio_start_sec + sum(num_interpolation_steps[:i]) / fps          # :755
                audio_duration = num_step / fps
                self.make_clip_frames(
                    c["prompt_a"], c["prompt_b"], c["seed_a"], c["seed_b"], num_interpolation_steps=num_step,
                    save_path=c["save_path"], num_inference_steps=num_inference_steps, guidance_scale=guidance_scale,
                    eta=eta, height=height, width=width, upsample=upsample, batch_size=batch_size,
                    T=get_timesteps_arr(audio_filepath, offset=audio_offset, duration=audio_duration, fps=fps,
                                        margin=margin, smooth=smooth) if audio_filepath else None,
                    frame_indices=frames, negative_prompt=negative_prompt, step=(i, len(prompts) - 1))
        finally:
            self._writer.close()
            self._writer = None
        if world_size > 1:
            parallel.barrier()      # all frames are on disk before rank 0 muxes
        if not make_video:
            return None
        result = None
        if rank == 0:
            for c in clips:
                i, num_step = c["i"], c["num_step"]
                audio_offset = audio_start_sec + sum(num_interpolation_steps[:i]) / fps
                make_video_pyav(c["save_path"], audio_filepath=audio_filepath, fps=fps, output_filepath=c["mp4"],
                                glob_pattern=f"*{image_file_ext}", audio_offset=audio_offset,
                                audio_duration=num_step / fps, sr=44100)                        # :787-796
            result = make_video_pyav(save_path_root, audio_filepath=audio_filepath, fps=fps, audio_offset=audio_start_sec,
                                     audio_duration=sum(num_interpolation_steps) / fps, output_filepath=output_filepath,
                                     glob_pattern=f"**/*{image_file_ext}", sr=44100)             # :798-807
        if world_size > 1:
            parallel.barrier()
            result = str(output_filepath) if result is None else result
        return result

"""``torch.ops.sdv.*`` - the hand-written gfx950 kernels of libsdv_hip.so registered as PyTorch custom ops
(``torch.library.custom_op``), the boundary BASELINE.json's north_star names ("Python host code calling hand-written CDNA4
HIP kernels through PyTorch-ROCm custom ops"; SURVEY.md 8b proposes exactly this op list).

Two layers, both in the ``sdv`` namespace:

* ``torch.ops.sdv.k_*`` (registered in ``hip.py``, 23 ops): ONE op per kernel launch of the C ABI (include/sdv_hip.h) - schema with
  the mutated outputs annotated, an implementation that passes ``data_ptr()`` + the current HIP stream to libsdv_hip.so, a
  Meta kernel.  **The product runs on these**: every wrapper in ``hip.py`` that ``engine.py`` / ``pipeline.py`` / ``text.py`` /
  ``esrgan.py`` call packs its arguments and calls the op (``tests/test_host.py::test_every_launch_is_a_torch_custom_op`` checks
  that nothing else touches ctypes), inside hipGraph capture as well.
* the ops below: tensor-in / tensor-out conveniences with PyTorch-style signatures (``linear``, ``conv3x3``, ``attention`` ...)
  built on the first layer - what an ATen-based caller, e.g. a diffusers model patched layer by layer, would use; they compose with
  ``torch.compile`` / FakeTensor shape propagation and show up by name in the PyTorch profiler.

There is NO CPU implementation in either layer: a CPU tensor raises ``SdvHipError``.

What each op replaces in the reference is cited on the C declaration it forwards to (include/sdv_hip.h):
    sdv::linear / sdv::conv3x3 / sdv::upsample_conv3x3 / sdv::attention / sdv::group_norm / sdv::layer_norm
                                            -> inside unet(...) / vae.decode(...)   stable_diffusion_pipeline.py:418, :433
    sdv::cfg_ddim_step                      -> :414, :422-426
    sdv::lerp_batch / sdv::slerp_batch      -> :467-468, utils.py:42-66
"""
from __future__ import annotations

from typing import Optional

import torch

from . import hip

_lib = torch.library


@_lib.custom_op("sdv::linear", mutates_args=())
def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           epi: int = 0, alpha: float = 1.0, alpha_cols: int = 0) -> torch.Tensor:
    """bf16 [M, K] x [N, K]^T (+ fp32 bias, + bf16 residual), epilogues 0 none / 1 GEGLU / 2 SiLU / 3-5 see sdv_hip.h."""
    return hip.linear(x, w, bias, residual=residual, epi=epi, alpha=alpha, alpha_cols=alpha_cols)


@linear.register_fake
def _(x, w, bias=None, residual=None, epi=0, alpha=1.0, alpha_cols=0):
    n = w.shape[0] // 2 if epi == 1 else w.shape[0]
    return x.new_empty((x.shape[0], n))


@_lib.custom_op("sdv::conv3x3", mutates_args=())
def conv3x3(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], nimg: int, H: int, W: int, stride: int = 1,
            residual: Optional[torch.Tensor] = None, circular: bool = False) -> torch.Tensor:
    """NHWC conv3x3 pad 1 over x [nimg*H*W, Cin] with OHWI weights [Cout, 9*Cin] (``weights.conv_w``); stride 1 or 2."""
    return hip.conv3x3(x, w, bias, nimg=nimg, H=H, W=W, mode=2 if stride == 2 else 1, residual=residual, circular=circular)


@conv3x3.register_fake
def _(x, w, bias, nimg, H, W, stride=1, residual=None, circular=False):
    ho, wo = ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)
    return x.new_empty((nimg * ho * wo, w.shape[0]))


@_lib.custom_op("sdv::upsample_conv3x3", mutates_args=())
def upsample_conv3x3(x: torch.Tensor, w4: torch.Tensor, bias: Optional[torch.Tensor], nimg: int, H: int, W: int,
                     circular: bool = False) -> torch.Tensor:
    """Upsample2D (nearest 2x, then conv3x3) in phase form; ``w4`` from ``weights.upconv_phase_w``."""
    return hip.upconv3x3_phase(x, w4, bias, nimg=nimg, H=H, W=W, circular=circular)


@upsample_conv3x3.register_fake
def _(x, w4, bias, nimg, H, W, circular=False):
    return x.new_empty((nimg * 4 * H * W, w4.shape[0] // 4))


@_lib.custom_op("sdv::attention", mutates_args=())
def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, heads: int, scale: float, causal: bool = False,
              q_prescaled: bool = False) -> torch.Tensor:
    """softmax(Q K^T scale) V without the score matrix.  q [B, Lq, C], k [B, Lk, C], vt = V TRANSPOSED [B, C, ldv] with
    ldv >= roundup(Lk, 64) and zero padding; C = heads * dh, dh in {40, 64, 80, 160}.  Returns [B, Lq, C]."""
    B, Lq, C = q.shape
    Lk = k.shape[1]
    out = torch.empty((B * Lq, C), dtype=torch.bfloat16, device=q.device)
    hip.attention(q.reshape(B * Lq, C), k.reshape(B * Lk, C), vt, out, B=B, H=heads, Lq=Lq, Lk=Lk, dh=C // heads, ldq=C, ldk=C,
                  ldv=vt.shape[2], ldo=C, scale=scale, causal=causal, q_prescaled=q_prescaled)
    return out.view(B, Lq, C)


@attention.register_fake
def _(q, k, vt, heads, scale, causal=False, q_prescaled=False):
    return q.new_empty(q.shape)


@_lib.custom_op("sdv::group_norm", mutates_args=())
def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, nimg: int, groups: int, eps: float,
               silu: bool = False) -> torch.Tensor:
    """GroupNorm (+ SiLU) over NHWC x [nimg*HW, C]."""
    return hip.groupnorm(x, gamma, beta, nimg=nimg, HW=x.shape[0] // nimg, groups=groups, eps=eps, silu=silu)


@group_norm.register_fake
def _(x, gamma, beta, nimg, groups, eps, silu=False):
    return x.new_empty(x.shape)


@_lib.custom_op("sdv::layer_norm", mutates_args=())
def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    return hip.layernorm(x, gamma, beta, eps)


@layer_norm.register_fake
def _(x, gamma, beta, eps=1e-5):
    return x.new_empty(x.shape)


@_lib.custom_op("sdv::cfg_ddim_step", mutates_args=("latents", "x2"))
def cfg_ddim_step(eps: torch.Tensor, latents: torch.Tensor, x2: torch.Tensor, coefs: torch.Tensor, step: torch.Tensor,
                  guidance: float, cfg: bool) -> None:
    """One fused classifier-free-guidance + DDIM update in place (latents fp32 NHWC; x2 = the next bf16 UNet input)."""
    hip.cfg_ddim_step(eps, latents, x2, coefs, step, None, guidance, cfg, latents.numel())


@_lib.custom_op("sdv::lerp_batch", mutates_args=())
def lerp_batch(a: torch.Tensor, b: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """out[f] = torch.lerp(a, b, T[f]) for every frame f in one launch (fp32)."""
    out = torch.empty((T.numel(),) + tuple(a.shape[1:] if a.shape[0] == 1 else a.shape), dtype=torch.float32, device=a.device)
    hip.lerp_batch(a.contiguous(), b.contiguous(), T, out_f32=out)
    return out


@lerp_batch.register_fake
def _(a, b, T):
    return a.new_empty((T.numel(),) + tuple(a.shape[1:] if a.shape[0] == 1 else a.shape), dtype=torch.float32)   # always fp32


@_lib.custom_op("sdv::slerp_batch", mutates_args=())
def slerp_batch(v0: torch.Tensor, v1: torch.Tensor, T: torch.Tensor, dot_threshold: float = 0.9995) -> torch.Tensor:
    """out[f] = slerp(T[f], v0, v1) over the WHOLE tensors (utils.py:42-66), all frames in one launch (fp32)."""
    stats = hip.slerp_stats(v0.contiguous(), v1.contiguous())
    out = hip.slerp_batch(v0.contiguous(), v1.contiguous(), stats, T, C_=1, HW=v0.numel(), to_hwc=False,
                          dot_threshold=dot_threshold)
    return out.view((T.numel(),) + tuple(v0.shape[1:] if v0.shape[0] == 1 else v0.shape))


@slerp_batch.register_fake
def _(v0, v1, T, dot_threshold=0.9995):
    return v0.new_empty((T.numel(),) + tuple(v0.shape[1:] if v0.shape[0] == 1 else v0.shape), dtype=torch.float32)


OPS = ("linear", "conv3x3", "upsample_conv3x3", "attention", "group_norm", "layer_norm", "cfg_ddim_step", "lerp_batch",
       "slerp_batch")

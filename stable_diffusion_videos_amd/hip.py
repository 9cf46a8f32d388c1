"""ctypes binding of libsdv_hip.so (C ABI: include/sdv_hip.h).

PyTorch is used for device memory and streams only: every wrapper takes torch tensors, checks
dtype/device/contiguity, and passes ``data_ptr()`` + the current HIP stream to the C entry point.
There is NO fallback: if the library is missing or a tensor is not on the GPU the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

import torch

# SDV_HIP_LIB: developer knob - load an alternative build of the SAME C ABI (tools/ubench/build_whatif.py timing variants)
_LIB_PATH = Path(os.environ["SDV_HIP_LIB"]) if os.environ.get("SDV_HIP_LIB") else Path(__file__).resolve().parent / "lib" / "libsdv_hip.so"
_lib = None


class SdvHipError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("X2", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("R", C.c_void_p),
        ("C", C.c_void_p), ("step_ptr", C.c_void_p), ("zero_page", C.c_void_p),
        ("sX", C.c_int64), ("sW", C.c_int64), ("sC", C.c_int64), ("sR", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("ldx", C.c_int32), ("ldx2", C.c_int32), ("C1", C.c_int32), ("ldw", C.c_int32), ("ldc", C.c_int32),
        ("ldr", C.c_int32),
        ("mode", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
        ("circular", C.c_int32),
        ("epi", C.c_int32), ("bias_mode", C.c_int32), ("bias_step_stride", C.c_int32),
        ("batch", C.c_int32), ("tile", C.c_int32), ("alpha", C.c_float),
        ("div_hw_mul", C.c_uint32), ("div_hw_shr", C.c_uint32), ("div_w_mul", C.c_uint32), ("div_w_shr", C.c_uint32),
        ("alpha_cols", C.c_int32),
        ("ln_stats", C.c_void_p), ("ln_s", C.c_void_p), ("stats_out", C.c_void_p), ("ln_side", C.c_int32), ("stats_p", C.c_int32),
        ("fp8", C.c_int32), ("out_mode", C.c_int32), ("out_f32", C.c_void_p), ("out_u8", C.c_void_p),
        ("gn_out", C.c_void_p), ("gn_ld", C.c_int32), ("split_k", C.c_int32),
    ]


ABI_VERSION = 12    # sdv_abi_version() of the library this binding (struct layouts, signatures) was written against


_SIGNATURES = {
    "sdv_last_error": (C.c_char_p, []),
    "sdv_abi_version": (C.c_int, []),
    "sdv_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "sdv_gemm_stats_slots": (C.c_int, [C.POINTER(GemmArgs)]),
    "sdv_gemm_split_k": (C.c_int, [C.POINTER(GemmArgs)]),
    "sdv_gemm_set_persistent": (C.c_int, [C.c_int]),
    "sdv_gemm_set_grid_limit": (C.c_int, [C.c_int]),
    "sdv_rowstats_finalize": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "sdv_ffn_geglu_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [C.c_int32, C.c_void_p]),
    "sdv_linear320_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sdv_linear640_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_float, C.c_void_p]),
    "sdv_attention_bf16": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 9 + [C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sdv_softmax_rows_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "sdv_softmax_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sdv_groupnorm_finalize": (C.c_int, ([C.c_void_p] + [C.c_int32] * 4 + [C.c_int64]) * 2 + [C.c_int32] * 3 + [C.c_void_p, C.c_void_p]),
    "sdv_groupnorm_stats": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]),
    "sdv_groupnorm_apply": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p] * 3 +
                            [C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "sdv_groupnorm_apply_fp8": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p] * 3 +
                                [C.c_float, C.c_int32, C.c_void_p, C.c_float, C.c_void_p]),
    "sdv_groupnorm_fp8_set_saturation_counter": (C.c_int, [C.c_void_p]),
    "sdv_layernorm_bf16": (C.c_int, [C.c_void_p] * 3 + [C.c_float, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "sdv_conv3x3_cin_small": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 6 + [C.c_void_p]),
    "sdv_im2col3x3_c4": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]),
    "sdv_latent_affine": (C.c_int, [C.c_void_p] * 3 + [C.c_float, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "sdv_slerp_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "sdv_slerp_batch": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_float, C.c_void_p, C.c_void_p]),
    "sdv_lerp_batch": (C.c_int, [C.c_void_p] * 3 + [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sdv_cfg_ddim_step": (C.c_int, [C.c_void_p] * 6 + [C.c_float, C.c_int32, C.c_int64, C.c_void_p]),
    "sdv_cfg_multistep_step": (C.c_int, [C.c_void_p] * 8 + [C.c_float, C.c_int32, C.c_int64, C.c_void_p]),
    "sdv_latents_to_unet_input": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "sdv_step_counter_add": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "sdv_timestep_embedding": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                         C.c_void_p]),
    "sdv_linear_small": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 4 + [C.c_void_p]),
    "sdv_nchw_to_nhwc_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sdv_nhwc_to_nchw_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sdv_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "sdv_embed_tokens": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sdv_rgb_u8_to_bf16_c4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "sdv_axpby_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32,
                                 C.c_float, C.c_float, C.c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib_path() -> Path:
    return _LIB_PATH


def load(build_if_missing: bool = False):
    """dlopen libsdv_hip.so and bind every symbol of include/sdv_hip.h.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        if build_if_missing:
            from . import build as _build
            _build.build()
        else:
            raise SdvHipError(
                f"{_LIB_PATH} is missing - build it with `python -m stable_diffusion_videos_amd.build` "
                "(there is no CPU / eager fallback for the hot path)")
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.sdv_abi_version() != ABI_VERSION:
        raise SdvHipError(f"{_LIB_PATH} reports ABI version {lib.sdv_abi_version()}, this binding needs {ABI_VERSION}: "
                          "rebuild it with `python -m stable_diffusion_videos_amd.build --force`")
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = _lib.sdv_last_error().decode(errors="replace") if _lib is not None else ""
        raise SdvHipError(f"{what} failed (rc={rc}): {msg}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor], dtype=None, name="tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise SdvHipError(f"{name} must live in GPU memory (got device {t.device}); the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise SdvHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


BF16 = torch.bfloat16
F32 = torch.float32
FP8 = torch.float8_e4m3fn      # OCP e4m3, what gfx950's fp8 MFMA and converters use
FP8_MAX = 448.0

_zero_pages = {}
FP8_MX = int(os.environ.get("SDV_FP8_MX", "1"))            # developer knob for A/B: 0 = fp8 operands on the plain (bf16-rate) fp8 MFMA
# Test / tools knob: launches that leave the tile to the cost model (tile=0) are forced onto this tile where it exists for them
# (6 = the persistent 256 x 320 tile).  With it - and sdv_gemm_set_grid_limit - a 2-sample forward runs the SAME tile code paths as
# the 256-sample benchmark forward, so they can be put under the oracle's block-wise absolute gate (tests/test_blockwise_gpu.py).
FORCE_TILE = int(os.environ.get("SDV_FORCE_TILE", "0"))

# Optional launch observer used by bench.py's roofline pass: called as hook(kind, info_dict, launch_fn).  The
# hook must call launch_fn() itself (it may bracket it with HIP events).  None = no overhead.
LAUNCH_HOOK = None


def _launch(kind: str, info: dict, fn):
    if LAUNCH_HOOK is None:
        fn()
    else:
        LAUNCH_HOOK(kind, info, fn)


# ------------------------------------------------------------------------------------------------
# torch.ops.sdv.k_* : every launch on the hot path is a PyTorch custom op
# ------------------------------------------------------------------------------------------------
# BASELINE.json's north_star: "Python host code calling hand-written CDNA4 HIP kernels through PyTorch-ROCm custom ops".  The public
# wrappers below (gemm / linear / conv3x3 / attention / groupnorm / cfg_*_step / lerp / slerp ...) - which is what engine.py and
# pipeline.py call - do not touch ctypes themselves: they pack their arguments and call ``torch.ops.sdv.k_<name>``, a dispatcher
# op registered here with a schema (mutated outputs annotated), an implementation that hands ``data_ptr()`` + the current HIP
# stream to the C ABI, and a Meta (fake) kernel.  Registered through ``torch.library.Library`` rather than
# ``torch.library.custom_op``: 6 us per call instead of 31 us (measured), which matters for the ~400 launches of an eager UNet
# forward and not at all once a step is a hipGraph replay.  There is still no CPU kernel: the one implementation raises
# ``SdvHipError`` for a tensor that is not in GPU memory.
_OPLIB = torch.library.Library("sdv", "FRAGMENT")
KERNEL_OPS = []


def _defop(schema: str, impl, fake=None):
    name = schema[:schema.index("(")]
    _OPLIB.define(schema)
    _OPLIB.impl(name, impl, "CompositeExplicitAutograd")
    _OPLIB.impl(name, fake if fake is not None else (lambda *a, **k: None), "Meta")
    KERNEL_OPS.append(name)
    return getattr(torch.ops.sdv, name)


def zero_page(device) -> torch.Tensor:
    key = str(device)
    if key not in _zero_pages:
        _zero_pages[key] = torch.zeros(256, dtype=torch.uint8, device=device)
    return _zero_pages[key]


# ------------------------------------------------------------------------------------------------
# GEMM / conv
# ------------------------------------------------------------------------------------------------
_GEMM_INTS = ("M", "N", "K", "ldx", "ldw", "ldc", "ldr", "C1", "ldx2", "epi", "mode", "Hin", "Win", "Hout", "Wout", "circular", "batch",
              "sX", "sW", "sC", "sR", "bias_mode", "bias_step_stride", "tile", "x_off", "w_off", "out_off", "alpha_cols", "out_mode")


GN_EPILOGUE = os.environ.get("SDV_GN_EPILOGUE", "1") != "0"     # A/B knob: 0 = every GroupNorm runs its own statistics pass
SPLIT_K = os.environ.get("SDV_SPLIT_K", "1") != "0"             # A/B knob: 0 = no split-K launches (small-batch regime)
LAST_SPLIT_K = 1


class GnStats:
    """GroupNorm statistics that left the producing igemm's epilogue (sdv_gemm_args.gn_out): ``p`` fp32 [blocks, 2, C] of
    (sum, sumsq) per 32-row block and channel of a tensor of ``nimg`` images with ``HW`` pixels each; ``nrep`` repetitions
    ``rep_stride`` blocks apart (the four phases of a phase-form up-conv).  Travels as ``tensor._sdv_gn`` with the tensor the
    producer returned; ``groupnorm`` uses it instead of a statistics pass when it matches what it is asked to normalise - same
    geometry AND the tensor's version counter still where it was when the statistics were attached (``ver``): any in-place torch
    writer since (``copy_``, ``add_``, an ``out=`` op) makes ``groupnorm`` fall back to its own statistics pass."""
    __slots__ = ("p", "C", "nimg", "HW", "bpi", "nrep", "rep_stride", "ver")

    def __init__(self, p, C, nimg, HW, bpi, nrep, rep_stride, ver=-1):
        self.p, self.C, self.nimg, self.HW, self.bpi, self.nrep, self.rep_stride, self.ver = p, C, nimg, HW, bpi, nrep, rep_stride, ver


def _ver(t: torch.Tensor) -> int:
    try:
        return t._version
    except RuntimeError:          # inference tensors do not track versions: nothing to compare, the geometry check stands alone
        return -1


def gn_repeat(t_src: torch.Tensor, t_dst: torch.Tensor, times: int):
    """``t_dst`` holds ``times`` copies of ``t_src``'s images back to back (the CFG-shared skip tensor): its statistics are the
    source's, repeated."""
    g = getattr(t_src, "_sdv_gn", None)
    if g is not None and g.nrep == 1:
        t_dst._sdv_gn = GnStats(torch.cat([g.p] * times), g.C, g.nimg * times, g.HW, g.bpi, 1, 0, _ver(t_dst))


def gn_slice(t: torch.Tensor, first_img: int, n_img: int, HW: int) -> torch.Tensor:
    """Rows of images [first_img, first_img + n_img) of ``t`` - with the matching slice of its epilogue statistics, so that a
    forward that walks the batch in chunks of images normalises with exactly the numbers the whole-batch forward uses."""
    out = t[first_img * HW:(first_img + n_img) * HW]
    g = getattr(t, "_sdv_gn", None)
    if g is not None and g.HW == HW:
        if g.nrep == 1:
            p = g.p[first_img * g.bpi:(first_img + n_img) * g.bpi]
            out._sdv_gn = GnStats(p, g.C, n_img, HW, g.bpi, 1, 0, g.ver)
        else:
            p = g.p.view(g.nrep, g.rep_stride, 2, g.C)[:, first_img * g.bpi:(first_img + n_img) * g.bpi].contiguous()
            out._sdv_gn = GnStats(p.view(-1, 2, g.C), g.C, n_img, HW, g.bpi, g.nrep, n_img * g.bpi, g.ver)
    return out


def gn_join(parts, whole: torch.Tensor):
    """The statistics of ``whole`` from those of its consecutive image chunks ``parts`` (all produced with nrep == 1)."""
    gs = [getattr(t, "_sdv_gn", None) for t in parts]
    if gs and all(g is not None and g.nrep == 1 for g in gs):
        whole._sdv_gn = GnStats(torch.cat([g.p for g in gs]), gs[0].C, sum(g.nimg for g in gs), gs[0].HW, gs[0].bpi, 1, 0, _ver(whole))


def gn_epilogue_ok(*, M, N, epi, mode, ldc, ldr, out, bias, residual, fp8, ln, want_stats, out_mode, HW_out, batch, sC, sR) -> bool:
    """Can this igemm launch emit the GroupNorm statistics of its output (sdv_hip.h gn_out)?  ``HW_out``: pixels per image of the
    OUTPUT tensor.  Mirrors the checks of sdv_gemm_bf16."""
    if not GN_EPILOGUE or epi != 0 or out_mode or fp8 or ln is not None or want_stats or out is None:
        return False
    rows_per_image = HW_out // 4 if mode == 4 else HW_out           # rows of ONE launch phase that belong to one image
    if M % 32 or rows_per_image % 32 or N % 8 or ldc % 8 or (residual is not None and ldr % 8) or (sC | sR) % 8:
        return False
    ptrs = out.data_ptr() | (bias.data_ptr() if bias is not None else 0) | (residual.data_ptr() if residual is not None else 0)
    return ptrs % 16 == 0


def _igemm_impl(x, w, out, bias, residual, x2, ln_stats, ln_s, step_ptr, out_f32, out_u8, gn_out, ints, alpha, ln_eps, want_stats):
    """sdv::k_igemm - the launch of ``sdv_gemm_bf16`` (+ ``sdv_rowstats_finalize`` when the row statistics are wanted)."""
    g = dict(zip(_GEMM_INTS, ints))
    M, N, K, batch, mode, epi = g["M"], g["N"], g["K"], g["batch"], g["mode"], g["epi"]
    lib = load()
    a = GemmArgs()
    fp8 = x.dtype == FP8
    if fp8:
        if w.dtype != FP8 or (x2 is not None and x2.dtype != FP8) or g["x_off"] or g["w_off"]:
            raise SdvHipError("gemm: fp8 activations need fp8 weights (and no operand offsets)")
        # 2 = the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 form (twice the MFMA rate; sdv_hip.h), 1 = v_mfma_f32_32x32x16_fp8_fp8
        a.fp8 = 2 if FP8_MX else 1
        a.X, a.X2, a.W = _ptr(x, FP8, "X"), _ptr(x2, FP8, "X2"), _ptr(w, FP8, "W")
    else:
        a.X = _ptr(x, BF16, "X") + 2 * g["x_off"]
        a.X2 = _ptr(x2, BF16, "X2")
        a.W = _ptr(w, BF16, "W") + 2 * g["w_off"]
    a.bias = _ptr(bias, F32, "bias")
    a.R = _ptr(residual, BF16, "R")
    a.C = _ptr(out, BF16, "C") + 2 * g["out_off"] if out is not None else None
    a.out_mode, a.out_f32, a.out_u8 = g["out_mode"], _ptr(out_f32, F32, "out_f32"), _ptr(out_u8, torch.uint8, "out_u8")
    a.step_ptr = _ptr(step_ptr, torch.int32, "step_ptr")
    a.zero_page = zero_page(x.device).data_ptr()
    a.sX, a.sW, a.sC, a.sR = g["sX"], g["sW"], g["sC"], g["sR"]
    a.M, a.N, a.K = M, N, K
    a.ldx, a.ldx2, a.C1, a.ldw, a.ldc, a.ldr = g["ldx"], g["ldx2"], g["C1"], g["ldw"], g["ldc"], g["ldr"]
    a.mode, a.Hin, a.Win, a.Hout, a.Wout, a.circular = mode, g["Hin"], g["Win"], g["Hout"], g["Wout"], g["circular"]
    a.epi, a.bias_mode, a.bias_step_stride = epi, (g["bias_mode"] if bias is not None else 0), g["bias_step_stride"]
    a.batch, a.tile, a.alpha, a.alpha_cols = batch, g["tile"], alpha, g["alpha_cols"]
    if FORCE_TILE and not g["tile"] and not g["out_mode"] and epi < 3:
        a.tile = FORCE_TILE
    if ln_stats is not None:
        a.ln_stats, a.ln_s, a.ln_side = _ptr(ln_stats, F32, "ln_stats"), _ptr(ln_s, F32, "ln_s"), 1
    if gn_out is not None:
        a.gn_out, a.gn_ld = _ptr(gn_out, F32, "gn_out"), gn_out.shape[-1]
    split_used = False
    # Split-K for the small-batch regime (sdv_hip.h split_k): plain launches with few rows and a long K ask the library how many
    # splits it would take and hand it the workspace (the GroupNorm statistics, if asked for, then come out of the second pass).
    # (alpha_cols - alpha on the leading columns only - is not something the second pass knows: such launches stay unsplit)
    if (SPLIT_K and not fp8 and ln_stats is None and not want_stats and not g["out_mode"] and epi == 0 and batch <= 1 and mode != 4
            and out_f32 is None and not g["alpha_cols"] and M <= 4096 and K * (1 if mode == 0 else 9) >= 2048):
        a.split_k = 8
        S = lib.sdv_gemm_split_k(C.byref(a))
        if S > 1:
            ws = torch.empty((S, M, N), dtype=F32, device=x.device)
            a.out_f32, a.split_k = ws.data_ptr(), S
            split_used = True
        else:
            a.split_k = 0
    global LAST_SPLIT_K
    LAST_SPLIT_K = a.split_k if split_used else 1    # (tests / tools: how the last igemm launch was split)
    partials = None
    if want_stats:
        a.stats_out = 16          # (non-null while planning: the tile choice depends on it)
        slots = lib.sdv_gemm_stats_slots(C.byref(a))
        if slots <= 0:
            _check(-1, "sdv_gemm_stats_slots")
        partials = torch.empty((max(batch, 1) * M, slots, 2), dtype=F32, device=x.device)
        a.stats_out = partials.data_ptr()
    taps = 1 if mode == 0 else 9
    # algorithmic work: the phase form (mode 4) is a nearest-2x upsample + conv3x3 on 4*M output pixels (9 taps each);
    # it EXECUTES 4 taps per output pixel
    flops = 2.0 * M * N * K * taps * (4 if mode == 4 else batch)
    _launch("gemm" if mode == 0 else "conv3x3",
            dict(M=M * (4 if mode == 4 else 1), N=N, K=K * taps, batch=batch, flops=flops, mode=mode, epi=epi),
            lambda: _check(lib.sdv_gemm_bf16(C.byref(a), _stream()), "sdv_gemm_bf16"))
    if want_stats:
        nout = N // 2 if epi == 1 else N
        stats = torch.empty((max(batch, 1) * M, 2), dtype=F32, device=x.device)
        _launch("ln_stats", dict(bytes=8.0 * partials.shape[0] * (partials.shape[1] + 1)),
                lambda: _check(lib.sdv_rowstats_finalize(partials.data_ptr(), partials.shape[0], partials.shape[1], nout, ln_eps,
                                                         stats.data_ptr(), _stream()), "sdv_rowstats_finalize"))
        return stats
    return torch.empty(0, dtype=F32, device=x.device)


def _igemm_fake(x, w, out, bias, residual, x2, ln_stats, ln_s, step_ptr, out_f32, out_u8, gn_out, ints, alpha, ln_eps, want_stats):
    g = dict(zip(_GEMM_INTS, ints))
    return x.new_empty((max(g["batch"], 1) * g["M"], 2) if want_stats else (0,), dtype=F32)


_k_igemm = _defop("k_igemm(Tensor x, Tensor w, Tensor(a!)? out, Tensor? bias, Tensor? residual, Tensor? x2, Tensor? ln_stats, "
                  "Tensor? ln_s, Tensor? step_ptr, Tensor(b!)? out_f32, Tensor(c!)? out_u8, Tensor(d!)? gn_out, int[] ints, float alpha, "
                  "float ln_eps, bool want_stats) -> Tensor", _igemm_impl, _igemm_fake)


def gemm(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int, ldx: int, ldw: int,
         ldc: int, bias: Optional[torch.Tensor] = None, bias_mode: int = 1, residual: Optional[torch.Tensor] = None,
         ldr: int = 0, x2: Optional[torch.Tensor] = None, C1: int = 0, ldx2: int = 0, alpha: float = 1.0,
         epi: int = 0, mode: int = 0, Hin: int = 0, Win: int = 0, Hout: int = 0, Wout: int = 0,
         circular: bool = False, batch: int = 1, sX: int = 0, sW: int = 0, sC: int = 0, sR: int = 0,
         step_ptr: Optional[torch.Tensor] = None, bias_step_stride: int = 0, tile: int = 0,
         x_off: int = 0, w_off: int = 0, out_off: int = 0, alpha_cols: int = 0, ln=None,
         want_stats: bool = False, ln_eps: float = 1e-5, out_mode: int = 0, out_f32: Optional[torch.Tensor] = None,
         out_u8: Optional[torch.Tensor] = None, gn_hw: int = 0):
    """``sdv_gemm_bf16`` through ``torch.ops.sdv.k_igemm`` (element offsets x_off / w_off / out_off select sub-matrices).
    ``out_mode`` 1 / 2: fp32 output / image epilogue into ``out_f32`` / ``out_u8`` (``out`` may be None), see sdv_hip.h.
    ``ln=(stats, s)``: LayerNorm folded into this GEMM (sdv_hip.h): ``stats`` fp32 [rows, 2] = (mean, rstd) from
    ``want_stats`` of the producer, ``s`` fp32 row sums of the gamma-scaled weights.  ``want_stats=True`` returns the
    (mean, rstd) [batch*M, 2] of this GEMM's OUTPUT rows over its N columns (for the next LayerNorm).
    ``gn_hw`` > 0: the output is an NHWC image tensor with ``gn_hw`` pixels per image whose next consumer is a GroupNorm - where the
    launch qualifies (``gn_epilogue_ok``) the epilogue also emits the per-channel statistics and ``out._sdv_gn`` carries them."""
    ints = [M, N, K, ldx, ldw, ldc, ldr, C1, ldx2, epi, mode, Hin, Win, Hout, Wout, int(circular), batch, sX, sW, sC, sR, bias_mode,
            bias_step_stride, tile, x_off, w_off, out_off, alpha_cols, out_mode]
    gn_out = None
    nb = max(batch, 1)
    if gn_hw and out_off == 0 and gn_epilogue_ok(M=M, N=N, epi=epi, mode=mode, ldc=ldc, ldr=ldr, out=out, bias=bias, residual=residual,
                                                 fp8=x.dtype == FP8, ln=ln, want_stats=want_stats, out_mode=out_mode, HW_out=gn_hw,
                                                 batch=batch, sC=sC, sR=sR):
        nrep = 4 if mode == 4 else 1
        rows = nb * M if mode != 4 else M                       # rows of one repetition
        gn_out = torch.empty((nrep * rows // 32, 2, N), dtype=F32, device=x.device)
    st = _k_igemm(x, w, out, bias, residual, x2, ln[0] if ln is not None else None, ln[1] if ln is not None else None, step_ptr,
                  out_f32, out_u8, gn_out, ints, float(alpha), float(ln_eps), bool(want_stats))
    if gn_out is not None:
        rows_per_image = gn_hw // 4 if mode == 4 else gn_hw
        nimg = (M if mode == 4 else nb * M) // rows_per_image
        out._sdv_gn = GnStats(gn_out, N, nimg, gn_hw, rows_per_image // 32, 4 if mode == 4 else 1, (M // 32) if mode == 4 else 0, _ver(out))
    elif out is not None and hasattr(out, "_sdv_gn"):
        del out._sdv_gn
    return st if want_stats else None


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, residual=None, out=None,
           epi: int = 0, x2: Optional[torch.Tensor] = None, alpha: float = 1.0, tile: int = 0, alpha_cols: int = 0, ln=None,
           want_stats: bool = False, ln_eps: float = 1e-5, gn_hw: int = 0):
    """y[M,N] = epi(x[M,K] @ w[N,K]^T + bias) (+ residual); x2 = optional second K-source (concat).
    ``ln`` / ``want_stats``: see ``gemm`` (with want_stats the return value is ``(y, stats)``)."""
    M, K1 = x.shape
    K2 = x2.shape[1] if x2 is not None else 0
    N, K = w.shape
    if K != K1 + K2:
        raise SdvHipError(f"linear: K mismatch {K} vs {K1}+{K2}")
    n_out = N // 2 if epi == 1 else N
    if out is None:
        out = torch.empty((M, n_out), dtype=BF16, device=x.device)
    st = gemm(x, w, out, M=M, N=N, K=K, ldx=x.stride(0), ldw=w.stride(0), ldc=out.stride(0), bias=bias,
              residual=residual, ldr=residual.stride(0) if residual is not None else 0, x2=x2, C1=K1 if x2 is not None else 0,
              ldx2=x2.stride(0) if x2 is not None else 0, alpha=alpha, epi=epi, tile=tile, alpha_cols=alpha_cols, ln=ln,
              want_stats=want_stats, ln_eps=ln_eps, gn_hw=gn_hw)
    return (out, st) if want_stats else out


def conv3x3(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, nimg: int, H: int, W: int,
            mode: int = 1, x2: Optional[torch.Tensor] = None, residual=None, circular: bool = False,
            step_ptr=None, bias_step_stride: int = 0, out=None, tile: int = 0, epi: int = 0,
            alpha: float = 1.0, out_mode: int = 0, out_f32=None, out_u8=None, gn: bool = False) -> torch.Tensor:
    """NHWC conv3x3 pad 1.  x: [nimg*H*W, C1] (+ x2 [.., C2]); w: [Cout, 9*(C1+C2)] (OHWI).
    mode 1: stride 1; 2: stride 2; 3: nearest-2x upsample then conv.  x / out / residual may be column
    slices of wider row-major buffers (row stride = .stride(0)): dense-block concat without copies.
    ``out_mode`` 1: fp32 result in ``out_f32`` [M, Cout]; 2: image epilogue (clamp(v/2+0.5), fp32 in ``out_f32`` and / or
    uint8 in ``out_u8``) - the Cout <= 4 output convolutions of the UNet / VAE on the matrix cores; nothing is returned."""
    C1 = x.shape[1]
    C2 = x2.shape[1] if x2 is not None else 0
    Cout = w.shape[0]
    if w.shape[1] != 9 * (C1 + C2):
        raise SdvHipError(f"conv3x3: weight K {w.shape[1]} != 9*{C1 + C2}")
    if mode == 4:
        return upconv3x3_phase(x, w, bias, nimg=nimg, H=H, W=W, circular=circular, out=out, tile=tile, gn=gn)
    if mode == 1:
        Ho, Wo = H, W
    elif mode == 2:
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
    else:
        Ho, Wo = 2 * H, 2 * W
    M = nimg * Ho * Wo
    if out_mode:
        if (out_f32 is None and out_u8 is None) or out is not None or residual is not None:
            raise SdvHipError("conv3x3: out_mode needs out_f32 / out_u8 (and takes no bf16 output / residual)")
        ldc = Cout
    else:
        if out is None:
            out = torch.empty((M, Cout), dtype=BF16, device=x.device)
        ldc = out.stride(0)
    gemm(x, w, out, M=M, N=Cout, K=C1 + C2, ldx=x.stride(0), ldw=w.stride(0), ldc=ldc, bias=bias,
         residual=residual, ldr=residual.stride(0) if residual is not None else 0, x2=x2,
         C1=C1 if x2 is not None else 0, ldx2=x2.stride(0) if x2 is not None else 0, mode=mode, Hin=H, Win=W,
         Hout=Ho, Wout=Wo, circular=circular, step_ptr=step_ptr, bias_step_stride=bias_step_stride, tile=tile,
         epi=epi, alpha=alpha, out_mode=out_mode, out_f32=out_f32, out_u8=out_u8, gn_hw=Ho * Wo if gn else 0)
    return out


def upconv3x3_phase(x: torch.Tensor, w4: torch.Tensor, bias: Optional[torch.Tensor], *, nimg: int, H: int, W: int,
                    circular: bool = False, out=None, tile: int = 0, gn: bool = False) -> torch.Tensor:
    """Upsample2D (nearest 2x, then conv3x3 pad 1) in phase form: x [nimg*H*W, Cin] -> [nimg*2H*2W, Cout].
    ``w4``: [4*Cout, 4*Cin] from ``weights.upconv_phase_w`` (phase-major; 2x2 taps with the coincident 3x3 taps summed)."""
    Cin = x.shape[1]
    if w4.shape[0] % 4 or w4.shape[1] != 4 * Cin:
        raise SdvHipError(f"upconv3x3_phase: weight shape {tuple(w4.shape)} does not match Cin={Cin}")
    Cout = w4.shape[0] // 4
    if out is None:
        out = torch.empty((nimg * 4 * H * W, Cout), dtype=BF16, device=x.device)
    gemm(x, w4, out, M=nimg * H * W, N=Cout, K=Cin, ldx=x.stride(0), ldw=w4.stride(0), ldc=out.stride(0), bias=bias, mode=4,
         Hin=H, Win=W, Hout=H, Wout=W, circular=circular, tile=tile, gn_hw=4 * H * W if gn else 0)
    return out


# ------------------------------------------------------------------------------------------------
# attention / norms
# ------------------------------------------------------------------------------------------------
def _attention_impl(q, k, vt, out, ints, scale, causal, q_prescaled, v_rowmajor):
    B, H, Lq, Lk, dh, ldq, ldk, ldv, ldo, q_off, k_off, v_off = ints
    lib = load()
    qp, kp, op = _ptr(q, BF16, "Q") + 2 * q_off, _ptr(k, BF16, "K") + 2 * k_off, _ptr(out, BF16, "O")
    vp = _ptr(vt, BF16, "V") + 2 * v_off
    _launch("attention", dict(B=B, H=H, Lq=Lq, Lk=Lk, dh=dh, flops=4.0 * B * H * Lq * Lk * dh),
            lambda: _check(lib.sdv_attention_bf16(qp, kp, vp, op, B, H, Lq, Lk, dh, ldq, ldk, ldv, ldo, scale, int(causal),
                                                  int(q_prescaled), int(v_rowmajor), _stream()),
                           "sdv_attention_bf16"))


_k_attention = _defop("k_attention(Tensor q, Tensor k, Tensor vt, Tensor(a!) out, int[] ints, float scale, bool causal, "
                      "bool q_prescaled, bool v_rowmajor) -> ()", _attention_impl)


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, *, B: int, H: int, Lq: int,
              Lk: int, dh: int, ldq: int, ldk: int, ldv: int, ldo: int, scale: float, q_off: int = 0, k_off: int = 0,
              v_off: int = 0, causal: bool = False, q_prescaled: bool = False, v_rowmajor: bool = False):
    """softmax(Q K^T scale) V (``torch.ops.sdv.k_attention``).  ``q_prescaled``: Q already holds q * scale * log2(e)
    (``LOG2E_SCALE(dh)`` applied as the ``alpha`` of its projection GEMM, one bf16 rounding in total) and ``scale`` is ignored.
    ``vt``: V transposed [B][H*dh][ldv] (the text context's V^T) or, with ``v_rowmajor``, V as a projection wrote it -
    [B*Lk][ldv] rows, head h at columns v_off + [h*dh, (h+1)*dh): the V third of a fused [Q | K | V] projection."""
    _k_attention(q, k, vt, out, [B, H, Lq, Lk, dh, ldq, ldk, ldv, ldo, q_off, k_off, v_off], float(scale), bool(causal),
                 bool(q_prescaled), bool(v_rowmajor))


def _ffn_geglu_impl(x, ln_stats, w1, w1x, w2p, bias2, out):
    lib = load()
    M, C = x.shape
    if tuple(w1.shape) != (8 * C, C) or tuple(w1x.shape) != (8 * C, 16) or tuple(w2p.shape) != (C, 4 * C) or not (
            w1.is_contiguous() and w1x.is_contiguous() and w2p.is_contiguous()):
        raise SdvHipError(f"ffn_geglu: weights must be contiguous [8C, C] / [8C, 16] / [C, 4C] for C = {C}, got {tuple(w1.shape)} / "
                          f"{tuple(w1x.shape)} / {tuple(w2p.shape)}")
    if ln_stats.numel() != 2 * M or bias2.numel() != C or out.shape != x.shape:
        raise SdvHipError("ffn_geglu: vector / output sizes do not match the activation")
    args = (_ptr(x, BF16, "X"), _ptr(ln_stats, F32, "ln_stats"), M, C, x.stride(0), _ptr(w1, BF16, "W1"), _ptr(w1x, BF16, "W1x"),
            _ptr(w2p, BF16, "W2p"), _ptr(bias2, F32, "bias2"), _ptr(out, BF16, "out"), out.stride(0))
    # algorithmic work: both projections (2 M (8C C + C 4C)); bytes: x in, out back (the hidden activations never touch HBM)
    _launch("ffn_geglu", dict(M=M, N=8 * C, K=C, flops=2.0 * M * 12 * C * C, bytes=4.0 * M * C),
            lambda: _check(lib.sdv_ffn_geglu_bf16(*args, _stream()), "sdv_ffn_geglu_bf16"))


_k_ffn_geglu = _defop("k_ffn_geglu(Tensor x, Tensor ln_stats, Tensor w1, Tensor w1x, Tensor w2p, Tensor bias2, Tensor(a!) out) -> ()", _ffn_geglu_impl)
FFN_FUSED = os.environ.get("SDV_FFN_FUSED", "1") != "0"     # A/B knob: 0 = the C = 320 feed-forward as two igemm launches (round 5)


def ffn_geglu(x: torch.Tensor, ln_stats: torch.Tensor, w1: torch.Tensor, w1x: torch.Tensor, w2p: torch.Tensor, bias2: torch.Tensor,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x + ff.net.2(GEGLU(ff.net.0(LayerNorm(x)))) in ONE launch (``torch.ops.sdv.k_ffn_geglu`` -> sdv_ffn_geglu_bf16, C = 320 only).
    ``w1``: the gamma-folded, GEGLU-interleaved ff.net.0 weight of ``weights.ln_fold(weights.geglu_interleave(...))``; ``w1x``:
    ``weights.ffn_fold_columns(s, t)`` of the same call; ``w2p``: ``weights.ffn_w2_permute(ff.net.2.weight)``; ``ln_stats`` [M, 2]:
    (mean, rstd) of the rows of ``x`` (the producer's ``want_stats``)."""
    if out is None:
        out = torch.empty_like(x)
    _k_ffn_geglu(x, ln_stats, w1, w1x, w2p, bias2, out)
    return out


def _linear320_impl(x, w, wx, ln_stats, alpha, residual, out, stats_out, eps, vt, hw):
    lib = load()
    M, N = x.shape[0], w.shape[0]
    if x.shape[1] != 320 or w.shape[1] != 320 or tuple(wx.shape) != (N, 16) or not (w.is_contiguous() and wx.is_contiguous()):
        raise SdvHipError(f"linear320: x [M, 320], w [N, 320], wx [N, 16] contiguous expected, got {tuple(x.shape)} / {tuple(w.shape)} / {tuple(wx.shape)}")
    if alpha is not None and alpha.numel() != N // 320:
        raise SdvHipError("linear320: alpha holds one factor per block of 320 output columns")
    args = (_ptr(x, BF16, "X"), M, x.stride(0), _ptr(w, BF16, "W"), _ptr(wx, BF16, "Wx"), N, _ptr(ln_stats, F32, "ln_stats"), _ptr(alpha, F32, "alpha"),
            _ptr(residual, BF16, "R"), residual.stride(0) if residual is not None else 0, _ptr(out, BF16, "out"), out.stride(0),
            _ptr(stats_out, F32, "stats_out"), float(eps), _ptr(vt, BF16, "Vt"), vt.stride(1) if vt is not None else 0, int(hw))
    if vt is not None and (vt.dim() != 3 or vt.shape[0] * hw != M or vt.shape[1] != 320 or vt.stride(2) != 1 or vt.stride(0) != 320 * vt.stride(1)):
        raise SdvHipError(f"linear320: vt must be [M / hw, 320, ldv] with unit token stride, got {tuple(vt.shape)} / strides {vt.stride()}")
    nbytes = 2.0 * M * (320 + N + (320 if residual is not None else 0))
    _launch("linear320", dict(M=M, N=N, K=320, flops=2.0 * M * N * 320, bytes=nbytes),
            lambda: _check(lib.sdv_linear320_bf16(*args, _stream()), "sdv_linear320_bf16"))


_k_linear320 = _defop("k_linear320(Tensor x, Tensor w, Tensor wx, Tensor? ln_stats, Tensor? alpha, Tensor? residual, Tensor(a!) out, Tensor(b!)? stats_out, "
                      "float eps, Tensor(c!)? vt, int hw) -> ()", _linear320_impl)
def _linear640_impl(x, w, wx, ln_stats, alpha, out, stats_out, eps):
    lib = load()
    M, N = x.shape[0], w.shape[0]
    if x.shape[1] != 640 or w.shape[1] != 640 or tuple(wx.shape) != (N, 16) or not (w.is_contiguous() and wx.is_contiguous()):
        raise SdvHipError(f"linear640: x [M, 640], w [N, 640], wx [N, 16] contiguous expected, got {tuple(x.shape)} / {tuple(w.shape)} / {tuple(wx.shape)}")
    if alpha is not None and alpha.numel() != N // 320:
        raise SdvHipError("linear640: alpha holds one factor per block of 320 output columns")
    args = (_ptr(x, BF16, "X"), M, x.stride(0), _ptr(w, BF16, "W"), _ptr(wx, BF16, "Wx"), N, _ptr(ln_stats, F32, "ln_stats"), _ptr(alpha, F32, "alpha"),
            _ptr(out, BF16, "out"), out.stride(0), _ptr(stats_out, F32, "stats_out"), float(eps))
    _launch("linear320", dict(M=M, N=N, K=640, flops=2.0 * M * N * 640, bytes=2.0 * M * (640 + N)),
            lambda: _check(lib.sdv_linear640_bf16(*args, _stream()), "sdv_linear640_bf16"))


_k_linear640 = _defop("k_linear640(Tensor x, Tensor w, Tensor wx, Tensor? ln_stats, Tensor? alpha, Tensor(a!) out, Tensor(b!)? stats_out, float eps) -> ()",
                      _linear640_impl)
LINEAR640 = os.environ.get("SDV_LINEAR640", "1") != "0"     # A/B knob: 0 = the residual-free C = 640 projections stay on the igemm tiles
# Rows (of the WHOLE call: samples x pixels of the level, so that the CFG-shared prefix and the unshared forward choose alike) below which
# a persistent panel kernel leaves most CUs idle and the igemm's small tiles win (tools/r6_small.sh, profiles/round6_small_batches.txt):
# the fused feed-forward, the C = 640 fused Q K V projection, the other C = 640 projections.  hip.FORCE_TILE - "run what the big batch
# runs" (bench.py's parity check, the forced-tile tests) - overrides them: a forced forward takes the panel kernels at any size.
PANEL_MIN_ROWS_FFN = int(os.environ.get("SDV_PANEL_MIN_ROWS_FFN", "16384"))
PANEL_MIN_ROWS_QKV640 = int(os.environ.get("SDV_PANEL_MIN_ROWS_QKV640", "32768"))
PANEL_MIN_ROWS_LIN640 = int(os.environ.get("SDV_PANEL_MIN_ROWS_LIN640", "8192"))
LINEAR320 = os.environ.get("SDV_LINEAR320", "1") != "0"     # A/B knob: 0 = the C = 320 projections on the igemm tiles (round 5)
QKV_VT = os.environ.get("SDV_QKV_VT", "1") != "0"           # A/B knob: 0 = V stays row-major in the fused [Q | K | V] buffer


def linear320(x: torch.Tensor, w: torch.Tensor, wx: torch.Tensor, *, ln_stats=None, alpha=None, residual=None, out=None, want_stats: bool = False,
              eps: float = 1e-5, stats_out=None, vt=None, hw: int = 0):
    """A C = 320 (or, residual-free, C = 640: ``k_linear640`` -> sdv_linear640_bf16, N = 640 / 1920) projection on the panel kernel
    (``torch.ops.sdv.k_linear320`` -> sdv_linear320_bf16): N = 320 or 960 output
    columns, bias / LayerNorm fold in ``wx`` (``weights.ffn_fold_columns(s, t)``), optional per-320-column ``alpha``, residual and the
    LayerNorm statistics of the stored rows (returned as ``(y, stats [M, 2])`` with ``want_stats``).  ``vt`` [M / hw, 320, ldv]
    (N = 960): the V third is stored transposed per sample of ``hw`` tokens and ``out`` [M, 640] holds only [Q | K]."""
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, w.shape[0] if vt is None else 640), dtype=BF16, device=x.device)
    st = (stats_out if stats_out is not None else torch.empty((M, 2), dtype=F32, device=x.device)) if want_stats else None
    if w.shape[1] == 640:      # the C = 640 form (sdv_linear640_bf16): N = 640 or 1920, no residual, no transposed V
        if residual is not None or vt is not None:
            raise SdvHipError("linear320: the K = 640 form takes neither a residual nor a transposed V output")
        _k_linear640(x, w, wx, ln_stats, alpha, out, st, float(eps))
    else:
        _k_linear320(x, w, wx, ln_stats, alpha, residual, out, st, float(eps), vt, int(hw))
    return (out, st) if want_stats else out


_fp8_sat_ref: Optional[torch.Tensor] = None


def _fp8_saturation_counter_impl(counter: Optional[torch.Tensor]):
    if counter is not None and (counter.dtype != torch.int32 or not counter.is_cuda or counter.numel() != 1):
        raise SdvHipError("fp8 saturation counter: one int32 element in GPU memory")
    _check(load().sdv_groupnorm_fp8_set_saturation_counter(counter.data_ptr() if counter is not None else None),
           "sdv_groupnorm_fp8_set_saturation_counter")
    global _fp8_sat_ref
    _fp8_sat_ref = counter      # the library holds a raw device pointer: keep the tensor alive for as long as it is registered


def set_fp8_saturation_counter(counter: Optional[torch.Tensor]):
    """Debug aid of the fp8 path: while a one-element int32 device tensor is registered, every e4m3 GroupNorm apply adds the number
    of elements it had to clamp at +-448 to it (None switches it off).  Register it before a step is captured into a hipGraph."""
    _fp8_saturation_counter_impl(counter)


def q_prescale(dh: int) -> float:
    """softmax scale * log2(e): what the attention kernels want Q multiplied by (they exponentiate in base 2)."""
    return dh ** -0.5 * 1.4426950408889634


def _softmax_rows_impl(s, rows, cols, ld):
    lib = load()
    _check(lib.sdv_softmax_rows_bf16(_ptr(s, BF16, "S"), rows, cols, ld, _stream()), "sdv_softmax_rows_bf16")


_k_softmax_rows = _defop("k_softmax_rows_(Tensor(a!) s, int rows, int cols, int ld) -> ()", _softmax_rows_impl)


def softmax_rows_(s: torch.Tensor, rows: int, cols: int, ld: int):
    _k_softmax_rows(s, rows, cols, ld)


def _softmax_rows_f32_impl(s, p, rows, cols, lds, ldp):
    lib = load()
    _launch("softmax_rows", dict(bytes=(2 * 4.0 + 2.0) * rows * cols),
            lambda: _check(lib.sdv_softmax_rows_f32(_ptr(s, F32, "S"), _ptr(p, BF16, "P"), rows, cols, lds, ldp, _stream()),
                           "sdv_softmax_rows_f32"))


_k_softmax_rows_f32 = _defop("k_softmax_rows_f32(Tensor s, Tensor(a!) p, int rows, int cols, int lds, int ldp) -> ()", _softmax_rows_f32_impl)


def softmax_rows_f32(s: torch.Tensor, p: torch.Tensor, rows: int, cols: int, lds: int, ldp: int):
    """bf16 probabilities ``p`` = softmax over each fp32 row of ``s`` (``torch.ops.sdv.k_softmax_rows_f32``)."""
    _k_softmax_rows_f32(s, p, rows, cols, lds, ldp)


def gn_splits(HW: int) -> int:
    """Pixel ranges the statistics pass cuts an image into.  The apply pass re-reduces the partials in every block, so it
    wants few of them, the statistics pass wants enough blocks to fill the chip: 256 pixels per split at the 64 x 64 level
    and above, 128 at 32 x 32, 64 below (tools/gn_bench.py at 128 samples: apply 152 -> 114 us on 64^2 x 320 channels)."""
    pix = 256 if HW >= 4096 else (128 if HW >= 1024 else 64)
    return max(1, min(64, HW // pix))


def _groupnorm_impl(x, gamma, beta, x2, out, p1, p2, ints, eps, fp8_scale):
    nimg, HW, groups, silu, bpi1, nrep1, rs1, bpi2, nrep2, rs2 = ints
    lib = load()
    C1 = x.shape[1]
    C2 = x2.shape[1] if x2 is not None else 0
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise SdvHipError("groupnorm: inputs must be contiguous")
    fp8 = fp8_scale > 0.0
    nbytes = 2.0 * nimg * HW * (C1 + C2)
    if p1 is not None:
        # statistics from the producers' epilogues (sdv_gemm_args.gn_out): only their per-block partials are read
        splits = max(1, min(64, (bpi1 * nrep1) // 16))          # 16 of an image's 32-row blocks per finalize workgroup
        partials = torch.empty((nimg, splits, groups, 2), dtype=F32, device=x.device)
        pp = _ptr(partials)
        q1, q2 = _ptr(p1, F32, "gn partials"), _ptr(p2, F32, "gn partials 2")
        ld1, ld2 = p1.shape[-1], (p2.shape[-1] if p2 is not None else 0)
        fbytes = 4.0 * (p1.numel() + (p2.numel() if p2 is not None else 0))
        _launch("gn_finalize", dict(bytes=fbytes),
                lambda: _check(lib.sdv_groupnorm_finalize(q1, C1, ld1, bpi1, nrep1, rs1, q2, C2, ld2, bpi2, nrep2, rs2, nimg, groups, splits,
                                                          pp, _stream()), "sdv_groupnorm_finalize"))
    else:
        splits = gn_splits(HW)
        partials = torch.empty((nimg, splits, groups, 2), dtype=F32, device=x.device)
        pp = _ptr(partials)
    xp, x2p, op = _ptr(x, BF16, "X"), _ptr(x2, BF16, "X2"), _ptr(out, FP8 if fp8 else BF16, "Y")
    gp, bp = _ptr(gamma, F32, "gamma"), _ptr(beta, F32, "beta")
    if p1 is None:
        _launch("gn_stats", dict(bytes=nbytes),
                lambda: _check(lib.sdv_groupnorm_stats(xp, x2p, C1, C2, nimg, HW, groups, splits, pp, _stream()),
                               "sdv_groupnorm_stats"))
    if fp8:
        _launch("gn_apply", dict(bytes=1.5 * nbytes),
                lambda: _check(lib.sdv_groupnorm_apply_fp8(xp, x2p, C1, C2, nimg, HW, groups, splits, pp, gp, bp, eps, int(silu),
                                                           op, 1.0 / fp8_scale, _stream()), "sdv_groupnorm_apply_fp8"))
        return
    _launch("gn_apply", dict(bytes=2 * nbytes),
            lambda: _check(lib.sdv_groupnorm_apply(xp, x2p, C1, C2, nimg, HW, groups, splits, pp, gp, bp, eps, int(silu),
                                                   op, _stream()), "sdv_groupnorm_apply"))


_k_groupnorm = _defop("k_groupnorm(Tensor x, Tensor gamma, Tensor beta, Tensor? x2, Tensor(a!) out, Tensor? gn1, Tensor? gn2, int[] ints, "
                      "float eps, float fp8_scale) -> ()", _groupnorm_impl)


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, nimg: int, HW: int, groups: int,
              eps: float, silu: bool, x2: Optional[torch.Tensor] = None, out=None, fp8_scale: Optional[float] = None) -> torch.Tensor:
    """GroupNorm(+SiLU) over NHWC [nimg*HW, C1] (++ [.., C2] concatenated on channels) -> bf16 [.., C1+C2]
    (``torch.ops.sdv.k_groupnorm``: statistics pass + apply pass).  When ``x`` (and ``x2``) carry the statistics their producer's
    epilogue emitted (``tensor._sdv_gn``, see ``gemm(gn_hw=)``) and those describe exactly this tensor, the statistics pass is
    replaced by ``sdv_groupnorm_finalize`` over the per-block partials.
    ``fp8_scale`` s: the result is written as OCP e4m3 bytes q = sat(y / s) instead (y ~ q * s), the fp8 conv's operand."""
    C1 = x.shape[1]
    C2 = x2.shape[1] if x2 is not None else 0
    if out is None:
        out = torch.empty((nimg * HW, C1 + C2), dtype=BF16 if fp8_scale is None else FP8, device=x.device)
    g1 = getattr(x, "_sdv_gn", None) if GN_EPILOGUE else None
    g2 = getattr(x2, "_sdv_gn", None) if (x2 is not None and GN_EPILOGUE) else None

    def fits(g, t):
        return (g is not None and g.nimg == nimg and g.HW == HW and g.C == t.shape[1] and t.shape[0] == nimg * HW
                and g.ver == _ver(t))

    geo = [0, 1, 0, 0, 1, 0]
    p1 = p2 = None
    if fits(g1, x) and (x2 is None or fits(g2, x2)):
        p1, geo[0:3] = g1.p, [g1.bpi, g1.nrep, g1.rep_stride]
        if x2 is not None:
            p2, geo[3:6] = g2.p, [g2.bpi, g2.nrep, g2.rep_stride]
    _k_groupnorm(x, gamma, beta, x2, out, p1, p2, [nimg, HW, groups, int(silu)] + geo, float(eps),
                 float(fp8_scale) if fp8_scale is not None else 0.0)
    return out


def _layernorm_impl(x, gamma, beta, eps, out):
    lib = load()
    rows, Cn = x.shape
    if not x.is_contiguous():
        raise SdvHipError("layernorm: input must be contiguous")
    xp, gp, bp, op = _ptr(x, BF16, "X"), _ptr(gamma, F32), _ptr(beta, F32), _ptr(out, BF16)
    _launch("layernorm", dict(bytes=4.0 * rows * Cn),
            lambda: _check(lib.sdv_layernorm_bf16(xp, gp, bp, eps, rows, Cn, op, _stream()), "sdv_layernorm_bf16"))


_k_layernorm = _defop("k_layernorm(Tensor x, Tensor gamma, Tensor beta, float eps, Tensor(a!) out) -> ()", _layernorm_impl)


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, out=None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(x)
    _k_layernorm(x, gamma, beta, float(eps), out)
    return out


# ------------------------------------------------------------------------------------------------
# small convs / latents
# ------------------------------------------------------------------------------------------------
def _conv3x3_cin_small_impl(x, w, bias, out, nimg, H, W, circular):
    lib = load()
    Cin, Cout = x.shape[1], w.shape[0]
    xp, wp, bp, op = _ptr(x, BF16, "X"), _ptr(w, BF16, "W"), _ptr(bias, F32), _ptr(out, BF16)
    _launch("conv_cin_small", dict(flops=18.0 * nimg * H * W * Cin * Cout, bytes=2.0 * nimg * H * W * (Cin + Cout)),
            lambda: _check(lib.sdv_conv3x3_cin_small(xp, wp, bp, op, nimg, H, W, Cin, Cout, int(circular), _stream()),
                           "sdv_conv3x3_cin_small"))


_k_conv3x3_cin_small = _defop("k_conv3x3_cin_small(Tensor x, Tensor w, Tensor? bias, Tensor(a!) out, int nimg, int H, int W, "
                              "bool circular) -> ()", _conv3x3_cin_small_impl)


def conv3x3_cin_small(x, w, bias, *, nimg, H, W, circular=False, out=None):
    if out is None:
        out = torch.empty((nimg * H * W, w.shape[0]), dtype=BF16, device=x.device)
    _k_conv3x3_cin_small(x, w, bias, out, nimg, H, W, bool(circular))
    return out


def _im2col3x3_c4_impl(x, cols, nimg, H, W, circular):
    lib = load()
    xp, cp = _ptr(x, BF16, "X"), _ptr(cols, BF16)
    _launch("im2col_c4", dict(bytes=2.0 * nimg * H * W * (4 + 64)),
            lambda: _check(lib.sdv_im2col3x3_c4(xp, cp, nimg, H, W, int(circular), _stream()), "sdv_im2col3x3_c4"))


_k_im2col3x3_c4 = _defop("k_im2col3x3_c4(Tensor x, Tensor(a!) cols, int nimg, int H, int W, bool circular) -> ()", _im2col3x3_c4_impl)


def conv3x3_c4(x, w_pad, bias, *, nimg, H, W, circular=False, out=None, gn: bool = False):
    """3x3 pad-1 conv of a 4-channel NHWC tensor on the matrix cores: im2col to 64-wide rows + K = 64 GEMM.
    ``w_pad``: [Cout, 64] = OHWI weights [Cout, 36] zero-padded (see ``weights.conv_w_c4``)."""
    if x.shape[1] != 4:
        raise SdvHipError(f"conv3x3_c4: expected 4 input channels, got {x.shape[1]}")
    cols = torch.empty((nimg * H * W, 64), dtype=BF16, device=x.device)
    _k_im2col3x3_c4(x, cols, nimg, H, W, bool(circular))
    return linear(cols, w_pad, bias, out=out, gn_hw=H * W if gn else 0)


def embed_tokens(ids: torch.Tensor, tok: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """CLIP input embeddings: ids int64 [B, L] -> bf16 [B*L, D] = tok[ids] + pos[position]."""
    if ids.dtype != torch.int64 or ids.ndim != 2:
        raise SdvHipError("embed_tokens: ids must be int64 [B, L]")
    B, L = ids.shape
    V, D = tok.shape
    if L > pos.shape[0] or pos.shape[1] != D:
        raise SdvHipError(f"embed_tokens: {L} positions / width {D} do not fit the position table {tuple(pos.shape)}")
    return _k_embed_tokens(ids.contiguous(), tok, pos)


def _embed_tokens_impl(ids, tok, pos):
    lib = load()
    B, L = ids.shape
    V, D = tok.shape
    out = torch.empty((B * L, D), dtype=BF16, device=ids.device)
    _check(lib.sdv_embed_tokens(_ptr(ids, torch.int64, "ids"), _ptr(tok, F32, "tok"), _ptr(pos, F32, "pos"), _ptr(out, BF16),
                                B * L, L, D, V, _stream()), "sdv_embed_tokens")
    return out


_k_embed_tokens = _defop("k_embed_tokens(Tensor ids, Tensor tok, Tensor pos) -> Tensor", _embed_tokens_impl,
                         lambda ids, tok, pos: tok.new_empty((ids.shape[0] * ids.shape[1], tok.shape[1]), dtype=BF16))


def rgb_u8_to_bf16_c4(img_u8: torch.Tensor, scale: float = 1.0 / 255.0) -> torch.Tensor:
    """uint8 RGB NHWC [..., 3] -> bf16 rows [npix, 4] = {r, g, b, 0} * scale."""
    if img_u8.shape[-1] != 3 or not img_u8.is_contiguous():
        raise SdvHipError("rgb_u8_to_bf16_c4: expected a contiguous [..., 3] uint8 tensor")
    return _k_rgb_u8_to_bf16_c4(img_u8, float(scale))


def _rgb_u8_to_bf16_c4_impl(img_u8, scale):
    lib = load()
    npix = img_u8.numel() // 3
    out = torch.empty((npix, 4), dtype=BF16, device=img_u8.device)
    _check(lib.sdv_rgb_u8_to_bf16_c4(_ptr(img_u8, torch.uint8, "img"), _ptr(out, BF16), npix, scale, _stream()),
           "sdv_rgb_u8_to_bf16_c4")
    return out


_k_rgb_u8_to_bf16_c4 = _defop("k_rgb_u8_to_bf16_c4(Tensor img_u8, float scale) -> Tensor", _rgb_u8_to_bf16_c4_impl,
                              lambda img, scale: img.new_empty((img.numel() // 3, 4), dtype=BF16))


def axpby(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, alpha: float, beta: float):
    """out = alpha*a + beta*b over [rows, cols] bf16 (column slices of wider buffers allowed)."""
    if b.shape != a.shape or out.shape != a.shape:
        raise SdvHipError("axpby: shape mismatch")
    for t in (a, b, out):
        if t.stride(1) != 1:
            raise SdvHipError("axpby: rows must be contiguous")
    _k_axpby(a, b, out, float(alpha), float(beta))


def _axpby_impl(a, b, out, alpha, beta):
    lib = load()
    rows, cols = a.shape
    _launch("axpby", dict(bytes=6.0 * rows * cols),
            lambda: _check(lib.sdv_axpby_bf16(_ptr(a, BF16, "a"), a.stride(0), _ptr(b, BF16, "b"), b.stride(0),
                                              _ptr(out, BF16, "out"), out.stride(0), rows, cols, alpha, beta, _stream()),
                           "sdv_axpby_bf16"))


_k_axpby = _defop("k_axpby(Tensor a, Tensor b, Tensor(a!) out, float alpha, float beta) -> ()", _axpby_impl)




def _latent_affine_impl(x, wpq, bias, in_scale, out, npix, Cn):
    lib = load()
    _check(lib.sdv_latent_affine(_ptr(x, F32), _ptr(wpq, F32), _ptr(bias, F32), in_scale, _ptr(out, BF16), npix, Cn,
                                 _stream()), "sdv_latent_affine")


_k_latent_affine = _defop("k_latent_affine(Tensor x, Tensor wpq, Tensor bias, float in_scale, Tensor(a!) out, int npix, int Cn) -> ()",
                          _latent_affine_impl)


def latent_affine(x, wpq, bias, in_scale, out, npix, Cn):
    _k_latent_affine(x, wpq, bias, float(in_scale), out, npix, Cn)


# ------------------------------------------------------------------------------------------------
# interpolation / scheduler step
# ------------------------------------------------------------------------------------------------
def _slerp_stats_impl(v0, v1):
    lib = load()
    stats = torch.empty(3, dtype=torch.float64, device=v0.device)
    _check(lib.sdv_slerp_stats(_ptr(v0, F32, "v0"), _ptr(v1, F32, "v1"), v0.numel(), _ptr(stats), _stream()),
           "sdv_slerp_stats")
    return stats


_k_slerp_stats = _defop("k_slerp_stats(Tensor v0, Tensor v1) -> Tensor", _slerp_stats_impl,
                        lambda v0, v1: v0.new_empty((3,), dtype=torch.float64))


def slerp_stats(v0, v1) -> torch.Tensor:
    return _k_slerp_stats(v0, v1)


def _slerp_batch_impl(v0, v1, stats, T, out, C_, HW, to_hwc, dot_threshold):
    lib = load()
    _check(lib.sdv_slerp_batch(_ptr(v0, F32), _ptr(v1, F32), _ptr(stats, torch.float64), _ptr(T, F32), T.numel(), C_, HW,
                               int(to_hwc), dot_threshold, _ptr(out, F32), _stream()), "sdv_slerp_batch")


_k_slerp_batch = _defop("k_slerp_batch(Tensor v0, Tensor v1, Tensor stats, Tensor T, Tensor(a!) out, int C_, int HW, bool to_hwc, "
                        "float dot_threshold) -> ()", _slerp_batch_impl)


def slerp_batch(v0, v1, stats, T, *, C_: int, HW: int, to_hwc: bool, dot_threshold: float = 0.9995, out=None):
    if out is None:
        out = torch.empty((T.numel(), C_ * HW), dtype=F32, device=v0.device)
    _k_slerp_batch(v0, v1, stats, T, out, C_, HW, bool(to_hwc), float(dot_threshold))
    return out


def _lerp_batch_impl(a, b, T, out_f32, out_bf16):
    lib = load()
    _check(lib.sdv_lerp_batch(_ptr(a, F32), _ptr(b, F32), _ptr(T, F32), T.numel(), a.numel(), _ptr(out_f32, F32),
                              _ptr(out_bf16, BF16), _stream()), "sdv_lerp_batch")


_k_lerp_batch = _defop("k_lerp_batch(Tensor a, Tensor b, Tensor T, Tensor(a!)? out_f32, Tensor(b!)? out_bf16) -> ()", _lerp_batch_impl)


def lerp_batch(a, b, T, *, out_f32=None, out_bf16=None):
    _k_lerp_batch(a, b, T, out_f32, out_bf16)


def _cfg_ddim_step_impl(eps, latents, x2, coefs, step_ptr, noise, guidance, cfg, n):
    lib = load()
    _check(lib.sdv_cfg_ddim_step(_ptr(eps, F32), _ptr(latents, F32), _ptr(x2, BF16), _ptr(coefs, F32),
                                 _ptr(step_ptr, torch.int32), _ptr(noise, F32), guidance, int(cfg), n, _stream()),
           "sdv_cfg_ddim_step")


_k_cfg_ddim_step = _defop("k_cfg_ddim_step(Tensor eps, Tensor(a!) latents, Tensor(b!) x2, Tensor coefs, Tensor? step_ptr, Tensor? noise, "
                          "float guidance, bool cfg, int n) -> ()", _cfg_ddim_step_impl)


def cfg_ddim_step(eps, latents, x2, coefs, step_ptr, noise, guidance: float, cfg: bool, n: int):
    _k_cfg_ddim_step(eps, latents, x2, coefs, step_ptr, noise, float(guidance), bool(cfg), n)


def _cfg_multistep_step_impl(eps, latents, x2, hist, xsave, table, step_ptr, noise, guidance, cfg, n):
    lib = load()
    _check(lib.sdv_cfg_multistep_step(_ptr(eps, F32), _ptr(latents, F32), _ptr(x2, BF16), _ptr(hist, F32), _ptr(xsave, F32),
                                      _ptr(table, F32), _ptr(step_ptr, torch.int32), _ptr(noise, F32), guidance, int(cfg), n,
                                      _stream()), "sdv_cfg_multistep_step")


_k_cfg_multistep_step = _defop("k_cfg_multistep_step(Tensor eps, Tensor(a!) latents, Tensor(b!) x2, Tensor(c!) hist, Tensor(d!) xsave, "
                               "Tensor table, Tensor? step_ptr, Tensor? noise, float guidance, bool cfg, int n) -> ()",
                               _cfg_multistep_step_impl)


def cfg_multistep_step(eps, latents, x2, hist, xsave, table, step_ptr, noise, guidance: float, cfg: bool, n: int):
    """One fused guidance + linear-multistep scheduler update (sdv_hip.h: every scheduler besides DDIM)."""
    _k_cfg_multistep_step(eps, latents, x2, hist, xsave, table, step_ptr, noise, float(guidance), bool(cfg), n)


def _latents_to_unet_input_impl(latents, x2, cfg, n):
    lib = load()
    _check(lib.sdv_latents_to_unet_input(_ptr(latents, F32), _ptr(x2, BF16), int(cfg), n, _stream()),
           "sdv_latents_to_unet_input")


_k_latents_to_unet_input = _defop("k_latents_to_unet_input(Tensor latents, Tensor(a!) x2, bool cfg, int n) -> ()",
                                  _latents_to_unet_input_impl)


def latents_to_unet_input(latents, x2, cfg: bool, n: int):
    _k_latents_to_unet_input(latents, x2, bool(cfg), n)


def _step_counter_add_impl(step_ptr, inc):
    lib = load()
    _check(lib.sdv_step_counter_add(_ptr(step_ptr, torch.int32), inc, _stream()), "sdv_step_counter_add")


_k_step_counter_add = _defop("k_step_counter_add(Tensor(a!) step_ptr, int inc) -> ()", _step_counter_add_impl)


def step_counter_add(step_ptr, inc: int = 1):
    _k_step_counter_add(step_ptr, inc)


def _timestep_embedding_impl(ts, dim, flip, freq_shift):
    lib = load()
    out = torch.empty((ts.numel(), dim), dtype=F32, device=ts.device)
    _check(lib.sdv_timestep_embedding(_ptr(ts, F32), ts.numel(), dim, int(flip), freq_shift, _ptr(out), _stream()),
           "sdv_timestep_embedding")
    return out


_k_timestep_embedding = _defop("k_timestep_embedding(Tensor ts, int dim, bool flip, float freq_shift) -> Tensor", _timestep_embedding_impl,
                               lambda ts, dim, flip, fs: ts.new_empty((ts.numel(), dim), dtype=F32))


def timestep_embedding(ts, dim: int, flip: bool, freq_shift: float):
    return _k_timestep_embedding(ts, dim, bool(flip), float(freq_shift))


def _linear_small_impl(x, w, b, add, silu_in):
    lib = load()
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=F32, device=x.device)
    _check(lib.sdv_linear_small(_ptr(x, F32), _ptr(w, BF16), _ptr(b, F32), _ptr(add, F32), _ptr(out), M, N, K,
                                int(silu_in), _stream()), "sdv_linear_small")
    return out


_k_linear_small = _defop("k_linear_small(Tensor x, Tensor w, Tensor? b, Tensor? add, bool silu_in) -> Tensor", _linear_small_impl,
                         lambda x, w, b, add, silu_in: x.new_empty((x.shape[0], w.shape[0]), dtype=F32))


def linear_small(x, w, b=None, add=None, silu_in=False):
    return _k_linear_small(x, w, b, add, bool(silu_in))


def _permute_f32_impl(x, to_nhwc):
    lib = load()
    if to_nhwc:
        n, c, h, w = x.shape
        out = torch.empty((n, h, w, c), dtype=F32, device=x.device)
        _check(lib.sdv_nchw_to_nhwc_f32(_ptr(x.contiguous(), F32), _ptr(out), n, c, h * w, _stream()), "sdv_nchw_to_nhwc_f32")
    else:
        n, h, w, c = x.shape
        out = torch.empty((n, c, h, w), dtype=F32, device=x.device)
        _check(lib.sdv_nhwc_to_nchw_f32(_ptr(x.contiguous(), F32), _ptr(out), n, c, h * w, _stream()), "sdv_nhwc_to_nchw_f32")
    return out


_k_permute_f32 = _defop("k_permute_f32(Tensor x, bool to_nhwc) -> Tensor", _permute_f32_impl,
                        lambda x, to_nhwc: x.new_empty((x.shape[0], x.shape[2], x.shape[3], x.shape[1]) if to_nhwc else
                                                       (x.shape[0], x.shape[3], x.shape[1], x.shape[2]), dtype=F32))


def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    return _k_permute_f32(x, True)


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    return _k_permute_f32(x, False)


def _f32_to_bf16_impl(x):
    lib = load()
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    _check(lib.sdv_f32_to_bf16(_ptr(x.contiguous(), F32), _ptr(out), x.numel(), _stream()), "sdv_f32_to_bf16")
    return out


_k_f32_to_bf16 = _defop("k_f32_to_bf16(Tensor x) -> Tensor", _f32_to_bf16_impl, lambda x: x.new_empty(x.shape, dtype=BF16))


def f32_to_bf16(x: torch.Tensor) -> torch.Tensor:
    return _k_f32_to_bf16(x)

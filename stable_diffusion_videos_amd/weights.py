"""Parameter tables for the hot-path networks: shapes in the diffusers state-dict schema
(SURVEY.md section 8a), seeded synthetic initialisation (there are no checkpoints offline), loading of
a local diffusers-layout model directory (``SDV_MODEL_DIR`` / ``from_pretrained(<dir>)``), and the
one-time re-layout of the weights for the HIP kernels (OHWI conv filters, fused QK, GEGLU row
interleave, bf16)."""
from __future__ import annotations

import json
import math
from collections import OrderedDict
from pathlib import Path
from typing import Dict, Optional

import torch

from .config import RRDBNetConfig, TextConfig, UNetConfig, VAEConfig

StateDict = Dict[str, torch.Tensor]


# ------------------------------------------------------------------------------------------------
# shape tables (key -> shape), diffusers naming
# ------------------------------------------------------------------------------------------------
def _conv(sd, name, cin, cout, k):
    sd[name + ".weight"] = (cout, cin, k, k)
    sd[name + ".bias"] = (cout,)


def _lin(sd, name, cin, cout, bias=True):
    sd[name + ".weight"] = (cout, cin)
    if bias:
        sd[name + ".bias"] = (cout,)


def _norm(sd, name, c):
    sd[name + ".weight"] = (c,)
    sd[name + ".bias"] = (c,)


def _resnet(sd, p, cin, cout, temb):
    _norm(sd, p + ".norm1", cin)
    _conv(sd, p + ".conv1", cin, cout, 3)
    if temb:
        _lin(sd, p + ".time_emb_proj", temb, cout)
    _norm(sd, p + ".norm2", cout)
    _conv(sd, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(sd, p + ".conv_shortcut", cin, cout, 1)


def _transformer(sd, p, c, ctx, linear_proj):
    _norm(sd, p + ".norm", c)
    if linear_proj:
        _lin(sd, p + ".proj_in", c, c)
    else:
        _conv(sd, p + ".proj_in", c, c, 1)
    b = p + ".transformer_blocks.0"
    _norm(sd, b + ".norm1", c)
    for a, kd in (("attn1", c), ("attn2", ctx)):
        _lin(sd, f"{b}.{a}.to_q", c, c, bias=False)
        _lin(sd, f"{b}.{a}.to_k", kd, c, bias=False)
        _lin(sd, f"{b}.{a}.to_v", kd, c, bias=False)
        _lin(sd, f"{b}.{a}.to_out.0", c, c)
        if a == "attn1":
            _norm(sd, b + ".norm2", c)
    _norm(sd, b + ".norm3", c)
    _lin(sd, b + ".ff.net.0.proj", c, 8 * c)
    _lin(sd, b + ".ff.net.2", 4 * c, c)
    if linear_proj:
        _lin(sd, p + ".proj_out", c, c)
    else:
        _conv(sd, p + ".proj_out", c, c, 1)


def unet_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    sd: "OrderedDict[str, tuple]" = OrderedDict()
    ch = cfg.block_out_channels
    temb = cfg.temb_dim
    ctx = cfg.cross_attention_dim
    lp = cfg.use_linear_projection
    _conv(sd, "conv_in", cfg.in_channels, ch[0], 3)
    _lin(sd, "time_embedding.linear_1", ch[0], temb)
    _lin(sd, "time_embedding.linear_2", temb, temb)
    cout = ch[0]
    for i, typ in enumerate(cfg.down_block_types):
        cin, cout = cout, ch[i]
        for j in range(cfg.layers_per_block):
            _resnet(sd, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, temb)
            if typ.startswith("CrossAttn"):
                _transformer(sd, f"down_blocks.{i}.attentions.{j}", cout, ctx, lp)
        if i != len(ch) - 1:
            _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    c = ch[-1]
    _resnet(sd, "mid_block.resnets.0", c, c, temb)
    _transformer(sd, "mid_block.attentions.0", c, ctx, lp)
    _resnet(sd, "mid_block.resnets.1", c, c, temb)
    rev = list(reversed(ch))
    cout = rev[0]
    for i, typ in enumerate(cfg.up_block_types):
        prev, cout = cout, rev[i]
        cin = rev[min(i + 1, len(ch) - 1)]
        n = cfg.layers_per_block + 1
        for j in range(n):
            skip = cin if j == n - 1 else cout
            rin = prev if j == 0 else cout
            _resnet(sd, f"up_blocks.{i}.resnets.{j}", rin + skip, cout, temb)
            if typ.startswith("CrossAttn"):
                _transformer(sd, f"up_blocks.{i}.attentions.{j}", cout, ctx, lp)
        if i != len(ch) - 1:
            _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    _norm(sd, "conv_norm_out", ch[0])
    _conv(sd, "conv_out", ch[0], cfg.out_channels, 3)
    return sd


def vae_decoder_shapes(cfg: VAEConfig) -> "OrderedDict[str, tuple]":
    sd: "OrderedDict[str, tuple]" = OrderedDict()
    ch = list(reversed(cfg.block_out_channels))
    _conv(sd, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    _conv(sd, "decoder.conv_in", cfg.latent_channels, ch[0], 3)
    _resnet(sd, "decoder.mid_block.resnets.0", ch[0], ch[0], None)
    a = "decoder.mid_block.attentions.0"
    _norm(sd, a + ".group_norm", ch[0])
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(sd, f"{a}.{n}", ch[0], ch[0])
    _resnet(sd, "decoder.mid_block.resnets.1", ch[0], ch[0], None)
    cout = ch[0]
    for i in range(len(ch)):
        cin, cout = cout, ch[i]
        for j in range(cfg.layers_per_block + 1):
            _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, None)
        if i != len(ch) - 1:
            _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    _norm(sd, "decoder.conv_norm_out", ch[-1])
    _conv(sd, "decoder.conv_out", ch[-1], cfg.out_channels, 3)
    return sd


def clip_text_shapes(cfg: TextConfig) -> "OrderedDict[str, tuple]":
    """``transformers.CLIPTextModel`` state-dict schema (5.x key names; older checkpoints prefix ``text_model.``)."""
    sd: "OrderedDict[str, tuple]" = OrderedDict()
    D, I = cfg.hidden_size, cfg.intermediate_size
    sd["embeddings.token_embedding.weight"] = (cfg.vocab_size, D)
    sd["embeddings.position_embedding.weight"] = (cfg.max_position_embeddings, D)
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}"
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            _lin(sd, f"{p}.self_attn.{n}", D, D)
        _norm(sd, f"{p}.layer_norm1", D)
        _lin(sd, f"{p}.mlp.fc1", D, I)
        _lin(sd, f"{p}.mlp.fc2", I, D)
        _norm(sd, f"{p}.layer_norm2", D)
    _norm(sd, "final_layer_norm", D)
    return sd


def load_clip_text(model_dir: Path, shapes) -> StateDict:
    d = Path(model_dir) / "text_encoder"
    path = next((p for p in (d / "model.safetensors", d / "model.fp16.safetensors", d / "pytorch_model.bin") if p.exists()), None)
    if path is None:
        raise FileNotFoundError(f"no text encoder weights under {d}")
    raw = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in _load_file(path).items()}
    missing = [k for k in shapes if k not in raw]
    if missing:
        raise KeyError(f"{path}: missing keys, e.g. {missing[:4]}")
    out: StateDict = OrderedDict()
    for k, shape in shapes.items():
        t = raw[k].float()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{k}: shape {tuple(t.shape)} != expected {shape}")
        out[k] = t
    return out


def rrdbnet_shapes(cfg: RRDBNetConfig) -> "OrderedDict[str, tuple]":
    """basicsr ``RRDBNet`` state-dict schema (the ``params_ema`` / ``params`` dict of RealESRGAN_x4plus.pth)."""
    sd: "OrderedDict[str, tuple]" = OrderedDict()
    nf, g = cfg.num_feat, cfg.num_grow_ch
    _conv(sd, "conv_first", cfg.num_in_ch, nf, 3)
    for i in range(cfg.num_block):
        for r in (1, 2, 3):
            for k in range(1, 6):
                _conv(sd, f"body.{i}.rdb{r}.conv{k}", nf + (k - 1) * g, g if k < 5 else nf, 3)
    for name in ("conv_body", "conv_up1", "conv_up2", "conv_hr"):
        _conv(sd, name, nf, nf, 3)
    _conv(sd, "conv_last", nf, cfg.num_out_ch, 3)
    return sd


def load_rrdbnet(path, shapes) -> StateDict:
    """RealESRGAN_x4plus.pth layout: ``{"params_ema": state_dict}`` (or ``params``, or the bare state dict)."""
    raw = _load_file(Path(path))
    for key in ("params_ema", "params"):
        if key in raw and isinstance(raw[key], dict):
            raw = raw[key]
            break
    missing = [k for k in shapes if k not in raw]
    if missing:
        raise KeyError(f"{path}: missing keys, e.g. {missing[:4]}")
    out: StateDict = OrderedDict()
    for k, shape in shapes.items():
        t = raw[k].float()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{k}: shape {tuple(t.shape)} != expected {shape}")
        out[k] = t
    return out


def count_params(shapes) -> int:
    return sum(math.prod(s) for s in shapes.values())


# ------------------------------------------------------------------------------------------------
# synthetic initialisation (seeded, CPU generator -> identical on every rank / machine)
# ------------------------------------------------------------------------------------------------
def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def synthetic_state_dict(shapes, seed: int = 0, bf16_exact: bool = True) -> StateDict:
    """Variance-preserving random weights: matrices ~ N(0, 1/fan_in), biases ~ N(0, 0.02^2), norm scales
    1 + 0.1 N, norm shifts 0.1 N.  With ``bf16_exact`` every matrix entry is bf16-representable, so
    the bf16 HIP path and an fp32 checker can share bit-identical weights."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out: StateDict = OrderedDict()
    for name, shape in shapes.items():
        is_norm = "norm" in name.split(".")[-2]
        if len(shape) == 1:
            v = torch.randn(shape, generator=g)
            if is_norm and name.endswith(".weight"):
                t = 1.0 + 0.1 * v
            elif is_norm:
                t = 0.1 * v
            else:
                t = 0.02 * v
        else:
            fan_in = math.prod(shape[1:])
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
            if bf16_exact:
                t = bf16_round(t)
        out[name] = t
    return out


# ------------------------------------------------------------------------------------------------
# local checkpoint loading (diffusers directory layout, safetensors or torch .bin)
# ------------------------------------------------------------------------------------------------
_VAE_OLD_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def _load_file(path: Path) -> StateDict:
    if path.suffix == ".safetensors":
        from safetensors.torch import load_file
        return load_file(str(path))
    return torch.load(str(path), map_location="cpu", weights_only=True)


def load_component(model_dir: Path, sub: str, shapes, prefix_filter: Optional[str] = None) -> StateDict:
    d = Path(model_dir) / sub
    cands = [d / "diffusion_pytorch_model.safetensors", d / "diffusion_pytorch_model.fp16.safetensors",
             d / "diffusion_pytorch_model.bin", d / "model.safetensors", d / "pytorch_model.bin"]
    path = next((p for p in cands if p.exists()), None)
    if path is None:
        raise FileNotFoundError(f"no weights under {d}")
    raw = _load_file(path)
    sd: StateDict = OrderedDict()
    for k, v in raw.items():
        for old, new in _VAE_OLD_ATTN.items():
            k = k.replace(f".attentions.0.{old}.", f".attentions.0.{new}.")
        sd[k] = v
    missing = [k for k in shapes if k not in sd]
    if missing:
        raise KeyError(f"{path}: missing keys, e.g. {missing[:4]}")
    out: StateDict = OrderedDict()
    for k, shape in shapes.items():
        t = sd[k].float()
        if tuple(t.shape) != tuple(shape):
            if t.numel() == math.prod(shape):
                t = t.reshape(shape)     # e.g. linear-vs-1x1-conv projections
            else:
                raise ValueError(f"{k}: shape {tuple(t.shape)} != expected {shape}")
        out[k] = t
    return out


# ------------------------------------------------------------------------------------------------
# re-layout for the kernels
# ------------------------------------------------------------------------------------------------
def conv_w(w: torch.Tensor, device) -> torch.Tensor:
    """[Cout, Cin, kh, kw] -> [Cout, kh*kw*Cin] (OHWI, K-contiguous rows) bf16."""
    co = w.shape[0]
    return w.permute(0, 2, 3, 1).reshape(co, -1).contiguous().to(device=device, dtype=torch.bfloat16)


def upconv_phase_w(w: torch.Tensor, device) -> torch.Tensor:
    """[Cout, Cin, 3, 3] filter of ``Upsample2D`` (nearest 2x, then conv3x3 pad 1) -> [4*Cout, 4*Cin] bf16: one 2x2 filter
    per output phase (py, px), phase-major, each OHWI.  Output pixel (2y+py, 2x+px) reads the upsampled rows 2y+py+dy
    (dy = -1, 0, 1) = low-resolution rows floor(.../2): {y-1, y, y} for py = 0 and {y, y, y+1} for py = 1, so the 3x3 taps
    that land on the same low-resolution pixel are summed (in fp32, then rounded to bf16 once): the result equals the 9-tap
    conv on the materialised 2x image up to that one weight rounding, with 4/9 of the multiplies."""
    co, ci = w.shape[:2]
    wf = w.to(torch.float32)
    groups = {0: ([0], [1, 2]), 1: ([0, 1], [2])}          # phase -> 3x3 tap indices summed into 2x2 tap 0 / tap 1
    out = torch.empty((2, 2, co, 2, 2, ci), dtype=torch.float32, device=wf.device)
    for py in (0, 1):
        for px in (0, 1):
            for ty in (0, 1):
                for tx in (0, 1):
                    out[py, px, :, ty, tx, :] = wf[:, :, groups[py][ty], :][:, :, :, groups[px][tx]].sum(dim=(2, 3))
    return out.reshape(4 * co, 4 * ci).contiguous().to(device=device, dtype=torch.bfloat16)


def conv_w_c4(w: torch.Tensor, device) -> torch.Tensor:
    """[Cout, 4, 3, 3] -> [Cout, 64]: OHWI rows (tap-major, 36 values) zero-padded to one 64-wide K tile, the layout
    ``sdv_im2col3x3_c4`` produces for the activations."""
    co = w.shape[0]
    flat = w.permute(0, 2, 3, 1).reshape(co, -1)
    out = torch.zeros((co, 64), dtype=flat.dtype)
    out[:, : flat.shape[1]] = flat
    return out.contiguous().to(device=device, dtype=torch.bfloat16)


def conv_w_kpad(w: torch.Tensor, device, scale: float = 1.0) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> OHWI [Cout, 9*Kp] with Cin zero-padded to Kp = roundup(Cin, 64), the K granularity of
    ``sdv_gemm_bf16`` (RRDB growth convs have Cin = 96 / 160: their padded channels read whatever the dense-block
    buffer holds there - always finite - against zero weights)."""
    co, ci = w.shape[:2]
    kp = (ci + 63) // 64 * 64
    out = torch.zeros((co, 3, 3, kp), dtype=torch.float32)
    out[..., :ci] = w.permute(0, 2, 3, 1) * scale
    return out.reshape(co, 9 * kp).contiguous().to(device=device, dtype=torch.bfloat16)


def lin_w(w: torch.Tensor, device) -> torch.Tensor:
    return w.reshape(w.shape[0], -1).contiguous().to(device=device, dtype=torch.bfloat16)


def vec(v: torch.Tensor, device) -> torch.Tensor:
    return v.contiguous().to(device=device, dtype=torch.float32)


def ln_fold(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: Optional[torch.Tensor], device, scale: float = 1.0):
    """Fold a LayerNorm (gamma, beta) that FEEDS the linear layer y = LN(x) W^T + b into the layer (sdv_hip.h ``ln_side``):

        W' = bf16(gamma o W)            the weights the GEMM multiplies the UN-normalised x with
        s  = rowsum(W')                  (of the rounded W', so that a constant row x = c 1 cancels exactly: x W'^T - mean s = 0)
        t  = scale * (W beta + b)        what the GEMM adds as its bias; ``scale`` = the GEMM's alpha on these rows (pre-scaled Q)

    so that LN(x) W^T + b = rstd (x W'^T - mean s) + t.  Returns (W' bf16 [N, K], s fp32 [N], t fp32 [N]) on ``device``."""
    wf = w.reshape(w.shape[0], -1).to(device=device, dtype=torch.float32)
    g = gamma.to(device=device, dtype=torch.float32)
    b = beta.to(device=device, dtype=torch.float32)
    wp = (wf * g[None, :]).to(torch.bfloat16)
    s = wp.to(torch.float32).sum(dim=1).contiguous()
    t = (wf * b[None, :]).sum(dim=1)
    if bias is not None:
        t = t + bias.to(device=device, dtype=torch.float32)
    return wp.contiguous(), s, (t * scale).contiguous()


def ffn_w2_permute(w: torch.Tensor) -> torch.Tensor:
    """ff.net.2.weight [C, 4C] for the fused feed-forward (sdv_ffn_geglu_bf16): inside every block of 16 hidden channels position
    8 a + 4 b + e holds channel 8 b + 4 a + e.  The GEGLU outputs of a lane sit in the 32x32 MFMA accumulator layout - lane half a
    owns channels 8 b + 4 a + e of a 16-channel tile - and are handed to ff.net.2's MFMA as its B operand as they are, where lane
    half a supplies K positions 8 a .. 8 a + 7: the contraction is unchanged as long as W2 walks K in the same order."""
    n, k = w.shape
    assert k % 16 == 0
    return w.reshape(n, k // 16, 2, 2, 4).permute(0, 1, 3, 2, 4).reshape(n, k).contiguous()


def ffn_fold_columns(s: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """The weight side of the fused feed-forward's fold k-step (sdv_ffn_geglu_bf16 ``W1x`` [N, 16] bf16): the LayerNorm fold's
    per-column terms of ``ln_fold`` - ``s`` (row sums of gamma o W) and ``t`` (W beta + b), fp32 - each split into three bf16
    pieces (h + m + l = the fp32 value to 24 bits) and laid out so that, against the kernel's token vector
    (m_h m_m m_h m_l m_h m_m r_h r_m | r_h r_l r_h r_m 0 0 0 0) with m = -mean, r = 1 / rstd, the twelve products are the six
    leading cross terms of (-mean) s and of t / rstd: the matrix core adds  - mean s + t / rstd  to x W'^T in fp32."""
    def split3(x):
        x = x.detach().to(torch.float32)
        h = x.to(torch.bfloat16)
        m = (x - h.float()).to(torch.bfloat16)
        lo = (x - h.float() - m.float()).to(torch.bfloat16)
        return h, m, lo
    sh, sm, sl = split3(s)
    th, tm, tl = split3(t)
    z = torch.zeros_like(sh)
    return torch.stack([sh, sh, sm, sh, sl, sm, th, th, tm, th, tl, tm, z, z, z, z], dim=1).contiguous()


def geglu_interleave(t: torch.Tensor) -> torch.Tensor:
    """ff.net.0.proj rows are [value(4C) | gate(4C)]; the GEGLU epilogue wants every 32-row MFMA tile to hold
    [16 value rows | the 16 gate rows of the same channels], so that value and gate of a channel meet in the same
    lane (accumulator quads g and g+2 of the 32x32 MFMA C layout)."""
    half = t.shape[0] // 2
    assert half % 16 == 0
    val = t[:half].reshape(half // 16, 16, *t.shape[1:])
    gate = t[half:].reshape(half // 16, 16, *t.shape[1:])
    return torch.stack([val, gate], dim=1).reshape(t.shape).contiguous()

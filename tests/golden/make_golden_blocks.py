#!/usr/bin/env python
"""Generator of tests/golden/blocks_f64.npz: a SECOND, independent restatement of every third-party block the oracle
restates (oracle/models.py), written from the published definitions in plain numpy float64 - explicit loops / einsums, no
torch.nn, so that a slip in one restatement (chunk order of the GEGLU, [cos|sin] order of the time embedding, padding of the
samplers, biased variance of the norms, head split of the attention ...) shows up as a disagreement between the two.

It does NOT pin the oracle to diffusers (diffusers cannot be installed here: parity stays "unpinned" for these blocks) - it
removes single-author slips, and the committed vectors make the oracle's arithmetic a regression-tested artefact.

    python tests/golden/make_golden_blocks.py        # rewrites tests/golden/blocks_f64.npz (seeded, deterministic)

What each function follows (the modules the reference's hot loop executes through diffusers,
/root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:418 and :433):
  timestep_embedding  Vaswani-style sinusoids, exponent -ln(10000) k / (half - freq_shift), [sin|cos] flipped to [cos|sin]
  resnet_block        GroupNorm(32) -> SiLU -> conv3x3 -> + Linear(SiLU(temb)) -> GroupNorm -> SiLU -> conv3x3 -> + shortcut
  attention           softmax(q k^T / sqrt(dh)) v, heads split on the channel axis, bias only on the output projection
  geglu_ff            Linear(C -> 8C) -> value * gelu_erf(gate) (value = FIRST half) -> Linear(4C -> C)
  basic_block         x + attn1(LN x);  x + attn2(LN x, ctx);  x + ff(LN x)
  transformer2d       GroupNorm(32, eps 1e-6) -> 1x1 conv (or Linear) in -> block -> out projection -> + input
  downsample          conv3x3 stride 2, padding 1;    upsample  nearest x2 then conv3x3 padding 1
  vae_attention       GroupNorm -> single-head attention over all channels with biased q/k/v/out -> + input
"""
from __future__ import annotations

import math
from pathlib import Path

import numpy as np
from scipy.special import erf

F64 = np.float64


# ---------------------------------------------------------------------------------------------- primitives
def conv2d(x, w, b, stride=1, pad=1):
    """x [N,C,H,W], w [O,C,kh,kw] -> [N,O,Ho,Wo]: direct sum over the taps."""
    n, c, h, wd = x.shape
    o, _, kh, kw = w.shape
    xp = np.zeros((n, c, h + 2 * pad, wd + 2 * pad), F64)
    xp[:, :, pad:pad + h, pad:pad + wd] = x
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    y = np.zeros((n, o, ho, wo), F64)
    for i in range(kh):
        for j in range(kw):
            win = xp[:, :, i:i + stride * (ho - 1) + 1:stride, j:j + stride * (wo - 1) + 1:stride]
            y += np.einsum("nchw,oc->nohw", win, w[:, :, i, j])
    return y + b[None, :, None, None]


def group_norm(x, groups, gamma, beta, eps):
    n, c, h, w = x.shape
    g = x.reshape(n, groups, (c // groups) * h * w)
    mu = g.mean(axis=2, keepdims=True)
    var = ((g - mu) ** 2).mean(axis=2, keepdims=True)            # biased, as torch.nn.GroupNorm
    y = ((g - mu) / np.sqrt(var + eps)).reshape(n, c, h, w)
    return y * gamma[None, :, None, None] + beta[None, :, None, None]


def layer_norm(x, gamma, beta, eps=1e-5):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def silu(x):
    return x / (1.0 + np.exp(-x))


def gelu_erf(x):
    return 0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))


def linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


def softmax(s):
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(axis=-1, keepdims=True)


# ---------------------------------------------------------------------------------------------- blocks
def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000.0):
    half = dim // 2
    out = np.zeros((len(t), dim), F64)
    for r, tv in enumerate(t):
        for k in range(half):
            ang = float(tv) * math.exp(-math.log(max_period) * k / (half - freq_shift))
            s, c = math.sin(ang), math.cos(ang)
            if flip_sin_to_cos:
                out[r, k], out[r, half + k] = c, s
            else:
                out[r, k], out[r, half + k] = s, c
    return out


def time_mlp(e, p):
    return linear(silu(linear(e, p["linear_1.weight"], p["linear_1.bias"])), p["linear_2.weight"], p["linear_2.bias"])


def resnet_block(x, temb, p, groups, eps):
    h = conv2d(silu(group_norm(x, groups, p["norm1.weight"], p["norm1.bias"], eps)), p["conv1.weight"], p["conv1.bias"])
    if temb is not None:
        h = h + linear(silu(temb), p["time_emb_proj.weight"], p["time_emb_proj.bias"])[:, :, None, None]
    h = conv2d(silu(group_norm(h, groups, p["norm2.weight"], p["norm2.bias"], eps)), p["conv2.weight"], p["conv2.bias"])
    sc = conv2d(x, p["conv_shortcut.weight"], p["conv_shortcut.bias"], pad=0) if "conv_shortcut.weight" in p else x
    return sc + h


def attention(x, ctx, p, prefix, heads):
    """x [B,L,C], ctx [B,Lc,D]"""
    q = linear(x, p[prefix + "to_q.weight"])
    k = linear(ctx, p[prefix + "to_k.weight"])
    v = linear(ctx, p[prefix + "to_v.weight"])
    b, l, c = q.shape
    dh = c // heads
    out = np.zeros_like(q)
    for bi in range(b):
        for hd in range(heads):
            sl = slice(hd * dh, (hd + 1) * dh)
            pr = softmax(q[bi, :, sl] @ k[bi, :, sl].T / math.sqrt(dh))
            out[bi, :, sl] = pr @ v[bi, :, sl]
    return linear(out, p[prefix + "to_out.0.weight"], p[prefix + "to_out.0.bias"])


def geglu_ff(x, p, prefix):
    hcat = linear(x, p[prefix + "net.0.proj.weight"], p[prefix + "net.0.proj.bias"])
    half = hcat.shape[-1] // 2
    value, gate = hcat[..., :half], hcat[..., half:]
    return linear(value * gelu_erf(gate), p[prefix + "net.2.weight"], p[prefix + "net.2.bias"])


def basic_block(x, ctx, p, heads, prefix="transformer_blocks.0."):
    n1 = layer_norm(x, p[prefix + "norm1.weight"], p[prefix + "norm1.bias"])
    x = x + attention(n1, n1, p, prefix + "attn1.", heads)             # self-attention: keys / values from the same tokens
    x = x + attention(layer_norm(x, p[prefix + "norm2.weight"], p[prefix + "norm2.bias"]), ctx, p, prefix + "attn2.", heads)
    x = x + geglu_ff(layer_norm(x, p[prefix + "norm3.weight"], p[prefix + "norm3.bias"]), p, prefix + "ff.")
    return x


def transformer2d(x, ctx, p, heads, groups, linear_proj):
    n, c, h, w = x.shape
    y = group_norm(x, groups, p["norm.weight"], p["norm.bias"], 1e-6)
    if linear_proj:
        t = linear(y.transpose(0, 2, 3, 1).reshape(n, h * w, c), p["proj_in.weight"], p["proj_in.bias"])
    else:
        t = conv2d(y, p["proj_in.weight"], p["proj_in.bias"], pad=0).transpose(0, 2, 3, 1).reshape(n, h * w, c)
    t = basic_block(t, ctx, p, heads)
    if linear_proj:
        o = linear(t, p["proj_out.weight"], p["proj_out.bias"]).reshape(n, h, w, c).transpose(0, 3, 1, 2)
    else:
        o = conv2d(t.reshape(n, h, w, c).transpose(0, 3, 1, 2), p["proj_out.weight"], p["proj_out.bias"], pad=0)
    return o + x


def downsample(x, p):
    return conv2d(x, p["conv.weight"], p["conv.bias"], stride=2, pad=1)


def upsample(x, p):
    return conv2d(np.repeat(np.repeat(x, 2, axis=2), 2, axis=3), p["conv.weight"], p["conv.bias"], pad=1)


def vae_attention(x, p, groups):
    n, c, h, w = x.shape
    t = group_norm(x, groups, p["group_norm.weight"], p["group_norm.bias"], 1e-6).reshape(n, c, h * w).transpose(0, 2, 1)
    q, k, v = (linear(t, p[f"to_{a}.weight"], p[f"to_{a}.bias"]) for a in "qkv")
    o = np.stack([softmax(q[i] @ k[i].T / math.sqrt(c)) @ v[i] for i in range(n)])
    o = linear(o, p["to_out.0.weight"], p["to_out.0.bias"])
    return o.transpose(0, 2, 1).reshape(n, c, h, w) + x


# ---------------------------------------------------------------------------------------------- fixture
def _params(rng, shapes):
    out = {}
    for k, s in shapes.items():
        if len(s) == 1:
            out[k] = (1.0 + 0.2 * rng.standard_normal(s)) if (k.endswith("weight") and "norm" in k) else 0.2 * rng.standard_normal(s)
        else:
            out[k] = rng.standard_normal(s) / math.sqrt(np.prod(s[1:]))
    return out


def build():
    rng = np.random.default_rng(20260921)
    fx = {}

    def put(block, params, **arrays):
        for k, v in params.items():
            fx[f"{block}::p::{k}"] = v
        for k, v in arrays.items():
            fx[f"{block}::{k}"] = np.asarray(v)

    # time embedding: SD values (dim 320, flipped, shift 0) at the first / a middle / the last DDIM timestep + an unflipped row
    t = np.array([981.0, 501.0, 1.0])
    fx["temb::t"] = t
    fx["temb::flip"] = timestep_embedding(t, 320, True, 0.0)
    fx["temb::noflip_shift1"] = timestep_embedding(t, 32, False, 1.0)
    p = _params(rng, {"linear_1.weight": (24, 16), "linear_1.bias": (24,), "linear_2.weight": (24, 24), "linear_2.bias": (24,)})
    e = timestep_embedding(t, 16)
    put("time_mlp", p, x=e, y=time_mlp(e, p))

    # ResnetBlock2D with and without shortcut / time embedding (8 groups, odd spatial size to catch H/W mix-ups)
    for name, cin, cout, temb_c in (("resnet_shortcut", 16, 24, 12), ("resnet_plain", 16, 16, 12), ("resnet_notemb", 16, 16, 0)):
        shapes = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,),
                  "norm2.weight": (cout,), "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
        if temb_c:
            shapes.update({"time_emb_proj.weight": (cout, temb_c), "time_emb_proj.bias": (cout,)})
        if cin != cout:
            shapes.update({"conv_shortcut.weight": (cout, cin, 1, 1), "conv_shortcut.bias": (cout,)})
        p = _params(rng, shapes)
        x = rng.standard_normal((2, cin, 5, 7))
        temb = rng.standard_normal((2, temb_c)) if temb_c else None
        put(name, p, x=x, y=resnet_block(x, temb, p, 8, 1e-5), **({"temb": temb} if temb_c else {}))

    # Transformer2DModel, conv and linear projections, 2 heads, context dim != channels, 5 context tokens
    for name, lin in (("transformer_conv", False), ("transformer_linear", True)):
        c, d = 16, 12
        b = "transformer_blocks.0."
        shapes = {"norm.weight": (c,), "norm.bias": (c,), "proj_in.weight": (c, c) if lin else (c, c, 1, 1), "proj_in.bias": (c,),
                  "proj_out.weight": (c, c) if lin else (c, c, 1, 1), "proj_out.bias": (c,)}
        for i in (1, 2, 3):
            shapes.update({f"{b}norm{i}.weight": (c,), f"{b}norm{i}.bias": (c,)})
        for a, kd in (("attn1", c), ("attn2", d)):
            shapes.update({f"{b}{a}.to_q.weight": (c, c), f"{b}{a}.to_k.weight": (c, kd), f"{b}{a}.to_v.weight": (c, kd),
                           f"{b}{a}.to_out.0.weight": (c, c), f"{b}{a}.to_out.0.bias": (c,)})
        shapes.update({f"{b}ff.net.0.proj.weight": (8 * c, c), f"{b}ff.net.0.proj.bias": (8 * c,),
                       f"{b}ff.net.2.weight": (c, 4 * c), f"{b}ff.net.2.bias": (c,)})
        p = _params(rng, shapes)
        x = rng.standard_normal((2, c, 3, 4))
        ctx = 2.0 * rng.standard_normal((2, 5, d))
        put(name, p, x=x, ctx=ctx, y=transformer2d(x, ctx, p, 2, 4, lin))

    p = _params(rng, {"conv.weight": (6, 6, 3, 3), "conv.bias": (6,)})
    x = rng.standard_normal((2, 6, 6, 8))
    put("downsample", p, x=x, y=downsample(x, p))
    x = rng.standard_normal((2, 6, 5, 7))                      # odd size: the stride-2 output is ceil(n/2)
    put("downsample_odd", p, x=x, y=downsample(x, p))
    x = rng.standard_normal((2, 6, 3, 4))
    put("upsample", p, x=x, y=upsample(x, p))

    c = 16
    p = _params(rng, {"group_norm.weight": (c,), "group_norm.bias": (c,), **{f"to_{a}.weight": (c, c) for a in "qkv"},
                      **{f"to_{a}.bias": (c,) for a in "qkv"}, "to_out.0.weight": (c, c), "to_out.0.bias": (c,)})
    x = 2.0 * rng.standard_normal((2, c, 3, 5))
    put("vae_attention", p, x=x, y=vae_attention(x, p, 4))
    return fx


if __name__ == "__main__":
    out = Path(__file__).resolve().parent / "blocks_f64.npz"
    fx = build()
    np.savez_compressed(out, **fx)
    print(f"wrote {out} ({out.stat().st_size >> 10} KiB, {len(fx)} arrays)")

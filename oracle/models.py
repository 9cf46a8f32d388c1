"""CPU oracle (test infrastructure): PyTorch fp32 restatement of the diffusers modules the
reference hot loop calls.

    unet(x, t, encoder_hidden_states=ctx).sample     stable_diffusion_pipeline.py:418
    vae.decode(z).sample                             stable_diffusion_pipeline.py:433

diffusers itself is an un-vendored, unpinned dependency (``pyproject.toml:14``); the idioms the
reference uses (``unet.in_channels`` :367, ``_optional_components`` :63) date it to roughly
diffusers 0.11-0.14.  The architecture restated here is the published SD-v1 / SD-v2
``UNet2DConditionModel`` and ``AutoencoderKL`` decoder; ``state_dict()`` keys follow the diffusers
schema so a real checkpoint directory can be loaded.  **parity
unpinned** for the UNet (no reference golden vectors exist for these modules); the VAE ``Decoder`` is pinned
against third-party code - the CompVis ``ldm`` Decoder shipped in ``transformers`` (JanusVQVAEDecoder) through the
published ldm -> diffusers weight conversion, tests/test_oracle.py::test_vae_decoder_pinned_against_the_ldm_decoder_in_transformers;
``BasicTransformerBlock`` is pinned against torch's own ``nn.TransformerDecoderLayer(norm_first=True)`` (same file).

Layout here is plain NCHW fp32 and nn.functional ops - deliberately the most literal form, so it
can serve as the checker for the NHWC/bf16 HIP path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
    up_block_types: Tuple[str, ...] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    # SD-1.x: number of heads (8); SD-2.x: per-level list of head counts (5,10,20,20)
    attention_head_dim: Union[int, Tuple[int, ...]] = 8
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False
    flip_sin_to_cos: bool = True
    freq_shift: int = 0

    def heads(self, level: int) -> int:
        a = self.attention_head_dim
        return a if isinstance(a, int) else a[level]


@dataclass
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


def sd14_unet_config() -> UNetConfig:
    return UNetConfig()


def sd21_unet_config() -> UNetConfig:
    return UNetConfig(sample_size=96, cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20),
                      use_linear_projection=True)


def sd_vae_config() -> VAEConfig:
    return VAEConfig()


# --------------------------------------------------------------------------------------
# shared blocks
# --------------------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int, flip_sin_to_cos: bool, freq_shift: float,
                       max_period: int = 10000) -> torch.Tensor:
    """diffusers ``get_timestep_embedding``: [cos | sin] of t * exp(-ln(1e4) k / (half - shift))."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_channels: Optional[int], groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, cout) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Attention(nn.Module):
    """diffusers ``CrossAttention``: softmax(q k^T / sqrt(dh)) v, bias only on to_out."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, qkv_bias=False):
        super().__init__()
        inner = heads * dim_head
        context_dim = context_dim or query_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=qkv_bias)
        self.to_k = nn.Linear(context_dim, inner, bias=qkv_bias)
        self.to_v = nn.Linear(context_dim, inner, bias=qkv_bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim)])

    def forward(self, x, context=None):
        context = x if context is None else context
        b, n, _ = x.shape
        h = self.heads
        q = self.to_q(x).view(b, n, h, -1).transpose(1, 2)
        k = self.to_k(context).view(b, context.shape[1], h, -1).transpose(1, 2)
        v = self.to_v(context).view(b, context.shape[1], h, -1).transpose(1, 2)
        attn = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * self.scale, dim=-1)
        out = torch.matmul(attn, v).transpose(1, 2).reshape(b, n, -1)
        return self.to_out[0](out)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Identity(), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, context_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, context_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, channels, context_dim, groups=32, use_linear_projection=False):
        super().__init__()
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        if use_linear_projection:
            self.proj_in = nn.Linear(channels, channels)
        else:
            self.proj_in = nn.Conv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, heads, dim_head, context_dim)])
        if use_linear_projection:
            self.proj_out = nn.Linear(channels, channels)
        else:
            self.proj_out = nn.Conv2d(channels, channels, 1)

    def forward(self, x, context):
        b, c, h, w = x.shape
        res = x
        x = self.norm(x)
        if not self.use_linear_projection:
            x = self.proj_in(x)
            x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
        else:
            x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
            x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, context)
        if not self.use_linear_projection:
            x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
            x = self.proj_out(x)
        else:
            x = self.proj_out(x)
            x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
        return x + res


# --------------------------------------------------------------------------------------
# UNet2DConditionModel
# --------------------------------------------------------------------------------------
class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, nlayers, groups, eps, attn: Optional[dict], add_down: bool):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(nlayers)])
        if attn is not None:
            self.attentions = nn.ModuleList([Transformer2DModel(channels=cout, **attn) for _ in range(nlayers)])
        else:
            self.attentions = None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, c, temb, groups, eps, attn: dict):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(channels=c, **attn)])

    def forward(self, x, temb, ctx):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ctx)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, temb, nlayers, groups, eps, attn: Optional[dict], add_up: bool):
        super().__init__()
        rs = []
        for i in range(nlayers):
            skip = cin if i == nlayers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(rs)
        if attn is not None:
            self.attentions = nn.ModuleList([Transformer2DModel(channels=cout, **attn) for _ in range(nlayers)])
        else:
            self.attentions = None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips: List[torch.Tensor], temb, ctx):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNet2DConditionModel(nn.Module):
    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        ch = cfg.block_out_channels
        temb = ch[0] * 4
        g, eps = cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)

        def attn_kwargs(level, c):
            heads = cfg.heads(level)
            return dict(heads=heads, dim_head=c // heads, context_dim=cfg.cross_attention_dim, groups=g,
                        use_linear_projection=cfg.use_linear_projection)

        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i, typ in enumerate(cfg.down_block_types):
            cin, cout = cout, ch[i]
            last = i == len(ch) - 1
            self.down_blocks.append(DownBlock(cin, cout, temb, cfg.layers_per_block, g, eps,
                                              attn_kwargs(i, cout) if typ.startswith("CrossAttn") else None,
                                              add_down=not last))
        self.mid_block = MidBlock(ch[-1], temb, g, eps, attn_kwargs(len(ch) - 1, ch[-1]))
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(ch))
        cout = rev[0]
        for i, typ in enumerate(cfg.up_block_types):
            prev, cout = cout, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            last = i == len(ch) - 1
            level = len(ch) - 1 - i
            self.up_blocks.append(UpBlock(cin, cout, prev, temb, cfg.layers_per_block + 1, g, eps,
                                          attn_kwargs(level, cout) if typ.startswith("CrossAttn") else None,
                                          add_up=not last))
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states):
        cfg = self.cfg
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.long)
        timestep = timestep.reshape(-1).expand(sample.shape[0])
        t_emb = timestep_embedding(timestep, cfg.block_out_channels[0], cfg.flip_sin_to_cos, cfg.freq_shift)
        emb = self.time_embedding(t_emb.to(sample.dtype))
        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, encoder_hidden_states)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return x


# --------------------------------------------------------------------------------------
# AutoencoderKL decoder
# --------------------------------------------------------------------------------------
class VAEAttention(nn.Module):
    """diffusers ``AttentionBlock`` (1 head over all channels), keys in the new to_q/to_k naming."""

    def __init__(self, c, groups=32):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        b, c, h, w = x.shape
        res = x
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        attn = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (c ** -0.5), dim=-1)
        t = self.to_out[0](torch.matmul(attn, v))
        return t.transpose(1, 2).reshape(b, c, h, w) + res


class VAEMid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, None, groups, 1e-6) for _ in range(2)])
        self.attentions = nn.ModuleList([VAEAttention(c, groups)])

    def forward(self, x):
        x = self.resnets[0](x)
        x = self.attentions[0](x)
        return self.resnets[1](x)


class VAEUpBlock(nn.Module):
    def __init__(self, cin, cout, nlayers, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, 1e-6) for i in range(nlayers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Decoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        ch = list(reversed(cfg.block_out_channels))
        g = cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, ch[0], 3, padding=1)
        self.mid_block = VAEMid(ch[0], g)
        self.up_blocks = nn.ModuleList()
        cout = ch[0]
        for i in range(len(ch)):
            cin, cout = cout, ch[i]
            self.up_blocks.append(VAEUpBlock(cin, cout, cfg.layers_per_block + 1, g, add_up=i != len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.conv_in(z)
        x = self.mid_block(x)
        for blk in self.up_blocks:
            x = blk(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLDecoder(nn.Module):
    """``AutoencoderKL.decode``: post_quant_conv (1x1) then the decoder."""

    def __init__(self, cfg: VAEConfig):
        super().__init__()
        self.cfg = cfg
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)
        self.decoder = Decoder(cfg)

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    forward = decode


def count_params(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())

"""Model configurations of the networks behind the walk hot path (diffusers ``config.json`` fields
that matter for inference).  Mirrors what ``pipe.unet.config`` / ``pipe.vae.config`` expose to the
reference (stable_diffusion_pipeline.py:158, :268, :367)."""
from __future__ import annotations

import json
from dataclasses import dataclass, field, fields
from pathlib import Path
from typing import Tuple, Union


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
    attention_head_dim: Union[int, Tuple[int, ...]] = 8   # SD-1.x: number of heads; SD-2.x: heads per level
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    use_linear_projection: bool = False
    flip_sin_to_cos: bool = True
    freq_shift: int = 0

    def heads(self, level: int) -> int:
        a = self.attention_head_dim
        return a if isinstance(a, int) else a[level]

    @property
    def temb_dim(self) -> int:
        return self.block_out_channels[0] * 4


@dataclass
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


@dataclass
class RRDBNetConfig:
    """Real-ESRGAN x4plus generator (reference upsampling.py:25)."""
    num_in_ch: int = 3
    num_out_ch: int = 3
    num_feat: int = 64
    num_block: int = 23
    num_grow_ch: int = 32
    scale: int = 4


@dataclass
class TextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    bos_token_id: int = 49406
    eos_token_id: int = 49407


def sd14_unet() -> UNetConfig:
    return UNetConfig()


def sd21_unet() -> UNetConfig:
    return UNetConfig(sample_size=96, cross_attention_dim=1024, attention_head_dim=(5, 10, 20, 20),
                      use_linear_projection=True)


def sd_vae() -> VAEConfig:
    return VAEConfig()


def sd14_text() -> TextConfig:
    return TextConfig()


def sd21_text() -> TextConfig:
    return TextConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                      hidden_act="gelu")


def tiny_unet(cross_attention_dim: int = 64) -> UNetConfig:
    """Small config with the full SD topology (4 levels, cross-attn at 3) for fast tests."""
    return UNetConfig(sample_size=16, block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2),
                      cross_attention_dim=cross_attention_dim)


def tiny_vae() -> VAEConfig:
    return VAEConfig(block_out_channels=(64, 64, 128, 128))


def tiny_text() -> TextConfig:
    return TextConfig(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                      num_attention_heads=1, bos_token_id=998, eos_token_id=999)     # head dim 64, as in both SD text encoders


def _from_json(cls, path: Path):
    data = json.loads(Path(path).read_text())
    names = {f.name for f in fields(cls)}
    kw = {}
    for k, v in data.items():
        if k in names:
            kw[k] = tuple(v) if isinstance(v, list) else v
    return cls(**kw)


def unet_from_json(path) -> UNetConfig:
    return _from_json(UNetConfig, path)


def vae_from_json(path) -> VAEConfig:
    return _from_json(VAEConfig, path)

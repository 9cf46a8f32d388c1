"""Shared builders for the parity tests: an oracle module (CPU fp32) and the matching HIP engine from ONE
seeded, bf16-exact synthetic state dict - so any difference is activation arithmetic, not weights."""
from __future__ import annotations

import torch

from oracle import models as om
from stable_diffusion_videos_amd import config as cfgs
from stable_diffusion_videos_amd import weights


def oracle_unet_cfg(c: cfgs.UNetConfig) -> om.UNetConfig:
    return om.UNetConfig(sample_size=c.sample_size, in_channels=c.in_channels, out_channels=c.out_channels,
                         block_out_channels=tuple(c.block_out_channels), down_block_types=tuple(c.down_block_types),
                         up_block_types=tuple(c.up_block_types), layers_per_block=c.layers_per_block,
                         cross_attention_dim=c.cross_attention_dim, attention_head_dim=c.attention_head_dim,
                         norm_num_groups=c.norm_num_groups, norm_eps=c.norm_eps,
                         use_linear_projection=c.use_linear_projection, flip_sin_to_cos=c.flip_sin_to_cos,
                         freq_shift=c.freq_shift)


def oracle_vae_cfg(c: cfgs.VAEConfig) -> om.VAEConfig:
    return om.VAEConfig(latent_channels=c.latent_channels, out_channels=c.out_channels,
                        block_out_channels=tuple(c.block_out_channels), layers_per_block=c.layers_per_block,
                        norm_num_groups=c.norm_num_groups, scaling_factor=c.scaling_factor)


def _materialise(module_cls, cfg, sd):
    with torch.device("meta"):
        m = module_cls(cfg)
    # proj_in/proj_out are stored as [C, C, 1, 1] convs or [C, C] linears depending on the config
    fixed = {k: v.reshape(m.state_dict()[k].shape) for k, v in sd.items()}
    m.load_state_dict(fixed, assign=True, strict=True)
    return m.eval()


def make_oracle_unet(c: cfgs.UNetConfig, sd):
    return _materialise(om.UNet2DConditionModel, oracle_unet_cfg(c), sd)


def make_oracle_vae(c: cfgs.VAEConfig, sd):
    return _materialise(om.AutoencoderKLDecoder, oracle_vae_cfg(c), sd)


def unet_pair(c: cfgs.UNetConfig, device, seed=0, tiled=False, fp8=False):
    from stable_diffusion_videos_amd.engine import UNetEngine
    sd = weights.synthetic_state_dict(weights.unet_shapes(c), seed=seed)
    return make_oracle_unet(c, sd), UNetEngine(c, sd, device, tiled=tiled, fp8=fp8)


def vae_pair(c: cfgs.VAEConfig, device, seed=1, tiled=False):
    from stable_diffusion_videos_amd.engine import VAEDecoderEngine
    sd = weights.synthetic_state_dict(weights.vae_decoder_shapes(c), seed=seed)
    return make_oracle_vae(c, sd), VAEDecoderEngine(c, sd, device, tiled=tiled)

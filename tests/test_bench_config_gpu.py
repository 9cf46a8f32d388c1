"""Parity at the configuration the headline number is MEASURED on: ``batch_size = 128`` (256 UNet samples per forward, M = 1 048 576-row
GEMMs, a 5.4 GB GEGLU tensor, 8.6 GB of VAE activations at 512 x 512).

The reference pushes whatever ``batch_size`` the caller gives through one ``__call__``
(/root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:472, :538-548), so a frame must not depend on the batch it
was generated in.  The CPU oracle cannot run 128 frames; what can be checked at this size:

* frames {0, 63, 64, 127} of a 128-frame call against the SAME frames generated in a 4-frame call (same embeddings, same noise)
  WITH THE SAME TILES (``hip.FORCE_TILE = 6``: the 4-frame call runs the 256 x 320 tile the 128-frame call picks by itself):
  **bit-identical**.  Every GEMM / conv / attention element sees the same sequence of MFMA k-steps whatever the batch, the
  GroupNorm statistics are split by image size only, attention is per (sample, head), and with the same tile the LayerNorm
  row statistics leave the producer GEMM in the same per-(N tile, wave column) partial sums - so nothing may depend on where
  in a 1 048 576-row tensor a frame sits.  This is the check that the 64-bit addressing of the big batch is right;
* the same comparison with the tiles the cost model picks for 8 samples (128 x 128 at M = 32 768 rows): NOT bit-identical and
  not close to it either - other partial-sum groupings move (mean, rstd) of the LayerNorm rows in their last fp32 bits, that
  flips a bf16 rounding here and there in the first transformer block, and from there on the two runs are two independent
  realisations of the bf16 roundings (DESIGN (c): a difference delta becomes sqrt(delta * ulp) at the next rounding).  Measured:
  uint8 mean-abs 1.9 / max-abs 17 after 2 of 50 steps (the decoded image is still mostly noise), i.e. 40.7 dB between the two
  - the distance each of them has to the fp32 oracle (41.7 dB).  Gate: PSNR >= 37 dB (3 dB under two independent realisations);
* frames 0 and 127 of the 128-frame call against the fp32 CPU oracle, the gate of ``test_sd14_full_size_two_steps`` (>= 39 dB);
* one 128-frame VAE decode against 4-frame decodes of the same latents: bit-identical (no statistic depends on the batch).
"""
import numpy as np
import pytest
import torch

from conftest import psnr, report

pytestmark = pytest.mark.gpu

PICK = [0, 63, 64, 127]


def _oracle_for(pipe_cpu_state):
    from helpers import make_oracle_unet, make_oracle_vae
    u, v = pipe_cpu_state
    return make_oracle_unet(u.config, u.state_dict), make_oracle_vae(v.config, v.state_dict)


def test_batch128_frames_match_batch4_and_the_oracle(hip, dev):
    from oracle.pipeline import denoise_and_decode, numpy_to_uint8
    from oracle.scheduler import DDIMScheduler as OracleDDIM
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline
    pipe = StableDiffusionWalkPipeline.from_pretrained("CompVis/stable-diffusion-v1-4", arch="sd14")
    o_unet, o_vae = _oracle_for((pipe.unet, pipe.vae))
    pipe.to(dev)
    B = 128
    T = np.linspace(0.0, 1.0, B)
    _, embeds, noise = next(pipe.generate_inputs("a cat", "a dog", 42, 1337, (1, 4, 64, 64), T, B))
    kw = dict(height=512, width=512, num_inference_steps=2, guidance_scale=7.5, eta=0.0, output_type="numpy_u8")
    big = pipe(latents=noise, text_embeddings=embeds, **kw)["images"]
    assert big.shape == (B, 512, 512, 3) and big.dtype == np.uint8
    idx = torch.tensor(PICK, device=embeds.device)
    skw = dict(latents=noise[idx].contiguous(), text_embeddings=embeds[idx].contiguous(), **kw)
    pipe.use_graphs = False          # (eager: the two 4-frame runs below differ only in a host-side knob a captured graph would freeze)
    hip.FORCE_TILE = 6
    try:
        same_tiles = pipe(**skw)["images"]
    finally:
        hip.FORCE_TILE = 0
    small = pipe(**skw)["images"]
    pipe.use_graphs = True
    d0 = np.abs(big[PICK].astype(int) - same_tiles.astype(int))
    d = np.abs(big[PICK].astype(int) - small.astype(int))
    p_small = psnr(torch.from_numpy(big[PICK].astype(np.float32)), torch.from_numpy(small.astype(np.float32)), peak=255.0)
    report(f"batch 128 vs batch 4, SD-1.4 512x512, 2 steps, frames {PICK}: SAME tiles uint8 max-abs {d0.max()} (must be 0); the "
           f"cost model's tiles: max-abs {d.max()}, mean-abs {d.mean():.3f}, {p_small:.1f} dB between the two bf16 realisations")
    assert d0.max() == 0
    assert p_small >= 37.0
    # distinct frames really are distinct (a batch-index bug that returned one frame 128 times would pass the line above)
    assert np.abs(big[0].astype(int) - big[127].astype(int)).mean() > 1.0
    # the two end frames against the CPU oracle
    uncond = pipe.embed_text("").cpu()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    for k in (0, 127):
        ref = denoise_and_decode(o_unet, o_vae, OracleDDIM(), embeds[k:k + 1].cpu(), uncond, noise[k:k + 1].cpu(),
                                 num_inference_steps=2, guidance_scale=7.5)
        p = psnr(torch.from_numpy(big[k:k + 1].astype(np.float32) / 255.0), torch.from_numpy(ref), peak=1.0)
        d8 = np.abs(big[k:k + 1].astype(int) - numpy_to_uint8(ref).astype(int))
        report(f"batch 128, frame {k} vs the fp32 oracle: PSNR {p:.1f} dB, uint8 max-abs {d8.max()} mean-abs {d8.mean():.3f}")
        assert p >= 39.0


def test_vae_decode_of_128_frames_matches_4_frame_decodes(hip, dev):
    from stable_diffusion_videos_amd import config as cfgs
    from helpers import vae_pair
    _, engine = vae_pair(cfgs.sd_vae(), dev)
    g = torch.Generator().manual_seed(11)
    lat = (torch.randn((128, 64, 64, 4), generator=g) * 0.18215 * 0.8).to(dev)
    big, _ = engine.decode(lat)
    torch.cuda.synchronize()
    assert big.shape == (128, 512, 512, 3)
    worst = 0
    for lo in (0, 60, 124):
        small, _ = engine.decode(lat[lo:lo + 4].contiguous())
        worst = max(worst, int((big[lo:lo + 4].int() - small.int()).abs().max()))
    report(f"VAE decode, 128 frames at 512x512 vs 4-frame decodes of the same latents: uint8 max-abs {worst}")
    assert worst == 0
    assert float((big[0].float() - big[127].float()).abs().mean()) > 1.0

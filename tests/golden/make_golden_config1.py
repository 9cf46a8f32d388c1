"""Generate the BASELINE config-1 fixture with the CPU oracle (test infrastructure; run once, here, on CPU):

    walk(['a cat', 'a dog'], seeds=[42, 1337], num_interpolation_steps=3, 512x512, 50 DDIM steps, CFG 7.5)
        /root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:556 (walk) -> :481 (make_clip_frames)
        -> :457-479 (lerp / slerp) -> :412-430 (the 50-step loop) -> :432-438, :450 (VAE decode, uint8)

SD-v1-4 architectures (859.52 M-parameter UNet, 49.49 M-parameter VAE decoder, 123.06 M-parameter CLIP text
tower) with the repo's seeded, bf16-exact synthetic weights (there are no checkpoints offline) and the synthetic
HashTokenizer - exactly what ``StableDiffusionWalkPipeline.from_pretrained("CompVis/stable-diffusion-v1-4")``
builds here, so the GPU test replays the same walk through the HIP path and compares.

Everything is fp32 PyTorch-eager on the CPU (``oracle/``): ~50 x 2 UNet forwards + 1 VAE decode per frame,
roughly 10-15 minutes per frame on 8 cores, which is why the result is committed as a fixture
(``config1_sd14_50steps.npz``: text embeddings, interpolated inputs, the latents after steps 1/10/25/50 and the
three uint8 frames) instead of being recomputed inside the test.

    python tests/golden/make_golden_config1.py [--steps 50] [--out tests/golden/config1_sd14_50steps.npz]
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden" / "config1_sd14_50steps.npz"))
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)

    from helpers import make_oracle_unet, make_oracle_vae
    from oracle import interp
    from oracle.clip import clip_text_forward
    from oracle.pipeline import decode_latents, denoise_and_decode, numpy_to_uint8
    from oracle.scheduler import DDIMScheduler as OracleDDIM
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline

    prompts, seeds, guidance = ["a cat", "a dog"], [42, 1337], 7.5
    pipe = StableDiffusionWalkPipeline.from_pretrained("CompVis/stable-diffusion-v1-4", arch="sd14")
    unet = make_oracle_unet(pipe.unet.config, pipe.unet.state_dict)
    vae = make_oracle_vae(pipe.vae.config, pipe.vae.state_dict)
    tc = pipe.text_encoder.config
    tsd = pipe.text_encoder.state_dict()

    def embed(text):                                                             # :809-820
        ids = pipe.tokenizer(text, padding="max_length", max_length=pipe.tokenizer.model_max_length, truncation=True,
                             return_tensors="pt").input_ids
        return clip_text_forward(tsd, ids, tc.num_attention_heads, tc.hidden_act)

    ea, eb, uncond = embed(prompts[0]), embed(prompts[1]), embed("")
    h = a.size // 8
    la, lb = interp.init_noise(seeds[0], (1, 4, h, h)), interp.init_noise(seeds[1], (1, 4, h, h))   # :461-462
    T = np.linspace(0.0, 1.0, a.frames)                                          # :509
    batches = list(interp.generate_inputs(ea, eb, la, lb, T, a.frames))          # :464-479, one batch
    _, embeds, noise = batches[0]

    snaps = {}
    keep = sorted({0, 9, 24, a.steps - 1} & set(range(a.steps)))
    t0 = time.time()

    def cb(i, t, latents):
        if i in keep:
            snaps[i] = latents.clone()
        print(f"[config1] step {i + 1}/{a.steps} t={int(t)}  {time.time() - t0:.0f} s", flush=True)

    lat = denoise_and_decode(unet, vae, OracleDDIM(), embeds, uncond, noise, a.steps, guidance, callback=cb,
                             return_latents=True)
    imgs = decode_latents(vae, lat)
    out = dict(prompt_embeds=torch.cat([ea, eb]).numpy(), uncond_embeds=uncond.numpy(), embeds=embeds.numpy(),
               noise=noise.numpy(), T=T, latents_final=lat.numpy(), frames_u8=numpy_to_uint8(imgs),
               steps=np.int64(a.steps), guidance=np.float64(guidance), seeds=np.asarray(seeds),
               oracle_seconds=np.float64(time.time() - t0), threads=np.int64(torch.get_num_threads()))
    for i, v in snaps.items():
        out[f"latents_step{i + 1}"] = v.numpy()
    np.savez_compressed(a.out, **out)
    print(f"[config1] wrote {a.out}: {a.frames} frames, {a.steps} steps, {time.time() - t0:.0f} s on "
          f"{torch.get_num_threads()} threads")


if __name__ == "__main__":
    main()

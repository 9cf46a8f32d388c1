#!/usr/bin/env python
"""Are a call's frames the same bits whether its steps ran eagerly, as replays of the captured step, or as the first call's mix
(step 0 eager, then the capture, then replays)?  usage: python tools/replay_determinism.py [arch] [B] [size] [steps]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import StableDiffusionWalkPipeline  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
size = int(sys.argv[3]) if len(sys.argv) > 3 else (128 if arch == "tiny" else 512)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
name = {"tiny": "tiny", "sd14": "CompVis/stable-diffusion-v1-4"}[arch]
pipe = StableDiffusionWalkPipeline.from_pretrained(name, arch=arch).to("cuda")
h = size // 8
_, e, n = next(pipe.generate_inputs("a cat", "a dog", 42, 1337, (1, 4, h, h), np.linspace(0, 1, B), B))


def run():
    out = pipe(latents=n, text_embeddings=e, height=size, width=size, num_inference_steps=steps, guidance_scale=7.5, eta=0.0,
               output_type="numpy_u8")["images"].astype(np.int32)
    torch.cuda.synchronize()
    return out


first = run()                       # step 0 eager, capture, replays
r1, r2 = run(), run()               # replays only
pipe.use_graphs = False
pipe._drop_graphs()
e1, e2 = run(), run()               # eager only
d = lambda a, b: int(np.abs(a - b).max())
print(f"{arch} B={B} {size}x{size} {steps} steps: first-call vs replay {d(first, r1)}   replay vs replay {d(r1, r2)}   eager vs eager {d(e1, e2)}   "
      f"replay vs eager {d(r1, e1)}   first-call vs eager {d(first, e1)}")

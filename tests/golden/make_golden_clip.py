#!/usr/bin/env python
"""Golden vectors for the text-encoder call (stable_diffusion_pipeline.py:819): the REAL ``transformers.CLIPTextModel``
run in this container on seeded weights and ids.  Writes tests/golden/clip_{quick_gelu,gelu}.npz holding the config,
the full state dict (tiny model), the ids and ``model(ids)[0]``.

    python tests/golden/make_golden_clip.py
"""
import os
from pathlib import Path

import numpy as np
import torch
import transformers
from transformers import CLIPTextConfig, CLIPTextModel

OUT = Path(__file__).resolve().parent


def main():
    for act, layers in (("quick_gelu", 2), ("gelu", 1)):
        torch.manual_seed(7)
        # head dim 64 = the head dim of both SD text encoders (ViT-L/14: 768/12, OpenCLIP-H: 1024/16)
        cfg = CLIPTextConfig(vocab_size=211, hidden_size=128, intermediate_size=128, num_hidden_layers=layers,
                             num_attention_heads=2, max_position_embeddings=77, hidden_act=act, bos_token_id=209,
                             eos_token_id=210, pad_token_id=210, projection_dim=64)
        model = CLIPTextModel(cfg).float().eval()
        with torch.no_grad():
            for p in model.parameters():                # default init is tiny (std 0.02): widen it so that the
                p.mul_(2.0)                             # non-linearities are exercised (x4 makes the net chaotic)
                p.copy_(p.to(torch.bfloat16).float())   # bf16-representable: the bf16 HIP path shares them exactly
        ids = torch.randint(1, 209, (3, 77))
        ids[:, 0] = 209
        ids[0, 9:] = 210
        ids[1, 40:] = 210
        ids[2, 76] = 210
        with torch.no_grad():
            out = model(ids)[0]
        # stored as the upper 16 bits of the fp32 pattern (exact, see above) to halve the fixture
        arrays = {"sd::" + k: (v.contiguous().view(torch.int32) >> 16).to(torch.int16).numpy()
                  for k, v in model.state_dict().items() if v.is_floating_point()}
        np.savez_compressed(OUT / f"clip_{act}.npz", ids=ids.numpy(), last_hidden_state=out.numpy(),
                            num_heads=np.int64(2), hidden_act=np.array(act), transformers_version=np.array(transformers.__version__),
                            **arrays)
        print(act, out.shape, float(out.abs().mean()), os.path.getsize(OUT / f"clip_{act}.npz"))


if __name__ == "__main__":
    main()

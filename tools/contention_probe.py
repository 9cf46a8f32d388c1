#!/usr/bin/env python
"""Which block of the UNet changes its output when ANOTHER process keeps the same GPU busy?
(tests/test_model_gpu.py::test_bench_launches_its_own_ranks: two ranks on one GPU - a frame generated while the other rank was
running differed, by a few uint8 steps and differently every time, from the same frame generated alone; one rank alone is
bit-reproducible; SDV_LN_FOLD=0 makes the difference go away.)
Foreground: the tiny UNet's eager forward with engine.TAP recording every block / sub-block output - once alone (baseline), then
RUNS times while a child process runs the same forward in a loop; per tap name the number of runs whose output differs from
the baseline, in forward order (the first name with a non-zero count is where it starts).
usage: python tools/contention_probe.py [runs] [arch] [noise arch]        (internal: ... noise <seconds> <arch>)"""
import subprocess
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, engine, hip  # noqa: E402

BF16 = torch.bfloat16


def setup(arch):
    name = {"tiny": "tiny", "sd14": "CompVis/stable-diffusion-v1-4"}[arch]
    pipe = StableDiffusionWalkPipeline.from_pretrained(name, arch=arch).to("cuda")
    B, h = 4, (16 if arch == "tiny" else 64)
    emb = pipe.embed_text(["a cat"] * B)
    ctx = torch.cat([pipe._uncond_embeddings(None, B), emb.float()])
    pipe._schedule(50, 0.0)
    pipe.unet.prepare_context(ctx)
    g = torch.Generator(device="cuda").manual_seed(3)
    x2 = torch.randn((2 * B * h * h, 4), device="cuda", generator=g).to(BF16)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    return pipe, x2, B, h, step


def noise(seconds, arch):
    pipe, x2, B, h, step = setup(arch)
    pipe.unet.forward(x2, 2 * B, h, h, step, cfg_shared=True)
    torch.cuda.synchronize()
    print("ready", flush=True)          # (tests/test_contention_gpu.py waits for this line before it starts comparing)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(5):
            pipe.unet.forward(x2, 2 * B, h, h, step, cfg_shared=True)
        torch.cuda.synchronize()


def tapped_forward(pipe, x2, B, h, step):
    rec = []
    engine.TAP = lambda name, d: rec.append((f"{name}:{d['kind']}", d["out"].detach().clone()))
    engine.TAP_AUX = lambda name, t: rec.append((name, t.detach().clone()))
    try:
        eps = pipe.unet.forward(x2, 2 * B, h, h, step, cfg_shared=True)
    finally:
        engine.TAP = engine.TAP_AUX = None
    torch.cuda.synchronize()
    rec.append(("eps", eps.clone()))
    return rec


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    arch = sys.argv[2] if len(sys.argv) > 2 else "tiny"
    pipe, x2, B, h, step = setup(arch)
    base = tapped_forward(pipe, x2, B, h, step)
    again = tapped_forward(pipe, x2, B, h, step)
    alone = sum(not torch.equal(a[1], b[1]) for a, b in zip(base, again))
    print(f"{arch}: {len(base)} taps; alone, second forward vs first: {alone} taps differ", flush=True)
    noise_arch = sys.argv[3] if len(sys.argv) > 3 else arch
    child = subprocess.Popen([sys.executable, __file__, "noise", "600", noise_arch], stdout=subprocess.PIPE, text=True)
    while "ready" not in child.stdout.readline():
        pass
    counts = [0] * len(base)
    firsts = {}
    try:
        for r in range(runs):
            rec = tapped_forward(pipe, x2, B, h, step)
            first = None
            for i, ((n0, a), (n1, b)) in enumerate(zip(base, rec)):
                if not torch.equal(a, b):
                    counts[i] += 1
                    if first is None:
                        first = n0
            if first is not None:
                firsts[first] = firsts.get(first, 0) + 1
    finally:
        child.kill()
    print(f"with a second process running the same forward: runs whose FIRST differing tap is ... {firsts or 'none: all runs identical'}")
    for (n, _), c in zip(base, counts):
        if c:
            print(f"  {n:70s} differs in {c}/{runs} runs")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "noise":
        noise(float(sys.argv[2]), sys.argv[3])
    else:
        main()

#!/usr/bin/env python
"""Real-ESRGAN x4 (RRDBNet, 23 blocks, synthetic weights) on n 512x512 uint8 frames: frames/s and algorithmic TFLOP/s
(9.4 TFLOP per 512^2 -> 2048^2 frame) plus the per-shape launch report."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402
from stable_diffusion_videos_amd.upsampling import RealESRGANModel  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
up = RealESRGANModel(None).to("cuda")
img = torch.randint(0, 256, (n, size, size, 3), dtype=torch.uint8, device="cuda")
up.upsample_u8(img)
torch.cuda.synchronize()


class Prof:
    def __init__(self):
        self.rec = []

    def __call__(self, kind, info, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.rec.append((kind, info, s, e))


t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    out = up.upsample_u8(img)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
flop = 2 * 4.70e12 * (size / 512) ** 2
print(f"esrgan x4: {n} frames {size}^2 -> {tuple(out.shape)}: {dt * 1e3 / n:.1f} ms/frame, {n / dt:.2f} frames/s, "
      f"{flop * n / dt / 1e12:.0f} TFLOP/s algorithmic")
p = Prof()
hip.LAUNCH_HOOK = p
up.upsample_u8(img)
torch.cuda.synchronize()
hip.LAUNCH_HOOK = None
agg = {}
for kind, info, s, e in p.rec:
    key = (kind, info.get("M"), info.get("N"), info.get("K"), info.get("mode"), info.get("epi"))
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += s.elapsed_time(e)
    a[2] += info.get("flops", 0.0)
tot = sum(a[1] for a in agg.values())
print(f"event-timed total {tot:.1f} ms for {n} frames")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(key, a[0], f"{a[1]:.2f} ms", f"{a[2] / (a[1] * 1e-3) / 1e12:.0f} TF" if a[2] else "")

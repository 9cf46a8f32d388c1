#!/usr/bin/env python
"""Time the UNet's self-attention shapes (HIP events).  usage: attn_bench.py [nimg] [qscale ...]
qscale multiplies Q: 1 = the N(0,1) logits of random-init weights (the O-rescale branch practically never fires), 4-6 = the
logit spread of a trained model's self-attention (it fires in most key tiles)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402

nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 64
qscales = [float(a) for a in sys.argv[2:]] or [1.0]
dev = torch.device("cuda")
for qs in qscales:
  for dh, L, heads in ((40, 4096, 8), (80, 1024, 8), (160, 256, 8)):
    C = dh * heads
    qk = torch.randn((nimg * L, 2 * C), device=dev)
    qk[:, :C] *= qs
    qk = qk.to(torch.bfloat16)
    vt = torch.randn((nimg, C, L), device=dev).to(torch.bfloat16)
    o = torch.empty((nimg * L, C), dtype=torch.bfloat16, device=dev)
    fn = lambda: hip.attention(qk, qk, vt, o, B=nimg, H=heads, Lq=L, Lk=L, dh=dh, ldq=2 * C, ldk=2 * C, ldv=L, ldo=C,
                               scale=dh ** -0.5, k_off=C)
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"qscale={qs} dh={dh} L={L} nimg={nimg}: {ms:.3f} ms  {4.0 * nimg * heads * L * L * dh / ms / 1e9:.0f} TFLOP/s algorithmic")
# text cross-attention (Lk = 77): Q + O streaming, two key tiles per query block
for dh, L, heads in ((40, 4096, 8), (80, 1024, 8), (160, 256, 8)):
    C = dh * heads
    q = torch.randn((nimg * L, C), device=dev).to(torch.bfloat16)
    k = torch.randn((nimg * 77, C), device=dev).to(torch.bfloat16)
    vt = torch.zeros((nimg, C, 128), dtype=torch.bfloat16, device=dev)
    vt[:, :, :77] = torch.randn((nimg, C, 77), device=dev).to(torch.bfloat16)
    o = torch.empty((nimg * L, C), dtype=torch.bfloat16, device=dev)
    fn = lambda: hip.attention(q, k, vt, o, B=nimg, H=heads, Lq=L, Lk=77, dh=dh, ldq=C, ldk=C, ldv=128, ldo=C, scale=dh ** -0.5)
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"cross dh={dh} Lq={L} Lk=77 nimg={nimg}: {ms:.3f} ms  {4.0 * nimg * heads * L * 77 * dh / ms / 1e9:.0f} TFLOP/s algorithmic  "
          f"{2 * 2.0 * nimg * L * C / ms / 1e6:.0f} GB/s of Q + O")

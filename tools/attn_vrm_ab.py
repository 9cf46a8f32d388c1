#!/usr/bin/env python
"""Where does the row-major-V form of the self-attention lose time against the transposed-V form?  Four operand layouts per UNet
self-attention shape, same arithmetic (HIP events, median of 5 x 5 launches):
  vt/2C   Q, K in a [tokens, 2C] buffer, V transposed [C, tokens]                      (rounds 2-4)
  vt/3C   Q, K read out of a [tokens, 3C] buffer, V transposed                         (only the K / Q row stride changes)
  vrm/3C  Q, K, V in one [tokens, 3C] buffer, V row-major                              (round 5, what UNetEngine launches)
  vrm/C   Q, K in [tokens, 2C], V row-major in its own dense [tokens, C] buffer        (row-major V without the 3C row stride)
usage: python tools/attn_vrm_ab.py [nimg]"""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402

nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
BF16 = torch.bfloat16


def timed(fn):
    ms = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms.append(s.elapsed_time(e) / 5)
    return statistics.median(ms)


for dh, L, heads in ((40, 4096, 8), (80, 1024, 8), (160, 256, 8), (64, 2304, 5)):
    C = dh * heads
    n = nimg if L <= 4096 else nimg // 4
    qkv = torch.randn((n * L, 3 * C), device=dev).to(BF16)
    qk = qkv[:, :2 * C].contiguous()
    v = qkv[:, 2 * C:].contiguous()
    vt = v.view(n, L, C).transpose(1, 2).contiguous()
    o = torch.empty((n * L, C), dtype=BF16, device=dev)
    kw = dict(B=n, H=heads, Lq=L, Lk=L, dh=dh, ldo=C, scale=dh ** -0.5, q_prescaled=True)
    forms = {
        "vt/2C": lambda: hip.attention(qk, qk, vt, o, ldq=2 * C, ldk=2 * C, ldv=L, k_off=C, **kw),
        "vt/3C": lambda: hip.attention(qkv, qkv, vt, o, ldq=3 * C, ldk=3 * C, ldv=L, k_off=C, **kw),
        "vrm/3C": lambda: hip.attention(qkv, qkv, qkv, o, ldq=3 * C, ldk=3 * C, ldv=3 * C, k_off=C, v_off=2 * C, v_rowmajor=True, **kw),
        "vrm/C": lambda: hip.attention(qk, qk, v, o, ldq=2 * C, ldk=2 * C, ldv=C, k_off=C, v_rowmajor=True, **kw),
    }
    outs = {}
    for name, fn in forms.items():
        fn()
        torch.cuda.synchronize()
        outs[name] = o.clone()
    assert all(torch.equal(outs["vt/2C"], x) for x in outs.values())
    res = {name: timed(fn) for name, fn in forms.items()}
    fl = 4.0 * n * heads * L * L * dh
    print(f"dh {dh:3d} L {L:4d} x {n:3d} samples: " + "   ".join(f"{k} {v:.3f} ms ({fl / v / 1e9:4.0f} TF)" for k, v in res.items()), flush=True)

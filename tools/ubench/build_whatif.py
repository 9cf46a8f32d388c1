#!/usr/bin/env python
"""Timing-only variants of the attention kernel (-DSDV_WHATIF=n removes one cost at a time: 1 exps, 2 barriers, 3 row max,
4 K/V staging, 5 LDS stores only, 6 global loads only; results are wrong by construction).  Builds tools/ubench/libsdv_whatif{n}.so with the same C ABI; run e.g.
    SDV_HIP_LIB=tools/ubench/libsdv_whatif1.so python tools/attn_bench.py 64"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402

b.build()
src = b.CSRC / "sdv_attention.hip"
others = [b.OBJDIR / f"{s.stem}.o" for s in b.sources() if s.name != src.name]
for n in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else (1, 2, 3, 4, 5, 6):
    obj = Path(__file__).resolve().parent / f"attn_whatif{n}.o"
    subprocess.run([b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS[src.name], f"-DSDV_WHATIF={n}", "-c", str(src), "-o", str(obj)],
                   check=True)
    out = Path(__file__).resolve().parent / f"libsdv_whatif{n}.so"
    subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", str(obj), *map(str, others), "-o", str(out)], check=True)
    obj.unlink()
    print("built", out)

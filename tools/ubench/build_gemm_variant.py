#!/usr/bin/env python
"""Diagnostic builds of csrc/sdv_gemm.hip: compiles it with the given -D flags and links it with the cached objects of the other kernel
files into tools/ubench/libsdv_gemm_<tag>.so (SDV_HIP_LIB=... python tools/contention_probe.py ...).   usage: build_gemm_variant.py tag -DX [...]"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402

b.build()
tag, flags = sys.argv[1], sys.argv[2:]
src = b.CSRC / "sdv_gemm.hip"
others = [str(b.OBJDIR / (s.stem + ".o")) for s in b.sources() if s.name != src.name]
obj = Path(f"/tmp/sdv_gemm_{tag}.o")
subprocess.run([b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS.get(src.name, []), *flags, "-c", str(src), "-o", str(obj)], check=True)
out = ROOT / "tools" / "ubench" / f"libsdv_gemm_{tag}.so"
subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", str(obj), *others, "-o", str(out)], check=True)
print("built", out)

#!/usr/bin/env python
"""Tools-only build of the igemm with per-phase cycle stamps (-DSDV_GEMM_TIMING=<workgroup>, 256 x 320 tile only):
workgroup <n>'s wave 0 writes s_memtime at kernel start (0), after tile setup (1), after the K loop (2), before / after the
epilogue barrier (5 / 3) and after the epilogue (4) for every tile it walks.  Builds tools/ubench/libsdv_gemm_timing.so with
the same C ABI + sdv_gemm_debug_timing(buf);  run   SDV_HIP_LIB=tools/ubench/libsdv_gemm_timing.so python tools/gemm_phases.py"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402

b.build()
src = b.CSRC / "sdv_gemm.hip"
wg = sys.argv[1] if len(sys.argv) > 1 else "0"
extra = sys.argv[2:]          # extra -D flags
name = "libsdv_gemm_timing.so" if wg != "notiming" else "libsdv_gemm_dbg%s.so" % "".join(c for c in "".join(extra) if c.isdigit())
others = [b.OBJDIR / f"{s.stem}.o" for s in b.sources() if s.name != src.name]
obj = Path(__file__).resolve().parent / "gemm_timing.o"
subprocess.run([b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS[src.name], *([f"-DSDV_GEMM_TIMING={wg}"] if wg != "notiming" else []), *extra, "-DSDV_GEMM_ONLY_TILE6", "-c",
                str(src), "-o", str(obj)], check=True)
out = Path(__file__).resolve().parent / name
subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", str(obj), *map(str, others), "-o", str(out)], check=True)
obj.unlink()
print("built", out)

#!/usr/bin/env python
"""A/B builds of sdv_gemm.hip with extra -D knobs (SDV_BF16_ROT_AH, SDV_EPI_VEC_AHEAD): tools/ubench/libsdv_<name>.so, same C ABI.
usage: build_variant.py name -DKNOB=V [-DKNOB=V ...]     then  SDV_HIP_LIB=tools/ubench/libsdv_<name>.so python tools/res_ab.py 256 5"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402

here = Path(__file__).resolve().parent
src = b.CSRC / "sdv_gemm.hip"
name, defs = sys.argv[1], sys.argv[2:]
obj = here / f"sdv_gemm_{name}.o"
subprocess.run([b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS[src.name], *defs, "-c", str(src), "-o", str(obj)], check=True)
others = [b.OBJDIR / f"{s.stem}.o" for s in b.sources() if s.name != src.name]
out = here / f"libsdv_{name}.so"
subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", str(obj), *map(str, others), "-o", str(out)], check=True)
obj.unlink()
print("built", out)

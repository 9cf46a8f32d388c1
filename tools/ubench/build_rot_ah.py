#!/usr/bin/env python
"""A/B builds of the bf16 K-loop fragment schedule (sdv_gemm.hip, SDV_BF16_ROT_AH): tools/ubench/libsdv_rot_ah{n}.so with the same
C ABI for n in argv (0 = the round-2 order: two whole fragment sets, compiler-scheduled; n > 0 = rotating W fragments, n in
flight ahead of the MFMAs).  Run e.g.  SDV_HIP_LIB=tools/ubench/libsdv_rot_ah0.so python tools/res_ab.py 256 5"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402

here = Path(__file__).resolve().parent
src = b.CSRC / "sdv_gemm.hip"


def one(n: int):
    obj = here / f"sdv_gemm_rot_ah{n}.o"
    subprocess.run([b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS[src.name], f"-DSDV_BF16_ROT_AH={n}", "-c", str(src), "-o", str(obj)],
                   check=True)
    others = [b.OBJDIR / f"{s.stem}.o" for s in b.sources() if s.name != src.name]
    out = here / f"libsdv_rot_ah{n}.so"
    subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", str(obj), *map(str, others), "-o", str(out)], check=True)
    obj.unlink()
    return out


if __name__ == "__main__":
    b.build()
    with ThreadPoolExecutor(max_workers=4) as ex:
        for out in ex.map(one, [int(a) for a in sys.argv[1:]] or [0, 1, 3]):
            print("built", out)

#!/usr/bin/env python
"""Compile one csrc/*.hip file with the build's flags + -Rpass-analysis=kernel-resource-usage and print, per kernel,
VGPRs / AGPRs / SGPRs / scratch bytes per lane / LDS / occupancy.   usage: kernel_resources.py sdv_gemm.hip [-DFOO ...]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sdv_gemm.hip"
cmd = [b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS.get(name, []), *sys.argv[2:], "-Rpass-analysis=kernel-resource-usage",
       "-c", str(ROOT / "stable_diffusion_videos_amd" / "csrc" / name), "-o", "/tmp/_kres.o"]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-4000:])
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
for blk in blocks:
    fn = blk.split()[0]
    g = lambda k: (re.search(k + r": (\d+)", blk) or [None, "?"])[1]
    m = re.search(r"igemm_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELi(\d+)", fn)
    short = "igemm<%s>" % ",".join(m.groups()) if m else fn[:70]
    scr, lds, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{short:40s} VGPR {g('VGPRs'):>4s} AGPR {g('AGPRs'):>4s} SGPR {g('SGPRs'):>4s} scratch {scr:>5s} LDS {lds:>7s} occ {occ}")

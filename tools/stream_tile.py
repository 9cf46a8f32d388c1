#!/usr/bin/env python
"""Build and run tools/ubench/stream_tile.hip (memory-stream probe for the K <= 640 GEMM access pattern: TB/s as a function of
the K slabs a CU keeps in flight, with and without the row-major stores and a stand-in for the MFMA time).
usage: python tools/stream_tile.py [rows]      (needs an MI355X; `--build-only` cross-compiles)"""
import subprocess
import sys
from pathlib import Path

here = Path(__file__).resolve().parent / "ubench"
exe = here / "stream_tile"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", str(here / "stream_tile.hip"), "-o", str(exe)],
               check=True)
if "--build-only" not in sys.argv:
    subprocess.run([str(exe), *[a for a in sys.argv[1:] if not a.startswith("--")]], check=True)

"""Build libsdv_hip.so (the gfx950 kernels behind the C ABI in include/sdv_hip.h) in-tree with hipcc.

``python -m stable_diffusion_videos_amd.build [--force]``.  hipcc cross-compiles for gfx950 without a
GPU; the resulting ``stable_diffusion_videos_amd/lib/libsdv_hip.so`` is git-ignored but travels with
the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
LIBDIR = PKG / "lib"
OBJDIR = PKG / "lib" / "obj"
LIB = LIBDIR / "libsdv_hip.so"
ARCH = "gfx950"

FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I", str(INCLUDE)]
# fast-math only where it buys throughput (MFMA epilogues / softmax); the interpolation + scheduler kernels keep
# IEEE division so that slerp(0) == v0 and slerp(1) == v1 hold exactly, as in the reference's numpy arithmetic.
FAST_MATH = {"sdv_gemm.hip", "sdv_attention.hip", "sdv_norm.hip", "sdv_ffn.hip"}
FAST_FLAGS = ["-ffast-math", "-fno-finite-math-only"]
# attention: MFMA results are consumed by the softmax VALU code straight away - keep them in VGPRs (no
# v_accvgpr_read/write traffic; 123+32 -> 128 registers, 3 -> 4 waves/SIMD for the 40/64-wide heads)
EXTRA_FLAGS = {"sdv_attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # igemm: + no SLP vectoriser - the one result that ever differed under a second process on the GPU came out of a
               # v_pk_* + transcendental sequence of a GEGLU epilogue and went away without the packed arithmetic (DESIGN.md "The
               # co-residency finding"); the forward is 0.1 - 0.4 % FASTER without it (same box: 12.33 / 12.36 vs 12.39 / 12.37 frames/s)
               "sdv_gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
               # fused feed-forward: one wave per SIMD with all 512 registers (AGPR-form MFMAs: no vgpr-form here); its VALU stream is
               # laid out by hand beside the MFMAs, where packed fp32 ops (v_pk_fma_f32 out of the SLP vectoriser) are an anti-lever
               "sdv_ffn.hip": ["-fno-slp-vectorize"]}


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: cannot build libsdv_hip.so")
    return exe


def sources():
    return sorted(CSRC.glob("*.hip"))


def _newest_dep() -> float:
    deps = list(CSRC.glob("*.h")) + list(INCLUDE.glob("*.h")) + [Path(__file__)]        # (a changed flag is a changed object)
    return max(p.stat().st_mtime for p in deps)


def _compile(src: Path, force: bool) -> Path:
    obj = OBJDIR / (src.stem + ".o")
    if (not force and obj.exists() and obj.stat().st_mtime >= src.stat().st_mtime
            and obj.stat().st_mtime >= _newest_dep()):
        return obj
    cmd = [hipcc(), *FLAGS, *(FAST_FLAGS if src.name in FAST_MATH else []), *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJDIR.mkdir(parents=True, exist_ok=True)
    srcs = sources()
    if not srcs:
        raise RuntimeError(f"no .hip sources under {CSRC}")
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not LIB.exists() or any(o.stat().st_mtime > LIB.stat().st_mtime for o in objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", *map(str, objs), "-o", str(LIB)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({LIB.stat().st_size >> 10} KiB) from {[s.name for s in srcs]}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)

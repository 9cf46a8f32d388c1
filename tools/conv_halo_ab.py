#!/usr/bin/env python
"""Prototype A/B: the halo-tile conv3x3 (tools/experiments/sdv_conv_halo.hip - the X window of a 256-pixel tile staged ONCE per
channel slab, nine taps read it at shifted LDS rows) against the shipped tap-major implicit GEMM (tile 6) on the ResBlock conv
shapes of a UNet forward.  Correctness: against the shipped kernel on all rows and against a float64 conv on the first two images.
usage: python tools/conv_halo_ab.py [nimg] [rounds]      (builds tools/ubench/libsdv_conv_halo.so with hipcc when it is missing)"""
import ctypes as C
import statistics
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402
from stable_diffusion_videos_amd import hip  # noqa: E402

import os  # noqa: E402

# HALO_SRC=sdv_conv_halo_persistent.hip: the tile-walk variant (persistent with -DHALO_PERSISTENT)
SRC = ROOT / "tools" / "experiments" / os.environ.get("HALO_SRC", "sdv_conv_halo.hip")
# HALO_DEFS="-DHALO_WHATIF_NO_EPILOGUE": timing-only what-if builds of the prototype (their outputs are wrong by construction)
DEFS = os.environ.get("HALO_DEFS", "").split()
# HALO_AGPR=1: compile without -amdgpu-mfma-vgpr-form (accumulators in AccVGPRs)
VGPR_FORM = [] if os.environ.get("HALO_AGPR") == "1" else ["-mllvm", "-amdgpu-mfma-vgpr-form"]
LIB = ROOT / "tools" / "ubench" / ("lib" + SRC.stem + "".join(d.replace("-D", "_").lower() for d in DEFS) + ("" if VGPR_FORM else "_agpr") + ".so")


def build():
    if LIB.exists() and LIB.stat().st_mtime >= SRC.stat().st_mtime:
        return
    b.build()
    obj = LIB.with_suffix(".o")
    subprocess.run([b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *VGPR_FORM, "-I", str(b.CSRC), *DEFS, "-c", str(SRC), "-o", str(obj)],
                   check=True)
    subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", str(obj), str(b.OBJDIR / "sdv_elementwise.o"), "-o", str(LIB)],
                   check=True)
    obj.unlink()


def timed(fn, reps=3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    build()
    lib = C.CDLL(str(LIB))
    lib.sdv_conv3x3_halo_bf16.restype = C.c_int
    lib.sdv_conv3x3_halo_bf16.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 11 + [C.c_void_p, C.c_int32, C.c_void_p]
    lib.sdv_last_error.restype = C.c_char_p
    dev = torch.device("cuda")
    hip.load()
    cases = [("conv 320->320 @64", 64, 320, 0, 320, True), ("conv 640->640 @32", 32, 640, 0, 640, True),
             ("conv 1280->1280 @16", 16, 1280, 0, 1280, True), ("conv 1280->1280 @8", 8, 1280, 0, 1280, True),
             ("conv 640+320->320 @64", 64, 640, 320, 320, False), ("conv 320->640 @32", 32, 320, 0, 640, False),
             ("conv 1280+1280->1280 @16", 16, 1280, 1280, 1280, False)]
    print(f"nimg={nimg} rounds={rounds}   TFLOP/s median (min..max): shipped igemm tile 6 | halo-tile prototype")
    stream = torch.cuda.current_stream().cuda_stream
    bn = 64 * int(next((d.split("=")[1] for d in DEFS if d.startswith("-DHALO_TN=")), "5"))     # the prototype's N tile
    for label, H, c1, c2, cout, use_res in cases:
        if cout % bn:
            continue
        M, K = nimg * H * H, c1 + c2
        g = torch.Generator(device=dev).manual_seed(1)
        x = (torch.randn((M, c1), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        x2 = (torch.randn((M, c2), device=dev, generator=g) * 0.5).to(torch.bfloat16) if c2 else None
        w = (torch.randn((cout, 9 * K), device=dev, generator=g) * (9 * K) ** -0.5).to(torch.bfloat16)
        bias = torch.randn(cout, device=dev, generator=g)
        res = torch.randn((M, cout), device=dev, generator=g).to(torch.bfloat16) if use_res else None
        out_a = torch.empty((M, cout), dtype=torch.bfloat16, device=dev)
        out_b = torch.full((M, cout), float("nan"), dtype=torch.bfloat16, device=dev)

        def run_a():
            hip.conv3x3(x, w, bias, nimg=nimg, H=H, W=H, x2=x2, residual=res, out=out_a, tile=6)

        def run_b():
            rc = lib.sdv_conv3x3_halo_bf16(x.data_ptr(), x2.data_ptr() if c2 else None, w.data_ptr(), bias.data_ptr(),
                                           res.data_ptr() if use_res else None, out_b.data_ptr(), nimg, H, H, c1, c2, cout, c1, c2 or 0,
                                           9 * K, cout, cout, None, 0, stream)
            if rc:
                raise RuntimeError(lib.sdv_last_error().decode())
        run_a()
        run_b()
        torch.cuda.synchronize()
        rel_ab = float((out_a.float() - out_b.float()).norm() / out_a.float().norm())
        n2 = 2 * H * H
        xx = torch.cat([x[:n2], x2[:n2]], dim=1) if c2 else x[:n2]
        ref = torch.nn.functional.conv2d(xx.double().view(2, H, H, K).permute(0, 3, 1, 2), w.double().view(cout, 3, 3, K).permute(0, 3, 1, 2),
                                         bias.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
        if use_res:
            ref = ref + res[:n2].double()
        rel_a = float((out_a[:n2].double() - ref).norm() / ref.norm())
        rel_b = float((out_b[:n2].double() - ref).norm() / ref.norm())
        ok = bool(torch.isfinite(out_b.float()).all())
        ms_a, ms_b = [], []
        for _ in range(rounds):
            ms_a.append(timed(run_a))
            ms_b.append(timed(run_b))
        fl = 18.0 * M * K * cout / 1e9
        f = lambda ms: f"{fl / statistics.median(ms):6.0f} ({fl / max(ms):5.0f}..{fl / min(ms):5.0f})"
        print(f"{label:26s} M={M:8d}  {f(ms_a)} | {f(ms_b)}  x{statistics.median(ms_a) / statistics.median(ms_b):.3f}   "
              f"rel-L2 vs f64: {rel_a:.2e} | {rel_b:.2e}   halo vs shipped {rel_ab:.2e}  finite={ok}", flush=True)
        del x, x2, w, out_a, out_b, res


if __name__ == "__main__":
    main()

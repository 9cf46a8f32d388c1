#!/usr/bin/env python
"""Where a 256 x 320 igemm tile's time goes: per-phase shader-clock stamps of one workgroup (tools/ubench/build_gemm_timing.py).
usage: SDV_HIP_LIB=tools/ubench/libsdv_gemm_timing.so python tools/gemm_phases.py [nimg]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import hip  # noqa: E402

NAMES = {0: "start", 1: "setup done", 2: "K loop done", 5: "epi: vectors staged", 3: "epi: barrier passed", 4: "epi done",
         6: "slab: barrier passed", 7: "slab: next issued", 8: "slab: MFMAs issued", 9: "slab: next landed",
         10: "last slab: barrier passed", 11: "last slab: next tile issued", 12: "last slab: MFMAs issued"}


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda")
    lib = hip.load()
    lib.sdv_gemm_debug_timing.restype = C.c_int
    lib.sdv_gemm_debug_timing.argtypes = [C.c_void_p]
    tbuf = torch.zeros(4096, dtype=torch.int64, device=dev)
    for label, H, cin, cout, use_res, geglu, conv in (("gemm 320->320 @64", 64, 320, 320, False, False, False),
                                                      ("gemm 320->320 @64 +res", 64, 320, 320, True, False, False),
                                                      ("gemm 320->2560 geglu @64", 64, 320, 2560, False, True, False),
                                                      ("gemm 1280->320 @64 +res", 64, 1280, 320, True, False, False),
                                                      ("gemm 640->640 @32", 32, 640, 640, False, False, False),
                                                      ("conv 320->320 @64", 64, 320, 320, False, False, True),
                                                      ("conv 320->320 @64 +res", 64, 320, 320, True, False, True)):
        M = nimg * H * H
        kk = 9 * cin if conv else cin
        x = (torch.randn((M, cin), device=dev) * 0.5).to(torch.bfloat16)
        w = (torch.randn((cout, kk), device=dev) * kk ** -0.5).to(torch.bfloat16)
        bias = torch.randn(cout, device=dev)
        nout = cout // 2 if geglu else cout
        res = torch.randn((M, nout), device=dev).to(torch.bfloat16) if use_res else None
        out = torch.zeros((M, nout), dtype=torch.bfloat16, device=dev)
        for pers in (0, 1):
            lib.sdv_gemm_set_persistent(pers)
            fn = (lambda: hip.conv3x3(x, w, bias, nimg=nimg, H=H, W=H, residual=res, out=out, tile=6)) if conv else \
                 (lambda: hip.linear(x, w, bias, residual=res, out=out, epi=1 if geglu else 0, tile=6))
            lib.sdv_gemm_debug_timing(None)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            tbuf.zero_()
            lib.sdv_gemm_debug_timing(C.c_void_p(tbuf.data_ptr()))
            fn()
            torch.cuda.synchronize()
            lib.sdv_gemm_debug_timing(None)
            st = [(int(v) >> 56, int(v) & ((1 << 56) - 1)) for v in tbuf.tolist() if v]
            # per-phase durations (shader clocks), averaged over the tiles the workgroup walked
            acc, cnt = {}, {}
            for (s0, t0), (s1, t1) in zip(st, st[1:]):
                key = f"{NAMES[s0]} -> {NAMES[s1]}"
                acc[key] = acc.get(key, 0) + (t1 - t0)
                cnt[key] = cnt.get(key, 0) + 1
            tiles = sum(1 for s, _ in st if s == 4)
            total = st[-1][1] - st[0][1] if st else 0
            print(f"{label}  persistent={pers}  tiles walked by the stamped workgroup: {tiles}  total {total} clk")
            for k in acc:
                print(f"    {k:48s} {acc[k] / cnt[k]:9.0f} clk  x{cnt[k]}")
    lib.sdv_gemm_set_persistent(1)


if __name__ == "__main__":
    main()

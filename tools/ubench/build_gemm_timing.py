#!/usr/bin/env python
"""Tools-only build of the igemm with per-phase cycle stamps (256 x 320 tile only), made from a patched COPY of
csrc/sdv_gemm.hip - the product source carries none of this: workgroup <n>'s wave 0 writes the shader clock at kernel start
(0), after tile setup (1), after the K loop (2), before / after the epilogue barrier (5 / 3) and after the epilogue (4) for every
tile it walks.  Builds tools/ubench/libsdv_gemm_timing.so with the same C ABI + sdv_gemm_debug_timing(buf); run
    SDV_HIP_LIB=tools/ubench/libsdv_gemm_timing.so python tools/gemm_phases.py
`build_gemm_timing.py notiming -DFOO=1` builds an unstamped variant with extra flags (libsdv_gemm_dbg<digits>.so)."""
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402

MACRO = """
__device__ long long* g_tbuf = nullptr;
#define SDV_STAMP(slot)                                                                        \\
    do {                                                                                       \\
        if (g_tbuf && blockIdx.x == SDV_GEMM_TIMING && threadIdx.x == 0 && tstamp < 4096) {    \\
            g_tbuf[tstamp++] = ((long long)(slot) << 56) | (__builtin_readcyclecounter() & 0xffffffffffffffLL); \\
        }                                                                                      \\
    } while (0)
"""
ENTRY = """extern "C" int sdv_gemm_debug_timing(void* buf) {
    long long* b = (long long*)buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tbuf), &b, sizeof(b));
}

"""
# (anchor in the product source, replacement) - every anchor must occur exactly once
PATCHES = [
    ("namespace {\n", "namespace {\n" + MACRO),
    ("    extern __shared__ __attribute__((aligned(16))) char smem[];\n",
     "    extern __shared__ __attribute__((aligned(16))) char smem[];\n    int tstamp = 0;\n    SDV_STAMP(0);\n"),
    ("                __builtin_amdgcn_s_barrier();\n                asm volatile(\"\" ::: \"memory\");\n                if constexpr (FEAT == 3) rowacc[lane_e] = 0.f;\n",
     "                SDV_STAMP(5);\n                __builtin_amdgcn_s_barrier();\n                asm volatile(\"\" ::: \"memory\");\n                SDV_STAMP(3);\n"
     "                if constexpr (FEAT == 3) rowacc[lane_e] = 0.f;\n"),
    ("        kloop();\n", "        SDV_STAMP(1);\n        kloop();\n        SDV_STAMP(2);\n"),
    ("        epilogue();\n", "        epilogue();\n        SDV_STAMP(4);\n"),
    # fine stamps inside the double-buffered K loop: 6 = barrier passed, 7 = next slab's pieces issued, 8 = slab's MFMAs issued,
    # 9 = next slab landed (vmcnt(0)); 10 / 11 / 12 = the last slab: barrier passed / next tile set up + its first slab issued / MFMAs issued
    ("        lds_barrier();\n        stage((slot0 + kt + 1) & 1);\n        compute((slot0 + kt) & 1);\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n",
     "        lds_barrier();\n        SDV_STAMP(6);\n        stage((slot0 + kt + 1) & 1);\n        SDV_STAMP(7);\n        compute((slot0 + kt) & 1);\n        SDV_STAMP(8);\n"
     "        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        SDV_STAMP(9);\n"),
    ("    lds_barrier();\n    if (PERSIST && has_next) {\n", "    lds_barrier();\n    SDV_STAMP(10);\n    if (PERSIST && has_next) {\n"),
    ("    compute((slot0 + nkt - 1) & 1);\n    }\n    };\n", "    SDV_STAMP(11);\n    compute((slot0 + nkt - 1) & 1);\n    SDV_STAMP(12);\n    }\n    };\n"),
    ('extern "C" int sdv_gemm_set_persistent(int on) {', ENTRY + 'extern "C" int sdv_gemm_set_persistent(int on) {'),
]


# Timing-only what-if variants of the epilogue (WRONG results, never shipped): `build_gemm_timing.py <wg> whatif=<name>[,<name>]`
WHATIF = {
    # no global stores at all (the values stay live): what the LDS staging + conversions cost on their own
    "nostore": [("                        __builtin_amdgcn_raw_buffer_store_b128(packed, rs_c, (int)vo_c, soff(pi, p.ldc), 0);\n",
                 "                        asm volatile(\"\" ::\"v\"(packed), \"v\"(vo_c));\n")],
    # no idle slots behind the stores (the gfx950 store hazard is then live: results are wrong)
    "nonop": [('                        asm volatile("s_nop 7" ::"v"(packed), "v"(vo_c) : "memory");\n',
               '                        asm volatile("" ::"v"(packed), "v"(vo_c) : "memory");\n')],
    # no residual loads (the fp32 pass structure stays)
    "nores": [("                            rres[pi % RING][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)b, soff(pi, p.ldr), 0);\n",
               "                            rres[pi % RING][it] = u32x4_t{(unsigned)b, 0u, 0u, 0u};\n")],
}


def stamped_source(text: str, whatif=()) -> str:
    for w in whatif:
        for old, new in WHATIF[w]:
            assert text.count(old) == 1, f"build_gemm_timing: what-if anchor {old[:60]!r} occurs {text.count(old)} times"
            text = text.replace(old, new)
    for old, new in PATCHES:
        assert text.count(old) == 1, f"build_gemm_timing: anchor {old[:50]!r} occurs {text.count(old)} times - update the patch"
        text = text.replace(old, new)
    return text


def main():
    b.build()
    src = b.CSRC / "sdv_gemm.hip"
    wg = sys.argv[1] if len(sys.argv) > 1 else "0"
    whatif = [w for a in sys.argv[2:] if a.startswith("whatif=") for w in a[7:].split(",") if w]
    extra = [a for a in sys.argv[2:] if not a.startswith("whatif=")]          # extra -D flags
    here = Path(__file__).resolve().parent
    name = ("libsdv_gemm_timing%s.so" % "".join("_" + w for w in whatif)) if wg != "notiming" else \
        "libsdv_gemm_dbg%s.so" % "".join(c for c in "".join(extra) if c.isdigit())
    others = [b.OBJDIR / f"{s.stem}.o" for s in b.sources() if s.name != src.name]
    with tempfile.TemporaryDirectory() as tmp:
        patched = Path(tmp) / "sdv_gemm_timing.hip"
        patched.write_text(stamped_source(src.read_text(), whatif) if wg != "notiming" else src.read_text())
        obj = Path(tmp) / "gemm_timing.o"
        subprocess.run([b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS[src.name], "-I", str(b.CSRC),
                        *([f"-DSDV_GEMM_TIMING={wg}"] if wg != "notiming" else []), *extra, "-DSDV_GEMM_ONLY_TILE6", "-c", str(patched),
                        "-o", str(obj)], check=True)
        out = here / name
        subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", str(obj), *map(str, others), "-o", str(out)], check=True)
    print("built", out)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Timing-only variants of the attention kernel: each removes ONE cost from a patched COPY of csrc/sdv_attention.hip (the
product source carries none of this) - 1 exps, 2 barriers (racy), 3 row max, 4 K/V staging (one tile reused), 5 LDS stores
only after the first tile, 6 global loads only for the first tile, 7 no O stores, 8 Q rows broadcast (one address per wave).  Results are wrong by construction.  Builds
tools/ubench/libsdv_whatif{n}.so with the same C ABI; run e.g.
    SDV_HIP_LIB=tools/ubench/libsdv_whatif1.so python tools/attn_bench.py 64"""
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402

STAGE = """            __syncthreads();  // previous tile fully consumed (and the pad zeroing is visible)
            store_tile(0);
            __syncthreads();
            if (t + 1 < ntiles) load_tile((t + 1) * 64);
"""
EXPS = "pr[e] = pack_bf16x2(__builtin_amdgcn_exp2f(a[8 * u + 2 * e]), __builtin_amdgcn_exp2f(a[8 * u + 2 * e + 1]));"
ROWMAX = """                float mx = a[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, a[r]);
"""
PATCHES = {
    1: (EXPS, "pr[e] = pack_bf16x2(a[8 * u + 2 * e] * 0.5f, a[8 * u + 2 * e + 1] * 0.5f);"),
    2: (STAGE, "            store_tile(0);\n            if (t + 1 < ntiles) load_tile((t + 1) * 64);\n"),
    3: (ROWMAX, "                float mx = a[0];\n                return fmaxf(mx, b[3]);\n"),
    4: (STAGE, "            if (t == 0) {\n                __syncthreads();\n                store_tile(0);\n                __syncthreads();\n            }\n"),
    5: (STAGE, "            __syncthreads();\n            if (t == 0) store_tile(0);\n            __syncthreads();\n"
               "            if (t + 1 < ntiles) load_tile((t + 1) * 64);\n"
               "            if (t + 1 == ntiles) asm volatile(\"\" ::\"v\"(kreg[0]), \"v\"(vreg[0]));\n"),
    6: (STAGE, "            __syncthreads();\n            store_tile(0);\n            __syncthreads();\n"),
    # (7 / 8 were measured on the tree BEFORE the Q / O slabs - commit 14d0cf9; in today's source they only touch the direct-access
    #  path that dh = 64 keeps)
    # 7 / 8 (round 4, the Lk = 77 cross-attention): no O stores (one lane of one block keeps the values alive) / Q rows not loaded
    # per lane (every lane reads row 0 of its block - one address per wave instruction) - what the 32-rows-per-instruction
    # MFMA-layout accesses of the prologue and the epilogue cost the texture addresser
    7: ("                        *(uint2*)(orow + d) = w;", "                        if (q0 < -1) *(uint2*)(orow + d) = w;"),
    8: ("            int q = q0 + qt * 32 + l31;\n            q = q < Lq ? q : Lq - 1;", "            int q = q0 + qt * 32;\n            q = q < Lq ? q : Lq - 1;"),
}

b.build()
src = b.CSRC / "sdv_attention.hip"
text = src.read_text()
others = [b.OBJDIR / f"{s.stem}.o" for s in b.sources() if s.name != src.name]
here = Path(__file__).resolve().parent
for n in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else sorted(PATCHES):
    old, new = PATCHES[n]
    assert text.count(old) == 1, f"what-if {n}: the product source changed, update the patch"
    with tempfile.TemporaryDirectory() as tmp:
        patched = Path(tmp) / "sdv_attention_whatif.hip"
        patched.write_text(text.replace(old, new))
        obj = Path(tmp) / "attn.o"
        subprocess.run([b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS[src.name], "-I", str(b.CSRC), "-c", str(patched), "-o", str(obj)],
                       check=True)
        out = here / f"libsdv_whatif{n}.so"
        subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", str(obj), *map(str, others), "-o", str(out)], check=True)
    print("built", out)

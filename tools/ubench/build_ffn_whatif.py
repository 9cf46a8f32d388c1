#!/usr/bin/env python
"""TIMING-ONLY variants of the fused feed-forward kernel: compiles csrc/sdv_ffn.hip with -DSDV_FFN_WHATIF=<bits> (see the kernel) and
links it with the cached objects of the other kernel files into tools/ubench/libsdv_ffn_w<bits>.so, for
`SDV_HIP_LIB=tools/ubench/libsdv_ffn_w<bits>.so python tools/ffn_ab.py`.   usage: build_ffn_whatif.py <bits> [<bits> ...]"""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from stable_diffusion_videos_amd import build as b  # noqa: E402

b.build()
src = b.CSRC / "sdv_ffn.hip"
others = [str(b.OBJDIR / (s.stem + ".o")) for s in b.sources() if s.name != "sdv_ffn.hip"]
for bits in sys.argv[1:]:
    obj = Path(f"/tmp/sdv_ffn_w{bits}.o")
    cmd = [b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS.get("sdv_ffn.hip", []), *(f"-D{d}" if "=" in d else f"-DSDV_FFN_WHATIF={d}" for d in bits.split(",")),
           "-c", str(src), "-o", str(obj)]
    subprocess.run(cmd, check=True)
    out = ROOT / "tools" / "ubench" / f"libsdv_ffn_w{bits.replace(',', '_').replace('=', '')}.so"
    subprocess.run([b.hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", str(obj), *others, "-o", str(out)], check=True)
    print("built", out)

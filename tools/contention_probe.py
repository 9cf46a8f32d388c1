#!/usr/bin/env python
"""Which block of the UNet changes its output when ANOTHER process keeps the same GPU busy?
(tests/test_model_gpu.py::test_bench_launches_its_own_ranks: two ranks on one GPU - a frame generated while the other rank was
running differed, by a few uint8 steps and differently every time, from the same frame generated alone; one rank alone is
bit-reproducible; SDV_LN_FOLD=0 makes the difference go away.)
Foreground: the tiny UNet's eager forward with engine.TAP recording every block / sub-block output - once alone (baseline), then
RUNS times while a child process runs the same forward in a loop; per tap name the number of runs whose output differs from
the baseline, in forward order (the first name with a non-zero count is where it starts).
usage: python tools/contention_probe.py [runs] [arch] [noise arch]        (internal: ... noise <seconds> <arch>)"""
import subprocess
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, engine, hip  # noqa: E402

BF16 = torch.bfloat16


def setup(arch):
    name = {"tiny": "tiny", "sd14": "CompVis/stable-diffusion-v1-4"}[arch]
    pipe = StableDiffusionWalkPipeline.from_pretrained(name, arch=arch).to("cuda")
    B, h = 4, (16 if arch == "tiny" else 64)
    emb = pipe.embed_text(["a cat"] * B)
    ctx = torch.cat([pipe._uncond_embeddings(None, B), emb.float()])
    pipe._schedule(50, 0.0)
    pipe.unet.prepare_context(ctx)
    g = torch.Generator(device="cuda").manual_seed(3)
    x2 = torch.randn((2 * B * h * h, 4), device="cuda", generator=g).to(BF16)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    return pipe, x2, B, h, step


def kernel_noise(seconds, kind):
    """ONE kind of kernel in a loop (which resource of the CU does the co-resident process have to use for the fault to show?):
    attn = flash attention (v_exp-heavy, little LDS-DMA), conv = the implicit-GEMM conv (LDS-DMA + MFMA, no transcendental),
    gn = GroupNorm apply (HBM streaming, no LDS), geglu = the 8-wave GEGLU GEMM (MFMA + transcendentals in the epilogue)"""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *sh: (torch.randn(sh, device=dev, generator=g) * 0.5).to(BF16)
    if kind == "attn":
        B, H, L, dh = 8, 8, 4096, 40
        q, k, v, o = rnd(B * L, H * dh), rnd(B * L, H * dh), rnd(B * L, H * dh), rnd(B * L, H * dh)
        fn = lambda: hip.attention(q, k, v, o, B=B, H=H, Lq=L, Lk=L, dh=dh, ldq=H * dh, ldk=H * dh, ldv=H * dh, ldo=H * dh, scale=dh ** -0.5, v_rowmajor=True)
    elif kind == "conv":
        nimg, Hh, C = 8, 64, 320
        x, w, b = rnd(nimg * Hh * Hh, C), rnd(C, 9 * C), torch.zeros(C, device=dev)
        out = torch.empty_like(x)
        fn = lambda: hip.conv3x3(x, w, b, nimg=nimg, H=Hh, W=Hh, out=out)
    elif kind == "gn":
        nimg, HW, C = 8, 4096, 320
        x, gm, bt = rnd(nimg * HW, C), torch.ones(C, device=dev), torch.zeros(C, device=dev)
        fn = lambda: hip.groupnorm(x, gm, bt, nimg=nimg, HW=HW, groups=32, eps=1e-5, silu=True)
    elif kind in ("tile1", "tile2", "tile3"):           # the 4-wave small tiles on a low-resolution shape (plain epilogue)
        M, C = 2048, 1280
        x, w, b = rnd(M, C), rnd(C, C), torch.zeros(C, device=dev)
        fn = lambda: hip.linear(x, w, b, tile=int(kind[-1]))
    elif kind == "splitk":                               # 8 x 8-level conv, split-K first pass + reduce
        nimg, Hh, C = 8, 8, 1280
        x, w, b = rnd(nimg * Hh * Hh, C), rnd(C, 9 * C), torch.zeros(C, device=dev)
        fn = lambda: hip.conv3x3(x, w, b, nimg=nimg, H=Hh, W=Hh)
    elif kind == "xattn":                                # text cross-attention (resident K / V form)
        B, H, L, dh = 8, 8, 4096, 40
        q, k, vt, o = rnd(B * L, H * dh), rnd(B * 77, H * dh), rnd(B, H * dh, 128), rnd(B * L, H * dh)
        fn = lambda: hip.attention(q, k, vt, o, B=B, H=H, Lq=L, Lk=77, dh=dh, ldq=H * dh, ldk=H * dh, ldv=128, ldo=H * dh, scale=dh ** -0.5)
    elif kind == "ffn":                                  # the fused feed-forward panel kernel (one 512-register wave per SIMD, 160 KB of LDS)
        from stable_diffusion_videos_amd.weights import ffn_fold_columns, ffn_w2_permute
        M, C = 32768, 320
        x, w1, w2 = rnd(M, C), rnd(8 * C, C), rnd(C, 4 * C)
        xf = x.float()
        st = torch.stack([xf.mean(1), torch.rsqrt(xf.var(1, unbiased=False) + 1e-5)], 1).contiguous()
        w1x, w2p, b2 = ffn_fold_columns(torch.zeros(8 * C, device=dev), torch.zeros(8 * C, device=dev)), ffn_w2_permute(w2), torch.zeros(C, device=dev)
        fn = lambda: hip.ffn_geglu(x, st, w1, w1x, w2p, b2)
    elif kind == "t1geglu":                              # the victim's own kind: fold + GEGLU on a small shape (the library decides the tile)
        M, C = 2048, 640
        x, w, b = rnd(M, C), rnd(8 * C, C), torch.zeros(8 * C, device=dev)
        xf = x.float()
        st = torch.stack([xf.mean(1), torch.rsqrt(xf.var(1, unbiased=False) + 1e-5)], 1).contiguous()
        sv = torch.zeros(8 * C, device=dev)
        fn = lambda: hip.linear(x, w, b, epi=1, ln=(st, sv))
    else:
        M, C = 32768, 640
        x, w, b = rnd(M, C), rnd(8 * C, C), torch.zeros(8 * C, device=dev)
        fn = lambda: hip.linear(x, w, b, epi=1, tile=6)
    fn()
    torch.cuda.synchronize()
    print("ready", flush=True)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()


def noise(seconds, arch):
    if arch in ("attn", "conv", "gn", "geglu", "tile1", "tile2", "tile3", "splitk", "xattn", "t1geglu", "ffn"):
        return kernel_noise(seconds, arch)
    pipe, x2, B, h, step = setup(arch)
    pipe.unet.forward(x2, 2 * B, h, h, step, cfg_shared=True)
    torch.cuda.synchronize()
    print("ready", flush=True)          # (tests/test_contention_gpu.py waits for this line before it starts comparing)
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(5):
            pipe.unet.forward(x2, 2 * B, h, h, step, cfg_shared=True)
        torch.cuda.synchronize()


def tapped_forward(pipe, x2, B, h, step):
    rec = []
    engine.TAP = lambda name, d: rec.append((f"{name}:{d['kind']}", d["out"].detach().clone()))
    engine.TAP_AUX = lambda name, t: rec.append((name, t.detach().clone()))
    try:
        eps = pipe.unet.forward(x2, 2 * B, h, h, step, cfg_shared=True)
    finally:
        engine.TAP = engine.TAP_AUX = None
    torch.cuda.synchronize()
    rec.append(("eps", eps.clone()))
    return rec


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    arch = sys.argv[2] if len(sys.argv) > 2 else "tiny"
    pipe, x2, B, h, step = setup(arch)
    base = tapped_forward(pipe, x2, B, h, step)
    again = tapped_forward(pipe, x2, B, h, step)
    alone = sum(not torch.equal(a[1], b[1]) for a, b in zip(base, again))
    print(f"{arch}: {len(base)} taps; alone, second forward vs first: {alone} taps differ", flush=True)
    noise_arch = sys.argv[3] if len(sys.argv) > 3 else arch
    child = subprocess.Popen([sys.executable, __file__, "noise", "600", noise_arch], stdout=subprocess.PIPE, text=True)
    while "ready" not in child.stdout.readline():
        pass
    counts = [0] * len(base)
    dumped = 0
    firsts = {}
    try:
        for r in range(runs):
            rec = tapped_forward(pipe, x2, B, h, step)
            first = None
            for i, ((n0, a), (n1, b)) in enumerate(zip(base, rec)):
                if not torch.equal(a, b):
                    counts[i] += 1
                    if first is None:
                        first = n0
                        if dumped < 6:          # where and by how much: the pattern says which instruction
                            dumped += 1
                            a2, b2 = a.reshape(-1, a.shape[-1]).float(), b.reshape(-1, b.shape[-1]).float()
                            bad = (a2 != b2).nonzero()
                            rows, cols = bad[:, 0].tolist(), bad[:, 1].tolist()
                            print(f"  run {r}: {n0} shape {tuple(a2.shape)}: {len(rows)} elements differ; rows {sorted(set(rows))[:40]} cols {sorted(set(cols))[:40]}")
                            for (rr, cc) in list(zip(rows, cols))[:12]:
                                print(f"      [{rr}, {cc}] alone {float(a2[rr, cc]):+.5f}  now {float(b2[rr, cc]):+.5f}")
            if first is not None:
                firsts[first] = firsts.get(first, 0) + 1
    finally:
        child.kill()
    print(f"with a second process running the same forward: runs whose FIRST differing tap is ... {firsts or 'none: all runs identical'}")
    for (n, _), c in zip(base, counts):
        if c:
            print(f"  {n:70s} differs in {c}/{runs} runs")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "noise":
        noise(float(sys.argv[2]), sys.argv[3])
    else:
        main()

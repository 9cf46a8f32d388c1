#!/usr/bin/env python
"""Is the tiny VAE decode independent of the igemm tile (cost model vs forced 256 x 320), block by block - and, inside the first block
that is not, which intermediate or which epilogue statistics tensor differs?  Found the re-associated GroupNorm statistics of the
phase-form up-conv (DESIGN.md status row 1b).  usage: python tools/vae_tile_probe.py [latents.pt]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, "/root/repo")
from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, engine, hip
pipe = StableDiffusionWalkPipeline.from_pretrained("tiny", arch="tiny").to("cuda")
g = torch.Generator(device="cuda").manual_seed(1)
lat = torch.randn((4, 16, 16, 4), device="cuda", generator=g) * 0.18215
dbg = Path(sys.argv[1]) if len(sys.argv) > 1 else Path("/nonexistent")     # optional: latents [B, 4, h, w] saved by a failing run
if dbg.exists():
    lat = hip.nchw_to_nhwc(torch.load(dbg).cuda())
    print("latents from", dbg, tuple(lat.shape), "abs max", float(lat.abs().max()), "std", float(lat.std()))


def run(tile):
    rec = []
    hip.FORCE_TILE = tile
    engine.TAP = lambda name, d: rec.append((f"{name}:{d['kind']}", d["out"].detach().clone(), getattr(d["out"], "_sdv_gn", None) is not None))
    try:
        u8, _ = pipe.vae.decode(lat)
    finally:
        engine.TAP = None
        hip.FORCE_TILE = 0
    torch.cuda.synchronize()
    rec.append(("u8", u8.clone(), False))
    return rec


a, b = run(0), run(6)
for (n, x, gx), (_, y, gy) in zip(a, b):
    same = torch.equal(x, y)
    print(f"{n:60s} {'same' if same else 'DIFFERENT max|d| %.4g' % float((x.float() - y.float()).abs().max())}   epilogue stats: tile0 {gx} tile6 {gy}")

# ---- inside the first differing block: every intermediate (and the epilogue statistics) under both tile choices -------------------
prev = dict((n, x) for n, x, _ in a)["decoder.up_blocks.2.resnets.2:resnet"]
blk = pipe.vae.up[2]
wu, bu = blk["up"]
r = pipe.vae.up[3]["res"][0]
nimg, H = 4, 64                      # tiny VAE: 16 -> 32 -> 64 -> 128; up_blocks.2's upsampler takes 64 x 64 to 128 x 128
res = {}
for tile in (0, 6):
    hip.FORCE_TILE = tile
    x = hip.upconv3x3_phase(prev, wu, bu, nimg=nimg, H=H, W=H, gn=True)
    HW = 4 * H * H
    h1 = hip.groupnorm(x, r.g1, r.b1, nimg=nimg, HW=HW, groups=r.groups, eps=r.eps, silu=True)
    c1 = hip.conv3x3(h1, r.w1, r.c1_bias, nimg=nimg, H=2 * H, W=2 * H, gn=True)
    h2 = hip.groupnorm(c1, r.g2, r.b2, nimg=nimg, HW=HW, groups=r.groups, eps=r.eps, silu=True)
    sc = hip.linear(x, r.ws, r.bs) if r.ws is not None else x
    out = hip.conv3x3(h2, r.w2, r.c2_bias, nimg=nimg, H=2 * H, W=2 * H, residual=sc, gn=True)
    torch.cuda.synchronize()
    res[tile] = dict(x=x, x_stats=x._sdv_gn.p, h1=h1, c1=c1, c1_stats=c1._sdv_gn.p, h2=h2, sc=sc, out=out, out_stats=out._sdv_gn.p)
hip.FORCE_TILE = 0
print("prev", tuple(prev.shape), "abs max", float(prev.float().abs().max()), " w1", tuple(r.w1.shape), " w2", tuple(r.w2.shape))
for k in res[0]:
    x, y = res[0][k], res[6][k]
    same = torch.equal(x, y)
    extra = "" if same else f"  DIFFERENT: {int((x != y).sum())} of {x.numel()} elements, max|d| {float((x.float() - y.float()).abs().max()):.6g}, max|x| {float(x.float().abs().max()):.6g}"
    print(f"  {k:10s} {'same' if same else ''}{extra}")

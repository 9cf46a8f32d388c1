import sys, os, subprocess, torch
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/stable_diffusion_videos_amd') else os.environ.get('GRAFT_REPO_ROOT','.'))
from stable_diffusion_videos_amd import hip
mode = sys.argv[1]          # 'dump' or 'check'
tag = sys.argv[2]
dev = torch.device('cuda')
hip.load()
torch.manual_seed(0)
res = {}
for nimg in (8, 256):
    for dh, L, heads in ((40, 4096, 8), (80, 1024, 8), (160, 256, 8)):
        C = dh * heads
        g = torch.Generator(device=dev).manual_seed(dh + nimg)
        qk = torch.randn((nimg * L, 2 * C), device=dev, generator=g).to(torch.bfloat16)
        vt = torch.randn((nimg, C, L), device=dev, generator=g).to(torch.bfloat16)
        for kind in ('self', 'cross'):
            if kind == 'self':
                o = torch.empty((nimg * L, C), dtype=torch.bfloat16, device=dev)
                fn = lambda: hip.attention(qk, qk, vt, o, B=nimg, H=heads, Lq=L, Lk=L, dh=dh, ldq=2 * C, ldk=2 * C, ldv=L, ldo=C, scale=dh ** -0.5, k_off=C)
            else:
                q = qk[:, :C].contiguous()
                k = torch.randn((nimg * 77, C), device=dev, generator=g).to(torch.bfloat16)
                v2 = torch.zeros((nimg, C, 128), dtype=torch.bfloat16, device=dev)
                v2[:, :, :77] = torch.randn((nimg, C, 77), device=dev, generator=g).to(torch.bfloat16)
                o = torch.empty((nimg * L, C), dtype=torch.bfloat16, device=dev)
                fn = lambda: hip.attention(q, k, v2, o, B=nimg, H=heads, Lq=L, Lk=77, dh=dh, ldq=C, ldk=C, ldv=128, ldo=C, scale=dh ** -0.5)
            fn(); torch.cuda.synchronize(); a = o.clone()
            ndiff = 0
            for _ in range(3):
                fn(); torch.cuda.synchronize()
                ndiff += int((o != a).sum())
            key = f"{kind}_dh{dh}_n{nimg}"
            res[key] = a.cpu()
            print(tag, key, 'run-to-run differing elements:', ndiff, 'nan:', int(torch.isnan(a.float()).sum()), flush=True)
if mode == 'dump':
    torch.save(res, f'/tmp/attn_{tag}.pt')
else:
    ref = torch.load(f'/tmp/attn_{sys.argv[3]}.pt')
    for k2, v in res.items():
        d = (v.float() - ref[k2].float()).abs()
        print('compare', k2, 'differing elements:', int((v != ref[k2]).sum()), 'of', v.numel(), 'max abs diff', float(d.max()), flush=True)

"""Generate the golden vectors under tests/golden/ by EXECUTING THE REFERENCE in the build
container (it cannot travel to the GPU box, so the outputs are committed as fixtures).

What can be executed: ``slerp`` from /root/reference/stable_diffusion_videos/utils.py:42-66.  The
module itself cannot be imported (``import librosa`` at utils.py:4 and torchvision at :8-9 are not
installed), so the ``slerp`` FunctionDef is lifted by AST and exec'd with only {np, torch} in scope -
the function body that runs is the reference's own, byte for byte.

Nothing else on the hot path can be run from the reference here: diffusers (UNet/VAE/DDIM) is not
installed and is un-vendored (pyproject.toml:14) -> those parts stay "parity unpinned".

Run:  python tests/golden/make_golden.py        (needs /root/reference; writes *.npz next to itself)
"""
import ast
import pathlib

import numpy as np
import torch

REF = pathlib.Path("/root/reference/stable_diffusion_videos/utils.py")
OUT = pathlib.Path(__file__).parent


def lift_reference_slerp():
    tree = ast.parse(REF.read_text())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "slerp"]
    assert len(fn) == 1 and fn[0].lineno == 42, "reference slerp moved"
    mod = ast.Module(body=fn, type_ignores=[])
    scope = {"np": np, "torch": torch}
    exec(compile(mod, str(REF), "exec"), scope)
    return scope["slerp"]


def noise(seed, shape, dtype=torch.float32):
    # stable_diffusion_pipeline.py:832-837 with a CPU generator
    return torch.randn(shape, generator=torch.Generator(device="cpu").manual_seed(seed), dtype=dtype)


def main():
    ref_slerp = lift_reference_slerp()
    ts = [0.0, 0.25, 0.5, 0.75, 1.0]
    # 1. the BASELINE config-1 endpoints: seeds 42 / 1337, (1,4,64,64) fp32
    v0, v1 = noise(42, (1, 4, 64, 64)), noise(1337, (1, 4, 64, 64))
    outs = {f"t{int(t * 100):03d}": ref_slerp(t, v0, v1).numpy() for t in ts}
    np.savez(OUT / "slerp_seed42_1337_fp32.npz", v0=v0.numpy(), v1=v1.numpy(), ts=np.array(ts), **outs)
    # 2. fp16 in -> fp16 out (reference computes the reductions in fp16 too), small latent
    h0, h1 = noise(7, (1, 4, 16, 16)).half(), noise(8, (1, 4, 16, 16)).half()
    outs = {f"t{int(t * 100):03d}": ref_slerp(t, h0, h1).numpy() for t in ts}
    np.savez(OUT / "slerp_seed7_8_fp16.npz", v0=h0.numpy(), v1=h1.numpy(), ts=np.array(ts), **outs)
    # 3. nearly parallel endpoints -> the |dot| > 0.9995 lerp branch (utils.py:52-53)
    p0 = noise(3, (1, 4, 8, 8))
    p1 = p0 + 1e-3 * noise(4, (1, 4, 8, 8))
    outs = {f"t{int(t * 100):03d}": ref_slerp(t, p0, p1).numpy() for t in ts}
    np.savez(OUT / "slerp_parallel_fp32.npz", v0=p0.numpy(), v1=p1.numpy(), ts=np.array(ts), **outs)
    # 4. anti-parallel (dot < -0.9995) also takes the lerp branch via abs()
    a1 = -p0 + 1e-3 * noise(5, (1, 4, 8, 8))
    outs = {f"t{int(t * 100):03d}": ref_slerp(t, p0, a1).numpy() for t in ts}
    np.savez(OUT / "slerp_antiparallel_fp32.npz", v0=p0.numpy(), v1=a1.numpy(), ts=np.array(ts), **outs)
    # 5. numpy (non-torch) inputs, float64 - the function's other entry form
    rng = np.random.default_rng(0)
    n0, n1 = rng.standard_normal((4, 8, 8)), rng.standard_normal((4, 8, 8))
    outs = {f"t{int(t * 100):03d}": ref_slerp(t, n0, n1) for t in ts}
    np.savez(OUT / "slerp_numpy_fp64.npz", v0=n0, v1=n1, ts=np.array(ts), **outs)
    print("wrote", sorted(p.name for p in OUT.glob("*.npz")))


if __name__ == "__main__":
    main()

"""CPU tests of the host side: the C ABI loads and exports every symbol include/sdv_hip.h declares,
weight schema / re-layout, the scheduler's fused coefficients, frame partitioning, tokenizer stub, and
that the product path fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def test_c_abi_exports_every_declared_symbol(hip):
    header = (ROOT / "include" / "sdv_hip.h").read_text()
    declared = set(re.findall(r"\b(sdv_[a-z0-9_]+)\s*\(", header))
    declared -= {"sdv_gemm_args"}
    assert len(declared) >= 20
    lib = ctypes.CDLL(str(hip.lib_path()))
    for name in sorted(declared):
        assert hasattr(lib, name), f"libsdv_hip.so does not export {name}"
    assert declared == set(hip.EXPORTED_SYMBOLS), declared ^ set(hip.EXPORTED_SYMBOLS)
    assert hip.load().sdv_abi_version() == hip.ABI_VERSION == 12


def test_gemm_args_struct_matches_header(hip):
    """Field order of the ctypes mirror == field order of struct sdv_gemm_args in the header."""
    header = (ROOT / "include" / "sdv_hip.h").read_text()
    body = header[header.index("typedef struct sdv_gemm_args {"):header.index("} sdv_gemm_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        decl = stmt.split()
        rest = stmt[len(decl[0]):] if decl[0] != "const" else stmt[len("const " + decl[1]):]
        names += [n.strip(" *") for n in rest.split(",")]
    assert names == [f[0] for f in hip.GemmArgs._fields_]


def test_argument_validation_without_gpu(hip):
    """Host-side argument checks run before any launch, so they are testable without a GPU."""
    lib = hip.load()
    a = hip.GemmArgs()
    assert lib.sdv_gemm_bf16(ctypes.byref(a), None) == -1
    assert b"null operand" in lib.sdv_last_error()
    a.X = a.W = a.C = 1
    a.M = a.N = 64
    a.K = 96
    assert lib.sdv_gemm_bf16(ctypes.byref(a), None) == -1 and b"multiple of 64" in lib.sdv_last_error()
    assert lib.sdv_attention_bf16(1, 1, 1, 1, 1, 1, 64, 64, 48, 64, 64, 64, 64, 1.0, 0, 0, 0, None) == -1
    assert b"unsupported head dim" in lib.sdv_last_error()
    # row-major V (ABI 10): ldv is a ROW stride and must cover the H * dh columns of a token
    assert lib.sdv_attention_bf16(16, 16, 16, 16, 1, 8, 64, 64, 40, 960, 960, 64, 320, 1.0, 0, 1, 1, None) == -1
    assert b"row-major V" in lib.sdv_last_error()
    a.K = 64
    a.tile = 14                                                 # the transposed tile of ABI 9
    assert lib.sdv_gemm_bf16(ctypes.byref(a), None) == -1 and b"bad tile" in lib.sdv_last_error()
    a.tile, a.ln_side, a.ln_stats, a.ln_s = 0, 2, 16, 16       # the column-side LayerNorm fold of ABI 9
    assert lib.sdv_gemm_bf16(ctypes.byref(a), None) == -1 and b"ln_side" in lib.sdv_last_error()


def test_ffn_argument_validation_without_gpu(hip):
    """sdv_ffn_geglu_bf16 rejects what it was not built for before it touches the device (C != 320, odd leading dimensions, nulls)"""
    lib = hip.load()
    ok = [16, 16, 4096, 320, 320, 16, 16, 16, 16, 16, 320, None]
    for pos, bad, word in ((3, 640, b"C = 320"), (4, 324, b"ldx"), (0, None, b"null"), (10, 100, b"ldo"), (2, 0, b"bad M")):
        a = list(ok)
        a[pos] = bad
        assert lib.sdv_ffn_geglu_bf16(*a) == -1 and word in lib.sdv_last_error(), (pos, lib.sdv_last_error())


def test_linear320_argument_validation_without_gpu(hip):
    """sdv_linear320_bf16: N is 320 or 960, a residual excludes fold / alpha / N = 960, statistics exist for N = 320 only"""
    lib = hip.load()
    ok = dict(X=16, M=4096, ldx=320, W=16, Wx=16, N=320, ln_stats=None, alpha=None, R=None, ldr=0, out=16, ldo=320, stats_out=None, eps=1e-5,
              Vt=None, ldvt=0, hw=0, stream=None)
    for change, word in ((dict(N=640), b"320 or 960"), (dict(R=16, ldr=320, ln_stats=16), b"residual"), (dict(N=960, ldo=960, stats_out=16), b"statistics"),
                         (dict(N=960, ldo=320), b"leading"), (dict(W=None), b"null"), (dict(X=18), b"unaligned"),
                         (dict(Vt=16, ldo=640, ldvt=4096, hw=4096), b"V^T"), (dict(N=960, ldo=640, Vt=16, ldvt=4096, hw=4000), b"V^T"),
                         (dict(N=960, ldo=640, Vt=16, ldvt=4096, hw=4096, M=4096 + 128), b"V^T")):
        a = dict(ok)
        a.update(change)
        assert lib.sdv_linear320_bf16(*a.values()) == -1 and word in lib.sdv_last_error(), (change, lib.sdv_last_error())


def test_linear640_argument_validation_without_gpu(hip):
    """sdv_linear640_bf16: N is 640 or 1920, statistics exist for N = 640 only, rows of 640 columns"""
    lib = hip.load()
    ok = dict(X=16, M=4096, ldx=640, W=16, Wx=16, N=640, ln_stats=None, alpha=None, out=16, ldo=640, stats_out=None, eps=1e-5, stream=None)
    for change, word in ((dict(N=320), b"640 or 1920"), (dict(N=1920, ldo=1920, stats_out=16), b"statistics"), (dict(ldx=320), b"leading"),
                         (dict(N=1920, ldo=640), b"leading"), (dict(out=None), b"null"), (dict(W=24), b"unaligned"), (dict(M=0), b"bad M")):
        a = dict(ok)
        a.update(change)
        assert lib.sdv_linear640_bf16(*a.values()) == -1 and word in lib.sdv_last_error(), (change, lib.sdv_last_error())


def test_split_k_planning_without_gpu(hip):
    """sdv_gemm_split_k is pure host logic (sdv_hip.h "split_k"): the small-batch shapes of the UNet split, everything the second pass
    cannot finish - or that fills the chip by itself - does not.  Shapes: ResBlock conv3x3 1280 -> 1280 at the 8 x 8 level of a 1 / 4 /
    16 / 128-frame call (M = 2 B x 64 rows), ff.net.2 of the mid block, a K = 320 projection."""
    lib = hip.load()

    def plan(M, N, K, mode=0, **kw):
        a = hip.GemmArgs()
        a.X = a.W = a.C = 16
        a.M, a.N, a.K, a.ldx, a.ldw, a.ldc, a.mode = M, N, K, K, K * (9 if mode else 1), N, mode
        if mode:
            a.Hin = a.Win = a.Hout = a.Wout = 8
        a.split_k = 8
        for k, v in kw.items():
            setattr(a, k, v)
        s = lib.sdv_gemm_split_k(ctypes.byref(a))
        assert s >= 1, lib.sdv_last_error()
        return s

    assert plan(128, 1280, 1280, mode=1) >= 4          # 1 frame per call: a handful of tiles, 180 K slabs each
    assert plan(512, 1280, 1280, mode=1) >= 2          # 4 frames per call
    assert plan(2048, 1280, 1280, mode=1) == 1         # 16 frames per call: 160 tiles of 128 x 128 - too many to split
    assert plan(16384, 1280, 1280, mode=1) == 1        # 128 frames per call: the 256 x 320 tile, one per CU
    assert plan(128, 1280, 5120) >= 2                  # ff.net.2 of the 8 x 8 block
    assert plan(128, 1280, 320) == 1                   # 5 K slabs: nothing to share out
    assert plan(128, 1280, 5120, epi=2) == 1           # an activation: the second pass only knows alpha / bias / residual
    assert plan(128, 1280, 5120, batch=2) == 1 and plan(128, 1280, 5120, out_mode=1, out_f32=16) == 1
    assert plan(128, 1280, 5120, ln_side=1, ln_stats=16, ln_s=16) == 1 and plan(128, 1280, 5120, stats_out=16) == 1
    assert plan(128, 1280, 5120, tile=6) == 1          # a forced 8-wave tile (tests: hip.FORCE_TILE) never splits
    # the second pass scales EVERY column by alpha and moves C / R as 8-byte pieces: a C caller that asks for alpha on the leading
    # columns only (a fused [Q | K] projection), or hands over a 2-byte aligned output, gets an unsplit launch, not a wrong one
    assert plan(128, 1280, 5120, alpha=0.125, alpha_cols=640) == 1 and plan(128, 1280, 5120, alpha=0.125) >= 2
    assert plan(128, 1280, 5120, C=18) == 1 and plan(128, 1280, 5120, R=18, ldr=1280) == 1 and plan(128, 1280, 5120, R=16, ldr=1280) >= 2
    for M in (64, 128, 512, 1024):                     # never fewer than 16 K slabs per split, never more than 8 splits
        s = plan(M, 1280, 1280, mode=1)
        assert s <= 8 and 180 // s >= 16


def test_product_fails_loudly_without_gpu(hip):
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, slerp
    with pytest.raises(hip.SdvHipError, match="GPU"):
        hip.linear(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))
    with pytest.raises(hip.SdvHipError):
        slerp(0.5, torch.zeros(4), torch.ones(4))
    if not torch.cuda.is_available():
        pipe = StableDiffusionWalkPipeline.from_pretrained("tiny")
        with pytest.raises(hip.SdvHipError, match="cuda"):
            pipe(prompt="a cat", height=64, width=64)


def test_product_never_imports_oracle():
    for py in (ROOT / "stable_diffusion_videos_amd").glob("*.py"):
        src = py.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), py
        assert "from oracle" not in src and "import oracle" not in src, py


def test_weight_schema_and_relayout():
    from stable_diffusion_videos_amd import config, weights
    shapes = weights.unet_shapes(config.sd14_unet())
    assert weights.count_params(shapes) == 859_520_964 and len(shapes) == 686
    assert weights.count_params(weights.unet_shapes(config.sd21_unet())) == 865_910_724
    assert weights.count_params(weights.vae_decoder_shapes(config.sd_vae())) == 49_490_199
    assert shapes["up_blocks.1.resnets.2.conv1.weight"] == (1280, 1920, 3, 3)
    assert shapes["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (320, 768)
    sd = weights.synthetic_state_dict(weights.unet_shapes(config.tiny_unet()), seed=3)
    sd2 = weights.synthetic_state_dict(weights.unet_shapes(config.tiny_unet()), seed=3)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)
    w = sd["conv_in.weight"]
    assert torch.equal(w, w.to(torch.bfloat16).float())          # bf16-exact
    assert abs(float(sd["conv_norm_out.weight"].mean()) - 1.0) < 0.1
    # OHWI relayout: row n, column (ky*3+kx)*Cin + c
    w = torch.arange(2 * 3 * 3 * 3, dtype=torch.float32).reshape(2, 3, 3, 3)
    r = weights.conv_w(w, "cpu").float()
    assert r.shape == (2, 27) and float(r[1, (2 * 3 + 1) * 3 + 2]) == float(w[1, 2, 2, 1])
    # GEGLU interleave: every 32-row tile = [16 value rows | their 16 gate rows]
    t = torch.arange(128, dtype=torch.float32)[:, None].repeat(1, 2)
    g = weights.geglu_interleave(t)
    assert g[:16, 0].tolist() == list(range(0, 16)) and g[16:32, 0].tolist() == list(range(64, 80))
    assert g[32:48, 0].tolist() == list(range(16, 32)) and g[48:64, 0].tolist() == list(range(80, 96))


def test_upconv_phase_weights_reproduce_upsample_conv():
    """weights.upconv_phase_w: the four 2x2 phase filters applied to the LOW-resolution image equal nearest-2x upsampling
    followed by the 3x3 conv (reference: diffusers Upsample2D inside unet(...) / vae.decode(...),
    stable_diffusion_pipeline.py:418, :433) - checked in fp64 with plain torch ops."""
    import torch.nn.functional as F
    from stable_diffusion_videos_amd.weights import upconv_phase_w
    g = torch.Generator().manual_seed(0)
    co, ci, H, W = 6, 5, 7, 9
    w = torch.randn((co, ci, 3, 3), generator=g, dtype=torch.float64)
    x = torch.randn((2, ci, H, W), generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
    w4 = upconv_phase_w(w, "cpu").to(torch.float64).reshape(2, 2, co, 2, 2, ci)         # bf16-rounded phase filters
    w4x = torch.empty_like(w4)                                                           # exact (unrounded) phase filters
    groups = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
    for py in (0, 1):
        for px in (0, 1):
            for ty in (0, 1):
                for tx in (0, 1):
                    w4x[py, px, :, ty, tx, :] = w[:, :, groups[py][ty], :][:, :, :, groups[px][tx]].sum(dim=(2, 3))
    xp = F.pad(x, (1, 1, 1, 1))
    for filt, tol in ((w4x, 1e-12), (w4, 2e-2)):
        out = torch.empty_like(ref)
        for py in (0, 1):
            for px in (0, 1):
                k = filt[py, px].permute(0, 3, 1, 2)                                      # [co, ci, 2, 2]
                out[:, :, py::2, px::2] = F.conv2d(xp[:, :, py:py + H + 1, px:px + W + 1], k)
        assert float((out - ref).abs().max()) <= tol * float(ref.abs().max())


def test_scheduler_coefficients_equal_oracle_step():
    from oracle.scheduler import DDIMScheduler as OracleDDIM
    from stable_diffusion_videos_amd.scheduler import DDIMScheduler
    for ptype in ("epsilon", "v_prediction"):
        for eta in (0.0, 0.7):
            o, s = OracleDDIM(prediction_type=ptype), DDIMScheduler(prediction_type=ptype)
            o.set_timesteps(20)
            s.set_timesteps(20)
            assert o.timesteps.tolist() == s.timesteps.tolist()
            tab = s.coefficient_table(eta)
            assert tab.shape == (20, 4)
            x, e, z = torch.randn(64), torch.randn(64), torch.randn(64)
            for i, t in enumerate(o.timesteps):
                ref = o.step(e, t, x, eta=eta, variance_noise=z)
                got = tab[i, 0] * x + tab[i, 1] * e + tab[i, 2] * z
                assert torch.allclose(got, ref, atol=2e-5, rtol=1e-5), (ptype, eta, i)
    with pytest.raises(ValueError):
        DDIMScheduler().set_timesteps(0)


@pytest.mark.parametrize("ptype", ["epsilon", "v_prediction"])
def test_coefficient_table_reproduces_the_forward_marginals(ptype):
    """The fused kernel applies x <- c_x x + c_e out per step with the PRODUCT scheduler's table: fed the true noise, the 50
    steps of the BASELINE schedule must walk x_t = sqrt(a_t) x0 + sqrt(1 - a_t) eps down the marginals to t = 0 (DDIM with
    eta = 0, Song et al. 2021 eq. 12) - a property of the published sampler, independent of the oracle restatement."""
    from stable_diffusion_videos_amd.scheduler import DDIMScheduler
    s = DDIMScheduler(prediction_type=ptype)
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(981, 0, -20))
    tab = s.coefficient_table(0.0).double()
    a = s.alphas_cumprod.double()
    g = torch.Generator().manual_seed(5)
    x0, eps = torch.randn(64, generator=g, dtype=torch.float64), torch.randn(64, generator=g, dtype=torch.float64)
    x = a[981].sqrt() * x0 + (1 - a[981]).sqrt() * eps
    for i, t in enumerate(s.timesteps.tolist()):
        out = eps if ptype == "epsilon" else a[t].sqrt() * eps - (1 - a[t]).sqrt() * x0
        x = tab[i, 0] * x + tab[i, 1] * out
        a_p = a[t - 20] if t - 20 >= 0 else a[0]
        assert torch.allclose(x, a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps, atol=2e-5), t      # (the table is fp32)
    assert float(tab[:, 2].abs().max()) == 0.0                                                  # eta = 0: no noise term


def test_partition_frames_covers_every_frame_once():
    from stable_diffusion_videos_amd.parallel import partition_frames
    for counts, skips in (([60], None), ([80, 80, 80], None), ([12, 6, 18], [3, 0, 17]), ([3, 3], None), ([1], None)):
        for ws in (1, 2, 3, 4, 8):
            seen = []
            sizes = []
            for r in range(ws):
                share = partition_frames(counts, ws, r, skips)
                n = 0
                for clip, a, b in share:
                    assert 0 <= a < b <= counts[clip]
                    seen += [(clip, k) for k in range(a, b)]
                    n += b - a
                sizes.append(n)
            sk = skips or [0] * len(counts)
            expect = [(i, k) for i, c in enumerate(counts) for k in range(sk[i], c)]
            assert seen == expect                  # contiguous, ordered, disjoint, complete
            assert max(sizes) - min(sizes) <= 1    # balanced


def test_partition_frame_list_with_holes():
    """Resume after a multi-rank crash: rank 0 died at frame 20 of its block [0, 50), rank 1 at frame 70 of
    [50, 100): the work list is the set of MISSING frames, and every one of them is generated exactly once."""
    from stable_diffusion_videos_amd.parallel import partition_frame_list
    todo = [[k for k in range(100) if not (k <= 20 or 50 <= k <= 70)], [], list(range(3))]
    seen = []
    for rank in range(3):
        for clip, a, b in partition_frame_list(todo, 3, rank):
            seen += [(clip, k) for k in range(a, b)]
    assert sorted(seen) == sorted((c, k) for c, fr in enumerate(todo) for k in fr) and len(seen) == len(set(seen))
    assert (0, 21) in seen and (0, 49) in seen and (0, 71) in seen and (0, 20) not in seen


def test_no_silent_synthetic_fallback_and_scheduler_error(tmp_path):
    """from_pretrained() of something that is neither a local directory nor an explicit synthetic request raises (ADVICE:
    a hub id or typo used to return a pipeline of random weights silently); non-DDIM schedulers get a clear error."""
    from types import SimpleNamespace
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline as P
    with pytest.raises(FileNotFoundError, match="SYNTHETIC"):
        P.from_pretrained("CompVis/stable-diffusion-v1-4")
    with pytest.raises(FileNotFoundError):
        P.from_pretrained(str(tmp_path / "typo"))
    pipe = P.from_pretrained("CompVis/stable-diffusion-v1-4", arch="tiny")
    assert pipe.synthetic is True
    assert P.from_pretrained("somewhere", synthetic=True, arch="tiny").synthetic
    pipe.scheduler = SimpleNamespace(config=SimpleNamespace(steps_offset=1, clip_sample=False))     # not a known scheduler class
    with pytest.raises(NotImplementedError, match="not one of the schedulers the reference accepts"):
        pipe._schedule(50, 0.0)


def test_torch_dtype_is_reported_not_swallowed():
    """The reference forwards torch_dtype to diffusers (its tests run float16, tests/test_pipeline.py:19-27); the HIP engines
    are bf16-storage only, so a float16 / float32 request warns loudly and ``pipe.torch_dtype`` says what really runs."""
    import torch
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline as P
    for dt in (torch.float16, torch.float32):
        with pytest.warns(UserWarning, match="runs in bfloat16"):
            pipe = P.from_pretrained("tiny", torch_dtype=dt)
        assert pipe.torch_dtype is torch.bfloat16 and pipe.requested_torch_dtype is dt
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert P.from_pretrained("tiny", torch_dtype=torch.bfloat16).torch_dtype is torch.bfloat16
        assert P.from_pretrained("tiny").torch_dtype is torch.bfloat16
    with pytest.raises(ValueError, match="unsupported torch_dtype"):
        P.from_pretrained("tiny", torch_dtype=torch.int8)


def test_frame_writer_renames_complete_files_into_place(tmp_path):
    from PIL import Image
    from stable_diffusion_videos_amd.utils import FrameWriter
    w = FrameWriter(workers=2)
    for k in range(4):
        w.submit(Image.fromarray(np.full((8, 8, 3), k, dtype=np.uint8)), tmp_path / f"frame{k:06d}.png")
    w.close()
    assert sorted(p.name for p in tmp_path.iterdir()) == [f"frame{k:06d}.png" for k in range(4)]
    assert np.asarray(Image.open(tmp_path / "frame000003.png"))[0, 0, 0] == 3


def test_torch_custom_ops_are_registered_with_fake_kernels(hip):
    """north_star: "... HIP kernels through PyTorch-ROCm custom ops" - torch.ops.sdv.* exist, carry schemas and meta
    (FakeTensor) implementations, and have NO CPU kernel (a CPU tensor raises, it does not fall back)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    import stable_diffusion_videos_amd  # noqa: F401
    from stable_diffusion_videos_amd import ops
    for name in ops.OPS:
        assert hasattr(torch.ops.sdv, name), name
    with FakeTensorMode():
        x = torch.empty((128, 64), dtype=torch.bfloat16)
        w = torch.empty((256, 64), dtype=torch.bfloat16)
        assert torch.ops.sdv.linear(x, w).shape == (128, 256)
        assert torch.ops.sdv.linear(x, w, None, None, 1).shape == (128, 128)               # GEGLU halves N
        assert torch.ops.sdv.conv3x3(x, torch.empty((32, 576), dtype=torch.bfloat16), None, 2, 8, 8, 2).shape == (32, 32)
        assert torch.ops.sdv.upsample_conv3x3(x, torch.empty((4 * 48, 256), dtype=torch.bfloat16), None, 2, 8, 8).shape == (512, 48)
        q = torch.empty((2, 100, 80), dtype=torch.bfloat16)
        assert torch.ops.sdv.attention(q, q, torch.empty((2, 80, 128), dtype=torch.bfloat16), 2, 0.158).shape == (2, 100, 80)
    with pytest.raises(hip.SdvHipError, match="GPU"):
        torch.ops.sdv.linear(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))


def test_every_launch_is_a_torch_custom_op(hip):
    """north_star: "Python host code calling hand-written CDNA4 HIP kernels through PyTorch-ROCm custom ops".  Every launch of
    the C ABI sits inside the implementation of a ``torch.ops.sdv.k_*`` op (hip.py); the wrappers the engines call only pack
    arguments and dispatch.  Checked on the source (no function but an ``*_impl`` touches ``lib.sdv_*``), on the registry (schema
    + Meta kernel for each), and on behaviour (a CPU tensor still raises SdvHipError - there is no CPU kernel behind the op)."""
    import ast
    import torch
    src = (ROOT / "stable_diffusion_videos_amd" / "hip.py").read_text()
    offenders = []
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and not node.name.endswith("_impl") and node.name != "load":
            if re.search(r"lib\.sdv_(?!abi_version|last_error)", ast.get_source_segment(src, node)):
                offenders.append(node.name)
    assert not offenders, offenders
    assert len(hip.KERNEL_OPS) >= 23
    for name in hip.KERNEL_OPS:
        op = getattr(torch.ops.sdv, name)
        assert str(op.default._schema).startswith(f"sdv::{name}(")
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"sdv::{name}", "Meta")
    # the engines' sources contain no ctypes at all
    for mod in ("engine.py", "pipeline.py", "text.py", "esrgan.py", "upsampling.py"):
        assert "import ctypes" not in (ROOT / "stable_diffusion_videos_amd" / mod).read_text(), mod
    x = torch.zeros((64, 64), dtype=torch.bfloat16)
    with pytest.raises(hip.SdvHipError, match="GPU memory"):
        torch.ops.sdv.k_igemm(x, x, x, None, None, None, None, None, None, None, None, None, [64, 64, 64, 64, 64, 64] + [0] * 24 + [-1], 1.0, 1e-5, False)
    with torch.device("meta"):
        m = torch.empty((128, 64), dtype=torch.bfloat16)
        ints = [128, 64, 64, 64, 64, 64, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1] + [0] * 13 + [-1]
        assert torch.ops.sdv.k_igemm(m, m, m, None, None, None, None, None, None, None, None, None, ints, 1.0, 1e-5, True).shape == (128, 2)
        assert torch.ops.sdv.k_slerp_stats(m.float(), m.float()).dtype == torch.float64


def test_hash_tokenizer_call_shape():
    from stable_diffusion_videos_amd import config
    from stable_diffusion_videos_amd.text import HashTokenizer
    tok = HashTokenizer(config.sd14_text())
    assert tok.model_max_length == 77
    ids = tok(["a cat", "a " + "very " * 100 + "long prompt"], padding="max_length", max_length=77, truncation=True,
              return_tensors="pt").input_ids
    assert ids.shape == (2, 77) and ids.dtype == torch.long
    assert ids[0, 0] == 49406 and ids[0, 3] == 49407 and ids[1, -1] == 49407
    assert torch.equal(tok("a cat", max_length=77, truncation=True).input_ids[0], ids[0])
    assert tok("a cat").input_ids[0, 1] == tok("A CAT").input_ids[0, 1]
    long = tok("x " * 100).input_ids                      # no truncation -> longer than 77, as the reference handles at :299
    assert long.shape[1] > 77


def test_pipeline_surface_matches_reference_signature():
    """Same public methods and keyword names as the reference class."""
    import inspect
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline as P
    walk = list(inspect.signature(P.walk).parameters)
    assert walk == ["self", "prompts", "seeds", "num_interpolation_steps", "output_dir", "name", "image_file_ext", "fps",
                    "num_inference_steps", "guidance_scale", "eta", "height", "width", "upsample", "batch_size", "resume",
                    "audio_filepath", "audio_start_sec", "margin", "smooth", "negative_prompt", "make_video"]
    call = list(inspect.signature(P.__call__).parameters)
    assert call[:16] == ["self", "prompt", "height", "width", "num_inference_steps", "guidance_scale", "negative_prompt",
                         "num_images_per_prompt", "eta", "generator", "latents", "output_type", "return_dict", "callback",
                         "callback_steps", "text_embeddings"]
    mcf = list(inspect.signature(P.make_clip_frames).parameters)
    assert mcf[:19] == ["self", "prompt_a", "prompt_b", "seed_a", "seed_b", "num_interpolation_steps", "save_path",
                        "num_inference_steps", "guidance_scale", "eta", "height", "width", "upsample", "batch_size",
                        "image_file_ext", "T", "skip", "negative_prompt", "step"]
    assert inspect.signature(P.walk).parameters["num_interpolation_steps"].default == 5
    assert inspect.signature(P.walk).parameters["make_video"].default is True
    for m in ("from_pretrained", "to", "enable_attention_slicing", "disable_attention_slicing", "generate_inputs",
              "embed_text", "init_noise"):
        assert hasattr(P, m)
    pipe = P.from_pretrained("tiny")
    assert pipe.vae_scale_factor == 8 and pipe.unet.in_channels == 4 and pipe.unet.config.sample_size == 16
    assert pipe.tokenizer.model_max_length == 77 and pipe.safety_checker is None and pipe.tiled is False
    assert P.from_pretrained("tiny", tiled=True).tiled is True
    from stable_diffusion_videos_amd.hip import SdvHipError
    with pytest.raises(SdvHipError, match="no CPU fallback"):           # the text encoder is native too: loud on CPU
        pipe.embed_text("a cat")
    n1, n2 = pipe.init_noise(42, (1, 4, 8, 8)), pipe.init_noise(42, (1, 4, 8, 8))
    assert torch.equal(n1, n2) and n1.shape == (1, 4, 8, 8)


def test_local_checkpoint_directory_round_trip(tmp_path):
    """from_pretrained(<diffusers-layout dir>) reads config.json + safetensors with the diffusers key schema
    (old VAE attention names included) - the path real SD weights take."""
    import json
    from safetensors.torch import save_file
    from stable_diffusion_videos_amd import StableDiffusionWalkPipeline, config, weights
    ucfg, vcfg = config.tiny_unet(), config.tiny_vae()
    u_sd = weights.synthetic_state_dict(weights.unet_shapes(ucfg), seed=11)
    v_sd = weights.synthetic_state_dict(weights.vae_decoder_shapes(vcfg), seed=12)
    (tmp_path / "unet").mkdir()
    (tmp_path / "vae").mkdir()
    (tmp_path / "unet" / "config.json").write_text(json.dumps(
        dict(sample_size=ucfg.sample_size, block_out_channels=list(ucfg.block_out_channels),
             attention_head_dim=list(ucfg.attention_head_dim), cross_attention_dim=ucfg.cross_attention_dim,
             _class_name="UNet2DConditionModel", act_fn="silu")))
    (tmp_path / "vae" / "config.json").write_text(json.dumps(
        dict(block_out_channels=list(vcfg.block_out_channels), latent_channels=4, _class_name="AutoencoderKL")))
    save_file({k: v.half() for k, v in u_sd.items()}, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    old = {k.replace(".to_q.", ".query.").replace(".to_k.", ".key.").replace(".to_v.", ".value.")
           .replace(".to_out.0.", ".proj_attn."): v for k, v in v_sd.items()}
    old["encoder.conv_in.weight"] = torch.zeros(1)          # extra (encoder) keys are ignored
    save_file(old, str(tmp_path / "vae" / "diffusion_pytorch_model.safetensors"))
    _write_clip_tokenizer(tmp_path / "tokenizer")
    pipe = StableDiffusionWalkPipeline.from_pretrained(str(tmp_path))
    from transformers import CLIPTokenizer
    assert isinstance(pipe.tokenizer, CLIPTokenizer) and not pipe.synthetic
    assert pipe.unet.config.block_out_channels == tuple(ucfg.block_out_channels)
    assert pipe.unet.config.attention_head_dim == tuple(ucfg.attention_head_dim)
    for k, v in u_sd.items():
        assert torch.allclose(pipe.unet.state_dict[k], v.half().float()), k
    for k, v in v_sd.items():
        assert torch.equal(pipe.vae.state_dict[k], v), k
    with pytest.raises(FileNotFoundError):
        weights.load_component(tmp_path, "text_encoder", {})


def _write_clip_tokenizer(d):
    """A real (tiny) CLIP byte-level BPE vocabulary: the 256-symbol alphabet, its word-final variants, four merges that build
    "cat</w>" and "dog</w>", and the two special tokens - enough for transformers.CLIPTokenizer to load and tokenise."""
    import json
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    chars = [chr(c) for c in cs]
    merges = [("c", "a"), ("ca", "t</w>"), ("d", "o"), ("do", "g</w>")]
    vocab = chars + [c + "</w>" for c in chars] + [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
    d.mkdir(parents=True)
    (d / "vocab.json").write_text(json.dumps({t: i for i, t in enumerate(vocab)}))
    (d / "merges.txt").write_text("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    (d / "tokenizer_config.json").write_text(json.dumps(dict(
        model_max_length=77, tokenizer_class="CLIPTokenizer", bos_token="<|startoftext|>", eos_token="<|endoftext|>",
        unk_token="<|endoftext|>", pad_token="<|endoftext|>")))
    return {t: i for i, t in enumerate(vocab)}


def test_real_clip_tokenizer_is_used_when_the_checkpoint_has_one(tmp_path):
    """embed_text (stable_diffusion_pipeline.py:809-820) tokenises with the checkpoint's CLIPTokenizer: a model directory
    with tokenizer/vocab.json must load transformers' real class (the hash tokenizer is only the no-vocabulary stand-in),
    and the call shape the pipeline uses - pad to 77 with EOS, truncate, BOS first - must come out of it."""
    from transformers import CLIPTokenizer
    from stable_diffusion_videos_amd import config
    from stable_diffusion_videos_amd.text import HashTokenizer, load_tokenizer
    v = _write_clip_tokenizer(tmp_path / "tokenizer")
    tok = load_tokenizer(tmp_path, config.tiny_text())
    assert isinstance(tok, CLIPTokenizer) and tok.model_max_length == 77
    out = tok(["a cat", "a dog " + "cat " * 200], padding="max_length", max_length=tok.model_max_length, truncation=True,
              return_tensors="pt")
    ids = out.input_ids
    bos, eos = v["<|startoftext|>"], v["<|endoftext|>"]
    assert ids.shape == (2, 77) and ids.dtype == torch.int64
    assert ids[0, :4].tolist() == [bos, v["a</w>"], v["cat</w>"], eos] and (ids[0, 4:] == eos).all()
    assert ids[1, 0] == bos and ids[1, 1:3].tolist() == [v["a</w>"], v["dog</w>"]] and ids[1, -1] == eos
    assert (ids[1, 3:-1] == v["cat</w>"]).all()                                   # truncated to 77, EOS kept last
    assert isinstance(load_tokenizer(tmp_path / "nothing-here", config.tiny_text()), HashTokenizer)


def test_realesrgan_surface_and_weights(tmp_path):
    """Host side of the Real-ESRGAN row (reference upsampling.py:13-99): constructor / method surface, the
    RealESRGAN_x4plus.pth container layouts, the K-padded weight re-layout, and loud failure without a GPU."""
    import inspect
    from stable_diffusion_videos_amd.config import RRDBNetConfig
    from stable_diffusion_videos_amd.upsampling import PipelineRealESRGAN, RealESRGANModel
    from stable_diffusion_videos_amd.weights import (conv_w_kpad, count_params, load_rrdbnet, rrdbnet_shapes,
                                                     synthetic_state_dict)
    assert PipelineRealESRGAN is RealESRGANModel
    assert list(inspect.signature(RealESRGANModel.__init__).parameters)[:6] == ["self", "model_path", "tile", "tile_pad",
                                                                                "pre_pad", "fp32"]
    assert list(inspect.signature(RealESRGANModel.forward).parameters) == ["self", "image", "outscale", "convert_to_pil"]
    assert list(inspect.signature(RealESRGANModel.upsample_imagefolder).parameters) == [
        "self", "in_dir", "out_dir", "suffix", "outfile_ext", "recursive", "force"]
    shapes = rrdbnet_shapes(RRDBNetConfig())
    assert count_params(shapes) == 16_697_987
    small = rrdbnet_shapes(RRDBNetConfig(num_block=1))
    sd = synthetic_state_dict(small, seed=3)
    for container in ({"params_ema": sd}, {"params": sd}, sd):
        f = tmp_path / "RealESRGAN_x4plus.pth"
        torch.save(container, f)
        back = load_rrdbnet(f, small)
        assert all(torch.equal(back[k], sd[k]) for k in small)
    with pytest.raises(KeyError):
        load_rrdbnet(f, shapes)                                       # 1-block file against the 23-block schema
    w = torch.randn(32, 96, 3, 3)
    wk = conv_w_kpad(w, "cpu").float().reshape(32, 3, 3, 128)
    assert torch.equal(wk[..., :96], w.permute(0, 2, 3, 1).to(torch.bfloat16).float()) and float(wk[..., 96:].abs().max()) == 0
    m = RealESRGANModel.from_pretrained(str(tmp_path / "nowhere"))   # nothing found offline -> synthetic, flagged
    assert m.synthetic and m.scale == 4
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.to("cpu")
    with pytest.raises(NotImplementedError):
        RealESRGANModel(None, tile=256)
    with pytest.raises(FileNotFoundError):
        m.upsample_imagefolder(tmp_path / "missing", tmp_path / "out")


def test_hot_kernels_do_not_spill():
    """Guard against register-pressure regressions in the kernels the roofline depends on: compile sdv_gemm.hip /
    sdv_attention.hip with the build's own flags plus -Rpass-analysis=kernel-resource-usage and check scratch use.
    (Adding two activation branches to the igemm epilogue once pushed the 256x320 tile from 64 B to 768 B of scratch per
    lane - every test still passed, the bench lost 4x.)"""
    import re
    import subprocess
    from stable_diffusion_videos_amd import build as b
    budget = {"sdv_gemm.hip": 128, "sdv_attention.hip": 0}       # bytes of scratch per lane allowed (epilogue-only slots)
    for name, limit in budget.items():
        src = b.CSRC / name
        cmd = [b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS.get(name, []), "-Rpass-analysis=kernel-resource-usage", "-c",
               str(src), "-o", "/dev/null"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        names = re.findall(r"Function Name: (\S+)", r.stderr)
        scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
        assert names and len(names) == len(scratch)
        # Kernels allowed MORE than the file's budget, each with the exact number of bytes measured when it was admitted (so that a
        # regression inside an exception still fails) - all of them tile-boundary slots, none inside a K loop (ISA lint below):
        #  * the block-scaled fp8 variants (FEAT 9) of the 256 x 320 tile: operands are aligned 8-register tuples, and at the tile
        #    boundary the allocator parks two to three accumulator tiles;
        #  * the GroupNorm-statistics variant (FEAT 4) of the 256 x 320 conv: the statistics butterfly's 16 + 16 values on top of the
        #    conv's addressing state (the statistics live in a variant of their own so that the plain kernels do not pay for them).
        # (round 4 had a fourth exception - the LayerNorm-fold variant of the ring tile 12 - which left with that tile in round 5)
        # (round 6: sdv_gemm.hip is built without the SLP vectoriser - DESIGN.md "The co-residency finding" -; the DENSE block-scaled
        #  variant, which no engine launches, went 160 -> 228 B with it, the two conv variants stayed inside their numbers)
        exact = {r"igemm_kernelILi4ELi2ELi2ELi5ELi64ELb0ELi2ELi9E": 228, r"igemm_kernelILi4ELi2ELi2ELi5ELi64ELb1ELi2ELi9E": 176,
                 r"igemm_kernelILi4ELi2ELi2ELi5ELi64ELb1ELi2ELi4E": 140}

        def allowed(n):
            for pat, nbytes in exact.items():
                if re.search(pat, n):
                    return nbytes
            return limit

        over = [(sc, allowed(n), n) for sc, n in zip(scratch, names) if sc > allowed(n)]
        assert not over, f"{name}: scratch per lane over budget (bytes, allowed, kernel): {over}"


def test_buffer_stores_are_followed_by_idle_slots_before_their_registers_change():
    """ISA lint for the store hazard found in round 2: on gfx950 a `buffer_store_dwordx4` with an SGPR offset
    does not sample all of its data at issue - a VALU write to the first data register in the next slot reached memory in
    lanes 12..15 of every 16 - and the compiler's hazard table has no wait state for that form.  The epilogue therefore puts
    `s_nop 7` behind every store and keeps data + address registers live across it.  This compiles the 256 x 320 tile to ISA
    and checks exactly that: between a store and its s_nop nothing writes the store's data or address VGPRs."""
    import re
    import subprocess
    import tempfile
    from stable_diffusion_videos_amd import build as b
    src = b.CSRC / "sdv_gemm.hip"
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "gemm6.s"
        cmd = [b.hipcc(), *b.FLAGS, *b.FAST_FLAGS, *b.EXTRA_FLAGS.get(src.name, []), "-DSDV_GEMM_ONLY_TILE6", "-S",
               "--cuda-device-only", "-o", str(out), str(src)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        raw = [ln.split(";")[0].strip() for ln in out.read_text().splitlines()]
    # (second lint on the same ISA: the steady-state K loop of every 256 x 320 kernel - a basic block with MFMAs that branches
    #  back to its own label - touches no scratch; the bytes test_hot_kernels_do_not_spill allows are tile-boundary slots)
    where, tightest = {}, {}
    for i, ln in enumerate(raw):
        if re.fullmatch(r"\.LBB\d+_\d+:", ln):
            where[ln[:-1]] = i
        elif ln.startswith(("s_cbranch", "s_branch")) and ln.split()[-1] in where:      # a backward branch: [label .. here] is a loop
            body = raw[where[ln.split()[-1]]:i]
            if any(x.startswith("v_mfma") for x in body):
                kern = ln.split()[-1].split("_")[0]
                if kern not in tightest or len(body) < len(tightest[kern]):
                    tightest[kern] = body                                               # the tightest loop around MFMAs
    assert len(tightest) >= 8, f"expected the K loops of the eight 256 x 320 kernels, found {len(tightest)}"
    for kern, body in tightest.items():
        assert not any(x.startswith(("scratch_", "buffer_store")) for x in body), f"scratch access inside the K loop of kernel {kern}"
    lines = [ln for ln in raw if ln and not ln.startswith(".") and not ln.endswith(":")]

    def vregs(op):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", op)
        return {int(m.group(1))} if m else set()

    def written(ln):
        mnem, _, rest = ln.partition(" ")
        if not mnem.startswith(("v_", "ds_read", "buffer_load", "global_load", "scratch_load")) or mnem.startswith("v_cmp"):
            return set()
        return vregs(rest.split(",")[0].strip())

    stores = [i for i, ln in enumerate(lines) if ln.startswith("buffer_store_dwordx4")]
    assert len(stores) >= 100, "expected the staged epilogues of the 256 x 320 kernels"
    for i in stores:
        ops = [o.strip() for o in lines[i].partition(" ")[2].split(",")]
        guarded = vregs(ops[0]) | vregs(ops[1])
        assert len(guarded) >= 5, lines[i]
        for j in range(i + 1, min(i + 400, len(lines))):      # (the scheduler may park a pass's conversions in between)
            if lines[j].startswith("s_nop 7"):
                break
            assert not lines[j].startswith(("buffer_store", "s_cbranch", "s_branch", "s_endpgm")), (lines[i], lines[j])
            assert not (written(lines[j]) & guarded), f"{lines[j]!r} overwrites a register of {lines[i]!r} before its s_nop"
        else:
            raise AssertionError(f"no s_nop 7 behind {lines[i]!r}")


def test_generate_images_layout_with_a_stub_pipeline(tmp_path):
    """generate_images (reference image_generation.py:108-215): batching, ``{seed}{ext}`` file names, prompt_config.json
    and the argument errors - exercised on CPU with a stub that has the pipeline's call shapes."""
    import json
    from types import SimpleNamespace
    from PIL import Image
    from stable_diffusion_videos_amd.image_generation import generate_images, generate_input_batches

    class Stub:
        device = torch.device("cpu")
        tiled = False
        unet = SimpleNamespace(in_channels=4)
        scheduler = SimpleNamespace(config=SimpleNamespace(beta_start=0.00085, prediction_type="epsilon"))
        calls = []

        def embed_text(self, text):
            return torch.zeros(1, 77, 8)

        def init_noise(self, seed, shape):
            return torch.full(shape, float(seed))

        def __call__(self, text_embeddings=None, latents=None, output_type="pil", **kw):
            self.calls.append((latents[:, 0, 0, 0].tolist(), kw["height"], output_type))
            return {"images": [Image.new("RGB", (kw["width"], kw["height"])) for _ in range(latents.shape[0])]}

    pipe = Stub()
    files = generate_images(pipe, "a cat", batch_size=2, num_batches=2, seeds=[5, 6, 7, 8], output_dir=tmp_path, name="run",
                            height=64, width=32, num_inference_steps=3, image_file_ext=".png")
    assert [Path(f).name for f in files] == ["5.png", "6.png", "7.png", "8.png"] and all(Path(f).exists() for f in files)
    assert pipe.calls == [([5.0, 6.0], 64, "pil"), ([7.0, 8.0], 64, "pil")]
    cfg = json.loads((tmp_path / "run" / "prompt_config.json").read_text())
    assert cfg["prompt"] == "a cat" and cfg["num_inference_steps"] == 3 and cfg["scheduler"]["prediction_type"] == "epsilon"
    assert [b[1].shape[0] for b in generate_input_batches(pipe, ["p"] * 5, list(range(5)), 2, 64, 64)] == [2, 2, 1]
    with pytest.raises(ValueError, match="seeds"):
        generate_images(pipe, "a cat", batch_size=2, num_batches=1, seeds=[1], output_dir=tmp_path, name="bad")
    with pytest.raises(ValueError, match="equal"):
        list(generate_input_batches(pipe, ["a"], [1, 2], 1, 64, 64))
    with pytest.raises(ValueError, match="repo_id"):
        generate_images(pipe, "a cat", push_to_hub=True, output_dir=tmp_path, name="hub")
    with pytest.raises(NotImplementedError):
        generate_images(pipe, "a cat", push_to_hub=True, repo_id="x/y", output_dir=tmp_path, name="hub2")
    with pytest.raises(FileExistsError):
        generate_images(pipe, "a cat", seeds=[1], output_dir=tmp_path, name="run")


def test_bench_reads_the_committed_pmc_profile(tmp_path):
    """bench.py folds the committed rocprofv3 --pmc summary of its default batch size into the JSON line (`roofline.traffic`,
    `attention.mfma_busy_pmc`) - but only a summary that was collected from THIS tree's kernel sources (the `# csrc=` fingerprint in
    its first line): the committed one must match, parse and carry the two kernels the bench looks up; a summary
    of other sources is refused, with the reason."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fp = bench.csrc_fingerprint()
    assert len(fp) == 16 and fp == bench.csrc_fingerprint()
    for name in ("round6_pmc_unet_b128.csv", "round6_bench_b128_kernel_stats.csv"):
        assert bench.profile_fingerprint(ROOT / "profiles" / name) == fp, \
            f"profiles/{name} was not collected from the kernel sources of this tree: re-run tools/run_profiles_r6.sh and commit its summaries"
    pmc = bench.pmc_profile(128)
    tr = bench.dominant_kernel_traffic(pmc)
    assert tr and tr["kernel"].startswith("igemm_kernel<4, 2, 2, 5, 64, true")
    assert 1e9 < tr["fetch_bytes"] < 2e10 and 1e8 < tr["write_bytes"] < 5e9
    shapes = [dict(kind="attention", dh=40, Lq=4096, Lk=4096, B=256, H=8, tflops=800.0)]
    att = bench.attention_object(shapes, pmc)
    assert att["shape"]["dh"] == 40 and 0.3 < att["mfma_busy_pmc"] < 0.9 and att["issued_tflops"] == 1120.0
    rp = bench.rocprof_conv_average(128)
    assert rp and 500 < rp["avg_launch_us"] < 5000 and rp["launches"] % 51 == 0
    assert bench.pmc_profile(7) == {}                      # no profile for that batch size: the fields stay null
    # a summary of OTHER kernel sources is not replayed
    real_root = bench.ROOT
    try:
        bench.ROOT = tmp_path
        (tmp_path / "profiles").mkdir()
        (tmp_path / "stable_diffusion_videos_amd").symlink_to(real_root / "stable_diffusion_videos_amd")
        (tmp_path / "include").symlink_to(real_root / "include")
        body = (real_root / "profiles" / "round6_pmc_unet_b128.csv").read_text().split("\n", 1)[1]
        (tmp_path / "profiles" / "round6_pmc_unet_b128.csv").write_text("# csrc=0123456789abcdef somebody else's kernels\n" + body)
        assert bench.pmc_profile(128) == {} and "not replayed" in bench.pmc_profile.stale
        (tmp_path / "profiles" / "round6_pmc_unet_b128.csv").write_text(body)             # no fingerprint at all (a round-4 file)
        assert bench.pmc_profile(128) == {} and "unrecorded" in bench.pmc_profile.stale
    finally:
        bench.ROOT = real_root



def test_resume_policies(tmp_path):
    """walk(resume=True), reference :741-753: "holes" (default) regenerates every missing / empty frame; "reference" continues behind the
    last frame on disk and reproduces the rule's quirks - a hole before the last frame stays, a clip missing only its last frame is
    skipped - and both skip a clip whose mp4 exists."""
    from stable_diffusion_videos_amd.pipeline import StableDiffusionWalkPipeline as P
    clip = tmp_path / "w_000000"
    clip.mkdir()
    mp4 = clip / "w_000000.mp4"
    for k in (0, 1, 2, 4, 5):
        (clip / f"frame{k:06d}.png").write_bytes(b"x")
    (clip / "frame000006.png").write_bytes(b"")                       # a zero-byte file: killed while writing
    assert P.resume_todo(clip, mp4, 10) == [3, 6, 7, 8, 9]
    assert P.resume_todo(clip, mp4, 10, policy="reference") == [7, 8, 9]      # behind the LAST file, hole 3 and the empty 6 stay
    assert P.resume_todo(clip, mp4, 8, policy="reference") is None            # last = num_step - 2: the :750 quirk skips the clip
    assert P.resume_todo(clip, mp4, 8) == [3, 6, 7]
    for k in (3, 6, 7):
        (clip / f"frame{k:06d}.png").write_bytes(b"x")
    assert P.resume_todo(clip, mp4, 8) is None and P.resume_todo(clip, mp4, 8, policy="reference") is None
    empty = tmp_path / "w_000001"
    empty.mkdir()
    assert P.resume_todo(empty, empty / "w_000001.mp4", 3) == [0, 1, 2] == P.resume_todo(empty, empty / "w_000001.mp4", 3, policy="reference")
    mp4.write_bytes(b"v")
    assert P.resume_todo(clip, mp4, 99) is None and P.resume_todo(clip, mp4, 99, policy="reference") is None
    with pytest.raises(ValueError):
        P.resume_todo(clip, mp4, 8, policy="other")
    assert P.resume_policy in ("holes", "reference")

"""CPU tests of the block-wise parity machinery (oracle/blockwise.py) - no GPU: the "engine" here is the oracle itself
with its block inputs / outputs rounded to bf16 (the ideal engine: ONE rounding per block), recorded in the engine's tap
format.  They check that (1) the comparison plumbing (layouts, names, time embedding, CFG-shared prefix) is right - an ideal
engine sits at EPS_BF16, well inside every bound; (2) every mutation in ``blockwise.mutations`` pushes at least one block it
touches over its bound, i.e. the gate the GPU tests apply (tests/test_blockwise_gpu.py) has the power it claims."""
import math

import pytest
import torch

from conftest import bf16_round
from helpers import make_oracle_unet, make_oracle_vae
from oracle import blockwise as bw
from oracle import models as om
from stable_diffusion_videos_amd import config as cfgs
from stable_diffusion_videos_amd import weights


def _tok(x):    # NCHW -> the engine's token-major layout
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


def ideal_engine_records(model, run, ctx=None):
    """Forward hooks that turn one oracle forward into tap records whose tensors are bf16-rounded."""
    recs, hooks = [], []

    def add_stages(mod, name):
        # the five stage records the engine emits per transformer block: every stage runs (in the oracle) on the rounded output
        # of the stage before it
        def hook(m, args, out):
            x = args[0]
            base = {"name": name, "nimg": x.shape[0], "H": x.shape[2], "W": x.shape[3]}
            cur = bf16_round(_tok(x))
            for kind in ("tf_in", "tf_attn1", "tf_attn2", "tf_ff", "tf_out"):
                rec = dict(base, kind=kind, x=cur)
                if kind == "tf_out":
                    rec["x2"] = bf16_round(_tok(x))
                rec["out"] = bf16_round(_tok(bw.oracle_block_output(model, rec, ctx=args[1])))
                recs.append(rec)
                cur = rec["out"]
        hooks.append(mod.register_forward_hook(hook))

    def add(mod, name, kind, out_round=True):
        def hook(m, args, out):
            x = args[0]
            recs.append({"name": name, "kind": kind, "x": bf16_round(_tok(x)), "nimg": x.shape[0], "H": x.shape[2],
                         "W": x.shape[3], "out": bf16_round(_tok(out)) if out_round else _tok(out)})
        hooks.append(mod.register_forward_hook(hook))

    for name, mod in model.named_modules():
        if isinstance(mod, om.ResnetBlock2D):
            add(mod, name, "resnet")
        elif isinstance(mod, om.Transformer2DModel):
            add(mod, name, "transformer")
            add_stages(mod, name)
        elif isinstance(mod, om.Downsample2D):
            add(mod, name, "down")
        elif isinstance(mod, om.Upsample2D):
            add(mod, name, "up")
        elif isinstance(mod, om.VAEAttention):
            add(mod, name, "vae_attention")
        elif name in ("conv_in", "decoder.conv_in"):
            add(mod, name, "conv")
    with torch.no_grad():
        run()
    for h in hooks:
        h.remove()
    return recs


@pytest.fixture(scope="module")
def tiny_unet_case():
    c = cfgs.tiny_unet()
    m = make_oracle_unet(c, weights.synthetic_state_dict(weights.unet_shapes(c), seed=0))
    g = torch.Generator().manual_seed(3)
    x = bf16_round(torch.randn((2, 4, 16, 16), generator=g))
    ctx = bf16_round(3.0 * torch.randn((2, 77, c.cross_attention_dim), generator=g))    # (x3: peaked cross-attention, as with real CLIP states)
    t = 501
    # the hooks round every block's INPUT too, so the recorded forward must run on rounded block inputs as well: do it by
    # rounding between blocks with pre-hooks
    pre = [mod.register_forward_pre_hook(lambda m, a: (bf16_round(a[0]),) + tuple(a[1:]))
           for _, mod in m.named_modules()
           if isinstance(mod, (om.ResnetBlock2D, om.Transformer2DModel, om.Downsample2D, om.Upsample2D))]
    recs = ideal_engine_records(m, lambda: m(x, torch.tensor(t), ctx))
    for h in pre:
        h.remove()
    return m, recs, t, ctx


def test_bounds_are_the_documented_function_of_the_rounding_count():
    assert abs(bw.EPS_BF16 - 1.63e-3) < 2e-5
    for kind, n in bw.BOUND_ROUNDINGS.items():
        assert bw.bound(kind) == pytest.approx(1.5 * bw.EPS_BF16 * math.sqrt(n))
    # a measured rounding agrees with the model: log-uniform magnitudes
    g = torch.Generator().manual_seed(0)
    v = torch.exp(torch.rand(1 << 20, generator=g) * 8 - 4) * torch.sign(torch.randn(1 << 20, generator=g))
    assert bw.rel_l2(bf16_round(v), v) == pytest.approx(bw.EPS_BF16, rel=0.05)


def test_ideal_engine_is_inside_every_bound(tiny_unet_case):
    m, recs, t, ctx = tiny_unet_case
    kinds = {r["kind"] for r in recs}
    assert {"conv", "resnet", "transformer", "down", "up", "tf_in", "tf_attn1", "tf_attn2", "tf_ff", "tf_out"} <= kinds
    rows = bw.compare(m, recs, timestep=t, ctx=ctx)
    assert len(rows) == len(recs) == 1 + 22 + 16 * 6 + 3 + 3
    for r in rows:
        # one output rounding: ~EPS_BF16 (less where a residual stream carries exact bits)
        assert r["rel_l2"] <= 1.25 * bw.EPS_BF16 < r["bound"], r


def test_every_unet_mutation_breaks_the_gate(tiny_unet_case):
    m, recs, t, ctx = tiny_unet_case
    muts = bw.mutations(m, ctx)
    assert len(muts) >= 10
    for label, (mm, kw, hits) in muts.items():
        args = dict(timestep=t, ctx=ctx)
        args.update(kw)
        rows = bw.compare(mm, [r for r in recs if hits(r)], **args)
        worst = max(r["rel_l2"] / r["bound"] for r in rows)
        frac = sum(r["rel_l2"] > r["bound"] for r in rows) / len(rows)
        print(f"{label:64s} worst block {worst:7.1f} x bound, {100 * frac:5.1f} % of the touched blocks over their bound")
        assert worst > 1.0, label


def test_blind_spots_are_blind(tiny_unet_case):
    """The documented limits of the gate (blockwise.blind_spots): an epsilon mix-up moves nothing measurable."""
    m, recs, t, ctx = tiny_unet_case
    spots = bw.blind_spots(m)
    assert len(spots) == 2
    for label, (mm, kw, hits) in spots.items():
        rows = bw.compare(mm, recs, timestep=t, ctx=ctx)
        assert max(r["rel_l2"] / r["bound"] for r in rows) < 1.0, label


def test_shared_prefix_and_concat_records(tiny_unet_case):
    """The two record shapes only the engine produces: a CFG-shared transformer input (nimg/2 samples in, nimg out) and a
    ResBlock with its skip tensor as a second input."""
    m, recs, t, ctx = tiny_unet_case
    r = next(r for r in recs if r["kind"] == "transformer")
    half = r["x"].shape[0] // 2
    x1 = r["x"][:half]
    ctx2 = torch.cat([ctx[:1], ctx[:1]])
    with torch.no_grad():
        full = m.get_submodule(r["name"])(bw.to_nchw(torch.cat([x1, x1]), 2, r["H"], r["W"]), ctx2)
    rec = dict(r, x=x1, nimg=1, shared_prefix=True, out=_tok(full))
    assert bw.compare(m, [rec], timestep=t, ctx=ctx2)[0]["rel_l2"] < 1e-6
    r = next(r for r in recs if r["kind"] == "resnet" and r["name"].startswith("up_blocks"))
    c1 = r["x"].shape[1] // 2
    rec = dict(r, x=r["x"][:, :c1].contiguous(), x2=r["x"][:, c1:].contiguous())
    a, b = bw.compare(m, [r, rec], timestep=t, ctx=ctx)
    assert a["rel_l2"] == b["rel_l2"]


def test_vae_records_and_mutations():
    c = cfgs.tiny_vae()
    m = make_oracle_vae(c, weights.synthetic_state_dict(weights.vae_decoder_shapes(c), seed=1))
    g = torch.Generator().manual_seed(5)
    lat = torch.randn((1, 4, 8, 8), generator=g) * 0.18215 * 0.6
    pre = [mod.register_forward_pre_hook(lambda mm, a: (bf16_round(a[0]),) + tuple(a[1:]))
           for _, mod in m.named_modules() if isinstance(mod, (om.ResnetBlock2D, om.Upsample2D, om.VAEAttention))]
    recs = ideal_engine_records(m, lambda: m.decoder(bf16_round(m.post_quant_conv(lat / 0.18215))))
    for h in pre:
        h.remove()
    recs.insert(0, {"name": "post_quant_conv", "kind": "post_quant", "x": _tok(lat), "nimg": 1, "H": 8, "W": 8,
                    "out": bf16_round(_tok(m.post_quant_conv(lat / 0.18215)))})
    rows = bw.compare(m, recs)
    assert {r["kind"] for r in rows} >= {"post_quant", "conv", "resnet", "vae_attention", "up"}
    for r in rows:
        assert r["rel_l2"] <= 1.25 * bw.EPS_BF16 < r["bound"], r
    for label, (mm, kw, hits) in bw.mutations(m).items():
        rows = bw.compare(mm, [r for r in recs if hits(r)], **kw)
        worst = max(r["rel_l2"] / r["bound"] for r in rows)
        print(f"{label:64s} worst block {worst:7.1f} x bound")
        assert worst > 1.0, label

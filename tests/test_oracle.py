"""CPU tests of the oracle itself: pinned against the reference where the reference can be executed
(slerp golden vectors), structural checks elsewhere (parameter counts, schedule known answers)."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import interp, models, pipeline
from oracle.scheduler import DDIMScheduler

GOLDEN_FILES = ["slerp_seed42_1337_fp32", "slerp_seed7_8_fp16", "slerp_parallel_fp32", "slerp_antiparallel_fp32",
                "slerp_numpy_fp64"]


@pytest.mark.parametrize("name", GOLDEN_FILES)
def test_slerp_bit_exact_vs_reference_golden(name):
    """tests/golden/*.npz hold outputs of the reference's own slerp (utils.py:42-66, AST-lifted)."""
    d = np.load(GOLDEN / f"{name}.npz")
    for t in d["ts"]:
        gold = d[f"t{int(t * 100):03d}"]
        got = interp.slerp_np(float(t), d["v0"], d["v1"])
        assert got.dtype == gold.dtype and np.array_equal(got, gold), (name, t)
    # torch entry form (utils.py:45-49, :63-64)
    got = interp.slerp(0.5, torch.from_numpy(d["v0"]), torch.from_numpy(d["v1"]))
    assert np.array_equal(got.numpy(), d["t050"])


def test_slerp_known_answers_from_survey():
    """SURVEY.md 8c known-answer vectors (seeds 42 / 1337, CPU generator, fp32)."""
    v0, v1 = interp.init_noise(42, (1, 4, 64, 64)), interp.init_noise(1337, (1, 4, 64, 64))
    assert np.allclose(v0.flatten()[:4].numpy(), [1.9269152879714966, 1.4872840642929077, 0.9007171988487244,
                                                  -2.1055209636688232])
    assert abs(float(v0.norm()) - 128.51612854) < 1e-3 and abs(float(v1.norm()) - 128.01329041) < 1e-3
    mid = interp.slerp(0.5, v0, v1)
    assert np.allclose(mid.flatten()[:4].numpy(), [1.4916281700, 1.0030324459, 0.3829366863, -2.1377933025], atol=1e-6)
    assert torch.equal(interp.slerp(0.0, v0, v1), v0) and torch.equal(interp.slerp(1.0, v0, v1), v1)
    with pytest.raises(TypeError):      # fact 5: the reference cannot slerp bf16
        interp.slerp(0.5, v0.bfloat16(), v1.bfloat16())


def test_generate_inputs_batching():
    ea, eb = torch.randn(1, 77, 8), torch.randn(1, 77, 8)
    la, lb = interp.init_noise(1, (1, 4, 8, 8)), interp.init_noise(2, (1, 4, 8, 8))
    out = list(interp.generate_inputs(ea, eb, la, lb, np.linspace(0, 1, 5), 2))
    assert [o[0] for o in out] == [0, 1, 2]
    assert [o[1].shape[0] for o in out] == [2, 2, 1]
    assert torch.equal(out[0][1][0], ea[0]) and torch.equal(out[2][1][0], eb[0])
    assert torch.equal(out[0][2][0], la[0]) and torch.equal(out[2][2][0], lb[0])


def test_parameter_counts_match_sd_checkpoints():
    with torch.device("meta"):
        assert models.count_params(models.UNet2DConditionModel(models.sd14_unet_config())) == 859_520_964
        assert models.count_params(models.UNet2DConditionModel(models.sd21_unet_config())) == 865_910_724
        assert models.count_params(models.AutoencoderKLDecoder(models.sd_vae_config())) == 49_490_199


def test_ddim_schedule_known_answers():
    s = DDIMScheduler()
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(981, 0, -20))
    assert abs(float(s.alphas_cumprod[0]) - 0.99915) < 1e-6
    assert abs(float(s.alphas_cumprod[999]) - 0.0046601) < 1e-5
    # eta = 0, eps = 0: x_prev = sqrt(a_prev / a_t) x
    x = torch.ones(4)
    out = s.step(torch.zeros(4), 981, x)
    assert torch.allclose(out, x * (s.alphas_cumprod[961] / s.alphas_cumprod[981]) ** 0.5)
    # last step falls back to final_alpha_cumprod = alphas_cumprod[0] (set_alpha_to_one=False)
    out = s.step(torch.zeros(4), 1, x)
    assert torch.allclose(out, x * (s.alphas_cumprod[0] / s.alphas_cumprod[1]) ** 0.5)
    # v-prediction identity: v = 0 -> eps = sqrt(1-a) x, x0 = sqrt(a) x
    v = DDIMScheduler(prediction_type="v_prediction")
    v.set_timesteps(50)
    a_t, a_p = v.alphas_cumprod[981], v.alphas_cumprod[961]
    out = v.step(torch.zeros(4), 981, x)
    assert torch.allclose(out, (a_p ** 0.5 * a_t ** 0.5 + (1 - a_p) ** 0.5 * (1 - a_t) ** 0.5) * x)


@pytest.mark.parametrize("ptype", ["epsilon", "v_prediction"])
def test_ddim_chain_reproduces_the_forward_marginals(ptype):
    """A check that does not lean on diffusers: DDIM with eta = 0 (Song et al. 2021, eq. 12) fed the TRUE noise maps
    x_t = sqrt(a_t) x0 + sqrt(1 - a_t) eps onto x_s = sqrt(a_s) x0 + sqrt(1 - a_s) eps for the same (x0, eps) - chained
    over all 50 steps of the BASELINE schedule it must land on the t = 0 marginal with final_alpha_cumprod = a[0]
    (set_alpha_to_one=False).  For v-prediction the model output is v = sqrt(a_t) eps - sqrt(1 - a_t) x0."""
    g = torch.Generator().manual_seed(3)
    x0, eps = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64), torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    s = DDIMScheduler(prediction_type=ptype)
    s.set_timesteps(50)
    a = s.alphas_cumprod.double()
    t0 = int(s.timesteps[0])
    x = a[t0].sqrt() * x0 + (1 - a[t0]).sqrt() * eps
    for t in s.timesteps.tolist():
        out = eps if ptype == "epsilon" else a[t].sqrt() * eps - (1 - a[t]).sqrt() * x0
        x = s.step(out, t, x)
        prev = t - 20
        a_p = a[prev] if prev >= 0 else a[0]
        assert torch.allclose(x, a_p.sqrt() * x0 + (1 - a_p).sqrt() * eps, atol=1e-6), t
    assert torch.allclose(x, a[0].sqrt() * x0 + (1 - a[0]).sqrt() * eps, atol=1e-6)


def test_tiny_oracle_end_to_end_runs():
    """The oracle's full control flow on a tiny architecture: shapes, range, determinism."""
    torch.manual_seed(0)
    ucfg = models.UNetConfig(sample_size=8, block_out_channels=(32, 64, 64, 64), attention_head_dim=(1, 2, 2, 2),
                             cross_attention_dim=32)
    vcfg = models.VAEConfig(block_out_channels=(32, 32, 64, 64))
    unet, vae = models.UNet2DConditionModel(ucfg).eval(), models.AutoencoderKLDecoder(vcfg).eval()
    ea, eb, un = torch.randn(1, 77, 32), torch.randn(1, 77, 32), torch.randn(1, 77, 32)
    frames = pipeline.make_clip_frames(unet, vae, DDIMScheduler(), ea, eb, un, 42, 1337, 3, 64, 64, batch_size=2,
                                       num_inference_steps=3)
    assert frames.shape == (3, 64, 64, 3) and frames.dtype == np.uint8
    again = pipeline.make_clip_frames(unet, vae, DDIMScheduler(), ea, eb, un, 42, 1337, 3, 64, 64, batch_size=1,
                                      num_inference_steps=3)
    assert np.abs(frames.astype(int) - again.astype(int)).max() <= 1
    with pytest.raises(ValueError, match="Unexpected T shape"):
        pipeline.make_clip_frames(unet, vae, DDIMScheduler(), ea, eb, un, 1, 2, 4, 64, 64, T=np.linspace(0, 1, 3))


def test_rrdbnet_oracle_schema_and_properties():
    """Real-ESRGAN x4plus generator restatement: published parameter count and state-dict schema, output geometry,
    and two exact properties of the architecture (zero network = zero image; the 0.2-scaled residual structure)."""
    from oracle import esrgan
    net = esrgan.RRDBNet().eval()
    assert sum(p.numel() for p in net.parameters()) == 16_697_987          # RealESRGAN_x4plus.pth (params_ema)
    keys = list(net.state_dict().keys())
    assert keys[0] == "conv_first.weight" and "body.22.rdb3.conv5.bias" in keys and keys[-1] == "conv_last.bias"
    assert net.state_dict()["body.0.rdb1.conv4.weight"].shape == (32, 160, 3, 3)
    small = esrgan.RRDBNet(esrgan.RRDBNetConfig(num_block=1)).eval()
    img = np.random.RandomState(0).randint(0, 256, (12, 20, 3), dtype=np.uint8)
    out = esrgan.enhance_rgb_u8(small, img)
    assert out.shape == (48, 80, 3) and out.dtype == np.uint8
    with torch.no_grad():
        for p in small.parameters():
            p.zero_()
        small.conv_last.bias.fill_(0.25)
        assert np.all(esrgan.enhance_rgb_u8(small, img) == 64)             # round(0.25 * 255) = 64 (63.75)
        # zero convs: every dense block returns 0 * 0.2 + x = x, so the RRDB returns x * 0.2 + x
        blk = esrgan.RRDB(64, 32)
        for p in blk.parameters():
            p.zero_()
        f = torch.randn(1, 64, 6, 6)
        assert torch.allclose(blk(f), 1.2 * f)
    with pytest.raises(ValueError):
        esrgan.RRDBNet(esrgan.RRDBNetConfig(scale=2))


def _clip_golden(act):
    from pathlib import Path
    z = np.load(Path(__file__).resolve().parent / "golden" / f"clip_{act}.npz")
    sd = {k[4:]: (torch.from_numpy(z[k].astype(np.int32)) << 16).view(torch.float32) for k in z.files if k.startswith("sd::")}
    return z, sd


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_clip_oracle_pinned_against_transformers(act):
    """oracle/clip.py vs last_hidden_state of the real transformers.CLIPTextModel: the committed vectors, and the live
    model when transformers can be imported (it is part of this image).  fp32 on both sides: 5e-4 absolute on values
    of magnitude ~15 (attention is evaluated in a different association order)."""
    from oracle.clip import clip_text_forward
    z, sd = _clip_golden(act)
    ids = torch.from_numpy(z["ids"])
    out = clip_text_forward(sd, ids, int(z["num_heads"]), act)
    assert float((out - torch.from_numpy(z["last_hidden_state"])).abs().max()) < 5e-4
    # prefixed (transformers < 5) state dicts are accepted too
    out2 = clip_text_forward({"text_model." + k: v for k, v in sd.items()}, ids, int(z["num_heads"]), act)
    assert torch.equal(out, out2)
    # causality: changing a later token never changes an earlier position
    ids2 = ids.clone()
    ids2[:, 40:] = 5
    assert torch.equal(clip_text_forward(sd, ids2, int(z["num_heads"]), act)[:, :40], out[:, :40])
    try:
        from transformers import CLIPTextConfig, CLIPTextModel
    except Exception:
        return
    D = sd["final_layer_norm.weight"].numel()
    nl = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    cfg = CLIPTextConfig(vocab_size=sd["embeddings.token_embedding.weight"].shape[0], hidden_size=D,
                         intermediate_size=sd["encoder.layers.0.mlp.fc1.weight"].shape[0], num_hidden_layers=nl,
                         num_attention_heads=int(z["num_heads"]), max_position_embeddings=77, hidden_act=act,
                         bos_token_id=209, eos_token_id=210, pad_token_id=210, projection_dim=64)
    model = CLIPTextModel(cfg).float().eval()
    missing = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k]
    with torch.no_grad():
        live = model(ids)[0]
    assert float((out - live).abs().max()) < 5e-4


def test_clip_shape_table_matches_transformers_schema():
    from stable_diffusion_videos_amd import config as cfgs
    from stable_diffusion_videos_amd.weights import clip_text_shapes, count_params
    _, sd = _clip_golden("quick_gelu")
    cfg = cfgs.TextConfig(vocab_size=211, hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2)
    shapes = clip_text_shapes(cfg)
    assert set(shapes) == set(sd) and all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    assert count_params(clip_text_shapes(cfgs.sd14_text())) == 123_060_480      # CLIP ViT-L/14 text tower


# ---------------------------------------------------------------------------------------------------------------------
# the third-party blocks (oracle/models.py) against a second, independent numpy-float64 restatement
# (tests/golden/make_golden_blocks.py -> tests/golden/blocks_f64.npz) and against torch built-ins
# ---------------------------------------------------------------------------------------------------------------------
def _block_fixture(block):
    z = np.load(GOLDEN / "blocks_f64.npz")
    pre = block + "::"
    params = {k[len(pre) + 3:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith(pre + "p::")}
    arrays = {k[len(pre):]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith(pre) and "::p::" not in k}
    return params, arrays


def _load(mod, params):
    mod = mod.double().eval()
    missing = mod.load_state_dict(params, strict=True)
    return mod


def test_timestep_embedding_closed_form():
    """320-d sinusoids, [cos | sin] of t * exp(-ln(1e4) k / 160) - checked against the committed numpy
    vectors AND a handful of closed-form values written out here."""
    import math
    z = np.load(GOLDEN / "blocks_f64.npz")
    t = torch.from_numpy(z["temb::t"])
    got = models.timestep_embedding(t, 320, True, 0)
    assert np.allclose(got.numpy(), z["temb::flip"], atol=2e-4)          # (the oracle evaluates the angles in fp32)
    assert abs(float(got[0, 0]) - math.cos(981.0)) < 1e-4 and abs(float(got[0, 160]) - math.sin(981.0)) < 1e-4
    k = 7
    f = math.exp(-math.log(10000.0) * k / 160)
    assert abs(float(got[1, k]) - math.cos(501.0 * f)) < 1e-4 and abs(float(got[1, 160 + k]) - math.sin(501.0 * f)) < 1e-4
    got = models.timestep_embedding(t, 32, False, 1)
    assert np.allclose(got.numpy(), z["temb::noflip_shift1"], atol=2e-4)
    p, a = _block_fixture("time_mlp")
    m = _load(models.TimestepEmbedding(16, 24), p)
    with torch.no_grad():
        assert torch.allclose(m(a["x"]), a["y"], atol=1e-10)


@pytest.mark.parametrize("name,cin,cout,temb", [("resnet_shortcut", 16, 24, 12), ("resnet_plain", 16, 16, 12), ("resnet_notemb", 16, 16, None)])
def test_resnet_block_vs_independent_restatement(name, cin, cout, temb):
    p, a = _block_fixture(name)
    m = _load(models.ResnetBlock2D(cin, cout, temb, groups=8, eps=1e-5), p)
    with torch.no_grad():
        got = m(a["x"], a.get("temb"))
    assert got.shape == a["y"].shape and float((got - a["y"]).abs().max()) < 1e-10


@pytest.mark.parametrize("name,lin", [("transformer_conv", False), ("transformer_linear", True)])
def test_transformer_block_vs_independent_restatement(name, lin):
    """GroupNorm(eps 1e-6) -> proj_in -> self-attn / cross-attn / GEGLU FF with pre-LayerNorms -> proj_out -> + input: the
    GEGLU chunk order (value = first half), the head split, the bias-free q/k/v, conv vs linear projections."""
    p, a = _block_fixture(name)
    m = _load(models.Transformer2DModel(heads=2, dim_head=8, channels=16, context_dim=12, groups=4, use_linear_projection=lin), p)
    with torch.no_grad():
        got = m(a["x"], a["ctx"])
    assert float((got - a["y"]).abs().max()) < 1e-10
    # the attention core against torch's own scaled_dot_product_attention
    att = m.transformer_blocks[0].attn2
    x = a["x"].permute(0, 2, 3, 1).reshape(2, 12, 16)
    with torch.no_grad():
        q = att.to_q(x).view(2, 12, 2, 8).transpose(1, 2)
        k = att.to_k(a["ctx"]).view(2, 5, 2, 8).transpose(1, 2)
        v = att.to_v(a["ctx"]).view(2, 5, 2, 8).transpose(1, 2)
        sdpa = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2, 12, 16)
        assert torch.allclose(att(x, a["ctx"]), att.to_out[0](sdpa), atol=1e-10)
        # GEGLU = value * exact (erf) GELU of the gate
        ff = m.transformer_blocks[0].ff.net[0]
        h = ff.proj(x)
        assert torch.allclose(ff(x), h[..., :64] * torch.nn.functional.gelu(h[..., 64:], approximate="none"), atol=1e-12)


def test_samplers_and_vae_attention_vs_independent_restatement():
    p, a = _block_fixture("downsample")
    m = _load(models.Downsample2D(6), p)
    with torch.no_grad():
        assert float((m(a["x"]) - a["y"]).abs().max()) < 1e-10
        p2, a2 = _block_fixture("downsample_odd")
        assert m(a2["x"]).shape == a2["y"].shape == (2, 6, 3, 4) and float((m(a2["x"]) - a2["y"]).abs().max()) < 1e-10
        pu, au = _block_fixture("upsample")
        mu = _load(models.Upsample2D(6), pu)
        assert float((mu(au["x"]) - au["y"]).abs().max()) < 1e-10
        pv, av = _block_fixture("vae_attention")
        mv = _load(models.VAEAttention(16, groups=4), pv)
        assert float((mv(av["x"]) - av["y"]).abs().max()) < 1e-10


def test_committed_block_vectors_are_what_the_generator_produces():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_blocks", GOLDEN / "make_golden_blocks.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fx = mod.build()
    z = np.load(GOLDEN / "blocks_f64.npz")
    assert set(fx) == set(z.files)
    assert all(np.array_equal(fx[k], z[k]) for k in fx)


def _ldm_to_diffusers_decoder_keys(ldm_sd, n_levels):
    """The published checkpoint conversion (diffusers ``convert_ldm_vae_checkpoint``): the SD VAE was trained with the CompVis
    ``ldm`` Decoder and its weights reach ``AutoencoderKL.decoder`` through exactly this renaming - 1x1 attention convs become
    Linear layers, ``nin_shortcut`` -> ``conv_shortcut``, ``norm_out`` -> ``conv_norm_out``, ``mid.block_k`` ->
    ``mid_block.resnets.k-1``, ``up`` levels in processing order -> ``up_blocks``."""
    out = {}
    for k, v in ldm_sd.items():
        k = k.replace("nin_shortcut", "conv_shortcut")
        if k.startswith("mid.block_"):
            k = k.replace("mid.block_1", "mid_block.resnets.0").replace("mid.block_2", "mid_block.resnets.1")
        elif k.startswith("mid.attn_1."):
            k = (k.replace("mid.attn_1.norm", "mid_block.attentions.0.group_norm")
                  .replace("mid.attn_1.q", "mid_block.attentions.0.to_q").replace("mid.attn_1.k", "mid_block.attentions.0.to_k")
                  .replace("mid.attn_1.v", "mid_block.attentions.0.to_v")
                  .replace("mid.attn_1.proj_out", "mid_block.attentions.0.to_out.0"))
            if v.ndim == 4:
                v = v[:, :, 0, 0]
        elif k.startswith("up."):
            _, lvl, kind, rest = k.split(".", 3)
            k = (f"up_blocks.{lvl}.resnets.{rest}" if kind == "block" else f"up_blocks.{lvl}.upsamplers.0.{rest}")
        elif k.startswith("norm_out."):
            k = "conv_" + k
        out[k] = v
    return out


def test_vae_decoder_pinned_against_the_ldm_decoder_in_transformers():
    """A THIRD-PARTY pin for row a15 ([3P] ``AutoencoderKL.decode``): ``transformers`` ships the CompVis ``ldm`` Decoder the SD
    VAE was trained with (``JanusVQVAEDecoder``: GroupNorm(32, eps 1e-6) - swish - conv ResnetBlocks with ``nin_shortcut``,
    single-head ``AttnBlock`` over 1x1 convs scaled by C^-0.5, nearest-2x + conv up-samplers, three ResnetBlocks per level,
    ``norm_out`` - swish - ``conv_out``).  Built at the SD-VAE sizes (128 x (1, 2, 4, 4), 2 + 1 blocks per level, 4 latent
    channels; its extra lowest-level attention blocks - ``attn_resolutions`` is empty in the SD config - are removed) with random
    weights, and carried into the oracle through the published ldm -> diffusers key conversion, it must produce the oracle's output.
    What stays unpinned: that diffusers executes those converted weights the way ldm did (its conversion script says so)."""
    mj = pytest.importorskip("transformers.models.janus.modeling_janus")
    from transformers.models.janus.configuration_janus import JanusVQVAEConfig
    from oracle import models
    cfg = models.sd_vae_config()
    torch.manual_seed(11)
    jc = JanusVQVAEConfig(base_channels=cfg.block_out_channels[0], channel_multiplier=[c // cfg.block_out_channels[0] for c in cfg.block_out_channels],
                          num_res_blocks=cfg.layers_per_block, latent_channels=cfg.latent_channels, out_channels=cfg.out_channels, dropout=0.0)
    ldm = mj.JanusVQVAEDecoder(jc).eval()
    for up in ldm.up:
        up.attn = torch.nn.ModuleList()                      # SD: attn_resolutions = [] (attention in the mid block only)
    with torch.no_grad():
        for n, p in ldm.named_parameters():                  # non-trivial norm scales / shifts and biases everywhere
            p.copy_(torch.randn(p.shape) * (0.05 if p.ndim > 1 else 0.3))
            if "norm" in n and n.endswith("weight"):
                p.add_(1.0)
    ours = models.Decoder(cfg).eval()
    sd = _ldm_to_diffusers_decoder_keys(ldm.state_dict(), len(cfg.block_out_channels))
    ours.load_state_dict(sd, strict=True)                    # every key of either side has its partner
    assert models.count_params(ours) == sum(p.numel() for p in ldm.parameters())
    z = torch.randn(2, cfg.latent_channels, 8, 8)
    with torch.no_grad():
        a = ldm(z.clone())
        b = ours(z)
    assert a.shape == b.shape == (2, 3, 64, 64)
    rel = float((a - b).norm() / a.norm())
    assert rel < 2e-5, rel
    # the blocks one by one as well (the mid attention is where a scale / head-count slip would hide behind the ResBlocks)
    x = torch.randn(2, 512, 8, 8)
    with torch.no_grad():
        assert float((ldm.mid.attn_1(x.clone()) - ours.mid_block.attentions[0](x)).abs().max()) < 1e-4
        assert float((ldm.up[1].upsample(x.clone()) - ours.up_blocks[1].upsamplers[0](x)).abs().max()) < 1e-4
        x2 = torch.randn(2, 512, 8, 8)
        assert float((ldm.up[2].block[0](x2.clone()) - ours.up_blocks[2].resnets[0](x2)).abs().max()) < 1e-4   # 512 -> 256, nin_shortcut


def test_basic_transformer_block_pinned_against_torch_transformer_decoder_layer():
    """A THIRD-PARTY pin for row a12 ([3P] ``BasicTransformerBlock``): torch's own ``nn.TransformerDecoderLayer`` with
    ``norm_first=True`` IS the block's data flow - x + SelfAttn(LN1 x), x + CrossAttn(LN2 x, context), x + FF(LN3 x), LayerNorm eps
    1e-5 - executed by code written by other people (``nn.MultiheadAttention``: packed q/k/v projection, the split into heads,
    1/sqrt(dh), softmax, merge, out projection with bias).  The two places diffusers differs are configured, not re-implemented:
    a context of another width (``kdim`` / ``vdim``) and the GEGLU feed-forward (``activation`` = value * exact GELU of the gate on
    ``linear1``'s fused output).  Random weights carried over by name; q / k / v have no bias in diffusers, so the in-projection
    biases are zero."""
    from oracle import models
    torch.manual_seed(5)
    d, heads, dh, ctx_dim, inner = 48, 4, 12, 40, 192
    ours = models.BasicTransformerBlock(d, heads, dh, ctx_dim).double().eval()
    with torch.no_grad():
        for n, p in ours.named_parameters():
            p.copy_(torch.randn(p.shape, dtype=torch.float64) * (0.15 if p.ndim > 1 else 0.3))
            if "norm" in n and n.endswith("weight"):
                p.add_(1.0)
    ref = torch.nn.TransformerDecoderLayer(d, heads, dim_feedforward=2 * inner, dropout=0.0, batch_first=True, norm_first=True,
                                           activation=lambda h: h[..., :inner] * torch.nn.functional.gelu(h[..., inner:]))
    ref.multihead_attn = torch.nn.MultiheadAttention(d, heads, dropout=0.0, batch_first=True, kdim=ctx_dim, vdim=ctx_dim)
    ref.linear2 = torch.nn.Linear(inner, d)
    ref = ref.double().eval()
    with torch.no_grad():
        a1, a2 = ours.attn1, ours.attn2
        ref.self_attn.in_proj_weight.copy_(torch.cat([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight]))
        ref.self_attn.in_proj_bias.zero_()
        ref.self_attn.out_proj.weight.copy_(a1.to_out[0].weight)
        ref.self_attn.out_proj.bias.copy_(a1.to_out[0].bias)
        ref.multihead_attn.q_proj_weight.copy_(a2.to_q.weight)
        ref.multihead_attn.k_proj_weight.copy_(a2.to_k.weight)
        ref.multihead_attn.v_proj_weight.copy_(a2.to_v.weight)
        ref.multihead_attn.in_proj_bias.zero_()
        ref.multihead_attn.out_proj.weight.copy_(a2.to_out[0].weight)
        ref.multihead_attn.out_proj.bias.copy_(a2.to_out[0].bias)
        for k in (1, 2, 3):
            getattr(ref, f"norm{k}").weight.copy_(getattr(ours, f"norm{k}").weight)
            getattr(ref, f"norm{k}").bias.copy_(getattr(ours, f"norm{k}").bias)
        ref.linear1.weight.copy_(ours.ff.net[0].proj.weight)
        ref.linear1.bias.copy_(ours.ff.net[0].proj.bias)
        ref.linear2.weight.copy_(ours.ff.net[2].weight)
        ref.linear2.bias.copy_(ours.ff.net[2].bias)
        x = torch.randn(3, 20, d, dtype=torch.float64)
        ctx = torch.randn(3, 7, ctx_dim, dtype=torch.float64) * 2.0
        got, want = ours(x, ctx), ref(x, ctx)
    assert got.shape == want.shape == (3, 20, d)
    assert float((got - want).abs().max()) < 1e-10 * float(want.abs().max())
    # ... and the comparison has teeth: swapping the GEGLU halves or dropping the softmax scale is far outside it
    with torch.no_grad():
        w = ours.ff.net[0].proj.weight.clone()
        ours.ff.net[0].proj.weight.copy_(torch.cat([w[inner:], w[:inner]]))
        assert float((ours(x, ctx) - want).abs().max()) > 1e-2

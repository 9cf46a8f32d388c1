"""The fused GEGLU feed-forward (sdv_ffn_geglu_bf16, csrc/sdv_ffn.hip; torch.ops.sdv.k_ffn_geglu) against a float64 evaluation of
norm3 -> ff.net.0 -> GEGLU -> ff.net.2 -> + residual (diffusers' BasicTransformerBlock inside unet(...),
stable_diffusion_pipeline.py:418) on the operands the kernel sees, element by element, and against the two-launch form it replaces."""
import pytest
import torch
import torch.nn.functional as F

from conftest import report

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32
C = 320


def _weights(dev, seed=0):
    from stable_diffusion_videos_amd.weights import ffn_fold_columns, ffn_w2_permute, geglu_interleave, ln_fold
    g = torch.Generator().manual_seed(seed)
    w1 = torch.randn(8 * C, C, generator=g) * C ** -0.5
    b1 = torch.randn(8 * C, generator=g) * 0.2
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).to(BF16)
    b2 = torch.randn(C, generator=g) * 0.2
    gamma, beta = 1.0 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    w1p, s1, t1 = ln_fold(geglu_interleave(w1), gamma, beta, geglu_interleave(b1), dev)
    return dict(w1p=w1p, s1=s1, t1=t1, w1x=ffn_fold_columns(s1, t1), w2=w2.to(dev), w2p=ffn_w2_permute(w2).to(dev), b2=b2.to(dev, F32))


def _rows(M, dev, seed, common_mode=3.0, K=C):
    """rows like the UNet's residual stream: a common mode of several sigma, sigma between 0.03 and 2"""
    g = torch.Generator(device=dev).manual_seed(seed)
    mu = torch.randn(M, 1, device=dev, generator=g) * common_mode
    sd = torch.exp(torch.empty(M, 1, device=dev).uniform_(-3.4, 0.7, generator=g))
    x = (mu + sd * torch.randn((M, K), device=dev, generator=g)).to(BF16)
    xf = x.float()
    st = torch.stack([xf.mean(1), torch.rsqrt(xf.var(1, unbiased=False) + 1e-5)], 1).contiguous()   # what the producer's epilogue emits
    return x, st


def _deinterleave(v):
    """undo weights.geglu_interleave on the last axis: [.., 32-blocks of (16 value | 16 gate)] -> (value [.., 4C], gate [.., 4C])"""
    b = v.reshape(*v.shape[:-1], -1, 2, 16)
    return b[..., 0, :].reshape(*v.shape[:-1], -1), b[..., 1, :].reshape(*v.shape[:-1], -1)


def _reference(x, st, P):
    """float64 on the bf16 operands: the fold exactly as sdv_hip.h states it, the hidden activations rounded to bf16 once (as both
    launch forms do), one rounding at the end left to the comparison"""
    x64, m64, r64 = x.double(), st[:, :1].double(), st[:, 1:].double()
    w1 = P["w1p"].double()
    pre = (x64 @ w1.T - m64 * P["s1"].double()[None]) * r64 + P["t1"].double()[None]
    mag1 = (x64.abs() @ w1.abs().T + m64.abs() * P["s1"].double().abs()[None]) * r64 + P["t1"].double().abs()[None]
    v, gt = _deinterleave(pre)
    hid = (v * F.gelu(gt)).to(BF16).double()
    w2 = P["w2"].double()
    ref = x64 + hid @ w2.T + P["b2"].double()[None]
    mag2 = x64.abs() + hid.abs() @ w2.abs().T + P["b2"].double().abs()[None]
    return ref, mag2, hid, float((mag1.max()))


@pytest.mark.parametrize("M", [128, 1000, 4096 * 9 + 77, 4096 * 40])
def test_ffn_geglu_is_deterministic_and_elementwise_bounded(hip, dev, M):
    """Every element of the fused launch against float64 within half a bf16 ulp of the result (the one rounding it performs) +
    1e-5 of the magnitudes summed (fp32 accumulation) + what a hidden activation that rounded the other way can move it by (the
    kernel's fp32 pre-activations differ from float64 in the last bits, so a few of the 1280 bf16 hidden values per row land on the
    neighbouring bf16: at most 4 such flips of the largest term are allowed for).  Four launches must agree bit for bit: one wave per
    SIMD walks panels with its weight stream running ahead across panel boundaries - a slab read before it landed would show here."""
    P = _weights(dev)
    x, st = _rows(M, dev, 7 + M)
    outs = [hip.ffn_geglu(x, st, P["w1p"], P["w1x"], P["w2p"], P["b2"]) for _ in range(4)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16)), "repeated launches differ"
    worst = 0.0
    for lo in range(0, M, 32768):            # (float64 in slabs: the 163 840-row case would need 3 GB at once)
        sl = slice(lo, min(M, lo + 32768))
        ref, mag2, hid, _ = _reference(x[sl], st[sl], P)
        ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)
        hid_ulp = torch.exp2(torch.floor(torch.log2(hid.abs().clamp_min(1e-30))) - 7)
        flip = 4.0 * (hid_ulp * 1.0).max(1, keepdim=True).values * P["w2"].double().abs().max(1).values[None]
        ratio = (outs[0][sl].double() - ref).abs() / (0.5 * ulp * (1 + 1e-3) + 1e-5 * mag2 + flip)
        worst = max(worst, float(ratio.max()))
        assert bool(torch.isfinite(outs[0][sl].float()).all())
    report(f"fused GEGLU feed-forward, M={M}: 4 launches bit-identical, worst element at {worst:.3f} of (half ulp + 1e-5 magnitudes + hidden flips)")
    assert worst <= 1.0


def test_ffn_geglu_agrees_with_the_two_launch_form(hip, dev):
    """ff.net.0 with the GEGLU epilogue + LayerNorm fold (sdv_gemm_bf16 epi 1, ln_side 1), then ff.net.2 + residual: the same
    roundings in the same places, so the two forms may differ by the fp32 summation order only - rarely, and by one bf16 ulp."""
    M = 4096 * 6 + 5
    P = _weights(dev, seed=3)
    x, st = _rows(M, dev, 99)
    a = hip.ffn_geglu(x, st, P["w1p"], P["w1x"], P["w2p"], P["b2"])
    g = hip.linear(x, P["w1p"], P["t1"], epi=1, ln=(st, P["s1"]))
    b = hip.linear(g, P["w2"], P["b2"], residual=x)
    torch.cuda.synchronize()
    differ = float((a != b).float().mean())
    # (ulp at the largest of the two values and the residual they both add: an output that cancels is the difference of larger terms)
    big = torch.maximum(torch.maximum(a.float().abs(), b.float().abs()), x.float().abs()).clamp_min(2.0 ** -4)
    ulp = torch.exp2(torch.floor(torch.log2(big)) - 7)
    worst = float(((a.float() - b.float()).abs() / ulp).max())
    report(f"fused vs two-launch feed-forward: {100 * differ:.3f} % of the elements differ, by at most {worst:.2f} bf16 ulp")
    assert differ < 0.01 and worst <= 2.0


def test_ffn_geglu_common_mode_does_not_leak(hip, dev):
    """The fold terms ride in the matrix product as three-way bf16 splits: a row that is a constant (LayerNorm of it = beta) must
    come out as x + b2 + W2 GEGLU(W1 beta + b1) whatever the constant - the - mean s term has to cancel x W'^T to fp32 accuracy."""
    P = _weights(dev, seed=5)
    vals = torch.tensor([0.0, 1.0, -7.5, 100.0, 3e-3], device=dev)
    x = vals[:, None].expand(5, C).to(BF16).contiguous()
    xf = x.float()
    st = torch.stack([xf.mean(1), torch.rsqrt(xf.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    out = hip.ffn_geglu(x, st, P["w1p"], P["w1x"], P["w2p"], P["b2"])
    v, gt = _deinterleave(P["t1"].double()[None])
    expect = (v * F.gelu(gt)).to(BF16).double() @ P["w2"].double().T + P["b2"].double()[None]      # the same for every row
    got = out.double() - x.double()
    tol = 2.0 ** -7 * (x.double().abs() + expect.abs()) + 2e-3
    assert bool(((got - expect).abs() <= tol).all()), float(((got - expect).abs() / tol).max())


# ------------------------------------------------------------------------------------------------
# the C = 320 projections on the panel kernel (sdv_linear320_bf16)
# ------------------------------------------------------------------------------------------------
def _lin_forms(hip, dev, K=C):
    """the projections of a transformer block with K channels (K = 320: sdv_linear320_bf16, K = 640: sdv_linear640_bf16)"""
    from stable_diffusion_videos_amd.weights import ffn_fold_columns, ln_fold
    g = torch.Generator().manual_seed(21 + K // 640)
    qs = hip.q_prescale(40)
    nb = K // C
    gamma, beta = 1.0 + 0.3 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    w = lambda: torch.randn(K, K, generator=g) * K ** -0.5
    bias = (torch.randn(K, generator=g) * 0.2).to(dev)
    forms = {"bias": dict(w=w().to(dev, BF16), s=torch.zeros(K, device=dev), t=bias, alpha=None)}
    parts = [ln_fold(w(), gamma, beta, None, dev) for _ in range(3)]
    W3, s3, t3 = (torch.cat([p_[j] for p_ in parts]).contiguous() for j in range(3))
    forms["qkv"] = dict(w=W3, s=s3, t=t3, alpha=torch.tensor([qs] * nb + [1.0] * (2 * nb), device=dev))
    wq, sq, tq = ln_fold(w(), gamma, beta, None, dev)
    forms["q2"] = dict(w=wq, s=sq, t=tq, alpha=torch.tensor([qs] * nb, device=dev))
    for f in forms.values():
        f["wx"] = ffn_fold_columns(f["s"], f["t"])
    return forms


@pytest.mark.parametrize("form,use_res,K", [("bias", False, 320), ("bias", True, 320), ("qkv", False, 320), ("q2", False, 320),
                                            ("bias", False, 640), ("qkv", False, 640), ("q2", False, 640)])
@pytest.mark.parametrize("M", [100, 4096 * 5 + 33, 4096 * 36])
def test_linear320_is_deterministic_and_elementwise_bounded(hip, dev, form, use_res, K, M):
    """proj_in / attn.to_out / attn2.to_q / the fused Q K V projection of the C = 320 transformer blocks (Transformer2DModel inside
    unet(...), stable_diffusion_pipeline.py:418) on the panel kernel: every element against float64 within half a bf16 ulp + 2e-5
    of the magnitudes that went into it (rows with a common mode of several sigma: the fold's - mean s rides in the matrix product
    as three-way bf16 splits and has to cancel to fp32 accuracy), four launches bit for bit equal, and the LayerNorm statistics it
    emits for its consumer against the statistics of the rows it stored."""
    f = _lin_forms(hip, dev, K)[form]
    x, st = _rows(M, dev, 31 + M, K=K)
    fold = form != "bias"
    r = (torch.randn((M, K), device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 2.0).to(BF16) if use_res else None
    run = lambda: hip.linear320(x, f["w"], f["wx"], ln_stats=st if fold else None, alpha=f["alpha"], residual=r, want_stats=not fold)
    outs = [run() for _ in range(4)]
    torch.cuda.synchronize()
    y0, s0 = outs[0] if not fold else (outs[0], None)
    for o in outs[1:]:
        yo = o[0] if not fold else o
        assert torch.equal(yo.view(torch.int16), y0.view(torch.int16)), "repeated launches differ"
    N = f["w"].shape[0]
    al = torch.ones(N, dtype=torch.float64, device=dev)
    if f["alpha"] is not None:
        al = f["alpha"].double().repeat_interleave(C)
    worst = 0.0
    for lo in range(0, M, 16384):
        sl = slice(lo, min(M, lo + 16384))
        x64, w64 = x[sl].double(), f["w"].double()
        m64 = st[sl, :1].double() if fold else torch.zeros((x64.shape[0], 1), dtype=torch.float64, device=dev)
        r64 = st[sl, 1:].double() if fold else torch.ones((x64.shape[0], 1), dtype=torch.float64, device=dev)
        ref = (x64 @ w64.T - m64 * f["s"].double()[None]) * (r64 * al[None]) + (f["t"].double() * al)[None]
        mag = (x64.abs() @ w64.abs().T + m64.abs() * f["s"].double().abs()[None]) * (r64 * al[None]) + (f["t"].double() * al).abs()[None]
        if use_res:
            ref, mag = ref + r[sl].double(), mag + r[sl].double().abs()
        ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)
        worst = max(worst, float(((y0[sl].double() - ref).abs() / (0.5 * ulp * (1 + 1e-3) + 2e-5 * mag)).max()))
    msg = f"linear{K} {form}{' + residual' if use_res else ''}, M={M}: 4 launches bit-identical, worst element at {worst:.3f} of (half ulp + 2e-5 magnitudes)"
    if s0 is not None:
        yf = y0.float()
        mean, rstd = yf.mean(1), torch.rsqrt(yf.var(1, unbiased=False) + 1e-5)
        e_mean = float((s0[:, 0] - mean).abs().max() / yf.abs().max())
        e_rstd = float(((s0[:, 1] - rstd) / rstd).abs().max())
        msg += f"; emitted LayerNorm statistics: mean within {e_mean:.1e} of the row scale, rstd within {e_rstd:.1e} (relative)"
        assert e_mean < 1e-5 and e_rstd < 2e-3
    report(msg)
    assert worst <= 1.0


@pytest.mark.parametrize("K", [320, 640])
def test_linear320_agrees_with_the_igemm_form(hip, dev, K):
    """The same projection as sdv_gemm_bf16 launches it (LayerNorm fold, alpha on the Q third): same roundings, fp32 summation order
    aside - and the statistics both forms emit lead a consumer to the same normalisation."""
    M = 4096 * 3 + 5
    f = _lin_forms(hip, dev, K)["qkv"]
    qs = hip.q_prescale(40)
    x, st = _rows(M, dev, 77, K=K)
    a = hip.linear320(x, f["w"], f["wx"], ln_stats=st, alpha=f["alpha"])
    t_scaled = torch.cat([f["t"][:K] * qs, f["t"][K:]])
    b = hip.linear(x, f["w"], t_scaled, alpha=qs, alpha_cols=K, ln=(st, f["s"]))
    fb = _lin_forms(hip, dev, K)["bias"]
    ya, sa = hip.linear320(x, fb["w"], fb["wx"], want_stats=True)
    yb, sb = hip.linear(x, fb["w"], fb["t"], want_stats=True)
    torch.cuda.synchronize()
    ulp = torch.exp2(torch.floor(torch.log2(torch.maximum(a.float().abs(), b.float().abs()).clamp_min(2.0 ** -6))) - 7)
    assert float((a != b).float().mean()) < 0.01 and float(((a.float() - b.float()).abs() / ulp).max()) <= 2.0
    assert torch.equal(ya, yb) or float((ya != yb).float().mean()) < 0.01
    assert float((sa[:, 0] - sb[:, 0]).abs().max()) < 1e-4 * float(ya.float().abs().max()) and float(((sa[:, 1] - sb[:, 1]) / sb[:, 1]).abs().max()) < 2e-3


@pytest.mark.parametrize("hw,nimg,pad", [(128, 3, 0), (1024, 5, 64), (4096, 4, 0), (4096, 37, 0)])
def test_linear320_transposed_v_is_the_same_rows_moved(hip, dev, hw, nimg, pad):
    """The fused Q K V projection with ``Vt``: attention's P V product wants V with the token axis contiguous (the attention forward
    inside unet(...), stable_diffusion_pipeline.py:418), so the panel kernel stores the V third as [sample][channel][token] straight
    from its epilogue.  A layout move only: [Q | K] and V^T hold bit for bit the values of the plain [M, 960] launch (sample counts
    that leave the persistent grid a ragged tail, a padded ldvt), and what lies beyond hw in a padded V^T row is not touched."""
    M = hw * nimg
    f = _lin_forms(hip, dev)["qkv"]
    x, st = _rows(M, dev, 91 + hw)
    full = hip.linear320(x, f["w"], f["wx"], ln_stats=st, alpha=f["alpha"])
    store = torch.full((nimg, C, hw + pad), 7.0, dtype=BF16, device=dev)
    vt = store[:, :, :hw]
    qk = hip.linear320(x, f["w"], f["wx"], ln_stats=st, alpha=f["alpha"], vt=vt, hw=hw)
    torch.cuda.synchronize()
    assert qk.shape == (M, 2 * C) and torch.equal(qk, full[:, :2 * C])
    assert torch.equal(vt, full[:, 2 * C:].reshape(nimg, hw, C).transpose(1, 2))
    assert pad == 0 or bool((store[:, :, hw:] == 7.0).all())
    report(f"linear320 Q K V with V^T, {nimg} samples of {hw} tokens (ldvt {hw + pad}): [Q | K] and V^T bit-identical to the row-major launch")

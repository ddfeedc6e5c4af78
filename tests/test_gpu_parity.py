"""Parity of the HIP path (through the C ABI) against (a) golden vectors captured from the reference
and (b) the CPU oracle on seeded inputs.  Needs an MI355X: `pytest -m gpu`.

Tolerances: BASELINE.json:north_star states 1e-3 relative fp32 on outputs; the tests hold forward
results to 1e-4 (max-abs error / max-abs reference) and single-op gradients to 1e-3."""
import math

import pytest
import torch

import seeded
from where2edit_amd.stylegan2 import freeze_conv_weights
from helpers import assert_close, assert_grad_close, golden, rel_err
from make_golden import MODCONV_CASES, UPFIRDN_CASES, _kernel, modconv_inputs
from oracle import ops as O
from oracle import stylegan2 as OG

pytestmark = pytest.mark.gpu
DEV = "cuda"
FWD_TOL = 1e-4
GRAD_TOL = 1e-3


def cu(t):
    return t.to(DEV)


# ------------------------------------------------------------------------------------------ upfirdn2d
@pytest.mark.parametrize("case", UPFIRDN_CASES, ids=[c[0] for c in UPFIRDN_CASES])
def test_upfirdn2d_golden(case):
    from where2edit_amd.op import upfirdn2d
    g = golden("ops")
    name, shape, kspec, gain, up, down, pad = case
    x = cu(seeded.tensor("upfirdn." + name, shape)).requires_grad_(True)
    y = upfirdn2d(x, cu(_kernel(kspec, gain)), up=up, down=down, pad=pad)
    assert_close(y, g[f"upfirdn.{name}.y"], 1e-5, "y")
    (gx,) = torch.autograd.grad(y, x, cu(seeded.tensor("upfirdn.gy." + name, y.shape)))
    assert_close(gx, g[f"upfirdn.{name}.gx"], 1e-5, "gx")


@pytest.mark.parametrize("shape,up,down,pad,ktaps", [
    ((2, 8, 65, 65), 1, 1, (1, 1), (1, 3, 3, 1)),      # tile kernel, ragged edge
    ((1, 3, 129, 257), 1, 1, (1, 1), (1, 3, 3, 1)),    # 2H+1 sizes, non-square
    ((2, 3, 32, 32), 2, 1, (2, 1), (1, 3, 3, 1)),      # RGB skip up-sampling
    ((1, 4, 64, 64), 1, 2, (1, 1), (1, 3, 3, 1)),      # down-sampling
    ((1, 2, 40, 24), 1, 1, (3, 3), (1, 4, 6, 4, 1)),   # 5-tap kernel through the tile kernel
    ((1, 1, 16, 16), 1, 1, (0, 0), (1,)),              # 1x1 kernel = identity
])
def test_upfirdn2d_vs_oracle_and_adjoint(shape, up, down, pad, ktaps):
    from where2edit_amd.op import upfirdn2d
    k = O.make_kernel(ktaps) * (up ** 2)
    x = seeded.tensor("ufd.x", shape)
    y_ref = O.upfirdn2d(x, k, up, down, pad)
    xg = cu(x).requires_grad_(True)
    y = upfirdn2d(xg, cu(k), up=up, down=down, pad=pad)
    assert_close(y, y_ref, 1e-5)
    gy = seeded.tensor("ufd.gy", y.shape)
    (gx,) = torch.autograd.grad(y, xg, cu(gy))
    # exact adjoint: <A x, gy> == <x, A^T gy>
    lhs = (y.detach().double().cpu() * gy.double()).sum()
    rhs = (x.double() * gx.double().cpu()).sum()
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))


def test_upfirdn2d_full_size_property():
    """BASELINE size (32 ch at 1025^2 -> 1024^2): linearity + DC gain of the blur (kernel sums to 4)."""
    from where2edit_amd.op import upfirdn2d
    k = cu(O.make_kernel((1, 3, 3, 1)) * 4)
    x = torch.randn(1, 32, 1025, 1025, device=DEV)
    ones = torch.ones(1, 1, 1025, 1025, device=DEV)
    y1 = upfirdn2d(ones, k, pad=(1, 1))
    assert y1.shape[-1] == 1024
    assert (y1[:, :, 2:-2, 2:-2] - 4.0).abs().max() < 1e-5
    a, b = upfirdn2d(x, k, pad=(1, 1)), upfirdn2d(2.5 * x, k, pad=(1, 1))
    assert (b - 2.5 * a).abs().max() <= 1e-5 * a.abs().max()


# ------------------------------------------------------------------------------------------ fused_leaky_relu
@pytest.mark.parametrize("name,shape", [("nchw", (2, 6, 5, 7)), ("seq3d", (2, 4, 6)), ("mat2d", (3, 6))])
def test_fused_leaky_relu_golden(name, shape):
    from where2edit_amd.op import fused_leaky_relu
    g = golden("ops")
    x = cu(seeded.tensor("flrelu." + name, shape)).requires_grad_(True)
    c = shape[-1] if len(shape) == 3 else shape[1]
    b = cu(seeded.tensor("flrelu.b." + name, (c,))).requires_grad_(True)
    y = fused_leaky_relu(x, b)
    gx, gb = torch.autograd.grad(y, (x, b), cu(seeded.tensor("flrelu.gy." + name, y.shape)))
    assert_close(y, g[f"flrelu.{name}.y"], 1e-6)
    assert_close(gx, g[f"flrelu.{name}.gx"], 1e-6)
    assert_close(gb, g[f"flrelu.{name}.gb"], 1e-5)


def test_fused_leaky_relu_module_and_slope():
    from where2edit_amd.op import FusedLeakyReLU, fused_leaky_relu
    g = golden("ops")
    y = fused_leaky_relu(cu(seeded.tensor("flrelu.slope", (2, 3, 4, 4))), cu(seeded.tensor("flrelu.slope.b", (3,))), 0.1, 1.5)
    assert_close(y, g["flrelu.slope.y"], 1e-6)
    m = FusedLeakyReLU(512).to(DEV)
    x = torch.randn(4, 512, 64, 64, device=DEV)
    assert_close(m(x), O.fused_leaky_relu(x.cpu(), m.bias.detach().cpu()), 1e-6)
    x = torch.randn(3, 18, 512, device=DEV)  # the 3-D branch (bias on the last dim) the mappers use
    assert_close(m(x), O.fused_leaky_relu(x.cpu(), m.bias.detach().cpu()), 1e-6)


# ------------------------------------------------------------------------------------------ ModulatedConv2d
def _our_modconv(cin, cout, k, demod, up, i):
    from where2edit_amd.stylegan2 import ModulatedConv2d
    m = ModulatedConv2d(cin, cout, k, 512, demodulate=demod, upsample=up)
    sd = {"weight": i["weight"], "modulation.weight": i["mod_w"], "modulation.bias": i["mod_b"]}
    if up:
        sd["blur.kernel"] = seeded.fir_kernel(gain=4.0)
    m.load_state_dict(sd, strict=True)
    return freeze_conv_weights(m.to(DEV))


@pytest.mark.parametrize("case", MODCONV_CASES, ids=[c[0] for c in MODCONV_CASES])
def test_modulated_conv_golden(case):
    g = golden("modconv")
    name, cin, cout, k, demod, up, b, h = case
    i = modconv_inputs(name, cin, cout, k, b, h)
    m = _our_modconv(cin, cout, k, demod, up, i)
    x, w = cu(i["x"]).requires_grad_(True), cu(i["w"]).requires_grad_(True)
    y, s = m(x, w)
    assert_close(y, g[f"{name}.y"], FWD_TOL, "y")
    assert_close(s, g[f"{name}.s"], 1e-5, "s")
    gx, gw = torch.autograd.grad(y, (x, w), cu(seeded.tensor(f"modconv.{name}.gy", y.shape)))
    assert_close(gx, g[f"{name}.gx"], GRAD_TOL, "gx")
    assert_close(gw, g[f"{name}.gw"], GRAD_TOL, "gw")
    with torch.no_grad():
        y2, _ = m(x.detach(), s.detach(), input_is_stylespace=True)
    assert_close(y2, g[f"{name}.y_sspace"], FWD_TOL, "sspace")


# every MFMA tile configuration / layer shape class of the 1024 generator, at batch 2
STYLED_SHAPES = [
    # cin, cout, H, up
    (512, 512, 4, False), (512, 512, 4, True), (512, 512, 8, False), (512, 512, 16, False), (512, 512, 16, True),
    (512, 256, 32, True), (256, 256, 64, False), (128, 128, 64, False), (128, 64, 32, True), (64, 64, 128, False),
    (64, 32, 64, True), (32, 32, 128, False), (24, 40, 12, False), (40, 24, 6, True), (10, 6, 5, False), (6, 10, 5, True),
]


@pytest.mark.parametrize("cin,cout,h,up", STYLED_SHAPES)
def test_styled_conv_vs_oracle(cin, cout, h, up):
    """StyledConv (fused conv+noise+bias+lrelu, blur for up) forward and all gradients vs the oracle."""
    from where2edit_amd.stylegan2 import StyledConv
    b = 2
    key = f"sc.{cin}.{cout}.{h}.{int(up)}"
    sd = {"conv.weight": seeded.tensor(key + ".w", (1, cout, cin, 3, 3)),
          "conv.modulation.weight": seeded.tensor(key + ".mw", (cin, 512)),
          "conv.modulation.bias": seeded.tensor(key + ".mb", (cin,), 0.05, 1.0),
          "noise.weight": seeded.tensor(key + ".nw", (1,), 0.3),
          "activate.bias": seeded.tensor(key + ".ab", (cout,), 0.3)}
    if up:
        sd["conv.blur.kernel"] = seeded.fir_kernel(gain=4.0)
    oh = 2 * h if up else h
    x = seeded.tensor(key + ".x", (b, cin, h, h))
    w = seeded.tensor(key + ".wl", (b, 512))
    noise = seeded.tensor(key + ".noise", (1, 1, oh, oh))
    gy = seeded.tensor(key + ".gy", (b, cout, oh, oh))
    # oracle
    osd = {"p." + k: v.clone().requires_grad_(k in ("noise.weight", "activate.bias")) for k, v in sd.items()}
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yo, so = OG.styled_conv(osd, "p", xo, wo, noise, upsample=up, input_is_stylespace=False)
    # A pre-activation within fp32 rounding of 0 may legitimately take the other LeakyReLU slope in a
    # different (equally valid) summation order; zero the incoming gradient there so the comparison
    # measures arithmetic, not which side of the kink a rounding error fell on.
    gy = gy * (yo.detach().abs() > 1e-4)
    go = torch.autograd.grad(yo, (xo, wo, osd["p.noise.weight"], osd["p.activate.bias"]), gy)
    # HIP
    m = StyledConv(cin, cout, 3, 512, upsample=up)
    m.load_state_dict(sd, strict=True)
    m = freeze_conv_weights(m.to(DEV))
    xg, wg = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
    y, s = m(xg, wg, noise=cu(noise))
    assert_close(y, yo, FWD_TOL, "y")
    assert_close(s, so, 1e-5, "s")
    gg = torch.autograd.grad(y, (xg, wg, m.noise.weight, m.activate.bias), cu(gy))
    assert_close(gg[0], go[0], GRAD_TOL, "gx")
    for a, r, what in zip(gg[1:], go[1:], ("g_latent", "g_noise_w", "g_bias")):
        assert_close(a, r, GRAD_TOL, what)


@pytest.mark.parametrize("cin,h,with_skip", [(512, 4, False), (512, 8, True), (256, 64, True), (32, 256, True), (20, 6, True)])
def test_to_rgb_vs_oracle(cin, h, with_skip):
    from where2edit_amd.stylegan2 import ToRGB
    b = 2
    key = f"rgb.{cin}.{h}"
    sd = {"bias": seeded.tensor(key + ".b", (1, 3, 1, 1), 0.3), "conv.weight": seeded.tensor(key + ".w", (1, 3, cin, 1, 1)),
          "conv.modulation.weight": seeded.tensor(key + ".mw", (cin, 512)),
          "conv.modulation.bias": seeded.tensor(key + ".mb", (cin,), 0.05, 1.0)}
    if with_skip:
        sd["upsample.kernel"] = seeded.fir_kernel(gain=4.0)
    x = seeded.tensor(key + ".x", (b, cin, h, h))
    w = seeded.tensor(key + ".wl", (b, 512))
    skip = seeded.tensor(key + ".skip", (b, 3, h // 2, h // 2)) if with_skip else None
    gy = seeded.tensor(key + ".gy", (b, 3, h, h))
    osd = {"p." + k: v.clone() for k, v in sd.items()}
    osd["p.bias"].requires_grad_(True)
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    so_ = skip.clone().requires_grad_(True) if with_skip else None
    yo, _ = OG.to_rgb(osd, "p", xo, wo, so_, input_is_stylespace=False)
    go = torch.autograd.grad(yo, [xo, wo, osd["p.bias"]] + ([so_] if with_skip else []), gy)
    m = ToRGB(cin, 512, upsample=with_skip)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    xg, wg = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
    sg = cu(skip).requires_grad_(True) if with_skip else None
    y, _ = m(xg, wg, sg)
    assert_close(y, yo, FWD_TOL, "y")
    gg = torch.autograd.grad(y, [xg, wg, m.bias] + ([sg] if with_skip else []), cu(gy))
    for a, r, what in zip(gg, go, ("gx", "g_latent", "g_bias", "g_skip")):
        assert_close(a, r, GRAD_TOL, what)


# ------------------------------------------------------------------------------------------ K5 / K6
@pytest.mark.parametrize("size", [1024, 256])
def test_clip_preprocess_golden(size):
    from where2edit_amd.functional import clip_preprocess
    g = golden("preproc")
    img = cu(seeded.tensor(f"preproc.img{size}", (1, 3, size, size))).requires_grad_(True)
    y = clip_preprocess(img)
    assert_close(y, g[f"clip{size}.y"], 1e-5)
    (gx,) = torch.autograd.grad(y, img, cu(seeded.tensor(f"preproc.gy{size}", y.shape)))
    assert_close(gx[:, :, :: size // 64, :: size // 64], g[f"clip{size}.gx_strided"], 1e-5)
    assert abs(gx.double().sum().item() - float(g[f"clip{size}.gx_sum"])) <= 1e-4 * abs(float(g[f"clip{size}.gx_sum"])) + 1e-4


@pytest.mark.parametrize("c,h,ms", [(512, 64, 64), (3, 64, 64), (16, 32, 8), (5, 12, 5)])
def test_mask_blend_vs_oracle(c, h, ms):
    from where2edit_amd.functional import mask_blend
    a, b = seeded.tensor("mb.a", (2, c, h, h)), seeded.tensor("mb.b", (2, c, h, h))
    mask = torch.rand(2, 1, ms, ms, generator=torch.Generator().manual_seed(1))
    ao, bo, mo = a.clone().requires_grad_(True), b.clone().requires_grad_(True), mask.clone().requires_grad_(True)
    yo = O.mask_blend(ao, bo, mo)
    gy = seeded.tensor("mb.gy", yo.shape)
    go = torch.autograd.grad(yo, (ao, bo, mo), gy)
    ag, bg, mg = cu(a).requires_grad_(True), cu(b).requires_grad_(True), cu(mask).requires_grad_(True)
    y = mask_blend(ag, bg, mg)
    assert_close(y, yo, 1e-6)
    gg = torch.autograd.grad(y, (ag, bg, mg), cu(gy))
    assert_close(gg[0], go[0], 1e-6), assert_close(gg[1], go[1], 1e-6), assert_close(gg[2], go[2], 1e-4)


# ------------------------------------------------------------------------------------------ generators
def _gen(size, cls=None):
    from where2edit_amd.stylegan2 import Generator
    g = (cls or Generator)(size, 512, 8)
    g.load_state_dict(seeded.generator_state_dict(size), strict=True)
    return freeze_conv_weights(g.to(DEV).eval())


def test_style_codes_equal_the_generator_pass():
    """Generator.style_codes: the W+ latent and the 26 S-space codes that forward(..., return_latents=True) returns, without
    the synthesis pass (z, W and W+ inputs, truncation)."""
    g = _gen(64)
    gen = torch.Generator().manual_seed(11)
    z, wp = torch.randn(3, 512, generator=gen).to(DEV), torch.randn(3, g.n_latent, 512, generator=gen).to(DEV)
    mean = g.mean_latent(256)
    for styles, kw in (([z], {}), ([z], {"truncation": 0.7, "truncation_latent": mean}), ([wp], {"input_is_latent": True})):
        with torch.no_grad():
            _, lat_ref, codes_ref = g(styles, return_latents=True, randomize_noise=False, **kw)
            lat, codes = g.style_codes(styles, **kw)
        assert torch.equal(lat, lat_ref) and len(codes) == len(codes_ref) == 2 * g.n_latent - 10 or len(codes) == len(codes_ref)
        for a, b in zip(codes, codes_ref):
            assert a.shape == b.shape and torch.equal(a, b)


def test_generator16_golden_all_modes():
    from where2edit_amd.attention_model import Generator as AttGenerator
    g = golden("generator16")
    gen = _gen(16)
    w = cu(seeded.wplus_latents(2, 6)).requires_grad_(True)
    img, lat, svec = gen([w], input_is_latent=True, randomize_noise=False, return_latents=True)
    assert_close(img, g["wplus.image"], FWD_TOL, "wplus image")
    for n, s in enumerate(svec):
        assert_close(s, g[f"wplus.style.{n}"], 1e-5)
    r = seeded.tensor("gen16.r", img.shape)
    (gw,) = torch.autograd.grad((img * cu(r)).sum(), w)
    assert_grad_close(gw, g["wplus.grad_w"], "grad_w")
    z, z2 = cu(seeded.tensor("gen16.z", (2, 512))), cu(seeded.tensor("gen16.z2", (2, 512)))
    tl = cu(seeded.tensor("gen16.trunc", (1, 512), 0.3))
    with torch.no_grad():
        assert_close(gen([z], truncation=0.7, truncation_latent=tl, randomize_noise=False)[0], g["z.image"], FWD_TOL, "z")
        assert_close(gen([z, z2], inject_index=3, randomize_noise=False)[0], g["mix.image"], FWD_TOL, "mix")
        assert_close(gen([w[:, 0].detach()], input_is_latent=True, randomize_noise=False)[0], g["wsingle.image"], FWD_TOL)
        assert gen([z], randomize_noise=False)[1] is None
    sv = [cu(g[f"wplus.style.{n}"]).requires_grad_(True) for n in range(8)]
    img_s = gen([sv], input_is_stylespace=True, randomize_noise=False)[0]
    assert_close(img_s, g["sspace.image"], FWD_TOL, "sspace")
    gs = torch.autograd.grad((img_s * cu(r)).sum(), sv)
    for n, t in enumerate(gs):
        assert_grad_close(t, g[f"sspace.grad.{n}"], f"sspace grad {n}")
    # attention generator: features + blends
    ga = _gen(16, AttGenerator)
    with torch.no_grad():
        img_f, _, _, feats = ga([w.detach()], input_is_latent=True, randomize_noise=False, return_features=True)
    assert_close(img_f, g["att.image"], FWD_TOL)
    for n, f in enumerate(feats):
        assert_close(f, g[f"att.feat.{n}"], FWD_TOL, f"feat {n}")
    for layer in (4, 3, 5, 1):
        w2 = (w.detach() + 0.2 * cu(seeded.tensor("gen16.dw", w.shape))).requires_grad_(True)
        mask = cu(g[f"blend{layer}.mask"]).requires_grad_(True)
        img_b, _, _, nf = ga([w2], input_is_latent=True, randomize_noise=False, return_features=True,
                             attention_layer=layer, attention_map=mask, feature_map=feats)
        assert_close(img_b, g[f"blend{layer}.image"], FWD_TOL, f"blend {layer}")
        assert_close(nf[layer - 1], g[f"blend{layer}.feat_at"], FWD_TOL)
        gw2, gm = torch.autograd.grad((img_b * cu(r)).sum(), (w2, mask))
        assert_grad_close(gw2, g[f"blend{layer}.grad_w"], f"blend {layer} grad_w")
        assert_grad_close(gm, g[f"blend{layer}.grad_mask"], f"blend {layer} grad_mask")
    sv2 = [cu(g[f"wplus.style.{n}"]) * 1.1 for n in range(8)]
    with torch.no_grad():
        img_sb = ga([sv2], input_is_stylespace=True, randomize_noise=False, return_features=True, attention_layer=4,
                    attention_map=cu(g["sblend.mask"]), feature_map=feats)[0]
    assert_close(img_sb, g["sblend.image"], FWD_TOL)


@pytest.mark.parametrize("size", [256, 1024])
def test_generator_big_golden(size):
    """config 1 shape (256) and the FFHQ-1024 generator against values captured from the reference."""
    from where2edit_amd.attention_model import Generator as AttGenerator
    g = golden("generator_big")
    gen = _gen(size, AttGenerator)
    w = cu(seeded.wplus_latents(1, gen.n_latent, salt=size))
    with torch.no_grad():
        img, _, svec, feats = gen([w], input_is_latent=True, randomize_noise=False, return_features=True)
    st = size // 32
    assert_close(img[:, :, ::st, ::st], g[f"g{size}.image_strided"], 1e-3, "image (north_star tolerance)")
    assert abs(img.double().abs().sum().item() / float(g[f"g{size}.image_abs_sum"]) - 1) < 1e-4
    assert [s.shape[2] for s in svec] == list(g[f"g{size}.style_dims"])
    import numpy as np
    for n, f in enumerate(feats):
        pos = seeded.sample_positions(f.numel(), 32, f"g{size}.feat.{n}")
        ref = torch.from_numpy(np.asarray(g[f"g{size}.feat_samples"][n]))
        assert (f.reshape(-1)[cu(pos)].cpu() - ref).abs().max() <= 1e-3 * g[f"g{size}.feat_stats"][n][2], f"layer {n}"


def test_generator64_batch_vs_oracle_with_grad():
    size = 64
    gen = _gen(size)
    sd = seeded.generator_state_dict(size)
    w = seeded.wplus_latents(3, gen.n_latent, salt=3)
    r = seeded.tensor("g64.r", (3, 3, size, size))
    wo = w.clone().requires_grad_(True)
    io, _ = OG.generator_forward(sd, [wo], size=size, input_is_latent=True, randomize_noise=False)
    (go,) = torch.autograd.grad((io * r).sum(), wo)
    wg = cu(w).requires_grad_(True)
    ig, _ = gen([wg], input_is_latent=True, randomize_noise=False)
    assert_close(ig, io, FWD_TOL, "image")
    (gg,) = torch.autograd.grad((ig * cu(r)).sum(), wg)
    assert_grad_close(gg, go, "grad_w at 64^2")
    # random per-sample noise path (randomize_noise=True): same ops unfused, just has to run and differ
    with torch.no_grad():
        a = gen([cu(w)], input_is_latent=True, randomize_noise=True)[0]
    assert a.shape == ig.shape and torch.isfinite(a).all() and rel_err(a, ig.detach()) > 1e-4


def test_cpu_tensor_is_refused():
    from where2edit_amd.op import upfirdn2d
    with pytest.raises(RuntimeError, match="GPU only"):
        upfirdn2d(torch.randn(1, 1, 8, 8), torch.ones(2, 2))


def test_empty_batch_passes_through():
    """Batch 0 (e.g. a rank whose shard is empty) behaves like the torch reference: empty outputs, no launch."""
    from where2edit_amd.functional import clip_preprocess, mask_blend
    from where2edit_amd.op import fused_leaky_relu, upfirdn2d
    from where2edit_amd.stylegan2 import StyledConv, ToRGB
    k = cu(O.make_kernel((1, 3, 3, 1)))
    assert upfirdn2d(torch.empty(0, 3, 8, 8, device=DEV), k, up=2, pad=(2, 1)).shape == (0, 3, 16, 16)
    assert fused_leaky_relu(torch.empty(0, 4, device=DEV), torch.zeros(4, device=DEV)).shape == (0, 4)
    m = freeze_conv_weights(StyledConv(8, 8, 3, 512, upsample=True).to(DEV))
    y, s = m(torch.empty(0, 8, 4, 4, device=DEV), torch.empty(0, 512, device=DEV), noise=torch.zeros(1, 1, 8, 8, device=DEV))
    assert y.shape == (0, 8, 8, 8) and s.shape == (0, 1, 8, 1, 1)
    r = ToRGB(8, 512, upsample=False).to(DEV)
    assert r(torch.empty(0, 8, 4, 4, device=DEV), torch.empty(0, 512, device=DEV))[0].shape == (0, 3, 4, 4)
    assert clip_preprocess(torch.empty(0, 3, 64, 64, device=DEV)).shape == (0, 3, 224, 224)
    assert mask_blend(torch.empty(0, 2, 4, 4, device=DEV), torch.empty(0, 2, 4, 4, device=DEV),
                      torch.empty(0, 1, 2, 2, device=DEV)).shape == (0, 2, 4, 4)


def test_argument_errors_surface_as_exceptions():
    """Error convention of the boundary: bad arguments -> non-zero status -> RuntimeError with the library's message."""
    from where2edit_amd.functional import _upfirdn2d_raw, clip_preprocess
    with pytest.raises(RuntimeError, match="multiple of 32"):
        clip_preprocess(torch.zeros(1, 3, 48, 48, device=DEV))
    with pytest.raises(RuntimeError, match="unsupported"):
        _upfirdn2d_raw(torch.zeros(1, 1, 40, 40, device=DEV), torch.ones(17, 17, device=DEV), 24, 24, 1, 1, 0, 0, True)
    with pytest.raises(RuntimeError, match="fp32"):
        from where2edit_amd.op import upfirdn2d
        upfirdn2d(torch.zeros(1, 1, 8, 8, device=DEV, dtype=torch.float16), torch.ones(2, 2, device=DEV))


@pytest.mark.parametrize("size,key,gkey,b", [(1024, "preproc.id", "id1024.y", 2), (256, "preproc.id256", "id256.y", 1)])
def test_id_preprocess_golden_and_adjoint(size, key, gkey, b):
    """K5b: pool(256) -> crop -> pool(112) of criteria/id_loss.py:19-23 in one kernel, against the reference chain."""
    from where2edit_amd.functional import id_preprocess
    g = golden("preproc")
    img = seeded.tensor(key, (b, 3, size, size))
    ig = cu(img).requires_grad_(True)
    y = id_preprocess(ig)
    assert_close(y, g[gkey], 1e-5)
    gy = seeded.tensor("idpre.gy", y.shape)
    (gx,) = torch.autograd.grad(y, ig, cu(gy))
    io = img.clone().requires_grad_(True)
    (go,) = torch.autograd.grad(O.id_preprocess(io), io, gy)
    assert_close(gx, go, 1e-5, "adjoint")


@pytest.mark.parametrize("cin,cout,h,up", [(32, 32, 1024, False), (64, 32, 512, True), (512, 512, 64, False)])
def test_modconv_full_size_properties(cin, cout, h, up):
    """BASELINE-size layers (too big for the CPU oracle in a test): size-independent properties of the modulated conv.
    (1) demodulation makes the output invariant to a rescaling of the style; (2) the layer is linear in x;
    (3) the input-gradient kernel is the exact adjoint: <y(x), g> == <x, dx(g)>; (4) a strided sample of the
    output equals the oracle evaluated on the receptive-field crop."""
    from where2edit_amd import functional as K
    b = 2
    w = torch.randn(cout, cin, 3, 3, device=DEV)
    scale = (cin * 9) ** -0.5
    packs = (K.conv_pack(w, scale, False, False), K.conv_pack(w, scale, True, not up))
    wsq = (w * scale).square().sum((2, 3)).contiguous()
    blur = cu(seeded.fir_kernel(gain=4.0)) if up else None
    x = torch.randn(b, cin, h, h, device=DEV)
    s = torch.randn(b, cin, device=DEV) + 1.0
    y = K.modconv(x, s, wsq, packs, blur, up)
    oh = 2 * h if up else h
    assert y.shape == (b, cout, oh, oh)
    # (rounding-level properties: the F(4x4,3x3) form the 512-channel layer takes by default sits ~1e-5 from the exact result,
    # the direct kernels ~3e-7 -- DESIGN.md section 4, K1w)
    f4 = (not up) and K._wino_form(x, cin, cout, h, h, None) == 4
    y2 = K.modconv(x, 3.0 * s, wsq, packs, blur, up)                      # (1)
    assert rel_err(y2, y) < (1e-4 if f4 else 2e-5)
    y3 = K.modconv(-2.0 * x, s, wsq, packs, blur, up)                     # (2)
    assert rel_err(y3, -2.0 * y) < 1e-5
    xg = x.clone().requires_grad_(True)                                    # (3)
    g = torch.randn_like(y)
    (dx,) = torch.autograd.grad(K.modconv(xg, s, wsq, packs, blur, up), xg, g)
    lhs, rhs = (y.double() * g.double()).sum(), (x.double() * dx.double()).sum()
    assert abs(lhs - rhs) <= (5e-5 if f4 else 1e-5) * (y.double().abs() * g.double().abs()).sum()
    # (4) 8x8 output crop at an interior position against the oracle on the input crop that feeds it
    c0 = h // 2 - (h // 2) % 2
    halo = 4
    xc = x[:1, :, c0 - halo:c0 + 8 + halo, c0 - halo:c0 + 8 + halo].cpu()
    sc = s[:1].cpu().view(1, 1, cin, 1, 1)
    ref, _ = OG.modulated_conv2d(xc, sc, w.cpu()[None], None, None, demodulate=True, upsample=up,
                                 input_is_stylespace=True, blur_kernel=seeded.fir_kernel(gain=4.0) if up else None)
    if up:
        got = y[:1, :, 2 * c0:2 * c0 + 16, 2 * c0:2 * c0 + 16]
        ref = ref[:, :, 2 * halo:2 * halo + 16, 2 * halo:2 * halo + 16]
    else:
        got = y[:1, :, c0:c0 + 8, c0:c0 + 8]
        ref = ref[:, :, halo:halo + 8, halo:halo + 8]
    assert_close(got, ref, FWD_TOL, "crop vs oracle")


def test_batched_style_affines_match_per_layer_path():
    """w2e_style_affine_fwd/bwd (all 26 modulation EqualLinears in one launch) against the per-layer EqualLinear path:
    styles, image and the W+ gradient."""
    g = _gen(64)
    for p in g.parameters():
        p.requires_grad_(False)
    w = cu(seeded.wplus_latents(3, g.n_latent))

    def run(batched):
        wl = w.clone().requires_grad_(True)
        if not batched:
            g._batched_styles = lambda latent, plan: None
        try:
            img, _, styles = g([wl], input_is_latent=True, randomize_noise=False, return_latents=True)
        finally:
            if not batched:
                del g._batched_styles
        gy = cu(seeded.tensor("style.gy", tuple(img.shape)))
        (gw,) = torch.autograd.grad(img, wl, gy)
        return img, styles, gw

    img_a, st_a, gw_a = run(True)
    img_b, st_b, gw_b = run(False)
    assert len(st_a) == len(st_b)
    for a, b in zip(st_a, st_b):
        assert a.shape == b.shape
        assert_close(a, b, 1e-5, "style")
    assert_close(img_a, img_b, 1e-4, "image")
    assert_grad_close(gw_a, gw_b, "w+ grad")


@pytest.mark.parametrize("b,k,n,h,w", [(3, 20, 36, 20, 36), (2, 7, 5, 7, 19), (1, 64, 96, 33, 17), (5, 8, 8, 64, 4),
                                        (2, 96, 40, 48, 80), (1, 130, 70, 16, 24)])
def test_modconv_abi_non_square_and_ragged_channels(b, k, n, h, w):
    """w2e_modconv3x3 through the C ABI on shapes the generator never produces -- H != W, channel counts that are not
    multiples of 8 / 32, odd sizes, batch 1..5 -- in all three modes, with the fused activation and the dot epilogue,
    against torch's float64 convolutions of the same shared-weight formula."""
    import torch.nn.functional as F
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(1000 * k + n)
    wt = torch.randn(n, k, 3, 3, generator=g).to(DEV)
    scale = (k * 9) ** -0.5
    x = torch.randn(b, k, h, w, generator=g).to(DEV)
    s_in = (torch.randn(b, k, generator=g) * 0.3 + 1).to(DEV)
    s_out = (torch.rand(b, n, generator=g) + 0.5).to(DEV)
    wd, xd = wt.double() * scale, x.double() * s_in.double()[:, :, None, None]
    so = s_out.double()[:, :, None, None]
    # SAME, plain / fused activation / dot epilogue
    ref = F.conv2d(xd, wd, padding=1) * so
    fwd = K.conv_pack(wt, scale, False, False)
    y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w)
    assert_close(y, ref, FWD_TOL, "same")
    noise = torch.randn(1, 1, h, w, generator=g).to(DEV)
    nw, bias = torch.randn(1, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    pre = ref + nw.double() * noise.double() + bias.double()[None, :, None, None]
    y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, act=(noise, nw, bias))
    assert_close(y, F.leaky_relu(pre, 0.2) * 2 ** 0.5, FWD_TOL, "same + act")
    dw = torch.randn(b, n, h, w, generator=g).to(DEV)
    y, dot = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, dot_with=dw)
    # dot_out[b,o] = sum_p (unscaled accumulator) * dot_with  (the epilogue multiplies by out_scale only on the stored y)
    assert_close(y, ref, FWD_TOL, "same (dot epilogue) y")
    assert_close(dot, (ref / so * dw.double()).sum((2, 3)), 5e-4, "dot_out")
    # UP: conv_transpose2d stride 2 -> T (2h+1) x (2w+1), phase-planar
    wt_t = torch.randn(k, n, 3, 3, generator=g).to(DEV)  # conv_transpose weight layout [in, out, 3, 3]
    up = K.conv_pack(wt_t.permute(1, 0, 2, 3).contiguous(), scale, False, False)
    t, _ = K._modconv_raw(K.MODE_UP, x, up, s_in, s_out, h, w)
    ref_t = F.conv_transpose2d(xd, wt_t.double() * scale, stride=2) * so
    assert_close(K.unplanar(t, w), ref_t, FWD_TOL, "up")
    if w >= 16:  # the 4x4 blur reading the phase-planar T directly (+ fused noise/bias/lrelu), and its plain-layout adjoint
        k4 = cu(seeded.fir_kernel(gain=4.0))
        noise2 = torch.randn(1, 1, 2 * h, 2 * w, generator=g).to(DEV)
        blurred = O.upfirdn2d(ref_t.float().cpu(), seeded.fir_kernel(gain=4.0), pad=(1, 1)).double().to(DEV)
        pre2 = blurred + nw.double() * noise2.double() + bias.double()[None, :, None, None]
        got = K._upfirdn2d_raw(t, k4, 2 * h, 2 * w, 1, 1, 1, 1, True, act=(None, noise2, nw, bias), planar_hw=(2 * h + 1, 2 * w + 1))
        assert_close(got, F.leaky_relu(pre2, 0.2) * 2 ** 0.5, FWD_TOL, "planar blur + act")
        gq = torch.randn(b, n, 2 * h, 2 * w, generator=g).to(DEV)
        adj = K._upfirdn2d_raw(gq, k4, 2 * h + 1, 2 * w + 1, 1, 1, 2, 2, False)
        tq = torch.randn(b, n, 2 * h + 1, 2 * w + 1, generator=g).to(DEV)
        fwd_q = K._upfirdn2d_raw(tq, k4, 2 * h, 2 * w, 1, 1, 1, 1, True)
        lhs, rhs = (fwd_q.double() * gq.double()).sum(), (tq.double() * adj.double()).sum()
        assert abs(lhs - rhs) <= 1e-5 * (fwd_q.double().abs() * gq.double().abs()).sum(), "blur adjoint"
    # DOWN: stride-2 conv of a (2h+1) x (2w+1) input
    xb = torch.randn(b, k, 2 * h + 1, 2 * w + 1, generator=g).to(DEV)
    y, _ = K._modconv_raw(K.MODE_DOWN, xb, fwd, s_in, s_out, h, w)
    assert_close(y, F.conv2d(xb.double() * s_in.double()[:, :, None, None], wd, stride=2) * so, FWD_TOL, "down")


@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("b,k,n,h,w", [(2, 20, 40, 37, 53), (1, 8, 8, 70, 33), (2, 33, 130, 16, 100)])
def test_modconv_lds_dma_pipeline_on_ragged_shapes(cfg, b, k, n, h, w, w2e_opt):
    """The LDS-DMA K-loop pipeline (the 512-thread tiles; picked by the cost model only for the large layers) forced onto
    shapes with ragged channel counts, odd sizes and partial tiles: the zero padding there is entirely the buffer range
    check of `buffer_load ... lds` (halo pixels, channels >= K, weight groups past K).  SAME plain / fused activation / dot
    epilogue and the all-phase UP form, against float64 convolutions."""
    import torch.nn.functional as F
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(77 * k + n + cfg)
    wt = torch.randn(n, k, 3, 3, generator=g).to(DEV)
    scale = (k * 9) ** -0.5
    x = torch.randn(b, k, h, w, generator=g).to(DEV)
    s_in = (torch.randn(b, k, generator=g) * 0.3 + 1).to(DEV)
    s_out = (torch.rand(b, n, generator=g) + 0.5).to(DEV)
    wd, xd = wt.double() * scale, x.double() * s_in.double()[:, :, None, None]
    so = s_out.double()[:, :, None, None]
    ref = F.conv2d(xd, wd, padding=1) * so
    fwd = K.conv_pack(wt, scale, False, False)
    w2e_opt("tune_cfg", f"{cfg},1,0")  # SAME launches only (w2e_set_option)
    y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w)
    assert_close(y, ref, FWD_TOL, "same")
    y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, None, None, h, w)  # no modulation: the in_scale table is all ones
    assert_close(y, F.conv2d(x.double(), wd, padding=1), FWD_TOL, "same, unmodulated")
    noise = torch.randn(1, 1, h, w, generator=g).to(DEV)
    nw, bias = torch.randn(1, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    pre = ref + nw.double() * noise.double() + bias.double()[None, :, None, None]
    y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, act=(noise, nw, bias))
    assert_close(y, F.leaky_relu(pre, 0.2) * 2 ** 0.5, FWD_TOL, "same + act")
    dw = torch.randn(b, n, h, w, generator=g).to(DEV)
    y, dot = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, dot_with=dw)
    assert_close(y, ref, FWD_TOL, "same (dot epilogue) y")
    assert_close(dot, (ref / so * dw.double()).sum((2, 3)), 5e-4, "dot_out")
    if cfg <= 1:  # all-phase UP tiles 0 / 1 (tile 1 takes the pipeline from K >= 256 on: covered by the full-size tests)
        w2e_opt("tune_cfg", f"{cfg},1,1")
        wt_t = torch.randn(k, n, 3, 3, generator=g).to(DEV)
        up = K.conv_pack(wt_t.permute(1, 0, 2, 3).contiguous(), scale, False, False)
        t, _ = K._modconv_raw(K.MODE_UP, x, up, s_in, s_out, h, w)
        assert_close(K.unplanar(t, w), F.conv_transpose2d(xd, wt_t.double() * scale, stride=2) * so, FWD_TOL, "up")


@pytest.mark.parametrize("cin,h,with_skip", [(32, 64, True), (12, 18, False), (512, 8, True)])
def test_to_rgb_passthrough_joins_gradients(cin, h, with_skip):
    """to_rgb(..., passthrough=True) hands x on to the next layer through the ToRGB node; the gradient that comes back
    through that second output is folded into the ToRGB input gradient by w2e_torgb_bwd_acc.  Same values and gradients
    as the plain form, where autograd adds the two."""
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(5 * cin + h)
    b = 2
    k4 = cu(seeded.fir_kernel(gain=4.0))
    data = [torch.randn(b, cin, h, h, generator=g), torch.randn(b, 3, cin, generator=g) * 0.2, torch.randn(1, 3, 1, 1, generator=g)]
    skip0 = torch.randn(b, 3, h // 2, h // 2, generator=g) if with_skip else None
    r, q = torch.randn(b, 3, h, h, generator=g).to(DEV), torch.randn(b, cin, h, h, generator=g).to(DEV)

    def run(passthrough):
        x, wmod, bias = [t.to(DEV).requires_grad_(True) for t in data]
        skip = skip0.to(DEV).requires_grad_(True) if with_skip else None
        if passthrough:
            y, xp = K.to_rgb(x, wmod, bias, skip, k4 if with_skip else None, True)
            assert xp.data_ptr() == x.data_ptr()
        else:
            y, xp = K.to_rgb(x, wmod, bias, skip, k4 if with_skip else None), x
        loss = (y * r).sum() + (torch.tanh(xp) * q).sum()
        grads = torch.autograd.grad(loss, [x, wmod, bias] + ([skip] if with_skip else []))
        return y.detach(), grads

    y_a, g_a = run(True)
    y_b, g_b = run(False)
    assert torch.equal(y_a, y_b)
    for a, bb, name in zip(g_a, g_b, ["gx", "gwmod", "gbias", "gskip"]):
        assert_close(a, bb, 1e-6, name)


def _to_planar(t, fill):
    """[N,C,2h+1,2w+1] -> the UP conv's phase-planar [N,C,2,2,h+1,WP] (WP = W2E_PLANAR_PITCH(w), include/w2e.h), padding = fill."""
    from where2edit_amd import functional as K
    n, c, ih, iw = t.shape
    hp, wp = (ih + 1) // 2, K.planar_pitch((iw - 1) // 2)
    out = torch.full((n, c, 2, 2, hp, wp), fill, device=t.device, dtype=t.dtype)
    for py in range(2):
        for px in range(2):
            sub = t[:, :, py::2, px::2]
            out[:, :, py, px, :sub.shape[2], :sub.shape[3]] = sub
    return out


@pytest.mark.parametrize("h,w,separable", [(128, 128, True), (130, 256, True), (96, 128, False), (33, 512, True)])
@pytest.mark.parametrize("act", [False, True])
def test_upfirdn_stream_planar_equals_reference_form(h, w, separable, act, w2e_opt):
    """The streaming 4x4 kernel (wide images) on the UP conv's phase-planar T': same result as the dense generic kernel on the
    plain image, strips that end mid-way, a non-separable kernel, NaN in the layout's padding (never read)."""
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(3 * h + w)
    c = 3
    t = torch.randn(2, c, 2 * h + 1, 2 * w + 1, generator=g).to(DEV)
    k4 = cu(seeded.fir_kernel(gain=4.0))
    if not separable:
        k4 = k4 + 0.05 * torch.randn(4, 4, generator=g).to(DEV)
    a = (None, torch.randn(1, 1, 2 * h, 2 * w, generator=g).to(DEV), torch.randn(1, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)) if act else None
    y = K._upfirdn2d_raw(_to_planar(t, float("nan")), k4, 2 * h, 2 * w, 1, 1, 1, 1, True, act=a, planar_hw=(2 * h + 1, 2 * w + 1))
    w2e_opt("tune_blur", "8")  # the tile kernels
    ref = K._upfirdn2d_raw(t, k4, 2 * h, 2 * w, 1, 1, 1, 1, True, act=a)
    assert torch.isfinite(y).all()
    assert_close(y, ref, 2e-6, "stream planar")


@pytest.mark.parametrize("h,w,separable", [(128, 128, True), (100, 256, True), (64, 130, False), (40, 514, True), (30, 131, True)])
def test_upfirdn_stream_dense_adjoint_equals_tile_kernel(h, w, separable, w2e_opt):
    """The streaming kernel on a dense source (the adjoint blur: pad 2, un-flipped taps, output one wider and taller than the
    input): odd output widths (the single last column), ragged widths, strips that end after one row."""
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(5 * h + w)
    gy = torch.randn(2, 3, 2 * h, 2 * w, generator=g).to(DEV)
    k4 = cu(seeded.fir_kernel(gain=1.0))
    if not separable:
        k4 = k4 + 0.05 * torch.randn(4, 4, generator=g).to(DEV)
    for oh, ow in ((2 * h + 1, 2 * w + 1), (2 * h, 2 * w - 1), (2 * h - 3, 2 * w - 2)):
        w2e_opt("tune_blur", "0")
        y = K._upfirdn2d_raw(gy, k4, oh, ow, 1, 1, 2, 2, False)
        w2e_opt("tune_blur", "8")
        ref = K._upfirdn2d_raw(gy, k4, oh, ow, 1, 1, 2, 2, False)
        assert_close(y, ref, 2e-6, f"stream dense {oh}x{ow}")


@pytest.mark.parametrize("cin,h,styled,with_acc,with_noise", [(32, 64, True, True, True), (12, 18, False, False, True), (512, 8, True, True, False),
                                                               (64, 128, False, True, True)])
def test_torgb_backward_with_fused_activation_backward(cin, h, styled, with_acc, with_noise):
    """w2e_torgb_bwd_actbwd: the ToRGB input gradient with the producing StyledConv's activation backward applied and that
    layer's three sums, against w2e_torgb_bwd_acc / w2e_torgb_styled_bwd followed by w2e_bias_act_bwd_reduce."""
    from where2edit_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(13 * cin + h)
    b = 3
    x = torch.randn(b, cin, h, h, generator=g).to(DEV)
    gy = torch.randn(b, 3, h, h, generator=g).to(DEV)
    acc = torch.randn(b, cin, h, h, generator=g).to(DEV) if with_acc else None
    noise = torch.randn(h * h, generator=g).to(DEV) if with_noise else None
    style = torch.randn(b, cin, generator=g).to(DEV) if styled else None
    wmod = (torch.randn(3, cin, generator=g) * 0.2).to(DEV) if styled else (torch.randn(b, 3, cin, generator=g) * 0.2).to(DEV)
    gx, gw_ref = torch.empty_like(x), torch.empty((b, cin) if styled else (b, 3, cin), device=DEV)
    if styled:
        call("w2e_torgb_styled_bwd", ptr(x), ptr(wmod), ptr(style), ptr(gy), ptr(acc), ptr(gx), ptr(gw_ref), b, cin, h, h, stream_ptr())
    else:
        call("w2e_torgb_bwd_acc", ptr(x), ptr(wmod), ptr(gy), ptr(acc), ptr(gx), ptr(gw_ref), b, cin, h, h, stream_ptr())
    gpre_ref, sums_ref = torch.empty_like(x), torch.empty(b, cin, 3, device=DEV)
    call("w2e_bias_act_bwd_reduce", ptr(gx), ptr(x), ptr(noise), ptr(gpre_ref), ptr(sums_ref), b, cin, h * h, 0.2, 2 ** 0.5, stream_ptr())
    gpre, gw, sums = torch.full_like(x, float("nan")), torch.full_like(gw_ref, float("nan")), torch.full((b, cin, 3), float("nan"), device=DEV)
    call("w2e_torgb_bwd_actbwd", ptr(x), ptr(wmod), ptr(style), ptr(gy), ptr(acc), ptr(noise), ptr(gpre), ptr(gw), ptr(sums), b, cin, h, h,
         0.2, 2 ** 0.5, stream_ptr())
    assert_close(gpre, gpre_ref, 2e-6, "pre-activation gradient")
    assert_close(gw, gw_ref, 1e-5, "weight / style gradient")
    scale = sums_ref.abs().max()
    assert float((sums - sums_ref).abs().max()) <= 2e-5 * float(scale), "sums"


@pytest.mark.parametrize("h,w,with_noise", [(128, 128, True), (130, 256, False), (65, 512, True)])
def test_blur_adjoint_with_fused_activation_backward(h, w, with_noise, w2e_opt):
    """w2e_blur_adjoint_actbwd (the StyledConv backward of a wide up-sampling layer in one pass): the same adjoint-blurred
    pre-activation gradient and the same three per-plane sums as w2e_bias_act_bwd_reduce followed by the adjoint blur."""
    from where2edit_amd import functional as K
    from where2edit_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(9 * h + w)
    b, c = 2, 5
    gy, y = torch.randn(b, c, 2 * h, 2 * w, generator=g).to(DEV), torch.randn(b, c, 2 * h, 2 * w, generator=g).to(DEV)
    noise = torch.randn(1, 1, 2 * h, 2 * w, generator=g).to(DEV) if with_noise else None
    k4 = cu(seeded.fir_kernel(gain=1.0))
    gpre, sums_ref = torch.empty_like(gy), torch.empty(b, c, 3, device=DEV)
    call("w2e_bias_act_bwd_reduce", ptr(gy), ptr(y), ptr(noise), ptr(gpre), ptr(sums_ref), b, c, 4 * h * w, 0.2, 2 ** 0.5, stream_ptr())
    gt_ref = K._upfirdn2d_raw(gpre, k4, 2 * h + 1, 2 * w + 1, 1, 1, 2, 2, False)
    gt, sums = torch.full((b, c, 2 * h + 1, 2 * w + 1), float("nan"), device=DEV), torch.full((b, c, 3), float("nan"), device=DEV)
    call("w2e_blur_adjoint_actbwd", ptr(gy), ptr(y), ptr(noise), ptr(k4), ptr(gt), ptr(sums), b * c, 2 * h, 2 * w, 0.2, 2 ** 0.5, stream_ptr())
    assert_close(gt, gt_ref, 2e-6, "adjoint-blurred pre-activation gradient")
    for i, name in enumerate(("sum gpre*pre", "sum gpre*noise", "sum gpre")):
        assert_close(sums[..., i] + 1.0, sums_ref[..., i] + 1.0, 2e-5, name)  # (+1: a sum near 0 is compared absolutely)


@pytest.mark.parametrize("cin,h,with_skip,prefix", [(32, 64, True, 0), (12, 18, False, 0), (512, 8, True, 1), (64, 128, True, 2)])
def test_to_rgb_styled_equals_per_sample_weight(cin, h, with_skip, prefix):
    """w2e_torgb_styled_*: the kernels form scale*W[c,i]*style[b,i] themselves and return the style gradient; same image and
    the same gradients as the [B,3,cin] per-sample weight built by an elementwise product (model.py:239 with k = 1), also
    with the rows of a no-grad prefix skipped (functional.nograd_prefix: their gradient rows are zero / never read)."""
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(7 * cin + h)
    b = 4
    k4 = cu(seeded.fir_kernel(gain=4.0))
    x0, wsc, st0 = torch.randn(b, cin, h, h, generator=g), (torch.randn(3, cin, generator=g) * 0.2).to(DEV), torch.randn(b, cin, generator=g)
    bias = torch.randn(1, 3, 1, 1, generator=g).to(DEV)
    skip0 = torch.randn(b, 3, h // 2, h // 2, generator=g) if with_skip else None
    r = torch.randn(b, 3, h, h, generator=g).to(DEV)

    def run(styled):
        x, st = x0.to(DEV).requires_grad_(True), st0.to(DEV).requires_grad_(True)
        skip = skip0.to(DEV).requires_grad_(True) if with_skip else None
        with K.nograd_prefix(prefix):
            if styled:
                y = K.to_rgb(x, wsc, bias, skip, k4 if with_skip else None, style=st)
            else:
                y = K.to_rgb(x, wsc.view(1, 3, cin) * st.view(b, 1, cin), bias, skip, k4 if with_skip else None)
        grads = torch.autograd.grad((y[prefix:] * r[prefix:]).sum(), [x, st] + ([skip] if with_skip else []))
        return y.detach(), [t[prefix:] for t in grads], grads[1][:prefix]

    y_a, g_a, head = run(True)
    y_b, g_b, _ = run(False)
    assert torch.equal(y_a, y_b)
    assert not head.any()
    for a, bb, name in zip(g_a, g_b, ["gx", "gstyle", "gskip"]):
        assert_close(a, bb, 2e-6, name)


@pytest.mark.parametrize("b,k,n,h,w,cfg", [(2, 20, 40, 37, 53, 0), (1, 8, 8, 70, 33, 1), (2, 33, 130, 16, 100, 2), (1, 24, 32, 40, 64, 8),
                                            (1, 256, 256, 64, 64, 0)])
def test_modconv_bf16x3_split_precision(b, k, n, h, w, cfg, w2e_opt):
    """Opt-in W2E_CONV_PRECISION=bf16x3: the SAME-mode tiles of the DMA pipeline compute each fp32 product as three bf16
    products (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, two taps per MFMA).  Same bar as the exact path:
    FWD_TOL = 1e-4 relative against float64 convolutions (north_star: 1e-3), plain / fused activation / dot epilogue,
    ragged channel counts and partial tiles included."""
    import torch.nn.functional as F
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(31 * k + n + cfg)
    wt = torch.randn(n, k, 3, 3, generator=g).to(DEV)
    scale = (k * 9) ** -0.5
    x = torch.randn(b, k, h, w, generator=g).to(DEV)
    s_in = (torch.randn(b, k, generator=g) * 0.3 + 1).to(DEV)
    s_out = (torch.rand(b, n, generator=g) + 0.5).to(DEV)
    wd, xd = wt.double() * scale, x.double() * s_in.double()[:, :, None, None]
    so = s_out.double()[:, :, None, None]
    ref = F.conv2d(xd, wd, padding=1) * so
    fwd = K.conv_pack(wt, scale, False, False)
    y_exact, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w)
    w2e_opt("tune_cfg", f"{cfg},1,0")
    w2e_opt("conv_precision", "bf16x3")
    y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w)
    assert_close(y, ref, FWD_TOL, "same")
    assert not torch.equal(y, y_exact), "the split path did not run"
    noise = torch.randn(1, 1, h, w, generator=g).to(DEV)
    nw, bias = torch.randn(1, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    pre = ref + nw.double() * noise.double() + bias.double()[None, :, None, None]
    y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, act=(noise, nw, bias))
    assert_close(y, F.leaky_relu(pre, 0.2) * 2 ** 0.5, FWD_TOL, "same + act")
    dw = torch.randn(b, n, h, w, generator=g).to(DEV)
    y, dot = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, dot_with=dw)
    assert_close(y, ref, FWD_TOL, "same (dot epilogue) y")
    assert_close(dot, (ref / so * dw.double()).sum((2, 3)), 5e-4, "dot_out")
    if cfg <= 2 and h * w >= 256:  # the all-phase UP tiles (two taps of the same output phase per MFMA)
        w2e_opt("tune_cfg", f"{cfg},1,1")
        wt_t = torch.randn(k, n, 3, 3, generator=g).to(DEV)
        up = K.conv_pack(wt_t.permute(1, 0, 2, 3).contiguous(), scale, False, False)
        ref_t = F.conv_transpose2d(xd, wt_t.double() * scale, stride=2) * so
        w2e_opt("conv_precision", "f32")
        t_exact, _ = K._modconv_raw(K.MODE_UP, x, up, s_in, s_out, h, w)
        w2e_opt("conv_precision", "bf16x3")
        t, _ = K._modconv_raw(K.MODE_UP, x, up, s_in, s_out, h, w)
        assert_close(K.unplanar(t, w), ref_t, FWD_TOL, "up")
        assert not torch.equal(t, t_exact), "the split path did not run (up)"


@pytest.mark.parametrize("b,k,n,h,w", [(2, 24, 136, 64, 64), (1, 33, 130, 70, 96)])
def test_modconv_bf16x3_down(b, k, n, h, w, w2e_opt):
    """bf16x3 on the DOWN tile (stride-2 conv = the input gradient of the up-sampling layers; taken from 64x64 outputs and 128 output channels up)."""
    import torch.nn.functional as F
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(13 * k + n)
    wt = torch.randn(n, k, 3, 3, generator=g).to(DEV)
    scale = (k * 9) ** -0.5
    xb = torch.randn(b, k, 2 * h + 1, 2 * w + 1, generator=g).to(DEV)
    s_in = (torch.randn(b, k, generator=g) * 0.3 + 1).to(DEV)
    s_out = (torch.rand(b, n, generator=g) + 0.5).to(DEV)
    fwd = K.conv_pack(wt, scale, False, False)
    ref = F.conv2d(xb.double() * s_in.double()[:, :, None, None], wt.double() * scale, stride=2) * s_out.double()[:, :, None, None]
    w2e_opt("conv_precision", "f32")
    y_exact, _ = K._modconv_raw(K.MODE_DOWN, xb, fwd, s_in, s_out, h, w)
    w2e_opt("conv_precision", "bf16x3")
    y, _ = K._modconv_raw(K.MODE_DOWN, xb, fwd, s_in, s_out, h, w)
    assert_close(y, ref, FWD_TOL, "down")
    assert not torch.equal(y, y_exact), "the split path did not run (down)"
    dw = torch.randn(b, n, h, w, generator=g).to(DEV)
    y, dot = K._modconv_raw(K.MODE_DOWN, xb, fwd, s_in, s_out, h, w, dot_with=dw)
    assert_close(y, ref, FWD_TOL, "down (dot epilogue) y")
    assert_close(dot, (ref / s_out.double()[:, :, None, None] * dw.double()).sum((2, 3)), 5e-4, "dot_out")


def test_generator1024_golden_with_bf16x3(w2e_opt):
    """The FFHQ-1024 generator against the values captured from the reference with the opt-in bf16x3 conv tiles: the
    north_star tolerance (1e-3 relative) holds through the 17 stacked layers."""
    from where2edit_amd.attention_model import Generator as AttGenerator
    w2e_opt("conv_precision", "bf16x3")
    size = 1024
    g = golden("generator_big")
    gen = _gen(size, AttGenerator)
    w = cu(seeded.wplus_latents(1, gen.n_latent, salt=size))
    with torch.no_grad():
        img, _, svec, feats = gen([w], input_is_latent=True, randomize_noise=False, return_features=True)
    st = size // 32
    assert_close(img[:, :, ::st, ::st], g[f"g{size}.image_strided"], 1e-3, "image (north_star tolerance)")
    assert abs(img.double().abs().sum().item() / float(g[f"g{size}.image_abs_sum"]) - 1) < 1e-4


@pytest.mark.parametrize("mode", ["same", "same_act", "same_dot", "same_split", "up", "down"])
def test_modconv_ragged_channels_never_write_past_the_output(mode, w2e_opt):
    """Guard band: the epilogue relies on the buffer range check to drop channels >= N when the tile is wider than
    N (N = 40 on 64- / 128-channel tiles) and pixels outside the image.  The output lives at the head of a larger allocation
    filled with a canary: nothing behind it (the next image's planes, in a batched tensor) may change, and the head must equal
    the separately allocated result -- SAME (plain / fused activation / dot epilogue / split-K atomics), UP and DOWN."""
    from where2edit_amd import functional as K
    from where2edit_amd._lib import call, ptr, stream_ptr
    b, k, n, h, w = 2, 24, 40, 19, 27
    g = torch.Generator().manual_seed(11)
    wt = torch.randn(n, k, 3, 3, generator=g).to(DEV)
    s_in, s_out = (torch.rand(b, k, generator=g) + 0.5).to(DEV), (torch.rand(b, n, generator=g) + 0.5).to(DEV)
    pack = K.conv_pack(wt, (k * 9) ** -0.5, False, False)
    kmode = {"up": K.MODE_UP, "down": K.MODE_DOWN}.get(mode, K.MODE_SAME)
    x = torch.randn(b, k, 2 * h + 1, 2 * w + 1, generator=g).to(DEV) if mode == "down" else torch.randn(b, k, h, w, generator=g).to(DEV)
    noise, nw, bias = torch.randn(1, 1, h, w, generator=g).to(DEV), torch.randn(1, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    dw = torch.randn(b, n, h, w, generator=g).to(DEV)
    act = (noise, nw, bias) if mode == "same_act" else None
    dot_with = dw if mode == "same_dot" else None
    if mode == "same_split":
        w2e_opt("tune_cfg", "6,4,0")  # 64x64 tile, 4 K-slices: the atomic epilogue
    ref, _ = K._modconv_raw(kmode, x, pack, s_in, s_out, h, w, act=act, dot_with=dot_with)
    numel = ref.numel()
    tail = 1 << 16
    buf = torch.full((numel + tail,), 1234.5, device=DEV)
    dot = torch.zeros(b, n, device=DEV) if dot_with is not None else None
    call("w2e_modconv3x3", kmode, ptr(x), ptr(pack), ptr(s_in), ptr(s_out), ptr(buf), b, k, n, h, w,
         K.planar_pitch(w) if mode == "up" else 0, int(act is not None),
         ptr(noise) if act else None, ptr(nw) if act else None, ptr(bias) if act else None, ptr(dot_with), ptr(dot), stream_ptr())
    torch.cuda.synchronize()
    assert torch.all(buf[numel:] == 1234.5), f"{mode}: wrote past the end of the output"
    if mode == "up":  # the pad columns of the phase-planar rows (W+1 .. WP-1) are never written: compare the image itself
        assert_close(K.unplanar(buf[:numel].view_as(ref), w), K.unplanar(ref, w), 1e-6, mode)
    else:
        assert_close(buf[:numel].view_as(ref), ref, 2e-6 if mode == "same_split" else 1e-6, mode)


def test_demod_coefficients_of_all_layers_in_one_launch():
    """w2e_demod_all_fwd (every demodulated layer of a generator pass in one launch) against rsqrt(s^2 @ wsq^T + eps)
    (model.py:241-243) in float64, on ragged layer shapes, and bit-equal to the per-layer kernel."""
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(17)
    b = 3
    shapes = [(512, 512), (512, 256), (72, 40), (32, 32), (64, 3)]  # (cin, cout)
    styles = [(torch.randn(b, cin, generator=g) + 1.0).to(DEV) for cin, _ in shapes]
    wsqs = [(torch.rand(cout, cin, generator=g) / cin).to(DEV) for cin, cout in shapes]
    ds = K.demod_coefficients_all(styles, wsqs)
    for s, w, d in zip(styles, wsqs, ds):
        ref = torch.rsqrt(s.double().cpu().square() @ w.double().cpu().t() + 1e-8)
        assert_close(d, ref, 1e-5, f"demod {tuple(w.shape)}")
        assert torch.equal(d, K.demod_coefficients(s, w))


def test_grad_pool_hands_out_disjoint_zeroed_regions():
    """functional.GradPool: regions are zero, aligned, disjoint, and a request past the chunk starts a new zeroed chunk."""
    from where2edit_amd import functional as K
    pool = K.GradPool(1000)
    a = pool.zeros((3, 7), DEV)
    b_ = pool.zeros((5, 64), DEV)
    c = pool.zeros((2000,), DEV)  # does not fit the rest of the first chunk
    d = pool.zeros((4, 4), DEV)
    for t in (a, b_, c, d):
        assert t.is_contiguous() and t.data_ptr() % 256 == 0 and float(t.abs().sum()) == 0.0
    a.fill_(1.0), b_.fill_(2.0), c.fill_(3.0), d.fill_(4.0)
    assert float(a.sum()) == 21.0 and float(b_.sum()) == 640.0 and float(c.sum()) == 6000.0 and float(d.sum()) == 64.0
    with K.grad_pool(16) as p:
        assert K._GRAD_POOL is p
    assert K._GRAD_POOL is None


def test_act_link_refuses_a_hooked_activation():
    """functional.ActLink: once the ToRGB backward has applied the producing StyledConv's activation backward to the gradient
    it returns, that gradient must reach the StyledConv backward as the very same tensor.  A tensor hook on the activation
    replaces it on the way: the StyledConv backward must raise instead of applying the activation backward a second time."""
    from where2edit_amd.stylegan2 import StyledConv, ToRGB
    cin, cout, h, b = 16, 16, 16, 2
    sc = freeze_conv_weights(StyledConv(cin, cout, 3, 512).to(DEV))
    rgb = freeze_conv_weights(ToRGB(cout, 512, upsample=False).to(DEV))
    x = cu(seeded.tensor("al.x", (b, cin, h, h))).requires_grad_(True)
    w = cu(seeded.tensor("al.w", (b, 512)))
    noise = cu(seeded.tensor("al.n", (1, 1, h, h)))

    def run(hook):
        y, _ = sc(x, w, noise=noise)
        assert sc._act_noise is not None
        if hook:
            y.register_hook(lambda g: g * 1.0)
        img, _ = rgb(y, w, producer_act=sc._act_noise)
        return torch.autograd.grad(img.sum(), x)[0]

    g_ok = run(False)
    assert torch.isfinite(g_ok).all()
    with pytest.raises(RuntimeError, match="ActLink"):
        run(True)


def test_trainable_decoder_weights_are_refused_not_run_on_a_second_backend():
    """Decoder fine-tuning is off this path (coach.py:174-180 optimises net.mapper only): a conv weight that requires grad is refused
    with an error that names the fix -- rounds 2-3 ran such a layer on stock MIOpen ops instead, a second backend inside
    ModulatedConv2d -- while the same generator with frozen conv weights runs, and still differentiates to its noise strengths / biases."""
    from where2edit_amd.stylegan2 import Generator
    size = 16
    sd = seeded.generator_state_dict(size)
    w = seeded.wplus_latents(2, OG.n_latent(size), salt=9)
    gen = Generator(size, 512, 8)
    gen.load_state_dict(sd, strict=True)
    gen = gen.to(DEV)  # (weights stay trainable: no freeze_conv_weights)
    with pytest.raises(RuntimeError, match="freeze_conv_weights"):
        gen([cu(w)], input_is_latent=True, randomize_noise=False)
    with torch.no_grad():  # (no gradient asked for: the kernels run)
        ig, _ = gen([cu(w)], input_is_latent=True, randomize_noise=False)
    io, _ = OG.generator_forward(sd, [w], size=size, input_is_latent=True, randomize_noise=False)
    assert_close(ig, io, FWD_TOL, "image under no_grad with trainable weights")
    freeze_conv_weights(gen)
    ig2, _ = gen([cu(w)], input_is_latent=True, randomize_noise=False)
    assert ig2.requires_grad  # (biases / noise strengths / affines are still trainable)
    assert_close(ig2, io, FWD_TOL, "image with frozen conv weights")


def test_wino_gemm_edge_shapes_and_argument_errors():
    """w2e_wino_gemm at the edges of its contract: the smallest legal layer (K = 8: one chunk, N = 64, one 4 x 4 image = one tile in a
    32-tile block), an empty batch, a K split that equals the chunk count, and the argument errors a caller can provoke (N % 64, a split
    count that leaves an empty slice, the fused dot on a tile count it cannot segment, noise on a ragged image) -- refused with a message,
    never launched."""
    import torch.nn.functional as F
    from where2edit_amd import functional as K
    from where2edit_amd._lib import call, ptr, stream_ptr
    g = torch.Generator().manual_seed(5)
    for (b, k, n, h, w, sp) in ((1, 8, 64, 4, 4, 0), (2, 32, 64, 8, 4, 4), (0, 64, 64, 16, 16, 0)):
        wt = torch.randn(n, k, 3, 3, generator=g).to(DEV)
        x = torch.randn(b, k, h, w, generator=g).to(DEV)
        pack = K.conv_pack(wt, 1.0, False, False)
        y = torch.full((b, n, h, w), 3.0, device=DEV)
        saved = K.GEMM_SPLITS
        K.GEMM_SPLITS = sp
        try:
            if b:
                K.wino_gemm_conv(x, pack, None, None, y, k, n, h, w)
        finally:
            K.GEMM_SPLITS = saved
        if b:
            assert_close(y, F.conv2d(x.double(), wt.double(), padding=1), FWD_TOL, f"wino gemm {b}x{k}->{n}@{h}x{w}")
    wt = torch.randn(64, 64, 3, 3, generator=g).to(DEV)
    x = torch.randn(2, 64, 16, 16, generator=g).to(DEV)
    uf = K._wino_weights_fused(K.conv_pack(wt, 1.0, False, False), 64, 64)
    vf, y, ws = torch.empty(36 * 64 * 32, device=DEV), torch.empty(2, 64, 16, 16, device=DEV), torch.empty(1 << 20, device=DEV)
    ok = ("w2e_wino_gemm", ptr(uf), ptr(vf), None, ptr(y), 2, 64, 64, 16, 16, 32, 1, ptr(ws), 0, None, None, None, None, None, None, stream_ptr())
    call("w2e_wino_pack_input", ptr(x), None, ptr(vf), 2, 64, 16, 16, 32, stream_ptr())
    call(*ok)
    for index, value in ((7, 48), (11, 9), (10, 5)):  # n_ch = 48 (N % 64), 9 splits of 8 chunks, tiles_padded = 5 for 32 tiles
        args = list(ok)
        args[index] = value
        with pytest.raises(RuntimeError, match="wino_gemm"):
            call(*args)
    dw = torch.randn(2, 64, 12, 12, generator=g).to(DEV)  # 9 tiles per plane: neither a multiple of 32 nor a power of two
    with pytest.raises(RuntimeError, match="fused dot"):
        call("w2e_wino_gemm", ptr(uf), ptr(vf), None, ptr(y), 2, 64, 64, 12, 12, 32, 1, ptr(ws), 0, None, None, None, None, ptr(dw), ptr(ws), stream_ptr())
    with pytest.raises(RuntimeError, match="not multiples of 4"):
        call("w2e_wino_gemm", ptr(uf), ptr(vf), None, ptr(y), 2, 64, 64, 14, 14, 32, 1, ptr(ws), 1, ptr(dw), ptr(ws), None, None, None, None, stream_ptr())


@pytest.mark.parametrize("m,b,k,n,h,w", [(4, 3, 64, 64, 16, 16), (4, 1, 64, 128, 32, 32), (4, 2, 512, 512, 32, 32), (4, 5, 128, 64, 64, 32),
                                         (4, 2, 128, 128, 16, 32), (4, 1, 192, 64, 8, 16),
                                         (11, 2, 512, 512, 16, 16), (11, 3, 64, 64, 16, 16), (11, 1, 128, 64, 32, 64), (12, 2, 256, 128, 32, 32),
                                         (8, 2, 32, 32, 32, 64), (8, 1, 64, 64, 16, 32), (8, 3, 128, 32, 48, 32), (8, 2, 64, 32, 32, 32), (8, 5, 32, 64, 64, 64),
                                         (8, 2, 32, 128, 32, 32), (8, 1, 32, 32, 64, 96), (8, 1, 256, 32, 16, 64), (8, 2, 256, 64, 32, 32)])
def test_modconv_winograd_form(m, b, k, n, h, w):
    """K1w / K1g: the Winograd F(4x4,3x3) forms of the same-resolution conv against float64 convolutions, every epilogue of w2e_modconv3x3
    -- plain, unmodulated, noise + bias + LeakyReLU, and the input-gradient pass (transposed + flipped pack) with the fused per-channel
    dot -- and against the direct kernel.  m = 4: the GEMM form on the OWN contraction kernel (w2e_wino_pack_input + w2e_wino_gemm:
    output transform in its epilogue; 8 / 16 / 32 / 64 / 128 tiles per plane: the segmented dot reduction and whole 32-tile segments;
    tile counts that pad to 32; K = 64 ... 512 incl. a non-power-of-two chunk count), m = 11 / 12: the same with K split 3 / 2 ways
    (slabs + w2e's finish kernel), all of them run twice and compared BIT FOR BIT (no atomics on any of its paths); m = 8: the FUSED
    kernel (w2e_wino_fused: persistent, transform + matrix waves, the patch by LDS-DMA; a grid of 3 workgroups so that each walks
    through several blocks, 1 to 4 output-channel blocks, K = 32 ... 128, both block shapes, the fused dot as per-block partials),
    image borders on every side.  Tolerances: FWD_TOL = 1e-4 against float64; the measured distance is printed by -s (~1e-5 at K = 512)."""
    import torch.nn.functional as F
    from where2edit_amd import functional as K
    g = torch.Generator().manual_seed(17 * k + n + h)
    wt = torch.randn(n, k, 3, 3, generator=g).to(DEV)
    scale = (k * 9) ** -0.5
    x = torch.randn(b, k, h, w, generator=g).to(DEV)
    s_in = (torch.randn(b, k, generator=g) * 0.3 + 1).to(DEV)
    s_out = (torch.rand(b, n, generator=g) + 0.5).to(DEV)
    wd, xd = wt.double() * scale, x.double() * s_in.double()[:, :, None, None]
    so = s_out.double()[:, :, None, None]
    ref = F.conv2d(xd, wd, padding=1) * so
    fwd, bwd = K.conv_pack(wt, scale, False, False), K.conv_pack(wt, scale, True, True)
    noise = torch.randn(1, 1, h, w, generator=g).to(DEV)
    nw, bias = torch.randn(1, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    pre = ref + nw.double() * noise.double() + bias.double()[None, :, None, None]
    gy = torch.randn(b, n, h, w, generator=g).to(DEV)          # the input-gradient pass: in_scale = s_out (demod), out_scale = s_in
    xdot = torch.randn(b, k, h, w, generator=g).to(DEV)
    graw = F.conv_transpose2d(gy.double() * so, wd, padding=1)
    saved, saved_f, saved_s = K.WINOGRAD, K.FUSED_WGS, K.GEMM_SPLITS
    if m in (11, 12):  # the GEMM form with a forced K split
        K.GEMM_SPLITS = {11: 3, 12: 2}[m]
        m = 4
    elif m == 8:  # 3 workgroups: several blocks each
        K.FUSED_WGS = 3
    try:
        K.set_winograd(False)
        y_direct, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, act=(noise, nw, bias))
        gx_direct, gs_direct = K._modconv_raw(K.MODE_SAME, gy, bwd, s_out, s_in, h, w, dot_with=xdot)
        K.set_winograd(m)
        assert K._wino_form(x, k, n, h, w, None) == m
        y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w)
        assert_close(y, ref, FWD_TOL, "winograd, plain epilogue")
        y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, None, None, h, w)
        assert_close(y, F.conv2d(x.double(), wd, padding=1), FWD_TOL, "winograd, unmodulated")
        y, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, act=(noise, nw, bias))
        assert_close(y, F.leaky_relu(pre, 0.2) * 2 ** 0.5, FWD_TOL, "winograd + act")
        assert_close(y, y_direct, 1e-4, "winograd == direct kernel")
        print(f"F({m}x{m},3x3) K={k}: |y - direct| / |direct| = {rel_err(y, y_direct):.2e}")
        K.WINO_LOG = []
        gx, gs = K._modconv_raw(K.MODE_SAME, gy, bwd, s_out, s_in, h, w, dot_with=xdot)
        log, K.WINO_LOG = K.WINO_LOG, None
        if K._wino_form(gy, n, k, h, w, xdot):  # (the form under test ran the gradient pass too -- unless its shapes exclude the transposed layer)
            assert len(log) == 1 and ("gemm" in log[0]) == (m == 4) and "dot" in log[0], log
        else:
            assert m == 8 and not log, log
        assert_close(gx, graw * s_in.double()[:, :, None, None], FWD_TOL, "winograd input gradient")
        assert_close(gs, (graw * xdot.double()).sum((2, 3)), FWD_TOL, "winograd fused dot")
        assert_close(gx, gx_direct, 1e-4, "input gradient: winograd == direct")
        assert_close(gs, gs_direct, 1e-4, "fused dot: winograd == direct")
        if m == 4:  # fixed summation order everywhere (K chunks ascending, slabs ascending, dot segments ascending): bit-identical reruns
            y2, _ = K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, act=(noise, nw, bias))
            gx2, gs2 = K._modconv_raw(K.MODE_SAME, gy, bwd, s_out, s_in, h, w, dot_with=xdot)
            assert torch.equal(y, y2) and torch.equal(gx, gx2) and torch.equal(gs, gs2), "the GEMM form is not bit-reproducible"
        out = torch.full((b + 1, n, h, w), 7.0, device=DEV)      # writing into the tail rows of a larger batch
        K._modconv_raw(K.MODE_SAME, x, fwd, s_in, s_out, h, w, out=out[1:])
        assert_close(out[1:], ref, FWD_TOL, "winograd into a view")
        assert float(out[0].min()) == 7.0 and float(out[0].max()) == 7.0
    finally:
        K.set_winograd(saved)
        K.FUSED_WGS = saved_f
        K.GEMM_SPLITS, K.WINO_LOG = saved_s, None

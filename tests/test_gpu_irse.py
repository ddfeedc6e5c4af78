"""N1: the IR-SE50 identity-loss network on the hand-written kernels (include/w2e_irse.h) -- the generic conv entry
point against torch convolutions in float64, the backbone against the fixture captured from the reference's Backbone
(tests/golden/irse.npz) and against the oracle (oracle/irse.py) including the input gradient, and IDLoss as a whole."""
import types

import pytest
import torch
import torch.nn.functional as F

import seeded
from helpers import assert_close, assert_grad_close, golden
from oracle import irse as OI

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("b,k,n,h,w", [(2, 64, 64, 28, 28), (1, 3, 64, 112, 112), (2, 24, 40, 14, 18), (3, 128, 128, 7, 7)])
def test_conv3x3_same_with_folded_bn_and_prelu(b, k, n, h, w):
    from where2edit_amd import functional as K
    from where2edit_amd import irse_hip as I
    g = torch.Generator().manual_seed(k * 7 + n)
    wt = torch.randn(n, k, 3, 3, generator=g) * (k * 9) ** -0.5
    x = torch.randn(b, k, h, w, generator=g)
    a, bias, slope = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g), torch.rand(n, generator=g) * 0.5 + 0.05
    ref = F.prelu(F.conv2d(x.double(), wt.double(), padding=1) * a.double()[None, :, None, None] + bias.double()[None, :, None, None],
                  slope.double())
    pack = K.conv_pack(wt.to(DEV), 1.0, False, False)
    y = I.conv3x3(x.to(DEV), pack, n, h, w, out_scale=a.to(DEV)[None].repeat(b, 1).contiguous(), bias=bias.to(DEV), slope=slope.to(DEV))
    assert_close(y, ref, 1e-4, "conv + BN + PReLU")
    y = I.conv3x3(x.to(DEV), pack, n, h, w, bias=bias.to(DEV))
    assert_close(y, F.conv2d(x.double(), wt.double(), padding=1) + bias.double()[None, :, None, None], 1e-4, "conv + bias")


@pytest.mark.parametrize("b,c,r,hw", [(3, 64, 4, 56 * 56), (2, 512, 32, 49), (1, 40, 5, 9)])
def test_se_gate_kernels_equal_the_stock_composition(b, c, r, hw):
    """w2e_se_gate_fwd / _bwd against mean -> fc1 -> ReLU -> fc2 -> sigmoid and its autograd gradient (helpers.py:56-72) in float64."""
    from where2edit_amd import irse_hip as I
    g = torch.Generator().manual_seed(c + r)
    plan = types.SimpleNamespace(depth=c, fc1=(torch.randn(r, c, generator=g) * c ** -0.5).to(DEV), fc2=torch.randn(c, r, generator=g).to(DEV))
    sums, dgate = torch.randn(b, c, generator=g) * hw, torch.randn(b, c, generator=g)
    gate, hidden = I.UnitPlan.gate(plan, sums.to(DEV), 1.0 / hw)
    pooled = (sums.double() / hw).requires_grad_(True)
    h_ref = torch.relu(pooled @ plan.fc1.double().cpu().t())
    g_ref = torch.sigmoid(h_ref @ plan.fc2.double().cpu().t())
    assert_close(hidden, h_ref, 1e-5, "SE hidden")
    assert_close(gate, g_ref, 1e-5, "SE gate")
    (gp_ref,) = torch.autograd.grad(g_ref, pooled, dgate.double())
    gpool = I.UnitPlan.gate_bwd(plan, dgate.to(DEV), gate, hidden, 1.0 / hw)
    assert_close(gpool, gp_ref / hw, 1e-5, "SE gate backward")


@pytest.mark.parametrize("b,k,n,h", [(2, 64, 64, 28), (1, 40, 72, 9), (2, 256, 512, 7)])
def test_conv3x3_stride2_pad1_and_its_adjoint(b, k, n, h):
    """DOWN with down_pad = nn.Conv2d(k, n, 3, 2, 1) on a [b,k,2h,2h] input; UP + the (+1,+1) crop = its input gradient."""
    from where2edit_amd import functional as K
    from where2edit_amd import irse_hip as I
    g = torch.Generator().manual_seed(k + n + h)
    wt = torch.randn(n, k, 3, 3, generator=g) * (k * 9) ** -0.5
    x = torch.randn(b, k, 2 * h, 2 * h, generator=g)
    a, bias = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    xd = x.double().requires_grad_(True)
    ref = F.conv2d(xd, wt.double(), stride=2, padding=1) * a.double()[None, :, None, None] + bias.double()[None, :, None, None]
    y = I.conv3x3(x.to(DEV), K.conv_pack(wt.to(DEV), 1.0, False, False), n, h, h, mode=K.MODE_DOWN, down_pad=1,
                  out_scale=a.to(DEV)[None].repeat(b, 1).contiguous(), bias=bias.to(DEV))
    assert_close(y, ref, 1e-4, "stride-2 pad-1 conv + BN")
    gy = torch.randn(b, n, h, h, generator=g)
    (gref,) = torch.autograd.grad(ref, xd, gy.double())
    tt = I.conv3x3(gy.to(DEV), K.conv_pack(wt.to(DEV), 1.0, True, False), k, h, h, mode=K.MODE_UP, in_scale=a.to(DEV)[None].repeat(b, 1).contiguous())
    gx = I.affine_act_bwd(tt, None, None, None, b, k, 2 * h, 2 * h, planar=True)
    assert_close(gx, gref, 1e-4, "input gradient of the stride-2 conv")
    # centre-tap pack = the 1x1 stride-2 shortcut convolution (helpers.py:103-106)
    w1 = torch.randn(n, k, 1, 1, generator=g) * k ** -0.5
    w9 = torch.zeros(n, k, 3, 3)
    w9[:, :, 1, 1] = w1[:, :, 0, 0]
    y = I.conv3x3(x.to(DEV), K.conv_pack(w9.to(DEV), 1.0, False, False), n, h, h, mode=K.MODE_DOWN, down_pad=1)
    assert_close(y, F.conv2d(x.double(), w1.double(), stride=2), 1e-4, "1x1 stride-2 shortcut")


def _backbone():
    from where2edit_amd.id_loss import Backbone
    net = Backbone(112, 50, drop_ratio=0.6, mode="ir_se").eval()
    sd = seeded.irse_fill(net.state_dict())
    net.load_state_dict(sd, strict=True)
    return net.to(DEV).requires_grad_(False), sd


def test_irse50_backbone_on_hip_matches_reference_fixture_and_oracle_gradient():
    net, sd = _backbone()
    x = seeded.tensor("irse.x", (2, 3, 112, 112), 0.5)
    xg = x.to(DEV).requires_grad_(True)
    y = net(xg)
    assert hasattr(net, "_plan"), "the HIP path did not run"
    assert_close(y, golden("irse")["feats"], 1e-4, "IR-SE50 features vs the reference's Backbone")
    r = seeded.tensor("irse.r", (2, 512))
    (gx,) = torch.autograd.grad((y * r.to(DEV)).sum(), xg)
    xo = x.clone().requires_grad_(True)
    yo = OI.backbone(sd, xo)
    (go,) = torch.autograd.grad((yo * r).sum(), xo)
    assert_grad_close(gx, go, "IR-SE50 input gradient")


def test_id_loss_on_hip_matches_oracle_with_gradient():
    """criteria/id_loss.py:19-40 end to end at 256^2 and 1024^2-shaped inputs: the fused pool-crop-pool, the batched
    [y_hat; y] embedding pass, loss value and d loss / d y_hat."""
    from where2edit_amd.id_loss import IDLoss
    mod = IDLoss(types.SimpleNamespace(ir_se50_weights=None))
    sd = seeded.irse_fill(mod.facenet.state_dict())
    mod.facenet.load_state_dict(sd, strict=True)
    mod = mod.to(DEV)
    for size in (256, 512):
        y = seeded.tensor(f"id.y{size}", (2, 3, size, size), 0.5)
        yh = (y + 0.2 * seeded.tensor(f"id.d{size}", (2, 3, size, size))).clone()
        yg = yh.to(DEV).requires_grad_(True)
        loss, zero = mod(yg, y.to(DEV))
        (g,) = torch.autograd.grad(loss, yg)
        yo = yh.clone().requires_grad_(True)
        lo = OI.id_loss(sd, yo, y)
        (go,) = torch.autograd.grad(lo, yo)
        assert zero == 0 and abs(float(loss) - float(lo)) <= 1e-4 * max(abs(float(lo)), 1e-3)
        assert_grad_close(g, go, f"d id_loss / d y_hat at {size}")


def test_e4e_encoder_on_hip_matches_reference_fixture():
    """N3: Encoder4Editing / GradualStyleEncoder (models/encoders/psp_encoders.py:58-200) with the IR-SE50 body, the
    map2style stride-2 stacks and the lateral 1x1 convolutions on the conv engine, against the reference's outputs."""
    import make_golden_e4e as ME
    from where2edit_amd.psp_encoders import Encoder4Editing, GradualStyleEncoder
    g = golden("e4e")
    x = seeded.tensor("e4e.x", (1, 3, 256, 256), 0.5).to(DEV)
    opts = types.SimpleNamespace(stylegan_size=1024)
    for name, cls in (("e4e", Encoder4Editing), ("gse", GradualStyleEncoder)):
        net = cls(50, "ir_se", opts).eval()
        net.load_state_dict(ME.encoder_state_dict(net.state_dict()), strict=True)
        net = net.to(DEV).requires_grad_(False)
        with torch.no_grad():
            w = net(x)
            assert hasattr(net, "_plan") and hasattr(net.styles[0], "_pack"), "the HIP path did not run"
            assert_close(w, g[name + ".w"], 1e-4, name + " W+ codes")
            if name == "e4e":
                c1, c2, c3 = net._taps(x)
                for t, key in ((c1, "c1"), (c2, "c2"), (c3, "c3")):
                    assert_close(t[:, ::8, ::4, ::4], g[f"e4e.{key}_strided"], 1e-4, key)
            w2 = net(torch.cat([x, x.flip(3)]))  # batch 2
            assert_close(w2[:1], g[name + ".w"], 1e-4, name + " batch-2, sample 0")
            # batch 4 (one GPU's share of BASELINE configs[4]): the work-gated Winograd forms of the stride-1 convs run here, not at batch 1
            from where2edit_amd import functional as KF
            KF.WINO_LOG = []
            try:
                w4 = net(torch.cat([x.flip(2), x.flip(3), x, x.flip(2).flip(3)]))
            finally:
                log, KF.WINO_LOG = KF.WINO_LOG, None
            assert sum("fused" in l for l in log) >= 4 and sum("gemm" in l for l in log) >= 4, log
            assert_close(w4[2:3], g[name + ".w"], 1e-4, name + " batch-4, sample 2")
            assert_close(w4[1:2], w2[1:2], 1e-4, name + " batch-4 sample 1 == batch-2 sample 1")


@pytest.mark.parametrize("shape,out", [((2, 16, 16, 16), (32, 32)), ((1, 5, 7, 9), (20, 31)), ((3, 8, 32, 32), (64, 64))])
def test_upsample_add_kernel_equals_bilinear_interpolate(shape, out):
    """w2e_upsample_add (the encoders' FPN merge, helpers.py:123-140) against F.interpolate(bilinear, align_corners=True) + y."""
    import torch.nn.functional as F
    from where2edit_amd.psp_encoders import _upsample_add
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(DEV)
    y = torch.randn(shape[0], shape[1], *out, generator=g).to(DEV)
    with torch.no_grad():
        got = _upsample_add(x, y)
    ref = F.interpolate(x.double(), size=out, mode="bilinear", align_corners=True) + y.double()
    assert_close(got, ref, 2e-6, "upsample + add")


@pytest.mark.parametrize("bsz", [1, 4])
def test_config5_invert_and_edit_pipeline_matches_oracle(bsz):
    """BASELINE configs[4] end to end -- e4e -> S codes -> features -> region-attention net -> masked 1024^2 generator
    (show_demo/try_demo.py:93-157), every stage against the oracle composition -- for one image and at the batch one GPU of the
    8-GPU configuration sees (32 / 8 = 4): the encoder's kernel selection is gated on the work per call (irse_hip._wino_form), so at
    batch 4 its 64- / 128-channel stages take the fused Winograd kernel and its 256- / 512-channel ones the GEMM form where batch 1
    keeps the direct kernel for most of them; the log of the forms that ran is asserted."""
    import make_golden_attention as MA
    import make_golden_e4e as ME
    from make_golden import CLIP_TINY as c
    from oracle import attention_net as OA
    from oracle import clip_model as OC
    from oracle import e4e as OE
    from oracle import ops as OO
    from oracle import stylegan2 as OG
    from where2edit_amd.attention_model import Generator
    from where2edit_amd.clip_loss import CLIPLoss
    from where2edit_amd.clip_vit import CLIP
    from where2edit_amd.demo_pipeline import gaussian_blur5, invert_and_edit
    from where2edit_amd.psp_encoders import Encoder4Editing
    from where2edit_amd.run_attention import FullSpaceMapperFEATClusterLinStyle_Net
    size, k, att = 1024, 20, 13
    opts = types.SimpleNamespace(stylegan_size=size)
    e4e = Encoder4Editing(50, "ir_se", opts).eval()
    esd = ME.encoder_state_dict(e4e.state_dict())
    e4e.load_state_dict(esd, strict=True)
    gsd = seeded.generator_state_dict(size)
    g = Generator(size, 512, 8)
    g.load_state_dict(gsd, strict=True)
    clip = CLIP(embed_dim=c["embed_dim"], vision_layers=c["vision_layers"], vision_width=c["vision_width"],
                context_length=c["context_length"], vocab_size=c["vocab_size"], transformer_width=c["text_width"],
                transformer_heads=1, transformer_layers=c["text_layers"])
    csd = seeded.clip_state_dict(**c)
    clip.load_state_dict(csd, strict=True)
    edim = c["embed_dim"]
    net = FullSpaceMapperFEATClusterLinStyle_Net(18, edim + 512, edim, attention_layer=att, channel_multiplier=2, cluster_layer=att,
                                                 clusters=k, cluster_dim=576)
    msd = MA.net_state_dict(net)
    msd["initial_bias"] = torch.tensor([0.9])  # about half of the 20 cluster means pass the 0.8 threshold
    img = seeded.tensor("cfg5.img", (bsz, 3, 256, 256), 0.5)
    text, att_text = seeded.tensor("cfg5.text", (bsz, edim), 0.3), seeded.tensor("cfg5.att", (bsz, edim), 0.3)
    # oracle, stage by stage
    with torch.no_grad():
        w_o = OE.encoder4editing(esd, img)
        _, _, codes_o = OG.generator_forward(gsd, [w_o], size=size, input_is_latent=True, randomize_noise=False, return_latents=True)
        img_o, _, _, feats_o = OG.generator_forward(gsd, [codes_o], size=size, input_is_stylespace=True, randomize_noise=False, return_features=True)
        feats_o = list(feats_o) + [gsd["input.input"].repeat(bsz, 1, 1, 1)]  # (run_attention.py:1110 appends the constant input, per sample)
        # centres = 20 pixels of the layer-13 activation (+ their positions): a non-trivial, well-separated assignment
        f13 = feats_o[att - 1]
        idx = torch.randperm(64 * 64, generator=torch.Generator().manual_seed(3))[:k]
        ys, xs = (idx // 64).float() * 2 / 63 - 1, (idx % 64).float() * 2 / 63 - 1
        msd["initial_state"] = torch.cat([f13[0].reshape(512, -1)[:, idx].t(), xs[:, None].repeat(1, 32), ys[:, None].repeat(1, 32)], 1)
        x = [torch.cat([text.unsqueeze(1), s[:, :, :, 0, 0]], -1) for s in codes_o]
        new_o, _, _, extra = OA.forward(msd, x, feats_o, 64, attention_text=att_text, attention_layer=att, cluster_layer=att, clusters=k,
                                        latent_dim=edim)
        # the demo's net returns the RAW cluster-pooled map (show_demo/utils_demo.py:135-139); one threshold + one blur (:154-155)
        mask_o = extra["same"].unsqueeze(1)
        mask_o = OA.gaussian_blur5(torch.where(mask_o < 0.8, torch.zeros_like(mask_o), mask_o))
        gen_o, _, _, _ = OG.generator_forward(gsd, [new_o], size=size, input_is_stylespace=True, randomize_noise=False, return_features=True,
                                              attention_layer=att, attention_map=mask_o, feature_map=feats_o)
        fo = OC.encode_image(csd, OO.clip_preprocess(gen_o, size))
    net.load_state_dict(msd, strict=True)
    from where2edit_amd import functional as KF
    KF.WINO_LOG = []
    try:
        out = invert_and_edit(img.to(DEV), e4e.to(DEV).requires_grad_(False), g.to(DEV).requires_grad_(False),
                              CLIPLoss(opts, model=clip).to(DEV), net.to(DEV).requires_grad_(False), text.to(DEV), att_text.to(DEV),
                              attention_layer=att)
    finally:
        log, KF.WINO_LOG = KF.WINO_LOG, None
    enc = [l for l in log if l.startswith("conv3x3")]
    n_fused, n_gemm = sum("fused" in l for l in enc), sum("gemm" in l for l in enc)
    print(f"config 5, batch {bsz}: encoder convs on the fused Winograd kernel: {n_fused}, on the GEMM form: {n_gemm}; generator layers: "
          f"{sum(l.startswith('modconv') for l in log)}")
    if bsz >= 4:  # (the forms the measured configuration runs are the ones this comparison covers)
        assert n_fused >= 4 and n_gemm >= 4, enc
    assert_close(out["latents"], w_o, 1e-4, "e4e W+")
    assert_close(out["img_orig"], img_o, 1e-4, "img_orig")
    assert 0.02 < float((mask_o > 0).float().mean()) < 0.98, "degenerate mask: the test would not exercise the blend"
    assert_close(out["mask"], mask_o, 1e-4, "mask")
    assert_close(out["img_gen"], gen_o, 1e-4, "edited image")
    assert_close(out["features_gen"], fo, 1e-3, "CLIP features of the edit")
    assert_close(gaussian_blur5(out["mask"]), OA.gaussian_blur5(mask_o), 1e-5, "blur")
    # the same pipeline replayed as one hipGraph, on new inputs copied into its static buffers
    from where2edit_amd.demo_pipeline import capture_invert_and_edit
    clip_dev = CLIPLoss(opts, model=clip).to(DEV)
    run = capture_invert_and_edit(torch.zeros_like(img).to(DEV), e4e, g, clip_dev, net, torch.zeros_like(text).to(DEV), torch.zeros_like(att_text).to(DEV),
                                  attention_layer=att)
    rep = run(img.to(DEV), text.to(DEV), att_text.to(DEV))
    for key in ("latents", "img_orig", "mask", "img_gen", "features_gen"):
        assert_close(rep[key], out[key], 1e-5, f"graph replay: {key}")  # (the direct kernels' split-K joins by fp32 atomics: run-to-run rounding)


@pytest.mark.parametrize("m,b,k,n,h,w", [(4, 3, 64, 128, 28, 28), (4, 2, 256, 256, 28, 28), (4, 1, 64, 64, 112, 112), (4, 5, 128, 64, 56, 28),
                                         (4, 16, 128, 256, 28, 28), (4, 16, 256, 256, 14, 14), (4, 8, 256, 256, 14, 14), (4, 3, 512, 512, 7, 7),
                                         (4, 2, 64, 128, 10, 6), (8, 2, 64, 64, 32, 64), (8, 3, 128, 256, 16, 32)])
def test_conv3x3_winograd_forms(m, b, k, n, h, w):
    """The stride-1 convs of the IR-SE50 / e4e encoders through the Winograd F(4x4,3x3) forms -- m = 4: the own contraction kernel
    (w2e_wino_pack_input + w2e_wino_gemm with the bias + PReLU epilogue of w2e_conv3x3, incl. the K split the plan picks for these small
    layers); m = 8: the fused kernel (w2e_wino_fused version 3) with the same epilogue: forward with BN scale, bias and PReLU, the
    input-gradient form (in_scale on the transposed + flipped pack), against float64 and against the direct kernel; tile counts per
    image that are no multiple of 32 (196, 49) and totals that pad (49 x 3 = 147 -> 160); image sizes that are no multiple of 4 -- 14^2,
    7^2 (IR-SE50's last two stages), 10 x 6: the ragged form, tiles hanging over the image, with and without a K split."""
    import torch.nn.functional as F
    from where2edit_amd import functional as K, irse_hip as I
    g = torch.Generator().manual_seed(5 * k + n + h)
    wt = (torch.randn(n, k, 3, 3, generator=g) * (k * 9) ** -0.5).to(DEV)
    x = torch.randn(b, k, h, w, generator=g).to(DEV)
    a = (torch.rand(b, n, generator=g) + 0.5).to(DEV)
    bias, slope = torch.randn(n, generator=g).to(DEV), (torch.rand(n, generator=g) * 0.5 + 0.05).to(DEV)
    fwd, bwd = K.conv_pack(wt, 1.0, False, False), K.conv_pack(wt, 1.0, True, True)
    pre = F.conv2d(x.double(), wt.double(), padding=1) * a.double()[:, :, None, None] + bias.double()[None, :, None, None]
    ref = torch.where(pre > 0, pre, pre * slope.double()[None, :, None, None])
    y0 = I.conv3x3(x, fwd, n, h, w, out_scale=a, bias=bias, slope=slope, form=0)
    K.WINO_LOG = []
    try:
        y = I.conv3x3(x, fwd, n, h, w, out_scale=a, bias=bias, slope=slope, form=m)
        assert len(K.WINO_LOG) == 1 and ("fused" in K.WINO_LOG[0]) == (m == 8) and ("gemm" in K.WINO_LOG[0]) == (m == 4), K.WINO_LOG  # (the form asked for is the one that ran)
    finally:
        K.WINO_LOG = None
    assert_close(y, ref, 1e-4, "winograd conv + BN + PReLU"), assert_close(y, y0, 1e-4, "winograd == direct")
    y = I.conv3x3(x, fwd, n, h, w, out_scale=a, bias=bias, form=m)
    assert_close(y, pre, 1e-4, "winograd conv + BN (no PReLU)")
    gy = torch.randn(b, n, h, w, generator=g).to(DEV)
    gx = I.conv3x3(gy, bwd, k, h, w, in_scale=a, form=m)
    assert_close(gx, F.conv_transpose2d(gy.double() * a.double()[:, :, None, None], wt.double(), padding=1), 1e-4, "winograd input gradient")
    out = torch.zeros(b + 2, k, h, w, device=DEV)
    I.conv3x3(gy, bwd, k, h, w, in_scale=a, out=out[:b], form=m)
    assert torch.equal(out[:b], gx), "winograd into a view: not bit-identical (no path of either form uses atomics here)"
    assert float(out[b:].abs().max()) == 0.0

"""Golden-vector generator.  Runs ONLY in the build container, where the reference
checkout exists read-only at /root/reference; the reference never travels, so
the outputs are committed as small .npz fixtures next to this script.

    python tests/golden/make_golden.py [case ...]

Every case imports the reference's own modules (models.stylegan2.*,
attention/attention_model.py, mapper.latent_mappers, mapper.training.ranger,
models.facial_recognition.model_irse), loads the deterministic synthetic
weights of tests/golden/seeded.py into them (strict=True, which also pins the
state_dict schema), runs them on CPU and stores inputs-by-name + outputs.
Harness-side adaptations, reference source untouched:
  * torch.Tensor.cuda -> identity (models/stylegan2/op/fused_act.py:25 hard-codes
    `.cuda()`; there is no GPU here);
  * CLIP: OpenAI `clip` is absent, so the CLIP fixtures come from the independent
    `transformers` implementation instead (cross-check, not reference parity).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import seeded  # noqa: E402


def _import_reference():
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "attention"))
    import models.stylegan2.model as ref_model
    import models.stylegan2.op as ref_op
    import attention_model as ref_att
    import mapper.latent_mappers as ref_mappers
    from mapper.training.ranger import Ranger
    return ref_model, ref_op, ref_att, ref_mappers, Ranger


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (_np(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------- ops
UPFIRDN_CASES = [
    # name, shape, taps(or 2-D key), gain, up, down, pad
    ("blur_up_conv", (2, 5, 17, 17), (1, 3, 3, 1), 4.0, 1, 1, (1, 1)),  # Blur after the stride-2 tconv (model.py:199-206)
    ("rgb_upsample", (2, 3, 8, 8), (1, 3, 3, 1), 4.0, 2, 1, (2, 1)),    # Upsample (model.py:31-49)
    ("downsample", (1, 4, 16, 12), (1, 3, 3, 1), 1.0, 1, 2, (1, 1)),    # Downsample (model.py:52-70)
    ("blur_down_conv", (1, 3, 12, 12), (1, 3, 3, 1), 1.0, 1, 1, (2, 2)),  # Blur before stride-2 conv (model.py:208-215)
    ("asym_kernel_updown", (2, 2, 7, 9), "asym5x3", 1.0, 3, 2, (2, 3)),  # non-symmetric kernel: pins the flip
    ("crop_negative_pad", (1, 2, 10, 10), (1, 2, 1), 1.0, 1, 1, (-1, -2)),  # negative pads crop (upfirdn2d.py:33-41)
    ("single_pixel", (1, 1, 1, 1), (1, 3, 3, 1), 4.0, 2, 1, (2, 1)),
]


def _kernel(spec, gain):
    if spec == "asym5x3":
        return seeded.tensor("asym5x3", (5, 3))
    return seeded.fir_kernel(spec, gain)


def case_ops():
    _, ref_op, _, _, _ = _import_reference()
    out = {}
    for name, shape, kspec, gain, up, down, pad in UPFIRDN_CASES:
        x = seeded.tensor("upfirdn." + name, shape).requires_grad_(True)
        k = _kernel(kspec, gain)
        y = ref_op.upfirdn2d(x, k, up=up, down=down, pad=pad)
        gy = seeded.tensor("upfirdn.gy." + name, y.shape)
        (gx,) = torch.autograd.grad(y, x, gy)
        out[f"upfirdn.{name}.y"] = y
        out[f"upfirdn.{name}.gx"] = gx
    for name, shape in [("nchw", (2, 6, 5, 7)), ("seq3d", (2, 4, 6)), ("mat2d", (3, 6))]:
        x = seeded.tensor("flrelu." + name, shape).requires_grad_(True)
        c = shape[-1] if len(shape) == 3 else shape[1]
        b = seeded.tensor("flrelu.b." + name, (c,)).requires_grad_(True)
        y = ref_op.fused_leaky_relu(x, b)
        gy = seeded.tensor("flrelu.gy." + name, y.shape)
        gx, gb = torch.autograd.grad(y, (x, b), gy)
        out[f"flrelu.{name}.y"], out[f"flrelu.{name}.gx"], out[f"flrelu.{name}.gb"] = y, gx, gb
    x = seeded.tensor("flrelu.slope", (2, 3, 4, 4))
    out["flrelu.slope.y"] = ref_op.fused_leaky_relu(x, seeded.tensor("flrelu.slope.b", (3,)), 0.1, 1.5)
    _save("ops", **out)


# ----------------------------------------------------------------------------- modconv
MODCONV_CASES = [
    # name, cin, cout, k, demod, up, B, H
    ("same3", 8, 16, 3, True, False, 2, 8),
    ("up3", 8, 12, 3, True, True, 2, 5),
    ("rgb1", 16, 3, 1, False, False, 3, 6),
    ("same3_wide", 32, 32, 3, True, False, 1, 16),
]


def modconv_inputs(name, cin, cout, k, b, h):
    return dict(
        x=seeded.tensor(f"modconv.{name}.x", (b, cin, h, h)),
        w=seeded.tensor(f"modconv.{name}.w", (b, 512)),
        weight=seeded.tensor(f"modconv.{name}.weight", (1, cout, cin, k, k)),
        mod_w=seeded.tensor(f"modconv.{name}.mod_w", (cin, 512)),
        mod_b=seeded.tensor(f"modconv.{name}.mod_b", (cin,), 0.05, 1.0),
    )


def case_modconv():
    ref_model, _, _, _, _ = _import_reference()
    out = {}
    for name, cin, cout, k, demod, up, b, h in MODCONV_CASES:
        i = modconv_inputs(name, cin, cout, k, b, h)
        m = ref_model.ModulatedConv2d(cin, cout, k, 512, demodulate=demod, upsample=up)
        sd = {"weight": i["weight"], "modulation.weight": i["mod_w"], "modulation.bias": i["mod_b"]}
        if up:
            sd["blur.kernel"] = seeded.fir_kernel(gain=4.0)
        m.load_state_dict(sd, strict=True)
        x = i["x"].clone().requires_grad_(True)
        w = i["w"].clone().requires_grad_(True)
        y, s = m(x, w)
        gy = seeded.tensor(f"modconv.{name}.gy", y.shape)
        gx, gw = torch.autograd.grad(y, (x, w), gy)
        out[f"{name}.y"], out[f"{name}.s"], out[f"{name}.gx"], out[f"{name}.gw"] = y, s, gx, gw
        # S-space entry: feed the post-affine style back (model.py:237-238)
        y2, _ = m(i["x"], s.detach(), input_is_stylespace=True)
        out[f"{name}.y_sspace"] = y2
    _save("modconv", **out)


# ----------------------------------------------------------------------------- generator
def _layer_stats(feats, key):
    rows, samples = [], []
    for n, f in enumerate(feats):
        rows.append([f.mean().item(), f.std().item(), f.abs().max().item()])
        pos = seeded.sample_positions(f.numel(), 32, f"{key}.{n}")
        samples.append(_np(f.reshape(-1)[pos]))
    return np.asarray(rows, dtype=np.float64), np.stack(samples)


def case_generator16():
    """size=16 (5 styled convs at 512 ch): full outputs, gradients, every input mode."""
    ref_model, _, ref_att, _, _ = _import_reference()
    size = 16
    sd = seeded.generator_state_dict(size)
    g = ref_model.Generator(size, 512, 8)
    g.load_state_dict(sd, strict=True)
    ga = ref_att.Generator(size, 512, 8)
    ga.load_state_dict(sd, strict=True)
    g.eval(), ga.eval()
    out = {}
    w = seeded.wplus_latents(2, g.n_latent).requires_grad_(True)
    img, lat, svec = g([w], input_is_latent=True, randomize_noise=False, return_latents=True)
    r = seeded.tensor("gen16.r", img.shape)
    (gw,) = torch.autograd.grad((img * r).sum(), w)
    out["wplus.image"], out["wplus.grad_w"] = img, gw
    for n, s in enumerate(svec):
        out[f"wplus.style.{n}"] = s
    # z input + truncation toward a seeded "mean latent"
    z = seeded.tensor("gen16.z", (2, 512))
    tl = seeded.tensor("gen16.trunc", (1, 512), 0.3)
    out["z.image"] = g([z], truncation=0.7, truncation_latent=tl, randomize_noise=False)[0]
    # style mixing with a fixed inject index
    z2 = seeded.tensor("gen16.z2", (2, 512))
    out["mix.image"] = g([z, z2], inject_index=3, randomize_noise=False)[0]
    # single [B,512] w broadcast (model.py:514-515)
    out["wsingle.image"] = g([w[:, 0].detach()], input_is_latent=True, randomize_noise=False)[0]
    # S-space round trip (model.py:559-566)
    svec_d = [s.detach().clone().requires_grad_(True) for s in svec]
    img_s = g([svec_d], input_is_stylespace=True, randomize_noise=False)[0]
    gs = torch.autograd.grad((img_s * r).sum(), svec_d)
    out["sspace.image"] = img_s
    for n, t in enumerate(gs):
        out[f"sspace.grad.{n}"] = t
    # attention generator: features, then a blended re-synthesis from edited W+ (attention_model.py:473-676)
    with torch.no_grad():
        img_f, _, _, feats = ga([w.detach()], input_is_latent=True, randomize_noise=False, return_features=True)
    out["att.image"] = img_f
    for n, f in enumerate(feats):
        out[f"att.feat.{n}"] = f
    w2 = (w.detach() + 0.2 * seeded.tensor("gen16.dw", w.shape)).requires_grad_(True)
    for layer in (4, 3, 5, 1):
        mask = torch.rand(2, 1, 4, 4, generator=torch.Generator().manual_seed(layer)).requires_grad_(True)
        img_b, _, _, nf = ga([w2], input_is_latent=True, randomize_noise=False, return_features=True,
                             attention_layer=layer, attention_map=mask, feature_map=[f.detach() for f in feats])
        gw2, gm = torch.autograd.grad((img_b * r).sum(), (w2, mask))
        out[f"blend{layer}.mask"], out[f"blend{layer}.image"] = mask, img_b
        out[f"blend{layer}.grad_w"], out[f"blend{layer}.grad_mask"] = gw2, gm
        out[f"blend{layer}.feat_at"] = nf[layer - 1]
    # S-space + blend (the mode run_attention.py:1245 uses)
    mask = torch.rand(2, 1, 8, 8, generator=torch.Generator().manual_seed(99))
    sv2 = [s.detach() * 1.1 for s in svec]
    out["sblend.mask"] = mask
    out["sblend.image"] = ga([sv2], input_is_stylespace=True, randomize_noise=False, return_features=True,
                             attention_layer=4, attention_map=mask, feature_map=[f.detach() for f in feats])[0]
    _save("generator16", **out)


def case_generator_big():
    """256 and 1024 generators: strided image sample + per-layer statistics."""
    _, _, ref_att, _, _ = _import_reference()
    out = {}
    for size, batch in ((256, 1), (1024, 1)):
        sd = seeded.generator_state_dict(size)
        g = ref_att.Generator(size, 512, 8)
        g.load_state_dict(sd, strict=True)
        g.eval()
        w = seeded.wplus_latents(batch, g.n_latent, salt=size)
        with torch.no_grad():
            img, _, svec, feats = g([w], input_is_latent=True, randomize_noise=False, return_features=True)
        stride = size // 32
        out[f"g{size}.image_strided"] = img[:, :, ::stride, ::stride]
        out[f"g{size}.image_sum"] = np.float64(img.double().sum().item())
        out[f"g{size}.image_abs_sum"] = np.float64(img.double().abs().sum().item())
        stats, samples = _layer_stats(feats, f"g{size}.feat")
        out[f"g{size}.feat_stats"], out[f"g{size}.feat_samples"] = stats, samples
        out[f"g{size}.style_dims"] = np.asarray([s.shape[2] for s in svec])
    _save("generator_big", **out)


# ----------------------------------------------------------------------------- mappers / step
class _Opts(types.SimpleNamespace):
    pass


def case_mappers():
    _, _, _, ref_mappers, _ = _import_reference()
    out = {}
    opts = _Opts(no_coarse_mapper=False, no_medium_mapper=False, no_fine_mapper=False)
    x = seeded.wplus_latents(3, 18, salt=5).requires_grad_(True)
    lm = ref_mappers.LevelsMapper(opts)
    lm.load_state_dict(seeded.mapper_state_dict(["course_mapping.", "medium_mapping.", "fine_mapping."]), strict=True)
    y = lm(x)
    gy = seeded.tensor("mappers.levels.gy", y.shape)
    gx = torch.autograd.grad(y, x, gy, retain_graph=True)[0]
    gp = torch.autograd.grad(y, lm.course_mapping.mapping[1].weight, gy)[0]
    out["levels.y"], out["levels.gx"], out["levels.g_course1_w"] = y, gx, gp
    opts2 = _Opts(no_coarse_mapper=True, no_medium_mapper=False, no_fine_mapper=True)
    lm2 = ref_mappers.LevelsMapper(opts2)
    lm2.load_state_dict(seeded.mapper_state_dict(["medium_mapping."]), strict=True)
    out["levels_medium_only.y"] = lm2(x.detach())
    sm = ref_mappers.SingleMapper(opts)
    sm.load_state_dict(seeded.mapper_state_dict(["mapping."]), strict=True)
    out["single.y"] = sm(x.detach())
    dims = ref_mappers.STYLESPACE_DIMENSIONS
    out["stylespace_dims"] = np.asarray(dims)
    xs = [seeded.tensor(f"mappers.s.{c}", (2, 1, d, 1, 1), 0.5, 1.0) for c, d in enumerate(dims)]
    fm = ref_mappers.FullStyleSpaceMapper(opts)
    fm.load_state_dict(seeded.mapper_state_dict([f"mapper_{c}." for c in range(len(dims))], dims), strict=True)
    for c, t in enumerate(fm(xs)):
        out[f"full_s.{c}"] = t
    wm = ref_mappers.WithoutToRGBStyleSpaceMapper(opts)
    idx = wm.STYLESPACE_INDICES_WITHOUT_TORGB
    out["without_torgb_indices"] = np.asarray(idx)
    wm.load_state_dict(seeded.mapper_state_dict([f"mapper_{c}." for c in idx], [dims[c] for c in idx]), strict=True)
    for c, t in enumerate(wm(xs)):
        out[f"wo_rgb_s.{c}"] = t
    _save("mappers", **out)


def case_step():
    """The CLIP-free part of Coach.train's step (coach.py:80-92) assembled from the
    reference's own Generator and LevelsMapper at size=64 (n_latent=10, so the
    coarse/medium/fine groups are 4/4/2 layers and all three sub-mappers train):
    x=G(w); w_hat=w+0.1*M(w); x_hat=G(w_hat); L = <x_hat,R>/numel + 0.8*MSE(w_hat,w);
    backward.  The optimizer is pinned separately (case_ranger)."""
    ref_model, _, _, ref_mappers, _ = _import_reference()
    size = 64
    out = {}
    g = ref_model.Generator(size, 512, 8)
    g.load_state_dict(seeded.generator_state_dict(size), strict=True)
    opts = _Opts(no_coarse_mapper=False, no_medium_mapper=False, no_fine_mapper=False)
    m = ref_mappers.LevelsMapper(opts)
    m.load_state_dict(seeded.mapper_state_dict(["course_mapping.", "medium_mapping.", "fine_mapping."]), strict=True)
    w = seeded.wplus_latents(1, g.n_latent, salt=11)
    r = seeded.tensor("step.r", (1, 3, size, size))
    with torch.no_grad():
        x, _ = g([w], input_is_latent=True, randomize_noise=False, truncation=1)
    w_hat = w + 0.1 * m(w)
    x_hat, w_hat, _ = g([w_hat], input_is_latent=True, return_latents=True, randomize_noise=False, truncation=1)
    l_img = (x_hat * r).sum() / x_hat.numel()
    l_l2 = torch.nn.functional.mse_loss(w_hat, w)
    loss = l_img + 0.8 * l_l2
    loss.backward()
    out["x"], out["x_hat"], out["w_hat"] = x, x_hat, w_hat
    out["losses"] = np.asarray([loss.item(), l_img.item(), l_l2.item()], dtype=np.float64)
    for name, p in m.named_parameters():
        pos = seeded.sample_positions(p.numel(), 64, "step." + name)
        out[f"grad.{name}.samples"] = p.grad.reshape(-1)[pos]
        out[f"grad.{name}.norm"] = np.float64(p.grad.double().norm().item())
    _save("step", **out)


RANGER_SHAPES = [("fc.weight", (8, 16)), ("fc.bias", (8,)), ("conv.weight", (4, 3, 3, 3))]


def case_ranger():
    """mapper/training/ranger.py driven with seeded gradients for 13 steps at the
    reference's default lr=0.5 (train_options.py:27): crosses the RAdam
    N_sma threshold switch and two k=6 lookahead syncs."""
    _, _, _, _, Ranger = _import_reference()
    import warnings
    params = [torch.nn.Parameter(seeded.tensor("ranger.p." + n, s)) for n, s in RANGER_SHAPES]
    opt = Ranger(params, lr=0.5)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(13):
            for (n, s), p in zip(RANGER_SHAPES, params):
                p.grad = seeded.tensor(f"ranger.g.{n}", s, salt=it)
            opt.step()
            for (n, _), p in zip(RANGER_SHAPES, params):
                out[f"step{it}.{n}"] = p.detach().clone()
    _save("ranger", **out)


# ----------------------------------------------------------------------------- preprocessing (stock torch modules, as the reference composes them)
def case_preproc():
    out = {}
    for size in (1024, 256):
        img = seeded.tensor(f"preproc.img{size}", (1, 3, size, size)).requires_grad_(True)
        up = torch.nn.Upsample(scale_factor=7)           # criteria/clip_loss.py:11
        pool = torch.nn.AvgPool2d(kernel_size=size // 32)  # criteria/clip_loss.py:12
        y = pool(up(img))
        gy = seeded.tensor(f"preproc.gy{size}", y.shape)
        (gx,) = torch.autograd.grad(y, img, gy)
        out[f"clip{size}.y"] = y
        out[f"clip{size}.gx_strided"] = gx[:, :, :: size // 64, :: size // 64]
        out[f"clip{size}.gx_sum"] = np.float64(gx.double().sum().item())
    img = seeded.tensor("preproc.id", (2, 3, 1024, 1024))
    p256 = torch.nn.AdaptiveAvgPool2d((256, 256))  # criteria/id_loss.py:13-14,19-23
    p112 = torch.nn.AdaptiveAvgPool2d((112, 112))
    out["id1024.y"] = p112(p256(img)[:, :, 35:223, 32:220])
    img256 = seeded.tensor("preproc.id256", (1, 3, 256, 256))
    out["id256.y"] = p112(img256[:, :, 35:223, 32:220])
    _save("preproc", **out)


# ----------------------------------------------------------------------------- CLIP cross-check (transformers, NOT the reference)
def _hf_to_openai(hf_sd, v_layers, t_layers):
    sd = {}
    g = lambda k: hf_sd[k].detach().clone()
    sd["visual.conv1.weight"] = g("vision_model.embeddings.patch_embedding.weight")
    sd["visual.class_embedding"] = g("vision_model.embeddings.class_embedding")
    sd["visual.positional_embedding"] = g("vision_model.embeddings.position_embedding.weight")
    for a, b in (("visual.ln_pre", "vision_model.pre_layrnorm"), ("visual.ln_post", "vision_model.post_layernorm"),
                 ("ln_final", "text_model.final_layer_norm")):
        sd[a + ".weight"], sd[a + ".bias"] = g(b + ".weight"), g(b + ".bias")
    sd["visual.proj"] = g("visual_projection.weight").t().contiguous()
    sd["text_projection"] = g("text_projection.weight").t().contiguous()
    sd["token_embedding.weight"] = g("text_model.embeddings.token_embedding.weight")
    sd["positional_embedding"] = g("text_model.embeddings.position_embedding.weight")
    sd["logit_scale"] = g("logit_scale")
    for dst, src, n in (("visual.transformer.resblocks", "vision_model.encoder.layers", v_layers),
                        ("transformer.resblocks", "text_model.encoder.layers", t_layers)):
        for i in range(n):
            p, q = f"{dst}.{i}", f"{src}.{i}"
            sd[p + ".attn.in_proj_weight"] = torch.cat([g(f"{q}.self_attn.{n_}_proj.weight") for n_ in "qkv"], 0)
            sd[p + ".attn.in_proj_bias"] = torch.cat([g(f"{q}.self_attn.{n_}_proj.bias") for n_ in "qkv"], 0)
            for a, b in ((".attn.out_proj", ".self_attn.out_proj"), (".ln_1", ".layer_norm1"), (".ln_2", ".layer_norm2"),
                         (".mlp.c_fc", ".mlp.fc1"), (".mlp.c_proj", ".mlp.fc2")):
                sd[p + a + ".weight"], sd[p + a + ".bias"] = g(q + b + ".weight"), g(q + b + ".bias")
    return sd


def _openai_to_hf(sd, model):
    """Inverse mapping: load seeded OpenAI-format weights into a transformers CLIPModel."""
    hf = model.state_dict()
    v_layers = model.config.vision_config.num_hidden_layers
    t_layers = model.config.text_config.num_hidden_layers
    probe = _hf_to_openai(hf, v_layers, t_layers)
    assert set(probe) == set(sd), sorted(set(probe) ^ set(sd))[:5]
    new = dict(hf)
    new["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    new["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
    new["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    for a, b in (("visual.ln_pre", "vision_model.pre_layrnorm"), ("visual.ln_post", "vision_model.post_layernorm"),
                 ("ln_final", "text_model.final_layer_norm")):
        new[b + ".weight"], new[b + ".bias"] = sd[a + ".weight"], sd[a + ".bias"]
    new["visual_projection.weight"] = sd["visual.proj"].t().contiguous()
    new["text_projection.weight"] = sd["text_projection"].t().contiguous()
    new["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    new["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    new["logit_scale"] = sd["logit_scale"]
    for dst, src, n in (("visual.transformer.resblocks", "vision_model.encoder.layers", v_layers),
                        ("transformer.resblocks", "text_model.encoder.layers", t_layers)):
        for i in range(n):
            p, q = f"{dst}.{i}", f"{src}.{i}"
            w3 = sd[p + ".attn.in_proj_weight"].chunk(3, 0)
            b3 = sd[p + ".attn.in_proj_bias"].chunk(3, 0)
            for j, n_ in enumerate("qkv"):
                new[f"{q}.self_attn.{n_}_proj.weight"], new[f"{q}.self_attn.{n_}_proj.bias"] = w3[j].contiguous(), b3[j].contiguous()
            for a, b in ((".attn.out_proj", ".self_attn.out_proj"), (".ln_1", ".layer_norm1"), (".ln_2", ".layer_norm2"),
                         (".mlp.c_fc", ".mlp.fc1"), (".mlp.c_proj", ".mlp.fc2")):
                new[q + b + ".weight"], new[q + b + ".bias"] = sd[p + a + ".weight"], sd[p + a + ".bias"]
    missing = model.load_state_dict(new, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k], missing


CLIP_TINY = dict(embed_dim=32, image_resolution=224, vision_layers=2, vision_width=128, vision_patch=32,
                 context_length=16, vocab_size=100, text_width=64, text_layers=2)


def _hf_model(cfg):
    from transformers import CLIPConfig, CLIPModel
    config = CLIPConfig(
        vision_config=dict(hidden_size=cfg["vision_width"], intermediate_size=4 * cfg["vision_width"],
                           num_hidden_layers=cfg["vision_layers"], num_attention_heads=cfg["vision_width"] // 64,
                           image_size=cfg["image_resolution"], patch_size=cfg["vision_patch"], hidden_act="quick_gelu",
                           layer_norm_eps=1e-5, projection_dim=cfg["embed_dim"]),
        text_config=dict(hidden_size=cfg["text_width"], intermediate_size=4 * cfg["text_width"],
                         num_hidden_layers=cfg["text_layers"], num_attention_heads=cfg["text_width"] // 64,
                         max_position_embeddings=cfg["context_length"], vocab_size=cfg["vocab_size"],
                         hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=cfg["embed_dim"],
                         eos_token_id=2, bos_token_id=0, pad_token_id=1),  # eos_token_id=2 -> legacy argmax pooling, as OpenAI
        projection_dim=cfg["embed_dim"])
    return CLIPModel(config).eval()


def case_clip():
    out = {}
    # tiny two-tower model: logits + image-feature gradient
    model = _hf_model(CLIP_TINY)
    sd = seeded.clip_state_dict(**CLIP_TINY)
    _openai_to_hf(sd, model)
    img = seeded.tensor("clip.tiny.img", (3, 3, 224, 224), 0.5).requires_grad_(True)
    g = torch.Generator().manual_seed(3)
    tokens = torch.randint(3, 99, (2, CLIP_TINY["context_length"]), generator=g)
    tokens[0, 9] = 99
    tokens[1, 15] = 99  # EOT = highest id -> argmax pooling position
    res = model(input_ids=tokens, pixel_values=img)
    out["tiny.tokens"] = tokens
    out["tiny.logits_per_image"] = res.logits_per_image
    out["tiny.image_embeds"], out["tiny.text_embeds"] = res.image_embeds, res.text_embeds
    (gi,) = torch.autograd.grad(res.logits_per_image.sum(), img)
    out["tiny.grad_img_strided"] = gi[:, :, ::8, ::8]
    # full-size ViT-B/32 visual tower, random weights: projected features only
    cfg = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch=32,
               context_length=8, vocab_size=64, text_width=64, text_layers=1)
    model = _hf_model(cfg)
    sd = seeded.clip_state_dict(**cfg)
    _openai_to_hf(sd, model)
    img = seeded.tensor("clip.b32.img", (2, 3, 224, 224), 0.5)
    with torch.no_grad():
        out["b32.image_features"] = model.visual_projection(model.vision_model(pixel_values=img).pooler_output)
    _save("clip_hf", **out)


# ----------------------------------------------------------------------------- IR-SE50
def case_irse():
    _import_reference()
    from models.facial_recognition.model_irse import Backbone
    net = Backbone(input_size=112, num_layers=50, drop_ratio=0.6, mode="ir_se").eval()
    sd = seeded.irse_fill(net.state_dict())
    net.load_state_dict(sd, strict=True)
    x = seeded.tensor("irse.x", (2, 3, 112, 112), 0.5)
    with torch.no_grad():
        y = net(x)
    _save("irse", feats=y, keys=np.asarray(sorted(k for k in sd)), shapes=np.asarray([str(tuple(sd[k].shape)) for k in sorted(sd)]))


CASES = dict(ops=case_ops, modconv=case_modconv, generator16=case_generator16, generator_big=case_generator_big,
             mappers=case_mappers, step=case_step, ranger=case_ranger, preproc=case_preproc, clip=case_clip, irse=case_irse)

if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for name in (sys.argv[1:] or list(CASES)):
        print("==", name)
        CASES[name]()

"""Deterministic synthetic weights/inputs shared by the golden-vector generator
(tests/golden/make_golden.py, runs where /root/reference exists) and by the tests
(run anywhere).  No reference code involved: every tensor is
``randn(shape, seed=crc32(key))`` shaped by the checkpoint schema of SURVEY.md 3.5,
so a fixture only has to store inputs' *names* and the expected outputs.
"""
import math
import zlib

import torch


def tensor(key, shape, std=1.0, mean=0.0, salt=0):
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * salt) % (2 ** 31))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * std + mean


def fir_kernel(taps=(1, 3, 3, 1), gain=1.0):
    k = torch.tensor(taps, dtype=torch.float32)
    k = torch.outer(k, k)
    return k / k.sum() * gain


def generator_channels(res, channel_multiplier=2):
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
            256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}[res]


def generator_state_dict(size, style_dim=512, n_mlp=8, salt=0, exercise_noise=True):
    """rosinality g_ema schema with real-magnitude values: conv / modulation / style
    weights ~ N(0,1) (style weights are stored /lr_mul), modulation bias ~ 1,
    and -- unlike the reference init (zeros) -- non-zero noise strengths and
    activation biases so those code paths are exercised."""
    sd = {}
    nz = 0.1 if exercise_noise else 0.0

    def t(key, shape, std=1.0, mean=0.0):
        sd[key] = tensor(key, shape, std, mean, salt)

    for i in range(1, n_mlp + 1):
        t(f"style.{i}.weight", (style_dim, style_dim), std=1.0 / 0.01)
        t(f"style.{i}.bias", (style_dim,), std=10.0 * nz)
    c4 = generator_channels(4)
    t("input.input", (1, c4, 4, 4))

    def styled(pre, cin, cout, up):
        t(pre + ".conv.weight", (1, cout, cin, 3, 3))
        if up:
            sd[pre + ".conv.blur.kernel"] = fir_kernel(gain=4.0)
        t(pre + ".conv.modulation.weight", (cin, style_dim))
        t(pre + ".conv.modulation.bias", (cin,), std=0.05, mean=1.0)
        t(pre + ".noise.weight", (1,), std=nz)
        t(pre + ".activate.bias", (cout,), std=nz)

    def rgb(pre, cin, up):
        t(pre + ".bias", (1, 3, 1, 1), std=nz)
        if up:
            sd[pre + ".upsample.kernel"] = fir_kernel(gain=4.0)
        t(pre + ".conv.weight", (1, 3, cin, 1, 1))
        t(pre + ".conv.modulation.weight", (cin, style_dim))
        t(pre + ".conv.modulation.bias", (cin,), std=0.05, mean=1.0)

    styled("conv1", c4, c4, False)
    rgb("to_rgb1", c4, False)
    log_size = int(math.log2(size))
    cin = c4
    for j, i in enumerate(range(3, log_size + 1)):
        cout = generator_channels(2 ** i)
        styled(f"convs.{2 * j}", cin, cout, True)
        styled(f"convs.{2 * j + 1}", cout, cout, False)
        rgb(f"to_rgbs.{j}", cout, True)
        cin = cout
    for layer in range((log_size - 2) * 2 + 1):
        res = (layer + 5) // 2
        t(f"noises.noise_{layer}", (1, 1, 2 ** res, 2 ** res))
    return sd


def mapper_state_dict(prefixes, dims=None, salt=0):
    """`<prefix>mapping.{1..4}.{weight,bias}` per Mapper (latent_mappers.py:10-30)."""
    sd = {}
    for n, pre in enumerate(prefixes):
        d = 512 if dims is None else dims[n]
        for i in range(1, 5):
            sd[f"{pre}mapping.{i}.weight"] = tensor(f"{pre}mapping.{i}.weight", (d, d), std=1.0 / 0.01, salt=salt)
            sd[f"{pre}mapping.{i}.bias"] = tensor(f"{pre}mapping.{i}.bias", (d,), std=1.0, salt=salt)
    return sd


def clip_state_dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch=32,
                    context_length=77, vocab_size=49408, text_width=512, text_layers=12, salt=0, text=True):
    """OpenAI-CLIP checkpoint schema (ViT variant) with transformer-ish magnitudes."""
    sd = {}

    def t(key, shape, std=1.0, mean=0.0):
        sd[key] = tensor(key, shape, std, mean, salt)

    def blocks(pre, width, layers):
        for i in range(layers):
            p = f"{pre}.{i}"
            t(p + ".ln_1.weight", (width,), 0.1, 1.0)
            t(p + ".ln_1.bias", (width,), 0.1)
            t(p + ".attn.in_proj_weight", (3 * width, width), width ** -0.5)
            t(p + ".attn.in_proj_bias", (3 * width,), 0.1)
            t(p + ".attn.out_proj.weight", (width, width), width ** -0.5)
            t(p + ".attn.out_proj.bias", (width,), 0.1)
            t(p + ".ln_2.weight", (width,), 0.1, 1.0)
            t(p + ".ln_2.bias", (width,), 0.1)
            t(p + ".mlp.c_fc.weight", (4 * width, width), width ** -0.5)
            t(p + ".mlp.c_fc.bias", (4 * width,), 0.1)
            t(p + ".mlp.c_proj.weight", (width, 4 * width), (4 * width) ** -0.5)
            t(p + ".mlp.c_proj.bias", (width,), 0.1)

    grid = image_resolution // vision_patch
    t("visual.conv1.weight", (vision_width, 3, vision_patch, vision_patch), (3 * vision_patch ** 2) ** -0.5)
    t("visual.class_embedding", (vision_width,), vision_width ** -0.5)
    t("visual.positional_embedding", (grid * grid + 1, vision_width), vision_width ** -0.5)
    for n in ("visual.ln_pre", "visual.ln_post"):
        t(n + ".weight", (vision_width,), 0.1, 1.0)
        t(n + ".bias", (vision_width,), 0.1)
    blocks("visual.transformer.resblocks", vision_width, vision_layers)
    t("visual.proj", (vision_width, embed_dim), vision_width ** -0.5)
    if text:
        t("token_embedding.weight", (vocab_size, text_width), 0.02)
        t("positional_embedding", (context_length, text_width), 0.01)
        blocks("transformer.resblocks", text_width, text_layers)
        t("ln_final.weight", (text_width,), 0.1, 1.0)
        t("ln_final.bias", (text_width,), 0.1)
        t("text_projection", (text_width, embed_dim), text_width ** -0.5)
        sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    return sd


def wplus_latents(batch, n_latent, salt=0):
    """W+-like codes: a common w per image plus small per-layer jitter."""
    base = tensor("wplus.base", (batch, 1, 512), salt=salt)
    return base + 0.1 * tensor("wplus.jitter", (batch, n_latent, 512), salt=salt)


def sample_positions(numel, count, key):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) % (2 ** 31))
    return torch.randint(0, numel, (count,), generator=g)


def irse_fill(state_dict, salt=0):
    """Deterministic IR-SE50 weights keyed by the reference's state_dict names (used by make_golden.case_irse and
    the tests): BN statistics near (0,1), 1-D affine/PReLU weights near 1, biases near 0, conv/linear ~ fan_in^-0.5."""
    sd = {}
    for k, v in state_dict.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = v.clone()
        elif k.endswith("running_var"):
            sd[k] = tensor("irse." + k, v.shape, 0.1, 1.0, salt).abs() + 0.5
        elif k.endswith("running_mean"):
            sd[k] = tensor("irse." + k, v.shape, 0.1, 0.0, salt)
        elif v.ndim == 1:
            sd[k] = tensor("irse." + k, v.shape, 0.05, 1.0 if k.endswith("weight") else 0.0, salt)
        else:
            sd[k] = tensor("irse." + k, v.shape, v[0].numel() ** -0.5, 0.0, salt)
    return sd

"""Oracle (CPU, test-only): StyleGAN2 generator, functional over a rosinality
``g_ema`` state_dict (schema: SURVEY.md 3.5).

Restates models/stylegan2/model.py:179-574 and the feature-recording /
region-blend forward of attention/attention_model.py:473-676.  The modulated
convolution is written per sample with the materialised per-sample weight,
i.e. the reference's own arithmetic order (model.py:239-247), NOT the
shared-weight form the HIP kernels use -- so a comparison against this file
also bounds the rounding-order difference of that reformulation.
"""
import math
import random

import torch
import torch.nn.functional as F

from . import ops

BLUR_TAPS = (1, 3, 3, 1)


def num_layers(size):
    return (int(math.log2(size)) - 2) * 2 + 1


def n_latent(size):
    return int(math.log2(size)) * 2 - 2


def channels(size_px, channel_multiplier=2):
    """model.py:392-402"""
    table = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
             128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
             512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
    return table[size_px]


def style_mlp(sd, z):
    """PixelNorm + n_mlp x EqualLinear(lr_mul=0.01, fused lrelu) (model.py:381-390)."""
    x = ops.pixel_norm(z, dim=1)
    i = 1
    while f"style.{i}.weight" in sd:
        x = ops.equal_linear(x, sd[f"style.{i}.weight"], sd[f"style.{i}.bias"], lr_mul=0.01, activation=True)
        i += 1
    return x


def mean_latent(sd, n, generator=None):
    """model.py:451-457"""
    z = torch.randn(n, sd["style.1.weight"].shape[1], generator=generator)
    return style_mlp(sd, z).mean(0, keepdim=True)


def modulated_conv2d(x, style, weight, mod_w, mod_b, *, demodulate=True, upsample=False,
                     input_is_stylespace=False, blur_kernel=None):
    """ModulatedConv2d.forward (model.py:234-276), upsample / same-resolution
    branches (the downsample branch has no caller on the path).

    weight [1,Cout,Cin,k,k]; returns (out, s) with s the post-affine style
    [B,1,Cin,1,1] (Q8)."""
    b, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    if not input_is_stylespace:
        s = ops.equal_linear(style, mod_w, mod_b).view(b, 1, cin, 1, 1)  # bias_init=1 lives in mod_b
    else:
        s = style
    scale = 1.0 / math.sqrt(cin * k * k)
    wmod = scale * weight * s  # [B,Cout,Cin,k,k]
    if demodulate:
        d = torch.rsqrt(wmod.pow(2).sum([2, 3, 4]) + 1e-8)
        wmod = wmod * d.view(b, cout, 1, 1, 1)
    outs = []
    for n in range(b):
        if upsample:
            # conv_transpose2d wants [Cin,Cout,k,k] (model.py:254-257)
            y = F.conv_transpose2d(x[n:n + 1], wmod[n].transpose(0, 1), stride=2, padding=0)
        else:
            y = F.conv2d(x[n:n + 1], wmod[n], padding=k // 2)
        outs.append(y)
    out = torch.cat(outs, 0)
    if upsample:
        # Blur(pad=(1,1), kernel*4) for k=3, 4-tap FIR, factor 2 (model.py:199-206,260)
        factor = 2
        kt = blur_kernel.shape[0]
        p = (kt - factor) - (k - 1)
        pad = ((p + 1) // 2 + factor - 1, p // 2 + 1)
        out = ops.upfirdn2d(out, blur_kernel, pad=pad)
    return out, s


def styled_conv(sd, pre, x, style, noise, *, upsample, input_is_stylespace):
    """StyledConv.forward (model.py:334-340): modconv -> +w*noise -> fused lrelu."""
    bk = sd.get(pre + ".conv.blur.kernel")
    out, s = modulated_conv2d(x, style, sd[pre + ".conv.weight"], sd[pre + ".conv.modulation.weight"],
                              sd[pre + ".conv.modulation.bias"], demodulate=True, upsample=upsample,
                              input_is_stylespace=input_is_stylespace, blur_kernel=bk)
    if noise is None:
        noise = torch.randn(out.shape[0], 1, out.shape[2], out.shape[3])  # model.py:286-288
    out = out + sd[pre + ".noise.weight"] * noise
    out = ops.fused_leaky_relu(out, sd[pre + ".activate.bias"])
    return out, s


def to_rgb(sd, pre, x, style, skip, *, input_is_stylespace):
    """ToRGB.forward (model.py:353-362): 1x1 modconv (no demod) + bias + upsampled skip."""
    out, s = modulated_conv2d(x, style, sd[pre + ".conv.weight"], sd[pre + ".conv.modulation.weight"],
                              sd[pre + ".conv.modulation.bias"], demodulate=False, upsample=False,
                              input_is_stylespace=input_is_stylespace)
    out = out + sd[pre + ".bias"]
    if skip is not None:
        k = sd[pre + ".upsample.kernel"]  # make_kernel(blur)*4, pad (2,1) (model.py:35-43)
        out = out + ops.upfirdn2d(skip, k, up=2, down=1, pad=(2, 1))
    return out, s


def layer_plan(size):
    """(kind, state-dict prefix, W+ index, upsample, noise index) in execution order
    (model.py:527-566: conv1/to_rgb1 then [conv_up, conv, to_rgb] per octave; W+ index
    advances by 2 per octave so to_rgb(n) and conv_up(n+1) share one -- Q9)."""
    plan = [("conv", "conv1", 0, False, 0), ("rgb", "to_rgb1", 1, False, None)]
    i = 1
    for j in range(int(math.log2(size)) - 2):
        plan.append(("conv", f"convs.{2 * j}", i, True, 1 + 2 * j))
        plan.append(("conv", f"convs.{2 * j + 1}", i + 1, False, 2 + 2 * j))
        plan.append(("rgb", f"to_rgbs.{j}", i + 2, False, None))
        i += 2
    return plan


def generator_forward(sd, styles, *, size, return_latents=False, return_features=False, inject_index=None,
                      truncation=1, truncation_latent=None, input_is_latent=False, input_is_stylespace=False,
                      noise=None, randomize_noise=True, attention_layer=0, attention_map=None, feature_map=None):
    """Generator.forward of both generator files.  With attention_map=None and
    return_features=False this is models/stylegan2/model.py:473-574; the extra
    arguments follow attention/attention_model.py:473-676."""
    if not input_is_latent and not input_is_stylespace:
        styles = [style_mlp(sd, s) for s in styles]
    nl = num_layers(size)
    if noise is None:
        noise = [None] * nl if randomize_noise else [sd[f"noises.noise_{i}"] for i in range(nl)]
    if truncation < 1 and not input_is_stylespace:
        styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
    nlat = n_latent(size)
    if input_is_stylespace:
        latent = styles[0]
    elif len(styles) < 2:
        latent = styles[0].unsqueeze(1).repeat(1, nlat, 1) if styles[0].ndim < 3 else styles[0]
    else:
        if inject_index is None:
            inject_index = random.randint(1, nlat - 1)
        latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                            styles[1].unsqueeze(1).repeat(1, nlat - inject_index, 1)], 1)

    batch = latent[0].shape[0] if input_is_stylespace else latent.shape[0]
    out = sd["input.input"].repeat(batch, 1, 1, 1)  # ConstantInput (model.py:299-303)
    skip = None
    style_vector = []
    recorded = []
    armed = False  # `this_layer` of attention_model.py:532
    for n, (kind, pre, widx, up, nidx) in enumerate(layer_plan(size)):
        sty = latent[n] if input_is_stylespace else latent[:, widx]
        if kind == "conv":
            out, s = styled_conv(sd, pre, out, sty, noise[nidx], upsample=up,
                                 input_is_stylespace=input_is_stylespace)
            cur = out
        else:
            skip, s = to_rgb(sd, pre, out, sty, skip, input_is_stylespace=input_is_stylespace)
            cur = skip
        if attention_map is not None:
            layer = n + 1
            hit = layer == attention_layer or (kind == "rgb" and armed)
            if hit:
                armed = kind == "conv"
                cur = ops.mask_blend(cur, feature_map[layer - 1], attention_map)
                if kind == "conv":
                    out = cur
                else:
                    skip = cur
        recorded.append(cur)
        style_vector.append(s)

    image = skip
    if return_latents:
        return image, latent, style_vector
    if return_features:
        return image, latent, style_vector, recorded
    return image, None

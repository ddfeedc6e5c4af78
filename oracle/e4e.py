"""Oracle (CPU, test-only): the pSp / e4e inversion encoders (models/encoders/psp_encoders.py:34-200,
models/encoders/helpers.py:123-140), functional over the reference's state_dict, on top of oracle/irse.py's bottleneck
units.  Eval mode.  Pinned by tests/golden/e4e.npz (captured from the reference's own classes)."""
import math

import torch
import torch.nn.functional as F

from . import irse, ops


def _body(sd, x):
    """input_layer + the 24 IR-SE units, returning the FPN taps after units 6, 20 and 23 (psp_encoders.py:173-183)."""
    x = F.prelu(irse._bn(sd, "input_layer.1", F.conv2d(x, sd["input_layer.0.weight"], padding=1)), sd["input_layer.2.weight"])
    taps = {}
    for i, (cin, depth, stride) in enumerate(irse.blocks()):
        x = irse._unit(sd, f"body.{i}", x, cin, depth, stride)
        if i in (6, 20, 23):
            taps[i] = x
    return taps[6], taps[20], taps[23]


def gradual_style_block(sd, pre, x, spatial):
    """GradualStyleBlock (:34-55): log2(spatial) x [Conv2d(.,512,3,2,1) + LeakyReLU(0.01)], flatten, EqualLinear(lr_mul=1)."""
    for j in range(int(math.log2(spatial))):
        x = F.leaky_relu(F.conv2d(x, sd[f"{pre}.convs.{2 * j}.weight"], sd[f"{pre}.convs.{2 * j}.bias"], stride=2, padding=1), 0.01)
    return ops.equal_linear(x.view(-1, 512), sd[f"{pre}.linear.weight"], sd[f"{pre}.linear.bias"], lr_mul=1)


def _upsample_add(x, y):
    """helpers.py:123-140"""
    return F.interpolate(x, size=y.shape[2:], mode="bilinear", align_corners=True) + y


def _spatial(i):
    return 16 if i < 3 else (32 if i < 7 else 64)  # coarse_ind = 3, middle_ind = 7 (:149-159)


def encoder4editing(sd, x, style_count=18, stage=18):
    """Encoder4Editing.forward (:173-200): w0 from the coarse features, repeated; deltas from the FPN levels."""
    c1, c2, c3 = _body(sd, x)
    w0 = gradual_style_block(sd, "styles.0", c3, 16)
    w = w0.repeat(style_count, 1, 1).permute(1, 0, 2).clone()
    features = c3
    for i in range(1, min(stage + 1, style_count)):
        if i == 3:
            p2 = _upsample_add(c3, F.conv2d(c2, sd["latlayer1.weight"], sd["latlayer1.bias"]))
            features = p2
        elif i == 7:
            p1 = _upsample_add(p2, F.conv2d(c1, sd["latlayer2.weight"], sd["latlayer2.bias"]))
            features = p1
        w[:, i] = w[:, i] + gradual_style_block(sd, f"styles.{i}", features, _spatial(i))
    return w


def gradual_style_encoder(sd, x, style_count=18):
    """GradualStyleEncoder.forward (:96-121), the pSp encoder."""
    c1, c2, c3 = _body(sd, x)
    lat = [gradual_style_block(sd, f"styles.{j}", c3, 16) for j in range(3)]
    p2 = _upsample_add(c3, F.conv2d(c2, sd["latlayer1.weight"], sd["latlayer1.bias"]))
    lat += [gradual_style_block(sd, f"styles.{j}", p2, 32) for j in range(3, 7)]
    p1 = _upsample_add(p2, F.conv2d(c1, sd["latlayer2.weight"], sd["latlayer2.bias"]))
    lat += [gradual_style_block(sd, f"styles.{j}", p1, 64) for j in range(7, style_count)]
    return torch.stack(lat, dim=1)

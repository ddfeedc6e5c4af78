"""Oracle (CPU, test-only): StyleCLIP latent mappers, functional over the mapper
state_dict (mapper/latent_mappers.py:10-128)."""
import torch

from . import ops

STYLESPACE_DIMENSIONS = [512] * 15 + [256] * 3 + [128] * 3 + [64] * 3 + [32] * 2  # latent_mappers.py:7


def mapper_mlp(sd, prefix, x):
    """Mapper.forward (latent_mappers.py:10-30): PixelNorm() with its default dim=1
    -- on a [B,n,512] group that is the *layer* axis, not the feature axis (Q2) --
    then 4 x EqualLinear(lr_mul=0.01, fused lrelu).  Keys `<prefix>mapping.{1..4}.*`."""
    x = ops.pixel_norm(x, dim=1)
    for i in range(1, 5):
        x = ops.equal_linear(x, sd[f"{prefix}mapping.{i}.weight"], sd[f"{prefix}mapping.{i}.bias"],
                             lr_mul=0.01, activation=True)
    return x


def single_mapper(sd, x):
    """SingleMapper (latent_mappers.py:33-44)."""
    return mapper_mlp(sd, "mapping.", x)


def levels_mapper(sd, x, no_coarse=False, no_medium=False, no_fine=False):
    """LevelsMapper.forward (latent_mappers.py:61-82); note the reference's
    attribute spelling `course_mapping`."""
    parts = []
    for name, sl, off in (("course_mapping.", slice(0, 4), no_coarse), ("medium_mapping.", slice(4, 8), no_medium),
                          ("fine_mapping.", slice(8, None), no_fine)):
        g = x[:, sl, :]
        parts.append(torch.zeros_like(g) if off else mapper_mlp(sd, name, g))
    return torch.cat(parts, dim=1)


def full_stylespace_mapper(sd, xs):
    """FullStyleSpaceMapper.forward (latent_mappers.py:93-101)."""
    return [mapper_mlp(sd, f"mapper_{c}.", x.view(x.shape[0], -1)).view(x.shape) for c, x in enumerate(xs)]


def without_torgb_stylespace_mapper(sd, xs):
    """WithoutToRGBStyleSpaceMapper.forward (latent_mappers.py:104-128): entries
    1,4,7,... (the ToRGB styles) get a zero delta."""
    torgb = set(range(1, len(STYLESPACE_DIMENSIONS), 3))
    out = []
    for c, x in enumerate(xs):
        if c in torgb:
            out.append(torch.zeros_like(x))
        else:
            out.append(mapper_mlp(sd, f"mapper_{c}.", x.view(x.shape[0], -1)).view(x.shape))
    return out

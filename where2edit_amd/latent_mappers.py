"""mapper/latent_mappers.py surface (same class names, `opts` fields, state_dict keys, Q2 PixelNorm axis).
The MLPs are [B*n,512]x[512,512] GEMMs -- rocBLAS through torch -- with the fused
bias+LeakyReLU on the HIP op; these are the trainable parameters whose gradients are all-reduced."""
import os

import torch
from torch import nn
from torch.nn import Module

from .stylegan2 import EqualLinear, PixelNorm

STYLESPACE_DIMENSIONS = [512 for _ in range(15)] + [256, 256, 256] + [128, 128, 128] + [64, 64, 64] + [32, 32]


class _ScaleParams(torch.autograd.Function):
    """(w_i * w_scale ..., b_i * b_scale ...) for a list of EqualLinear layers that share scale and lr_mul: ONE multi-tensor
    kernel per group in each direction instead of two elementwise kernels per layer per direction (model.py:151-158 does the
    products per layer; the values are the same)."""

    @staticmethod
    def forward(ctx, w_scale, b_scale, n, *params):
        ctx.cfg = (w_scale, b_scale, n)
        ws = torch._foreach_mul(list(params[:n]), w_scale)
        bs = torch._foreach_mul(list(params[n:]), b_scale)
        return (*ws, *bs)

    @staticmethod
    def backward(ctx, *grads):
        w_scale, b_scale, n = ctx.cfg
        out = [None] * len(grads)
        for lo, hi, sc in ((0, n, w_scale), (n, len(grads), b_scale)):
            idx = [i for i in range(lo, hi) if grads[i] is not None]
            if idx:
                for i, g in zip(idx, torch._foreach_mul([grads[i] for i in idx], sc)):
                    out[i] = g
        return (None, None, None, *out)


def scaled_parameters(mappers):
    """{EqualLinear: (weight*scale, bias*lr_mul)} for every EqualLinear of `mappers`, or None when the layers do not share
    one scale / lr_mul (the style-space mappers: per-layer widths) or nothing needs a gradient."""
    if os.environ.get("W2E_TUNE_NO_PSCALE"):  # tuning aid: the per-layer products
        return None
    layers = [m for mp in mappers for m in mp.mapping if isinstance(m, EqualLinear)]
    if not layers or any(m.bias is None or m.scale != layers[0].scale or m.lr_mul != layers[0].lr_mul for m in layers):
        return None
    if not (torch.is_grad_enabled() and any(m.weight.requires_grad for m in layers)):
        return None  # frozen / no_grad: EqualLinear's own cached products
    outs = _ScaleParams.apply(layers[0].scale, layers[0].lr_mul, len(layers), *[m.weight for m in layers], *[m.bias for m in layers])
    return {m: (outs[i], outs[len(layers) + i]) for i, m in enumerate(layers)}


class Mapper(Module):
    """latent_mappers.py:10-30.  PixelNorm() keeps the reference's default dim=1: on a [B,n,512] group
    that normalises over the layer axis n (Q2) -- reproduced on purpose."""

    def __init__(self, opts, latent_dim=512):
        super().__init__()
        self.opts = opts
        layers = [PixelNorm()]
        for _ in range(4):
            layers.append(EqualLinear(latent_dim, latent_dim, lr_mul=0.01, activation="fused_lrelu"))
        self.mapping = nn.Sequential(*layers)

    def forward(self, x, scaled=None):
        if scaled is None:
            scaled = scaled_parameters([self])
        if scaled is None:
            return self.mapping(x)
        for m in self.mapping:
            x = m(x, scaled[m]) if isinstance(m, EqualLinear) else m(x)
        return x


class SingleMapper(Module):
    """latent_mappers.py:33-44"""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.mapping = Mapper(opts)

    def forward(self, x):
        if x.is_cuda and x.ndim == 3 and not os.environ.get("W2E_MAPPER_STOCK"):
            from . import mapper_hip  # one Mapper over all latents = LevelsMapper's kernels with a single level
            out = mapper_hip.levels_mlp(x, [(self.mapping, 0, x.shape[1])])
            if out is not None:
                return out
        return self.mapping(x)


class LevelsMapper(Module):
    """latent_mappers.py:47-82 (attribute `course_mapping` spelled as in the reference: checkpoint keys)."""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        if not opts.no_coarse_mapper:
            self.course_mapping = Mapper(opts)
        if not opts.no_medium_mapper:
            self.medium_mapping = Mapper(opts)
        if not opts.no_fine_mapper:
            self.fine_mapping = Mapper(opts)

    def forward(self, x):
        if x.is_cuda and x.ndim == 3 and x.shape[1] > 8 and not os.environ.get("W2E_MAPPER_STOCK"):
            # the three MLPs as one node on the library's mapper kernels (forward, weight and bias gradients): mapper_hip.py
            from . import mapper_hip
            levels = [(getattr(self, n), l0, ln) for n, off, l0, ln in (("course_mapping", self.opts.no_coarse_mapper, 0, 4),
                                                                         ("medium_mapping", self.opts.no_medium_mapper, 4, 4),
                                                                         ("fine_mapping", self.opts.no_fine_mapper, 8, x.shape[1] - 8)) if not off]
            out = mapper_hip.levels_mlp(x, levels) if levels else None
            if out is not None:
                return out
        x_coarse, x_medium, x_fine = x[:, :4, :], x[:, 4:8, :], x[:, 8:, :]
        active = [getattr(self, n) for n, off in (("course_mapping", self.opts.no_coarse_mapper), ("medium_mapping", self.opts.no_medium_mapper),
                                                  ("fine_mapping", self.opts.no_fine_mapper)) if not off]
        sc = scaled_parameters(active) if active else None  # the weight*scale / bias*lr_mul products of all levels at once
        x_coarse = self.course_mapping(x_coarse, sc) if not self.opts.no_coarse_mapper else torch.zeros_like(x_coarse)
        x_medium = self.medium_mapping(x_medium, sc) if not self.opts.no_medium_mapper else torch.zeros_like(x_medium)
        x_fine = self.fine_mapping(x_fine, sc) if not self.opts.no_fine_mapper else torch.zeros_like(x_fine)
        return torch.cat([x_coarse, x_medium, x_fine], dim=1)


class FullStyleSpaceMapper(Module):
    """latent_mappers.py:84-101"""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        for c, c_dim in enumerate(STYLESPACE_DIMENSIONS):
            setattr(self, f"mapper_{c}", Mapper(opts, latent_dim=c_dim))

    def forward(self, x):
        if x and x[0].is_cuda and not os.environ.get("W2E_MAPPER_STOCK"):
            from . import mapper_hip  # the 26 MLPs as one node on the library's style-space mapper kernels (w2e_ssmapper_*)
            out = mapper_hip.stylespace_mlp(list(x), [getattr(self, f"mapper_{c}") for c in range(len(x))])
            if out is not None:
                return out
        out = []
        for c, x_c in enumerate(x):
            out.append(getattr(self, f"mapper_{c}")(x_c.view(x_c.shape[0], -1)).view(x_c.shape))
        return out


class WithoutToRGBStyleSpaceMapper(Module):
    """latent_mappers.py:104-128"""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        indices_without_torgb = list(range(1, len(STYLESPACE_DIMENSIONS), 3))
        self.STYLESPACE_INDICES_WITHOUT_TORGB = [i for i in range(len(STYLESPACE_DIMENSIONS))
                                                 if i not in indices_without_torgb]
        for c in self.STYLESPACE_INDICES_WITHOUT_TORGB:
            setattr(self, f"mapper_{c}", Mapper(opts, latent_dim=STYLESPACE_DIMENSIONS[c]))

    def forward(self, x):
        if x and x[0].is_cuda and not os.environ.get("W2E_MAPPER_STOCK"):
            from . import mapper_hip  # the 17 MLPs as one node (w2e_ssmapper_*); the ToRGB codes map to zeros
            idx = self.STYLESPACE_INDICES_WITHOUT_TORGB
            mapped = mapper_hip.stylespace_mlp([x[c] for c in idx], [getattr(self, f"mapper_{c}") for c in idx])
            if mapped is not None:
                by_c = dict(zip(idx, mapped))
                return [by_c[c] if c in by_c else torch.zeros_like(x[c]) for c in range(len(STYLESPACE_DIMENSIONS))]
        out = []
        for c in range(len(STYLESPACE_DIMENSIONS)):
            x_c = x[c]
            if c in self.STYLESPACE_INDICES_WITHOUT_TORGB:
                out.append(getattr(self, f"mapper_{c}")(x_c.view(x_c.shape[0], -1)).view(x_c.shape))
            else:
                out.append(torch.zeros_like(x_c))
        return out

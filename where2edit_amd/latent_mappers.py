"""mapper/latent_mappers.py surface (same class names, `opts` fields, state_dict keys, Q2 PixelNorm axis).
The MLPs are [B*n,512]x[512,512] GEMMs -- rocBLAS through torch -- with the fused
bias+LeakyReLU on the HIP op; these are the trainable parameters whose gradients are all-reduced."""
import torch
from torch import nn
from torch.nn import Module

from .stylegan2 import EqualLinear, PixelNorm

STYLESPACE_DIMENSIONS = [512 for _ in range(15)] + [256, 256, 256] + [128, 128, 128] + [64, 64, 64] + [32, 32]


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

    def forward(self, x):
        return self.mapping(x)


class SingleMapper(Module):
    """latent_mappers.py:33-44"""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.mapping = Mapper(opts)

    def forward(self, x):
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
        x_coarse, x_medium, x_fine = x[:, :4, :], x[:, 4:8, :], x[:, 8:, :]
        x_coarse = self.course_mapping(x_coarse) if not self.opts.no_coarse_mapper else torch.zeros_like(x_coarse)
        x_medium = self.medium_mapping(x_medium) if not self.opts.no_medium_mapper else torch.zeros_like(x_medium)
        x_fine = self.fine_mapping(x_fine) if not self.opts.no_fine_mapper else torch.zeros_like(x_fine)
        return torch.cat([x_coarse, x_medium, x_fine], dim=1)


class FullStyleSpaceMapper(Module):
    """latent_mappers.py:84-101"""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        for c, c_dim in enumerate(STYLESPACE_DIMENSIONS):
            setattr(self, f"mapper_{c}", Mapper(opts, latent_dim=c_dim))

    def forward(self, x):
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
        out = []
        for c in range(len(STYLESPACE_DIMENSIONS)):
            x_c = x[c]
            if c in self.STYLESPACE_INDICES_WITHOUT_TORGB:
                out.append(getattr(self, f"mapper_{c}")(x_c.view(x_c.shape[0], -1)).view(x_c.shape))
            else:
                out.append(torch.zeros_like(x_c))
        return out

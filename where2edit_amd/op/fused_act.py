"""models/stylegan2/op/fused_act.py surface: `FusedLeakyReLU(channel)` with its `.bias` Parameter and
`fused_leaky_relu(input, bias, negative_slope=0.2, scale=sqrt 2)` -- one HIP kernel instead of
add + leaky_relu + mul, and without the reference's hard-coded `input.cuda()` (fused_act.py:25)."""
import torch
from torch import nn

from ..functional import fused_leaky_relu


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


__all__ = ["FusedLeakyReLU", "fused_leaky_relu"]

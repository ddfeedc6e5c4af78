"""The reference's operator seam, `models/stylegan2/op/__init__.py:1-2`: exactly these three names."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d"]

"""models/stylegan2/op/upfirdn2d.py surface: `upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))`."""
from ..functional import upfirdn2d

__all__ = ["upfirdn2d"]

"""Oracle (CPU, test-only) for the L0 ops of the hot path.

Each function names the reference lines it restates (paths relative to
/root/reference).  Stock torch ops only; differentiable through torch autograd.
"""
import math

import torch
import torch.nn.functional as F


def make_kernel(taps):
    """FIR taps -> normalised 2-D kernel (models/stylegan2/model.py:20-28)."""
    k = torch.as_tensor(taps, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Zero-stuff by `up`, pad/crop, true 2-D convolution with `kernel`, keep
    every `down`-th sample (models/stylegan2/op/upfirdn2d.py:11-60).

    x [N,C,H,W], kernel [kh,kw].  Output size per axis:
    (H*up + pad0 + pad1 - kh)//down + 1.
    """
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    if up > 1:
        z = x.new_zeros(n, c, h * up, w * up)
        z[:, :, ::up, ::up] = x  # sample, then up-1 zeros (op/upfirdn2d.py:29-31)
    else:
        z = x
    z = F.pad(z, (p0, p1, p0, p1))  # negative pad = crop (op/upfirdn2d.py:33-41)
    wk = torch.flip(kernel, (0, 1)).to(x.dtype).reshape(1, 1, kh, kw).expand(c, 1, kh, kw)
    y = F.conv2d(z, wk, groups=c)  # correlation with the flipped kernel (op/upfirdn2d.py:47-48)
    return y[:, :, ::down, ::down]


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=math.sqrt(2.0)):
    """lrelu(x + bias) * scale; bias on dim 1, except 3-D inputs where it sits on
    the last dim (models/stylegan2/op/fused_act.py:23-39).  No device moves (Q1).
    """
    if x.ndim == 3:
        b = bias.view(1, 1, -1)
    else:
        b = bias.view(1, -1, *([1] * (x.ndim - 2)))
    return F.leaky_relu(x + b, negative_slope) * scale


def pixel_norm(x, dim=1):
    """models/stylegan2/model.py:11-17."""
    return x * torch.rsqrt(torch.mean(x * x, dim=dim, keepdim=True) + 1e-8)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """EqualLinear.forward (models/stylegan2/model.py:149-159): weight is stored
    divided by lr_mul, applied as weight*scale with scale = lr_mul/sqrt(in);
    bias enters as bias*lr_mul; optional fused lrelu*sqrt2."""
    scale = lr_mul / math.sqrt(weight.shape[1])
    y = F.linear(x, weight * scale)
    if bias is not None:
        if activation:
            return fused_leaky_relu(y, bias * lr_mul)
        y = y + bias * lr_mul
    elif activation:
        raise ValueError("activation needs a bias (fused_act.py:23)")
    return y


def clip_preprocess(img, stylegan_size=None):
    """Upsample(scale_factor=7, nearest) then AvgPool2d(stylegan_size // 32)
    (criteria/clip_loss.py:11-12,15).  Literal chain -- materialises the 7x image."""
    size = img.shape[-1] if stylegan_size is None else stylegan_size
    up = F.interpolate(img, scale_factor=7, mode="nearest")
    return F.avg_pool2d(up, kernel_size=size // 32)


def mask_blend(new, old, mask):
    """Region-attention blend (attention/attention_model.py:548-549): nearest
    resize of mask [B,1,s,s] to the feature size, broadcast over channels,
    m*new + (1-m)*old."""
    m = F.interpolate(mask, size=new.shape[-1])  # default mode: nearest (Q7)
    m = m.expand(-1, new.shape[1], -1, -1)
    return m * new + (1 - m) * old


def id_preprocess(img):
    """IDLoss.extract_feats front (criteria/id_loss.py:19-23): pool to 256 unless
    already 256, crop [35:223, 32:220], adaptive-pool to 112."""
    if img.shape[2] != 256:
        img = F.adaptive_avg_pool2d(img, (256, 256))
    img = img[:, :, 35:223, 32:220]
    return F.adaptive_avg_pool2d(img, (112, 112))

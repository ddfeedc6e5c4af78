"""models/stylegan2/model.py surface on the MI355X kernels.

Same class names, constructor arguments, forward keyword arguments, return arities and state_dict keys
as the reference (models/stylegan2/model.py:11-574), so `from where2edit_amd.stylegan2 import Generator`
drops in where the reference does `from models.stylegan2.model import Generator` and loads the same
rosinality `g_ema` checkpoints.  The arithmetic is different by design:

  * ModulatedConv2d never materialises the per-sample weight [B,Cout,Cin,k,k] (model.py:239-247); it
    runs y = demod[b,o] * conv(scale*W, s[b,i]*x) on fp32 MFMA with one packed weight for the batch;
  * StyledConv is ONE kernel for conv + noise + bias + LeakyReLU*sqrt2 (3 extra passes in the reference),
    plus the FIR blur for the up-sampling layers;
  * ToRGB is ONE kernel for the 1x1 modconv + bias + FIR-upsampled skip.
"""
import math
import os
import random

import torch
from torch import nn
from torch.nn import functional as F

from . import functional as K
from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d


def invalidate_caches(module):
    """Drop every derived-weight cache under `module` (packed conv weights, sum_k W^2, scaled EqualLinear / ToRGB weights,
    the stacked style affines).  The caches are keyed on (data_ptr, Tensor._version): optimizer steps, load_state_dict and
    any in-place op bump the version and rebuild them by themselves, but writes through `.data` (the reference Ranger's
    `p.data.copy_`, rosinality's EMA `accumulate`) do NOT -- call this after such writes."""
    for m in module.modules():
        for attr in ("_cache_key", "_scaled_key", "_wsc_key", "_style_pack_key"):
            if hasattr(m, attr):
                setattr(m, attr, None)
        if hasattr(m, "_text_cache"):
            m._text_cache = None


def _trainable_weight(mod):
    """Refuses a conv weight that autograd expects a gradient for.  The packed-weight kernels (K1) compute none: the path this package
    accelerates never optimises the decoder (mapper/training/coach.py:174-180 hands only net.mapper's parameters to the optimizer), and a
    second, stock-op backend inside ModulatedConv2d for that case (rounds 2-3 ran the reference's per-sample-weight grouped convolution on
    MIOpen here) is exactly the silent fallback this package does not have.  One kernel path, or an error that says what to do."""
    if torch.is_grad_enabled() and mod.weight.requires_grad:
        raise RuntimeError(
            "where2edit_amd: a ModulatedConv2d weight requires grad (decoder fine-tuning).  The HIP kernels treat the decoder as frozen -- "
            "as the path they serve does (coach.py:174-180 optimises net.mapper only) -- and produce no conv-weight gradient; there is no "
            "stock-op fallback.  Call stylegan2.freeze_conv_weights(decoder) (or decoder.requires_grad_(False)); gradients to latents, "
            "styles, noise strengths and biases are unaffected.")
    return False


def freeze_conv_weights(module):
    """requires_grad_(False) on exactly the parameters this implementation cannot differentiate: the 3x3 (and generic 1x1)
    ModulatedConv2d weights consumed through the packed-weight kernels.  Everything else stays as it was."""
    for m in module.modules():
        if isinstance(m, ModulatedConv2d):
            m.weight.requires_grad_(False)
    return module


class PixelNorm(nn.Module):
    """model.py:11-17 ([B,512]-sized: stock torch ops)."""

    def __init__(self, dim=1):
        super().__init__()
        self.dim = dim

    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input ** 2, dim=self.dim, keepdim=True) + 1e-8)


def make_kernel(k):
    """model.py:20-28"""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    k /= k.sum()
    return k


class Upsample(nn.Module):
    """model.py:31-49"""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        kernel = make_kernel(kernel) * (factor ** 2)
        self.register_buffer("kernel", kernel)
        p = kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    """model.py:52-70"""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        kernel = make_kernel(kernel)
        self.register_buffer("kernel", kernel)
        p = kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    """model.py:73-89"""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module):
    """model.py:92-127.  Only the (unused) Discriminator instantiates it in the reference; kept for the
    import surface, on stock ops."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        return F.conv2d(input, self.weight * self.scale, bias=self.bias, stride=self.stride, padding=self.padding)

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]},"
                f" {self.weight.shape[2]}, stride={self.stride}, padding={self.padding})")


class EqualLinear(nn.Module):
    """model.py:130-164.  [B,512]x[512,C] GEMMs: rocBLAS through torch, the fused
    bias+lrelu through the HIP op."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def _scaled(self):
        """weight*scale and bias*lr_mul (model.py:151-158).  For frozen parameters (the decoder on the mapper path)
        the products are cached until the parameters change instead of being recomputed by two kernels per call."""
        w, b = self.weight, self.bias
        if torch.is_grad_enabled() and (w.requires_grad or (b is not None and b.requires_grad)):
            return w * self.scale, (b * self.lr_mul if b is not None else None)
        key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version))
        if getattr(self, "_scaled_key", None) != key:
            with torch.no_grad():
                self._scaled_val = ((w * self.scale).detach(), (b * self.lr_mul).detach() if b is not None else None)
            self._scaled_key = key
        return self._scaled_val

    def forward(self, input, scaled=None):
        """`scaled` = (weight*scale, bias*lr_mul) computed by the caller (latent_mappers batches those products over all
        of a mapper's layers); default: this layer's own."""
        w, b = scaled if scaled is not None else self._scaled()
        if self.activation:
            return fused_leaky_relu(F.linear(input, w), b)
        return F.linear(input, w, bias=b)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})"


class ScaledLeakyReLU(nn.Module):
    """model.py:167-176"""

    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * math.sqrt(2)


class ModulatedConv2d(nn.Module):
    """model.py:179-276: same parameters (`weight [1,Cout,Cin,k,k]`, `modulation.*`, `blur.kernel`) and the
    same `(out, style)` return, where `style` is the post-affine [B,1,Cin,1,1] tensor (the S-space code)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        fan_in = in_channel * kernel_size ** 2
        self.scale = 1 / math.sqrt(fan_in)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self._cache_key = None
        self._cache = None

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, "
                f"upsample={self.upsample}, downsample={self.downsample})")

    # ---- frozen-weight derived tensors, rebuilt whenever the parameter changes (in-place updates bump _version)
    def _derived(self):
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._cache_key != key:
            with torch.no_grad():
                w4 = w.detach()[0].to(torch.float32)
                if self.kernel_size == 1:
                    w9 = torch.zeros(self.out_channel, self.in_channel, 3, 3, device=w.device)
                    w9[:, :, 1, 1] = w4[:, :, 0, 0]  # 1x1 on the 3x3 engine (generic Cout; ToRGB has its own kernel)
                else:
                    w9 = w4.contiguous()
                fwd = K.conv_pack(w9, self.scale, transpose=False, flip=False)
                bwd = K.conv_pack(w9, self.scale, transpose=True, flip=not self.upsample)
                wsq = (w4 * self.scale).square().sum((2, 3)).contiguous()  # [Cout,Cin]: sum_k (scale*W)^2
            self._cache = (fwd, bwd, wsq)
            self._cache_key = key
        return self._cache

    def _style(self, style, batch, input_is_stylespace):
        if not input_is_stylespace:
            style = self.modulation(style).view(batch, 1, self.in_channel, 1, 1)
        return style

    def forward(self, input, style, input_is_stylespace=False):
        if self.kernel_size not in (1, 3):
            raise NotImplementedError("ModulatedConv2d kernels exist for kernel_size 1 and 3 (the sizes the generator uses)")
        batch, in_channel, height, width = input.shape
        style = self._style(style, batch, input_is_stylespace)
        if _trainable_weight(self):
            return _modconv_trainable(self, input, style), style
        s2d = style.reshape(batch, in_channel)
        fwd, bwd, wsq = self._derived()
        if not self.demodulate:
            wsq = None  # demod[b,o] = rsqrt(s^2 @ wsq^T + eps) is computed inside the kernels (model.py:241-243)
        if self.downsample:
            if torch.is_grad_enabled() and (input.requires_grad or style.requires_grad):
                raise NotImplementedError("down-sampling ModulatedConv2d (no caller in the generator) is forward-only")
            x = self.blur(input)
            d = K.demod_coefficients(s2d.contiguous(), wsq) if wsq is not None else None
            out = K.modconv_down_plain(x, s2d, d, fwd, (x.shape[2] - 1) // 2, (x.shape[3] - 1) // 2)
        else:
            if self.upsample and (self.kernel_size != 3 or tuple(self.blur.kernel.shape) != (4, 4)):
                raise NotImplementedError("up-sampling ModulatedConv2d: kernel_size 3 with a 4-tap blur")
            out = K.modconv(input, s2d, wsq, (fwd, bwd), self.blur.kernel if self.upsample else None, self.upsample)
        return out, style


class NoiseInjection(nn.Module):
    """model.py:279-290 (standalone form; inside StyledConv the add is fused into the conv epilogue)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    """model.py:293-303"""

    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    """model.py:306-340: conv -> noise -> fused lrelu, here a single fused launch (two for upsample)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None, input_is_stylespace=False, demod=None):
        conv = self.conv
        batch = input.shape[0]
        fusable = (conv.kernel_size == 3 and not conv.downsample and noise is not None and noise.ndim == 4
                   and noise.shape[0] == 1 and noise.shape[1] == 1 and not _trainable_weight(conv)
                   and not (conv.upsample and tuple(conv.blur.kernel.shape) != (4, 4)))
        self._act_noise = None  # (set below when the fused epilogue produced the output: what a following ToRGB may fold, see _synthesis)
        if not fusable:
            # per-sample / random noise (randomize_noise=True) or a 1x1 StyledConv: unfused composition of the same ops
            out, style = conv(input, style, input_is_stylespace=input_is_stylespace)
            out = self.noise(out, noise=noise)
            return self.activate(out), style
        style = conv._style(style, batch, input_is_stylespace)
        s2d = style.reshape(batch, conv.in_channel)
        fwd, bwd, wsq = conv._derived()
        noise_c = noise.contiguous()
        link = K.ActLink(noise_c) if (torch.is_grad_enabled() and not conv.upsample) else None
        out = K.styled_conv(input, s2d, wsq if conv.demodulate else None, noise_c, self.noise.weight, self.activate.bias, (fwd, bwd),
                            conv.blur.kernel if conv.upsample else None, conv.upsample, link=link,
                            demod=demod if conv.demodulate else None)
        self._act_noise = link
        return out, style


_NO_RGBPASS = bool(os.environ.get("W2E_TUNE_NO_RGBPASS"))  # tuning aid: autograd's own accumulation instead
_NO_RGBACT = os.environ.get("W2E_TUNE_NO_RGBACT") is not None  # tuning aid: the activation backward as its own pass


class ToRGB(nn.Module):
    """model.py:343-362"""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None, input_is_stylespace=False, passthrough=False, producer_act=None):
        """`passthrough` (synthesis loop, training): returns (rgb, style, input) -- see functional._ToRGB.  `producer_act`: the
        ActLink of the fused StyledConv whose output `input` is, when this ToRGB is its only consumer: the ToRGB backward then
        applies that layer's activation backward to the gradient it returns (w2e_torgb_bwd_actbwd)."""
        conv = self.conv
        batch = input.shape[0]
        style = conv._style(style, batch, input_is_stylespace)
        # wmod[b,c,i] = scale * W[c,i] * s[b,i]  (model.py:239 with k=1, demodulate=False)
        w = conv.weight
        st = None
        if torch.is_grad_enabled() and w.requires_grad:
            wmod = (conv.scale * w.view(1, 3, conv.in_channel)) * style.reshape(batch, 1, conv.in_channel)
        else:  # frozen decoder: scale*W is cached until the parameter changes, and the kernels apply the style themselves
            key = (w.data_ptr(), w._version)
            if getattr(self, "_wsc_key", None) != key:
                with torch.no_grad():
                    self._wsc = (conv.scale * w.detach().view(3, conv.in_channel)).contiguous()
                self._wsc_key = key
            wmod, st = self._wsc, style.reshape(batch, conv.in_channel)
        fuse_skip = skip is not None and self.upsample.kernel.shape == (4, 4) and self.upsample.factor == 2
        passed = None
        if passthrough:
            out, passed = K.to_rgb(input, wmod, self.bias, skip if fuse_skip else None, self.upsample.kernel if fuse_skip else None,
                                   True, style=st, producer_act=None if _NO_RGBACT else producer_act)
        else:
            out = K.to_rgb(input, wmod, self.bias, skip if fuse_skip else None, self.upsample.kernel if fuse_skip else None,
                           style=st, producer_act=None if _NO_RGBACT else producer_act)
        if skip is not None and not fuse_skip:
            out = out + self.upsample(skip)
        if passthrough:
            return out, style, passed
        return out, style


class Generator(nn.Module):
    """model.py:365-574: same constructor, buffers, state_dict keys and forward contract
    (`(image, None)`, or `(image, latent, style_vector)` with return_latents -- Q10)."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01):
        super().__init__()
        self.size = size
        self.style_dim = style_dim
        layers = [PixelNorm()]
        for _ in range(n_mlp):
            layers.append(EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu"))
        self.style = nn.Sequential(*layers)
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier,
                         128: 128 * channel_multiplier, 256: 64 * channel_multiplier,
                         512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        in_channel = self.channels[4]
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2

    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 2 ** 2, 2 ** 2, device=device)]
        for i in range(3, self.log_size + 1):
            for _ in range(2):
                noises.append(torch.randn(1, 1, 2 ** i, 2 ** i, device=device))
        return noises

    def mean_latent(self, n_latent):
        latent_in = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(latent_in).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    # ---- shared front end of both Generator.forward variants (model.py:485-525)
    def _prepare(self, styles, inject_index, truncation, truncation_latent, input_is_latent, input_is_stylespace, noise,
                 randomize_noise):
        if not input_is_latent and not input_is_stylespace:
            styles = [self.style(s) for s in styles]
        if noise is None:
            if randomize_noise:
                noise = [None] * self.num_layers
            else:
                noise = [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]
        if truncation < 1 and not input_is_stylespace:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if input_is_stylespace:
            latent = styles[0]
        elif len(styles) < 2:
            latent = styles[0].unsqueeze(1).repeat(1, self.n_latent, 1) if styles[0].ndim < 3 else styles[0]
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)
        return latent, noise

    def _layers(self):
        """(module, is_rgb, W+ index, noise index) in execution order; W+ index advances by 2 per octave
        (to_rgb(n) and conv_up(n+1) share one), the S-space index by 1 per layer (model.py:527-566, Q9)."""
        plan = [(self.conv1, False, 0, 0), (self.to_rgb1, True, 1, None)]
        i = 1
        for j in range(len(self.to_rgbs)):
            plan.append((self.convs[2 * j], False, i, 1 + 2 * j))
            plan.append((self.convs[2 * j + 1], False, i + 1, 2 + 2 * j))
            plan.append((self.to_rgbs[j], True, i + 2, None))
            i += 2
        return plan

    def _style_pack(self, plan):
        """All modulation affines stacked for w2e_style_affine_* (rebuilt when any of their parameters changes)."""
        mods = [m.conv.modulation for m, _, _, _ in plan]
        key = tuple((m.weight.data_ptr(), m.weight._version, m.bias.data_ptr(), m.bias._version) for m in mods)
        if getattr(self, "_style_pack_key", None) != key:
            with torch.no_grad():
                w = torch.cat([m.weight.detach().float() * m.scale for m in mods]).contiguous()
                b = torch.cat([m.bias.detach().float() * m.lr_mul for m in mods]).contiguous()
                meta, off = [], 0
                for (_, _, widx, _), m in zip(plan, mods):
                    cw = m.weight.shape[0]
                    meta.append(torch.stack([torch.full((cw,), widx), torch.full((cw,), off), torch.full((cw,), cw),
                                             torch.arange(cw)], 1))
                    off += cw
                meta = torch.cat(meta).to(device=w.device, dtype=torch.int32).contiguous()
            self._style_pack_val = (w, b, meta, [m.weight.shape[0] for m in mods])
            self._style_pack_key = key
        return self._style_pack_val

    def _batched_styles(self, latent, plan):
        """The 26 per-layer style vectors from one launch, when the modulation layers are frozen (they are on every
        path of SURVEY 8: the decoder is never optimised) and the input is a W+ tensor; None = per-layer path."""
        if not (torch.is_tensor(latent) and latent.ndim == 3 and latent.is_cuda):
            return None
        mods = [m.conv.modulation for m, _, _, _ in plan]
        if any(m.bias is None or m.weight.shape[0] % 32 for m in mods):
            return None
        if torch.is_grad_enabled() and any(m.weight.requires_grad or m.bias.requires_grad for m in mods):
            return None
        return K.style_affine_all(latent, self._style_pack(plan))

    def style_codes(self, styles, inject_index=None, truncation=1, truncation_latent=None, input_is_latent=False):
        """(latent, the 26 S-space codes [B,1,C,1,1]) -- the second and third values of forward(..., return_latents=True)
        (model.py:527-574) WITHOUT running the synthesis network: for callers that invert to W+ and only need the codes
        (show_demo/try_demo.py:99-101 runs the whole generator for them and drops the image)."""
        latent, _ = self._prepare(styles, inject_index, truncation, truncation_latent, input_is_latent, False, None, False)
        plan = self._layers()
        batch = latent.shape[0]
        batched = self._batched_styles(latent, plan)
        if batched is not None:
            return latent, [s.view(batch, 1, s.shape[1], 1, 1) for s in batched]
        return latent, [mod.conv._style(latent[:, widx], batch, False) for mod, _, widx, _ in plan]

    def _synthesis(self, latent, noise, input_is_stylespace, on_layer=None, hook_layers=None):
        """`on_layer(n, is_rgb, act) -> act` is called after layer n of the plan (every layer, or only those in `hook_layers`);
        layers without a hook take the fused training forms (ToRGB pass-through, activation backward inside the ToRGB backward)."""
        if not torch.is_grad_enabled():
            return self._synthesis_pass(latent, noise, input_is_stylespace, on_layer, hook_layers)
        # the [B,C]-sized gradient accumulators of this pass's backward nodes come out of one zero-filled buffer (K.GradPool)
        batch = (latent[0] if input_is_stylespace else latent).shape[0]
        widths = sum(m.conv.in_channel for m, _, _, _ in self._layers())
        with K.grad_pool(batch * (3 * widths + 64 * (len(self._layers()) + 2) + self.n_latent * self.style_dim)):
            return self._synthesis_pass(latent, noise, input_is_stylespace, on_layer, hook_layers)

    def _synthesis_pass(self, latent, noise, input_is_stylespace, on_layer, hook_layers):
        hook_all = on_layer
        batch_ref = latent[0] if input_is_stylespace else latent
        out = self.input(batch_ref)
        skip = None
        style_vector = []
        plan = self._layers()
        batched = None if input_is_stylespace else self._batched_styles(latent, plan)
        if batched is not None:
            batch = latent.shape[0]
            latent = [s.view(batch, 1, s.shape[1], 1, 1) for s in batched]
            input_is_stylespace = True
        demods = {}
        if batched is not None and batched[0].is_cuda:  # every layer's style is known: all demodulation vectors in one launch
            idx = [n for n, (m, is_rgb, _, _) in enumerate(plan) if not is_rgb and m.conv.demodulate and m.conv.kernel_size == 3]
            if idx:
                demods = dict(zip(idx, K.demod_coefficients_all([batched[n] for n in idx], [plan[n][0].conv._derived()[2] for n in idx])))
        producer = None  # the fused-epilogue record of the StyledConv whose output `out` currently is
        for n, (mod, is_rgb, widx, nidx) in enumerate(plan):
            sty = latent[n] if input_is_stylespace else latent[:, widx]
            on_layer = hook_all if (hook_layers is None or n in hook_layers) else None
            if is_rgb:
                # the activation feeds this ToRGB and the next conv: route it THROUGH the ToRGB node when gradients flow
                # (one consumer, the two gradients are joined inside torgb_bwd instead of by an elementwise add -- and the
                # producing StyledConv's activation backward is applied there too)
                if on_layer is None and n + 1 < len(plan) and torch.is_grad_enabled() and out.requires_grad and not _NO_RGBPASS:
                    skip, s, out = mod(out, sty, skip, input_is_stylespace=input_is_stylespace, passthrough=True, producer_act=producer)
                elif on_layer is None and n + 1 == len(plan) and torch.is_grad_enabled() and out.requires_grad and not _NO_RGBPASS:
                    skip, s = mod(out, sty, skip, input_is_stylespace=input_is_stylespace, producer_act=producer)  # (the last layer: sole consumer too)
                else:
                    skip, s = mod(out, sty, skip, input_is_stylespace=input_is_stylespace)
                if on_layer is not None:
                    skip = on_layer(n, True, skip)
                producer = None
            else:
                out, s = mod(out, sty, noise=noise[nidx], input_is_stylespace=input_is_stylespace, demod=demods.get(n))
                producer = getattr(mod, "_act_noise", None)
                if on_layer is not None:
                    out = on_layer(n, False, out)
                    producer = None  # (the hook may have replaced the activation)
            style_vector.append(s)
        return skip, style_vector

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, input_is_stylespace=False, noise=None, randomize_noise=True):
        latent, noise = self._prepare(styles, inject_index, truncation, truncation_latent, input_is_latent,
                                      input_is_stylespace, noise, randomize_noise)
        image, style_vector = self._synthesis(latent, noise, input_is_stylespace)
        if return_latents:
            return image, latent, style_vector
        return image, None

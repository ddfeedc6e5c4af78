"""models/encoders/psp_encoders.py surface: the pSp / e4e GAN-inversion encoders (`GradualStyleBlock`,
`GradualStyleEncoder`, `Encoder4Editing`; same constructor arguments and state_dict keys, so e4e checkpoints'
`encoder.*` tensors load strict) -- the front of BASELINE configs[4] (show_demo/try_demo.py:93-100,
utils.py:622-636).

The body is IR-SE50's (models/encoders/helpers.py = models/facial_recognition/helpers.py): on the GPU, in eval mode with
frozen weights, it runs on the kernels of irse_hip.py; the map2style stacks (stride-2 3x3 conv + bias + LeakyReLU 0.01)
and the 1x1 lateral convolutions run on the same conv engine (w2e_conv3x3: DOWN with padding, centre-tap packs).
Inference only on that path (the encoder is frozen everywhere the reference uses it); CPU / train mode: stock modules."""
import math
import types
from enum import Enum

import numpy as np
import torch
from torch import nn
from torch.nn import BatchNorm2d, Conv2d, Module, PReLU, Sequential
from torch.nn import functional as F

from .id_loss import bottleneck_IR, bottleneck_IR_SE, get_blocks
from .stylegan2 import EqualLinear


class ProgressiveStage(Enum):
    """psp_encoders.py:12-31"""
    WTraining = 0
    Delta1Training = 1
    Delta2Training = 2
    Delta3Training = 3
    Delta4Training = 4
    Delta5Training = 5
    Delta6Training = 6
    Delta7Training = 7
    Delta8Training = 8
    Delta9Training = 9
    Delta10Training = 10
    Delta11Training = 11
    Delta12Training = 12
    Delta13Training = 13
    Delta14Training = 14
    Delta15Training = 15
    Delta16Training = 16
    Delta17Training = 17
    Inference = 18


def _upsample_add(x, y):
    """models/encoders/helpers.py:123-140.  Inference on the GPU: one kernel (w2e_upsample_add; the stock bilinear op takes 1.4 ms
    for an [8,512,64,64] output on this stack, the kernel the 30 us the bytes take)."""
    _, _, h, w = y.size()
    if x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and not (torch.is_grad_enabled() and (x.requires_grad or y.requires_grad)):
        from ._lib import call, ptr, stream_ptr
        xc, yc = x.contiguous(), y.contiguous()
        out = torch.empty_like(yc)
        call("w2e_upsample_add", ptr(xc), ptr(yc), ptr(out), xc.shape[0] * xc.shape[1], xc.shape[2], xc.shape[3], h, w, stream_ptr())
        return out
    return F.interpolate(x, size=(h, w), mode="bilinear", align_corners=True) + y


class GradualStyleBlock(Module):
    """psp_encoders.py:34-55"""

    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c = out_c
        self.spatial = spatial
        num_pools = int(np.log2(spatial))
        modules = [Conv2d(in_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        for _ in range(num_pools - 1):
            modules += [Conv2d(out_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        self.convs = nn.Sequential(*modules)
        self.linear = EqualLinear(out_c, out_c, lr_mul=1)

    def _hip_ok(self, x):
        """Inference on the GPU: the stride-2 convolutions run on the MFMA conv engine (forward only)."""
        return x.is_cuda and x.dtype == torch.float32 and (not torch.is_grad_enabled() or not (
            x.requires_grad or any(p.requires_grad for p in self.parameters())))

    def _packs(self):
        from . import functional as K
        key = tuple((p.data_ptr(), p._version) for p in self.convs.parameters())
        if getattr(self, "_pack_key", None) != key:
            with torch.no_grad():
                self._pack = [(K.conv_pack(m.weight.detach().float(), 1.0, False, False), m.bias.detach().float().contiguous(),
                               torch.full((m.weight.shape[0],), m_act.negative_slope, device=m.weight.device))
                              for m, m_act in zip(self.convs[0::2], self.convs[1::2])]
            self._pack_key = key
        return self._pack

    def forward(self, x):
        if self._hip_ok(x):
            from . import functional as K
            from . import irse_hip
            with torch.no_grad():
                for wp, bias, slope in self._packs():
                    h, w = x.shape[2] // 2, x.shape[3] // 2
                    x = irse_hip.conv3x3(x.contiguous(), wp, self.out_c, h, w, mode=K.MODE_DOWN, down_pad=1, bias=bias, slope=slope)
        else:
            x = self.convs(x)
        x = x.view(-1, self.out_c)
        return self.linear(x)


class _EncoderBase(Module):
    def __init__(self, num_layers, mode, opts):
        super().__init__()
        assert num_layers in [50, 100, 152], "num_layers should be 50,100, or 152"
        assert mode in ["ir", "ir_se"], "mode should be ir or ir_se"
        unit_module = bottleneck_IR if mode == "ir" else bottleneck_IR_SE
        self.input_layer = Sequential(Conv2d(3, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        self.body = Sequential(*[unit_module(b.in_channel, b.depth, b.stride) for blk in get_blocks(num_layers) for b in blk])
        self.styles = nn.ModuleList()
        log_size = int(math.log(opts.stylegan_size, 2))
        self.style_count = 2 * log_size - 2
        self.coarse_ind = 3
        self.middle_ind = 7
        for i in range(self.style_count):
            self.styles.append(GradualStyleBlock(512, 512, 16 if i < self.coarse_ind else (32 if i < self.middle_ind else 64)))
        self.latlayer1 = nn.Conv2d(256, 512, kernel_size=1, stride=1, padding=0)
        self.latlayer2 = nn.Conv2d(128, 512, kernel_size=1, stride=1, padding=0)
        self._mode = mode

    def _hip(self, x):
        return (x.is_cuda and x.dtype == torch.float32 and not self.training and self._mode == "ir_se"
                and not any(p.requires_grad for n, p in self.named_parameters() if n.startswith(("input_layer", "body"))))

    def _taps(self, x):
        """c1, c2, c3 = the body's outputs after units 6, 20, 23 (psp_encoders.py:176-183)."""
        if self._hip(x):
            from . import irse_hip
            key = tuple((p.data_ptr(), p._version) for n, p in self.named_parameters() if n.startswith(("input_layer", "body")))
            if getattr(self, "_plan_key", None) != key:
                plan = types.SimpleNamespace()
                with torch.no_grad():
                    il = self.input_layer
                    a, b = irse_hip._bn_affine(il[1])
                    w0 = il[0].weight.detach().float()
                    from . import functional as K
                    plan.input = {"wf": K.conv_pack(w0, 1.0, False, False), "wb": K.conv_pack(w0, 1.0, True, True), "a": a, "b": b,
                                  "slope": il[2].weight.detach().float().contiguous()}
                    specs = [b_ for blk in get_blocks(len_to_layers(len(self.body))) for b_ in blk]
                    plan.units = [irse_hip.UnitPlan(u, s.in_channel, s.depth, s.stride) for u, s in zip(self.body, specs)]
                self._plan, self._plan_key = plan, key
            y = irse_hip._InputLayer.apply(x, self._plan.input, None)
            taps = {}
            for i, u in enumerate(self._plan.units):
                y = irse_hip._IRUnit.apply(y, u, None)
                if i in (6, 20, 23):
                    taps[i] = y
            return taps[6], taps[20], taps[23]
        x = self.input_layer(x)
        taps = {}
        for i, l in enumerate(self.body):
            x = l(x)
            if i in (6, 20, 23):
                taps[i] = x
        return taps[6], taps[20], taps[23]

    def _lateral(self, conv, x):
        """nn.Conv2d(C, 512, 1) with bias: the centre tap of the 3x3 engine on the GPU inference path."""
        if x.is_cuda and not torch.is_grad_enabled():
            from . import functional as K
            from . import irse_hip
            key = (conv.weight.data_ptr(), conv.weight._version, conv.bias.data_ptr(), conv.bias._version)
            cache = self.__dict__.setdefault("_lat", {})
            if cache.get(id(conv), (None,))[0] != key:
                w9 = torch.zeros(conv.weight.shape[0], conv.weight.shape[1], 3, 3, device=x.device)
                w9[:, :, 1, 1] = conv.weight.detach()[:, :, 0, 0]
                cache[id(conv)] = (key, K.conv_pack(w9, 1.0, False, False), conv.bias.detach().float().contiguous())
            _, wp, bias = cache[id(conv)]
            return irse_hip.conv3x3(x.contiguous(), wp, conv.weight.shape[0], x.shape[2], x.shape[3], bias=bias)
        return conv(x)


def len_to_layers(n_units):
    return {24: 50, 49: 100, 50: 152}.get(n_units, 50)


class GradualStyleEncoder(_EncoderBase):
    """psp_encoders.py:58-121 (the pSp encoder)."""

    def __init__(self, num_layers, mode="ir", opts=None):
        super().__init__(num_layers, mode, opts)

    def forward(self, x):
        c1, c2, c3 = self._taps(x)
        latents = [self.styles[j](c3) for j in range(self.coarse_ind)]
        p2 = _upsample_add(c3, self._lateral(self.latlayer1, c2))
        latents += [self.styles[j](p2) for j in range(self.coarse_ind, self.middle_ind)]
        p1 = _upsample_add(p2, self._lateral(self.latlayer2, c1))
        latents += [self.styles[j](p1) for j in range(self.middle_ind, self.style_count)]
        return torch.stack(latents, dim=1)


class Encoder4Editing(_EncoderBase):
    """psp_encoders.py:124-200 (e4e)."""

    def __init__(self, num_layers, mode="ir", opts=None):
        super().__init__(num_layers, mode, opts)
        self.progressive_stage = ProgressiveStage.Inference

    def get_deltas_starting_dimensions(self):
        return list(range(self.style_count))

    def set_progressive_stage(self, new_stage):
        self.progressive_stage = new_stage
        print("Changed progressive stage to: ", new_stage)

    def forward(self, x):
        c1, c2, c3 = self._taps(x)
        w0 = self.styles[0](c3)
        w = w0.repeat(self.style_count, 1, 1).permute(1, 0, 2)
        stage = self.progressive_stage.value
        features = c3
        for i in range(1, min(stage + 1, self.style_count)):
            if i == self.coarse_ind:
                p2 = _upsample_add(c3, self._lateral(self.latlayer1, c2))
                features = p2
            elif i == self.middle_ind:
                p1 = _upsample_add(p2, self._lateral(self.latlayer2, c1))
                features = p1
            w[:, i] += self.styles[i](features)
        return w


def load_e4e_standalone(checkpoint_path, device="cuda"):
    """utils.py:622-636: an e4e checkpoint {'opts', 'state_dict' (encoder.* / decoder.*), 'latent_avg'} -> the encoder with
    the latent-average forward hook."""
    import argparse
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    opts = argparse.Namespace(**ckpt["opts"])
    e4e = Encoder4Editing(50, "ir_se", opts)
    e4e.load_state_dict({k.replace("encoder.", ""): v for k, v in ckpt["state_dict"].items() if k.startswith("encoder.")})
    e4e.eval().requires_grad_(False)
    e4e = e4e.to(device)
    latent_avg = ckpt["latent_avg"].to(device)

    def add_latent_avg(model, inputs, outputs):
        return outputs + latent_avg.repeat(outputs.shape[0], 1, 1)

    e4e.register_forward_hook(add_latent_avg)
    return e4e

"""The latent mapper MLPs on libw2e.so (csrc/mapper.hip, K7): LevelsMapper's three Mappers (mapper/latent_mappers.py:10-82) as ONE
autograd node -- PixelNorm + 4 EqualLinear layers of all levels, one launch per layer and direction, weight and bias gradients
included (these are the parameters the step trains).  About 120 stock-op launches per step (rocBLAS GEMMs of 16-80 rows, bias /
activation, PixelNorm, reductions) become 13."""
import ctypes

import torch
from torch.autograd.function import once_differentiable

from ._lib import call, ptr, stream_ptr

DIM = 512
LAYERS = 4


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


def _int_array(values):
    return (ctypes.c_int * len(values))(*values)


class _LevelsMLP(torch.autograd.Function):
    """out[b, l0_g + l] = Mapper_g(x[b, l0_g : l0_g + len_g])[l] for every level g = (l0_g, len_g); latents outside every level: 0.
    `params`: LAYERS * G weights (layer-major: layer 0 of every group, then layer 1, ...) followed by the biases in the same order.
    Differentiable in the parameters only (the W+ latents of the training set carry no gradient, coach.py:79-84)."""

    @staticmethod
    def forward(ctx, x, levels, w_scale, b_scale, *params):
        x = x.contiguous()
        b, n_latent, d = x.shape
        g = len(levels)
        l0, ln = _int_array([lv[0] for lv in levels]), _int_array([lv[1] for lv in levels])
        rows = b * sum(lv[1] for lv in levels)
        weights, biases = params[:LAYERS * g], params[LAYERS * g:]
        dev = x.device
        h = [torch.empty((rows, DIM), device=dev, dtype=torch.float32) for _ in range(LAYERS)]  # h[0] = PixelNorm(x), h[j] = layer j's output
        covered = sum(lv[1] for lv in levels) == n_latent
        out = (torch.empty if covered else torch.zeros)((b, n_latent, DIM), device=dev, dtype=torch.float32)
        st = stream_ptr()
        call("w2e_mapper_pixelnorm", ptr(x), ptr(h[0]), b, n_latent, g, l0, ln, st)
        for j in range(LAYERS):
            last = j == LAYERS - 1
            call("w2e_mapper_linear", 0, ptr(h[j]), None, ptr(out if last else h[j + 1]), _ptr_array(weights[j * g:(j + 1) * g]),
                 _ptr_array(biases[j * g:(j + 1) * g]), b, n_latent, g, l0, ln, float(w_scale), float(b_scale), int(last), st)
        ctx.save_for_backward(out, *h, *weights)
        ctx.geom = (levels, w_scale, b_scale, b, n_latent)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        levels, w_scale, b_scale, b, n_latent = ctx.geom
        g = len(levels)
        saved = ctx.saved_tensors
        out, h, weights = saved[0], saved[1:1 + LAYERS], saved[1 + LAYERS:]
        l0, ln = _int_array([lv[0] for lv in levels]), _int_array([lv[1] for lv in levels])
        dev = out.device
        rows = h[0].shape[0]
        gout = gout.contiguous()
        st = stream_ptr()
        gw = torch.empty((LAYERS * g, DIM, DIM), device=dev, dtype=torch.float32)
        gb = torch.empty((LAYERS * g, DIM), device=dev, dtype=torch.float32)
        # transposed weights of layers 1..3 (the input gradient of layer 0 is not needed)
        wt = torch.empty(((LAYERS - 1) * g, DIM, DIM), device=dev, dtype=torch.float32)
        call("w2e_mapper_transpose", _ptr_array(weights[g:]), (LAYERS - 1) * g, ptr(wt), st)
        gy, y, gathered = gout, out, 0  # the last layer's output and its gradient are [B, n_latent, 512]
        for j in range(LAYERS - 1, -1, -1):
            call("w2e_mapper_wgrad", ptr(gy), ptr(y), ptr(h[j]), _ptr_array([gw[j * g + i] for i in range(g)]),
                 _ptr_array([gb[j * g + i] for i in range(g)]), b, n_latent, g, l0, ln, float(w_scale), float(b_scale), gathered, st)
            if j == 0:
                break
            if not gathered:  # group-major copies of the last layer's (gradient, output) for the input-gradient product
                ga, ya = torch.empty((rows, DIM), device=dev, dtype=torch.float32), torch.empty((rows, DIM), device=dev, dtype=torch.float32)
                call("w2e_mapper_gather", ptr(gy), ptr(ga), b, n_latent, g, l0, ln, st)
                call("w2e_mapper_gather", ptr(y), ptr(ya), b, n_latent, g, l0, ln, st)
                gy, y, gathered = ga, ya, 1
            gh = torch.empty((rows, DIM), device=dev, dtype=torch.float32)
            call("w2e_mapper_linear", 1, ptr(gy), ptr(y), ptr(gh), _ptr_array([wt[(j - 1) * g + i] for i in range(g)]), None, b, n_latent, g, l0, ln,
                 float(w_scale), 0.0, 0, st)
            gy, y = gh, h[j]
        grads = [gw[i] for i in range(LAYERS * g)] + [gb[i] for i in range(LAYERS * g)]
        return (None, None, None, None, *grads)


def levels_mlp(x, mappers_and_ranges):
    """mappers_and_ranges: [(Mapper, l0, len), ...] -> the [B, n_latent, 512] output of LevelsMapper.forward, or None when this path
    does not apply (CPU tensors, a latent that requires grad, a Mapper that is not PixelNorm + 4 x EqualLinear(512, 512, fused_lrelu)
    with one shared scale / lr_mul): the caller then composes the stock modules."""
    from .stylegan2 import EqualLinear, PixelNorm
    if not (torch.is_tensor(x) and x.is_cuda and x.ndim == 3 and x.shape[-1] == DIM and x.dtype == torch.float32) or x.requires_grad:
        return None
    if not mappers_and_ranges or len(mappers_and_ranges) > 4:
        return None
    layers = []
    for mp, _, ln in mappers_and_ranges:
        seq = list(mp.mapping)
        lin = seq[1:]
        if not (isinstance(seq[0], PixelNorm) and seq[0].dim == 1 and len(lin) == LAYERS and all(isinstance(m, EqualLinear) for m in lin)):
            return None
        if any(m.weight.shape != (DIM, DIM) or m.bias is None or m.activation != "fused_lrelu" or not m.weight.is_cuda for m in lin):
            return None
        if ln * x.shape[0] > 1152:
            return None
        layers.append(lin)
    first = layers[0][0]
    if any(m.scale != first.scale or m.lr_mul != first.lr_mul for lin in layers for m in lin):
        return None
    weights = [layers[gi][j].weight for j in range(LAYERS) for gi in range(len(layers))]
    biases = [layers[gi][j].bias for j in range(LAYERS) for gi in range(len(layers))]
    levels = tuple((l0, ln) for _, l0, ln in mappers_and_ranges)
    return _LevelsMLP.apply(x, levels, first.scale, first.lr_mul, *weights, *biases)


# ---- the style-space mappers (latent_mappers.py:84-128): one Mapper per S-space code, width = that code's channel count ----------
def _float_array(values):
    return (ctypes.c_float * len(values))(*values)


class _StyleSpaceMLP(torch.autograd.Function):
    """out_c = Mapper_c(x_c) for G codes x_c [B, C_c] (PixelNorm over the features + 4 EqualLinear(C_c, C_c, fused_lrelu)), every
    layer of ALL codes one launch per direction (w2e_ssmapper_*).  `params`: LAYERS * G weights (layer-major), then the biases in
    the same order.  Returns G tensors [B, C_c] (views of one packed buffer).  Differentiable in the parameters only."""

    @staticmethod
    def forward(ctx, dims, w_scales, b_scale, n_codes, *args):
        xs, params = args[:n_codes], args[n_codes:]
        g = n_codes
        b = xs[0].shape[0]
        dev = xs[0].device
        total = b * sum(dims)
        dims_c, ws_c = _int_array(dims), _float_array(w_scales)
        weights, biases = params[:LAYERS * g], params[LAYERS * g:]
        h = [torch.empty(total, device=dev, dtype=torch.float32) for _ in range(LAYERS + 1)]  # h[0] = PixelNorm(x), h[j+1] = layer j's output
        st = stream_ptr()
        xs = [x.contiguous() for x in xs]
        call("w2e_ssmapper_pixelnorm", _ptr_array(xs), ptr(h[0]), b, g, dims_c, st)
        for j in range(LAYERS):
            call("w2e_ssmapper_linear", 0, ptr(h[j]), None, ptr(h[j + 1]), _ptr_array(weights[j * g:(j + 1) * g]),
                 _ptr_array(biases[j * g:(j + 1) * g]), b, g, dims_c, ws_c, float(b_scale), st)
        ctx.save_for_backward(*h, *weights)
        ctx.geom = (tuple(dims), tuple(w_scales), b_scale, b)
        outs, off = [], 0
        for d in dims:
            outs.append(h[LAYERS][off:off + b * d].view(b, d))
            off += b * d
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        dims, w_scales, b_scale, b = ctx.geom
        g = len(dims)
        saved = ctx.saved_tensors
        h, weights = saved[:LAYERS + 1], saved[LAYERS + 1:]
        dev = h[0].device
        dims_c, ws_c = _int_array(dims), _float_array(w_scales)
        st = stream_ptr()
        gsrc = [go.contiguous() if go is not None else None for go in gouts]
        gy = torch.empty_like(h[0])
        call("w2e_ssmapper_gather", _ptr_array(gsrc), ptr(gy), b, g, dims_c, st)
        gw = [torch.empty((d, d), device=dev, dtype=torch.float32) for _ in range(LAYERS) for d in dims]
        gb = [torch.empty((d,), device=dev, dtype=torch.float32) for _ in range(LAYERS) for d in dims]
        for j in range(LAYERS - 1, -1, -1):
            call("w2e_ssmapper_wgrad", ptr(gy), ptr(h[j + 1]), ptr(h[j]), _ptr_array(gw[j * g:(j + 1) * g]), _ptr_array(gb[j * g:(j + 1) * g]),
                 b, g, dims_c, ws_c, float(b_scale), st)
            if j == 0:
                break
            gx = torch.empty_like(gy)
            call("w2e_ssmapper_linear", 1, ptr(gy), ptr(h[j + 1]), ptr(gx), _ptr_array(weights[j * g:(j + 1) * g]), None, b, g, dims_c, ws_c,
                 0.0, st)
            gy = gx
        return (None, None, None, None, *([None] * g), *gw, *gb)


def stylespace_mlp(xs, mappers):
    """xs: the G code tensors (any shape with B leading and C_c elements per sample), mappers: their G `Mapper`s -> list of G outputs
    shaped like the inputs, or None when this path does not apply (CPU tensors, codes that require grad, batch > 16, more than 32
    codes, a Mapper that is not PixelNorm + 4 x EqualLinear(C, C, fused_lrelu) with one lr_mul): the caller then composes the stock
    modules."""
    from .stylegan2 import EqualLinear, PixelNorm
    if not xs or len(xs) != len(mappers) or len(xs) > 32:
        return None
    b = xs[0].shape[0]
    if b < 1 or b > 16:
        return None
    dims, layers = [], []
    for x, mp in zip(xs, mappers):
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.shape[0] == b) or x.requires_grad:
            return None
        seq = list(mp.mapping)
        lin = seq[1:]
        d = x.numel() // b
        if not (isinstance(seq[0], PixelNorm) and seq[0].dim == 1 and len(lin) == LAYERS and all(isinstance(m, EqualLinear) for m in lin)):
            return None
        if any(m.weight.shape != (d, d) or m.bias is None or m.activation != "fused_lrelu" or not m.weight.is_cuda for m in lin):
            return None
        dims.append(d)
        layers.append(lin)
    lr_mul = layers[0][0].lr_mul
    if any(m.lr_mul != lr_mul for lin in layers for m in lin) or any(lin[j].scale != lin[0].scale for lin in layers for j in range(LAYERS)):
        return None
    w_scales = [lin[0].scale for lin in layers]
    weights = [layers[gi][j].weight for j in range(LAYERS) for gi in range(len(layers))]
    biases = [layers[gi][j].bias for j in range(LAYERS) for gi in range(len(layers))]
    outs = _StyleSpaceMLP.apply(dims, w_scales, lr_mul, len(xs), *[x.reshape(b, -1) for x in xs], *weights, *biases)
    return [o.view(x.shape) for o, x in zip(outs, xs)]

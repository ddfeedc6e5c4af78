"""torch.autograd.Functions over the C ABI of libw2e.so (include/w2e.h).

PyTorch is plumbing here: it owns the device memory (outputs are torch.empty on the input's device),
the stream (kernels are enqueued on torch's current HIP stream) and the autograd tape.  All arithmetic
on image-sized tensors happens in the HIP kernels; torch ops are used only on [B,C]-sized style math.
"""
import ctypes
import math
import os

import torch
from torch.autograd.function import once_differentiable

from . import _lib, profiling
from ._lib import call, ptr, stream_ptr

SQRT2 = math.sqrt(2.0)

# ---- "the first n samples of the batch carry no gradient".  Coach runs x = G(w) (no grad) and x_hat = G(w_hat) as ONE
# generator pass over the batch [w; w_hat] (twice the rows per launch: the same kernels run ~9 % faster per image at twice
# the batch); the backward of every generator node then works on the rows [n:] only -- the saved activations and the incoming
# gradient are batch-major, so the slices are contiguous views -- and leaves rows [:n] of the gradients it returns
# unwritten: nothing reads them (each node's only producers / consumers are nodes that slice the same way).
_NOGRAD_PREFIX = 0
# Debug aid (W2E_DEBUG_POISON=1 / set_debug_poison): the prefix rows that the nodes above leave unwritten are filled with NaN,
# so that a consumer which DOES read them (a stock op slipped between `both` and a generator node, a node that forgot to slice)
# turns the loss / gradients into NaN instead of silently using uninitialised memory.  One fill per buffer: off by default.
_POISON = os.environ.get("W2E_DEBUG_POISON", "0") == "1"


def set_debug_poison(on=True):
    global _POISON
    _POISON = bool(on)


def _grad_rows(shape, n_skip, device, like=None):
    """An uninitialised full-batch gradient buffer whose rows [n_skip:] a kernel is about to write (rows [:n_skip] belong to the
    no-grad half of a merged pass: unwritten by design, NaN under the debug option)."""
    buf = torch.empty(shape, device=device, dtype=torch.float32) if like is None else torch.empty_like(like)
    if _POISON and n_skip:
        buf[:n_skip].fill_(float("nan"))
    return buf


class nograd_prefix:
    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        global _NOGRAD_PREFIX
        self.prev, _NOGRAD_PREFIX = _NOGRAD_PREFIX, self.n
        return self

    def __exit__(self, *exc):
        global _NOGRAD_PREFIX
        _NOGRAD_PREFIX = self.prev


class _TailRows(torch.autograd.Function):
    """x[n:] of a batch whose first n rows carry no gradient (the merged generator pass): the backward hands the producer a
    full-batch buffer whose tail holds the gradient and whose first n rows are unwritten -- every generator node slices them off
    (`nograd_prefix`) -- instead of autograd's zero-fill of the whole batch + copy (SliceBackward)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.full = n, x.shape[0]
        return x[n:]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        out = _grad_rows((ctx.full,) + tuple(g.shape[1:]), ctx.n, g.device)
        out[ctx.n:].copy_(g)
        return out, None


def tail_rows(x, n):
    return _TailRows.apply(x, n)


class GradPool:
    """Zeroed scratch for the [B,C]-sized gradient accumulators of ONE generator pass: its backward nodes (17 StyledConv style
    gradients, 9 ToRGB ones, the W+ gradient) carve their buffers out of one zero-filled tensor -- one fill launch instead of 27.
    Created per pass by `grad_pool` (the buffer itself on first use, in the backward), handed out once per region, never reused:
    nothing is shared between steps or between graph captures."""

    __slots__ = ("buf", "off", "hint")

    def __init__(self, hint):
        self.buf, self.off, self.hint = None, 0, int(hint)

    def zeros(self, shape, device):
        n = 1
        for v in shape:
            n *= int(v)
        n_al = (n + 63) & ~63  # 256-byte granules: float4 / b128 accesses of the consumers stay aligned
        if self.buf is None or self.off + n_al > self.buf.numel() or self.buf.device != torch.device(device):
            self.buf, self.off = torch.zeros(max(n_al, self.hint), device=device, dtype=torch.float32), 0
        out = self.buf[self.off:self.off + n].view(shape)
        self.off += n_al
        return out


_GRAD_POOL = None


class grad_pool:
    """`with grad_pool(hint_floats):` -- the autograd nodes created inside share one GradPool for their small zeroed gradients."""

    def __init__(self, hint):
        self.pool = GradPool(hint)

    def __enter__(self):
        global _GRAD_POOL
        self.prev, _GRAD_POOL = _GRAD_POOL, self.pool
        return self.pool

    def __exit__(self, *exc):
        global _GRAD_POOL
        _GRAD_POOL = self.prev
        return False


def _zeros_with_tail(full, n, tail_shape, device, pool=None):
    """-> (buf, tail): a zeroed [full, *tail_shape] fp32 buffer and its rows [n:] (a contiguous view) for a kernel to write:
    the [B,C]-sized gradients of a node inside `nograd_prefix` in one fill, no concatenation.  `pool`: the pass's GradPool."""
    shape = (full,) + tuple(tail_shape)
    buf = pool.zeros(shape, device) if pool is not None else torch.zeros(shape, device=device, dtype=torch.float32)
    return buf, (buf[n:] if n else buf)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------ K2
def _upfirdn2d_raw(x, kernel, out_h, out_w, up, down, pad_x0, pad_y0, flip, act=None, planar_hw=None, out=None):
    """One launch of w2e_upfirdn2d.  `act` = (out_scale[planes]|None, noise[HW]|None, noise_w|None, bias[C]|None).
    planar_hw=(in_h, in_w): x is the phase-planar [N,C,2,2,(in_h+1)/2,(in_w+1)/2] image the UP conv writes.
    `out`: a contiguous [n,c,out_h,out_w] tensor to write (e.g. the tail rows of a larger batch) instead of allocating."""
    n, c = x.shape[0], x.shape[1]
    h, w = planar_hw if planar_hw is not None else (x.shape[2], x.shape[3])
    if planar_hw is not None and tuple(x.shape[2:]) != (2, 2, (h + 1) // 2, planar_pitch((w - 1) // 2)):
        raise RuntimeError(f"upfirdn2d: a phase-planar input of a {h}x{w} image must be [N,C,2,2,{(h + 1) // 2},{planar_pitch((w - 1) // 2)}] "
                           f"(W2E_PLANAR_PITCH, include/w2e.h), got {tuple(x.shape)}")
    kh, kw = kernel.shape
    y = out if out is not None else torch.empty((n, c, out_h, out_w), device=x.device, dtype=torch.float32)
    if act is None:
        a = (0, None, None, None, None, 1, 0.2, SQRT2)
    else:
        out_scale, noise, noise_w, bias = act
        a = (1, ptr(out_scale), ptr(noise), ptr(noise_w), ptr(bias), c, 0.2, SQRT2)
    sp = profiling.span("upfirdn2d", 4.0 * n * c * (h * w + out_h * out_w))  # algorithmic bytes: read x + write y
    call("w2e_upfirdn2d", ptr(x), ptr(kernel), ptr(y), n * c, h, w, out_h, out_w, kh, kw, up, down, pad_x0, pad_y0,
         int(flip), int(planar_hw is not None), x.shape[-1] if planar_hw is not None else 0, *a, stream_ptr())
    if sp is not None:
        sp.end()
    return y


class _UpFirDn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, up, down, pad0, pad1):
        x = _c(x)
        kernel = _c(kernel.to(torch.float32))
        n, c, h, w = x.shape
        kh, kw = kernel.shape
        out_h = (h * up + pad0 + pad1 - kh) // down + 1
        out_w = (w * up + pad0 + pad1 - kw) // down + 1
        if out_h <= 0 or out_w <= 0:
            raise RuntimeError(f"upfirdn2d: empty output {out_h}x{out_w}")
        ctx.save_for_backward(kernel)
        ctx.geom = (h, w, up, down, pad0)
        return _upfirdn2d_raw(x, kernel, out_h, out_w, up, down, pad0, pad0, True)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (kernel,) = ctx.saved_tensors
        h, w, up, down, pad0 = ctx.geom
        kh, kw = kernel.shape
        # adjoint: swap up/down, un-flipped taps, leading pad k-1-pad0, output cropped to the input size
        gx = _upfirdn2d_raw(_c(gy), kernel, h, w, down, up, kw - 1 - pad0, kh - 1 - pad0, False)
        return gx, None, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """models/stylegan2/op/upfirdn2d.py:11 -- same signature and semantics, HIP kernel underneath."""
    return _UpFirDn2d.apply(input, kernel, up, down, pad[0], pad[1])


# ------------------------------------------------------------------------------------------ K3
class _BiasAct(torch.autograd.Function):
    """y = lrelu(x + bias (+ w*noise), slope) * gain with x viewed as [outer, C, inner]."""

    @staticmethod
    def forward(ctx, x, bias, slope, gain, channel_last):
        x = _c(x)
        bias = _c(bias)
        if channel_last:
            outer, ch, inner = x.numel() // x.shape[-1], x.shape[-1], 1
        else:
            outer, ch, inner = x.shape[0], x.shape[1], x.numel() // (x.shape[0] * x.shape[1])
        if bias.numel() != ch:
            raise RuntimeError(f"fused_leaky_relu: bias has {bias.numel()} entries, expected {ch}")
        y = torch.empty_like(x)
        call("w2e_bias_act_fwd", ptr(x), ptr(bias), None, None, ptr(y), outer, ch, inner, slope, gain, stream_ptr())
        ctx.save_for_backward(y)
        ctx.cfg = (slope, gain, channel_last, ch)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        slope, gain, channel_last, ch = ctx.cfg
        gy = _c(gy)
        gx = torch.empty_like(y)
        call("w2e_bias_act_bwd", ptr(gy), ptr(y), ptr(gx), y.numel(), slope, gain, stream_ptr())
        gb = None
        if ctx.needs_input_grad[1]:
            gb = gx.reshape(-1, ch).sum(0) if channel_last else gx.reshape(gx.shape[0], ch, -1).sum((0, 2))
        return gx, gb, None, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    """models/stylegan2/op/fused_act.py:23 -- bias on dim 1, except 3-D inputs (bias on the last dim).
    No device move inside the op (the reference's `input.cuda()` is a bug on CPU boxes, Q1)."""
    if input.ndim < 2:
        raise RuntimeError("fused_leaky_relu expects at least 2 dims")
    channel_last = input.ndim == 3 or input.ndim == 2
    return _BiasAct.apply(input, bias, float(negative_slope), float(scale), channel_last)


# ------------------------------------------------------------------------------------------ K1
def conv_pack(weight, scale, transpose, flip):
    """weight [Cout,Cin,3,3] -> [ceil(K/8)][9][2][N][4] (w2e_conv_pack): K the reduced, N the produced channels."""
    cout, cin = weight.shape[0], weight.shape[1]
    kd, nd = (cout, cin) if transpose else (cin, cout)
    wp = torch.empty(((kd + 7) // 8, 9, 2, nd, 4), device=weight.device, dtype=torch.float32)
    call("w2e_conv_pack", ptr(_c(weight.detach())), ptr(wp), cout, cin, float(scale), int(transpose), int(flip),
         stream_ptr())
    return wp


MODE_SAME, MODE_UP, MODE_DOWN = 0, 1, 2

# ---- Winograd F(4x4,3x3) forms of the same-resolution layers (include/w2e.h, K1w / K1g).  Two own kernels, no vendor GEMM:
#   form 4      the GEMM form for the WIDE layers: w2e_wino_pack_input (V in MFMA operand order) -> w2e_wino_gemm (the 36 contractions on fp32
#               MFMA, operands L2 -> registers, the output transform and the direct kernel's epilogues in its epilogue: M never reaches HBM)
#   form FUSED  w2e_wino_fused for the narrow high-resolution layers (nothing transform-domain in HBM at all)
# WINOGRAD (W2E_WINOGRAD, read once here):
#   "auto"  per layer, what measures fastest: form 4 for K, N >= 128 at 16^2 ... 128^2 (K = N = 256 / 512), FUSED for 128 @ 256^2, 64 @ 512^2,
#           32 @ 1024^2, the direct kernels for the rest (up-sampling layers, <= 8^2).  Rounding ~1e-5 relative (direct: 3e-7)
#   False   direct kernels only ("0");  4 / 8: that form wherever the shapes allow (tests, tools)
def _parse_winograd(v):
    table = {"": "auto", "auto": "auto", "0": False, "4": 4, "8": 8}
    if v not in table:  # (a typo used to mean "auto": a user asking for the exact direct kernels got the 1e-5 forms silently)
        raise ValueError(f"W2E_WINOGRAD={v!r}: expected 'auto', '0' (direct kernels), '4' (GEMM form) or '8' (fused form)")
    return table[v]


WINOGRAD = _parse_winograd(os.environ.get("W2E_WINOGRAD", ""))
WINO_LOG = None  # a list: every Winograd-form conv appends one line in the format of the library's tune_print (tests, tools/cfg_selections.py)


FUSED = 8  # form code of the fused F(4x4,3x3) kernel (w2e_wino_fused)


def set_winograd(mode):
    """mode: "auto" | False | 4 | 8 (4: the GEMM form, 8: the fused F(4x4,3x3) kernel, each wherever its shapes allow)"""
    global WINOGRAD
    if mode not in ("auto", False, 4, 8):
        raise ValueError("set_winograd: 'auto', False, 4 or 8")
    WINOGRAD = mode


FUSED_WGS = 0           # > 0: cap of the fused kernel's persistent grid (tests)
X_LIMIT = 2 ** 32 - 64  # bytes: the fused kernel addresses the whole input through ONE buffer descriptor


def _fused_shape_ok(b, k, n, h, w, dot=False):
    """w2e_wino_fused's shape rules (include/w2e.h, K1w).  It range-checks its loads against one descriptor over x, so it needs
    x < 4 GB (total batch 32 of the 32 @ 1024^2 layer is 4 GiB: the merged [w; w_hat] pass reaches that at batch_size 16) -- past it
    the selection falls back to the GEMM form or the direct kernel, which only needs one image under 4 GB."""
    if not (b > 0 and n % 32 == 0 and h % 16 == 0 and w % 32 == 0 and b * (h // 16) * (w // 32) < 2 ** 31):
        return False
    return 32 <= k <= 256 and k & (k - 1) == 0 and 4 * b * k * h * w < X_LIMIT


def _gemm_shape_ok(b, k, n, h, w, dot=True, ragged=False):
    """w2e_wino_gemm's shape rules (include/w2e.h, K1g).  `ragged`: the caller accepts the form for H, W that are not multiples of 4
    (tiles hang over the image; plain and bias + PReLU epilogues only: the encoders' 14^2 / 7^2 stages)."""
    if b <= 0 or k % 8 or n % 64 or h < 4 or w < 4:
        return False
    if (h % 4 or w % 4) and (dot or not ragged):
        return False
    tiles = -(-h // 4) * -(-w // 4)
    tp = (b * tiles + 31) & ~31
    if 36 * k * tp * 4 >= X_LIMIT or 36 * k * n * 4 >= X_LIMIT or b * tiles >= 2 ** 30:
        return False
    return (not dot) or tiles % 32 == 0 or (tiles < 32 and tiles & (tiles - 1) == 0)  # (the fused dot reduces over half-wave segments)


def _wino_form(x, k, n, h, w, dot_with):
    """0 (direct kernel), 4 (the GEMM form: w2e_wino_pack_input + w2e_wino_gemm) or FUSED (w2e_wino_fused) for one W2E_CONV_SAME call."""
    if WINOGRAD is False:
        return 0
    b = x.shape[0]
    dot = dot_with is not None
    if WINOGRAD == "auto":
        m = 4 if (k >= 128 and n >= 128 and 16 <= h <= 256 and 16 <= w <= 256) else 0
        if k <= 128 and n <= 128 and h >= 256 and w >= 256 and _fused_shape_ok(b, k, n, h, w, dot):
            m = FUSED  # 128 @ 256^2, 64 @ 512^2, 32 @ 1024^2
    else:
        m = WINOGRAD
    if m == FUSED:
        if not _fused_shape_ok(b, k, n, h, w, dot):
            return 0
    elif not m or not _gemm_shape_ok(b, k, n, h, w, dot):
        return 0
    if _lib.get_option("conv_precision") != 0:
        return 0
    # (bit-reproducible mode keeps both forms: the fused kernel leaves per-block partials of its fused dot, the GEMM form per-segment
    # partials and K-split slabs -- all summed in a fixed order, no atomics)
    return m if _lib.get_option("tune_cfg") < 0 else 0  # (a forced direct tile: tests, tools/layer_bench.py)


def _wino_weights_fused(wp, k, n):
    uf = getattr(wp, "_w2e_wino_uf", None)
    if uf is None:
        uf = torch.empty((36, k // 8, 2, n, 4), device=wp.device, dtype=torch.float32)
        call("w2e_wino_weights_fused", ptr(wp), ptr(uf), k, n, stream_ptr())
        wp._w2e_wino_uf = uf
    return uf


GEMM_SPLITS = 0  # > 0: force w2e_wino_gemm's K split (tests); 0: the library's plan


def wino_gemm_conv(x, wp, in_scale, out_scale, y, k, n, h, w, act_code=0, noise=None, noise_w=None, bias=None, slope=None,
                   dot_with=None, dot=None, tag="modconv mode 0"):
    """One same-resolution 3x3 conv in the F(4x4,3x3) GEMM form (K1g): w2e_wino_pack_input -> w2e_wino_gemm; `dot` [b,n] is accumulated
    into (in a fixed order, no atomics) when dot_with is given."""
    b = x.shape[0]
    tp, sp, ws = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
    call("w2e_wino_gemm_plan", b, k, n, h, w, ctypes.byref(tp), ctypes.byref(sp), ctypes.byref(ws))
    splits = sp.value
    ws_floats = ws.value
    if GEMM_SPLITS > 0 and GEMM_SPLITS != splits:
        splits = min(GEMM_SPLITS, k // 8)
        kcs = -(-(k // 8) // splits)
        splits = -(-(k // 8) // kcs)
        ws_floats = b * n * max(1, -(-h // 4) * -(-w // 4) // 32) + (splits * b * n * h * w if splits > 1 else 0)
    if WINO_LOG is not None:
        WINO_LOG.append(f"{tag} (winograd F(4x4,3x3) gemm{', dot' if dot_with is not None else ''}) K {k} N {n} {h}x{w} B {b} -> "
                        f"36 x [{n}x{k}] x [{k}x{tp.value}], {splits} K split(s)")
    vf = torch.empty(36 * k * tp.value, device=x.device, dtype=torch.float32)
    work = torch.empty(ws_floats, device=x.device, dtype=torch.float32) if (splits > 1 or dot_with is not None) else None
    st = stream_ptr()
    call("w2e_wino_pack_input", ptr(x), ptr(in_scale), ptr(vf), b, k, h, w, tp.value, st)
    call("w2e_wino_gemm", ptr(_wino_weights_fused(wp, k, n)), ptr(vf), ptr(out_scale), ptr(y), b, k, n, h, w, tp.value, splits, ptr(work),
         act_code, ptr(noise), ptr(noise_w), ptr(bias), ptr(slope), ptr(dot_with), ptr(dot), st)


def _modconv_wino(m, x, wp, in_scale, out_scale, y, k, n, h, w, act, dot_with, dot):
    b = x.shape[0]
    noise = noise_w = bias = None
    if act is not None:
        noise, noise_w, bias = act
    if m == FUSED:
        if WINO_LOG is not None:
            WINO_LOG.append(f"modconv mode 0 (winograd F(4x4,3x3) fused{', dot' if dot_with is not None else ''}) "
                            f"K {k} N {n} {h}x{w} B {b} -> {b * (h // 16) * (w // 32)} blocks of 32 tiles")
        if dot_with is not None:  # one partial per (spatial block, channel): summed here, in a fixed order
            nblk = (h // 16) * (w // 32)
            part = torch.empty((b, n, nblk), device=x.device, dtype=torch.float32)
            call("w2e_wino_fused", ptr(x), ptr(in_scale), ptr(_wino_weights_fused(wp, k, n)), ptr(out_scale), ptr(y), b, k, n, h, w,
                 0, None, None, None, None, ptr(dot_with), ptr(part), FUSED_WGS, stream_ptr())
            # (a kernel, not aten::sum: that one memsets under capture.)  `dot` arrives zeroed and nothing else adds to it before this
            # call returns -- the partial sums ARE its value: written in place (round 3 summed into a temporary and added that: 2 more launches)
            call("w2e_channel_sums", ptr(part), None, ptr(dot), b, n, nblk, stream_ptr())
            return
        call("w2e_wino_fused", ptr(x), ptr(in_scale), ptr(_wino_weights_fused(wp, k, n)), ptr(out_scale), ptr(y), b, k, n, h, w,
             int(act is not None), ptr(noise), ptr(noise_w), ptr(bias), None, None, None, FUSED_WGS, stream_ptr())
        return
    wino_gemm_conv(x, wp, in_scale, out_scale, y, k, n, h, w, int(act is not None), noise, noise_w, bias, None, dot_with, dot)


def _modconv_raw(mode, x, wp, in_scale, out_scale, h, w, act=None, dot_with=None, out=None, dot_out=None):
    """One call of w2e_modconv3x3.  h,w: input size for SAME/UP, output size for DOWN.  `out`: write into this
    (contiguous, [b,n,h,w]) tensor -- e.g. the tail rows of a larger batch -- instead of allocating; `dot_out`: likewise a
    ZEROED [b,n] tensor for the fused per-channel dot (accumulated with atomics)."""
    b, k = x.shape[0], x.shape[1]
    n = wp.shape[3]
    if wp.shape[0] != (k + 7) // 8:
        raise RuntimeError(f"modconv: packed weight holds {wp.shape[0]} 8-channel groups, input has {k} channels")
    if mode == MODE_UP:  # phase-planar T: T[Y][X] = y[Y&1][X&1][Y>>1][X>>1]  (unit-stride stores per output phase)
        y = torch.empty((b, n, 2, 2, h + 1, planar_pitch(w)), device=x.device, dtype=torch.float32)  # sector-aligned rows
    elif out is not None:
        y = out
    else:
        y = torch.empty((b, n, h, w), device=x.device, dtype=torch.float32)
    dot = None
    if dot_with is not None:
        dot = dot_out if dot_out is not None else torch.zeros((b, n), device=x.device, dtype=torch.float32)
    noise = noise_w = bias = None
    if act is not None:
        noise, noise_w, bias = act
    # algorithmic FLOPs: 2*K*N*9 per domain pixel (MACs actually needed; SURVEY 2.3 convention)
    form = _wino_form(x, k, n, h, w, dot_with) if mode == MODE_SAME else 0
    sp = profiling.span("modconv3x3_wino4" if form else "modconv3x3", 2.0 * b * k * n * 9 * h * w)
    if form:
        _modconv_wino(form, x, wp, in_scale, out_scale, y, k, n, h, w, act, dot_with, dot)
    else:
        call("w2e_modconv3x3", mode, ptr(x), ptr(wp), ptr(in_scale), ptr(out_scale), ptr(y), b, k, n, h, w,
             y.shape[-1] if mode == MODE_UP else 0, int(act is not None), ptr(noise), ptr(noise_w), ptr(bias), ptr(dot_with), ptr(dot), stream_ptr())
    if sp is not None:
        sp.end()
    return y, dot


def _channel_dot(a, b, out=None):
    """[B,C] = sum_p a*b per plane (w2e_channel_sums: one wave per plane, fixed reduction order)."""
    n, c = a.shape[0], a.shape[1]
    if out is None:
        out = torch.empty((n, c), device=a.device, dtype=torch.float32)
    call("w2e_channel_sums", ptr(a), ptr(b), ptr(out), n, c, a.shape[2] * a.shape[3], stream_ptr())
    return out


def _scale_planes(a, s):
    """a[b,c,:,:] * s[b,c]  (w2e_se_apply_bwd with a zero offset)."""
    n, c = a.shape[0], a.shape[1]
    out = torch.empty_like(a)
    zero = torch.zeros((n, c), device=a.device, dtype=torch.float32)
    call("w2e_se_apply_bwd", ptr(a), ptr(_c(s)), ptr(zero), ptr(out), n, c, a.shape[2] * a.shape[3], stream_ptr())
    return out


def demod_coefficients(s, wsq, eps=1e-8):
    """d[b,o] = rsqrt(sum_i s[b,i]^2 wsq[o,i] + eps) (w2e_demod_fwd)."""
    b, cin = s.shape
    cout = wsq.shape[0]
    d = torch.empty((b, cout), device=s.device, dtype=torch.float32)
    call("w2e_demod_fwd", ptr(s), ptr(wsq), ptr(d), b, cin, cout, float(eps), stream_ptr())
    return d


def demod_coefficients_all(styles, wsqs, eps=1e-8):
    """[demod_coefficients(s, wsq) for s, wsq in zip(styles, wsqs)] in ONE launch (w2e_demod_all_fwd): the demodulation vectors of
    every layer of a generator pass, when the styles of all layers are known up front.  styles[j]: contiguous [B,cin_j]."""
    b = styles[0].shape[0]
    couts = [w.shape[0] for w in wsqs]
    buf = torch.empty(b * sum(couts), device=styles[0].device, dtype=torch.float32)
    descs = (_lib.DemodLayer * len(styles))()
    out, off = [], 0
    for j, (s, w) in enumerate(zip(styles, wsqs)):
        if s.shape != (b, w.shape[1]) or not s.is_contiguous() or not w.is_contiguous():
            raise RuntimeError(f"demod_coefficients_all: layer {j}: style {tuple(s.shape)} for wsq {tuple(w.shape)}")
        d = buf[off:off + b * couts[j]].view(b, couts[j])
        off += b * couts[j]
        descs[j].s, descs[j].wsq, descs[j].d, descs[j].cin, descs[j].cout = ptr(s).value, ptr(w).value, ptr(d).value, w.shape[1], couts[j]
        out.append(d)
    call("w2e_demod_all_fwd", descs, len(styles), b, float(eps), stream_ptr())
    return out


def planar_pitch(w):
    """W2E_PLANAR_PITCH (include/w2e.h): row pitch of the phase planes of the transposed-conv output for an input w wide."""
    return (w + 1 + 15) & ~15


def unplanar(t, in_w):
    """[B,N,2,2,H+1,WP] phase-planar transposed-conv output (WP = W+1 padded to a multiple of 4) -> the plain
    [B,N,2H+1,2W+1] image."""
    b, n, _, _, hp, wpp = t.shape
    w = in_w
    return t.permute(0, 1, 4, 2, 5, 3).reshape(b, n, 2 * hp, 2 * wpp)[:, :, :2 * hp - 1, :2 * w + 1].contiguous()


class ActLink:
    """Pairs ONE fused StyledConv node with the ToRGB node that is the only consumer of its output (the pass-through form of the
    synthesis loop, or the last layer).  The ToRGB backward, which has the activation and the gradient it returns in registers,
    applies the StyledConv's activation backward itself (w2e_torgb_bwd_actbwd) and leaves the pre-activation gradient tensor and
    its three per-(sample, channel) sums here; the StyledConv backward takes them if -- and only if -- the gradient it receives
    is that very tensor (the engine hands it on untouched: nothing else contributes to it)."""

    __slots__ = ("noise", "gpre", "sums")

    def __init__(self, noise):
        self.noise, self.gpre, self.sums = noise, None, None

    def take(self, gout):
        gpre, sums = self.gpre, self.sums
        self.gpre = self.sums = None
        if gpre is None:
            return None  # the ToRGB node did not run its backward (its output was unused): the ordinary path
        if gpre is not gout:
            # The ToRGB backward has ALREADY applied this layer's activation backward to the gradient it returned, but what arrives
            # here is another tensor: a hook / retain_grad on the activation, a second consumer, or an engine-side copy or
            # accumulation changed it on the way.  Running the activation backward again would be silently wrong.
            raise RuntimeError("where2edit_amd: the gradient of a StyledConv output that was routed through its ToRGB node "
                               "(ActLink) reached the StyledConv backward as a different tensor -- a tensor hook, retain_grad() "
                               "or a second consumer on that activation is not supported on the fused training path "
                               "(use return_features=True / the unfused modules to tap activations)")
        return sums


class _StyledConv(torch.autograd.Function):
    """Fused StyledConv: out = lrelu(d * conv(Wp, s*x) [blur] + nw*noise + bias) * sqrt2 with
    d = rsqrt(s^2 @ wsq^T + eps) (model.py:234-276, 285-290, op/fused_act.py); with fuse_act=False just
    d * conv(Wp, s*x) [blur] (a bare ModulatedConv2d).  Differentiable in x, s (direct + demodulation paths in one
    gradient), noise_w, bias.  The conv weight is treated as frozen (no weight gradient is produced -- the decoder is
    never optimised on this path: coach.py:174-180 optimises net.mapper only)."""

    @staticmethod
    def forward(ctx, x, s, wsq, noise, noise_w, bias, packs, blur_kernel, upsample, fuse_act, link=None, d_pre=None):
        """`d_pre`: the layer's demodulation vector when the caller already has it (demod_coefficients_all)."""
        x, s = _c(x), _c(s)
        b, cin, h, w = x.shape
        wp_f, wp_b = packs
        d = None
        if wsq is not None:
            d = d_pre if d_pre is not None else demod_coefficients(s, wsq)
        act = (noise, noise_w, bias) if fuse_act else None
        if upsample:
            t, _ = _modconv_raw(MODE_UP, x, wp_f, s, d, h, w)
            if w >= 16:   # the tile kernel reads the phase-planar layout directly
                out = _upfirdn2d_raw(t, blur_kernel, 2 * h, 2 * w, 1, 1, 1, 1, True,
                                     act=((None,) + act) if fuse_act else None, planar_hw=(2 * h + 1, 2 * w + 1))
            else:         # tiny images: re-interleave (a [B,C,<=17,<=17] copy) and use the generic kernel
                out = _upfirdn2d_raw(unplanar(t, w), blur_kernel, 2 * h, 2 * w, 1, 1, 1, 1, True,
                                     act=((None,) + act) if fuse_act else None)
        else:
            out, _ = _modconv_raw(MODE_SAME, x, wp_f, s, d, h, w, act=act)
        ctx.save_for_backward(x, s, d, wsq, noise, noise_w, bias, out, wp_b, blur_kernel)
        ctx.cfg = (upsample, fuse_act)
        ctx.link = link  # (an ActLink shared with the consuming ToRGB node, or None)
        ctx.n_skip = _NOGRAD_PREFIX if _NOGRAD_PREFIX < b else 0
        ctx.pool = _GRAD_POOL
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x, s, d, wsq, noise, noise_w, bias, out, wp_b, blur_kernel = ctx.saved_tensors
        upsample, fuse_act = ctx.cfg
        n_skip, full = ctx.n_skip, x.shape[0]
        gout_full = gout
        gout = _c(gout)
        if n_skip:  # rows [:n_skip] carry no gradient: batch-major tensors, so the tails are contiguous views
            x, s, out, gout = x[n_skip:], s[n_skip:], out[n_skip:], gout[n_skip:]
            d = d[n_skip:] if d is not None else None
        b, cin, h, w = x.shape
        cout, oh, ow = out.shape[1], out.shape[2], out.shape[3]
        # inside `nograd_prefix`: full-batch gradient buffers whose tails the kernels write in place; the prefix rows of the
        # image-sized one stay unwritten (the producer node slices them off the same way), the [B,C]-sized one is zeroed
        gx_full = _grad_rows((full, cin, h, w), n_skip, x.device) if n_skip else None
        gx_out = gx_full[n_skip:] if n_skip else None
        gs_full, gs_out = _zeros_with_tail(full, n_skip, (cin,), x.device, ctx.pool)
        g_bias = g_nw = sums = dz = None
        blurred = False
        pre_sums = ctx.link.take(gout_full) if (fuse_act and ctx.link is not None) else None
        if pre_sums is not None:  # gout already IS the pre-activation gradient (the ToRGB backward applied the activation backward)
            gpre, sums = gout, pre_sums
        elif fuse_act:
            sums = torch.empty((b, cout, 3), device=x.device, dtype=torch.float32)
            if (upsample and ow >= 256 and ow % 4 == 0 and tuple(blur_kernel.shape) == (4, 4) and not _lib.get_option("deterministic")
                    and _lib.get_option("tune_fuse") != 0):  # (tune_fuse = 0: the two-kernel form, for comparison)
                # wide up-sampling layers: activation backward, its three reductions and the adjoint blur in ONE pass over
                # (gout, out) -- the pre-activation gradient is never written (3 tensor passes instead of 5)
                gpre = torch.empty((b, cout, oh + 1, ow + 1), device=x.device, dtype=torch.float32)
                sp = profiling.span("upfirdn2d", 4.0 * b * cout * (2 * oh * ow + (oh + 1) * (ow + 1)))  # algorithmic bytes: gout + out + gT
                call("w2e_blur_adjoint_actbwd", ptr(gout), ptr(out), ptr(noise), ptr(blur_kernel), ptr(gpre), ptr(sums), b * cout, oh, ow,
                     0.2, SQRT2, stream_ptr())
                if sp is not None:
                    sp.end()
                blurred = True
            else:
                gpre = torch.empty_like(out)
                call("w2e_bias_act_bwd_reduce", ptr(gout), ptr(out), ptr(noise), ptr(gpre), ptr(sums), b, cout, oh * ow,
                     0.2, SQRT2, stream_ptr())
        else:
            gpre = gout
            if d is not None:
                dz = (gout * out).sum((2, 3))
        if fuse_act:
            if bias is not None and ctx.needs_input_grad[5]:
                g_bias = sums[..., 2].sum(0)
            if noise is not None and ctx.needs_input_grad[4]:
                g_nw = sums[..., 1].sum().reshape(1)
        if upsample and not blurred:
            # adjoint of Blur(pad=(1,1)) back onto the (2h+1)x(2w+1) transposed-conv grid, then the
            # stride-2 conv that is the adjoint of conv_transpose2d
            gpre = _upfirdn2d_raw(gpre, blur_kernel, 2 * h + 1, 2 * w + 1, 1, 1, 2, 2, False)
        mode = MODE_DOWN if upsample else MODE_SAME
        if _lib.get_option("deterministic") and not (mode == MODE_SAME and _wino_form(gpre, gpre.shape[1], cin, h, w, x)):
            # (both Winograd forms sum the partials of their fused dot in a fixed order: they stay)
            # the direct kernel's fused dot epilogue joins the workgroups of a (b, channel) with fp32 atomics; here instead: the unscaled
            # input gradient, its per-channel dot with x by a fixed-order wave reduction, then the out_scale
            raw, _ = _modconv_raw(mode, gpre, wp_b, d, None, h, w)
            gs = _channel_dot(raw, x, out=gs_out)
            gx = _scale_planes(raw, s)
            if n_skip:
                gx_out.copy_(gx)
        else:
            gx, gs = _modconv_raw(mode, gpre, wp_b, d, s, h, w, dot_with=x, out=gx_out, dot_out=gs_out)
        if d is not None:  # + the demodulation path: gs -= s * (dz*d^2) @ wsq, with dz = sum_p gpre*(pre - nw*noise - bias)
            call("w2e_demod_bwd", ptr(sums), ptr(dz), ptr(noise_w) if (fuse_act and noise is not None) else None,
                 ptr(bias) if fuse_act else None, ptr(d), ptr(s), ptr(wsq), ptr(gs), None, b, cin, cout, stream_ptr())
        if n_skip:
            gx = gx_full
        return gx, gs_full, None, None, g_nw, g_bias, None, None, None, None, None, None


def styled_conv(x, s, wsq, noise, noise_w, bias, packs, blur_kernel, upsample, link=None, demod=None):
    return _StyledConv.apply(x, s, wsq, noise, noise_w, bias, packs, blur_kernel, upsample, True, link, demod)


def modconv(x, s, wsq, packs, blur_kernel, upsample):
    """Bare ModulatedConv2d (3x3): d * conv(Wp, s*x), with the FIR blur for the up-sampling variant."""
    return _StyledConv.apply(x, s, wsq, None, None, None, packs, blur_kernel, upsample, False)


def modconv_down_plain(x, s, d, wp_f, h, w):
    """Forward-only stride-2 conv on a (2h+1)x(2w+1) input."""
    y, _ = _modconv_raw(MODE_DOWN, _c(x), wp_f, _c(s) if s is not None else None, _c(d) if d is not None else None, h, w)
    return y


# ------------------------------------------------------------------------------------------ style affines
class _StyleAffineAll(torch.autograd.Function):
    """Every `style = self.modulation(style)` of one generator pass (model.py:211, 26 EqualLinear calls) in one launch.
    pack = (w [R,D] = stacked weight*scale, bias [R] = stacked bias*lr_mul, meta int32 [R,4], widths): see w2e.h.
    Returns one [B, cin_l] tensor per layer; differentiable in the latent only (the decoder is frozen on this path)."""

    @staticmethod
    def forward(ctx, latent, pack):
        import ctypes
        latent = _c(latent)
        b, n_latent, dim = latent.shape
        w, bias, meta, widths = pack
        rows = w.shape[0]
        out = torch.empty(b * rows, device=latent.device, dtype=torch.float32)
        call("w2e_style_affine_fwd", ptr(latent), ptr(w), ptr(bias), ctypes.c_void_p(meta.data_ptr()), ptr(out), b, n_latent,
             dim, rows, stream_ptr())
        ctx.pack = pack
        ctx.geom = (b, n_latent, dim)
        ctx.n_skip = _NOGRAD_PREFIX if _NOGRAD_PREFIX < b else 0
        ctx.pool = _GRAD_POOL
        outs, off = [], 0
        for cw in widths:
            outs.append(out[off * b:(off + cw) * b].view(b, cw))
            off += cw
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        import ctypes
        b, n_latent, dim = ctx.geom
        n_skip = ctx.n_skip
        b -= n_skip
        w, bias, meta, widths = ctx.pack
        parts = [(g[n_skip:].reshape(-1) if g is not None else torch.zeros(b * cw, device=w.device, dtype=torch.float32))
                 for g, cw in zip(gs, widths)]
        flat = torch.cat(parts)
        glat_full, glat = _zeros_with_tail(b + n_skip, n_skip, (n_latent, dim), w.device, ctx.pool)
        call("w2e_style_affine_bwd", ptr(flat), ptr(w), ctypes.c_void_p(meta.data_ptr()), ptr(glat), b, n_latent, dim,
             w.shape[0], stream_ptr())
        return glat_full, None


def style_affine_all(latent, pack):
    return _StyleAffineAll.apply(latent, pack)


# ------------------------------------------------------------------------------------------ K1r
class _ToRGB(torch.autograd.Function):
    """`style` None: wmod is the per-sample [B,3,cin] weight; else wmod is the shared (frozen) [3,cin] scale*W and the kernels
    form wmod[c,i]*style[b,i] themselves, the backward returning the style gradient directly (w2e_torgb_styled_*).
    `passthrough`: also return x itself as a second output for the NEXT layer to consume.  x then has this node as its
    only consumer, and the gradient coming back through the next layer arrives here as `gx_next` and is folded into the
    ToRGB input gradient by the kernel (w2e_torgb_bwd_acc) -- instead of autograd adding two activation-sized tensors."""

    @staticmethod
    def forward(ctx, x, wmod, style, bias, skip, upk, passthrough=False, producer_act=None):
        """`producer_act` = the ActLink of the fused StyledConv (slope 0.2, gain sqrt 2) whose output x is, when this node is its only
        consumer (the pass-through form, or the last layer): the backward then returns that layer's PRE-activation gradient."""
        x_in = x
        x, wmod = _c(x), _c(wmod)
        b, cin, h, w = x.shape
        y = torch.empty((b, 3, h, w), device=x.device, dtype=torch.float32)
        skip_c = _c(skip) if skip is not None else None
        bias_c = _c(bias.reshape(-1)) if bias is not None else None
        if style is None:
            call("w2e_torgb_fwd", ptr(x), ptr(wmod), ptr(bias_c), ptr(skip_c), ptr(upk) if skip is not None else None,
                 ptr(y), b, cin, h, w, stream_ptr())
        else:
            style = _c(style.reshape(b, cin))
            if wmod.numel() != 3 * cin:
                raise RuntimeError(f"to_rgb: the shared weight must be [3,{cin}], got {tuple(wmod.shape)}")
            call("w2e_torgb_styled_fwd", ptr(x), ptr(wmod), ptr(style), ptr(bias_c), ptr(skip_c),
                 ptr(upk) if skip is not None else None, ptr(y), b, cin, h, w, stream_ptr())
        ctx.save_for_backward(x, wmod, style, upk if skip is not None else None, producer_act.noise if producer_act is not None else None)
        ctx.link = producer_act  # (the caller vouches that this node is x's only consumer)
        ctx.has = (bias is not None, skip is not None, tuple(bias.shape) if bias is not None else None)
        ctx.n_skip = _NOGRAD_PREFIX if _NOGRAD_PREFIX < b else 0
        ctx.pool = _GRAD_POOL
        if passthrough:
            return y, x_in.view_as(x_in)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gx_next=None):
        x, wmod, style, upk, act_noise = ctx.saved_tensors
        has_bias, has_skip, bias_shape = ctx.has
        n_skip = ctx.n_skip
        b, cin, h, w = x.shape
        if gy is None:  # only the pass-through output was used
            return gx_next, None, None, None, None, None, None, None
        gy = _c(gy)
        gx = _grad_rows(None, n_skip, x.device, like=x)
        acc = _c(gx_next) if gx_next is not None else None
        if n_skip:  # the tails of batch-major tensors: contiguous views, written / read in place
            xs, gys, gxs = x[n_skip:], gy[n_skip:], gx[n_skip:]
            accs = acc[n_skip:] if acc is not None else None
        else:
            xs, gys, gxs, accs = x, gy, gx, acc
        ws = wmod if style is not None else (wmod[n_skip:] if n_skip else wmod)
        sts = None if style is None else (style[n_skip:] if n_skip else style)
        gw_full, gw = _zeros_with_tail(b, n_skip, (cin,) if style is not None else (3, cin), x.device, ctx.pool)
        if ctx.link is not None:
            # x is the activated output of the StyledConv below: hand that layer its pre-activation gradient and sums directly
            sums3 = torch.empty((b - n_skip, cin, 3), device=x.device, dtype=torch.float32)
            call("w2e_torgb_bwd_actbwd", ptr(xs), ptr(ws), ptr(sts), ptr(gys), ptr(accs), ptr(act_noise), ptr(gxs), ptr(gw), ptr(sums3),
                 b - n_skip, cin, h, w, 0.2, SQRT2, stream_ptr())
            ctx.link.gpre, ctx.link.sums = gx, sums3
        elif style is None:
            call("w2e_torgb_bwd_acc", ptr(xs), ptr(ws), ptr(gys), ptr(accs), ptr(gxs), ptr(gw), b - n_skip, cin, h, w, stream_ptr())
        else:
            call("w2e_torgb_styled_bwd", ptr(xs), ptr(ws), ptr(sts), ptr(gys), ptr(accs), ptr(gxs), ptr(gw), b - n_skip, cin, h, w,
                 stream_ptr())
        g_wmod, g_style = (gw_full.view_as(wmod), None) if style is None else (None, gw_full)
        gb = gys.sum((0, 2, 3)).reshape(bias_shape) if (has_bias and ctx.needs_input_grad[3]) else None
        gskip = None
        if has_skip:  # adjoint of Upsample(up=2, pad=(2,1)): down=2, un-flipped taps, leading pad 4-1-2
            gskip = _grad_rows((b, 3, h // 2, w // 2), n_skip, x.device)  # rows [:n_skip]: see _StyledConv
            _upfirdn2d_raw(gys, upk, h // 2, w // 2, 1, 2, 1, 1, False, out=gskip[n_skip:] if n_skip else gskip)
        return gx, g_wmod, g_style, gb, gskip, None, None, None


def to_rgb(x, wmod, bias, skip, upk, passthrough=False, style=None, producer_act=None):
    """y = sum_i wmod[b,c,i] x[b,i] + bias + Upsample(skip)   (model.py:353-362); with `passthrough` -> (y, x).
    With `style` [B,cin]: wmod is the shared [3,cin] scale*W (treated as frozen) and the weight of sample b is
    wmod[c,i]*style[b,i]."""
    return _ToRGB.apply(x, wmod, style, bias, skip, upk, passthrough, producer_act)


# ------------------------------------------------------------------------------------------ K5
class _ClipPreproc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img):
        img = _c(img)
        n, c, h, w = img.shape
        if h != w:
            raise RuntimeError("clip_preprocess expects square images")
        out = torch.empty((n, c, 224, 224), device=img.device, dtype=torch.float32)
        call("w2e_clip_preproc_fwd", ptr(img), ptr(out), n * c, h, stream_ptr())
        ctx.shape = img.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        n, c, h, w = ctx.shape
        gimg = torch.empty(ctx.shape, device=gout.device, dtype=torch.float32)
        call("w2e_clip_preproc_bwd", ptr(_c(gout)), ptr(gimg), n * c, h, stream_ptr())
        return gimg


def clip_preprocess(img):
    """AvgPool2d(size//32)(Upsample(scale_factor=7)(img)) in one closed-form pass (clip_loss.py:11-12,15)."""
    return _ClipPreproc.apply(img)


class _IdPreproc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img):
        img = _c(img)
        n, c, h, w = img.shape
        out = torch.empty((n, c, 112, 112), device=img.device, dtype=torch.float32)
        call("w2e_id_preproc_fwd", ptr(img), ptr(out), n * c, h, stream_ptr())
        ctx.shape = img.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        n, c, h, w = ctx.shape
        gimg = torch.empty(ctx.shape, device=gout.device, dtype=torch.float32)
        call("w2e_id_preproc_bwd", ptr(_c(gout)), ptr(gimg), n * c, h, stream_ptr())
        return gimg


def id_preprocess(img):
    """face_pool(pool(img)[:, :, 35:223, 32:220]) of criteria/id_loss.py:19-23 in one pass (square, size % 256 == 0)."""
    return _IdPreproc.apply(img)


# ------------------------------------------------------------------------------------------ K6
class _MaskBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mask):
        a, b, mask = _c(a), _c(b), _c(mask)
        n, c, h, w = a.shape
        if b.shape != a.shape or mask.shape[0] != n or mask.shape[1] != 1 or mask.shape[2] != mask.shape[3]:
            raise RuntimeError(f"mask_blend: shapes {tuple(a.shape)} {tuple(b.shape)} {tuple(mask.shape)}")
        out = torch.empty_like(a)
        call("w2e_mask_blend_fwd", ptr(a), ptr(b), ptr(mask), ptr(out), n, c, h, w, mask.shape[2], stream_ptr())
        ctx.save_for_backward(a, b, mask)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        a, b, mask = ctx.saved_tensors
        n, c, h, w = a.shape
        gout = _c(gout)
        ga = torch.empty_like(a)
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        gm = torch.empty_like(mask) if ctx.needs_input_grad[2] else None
        call("w2e_mask_blend_bwd", ptr(gout), ptr(a), ptr(b), ptr(mask), ptr(ga), ptr(gb), ptr(gm), n, c, h, w,
             mask.shape[2], stream_ptr())
        return ga, gb, gm


def mask_blend(new, old, mask):
    """m*new + (1-m)*old with m = nearest-resized mask (attention_model.py:548-549)."""
    return _MaskBlend.apply(new, old, mask)

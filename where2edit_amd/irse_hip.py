"""The IR-SE50 ArcFace body (models/facial_recognition/model_irse.py:9-48, helpers.py:56-119) on the hand-written
kernels (include/w2e_irse.h): every 3x3 / shortcut convolution runs on the fp32-MFMA engine of the StyleGAN2 layers
(w2e_conv3x3) with eval-mode BatchNorm folded into its scales and bias and PReLU in its epilogue; the SE block is a
one-wave-per-plane pooling kernel, a one-workgroup-per-sample gate kernel (fc1 / ReLU / fc2 / sigmoid) and a gate-and-add kernel.

Eval mode only (criteria/id_loss.py:14 calls facenet.eval()) and frozen weights: forward + INPUT gradients, which is what
the identity loss needs (the gradient flows back to the generated image).  One autograd node per bottleneck unit."""
import os

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from . import functional as K
from ._lib import call, ptr, stream_ptr

_I = __import__("ctypes").c_int
_P = __import__("ctypes").c_void_p
_L = __import__("ctypes").c_int64
_F = __import__("ctypes").c_float
PROTOS = {
    "w2e_conv3x3": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "w2e_affine_act_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _L, _P]),
    "w2e_affine_act_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "w2e_channel_sums": (_I, [_P, _P, _P, _I, _I, _L, _P]),
    "w2e_se_gate_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "w2e_se_gate_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "w2e_se_apply_fwd": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "w2e_se_apply_bwd": (_I, [_P, _P, _P, _P, _I, _I, _L, _P]),
    "w2e_shortcut_add_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "w2e_upsample_add": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _P]),
}


def declare(lib):
    for name, (res, args) in PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args


# ---------------------------------------------------------------------------------------------- raw kernel calls
FUSED_ENCODER = os.environ.get("W2E_FUSED_ENCODER", "1") != "0"  # (A/B aid: the fused Winograd kernel for the encoders' stride-1 convs)


def _wino_form(b, k, n, h, w):
    """The Winograd F(4x4,3x3) form (functional.py, K1w / K1g) of one stride-1 3x3 conv of the IR-SE50 / e4e encoders, or 0: the fused kernel
    for the narrow high-resolution stages, the GEMM form (own MFMA contraction, output transform in its epilogue) where the channel counts
    allow -- also for 14^2 / 7^2, whose last tile row / column hangs over the image (the ragged form) --, the direct kernel where the call
    is too small to pay for two or three launches (profiles/r05_irse_shapes.txt)."""
    if K.WINOGRAD is False:
        return 0
    if _lib.get_option("conv_precision") != 0 or _lib.get_option("tune_cfg") >= 0:
        return 0
    # Measured per shape (profiles/r05_irse_shapes.txt, the fused kernel with 64 output channels per workgroup where the grid fills the
    # chip; batch 8 of the e4e encoder / batch 16 of the id-loss network, us, direct / GEMM form / fused): 64 -> 64 @ 256^2 306 / 255 /
    # 137; 64 -> 64 @ 128^2 81 / 49 / 37; 64 -> 128 @ 128^2 148 / 73 / 73; 128 -> 128 @ 64^2 81 / 51 / 41 (32-channel workgroups: 64
    # blocks); 128 -> 256 @ 64^2 152 / 58 / 59; 256 -> 256 @ 32^2 88 / 44 / 103; 64 -> 64 @ 112^2 (batch 16) 163 / 89 / -;
    # 64 -> 128 @ 56^2 87 / 42; 128 -> 256 @ 28^2 80 / 55; 256 -> 256 @ 14^2 54 / 45; 256 -> 512 @ 14^2 89 / 53; 512 -> 512 @ 7^2 65 / 50;
    # below ~3 GFLOP of direct work per call the direct kernel wins (64 -> 64 @ 56^2, batch 8: 26 / 39; 256 -> 256 @ 14^2, batch 8: 33 / 42).  The fused kernel repeats its input transform per 32 output channels, the GEMM form pays one pass over V:
    # fused where the layer does not widen (N <= K <= 128), the GEMM form otherwise.
    work_ok = 18.0 * b * k * n * h * w >= 3e9
    fused_ok = FUSED_ENCODER and K._fused_shape_ok(b, k, n, h, w)
    gemm_ok = K._gemm_shape_ok(b, k, n, h, w, dot=False, ragged=True)
    if K.WINOGRAD == K.FUSED:
        return K.FUSED if fused_ok else 0
    if K.WINOGRAD == 4:
        return 4 if gemm_ok else 0
    if not work_ok:
        return 0
    if fused_ok and k <= 128 and n <= k:
        return K.FUSED
    if gemm_ok and k >= 64:
        return 4
    return K.FUSED if (fused_ok and k <= 128) else 0


def conv3x3(x, wp, n_out, h, w, mode=K.MODE_SAME, down_pad=0, in_scale=None, out_scale=None, bias=None, slope=None, out=None, form=None):
    """w2e_conv3x3.  h,w: input size for SAME / UP, output size for DOWN.  in_scale [B,K] / out_scale [B,N] / bias, slope [N].
    `out`: a contiguous [B,n_out,h,w] tensor to write (SAME / DOWN).  `form`: None = the library's / _wino_form's choice;
    0 / 4 / 8 force the direct kernel / a Winograd form (tools/irse_shapes.py)."""
    b, k = x.shape[0], x.shape[1]
    if mode == K.MODE_SAME and b > 0:
        m = _wino_form(b, k, n_out, h, w) if form is None else form
        if m:
            y = out if out is not None else torch.empty((b, n_out, h, w), device=x.device, dtype=torch.float32)
            if out is not None:
                assert out.shape == (b, n_out, h, w) and out.is_contiguous()
            epi = 2 if (bias is not None or slope is not None) else 0
            if m == K.FUSED:
                if K.WINO_LOG is not None:
                    K.WINO_LOG.append(f"conv3x3 (winograd F(4x4,3x3) fused) K {k} N {n_out} {h}x{w} B {b}")
                call("w2e_wino_fused", ptr(x), ptr(in_scale), ptr(K._wino_weights_fused(wp, k, n_out)), ptr(out_scale), ptr(y), b, k, n_out, h, w,
                     epi, None, None, ptr(bias), ptr(slope), None, None, K.FUSED_WGS, stream_ptr())
            else:
                K.wino_gemm_conv(x, wp, in_scale, out_scale, y, k, n_out, h, w, epi, None, None, bias, slope, tag="conv3x3")
            return y
    if out is not None:
        assert mode != K.MODE_UP and out.shape == (b, n_out, h, w) and out.is_contiguous()
        y = out
    elif mode == K.MODE_UP:
        y = torch.empty((b, n_out, 2, 2, h + 1, K.planar_pitch(w)), device=x.device, dtype=torch.float32)
    else:
        y = torch.empty((b, n_out, h, w), device=x.device, dtype=torch.float32)
    call("w2e_conv3x3", mode, ptr(x), ptr(wp), ptr(in_scale), ptr(out_scale), ptr(y), b, k, n_out, h, w,
         y.shape[-1] if mode == K.MODE_UP else 0, down_pad, ptr(bias), ptr(slope), stream_ptr())
    return y


def affine_act(x, a=None, b=None, slope=None):
    y = torch.empty_like(x)
    call("w2e_affine_act_fwd", ptr(x), ptr(a), ptr(b), ptr(slope), ptr(y), x.shape[0], x.shape[1], x.shape[2] * x.shape[3], stream_ptr())
    return y


def affine_act_bwd(gy, y, a, slope, batch, channels, height, width, planar=False):
    gx = torch.empty((batch, channels, height, width), device=gy.device, dtype=torch.float32)
    if planar and tuple(gy.shape[-4:]) != (2, 2, height // 2 + 1, K.planar_pitch(width // 2)):
        raise RuntimeError(f"affine_act_bwd: the phase-planar gradient of a {height}x{width} image must end in "
                           f"[2,2,{height // 2 + 1},{K.planar_pitch(width // 2)}] (W2E_PLANAR_PITCH), got {tuple(gy.shape)}")
    call("w2e_affine_act_bwd", ptr(gy), ptr(y), ptr(a), ptr(slope), ptr(gx), batch, channels, height, width,
         gy.shape[-1] if planar else 0, stream_ptr())  # (the pitch the tensor WAS allocated with: the library checks it against its own)
    return gx


def channel_sums(x, y=None):
    b, c = x.shape[0], x.shape[1]
    out = torch.empty((b, c), device=x.device, dtype=torch.float32)
    call("w2e_channel_sums", ptr(x), ptr(y), ptr(out), b, c, x.shape[2] * x.shape[3], stream_ptr())
    return out


# ---------------------------------------------------------------------------------------------- folded parameters
def _bn_affine(bn):
    a = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
    return a.contiguous(), (bn.bias.detach() - bn.running_mean * a).contiguous()


class UnitPlan:
    """Folded, packed, frozen parameters of one bottleneck_IR_SE unit (helpers.py:97-119)."""

    def __init__(self, unit, in_channel, depth, stride):
        res = unit.res_layer
        self.cin, self.depth, self.stride = in_channel, depth, stride
        with torch.no_grad():
            self.a1, self.b1 = _bn_affine(res[0])
            w1, w2 = res[1].weight.detach().float(), res[3].weight.detach().float()
            self.w1f = K.conv_pack(w1, 1.0, transpose=False, flip=False)
            self.w1b = K.conv_pack(w1, 1.0, transpose=True, flip=True)
            self.slope = res[2].weight.detach().float().contiguous()
            self.fused_prelu = bool((self.slope > 0).all().item())  # backward recovers the PReLU branch from sign(output)
            self.w2f = K.conv_pack(w2, 1.0, transpose=False, flip=False)
            self.w2b = K.conv_pack(w2, 1.0, transpose=True, flip=(stride == 1))
            self.a2, self.b2 = _bn_affine(res[4])
            se = res[5]
            self.fc1 = se.fc1.weight.detach().float().reshape(se.fc1.weight.shape[0], -1).contiguous()
            self.fc2 = se.fc2.weight.detach().float().reshape(se.fc2.weight.shape[0], -1).contiguous()
            self.conv_shortcut = in_channel != depth
            if self.conv_shortcut:  # Conv2d(in, depth, 1, stride) + BN: the centre tap of a 3x3 (helpers.py:103-106)
                ws = unit.shortcut_layer[0].weight.detach().float()
                w9 = torch.zeros(depth, in_channel, 3, 3, device=ws.device)
                w9[:, :, 1, 1] = ws[:, :, 0, 0]
                self.wsf = K.conv_pack(w9, 1.0, transpose=False, flip=False)
                self.wsb = K.conv_pack(w9, 1.0, transpose=True, flip=(stride == 1))
                self.a_s, self.b_s = _bn_affine(unit.shortcut_layer[1])
        if not self.fused_prelu:
            raise RuntimeError("IR-SE50 on the HIP kernels needs positive PReLU slopes (the backward reads the branch from the sign "
                               "of the output); this checkpoint has a non-positive one")

    def gate(self, sums, inv_hw):
        """(gate [B,C], hidden [B,R]) of the SE block from the per-plane sums (helpers.py:66-71): one launch."""
        b, r = sums.shape[0], self.fc1.shape[0]
        gate = torch.empty((b, self.depth), device=sums.device, dtype=torch.float32)
        hidden = torch.empty((b, r), device=sums.device, dtype=torch.float32)
        call("w2e_se_gate_fwd", ptr(sums), ptr(self.fc1), ptr(self.fc2), ptr(gate), ptr(hidden), b, self.depth, r, float(inv_hw), stream_ptr())
        return gate, hidden

    def gate_bwd(self, dgate, gate, hidden, inv_hw):
        """gradient at the pooled mean, already divided by H*W (what w2e_se_apply_bwd adds to every pixel of the plane)"""
        b, r = dgate.shape[0], self.fc1.shape[0]
        gpool = torch.empty_like(dgate)
        call("w2e_se_gate_bwd", ptr(dgate), ptr(gate), ptr(hidden), ptr(self.fc1), ptr(self.fc2), ptr(gpool), b, self.depth, r, float(inv_hw),
             stream_ptr())
        return gpool


_REP = {}


def _rep(v, batch):
    """[C] -> [batch, C] (the conv engine takes per-sample scales); cached: the vectors are frozen parameters."""
    key = (v.data_ptr(), v._version, batch)
    hit = _REP.get(key)
    if hit is None or hit[0] is not v:
        if len(_REP) > 4096:
            _REP.clear()
        hit = (v, v.unsqueeze(0).expand(batch, -1).contiguous())
        _REP[key] = hit
    return hit[1]


class _IRUnit(torch.autograd.Function):
    """One bottleneck_IR_SE unit.  `n_grad`: only the first n_grad samples of the batch take part in the backward (the id
    loss embeds [generated; original] in one batch and differentiates the generated half only, id_loss.py:30-33)."""

    @staticmethod
    def forward(ctx, x, plan, n_grad):
        x = x if x.is_contiguous() else x.contiguous()
        b, cin, h, w = x.shape
        p = plan
        s = p.stride
        oh, ow = h // s, w // s
        t0 = affine_act(x, p.a1, p.b1)                                                   # BN1 (helpers.py:111)
        t1 = conv3x3(t0, p.w1f, p.depth, h, w, slope=p.slope)                            # conv + PReLU (:112-113)
        a2 = _rep(p.a2, b)
        if s == 1:
            t2 = conv3x3(t1, p.w2f, p.depth, h, w, out_scale=a2, bias=p.b2)              # conv + BN2 (:114-115)
        else:
            t2 = conv3x3(t1, p.w2f, p.depth, oh, ow, mode=K.MODE_DOWN, down_pad=1, out_scale=a2, bias=p.b2)
        gate, hidden = p.gate(channel_sums(t2), 1.0 / float(oh * ow))                    # SE (:56-72)
        out = torch.empty_like(t2)
        if p.conv_shortcut:
            a_s = _rep(p.a_s, b)
            if s == 1:
                sc = conv3x3(x, p.wsf, p.depth, h, w, out_scale=a_s, bias=p.b_s)
            else:
                sc = conv3x3(x, p.wsf, p.depth, oh, ow, mode=K.MODE_DOWN, down_pad=1, out_scale=a_s, bias=p.b_s)
            call("w2e_se_apply_fwd", ptr(t2), ptr(gate), ptr(sc), 0, ptr(out), b, p.depth, oh, ow, stream_ptr())
        else:  # MaxPool2d(1, stride): the strided samples of x (:100-101)
            call("w2e_se_apply_fwd", ptr(t2), ptr(gate), ptr(x), s, ptr(out), b, p.depth, oh, ow, stream_ptr())
        ctx.plan, ctx.geom = plan, (n_grad if n_grad is not None else b, cin, h, w, oh, ow)
        ctx.save_for_backward(t1, t2, gate, hidden)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        t1, t2, gate, hidden = ctx.saved_tensors
        p = ctx.plan
        n, cin, h, w, oh, ow = ctx.geom
        s = p.stride
        full = gout.shape[0]
        gout = (gout if gout.is_contiguous() else gout.contiguous())[:n]
        t1, t2, gate, hidden = t1[:n], t2[:n], gate[:n], hidden[:n]  # leading rows of contiguous tensors: contiguous
        # SE backward: d gate = sum_p gout*t2, then through sigmoid / fc2 / ReLU / fc1 back to the mean (one launch)
        gpool = p.gate_bwd(channel_sums(gout, t2), gate, hidden, 1.0 / float(oh * ow))
        g_t2 = torch.empty_like(gout)
        call("w2e_se_apply_bwd", ptr(gout), ptr(gate), ptr(gpool), ptr(g_t2), n, p.depth, oh * ow, stream_ptr())
        a2 = _rep(p.a2, n)
        if s == 1:
            g_t1 = conv3x3(g_t2, p.w2b, p.depth, h, w, in_scale=a2)
            g_c1 = affine_act_bwd(g_t1, t1, None, p.slope, n, p.depth, h, w)
        else:  # adjoint of the padded stride-2 conv: UP, then the (+1,+1) crop folded into the PReLU backward
            tt = conv3x3(g_t2, p.w2b, p.depth, oh, ow, mode=K.MODE_UP, in_scale=a2)
            g_c1 = affine_act_bwd(tt, t1, None, p.slope, n, p.depth, h, w, planar=True)
        # the samples past n take no gradient: the convolution writes the head of a full-batch tensor, the tail is zeroed
        gx_full = torch.empty((full, cin, h, w), device=gout.device, dtype=torch.float32)
        gx = conv3x3(g_c1, p.w1b, cin, h, w, out_scale=_rep(p.a1, n), out=gx_full[:n])
        if n < full:
            gx_full[n:].zero_()
        if p.conv_shortcut:
            a_s = _rep(p.a_s, n)
            if s == 1:
                gs = conv3x3(gout, p.wsb, cin, h, w, in_scale=a_s)
                call("w2e_shortcut_add_bwd", ptr(gx), ptr(gs), n, cin, h, w, 1, 0, stream_ptr())
            else:
                ts = conv3x3(gout, p.wsb, cin, oh, ow, mode=K.MODE_UP, in_scale=a_s)
                call("w2e_shortcut_add_bwd", ptr(gx), ptr(ts), n, cin, h, w, 1, ts.shape[-1], stream_ptr())
        else:
            call("w2e_shortcut_add_bwd", ptr(gx), ptr(gout), n, cin, oh, ow, s, 0, stream_ptr())
        return gx_full, None, None


class _InputLayer(torch.autograd.Function):
    """Conv2d(3,64,3,1,1) -> BatchNorm2d -> PReLU (model_irse.py:20-22) as one conv launch."""

    @staticmethod
    def forward(ctx, x, plan, n_grad):
        x = x if x.is_contiguous() else x.contiguous()
        b, _, h, w = x.shape
        y = conv3x3(x, plan["wf"], 64, h, w, out_scale=_rep(plan["a"], b), bias=plan["b"], slope=plan["slope"])
        ctx.plan, ctx.n = plan, (n_grad if n_grad is not None else b)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        plan, n = ctx.plan, ctx.n
        full, _, h, w = y.shape
        g = affine_act_bwd((gy if gy.is_contiguous() else gy.contiguous())[:n], y[:n], plan["a"], plan["slope"], n, 64, h, w)
        gx_full = torch.empty((full, 3, h, w), device=g.device, dtype=torch.float32)
        conv3x3(g, plan["wb"], 3, h, w, out=gx_full[:n])
        if n < full:
            gx_full[n:].zero_()
        return gx_full, None, None


class BackbonePlan:
    """Everything `Backbone(112, 50, 'ir_se')` needs on the device, folded and packed once (the weights are frozen)."""

    def __init__(self, backbone):
        from .id_loss import get_blocks
        il, ol = backbone.input_layer, backbone.output_layer
        with torch.no_grad():
            a, b = _bn_affine(il[1])
            w0 = il[0].weight.detach().float()
            slope = il[2].weight.detach().float().contiguous()
            if not bool((slope > 0).all().item()):
                raise RuntimeError("IR-SE50 on the HIP kernels needs positive PReLU slopes")
            self.input = {"wf": K.conv_pack(w0, 1.0, False, False), "wb": K.conv_pack(w0, 1.0, True, True), "a": a, "b": b, "slope": slope}
            specs = [blk for stage in get_blocks(50 if len(backbone.body) == 24 else (100 if len(backbone.body) == 49 else 152)) for blk in stage]
            self.units = [UnitPlan(u, s.in_channel, s.depth, s.stride) for u, s in zip(backbone.body, specs)]
            self.out_a, self.out_b = _bn_affine(ol[0])
            bn1 = ol[4]
            a1 = torch.rsqrt(bn1.running_var + bn1.eps) * (bn1.weight.detach() if bn1.affine else 1.0)
            b1 = (bn1.bias.detach() if bn1.affine else 0.0) - bn1.running_mean * a1
            # Linear(25088,512) then BatchNorm1d: fold the BN into the weight rows and the bias (model_irse.py:24-28)
            self.fc_w = (ol[3].weight.detach().float() * a1[:, None]).contiguous()
            self.fc_b = (ol[3].bias.detach().float() * a1 + b1).contiguous()


class _OutputBN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, b):
        ctx.save_for_backward(a)
        return affine_act(x if x.is_contiguous() else x.contiguous(), a, b)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (a,) = ctx.saved_tensors
        n, c, h, w = gy.shape
        return affine_act_bwd(gy if gy.is_contiguous() else gy.contiguous(), None, a, None, n, c, h, w), None, None


def backbone_forward(plan, x, n_grad=None):
    """Backbone.forward (model_irse.py:44-48): [B,3,112,112] -> L2-normalised [B,512]."""
    y = _InputLayer.apply(x, plan.input, n_grad)
    for u in plan.units:
        y = _IRUnit.apply(y, u, n_grad)
    y = _OutputBN.apply(y, plan.out_a, plan.out_b)                # BatchNorm2d(512); Dropout is the identity in eval
    y = torch.nn.functional.linear(y.flatten(1), plan.fc_w, plan.fc_b)  # [B,25088]x[25088,512] on rocBLAS (+ folded BatchNorm1d)
    return y / torch.norm(y, 2, 1, True)

"""The CLIP visual tower on the hand-written kernels (include/w2e_vit.h): autograd Functions for
Linear (+bias, +residual, QuickGELU prologue / derivative epilogue), LayerNorm and the attention core,
and `vision_forward`, the HIP execution of clip_vit.VisionTransformer.forward.

CLIP is a frozen critic here (criteria/clip_loss.py builds it once and never optimises it), so the
Functions return input gradients only; asking for a weight gradient raises."""
import os

import torch
from torch.autograd.function import once_differentiable

from . import profiling
from ._lib import call, ptr, stream_ptr
from .clip_vit import patchify


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _gemm(a, b, trans_b, bias=None, residual=None, a_gelu=False, gelu_grad_aux=None, out=None):
    """C[M,N] = epi(pro(A) x B): B is [N,K] (trans_b) or [K,N].  `out`: a ZERO-filled [M,N] buffer to write into (a slice
    of a per-pass arena: split-K launches then skip their own memset)."""
    m, k = a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    c = out if out is not None else torch.empty((m, n), device=a.device, dtype=torch.float32)
    sp = profiling.span("vit_gemm", 2.0 * m * n * k)
    call("w2e_gemm_ex", ptr(a), ptr(b), ptr(c), m, n, k, a.stride(0), b.stride(0), n, int(trans_b), int(a_gelu),
         ptr(bias), ptr(residual), ptr(gelu_grad_aux), int(out is not None), stream_ptr())
    if sp is not None:
        sp.end()
    return c


def _frozen(*tensors):
    for t in tensors:
        if t is not None and t.requires_grad:
            raise RuntimeError("where2edit_amd ViT kernels compute input gradients only: freeze the CLIP weights "
                               "(CLIPLoss does) -- weight gradients are not implemented")


class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ residual)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        _frozen(weight, bias)
        x2 = _c(x).reshape(-1, x.shape[-1])
        r2 = _c(residual).reshape(-1, weight.shape[0]) if residual is not None else None
        y = _gemm(x2, _c(weight), True, bias=bias, residual=r2)
        ctx.save_for_backward(weight)
        ctx.has_res = residual is not None
        return y.reshape(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (weight,) = ctx.saved_tensors
        g2 = _c(gy).reshape(-1, gy.shape[-1])
        gx = _gemm(g2, _c(weight), False)
        return gx.reshape(*gy.shape[:-1], weight.shape[1]), None, None, (gy if ctx.has_res else None)


class _GeluLinear(torch.autograd.Function):
    """y = QuickGELU(h) W^T + b + residual; gelu(h) is never materialised (applied while A is staged) and the
    backward multiplies by QuickGELU'(h) in the GEMM epilogue."""

    @staticmethod
    def forward(ctx, h, weight, bias, residual):
        _frozen(weight, bias)
        h2 = _c(h).reshape(-1, h.shape[-1])
        r2 = _c(residual).reshape(-1, weight.shape[0])
        y = _gemm(h2, _c(weight), True, bias=bias, residual=r2, a_gelu=True)
        ctx.save_for_backward(h2, weight)
        ctx.shape = h.shape
        return y.reshape(*h.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        h2, weight = ctx.saved_tensors
        g2 = _c(gy).reshape(-1, gy.shape[-1])
        gh = _gemm(g2, _c(weight), False, gelu_grad_aux=h2)
        return gh.reshape(ctx.shape), None, None, gy


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _frozen(weight, bias)
        x2 = _c(x).reshape(-1, x.shape[-1])
        rows, dim = x2.shape
        y = torch.empty_like(x2)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        call("w2e_layernorm_fwd", ptr(x2), ptr(_c(weight)), ptr(_c(bias)), ptr(y), ptr(mean), ptr(rstd), rows, dim,
             float(eps), stream_ptr())
        ctx.save_for_backward(x2, weight, mean, rstd)
        ctx.shape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x2, weight, mean, rstd = ctx.saved_tensors
        g2 = _c(gy).reshape(x2.shape)
        gx = torch.empty_like(x2)
        call("w2e_layernorm_bwd", ptr(g2), ptr(x2), ptr(_c(weight)), ptr(mean), ptr(rstd), ptr(gx), x2.shape[0],
             x2.shape[1], stream_ptr())
        return gx.reshape(ctx.shape), None, None, None


class _Attention(torch.autograd.Function):
    """softmax(Q K^T / sqrt(64)) V per (batch, head) on the packed in_proj output [B, L, 3*H*64]."""

    @staticmethod
    def forward(ctx, qkv, heads):
        qkv = _c(qkv)
        b, l, d3 = qkv.shape
        if d3 != 3 * heads * 64:
            raise RuntimeError(f"attention kernel needs head_dim 64 (got width {d3 // 3} with {heads} heads)")
        out = torch.empty((b, l, heads * 64), device=qkv.device, dtype=torch.float32)
        call("w2e_attn_fwd", ptr(qkv), ptr(out), b, l, heads, stream_ptr())
        ctx.save_for_backward(qkv)
        ctx.heads = heads
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (qkv,) = ctx.saved_tensors
        b, l, _ = qkv.shape
        gqkv = torch.empty_like(qkv)
        call("w2e_attn_bwd", ptr(qkv), ptr(_c(gout)), ptr(gqkv), b, l, ctx.heads, stream_ptr())
        return gqkv, None


def linear(x, weight, bias=None, residual=None):
    return _Linear.apply(x, weight, bias, residual)


def layer_norm(x, ln):
    return _LayerNorm.apply(x, ln.weight, ln.bias, ln.eps)


def _ln_fwd(x2, w, b, eps):
    rows, dim = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(rows, device=x2.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x2.device, dtype=torch.float32)
    call("w2e_layernorm_fwd", ptr(x2), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, dim, float(eps), stream_ptr())
    return y, mean, rstd


class _ResBlock(torch.autograd.Function):
    """One whole ResidualAttentionBlock as ONE autograd node (9 launches forward, 9 backward): the per-op Functions
    above cost ~25 us of host time per kernel in autograd bookkeeping, which starves the GPU on these 5-20 us kernels;
    the residual joins of the backward ride in the LayerNorm-backward kernel instead of two extra `add` launches."""

    @staticmethod
    def forward(ctx, x, heads, eps1, eps2, ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, fc_w, fc_b, proj_w, proj_b, arena):
        _frozen(ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, fc_w, fc_b, proj_w, proj_b)
        b, l, dim = x.shape
        if dim != heads * 64:
            raise RuntimeError(f"attention kernel needs head_dim 64 (got width {dim} with {heads} heads)")
        x2 = _c(x).reshape(b * l, dim)
        y1, mean1, rstd1 = _ln_fwd(x2, ln1_w, ln1_b, eps1)
        qkv = _gemm(y1, in_w, True, bias=in_b)
        att = torch.empty((b * l, dim), device=x.device, dtype=torch.float32)
        call("w2e_attn_fwd", ptr(qkv), ptr(att), b, l, heads, stream_ptr())
        # the two N = width outputs are split-K launches: they land in zero-filled slices of the pass's arena
        z = arena.take(2, b * l, dim) if arena is not None else (None, None)
        x_mid = _gemm(att, out_w, True, bias=out_b, residual=x2, out=z[0])
        y2, mean2, rstd2 = _ln_fwd(x_mid, ln2_w, ln2_b, eps2)
        h = _gemm(y2, fc_w, True, bias=fc_b)
        out = _gemm(h, proj_w, True, bias=proj_b, residual=x_mid, a_gelu=True, out=z[1])
        ctx.save_for_backward(x2, mean1, rstd1, qkv, x_mid, mean2, rstd2, h, ln1_w, in_w, out_w, ln2_w, fc_w, proj_w)
        ctx.geom = (b, l, dim, heads)
        ctx.arena = arena
        return out.reshape(b, l, dim)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        x2, mean1, rstd1, qkv, x_mid, mean2, rstd2, h, ln1_w, in_w, out_w, ln2_w, fc_w, proj_w = ctx.saved_tensors
        b, l, dim, heads = ctx.geom
        rows = b * l
        g = _c(gout).reshape(rows, dim)
        gh = _gemm(g, proj_w, False, gelu_grad_aux=h)           # through c_proj and QuickGELU'
        z = ctx.arena.take_bwd(3, rows, dim) if ctx.arena is not None else (None, None, None)
        gy2 = _gemm(gh, fc_w, False, out=z[0])                  # through c_fc
        g_mid = torch.empty_like(x_mid)                         # through ln_2, + the residual branch
        call("w2e_layernorm_bwd_add", ptr(gy2), ptr(x_mid), ptr(ln2_w), ptr(mean2), ptr(rstd2), ptr(g), ptr(g_mid), rows, dim,
             stream_ptr())
        ga = _gemm(g_mid, out_w, False, out=z[1])               # through out_proj
        gqkv = torch.empty_like(qkv)
        call("w2e_attn_bwd", ptr(qkv), ptr(ga), ptr(gqkv), b, l, heads, stream_ptr())
        gy1 = _gemm(gqkv, in_w, False, out=z[2])                # through in_proj
        gx = torch.empty_like(x2)                               # through ln_1, + the residual branch
        call("w2e_layernorm_bwd_add", ptr(gy1), ptr(x2), ptr(ln1_w), ptr(mean1), ptr(rstd1), ptr(g_mid), ptr(gx), rows, dim,
             stream_ptr())
        return (gx.reshape(b, l, dim),) + (None,) * 16


class _ZeroArena:
    """Zero-filled scratch for the split-K GEMM outputs of one transformer pass: ONE fill launch for all blocks of the
    forward (and one more, on first use, for the backward) instead of a memset per GEMM."""

    def __init__(self, n_blocks, device):
        self.n_blocks, self.device = n_blocks, device
        self.fwd = self.bwd = None
        self.i = self.j = 0

    def _take(self, buf, idx, count, rows, dim):
        if idx >= self.n_blocks:  # a second backward through the same graph: its buffers must be zero again
            return tuple(torch.zeros((rows, dim), device=self.device, dtype=torch.float32) for _ in range(count))
        return tuple(buf[idx * count + c] for c in range(count))

    def take(self, count, rows, dim):
        if self.fwd is None:
            self.fwd = torch.zeros((self.n_blocks * count, rows, dim), device=self.device, dtype=torch.float32)
        out = self._take(self.fwd, self.i, count, rows, dim)
        self.i += 1
        return out

    def take_bwd(self, count, rows, dim):
        if self.bwd is None:
            self.bwd = torch.zeros((self.n_blocks * count, rows, dim), device=self.device, dtype=torch.float32)
        out = self._take(self.bwd, self.j, count, rows, dim)
        self.j += 1
        return out


def resblock_forward(blk, x, heads, arena=None):
    """ResidualAttentionBlock: x += out_proj(attn(in_proj(ln_1 x))); x += c_proj(QuickGELU(c_fc(ln_2 x)))."""
    return _ResBlock.apply(x, heads, blk.ln_1.eps, blk.ln_2.eps, blk.ln_1.weight, blk.ln_1.bias, blk.attn.in_proj_weight,
                           blk.attn.in_proj_bias, blk.attn.out_proj.weight, blk.attn.out_proj.bias, blk.ln_2.weight,
                           blk.ln_2.bias, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias,
                           arena)


# ---------------------------------------------------------------------------------------------- the M = 50*batch tower (csrc/vit2.hip, vit3.hip)
def _reduce_ln(part, bias, residual, gamma, beta, eps, want_x=True, want_y=True, mpad=0):
    """x = sum of the slabs (+ bias + residual); y = LayerNorm(x).  part: [S, M, D].  mpad > 0: y comes back K-quad-major
    ([D/4, mpad, 4]: w2e_gemm_pk's A operand)."""
    s, m, d = part.shape
    dev = part.device
    x = torch.empty((m, d), device=dev, dtype=torch.float32) if want_x else None
    y = mean = rstd = None
    if want_y:
        y = torch.empty((d // 4, mpad, 4) if mpad else (m, d), device=dev, dtype=torch.float32)
        mean = torch.empty(m, device=dev, dtype=torch.float32)
        rstd = torch.empty(m, device=dev, dtype=torch.float32)
    call("w2e_reduce_ln_fwd", ptr(part), s, m * d, ptr(bias), ptr(residual), ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean),
         ptr(rstd), m, d, float(eps), mpad, stream_ptr())
    return x, y, mean, rstd


def _ln_bwd_part(gpart, x, gamma, mean, rstd, add, mpad=0):
    """gx = LN'(sum of the slabs) + add; mpad > 0: also its K-quad-major copy (the next GEMM's A operand): returns (gx, gx_packed)."""
    s, m, d = gpart.shape
    gx = torch.empty((m, d), device=gpart.device, dtype=torch.float32)
    gxp = torch.empty((d // 4, mpad, 4), device=gpart.device, dtype=torch.float32) if mpad else None
    call("w2e_layernorm_bwd_part", ptr(gpart), s, m * d, ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(add), ptr(gx), m, d,
         ptr(gxp), mpad, stream_ptr())
    return (gx, gxp) if mpad else gx


# ---------------------------------------------------------------------------------------------- third-generation tower (csrc/vit3.hip)
def _pad(v, q):
    return -(-v // q) * q


class _WeightsPk:
    """K-quad-major packed copies of the frozen Linear weights (w2e_pack_kq): `fwd(w)` for y = x W^T (contraction over W's columns),
    `bwd(w)` for gx = gy W (contraction over W's rows: the packed form of W^T).  Built once per weight version: +2 x 340 MB for ViT-B/32."""

    def __init__(self, half=False):
        self.cache = {}
        self.half = bool(half)  # opt-in: fp16 packs for w2e_gemm_pk_h (VisionTransformer.set_precision("f16"))

    def _get(self, w, transposed):
        key = (w.data_ptr(), w._version)
        hit = self.cache.get((id(w), transposed))
        if hit is None or hit[0] != key:
            wd = w.detach()
            n, k = (wd.shape[1], wd.shape[0]) if transposed else (wd.shape[0], wd.shape[1])  # output columns, contraction length
            npad = _pad(n, 64)
            if self.half:
                if k % 16:
                    raise ValueError(f"fp16 tower GEMM: contraction length {k} is not a multiple of 16")
                p = torch.empty((k // 8, npad, 8), device=w.device, dtype=torch.float16)
                call("w2e_pack_kq_h", ptr(_c(wd)), p.data_ptr(), n, npad, k, wd.stride(0), int(transposed), stream_ptr())
            else:
                p = torch.empty((k // 4, npad, 4), device=w.device, dtype=torch.float32)
                call("w2e_pack_kq", ptr(_c(wd)), ptr(p), n, npad, k, wd.stride(0), int(transposed), stream_ptr())
            hit = (key, (p, n, k, npad, self.half))
            self.cache[(id(w), transposed)] = hit
        return hit[1]

    def fwd(self, w):
        return self._get(w, False)

    def bwd(self, w):
        return self._get(w, True)


_PK_SPLITS = {}


def _gemm_pk(a_packed, m, mpad, wpk):
    """[S, m, n] slabs = A x W^T on w2e_gemm_pk: both operands K-quad-major, one independent wave per (32 rows, 64 columns, K slice)."""
    wp, n, k, npad, half = wpk
    key = (m, n, k, a_packed.device.index, half)  # (the plan depends on the CU count of the device the launch goes to)
    if key not in _PK_SPLITS:
        from . import _lib
        _PK_SPLITS[key] = (_lib.load().w2e_gemm_pk_h_splits if half else _lib.load().w2e_gemm_pk_splits)(m, n, k)
    sp = _PK_SPLITS[key]
    c = torch.empty((sp, m, n), device=a_packed.device, dtype=torch.float32)
    prof = profiling.span("vit_gemm", 2.0 * m * n * k)
    if half:  # (A stays the fp32 packed operand its producer wrote: rounded to fp16 in the kernel's registers)
        call("w2e_gemm_pk_h", ptr(a_packed), wp.data_ptr(), ptr(c), m, n, k, mpad, npad, n, sp, stream_ptr())
    else:
        call("w2e_gemm_pk", ptr(a_packed), ptr(wp), ptr(c), m, n, k, mpad, npad, n, sp, stream_ptr())
    if prof is not None:
        prof.end()
    return c


class _TransformerV3(torch.autograd.Function):
    """All residual blocks of the visual tower as ONE autograd node on the M = 50*batch kernels: per block 7 launches forward (reduce+LN,
    QKV GEMM, attention, out-proj GEMM, reduce+LN, c_fc GEMM, reduce + QuickGELU pair -- then c_proj's GEMM opens the next block's
    reduce+LN) and 7 backward.  The four GEMMs run on w2e_gemm_pk: every tensor a GEMM consumes is WRITTEN K-quad-major by its producer
    (LayerNorm, attention, the QuickGELU pair and their backward counterparts take a `packed_rows` argument), the weights are packed
    once; split-K slabs are summed by their consumers in a fixed order, never added atomically."""

    @staticmethod
    def forward(ctx, x, heads, wpk, blocks):
        b, l, dim = x.shape
        m = b * l
        mpad = _pad(m, 32)
        x2 = _c(x).reshape(m, dim)
        saved = []
        pend, pbias, pres = x2.view(1, m, dim), None, None
        dev = x.device
        for blk in blocks:
            p = _block_params(blk)
            _frozen(*p)
            ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, fc_w, fc_b, proj_w, proj_b = p
            xr, y1, mean1, rstd1 = _reduce_ln(pend, pbias, pres, ln1_w, ln1_b, blk.ln_1.eps, mpad=mpad)
            qkv = _gemm_pk(y1, m, mpad, wpk.fwd(in_w))
            att = torch.empty((dim // 4, mpad, 4), device=dev, dtype=torch.float32)
            call("w2e_attn2_fwd", ptr(qkv), qkv.shape[0], m * 3 * dim, ptr(in_b), ptr(att), b, l, heads, mpad, stream_ptr())
            o = _gemm_pk(att, m, mpad, wpk.fwd(out_w))
            x_mid, y2, mean2, rstd2 = _reduce_ln(o, out_b, xr, ln2_w, ln2_b, blk.ln_2.eps, mpad=mpad)
            hp = _gemm_pk(y2, m, mpad, wpk.fwd(fc_w))
            n_fc = fc_w.shape[0]
            h = torch.empty((m, n_fc), device=dev, dtype=torch.float32)
            g = torch.empty((n_fc // 4, mpad, 4), device=dev, dtype=torch.float32)
            call("w2e_reduce_gelu", ptr(hp), hp.shape[0], m * n_fc, ptr(fc_b), None, ptr(h), ptr(g), m, n_fc, 0, mpad, stream_ptr())
            pend = _gemm_pk(g, m, mpad, wpk.fwd(proj_w))
            pbias, pres = proj_b, x_mid
            saved.append((xr, mean1, rstd1, qkv, x_mid, mean2, rstd2, h))
        out, _, _, _ = _reduce_ln(pend, pbias, pres, None, None, 0.0, want_y=False)
        ctx.save_for_backward(*[t for blk_saved in saved for t in blk_saved])
        ctx.blocks, ctx.wpk, ctx.geom = blocks, wpk, (b, l, dim, heads)
        return out.reshape(b, l, dim)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        b, l, dim, heads = ctx.geom
        m = b * l
        mpad = _pad(m, 32)
        wpk = ctx.wpk
        g = _c(gout).reshape(m, dim)
        dev = g.device
        gp = torch.empty((dim // 4, mpad, 4), device=dev, dtype=torch.float32)  # the one stand-alone packing pass of a step
        call("w2e_pack_kq", ptr(g), ptr(gp), m, mpad, dim, dim, 0, stream_ptr())
        flat = ctx.saved_tensors
        saved = [flat[8 * i:8 * i + 8] for i in range(len(ctx.blocks))]
        for blk, (xr, mean1, rstd1, qkv, x_mid, mean2, rstd2, h) in zip(reversed(ctx.blocks), reversed(saved)):
            ln1_w, _, in_w, in_b, out_w, _, ln2_w, _, fc_w, _, proj_w, _ = _block_params(blk)
            n_fc = fc_w.shape[0]
            ghp = _gemm_pk(gp, m, mpad, wpk.bwd(proj_w))                                  # through c_proj ...
            gh = torch.empty((n_fc // 4, mpad, 4), device=dev, dtype=torch.float32)
            call("w2e_reduce_gelu", ptr(ghp), ghp.shape[0], m * n_fc, None, ptr(h), ptr(gh), None, m, n_fc, 1, mpad, stream_ptr())  # ... and QuickGELU'
            gy2 = _gemm_pk(gh, m, mpad, wpk.bwd(fc_w))
            g_mid, g_mid_p = _ln_bwd_part(gy2, x_mid, ln2_w, mean2, rstd2, g, mpad=mpad)   # through ln_2, + the residual branch
            ga = _gemm_pk(g_mid_p, m, mpad, wpk.bwd(out_w))
            gqkv = torch.empty((3 * dim // 4, mpad, 4), device=dev, dtype=torch.float32)
            call("w2e_attn2_bwd", ptr(qkv), qkv.shape[0], m * 3 * dim, ptr(in_b), ptr(ga), ga.shape[0], m * dim, ptr(gqkv), b, l, heads,
                 mpad, stream_ptr())
            gy1 = _gemm_pk(gqkv, m, mpad, wpk.bwd(in_w))
            g, gp = _ln_bwd_part(gy1, xr, ln1_w, mean1, rstd1, g_mid, mpad=mpad)           # through ln_1, + the residual branch
        return g.reshape(b, l, dim), None, None, None


def _block_params(blk):
    return (blk.ln_1.weight, blk.ln_1.bias, blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.attn.out_proj.weight,
            blk.attn.out_proj.bias, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, blk.mlp.c_proj.weight,
            blk.mlp.c_proj.bias)


def _v2_ok(vit, width):
    return width in (512, 768, 1024) and width == vit.heads * 64 and not os.environ.get("W2E_VIT_V1")


def vision_forward(vit, image):
    """VisionTransformer.forward: patch embed (a GEMM on the re-laid-out image) -> [cls; patches] + pos ->
    ln_pre -> 12 blocks -> ln_post(cls) @ proj."""
    w = vit.conv1.weight
    width = w.shape[0]
    patches = patchify(image, vit.patch_size)  # [B, 49, 3072]: a pure re-layout (torch copy kernel)
    x = linear(patches, w.reshape(width, -1))
    b = x.shape[0]
    x = torch.cat([vit.class_embedding.view(1, 1, width).expand(b, 1, width), x], dim=1) + vit.positional_embedding
    x = layer_norm(x, vit.ln_pre)
    if _v2_ok(vit, width):
        half = getattr(vit, "_w2e_precision", "f32") == "f16"
        if not hasattr(vit, "_wpk") or vit._wpk.half != half:
            vit._wpk = _WeightsPk(half)
        x = _TransformerV3.apply(x, vit.heads, vit._wpk, list(vit.transformer.resblocks))
    else:  # widths the M = 50*batch kernels are not instantiated for (the tests' tiny tower): first-generation kernels
        arena = None if os.environ.get("W2E_TUNE_NO_ARENA") else _ZeroArena(len(vit.transformer.resblocks), x.device)
        for blk in vit.transformer.resblocks:
            x = resblock_forward(blk, x, vit.heads, arena)
    x = layer_norm(x[:, 0, :], vit.ln_post)
    return linear(x, vit.proj.t())  # [B,768] x [768,512]


# ---------------------------------------------------------------------------------------------- the scalar tail of a step
class _ClipLogits(torch.autograd.Function):
    """exp(logit_scale) * cos(image features, text features) [B,T] -- or, `similarity`, the loss's 1 - that / 100 -- as ONE launch
    (w2e_clip_logits_fwd: the two normalisations, the scale and the product; criteria/clip_loss.py:16 + the tail of clip.model.CLIP.forward),
    and one for the gradient to the image features.  Text features and the scale are constants here (CLIP is a frozen critic)."""

    @staticmethod
    def forward(ctx, feat, text_feat, logit_scale, similarity):
        feat, text_feat = _c(feat), _c(text_feat.detach())
        b, d = feat.shape
        t = text_feat.shape[0]
        ls = logit_scale.detach().reshape(1)
        out = torch.empty((b, t), device=feat.device, dtype=torch.float32)
        call("w2e_clip_logits_fwd", ptr(feat), ptr(text_feat), ptr(ls), ptr(out), b, t, d, int(similarity), stream_ptr())
        ctx.save_for_backward(feat, text_feat, ls)
        ctx.similarity = bool(similarity)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        feat, text_feat, ls = ctx.saved_tensors
        b, d = feat.shape
        gf = torch.empty_like(feat)
        call("w2e_clip_logits_bwd", ptr(_c(gout)), ptr(feat), ptr(text_feat), ptr(ls), ptr(gf), b, text_feat.shape[0], d,
             int(ctx.similarity), stream_ptr())
        return gf, None, None, None


def clip_logits(feat, text_feat, logit_scale, similarity=False):
    return _ClipLogits.apply(feat, text_feat, logit_scale, similarity)


def clip_logits_ok(feat, text_feat, logit_scale):
    """The fused tail applies when only the image features carry a gradient (the path's case) and everything is fp32 on the GPU."""
    return (feat.is_cuda and feat.dtype == torch.float32 and feat.ndim == 2 and text_feat.ndim == 2 and not text_feat.requires_grad
            and feat.shape[1] == text_feat.shape[1] and text_feat.is_cuda and text_feat.dtype == torch.float32
            and not logit_scale.requires_grad)


class _StepLoss(torch.autograd.Function):
    """[loss, loss_clip, loss_l2] = [clip_lambda * mean(sim) + l2_lambda * MSE(w_hat, w), mean(sim), MSE(w_hat, w)] in one launch, their
    gradients to sim and w_hat in one (mapper/training/coach.py:223-245 without the id term).  The two loss terms are returned for the
    log only: their gradient is not propagated (the reference detaches what it logs)."""

    @staticmethod
    def forward(ctx, sim, w_hat, w, clip_lambda, l2_lambda):
        if w.shape != w_hat.shape or w.dtype != w_hat.dtype or w.device != w_hat.device:
            # (the kernel walks both with w_hat's element count; nn.MSELoss would broadcast or raise here -- never read past `w`)
            raise ValueError(f"step_loss: w {tuple(w.shape)} {w.dtype} {w.device} and w_hat {tuple(w_hat.shape)} {w_hat.dtype} "
                             f"{w_hat.device} must have the same shape, dtype and device")
        sim, w_hat, w = _c(sim), _c(w_hat), _c(w.detach())
        out = torch.empty(3, device=w_hat.device, dtype=torch.float32)
        call("w2e_step_loss_fwd", ptr(sim), sim.numel(), ptr(w_hat), ptr(w), w_hat.numel(), float(clip_lambda), float(l2_lambda),
             ptr(out), stream_ptr())
        ctx.save_for_backward(w_hat, w)
        ctx.cfg = (tuple(sim.shape), float(clip_lambda), float(l2_lambda))
        loss, l_clip, l_l2 = out[0], out[1], out[2]  # (three 0-dim outputs: no select node -- and no zero-filled select gradient -- on the tape)
        ctx.mark_non_differentiable(l_clip, l_l2)
        ctx.set_materialize_grads(False)
        return loss, l_clip, l_l2

    @staticmethod
    @once_differentiable
    def backward(ctx, gout, _g1=None, _g2=None):
        if gout is None:
            return None, None, None, None, None
        w_hat, w = ctx.saved_tensors
        sim_shape, cl, l2 = ctx.cfg
        g0 = _c(gout).reshape(1)
        n_sim = 1
        for v in sim_shape:
            n_sim *= v
        g_sim = torch.empty(sim_shape, device=w_hat.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        g_what = torch.empty_like(w_hat) if ctx.needs_input_grad[1] else None
        call("w2e_step_loss_bwd", ptr(g0), n_sim, ptr(w_hat), ptr(w), w_hat.numel(), cl, l2, ptr(g_sim), ptr(g_what), stream_ptr())
        return g_sim, g_what, None, None, None


def step_loss(sim, w_hat, w, clip_lambda, l2_lambda):
    return _StepLoss.apply(sim, w_hat, w, clip_lambda, l2_lambda)

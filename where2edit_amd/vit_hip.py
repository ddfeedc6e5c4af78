"""The CLIP visual tower on the hand-written kernels (include/w2e_vit.h): autograd Functions for
Linear (+bias, +residual, QuickGELU prologue / derivative epilogue), LayerNorm and the attention core,
and `vision_forward`, the HIP execution of clip_vit.VisionTransformer.forward.

CLIP is a frozen critic here (criteria/clip_loss.py builds it once and never optimises it), so the
Functions return input gradients only; asking for a weight gradient raises."""
import torch
from torch.autograd.function import once_differentiable

from . import profiling
from ._lib import call, ptr, stream_ptr
from .clip_vit import patchify


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _gemm(a, b, trans_b, bias=None, residual=None, a_gelu=False, gelu_grad_aux=None):
    """C[M,N] = epi(pro(A) x B): B is [N,K] (trans_b) or [K,N]."""
    m, k = a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    c = torch.empty((m, n), device=a.device, dtype=torch.float32)
    sp = profiling.span("vit_gemm", 2.0 * m * n * k)
    call("w2e_gemm", ptr(a), ptr(b), ptr(c), m, n, k, a.stride(0), b.stride(0), n, int(trans_b), int(a_gelu),
         ptr(bias), ptr(residual), ptr(gelu_grad_aux), stream_ptr())
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


def resblock_forward(blk, x, heads):
    """ResidualAttentionBlock: x += out_proj(attn(in_proj(ln_1 x))); x += c_proj(QuickGELU(c_fc(ln_2 x)))."""
    qkv = linear(layer_norm(x, blk.ln_1), blk.attn.in_proj_weight, blk.attn.in_proj_bias)
    x = linear(_Attention.apply(qkv, heads), blk.attn.out_proj.weight, blk.attn.out_proj.bias, residual=x)
    h = linear(layer_norm(x, blk.ln_2), blk.mlp.c_fc.weight, blk.mlp.c_fc.bias)
    return _GeluLinear.apply(h, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, x)


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
    for blk in vit.transformer.resblocks:
        x = resblock_forward(blk, x, vit.heads)
    x = layer_norm(x[:, 0, :], vit.ln_post)
    return linear(x, vit.proj.t())  # [B,768] x [768,512]

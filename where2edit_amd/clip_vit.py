"""CLIP ViT-B/32 (the model behind criteria/clip_loss.py:10,16) with the OpenAI-CLIP module/parameter
names (`visual.conv1.weight`, `visual.transformer.resblocks.N.attn.in_proj_weight`, `text_projection`,
`logit_scale`, ...), so OpenAI checkpoints' state_dicts load unchanged.  OpenAI `clip` itself is a
third-party package absent from the reference tree and from this image; the architecture follows the
published `clip/model.py` (VisionTransformer, ResidualAttentionBlock, QuickGELU, LayerNorm, CLIP).

fp32 throughout.  The visual tower runs its LayerNorm / QKV / attention / projection / MLP on the hand-written
kernels of libw2e.so (include/w2e_vit.h, vit_hip.vision_forward) and has no other execution: a CPU tensor raises.
The text tower (`Transformer` below, stock PyTorch-ROCm ops) runs once on constant tokens and is cached (coach.py:55
tokenises the description once).  A stock-op composition of the visual tower for A/B timing lives in tools/vit_stock.py.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


class LayerNorm(nn.LayerNorm):
    """fp32 LayerNorm, eps 1e-5."""

    def forward(self, x):
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class _SelfAttention(nn.Module):
    """Parameter layout of nn.MultiheadAttention (in_proj_weight [3d,d], in_proj_bias, out_proj)."""

    def __init__(self, d_model, n_head):
        super().__init__()
        self.n_head = n_head
        self.in_proj_weight = nn.Parameter(torch.randn(3 * d_model, d_model) * d_model ** -0.5)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)

    def forward(self, x, causal):
        b, l, d = x.shape
        hd = d // self.n_head
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias).view(b, l, 3, self.n_head, hd).permute(2, 0, 3, 1, 4)
        att = (qkv[0] @ qkv[1].transpose(-1, -2)) / math.sqrt(hd)
        if causal:
            att = att + torch.full((l, l), float("-inf"), device=x.device, dtype=x.dtype).triu(1)
        y = (att.softmax(-1) @ qkv[2]).transpose(1, 2).reshape(b, l, d)
        return self.out_proj(y)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, causal=False):
        super().__init__()
        self.attn = _SelfAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.causal = causal

    def forward(self, x):  # x [B, L, D] (batch-first; OpenAI runs L,B,D -- same math)
        x = x + self.attn(self.ln_1(x), self.causal)
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, causal=False):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, causal) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


def patchify(x, patch):
    """[B,3,H,W] -> [B, (H/p)*(W/p), 3*p*p]: the im2col of a stride-p, kernel-p convolution is a pure re-layout."""
    b, c, h, w = x.shape
    gh, gw = h // patch, w // patch
    return x.reshape(b, c, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, c * patch * patch)


def patch_embed(x, weight):
    """conv1 (kernel = stride = patch, no bias) as a GEMM [B*49, 3072] x [3072, width]."""
    return F.linear(patchify(x, weight.shape[-1]), weight.reshape(weight.shape[0], -1))


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim, self.patch_size, self.heads = input_resolution, output_dim, patch_size, heads
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def forward(self, x):
        from . import vit_hip
        return vit_hip.vision_forward(self, x)

    def set_precision(self, precision):
        """"f32" (default: every GEMM on the exact fp32 MFMA) or the OPT-IN "f16": the four Linear layers of every block take fp16
        operands with fp32 accumulation (w2e_gemm_pk_h) -- the arithmetic of the model the reference loads on a GPU
        (criteria/clip_loss.py:10: `clip.load(..., device="cuda")` is OpenAI's fp16 tower); LayerNorm, attention, GELU and the
        residual stream stay fp32.  Widths the packed-operand kernels are not instantiated for ignore it."""
        if precision not in ("f32", "f16"):
            raise ValueError("precision must be 'f32' or 'f16'")
        self._w2e_precision = precision
        return self


class CLIP(nn.Module):
    """ViT variants of OpenAI CLIP.  Default arguments = "ViT-B/32"."""

    def __init__(self, embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
                 context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12):
        super().__init__()
        self.context_length = context_length
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads, causal=True)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.randn(context_length, transformer_width) * 0.01)
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.randn(transformer_width, embed_dim) * transformer_width ** -0.5)
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))
        self._text_cache = None

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def set_precision(self, precision):
        """The image tower's GEMM operand precision: see VisionTransformer.set_precision (parameters and I/O stay fp32)."""
        self.visual.set_precision(precision)
        return self

    def encode_image(self, image):
        return self.visual(image.to(self.dtype))

    def encode_text(self, text):
        x = self.token_embedding(text) + self.positional_embedding[: text.shape[1]]
        x = self.ln_final(self.transformer(x))
        return x[torch.arange(x.shape[0], device=x.device), text.argmax(dim=-1)] @ self.text_projection

    def encode_text_cached(self, text):
        """The text tower on constant tokens is recomputed every step by the reference (clip_loss.py:16);
        its output cannot change while the weights are frozen, so it is computed once per token tensor."""
        c = self._text_cache
        # fast path: the very same tensor object, unmodified (the cache holds a reference, so its address cannot be
        # recycled for another prompt); otherwise compare contents with the cached copy (one small D2H sync)
        if c is not None and ((c[0] is text and c[1] == text._version) or
                              (c[2].shape == text.shape and c[2].device == text.device and torch.equal(c[2], text))):
            if c[0] is not text or c[1] != text._version:
                self._text_cache = (text, text._version, c[2], c[3])
            return c[3]
        with torch.no_grad():
            self._text_cache = (text, text._version, text.detach().clone(), self.encode_text(text))
        return self._text_cache[3]

    def load_state_dict(self, *args, **kwargs):
        self._text_cache = None  # new weights: the cached text features are stale
        return super().load_state_dict(*args, **kwargs)

    def logits_per_image(self, image, text, similarity=False):
        """forward()[0] -- or, `similarity`, the loss's 1 - that / 100 (criteria/clip_loss.py:16) -- with the normalisations, the scale
        and the product as one launch when only the image carries a gradient (vit_hip.clip_logits); else forward()'s composition."""
        from . import vit_hip
        text_frozen = not any(p.requires_grad for p in self.transformer.parameters())
        image_features = self.encode_image(image)
        text_features = self.encode_text_cached(text) if text_frozen else self.encode_text(text)
        if text_frozen and vit_hip.clip_logits_ok(image_features, text_features, self.logit_scale):
            return vit_hip.clip_logits(image_features, text_features, self.logit_scale, similarity)
        image_features = image_features / image_features.norm(dim=1, keepdim=True)
        text_features = text_features / text_features.norm(dim=1, keepdim=True)
        logits = self.logit_scale.exp() * image_features @ text_features.t()
        return 1 - logits / 100 if similarity else logits

    def forward(self, image, text):
        image_features = self.encode_image(image)
        text_features = self.encode_text_cached(text) if not any(p.requires_grad for p in self.transformer.parameters()) \
            else self.encode_text(text)
        image_features = image_features / image_features.norm(dim=1, keepdim=True)
        text_features = text_features / text_features.norm(dim=1, keepdim=True)
        logits_per_image = self.logit_scale.exp() * image_features @ text_features.t()
        return logits_per_image, logits_per_image.t()

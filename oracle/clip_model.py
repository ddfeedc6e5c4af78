"""Oracle (CPU, test-only): CLIP ViT-B/32 towers + the CLIP loss of
criteria/clip_loss.py:6-17, functional over an OpenAI-CLIP format state_dict.

PARITY WITH THE REFERENCE IS UNPINNED for the towers: OpenAI `clip`
(requirements.txt:31 `clip=1.0`, cog.yaml:22 openai/CLIP@8a665a68) is not under
/root/reference and not installed.  This restates the published architecture
(`clip/model.py` of that commit: VisionTransformer, ResidualAttentionBlock,
QuickGELU, LayerNorm, CLIP.encode_text / forward) and is cross-checked against
`transformers.CLIPModel` (tests/golden/clip_hf_tiny.npz).  fp32 throughout: the
fp16 weights `clip.load(device="cuda")` produces are a storage choice of the
reference run, the stated tolerance is against this fp32 restatement.
"""
import math

import torch
import torch.nn.functional as F

from . import ops


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def _ln(sd, pre, x):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"], sd[pre + ".bias"], 1e-5)


def _attention(x, w_in, b_in, w_out, b_out, heads, causal):
    """nn.MultiheadAttention self-attention, batch-first restatement. x [B,L,D]."""
    b, l, d = x.shape
    hd = d // heads
    qkv = F.linear(x, w_in, b_in).view(b, l, 3, heads, hd).permute(2, 0, 3, 1, 4)  # [3,B,H,L,hd]
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if causal:
        att = att + torch.full((l, l), float("-inf"), dtype=x.dtype).triu(1)
    att = att.softmax(-1)
    y = (att @ v).transpose(1, 2).reshape(b, l, d)
    return F.linear(y, w_out, b_out)


def resblock(sd, pre, x, heads, causal):
    """ResidualAttentionBlock: x += attn(ln_1 x); x += c_proj(QuickGELU(c_fc(ln_2 x)))."""
    h = _ln(sd, pre + ".ln_1", x)
    x = x + _attention(h, sd[pre + ".attn.in_proj_weight"], sd[pre + ".attn.in_proj_bias"],
                       sd[pre + ".attn.out_proj.weight"], sd[pre + ".attn.out_proj.bias"], heads, causal)
    h = _ln(sd, pre + ".ln_2", x)
    h = quick_gelu(F.linear(h, sd[pre + ".mlp.c_fc.weight"], sd[pre + ".mlp.c_fc.bias"]))
    return x + F.linear(h, sd[pre + ".mlp.c_proj.weight"], sd[pre + ".mlp.c_proj.bias"])


def _n_blocks(sd, pre):
    n = 0
    while f"{pre}.{n}.ln_1.weight" in sd:
        n += 1
    return n


def encode_image(sd, img):
    """VisionTransformer.forward: patch conv (no bias) -> [cls; patches] + pos ->
    ln_pre -> blocks -> ln_post(cls) @ proj.  heads = width // 64."""
    w = sd["visual.conv1.weight"]
    patch = w.shape[-1]
    x = F.conv2d(img, w, stride=patch)
    b, d = x.shape[0], x.shape[1]
    x = x.reshape(b, d, -1).permute(0, 2, 1)
    cls = sd["visual.class_embedding"].view(1, 1, d).expand(b, 1, d)
    x = torch.cat([cls, x], 1) + sd["visual.positional_embedding"]
    x = _ln(sd, "visual.ln_pre", x)
    for i in range(_n_blocks(sd, "visual.transformer.resblocks")):
        x = resblock(sd, f"visual.transformer.resblocks.{i}", x, d // 64, causal=False)
    x = _ln(sd, "visual.ln_post", x[:, 0, :])
    return x @ sd["visual.proj"]


def encode_text(sd, tokens, heads=None):
    """CLIP.encode_text: embed + pos -> causal blocks -> ln_final -> take the
    feature at the EOT token (= argmax token id) @ text_projection."""
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"][: tokens.shape[1]]
    d = x.shape[-1]
    heads = heads or d // 64
    for i in range(_n_blocks(sd, "transformer.resblocks")):
        x = resblock(sd, f"transformer.resblocks.{i}", x, heads, causal=True)
    x = _ln(sd, "ln_final", x)
    x = x[torch.arange(x.shape[0]), tokens.argmax(-1)]
    return x @ sd["text_projection"]


def clip_logits(sd, img, tokens):
    """CLIP.forward -> logits_per_image [B, n_text]."""
    fi = encode_image(sd, img)
    ft = encode_text(sd, tokens)
    fi = fi / fi.norm(dim=1, keepdim=True)
    ft = ft / ft.norm(dim=1, keepdim=True)
    return sd["logit_scale"].exp() * fi @ ft.t()


def clip_loss(sd, image, tokens, stylegan_size):
    """CLIPLoss.forward (criteria/clip_loss.py:14-17): 1 - logits/100 on the
    7x-nearest / avg-pooled image; no mean/std normalisation, no clamp (Q6)."""
    img = ops.clip_preprocess(image, stylegan_size)
    return 1 - clip_logits(sd, img, tokens) / 100

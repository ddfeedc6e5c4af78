#!/usr/bin/env python3
"""A/B aid, not product code: the CLIP visual tower composed from stock PyTorch-ROCm ops (rocBLAS GEMMs, torch softmax /
LayerNorm) on the SAME module and weights, timed against the library's kernels (vit_hip.vision_forward).

    python tools/vit_stock.py [--batch 4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from where2edit_amd.clip_vit import CLIP, patch_embed  # noqa: E402


def stock_vision_forward(vit, x):
    """clip.model.VisionTransformer.forward on stock ops (batch-first)."""
    x = patch_embed(x, vit.conv1.weight)
    b, d = x.shape[0], x.shape[2]
    x = torch.cat([vit.class_embedding.view(1, 1, d).expand(b, 1, d), x], dim=1) + vit.positional_embedding
    x = vit.transformer(vit.ln_pre(x))
    return vit.ln_post(x[:, 0, :]) @ vit.proj


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    args = ap.parse_args()
    m = CLIP().cuda().eval().requires_grad_(False)
    x = torch.randn(args.batch, 3, 224, 224, device="cuda", requires_grad=True)
    for name, fn in (("libw2e kernels", lambda: m.visual(x)), ("stock ops", lambda: stock_vision_forward(m.visual, x))):
        for _ in range(3):
            fn().sum().backward()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            f = fn()
            f.sum().backward()
        b.record()
        torch.cuda.synchronize()
        print(f"ViT-B/32 visual fwd+bwd, batch {args.batch}, {name}: {a.elapsed_time(b) / 10:.3f} ms")
    d = (m.visual(x) - stock_vision_forward(m.visual, x)).abs().max() / stock_vision_forward(m.visual, x).abs().max()
    print(f"max-norm relative difference of the features: {float(d):.2e}")


if __name__ == "__main__":
    main()

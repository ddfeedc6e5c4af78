#!/usr/bin/env python3
"""How far the opt-in W2E_CONV_PRECISION=bf16x3 moves the results of the exact fp32 path: the 1024^2 generator image, the
loss of a mapper step and the mapper gradients, same seeded weights and latents (tests/golden/seeded.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(mode):
    if mode == "bf16x3":
        _lib.set_option("conv_precision", "bf16x3")
    else:
        _lib.set_option("conv_precision", "f32")
    torch.manual_seed(0)
    coach = bench.build_coach(1024, 2, "cuda:0", False, "hip", 2)
    w = bench.synthetic_latents(coach.net.decoder, 2, 0)
    coach.optimizer.zero_grad()
    x, x_hat, w_hat = coach.forward_pair(w)
    loss, d = coach.calc_loss(w, x, w_hat, x_hat)
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in coach.net.mapper.parameters()]).double()
    return x.detach().double(), x_hat.detach().double(), float(loss.detach()), grads


xa, xha, la, ga = run("f32")
xb, xhb, lb, gb = run("bf16x3")
rel = lambda a, b: ((a - b).abs().max() / a.abs().max()).item()
print(f"image G(w):      max |diff| / max |ref| = {rel(xa, xb):.3e}")
print(f"image G(w_hat):  max |diff| / max |ref| = {rel(xha, xhb):.3e}")
print(f"loss: {la:.9f} vs {lb:.9f}  (rel {abs(la - lb) / abs(la):.3e})")
print(f"mapper gradient: max |diff| / max |ref| = {rel(ga, gb):.3e}, cosine = {torch.nn.functional.cosine_similarity(ga, gb, dim=0).item():.9f}")

#!/usr/bin/env python3
"""ViT-B/32 visual tower forward+backward time at batch 4 (HIP kernels), per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from where2edit_amd.clip_vit import CLIP
m = CLIP().cuda().eval()
for p in m.parameters():
    p.requires_grad_(False)
x = torch.randn(4, 3, 224, 224, device="cuda", requires_grad=True)
def step():
    f = m.encode_image(x)
    f.sum().backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    step()
b.record()
torch.cuda.synchronize()
print(f"ViT-B/32 fwd+bwd, batch 4: {a.elapsed_time(b) / 10:.3f} ms")

#!/usr/bin/env python3
"""Memset / memcpy operations of one eager mapper step (they become memset / memcpy NODES under hipGraph capture; on this ROCm
stack a captured hipMemsetAsync was observed not to be replayed, so a step that is to be captured should contain none that
matter).  usage: graph_safety.py [workload batch]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

WL = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
coach = bench.build_coach(1024, B, "cuda:0", False, "hip", WL)
w = bench.synthetic_latents(coach.net.decoder, B, 0)
mask = bench.make_mask(coach, B, 1024, 0, "cuda:0", False) if WL == 3 else None
for _ in range(3):
    coach.train_step(w, mask)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], record_shapes=True) as prof:
    coach.optimizer.zero_grad()
    x, x_hat, w_hat = coach.forward_pair(w, mask)
    loss, d = coach.calc_loss(w, x, w_hat, x_hat)
    loss.backward()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    for k in (e.kernels or []):
        if "emcpy" in k.name or "copyBuffer" in k.name or "emset" in k.name or "fillBuffer" in k.name:
            par, chain = e, []
            while par is not None and len(chain) < 4:
                chain.append(par.name)
                par = par.cpu_parent
            cnt[(k.name[:32], " <- ".join(chain), str(e.input_shapes)[:70])] += 1
print(f"# memset / memcpy operations in zero_grad + forward + losses + backward (workload {WL}, batch {B}): {sum(cnt.values())}")
for (k, chain, shp), n in cnt.most_common(60):
    print(n, k, "|", chain, "|", shp)

#!/usr/bin/env python3
"""Which host lines launch the stock (non-libw2e) kernels of one eager mapper step: count and GPU time per source line.

usage: op_census.py [workload batch]   (one GPU; eager step, torch.profiler with stacks)"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

WL = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
STEPS = 3
coach = bench.build_coach(1024, B, "cuda:0", False, "hip", WL)
w = bench.synthetic_latents(coach.net.decoder, B, 0)
mask = bench.make_mask(coach, B, 1024, 0, "cuda:0", False) if WL == 3 else None
for _ in range(3):
    coach.train_step(w, mask)
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

VIEW_OPS = ("view", "reshape", "expand", "slice", "select", "t.default", "transpose", "permute", "unsqueeze", "squeeze", "detach",
            "alias", "as_strided", "unbind", "split", "chunk", "narrow", "empty", "_unsafe_view", "sym_", "stride", "size", "numel",
            "is_", "dim", "result_type", "_local_scalar", "item", "unfold", "lift_fresh", "set_", "resize_")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_line = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(v) or ("." + v) in name for v in VIEW_OPS):
            where = "(engine)"
            for fr in reversed(traceback.extract_stack()):
                if "where2edit_amd/" in fr.filename and "op_census" not in fr.filename:
                    where = f"{fr.filename.split('where2edit_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            self.by_line[(where, name)] += 1
        return func(*args, **(kwargs or {}))


census = Census()
with census:
    coach.train_step(w, mask)
torch.cuda.synchronize()
print(f"# aten ops dispatched in one eager step (views dropped): {sum(census.by_line.values())} (workload {WL}, batch {B})")
for (where, op), n in sorted(census.by_line.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:5d}x  {op:36.36s} {where}")

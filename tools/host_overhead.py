#!/usr/bin/env python3
"""Host-side enqueue time of one mapper step vs its GPU time (is the step launch-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

WL = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
coach = bench.build_coach(1024, B, "cuda:0", False, "hip", WL)
w = bench.synthetic_latents(coach.net.decoder, B, 0)
mask = torch.rand(B, 1, 64, 64, device="cuda:0") if WL == 3 else None
_step = coach.train_step
coach.train_step = lambda w_: _step(w_, mask)
for _ in range(3):
    coach.train_step(w)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    coach.train_step(w)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0), total.append(t2 - t0)
print(f"host enqueue {1e3 * sum(host) / 10:.1f} ms/step, step wall {1e3 * sum(total) / 10:.1f} ms (cpu count {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))})")

#!/usr/bin/env python3
"""Host-side enqueue time of one mapper step vs its GPU time (is the step launch-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

coach = bench.build_coach(1024, 4, "cuda:0", False, "hip")
w = bench.synthetic_latents(coach.net.decoder, 4, 0)
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

#!/usr/bin/env python3
"""RCCL self-test on ONE GPU: a world-size-1 nccl process group (its watchdog thread included), the data-parallel Coach with the
flat gradient bucket, eager steps, then the hipGraph-captured step with the all-reduce after each replay -- what
`bench.py --gpus N` runs per rank; with a third argument `pipeline`, also BASELINE configs[4]'s inference pipeline captured and replayed
while the group is alive (what `bench.py --workload 5 --gpus N` does per rank).  usage: dp_rccl_selftest.py [size batch [pipeline]]"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import socket  # noqa: E402

_s = socket.socket()
_s.bind(("127.0.0.1", 0))  # (a free port: a fixed one collides with whatever else rendezvous on the box)
_port = _s.getsockname()[1]
_s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
SIZE = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
from where2edit_amd import dist as wd  # noqa: E402

wd.init_from_env(backend="nccl", timeout_s=300, force_group=True)  # the very call bench.py makes (device_id, timeout)
import bench  # noqa: E402

coach = bench.build_coach(SIZE, B, "cuda:0", True, "hip")
w = bench.synthetic_latents(coach.net.decoder, B, 0)
for _ in range(3):
    d = coach.train_step(w)
dist.all_reduce(coach.bucket.flat)  # (GradBucket.all_reduce_mean skips the collective at world size 1: issue it here)
dist.barrier()
torch.cuda.synchronize()
print("RCCL world-1 DP eager step ok, loss", float(d["loss"]), "bucket MB", coach.bucket.nbytes / 1e6, flush=True)
step = coach.capture_step(w)
for _ in range(5):
    d = step(w)
dist.barrier()
torch.cuda.synchronize()
loss = float(d["loss"])
assert loss == loss, "NaN loss after graph replays"
print("RCCL world-1 DP graphed step ok, loss", loss, flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "pipeline":
    from where2edit_amd.demo_pipeline import capture_invert_and_edit, invert_and_edit  # noqa: E402
    del coach, step
    torch.cuda.empty_cache()
    imgs, e4e, g, clip, net, text, att = bench.build_config5("cuda:0", 1, 0)
    eager = invert_and_edit(imgs, e4e, g, clip, net, text, att, attention_layer=13)
    run = capture_invert_and_edit(torch.zeros_like(imgs), e4e, g, clip, net, torch.zeros_like(text), torch.zeros_like(att), attention_layer=13)
    dist.barrier()
    rep = run(imgs, text, att)
    torch.cuda.synchronize()
    for key in ("img_orig", "mask", "img_gen", "features_gen"):
        err = float((rep[key] - eager[key]).abs().max() / eager[key].abs().max().clamp_min(1e-30))
        assert err <= 1e-5, (key, err)
    print("RCCL world-1 pipeline captured with the group alive and replayed ok", flush=True)
dist.destroy_process_group()

#!/usr/bin/env python3
"""Per-layer microbenchmark of the modulated-conv kernels at the FFHQ-1024 generator's layer shapes
(HIP events on the launch stream, random data).  python tools/layer_bench.py [--batch 4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from where2edit_amd import functional as K  # noqa: E402

LAYERS = [  # cin, cout, input res, upsample
    (512, 512, 4, False), (512, 512, 4, True), (512, 512, 8, False), (512, 512, 8, True), (512, 512, 16, False),
    (512, 512, 16, True), (512, 512, 32, False), (512, 512, 32, True), (512, 512, 64, False), (512, 256, 64, True),
    (256, 256, 128, False), (256, 128, 128, True), (128, 128, 256, False), (128, 64, 256, True), (64, 64, 512, False),
    (64, 32, 512, True), (32, 32, 1024, False)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", type=str, default="", help="comma list of layer indices")
    ap.add_argument("--warm", type=float, default=0.0, help="seconds of GPU load before the first measurement (a fresh box "
                    "starts at idle clocks: short runs of the first layers read low without it)")
    args = ap.parse_args()
    if args.warm > 0:
        import time
        a = torch.randn(4096, 4096, device="cuda")
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.warm:
            (a @ a).sum().item()
    B = args.batch
    dev = "cuda"
    tot = {"fwd": [0.0, 0.0], "bwd": [0.0, 0.0]}
    print(f"{'layer':28s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF/s':>7s} | {'dgrad ms':>8s} {'TF/s':>7s}")
    sel = [int(i) for i in args.only.split(',')] if args.only else range(len(LAYERS))
    for cin, cout, h, up in [LAYERS[i] for i in sel]:
        w = torch.randn(cout, cin, 3, 3, device=dev)
        scale = (cin * 9) ** -0.5
        fwd = K.conv_pack(w, scale, False, False)
        bwd = K.conv_pack(w, scale, True, not up)
        x = torch.randn(B, cin, h, h, device=dev)
        s = torch.randn(B, cin, device=dev)
        d = torch.rand(B, cout, device=dev) + 0.5
        oh = 2 * h if up else h
        noise = torch.randn(1, 1, oh, oh, device=dev)
        nw = torch.randn(1, device=dev)
        bias = torch.randn(cout, device=dev)
        flop = 2.0 * B * cin * cout * 9 * h * h
        if up:
            f = lambda: K._modconv_raw(K.MODE_UP, x, fwd, s, d, h, h)
            g = torch.randn(B, cout, 2 * h + 1, 2 * h + 1, device=dev)
            r = lambda: K._modconv_raw(K.MODE_DOWN, g, bwd, d, s, h, h, dot_with=x)
        else:
            f = lambda: K._modconv_raw(K.MODE_SAME, x, fwd, s, d, h, h, act=(noise, nw, bias))
            g = torch.randn(B, cout, h, h, device=dev)
            r = lambda: K._modconv_raw(K.MODE_SAME, g, bwd, d, s, h, h, dot_with=x)
        tf, tb = timeit(f, args.iters), timeit(r, args.iters)
        tot["fwd"][0] += tf; tot["fwd"][1] += flop; tot["bwd"][0] += tb; tot["bwd"][1] += flop
        name = f"{cin:3d}->{cout:3d} @{h:4d} {'up  ' if up else 'same'}"
        print(f"{name:28s} {flop / 1e9:8.2f} | {tf:8.3f} {flop / tf / 1e9:7.1f} | {tb:8.3f} {flop / tb / 1e9:7.1f}")
    for k, (ms, fl) in tot.items():
        print(f"total {k}: {ms:.3f} ms, {fl / 1e9:.1f} GFLOP, {fl / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()

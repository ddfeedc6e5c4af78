#!/usr/bin/env python3
"""Every IR-SE50 3x3 layer shape of BASELINE configs[2] (batch 16 = 2 x 8 faces at 112^2, and the e4e encoder's batch-8 shapes):
the library's automatic tile / split-K choice against an exhaustive sweep of `tune_cfg` (all tile configurations x split-K 1..8).

    python tools/irse_shapes.py > profiles/rNN_irse_shapes.txt
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from where2edit_amd import functional as K, irse_hip as I, _lib
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
shapes = [(16, 64, 56), (16, 128, 28), (16, 256, 14), (16, 512, 7), (8, 256, 14), (8, 256, 32), (8, 512, 16)]
for B, C, H in shapes:
    w = torch.randn(C, C, 3, 3, device="cuda")
    wp = K.conv_pack(w, 1.0, False, False)
    x = torch.randn(B, C, H, H, device="cuda")
    bias, slope = torch.randn(C, device="cuda"), torch.rand(C, device="cuda")
    flop = 2.0 * B * C * C * 9 * H * H
    res = []
    for prelu in (True, False):
        base = timeit(lambda: I.conv3x3(x, wp, C, H, H, bias=bias if prelu else None, slope=slope if prelu else None))
        res.append(f"{'prelu' if prelu else 'plain'} auto {base:6.1f}us ({flop / base / 1e6:5.1f} TF/s)")
        best = (1e9, None)
        for cfg in range(0, 11):
            for sp in (1, 2, 4, 8):
                _lib.set_option("tune_cfg", f"{cfg},{sp},0")
                try:
                    t = timeit(lambda: I.conv3x3(x, wp, C, H, H, bias=bias if prelu else None, slope=slope if prelu else None), 10)
                except RuntimeError:
                    continue
                if t < best[0]: best = (t, (cfg, sp))
        _lib.set_option("tune_cfg", "")
        res.append(f"best cfg {best[1]} {best[0]:6.1f}us ({flop / best[0] / 1e6:5.1f} TF/s)")
    print(f"B{B} C{C} {H}x{H} {flop/1e9:5.2f} GF: " + " | ".join(res), flush=True)

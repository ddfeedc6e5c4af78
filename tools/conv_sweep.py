#!/usr/bin/env python3
"""The library's automatic tile / split-K / all-phase choice for the generator's UP layers (forward at batch 8 = the merged pass) and their
stride-2 adjoints with the fused dot (batch 4) against an exhaustive sweep of tune_cfg x splits x tune_upall.

    python tools/conv_sweep.py > profiles/rNN_conv_sweep.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from where2edit_amd import _lib, functional as K  # noqa: E402

LAYERS = [(512, 512, 8), (512, 512, 16), (512, 512, 32), (512, 256, 64), (256, 128, 128), (128, 64, 256), (64, 32, 512)]


def timeit(fn, iters=8):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


warm = torch.randn(4096, 4096, device="cuda")
for _ in range(30):
    (warm @ warm).sum().item()
for cin, cout, h in LAYERS:
    for which, B in (("up fwd", 8), ("down+dot bwd", 4)):
        w = torch.randn(cout, cin, 3, 3, device="cuda")
        scale = (cin * 9) ** -0.5
        fwd, bwd = K.conv_pack(w, scale, False, False), K.conv_pack(w, scale, True, False)
        x = torch.randn(B, cin, h, h, device="cuda")
        s, d = torch.randn(B, cin, device="cuda"), torch.rand(B, cout, device="cuda") + 0.5
        if which == "up fwd":
            run = lambda: K._modconv_raw(K.MODE_UP, x, fwd, s, d, h, h)
            mode = 1
        else:
            g = torch.randn(B, cout, 2 * h + 1, 2 * h + 1, device="cuda")
            run = lambda: K._modconv_raw(K.MODE_DOWN, g, bwd, d, s, h, h, dot_with=x)
            mode = 2
        flop = 2.0 * B * cin * cout * 9 * h * h
        auto = timeit(run, 20)
        best = (auto, "auto")
        for upall in ((0, 1) if mode == 1 else (-1,)):
            _lib.set_option("tune_upall", "" if upall < 0 else str(upall))
            for cfg in range(12 if (mode == 1 and upall == 1) else 11):  # (tile 11 exists in the all-phase UP form only)
                for sp in (1, 2, 4, 8):
                    _lib.set_option("tune_cfg", f"{cfg},{sp},{mode}")
                    try:
                        t = timeit(run, 4)
                    except RuntimeError:
                        continue
                    if t < best[0]:
                        best = (t, f"cfg {cfg} x{sp}" + (f" upall {upall}" if mode == 1 else ""))
        _lib.set_option("tune_cfg", "")
        _lib.set_option("tune_upall", "")
        def forced(tag):
            if tag == "auto":
                _lib.set_option("tune_cfg", "")
                _lib.set_option("tune_upall", "")
                return
            cfg, sp = tag.split()[1], tag.split()[2][1:]
            _lib.set_option("tune_upall", tag.split()[-1] if mode == 1 else "")
            _lib.set_option("tune_cfg", f"{cfg},{sp},{mode}")

        # the verdict: auto and the sweep's winner timed ALTERNATELY, 3 x 20 launches each (a first measurement right after the
        # allocations reads up to 10 % high: clocks and caches), best of three
        ta, tb = [], []
        for _ in range(3):
            forced("auto")
            ta.append(timeit(run, 20))
            forced(best[1])
            tb.append(timeit(run, 20))
        forced("auto")
        ta, tb = min(ta), min(tb)
        print(f"{cin:3d}->{cout:3d} @{h:4d} {which:13s} B{B}: auto {ta * 1e3:7.1f} us ({flop / ta / 1e9:6.1f} TF/s) | sweep winner {tb * 1e3:7.1f} us "
              f"({flop / tb / 1e9:6.1f}) {best[1]}  {'<-- ' + format((ta - tb) * 1e3, '.1f') + ' us' if tb < 0.98 * ta else ''}", flush=True)

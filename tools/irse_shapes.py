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
# (batch, cin, cout, size): IR-SE50 at 112^2 (id loss: batch 16 = 8 generated + 8 originals; its backward: batch 8) and at 256^2 (e4e, batch 8)
shapes = [(16, 64, 64, 112), (16, 64, 64, 56), (16, 64, 128, 56), (16, 128, 128, 28), (16, 128, 256, 28), (16, 256, 256, 14), (16, 256, 512, 14),
          (16, 512, 512, 7), (8, 64, 64, 56), (8, 128, 128, 28), (8, 256, 256, 14),
          (8, 64, 64, 256), (8, 64, 64, 128), (8, 64, 128, 128), (8, 128, 128, 64), (8, 128, 256, 64), (8, 256, 256, 32), (8, 256, 512, 32), (8, 512, 512, 16)]
sweep = "--sweep" in sys.argv  # also the exhaustive tile / split-K sweep of the direct kernel (slow)
for B, Ci, C, H in shapes:
    w = torch.randn(C, Ci, 3, 3, device="cuda")
    wp = K.conv_pack(w, 1.0, False, False)
    x = torch.randn(B, Ci, H, H, device="cuda")
    bias, slope = torch.randn(C, device="cuda"), torch.rand(C, device="cuda")
    flop = 2.0 * B * C * Ci * 9 * H * H
    res = []
    run = lambda form: I.conv3x3(x, wp, C, H, H, bias=bias, slope=slope, form=form)
    base = timeit(lambda: run(0))
    res.append(f"direct {base:6.1f}us ({flop / base / 1e6:5.1f} TF/s)")
    if K._gemm_shape_ok(B, Ci, C, H, H, dot=False, ragged=True):
        t = timeit(lambda: run(4))
        res.append(f"F(4x4) gemm {t:6.1f}us ({flop / t / 1e6:5.1f})")
    if K._fused_shape_ok(B, Ci, C, H, H):
        t = timeit(lambda: run(K.FUSED))
        res.append(f"fused {t:6.1f}us ({flop / t / 1e6:5.1f})")
    res.append(f"chosen: {({0: 'direct', K.FUSED: 'fused'}).get(I._wino_form(B, Ci, C, H, H), I._wino_form(B, Ci, C, H, H))}")
    if sweep:
        best = (1e9, None)
        for cfg in range(0, 11):
            for sp in (1, 2, 4, 8):
                _lib.set_option("tune_cfg", f"{cfg},{sp},0")
                try:
                    t = timeit(lambda: run(0), 10)
                except RuntimeError:
                    continue
                if t < best[0]: best = (t, (cfg, sp))
        _lib.set_option("tune_cfg", "")
        res.append(f"best direct cfg {best[1]} {best[0]:6.1f}us ({flop / best[0] / 1e6:5.1f} TF/s)")
    print(f"B{B} {Ci}->{C} {H}x{H} {flop/1e9:6.2f} GF: " + " | ".join(res), flush=True)

#!/usr/bin/env python3
"""w2e_gemm_fm per shape of the ViT-B/32 tower at M = 50*batch: microseconds per launch for every admissible K-split
(HIP events over back-to-back launches, L2-warm), next to w2e_gemm_fm_splits' choice and the first-generation kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from where2edit_amd import vit_hip as V  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
M = 50 * B
SHAPES = [("qkv", 2304, 768), ("out_proj / out^T", 768, 768), ("c_fc / proj^T", 3072, 768), ("c_proj / fc^T", 768, 3072), ("in^T", 768, 2304)]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for name, n, k in SHAPES:
    a = torch.randn(M, k, device="cuda")
    w = torch.randn(n, k, device="cuda")
    steps = k // 32
    res = []
    for sp in range(1, 17):
        per = -(-steps // sp)
        if -(-steps // per) != sp or (sp > 1 and per < 2):
            continue
        t = timeit(lambda: V._gemm_fm(a, w, V.EPI_PARTIAL, splits=sp))
        res.append((sp, t))
    old = timeit(lambda: V._gemm(a, w, True))
    pick = V._fm_splits(M, n, k)
    gf = 2.0 * M * n * k / 1e9
    print(f"{name:18s} M={M} N={n} K={k} ({gf:.2f} GF)  pick x{pick}  v1 {old:6.1f} us | " + "  ".join(f"x{sp}:{t:5.1f}" for sp, t in res), flush=True)

#!/usr/bin/env python3
"""The ViT-B/32 GEMM shapes of one step at batch 4 (M = 200): wall time per call of w2e_gemm vs torch (hipBLASLt).
Back-to-back calls of 10-25 us kernels are partly host-bound (ctypes + allocation ~15 us per call): for kernel durations
use rocprofv3 --kernel-trace on tools/vit_bench.py; this tool is for relative comparisons and split-K sweeps
(W2E_TUNE_GEMM_S=<splits>)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from where2edit_amd import vit_hip  # noqa: E402

SHAPES = [  # name, m, n, k, trans_b, kwargs
    ("qkv fwd", 200, 2304, 768, True, {}), ("out_proj fwd", 200, 768, 768, True, {"res": True}),
    ("c_fc fwd", 200, 3072, 768, True, {}), ("c_proj fwd", 200, 768, 3072, True, {"a_gelu": True, "res": True}),
    ("c_proj dgrad", 200, 3072, 768, False, {"aux": True}), ("c_fc dgrad", 200, 768, 3072, False, {}),
    ("out_proj dgrad", 200, 768, 768, False, {}), ("qkv dgrad", 200, 768, 2304, False, {}),
    ("patch embed", 196, 768, 3072, True, {}),
]


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev = "cuda"
    tot = [0.0, 0.0]
    for name, m, n, k, tb, kw in SHAPES:
        a = torch.randn(m, k, device=dev)
        b = torch.randn((n, k) if tb else (k, n), device=dev)
        bias = torch.randn(n, device=dev)
        res = torch.randn(m, n, device=dev) if kw.get("res") else None
        aux = torch.randn(m, n, device=dev) if kw.get("aux") else None
        f = lambda: vit_hip._gemm(a, b, tb, bias=None if aux is not None else bias, residual=res,
                                  a_gelu=bool(kw.get("a_gelu")), gelu_grad_aux=aux)
        g = (lambda: torch.addmm(bias, a, b.t())) if tb else (lambda: a @ b)
        us, ut = timeit(f), timeit(g)
        fl = 2.0 * m * n * k
        tot[0] += us; tot[1] += ut
        print(f"{name:16s} M{m} N{n:5d} K{k:5d} {'NT' if tb else 'NN'} | w2e {us:7.1f} us {fl / us / 1e6:6.1f} TF | torch {ut:7.1f} us {fl / ut / 1e6:6.1f} TF")
    print(f"sum: w2e {tot[0]:.1f} us, torch {tot[1]:.1f} us  (x12 layers per pass)")


if __name__ == "__main__":
    main()

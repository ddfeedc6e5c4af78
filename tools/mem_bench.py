#!/usr/bin/env python3
"""Microbenchmark of the HBM-bound kernels at the 1024^2 generator's shapes (GB/s of algorithmic bytes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from where2edit_amd import functional as K  # noqa: E402
from where2edit_amd._lib import call, ptr, stream_ptr  # noqa: E402


def timeit(fn, iters=10):
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
    dev = "cuda"
    B = 4
    k4 = (torch.tensor([1., 3., 3., 1.])[:, None] * torch.tensor([1., 3., 3., 1.])[None, :] / 16).to(dev)
    print(f"{'kernel':44s} {'MB':>8s} {'ms':>8s} {'GB/s':>8s}")
    for c, h in ((32, 512), (64, 256), (128, 128), (256, 64), (512, 32)):
        t = torch.randn(B, c, 2 * h + 1, 2 * h + 1, device=dev)
        noise = torch.randn(1, 1, 2 * h, 2 * h, device=dev)
        nw, bias = torch.randn(1, device=dev), torch.randn(c, device=dev)
        by = 4.0 * B * c * ((2 * h + 1) ** 2 + (2 * h) ** 2)
        ms = timeit(lambda: K._upfirdn2d_raw(t, k4, 2 * h, 2 * h, 1, 1, 1, 1, True, act=(None, noise, nw, bias)))
        print(f"{'blur+act ' + str(c) + 'ch -> ' + str(2 * h):44s} {by / 1e6:8.1f} {ms:8.3f} {by / ms / 1e6:8.0f}")
        tp = torch.randn(B, c, 2, 2, h + 1, K.planar_pitch(h), device=dev)  # phase-planar T' as the UP conv writes it
        ms = timeit(lambda: K._upfirdn2d_raw(tp, k4, 2 * h, 2 * h, 1, 1, 1, 1, True, act=(None, noise, nw, bias),
                                             planar_hw=(2 * h + 1, 2 * h + 1)))
        print(f"{'blur+act planar ' + str(c) + 'ch -> ' + str(2 * h):44s} {by / 1e6:8.1f} {ms:8.3f} {by / ms / 1e6:8.0f}")
        g = torch.randn(B, c, 2 * h, 2 * h, device=dev)
        ms = timeit(lambda: K._upfirdn2d_raw(g, k4, 2 * h + 1, 2 * h + 1, 1, 1, 2, 2, False))
        print(f"{'blur adjoint ' + str(c) + 'ch ' + str(2 * h) + ' -> ' + str(2 * h + 1):44s} {by / 1e6:8.1f} {ms:8.3f} {by / ms / 1e6:8.0f}")
    for c, h in ((32, 1024), (64, 512), (128, 256), (512, 64)):
        gy, y = torch.randn(B, c, h, h, device=dev), torch.randn(B, c, h, h, device=dev)
        noise = torch.randn(h * h, device=dev)
        gx, sums = torch.empty_like(gy), torch.empty(B, c, 3, device=dev)
        by = 4.0 * 3 * gy.numel()
        ms = timeit(lambda: call("w2e_bias_act_bwd_reduce", ptr(gy), ptr(y), ptr(noise), ptr(gx), ptr(sums), B, c, h * h, 0.2,
                                 2 ** 0.5, stream_ptr()))
        print(f"{'bias_act_bwd_reduce ' + str(c) + 'ch @' + str(h):44s} {by / 1e6:8.1f} {ms:8.3f} {by / ms / 1e6:8.0f}")
    for c, h in ((32, 1024), (64, 512), (512, 64)):
        x = torch.randn(B, c, h, h, device=dev)
        wmod = torch.randn(B, 3, c, device=dev)
        bias = torch.randn(3, device=dev)
        skip = torch.randn(B, 3, h // 2, h // 2, device=dev)
        y = torch.empty(B, 3, h, h, device=dev)
        by = 4.0 * (x.numel() + y.numel() + skip.numel())
        ms = timeit(lambda: call("w2e_torgb_fwd", ptr(x), ptr(wmod), ptr(bias), ptr(skip), ptr(k4 * 4), ptr(y), B, c, h, h, stream_ptr()))
        print(f"{'torgb_fwd ' + str(c) + 'ch @' + str(h):44s} {by / 1e6:8.1f} {ms:8.3f} {by / ms / 1e6:8.0f}")
        gy = torch.randn(B, 3, h, h, device=dev)
        gx, gw = torch.empty_like(x), torch.empty_like(wmod)
        by = 4.0 * (2 * x.numel() + gy.numel())
        ms = timeit(lambda: call("w2e_torgb_bwd", ptr(x), ptr(wmod), ptr(gy), ptr(gx), ptr(gw), B, c, h, h, stream_ptr()))
        print(f"{'torgb_bwd ' + str(c) + 'ch @' + str(h):44s} {by / 1e6:8.1f} {ms:8.3f} {by / ms / 1e6:8.0f}")
    n = 1 << 28
    a, b2 = torch.empty(n // 4, device=dev), torch.empty(n // 4, device=dev)
    ms = timeit(lambda: b2.copy_(a))
    print(f"{'torch copy_ 268 MB (reference point)':44s} {2 * n / 4 * 4 / 1e6:8.1f} {ms:8.3f} {2 * n / ms / 1e6:8.0f}")


if __name__ == "__main__":
    main()

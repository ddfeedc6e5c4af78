#!/usr/bin/env python3
"""w2e_gemm_pk (packed operands, register-fed, one wave per tile) per shape of the ViT-B/32 tower at M = 50*batch: correctness against
float64 and microseconds per launch for several K splits (HIP events over back-to-back launches, L2-warm), the split w2e_gemm_pk_splits
picks, and the stand-alone packing pass.  (Round 2's w2e_gemm_fm on the same shapes, batch 4: 14.4 / 9.7 / 17.0 / 17.0 / 14.4 us.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from where2edit_amd import vit_hip as V  # noqa: E402
from where2edit_amd._lib import call, ptr, stream_ptr  # noqa: E402

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


def pack(x, pad):
    rows, k = x.shape
    rp = -(-rows // pad) * pad
    p = torch.empty((k // 4, rp, 4), device=x.device, dtype=torch.float32)
    call("w2e_pack_kq", ptr(x), ptr(p), rows, rp, k, x.stride(0), 0, stream_ptr())
    return p, rp


for name, n, k in SHAPES:
    a = torch.randn(M, k, device="cuda")
    w = torch.randn(n, k, device="cuda")
    ap, mp = pack(a, 32)
    wp, np_ = pack(w, 64)
    ref = a.double() @ w.double().t()
    res = []
    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
        chunks = k // 8
        per = -(-(-(-chunks // sp)) // 4) * 4
        if (sp - 1) * per >= chunks:
            continue
        c = torch.empty((sp, M, n), device="cuda")
        run = lambda: call("w2e_gemm_pk", ptr(ap), ptr(wp), ptr(c), M, n, k, mp, np_, n, sp, stream_ptr())
        run()
        err = ((c.double().sum(0) - ref).abs().max() / ref.abs().max()).item()
        assert err < 1e-5, (name, sp, err)
        res.append((sp, timeit(run)))
    from where2edit_amd import _lib
    pick = _lib.load().w2e_gemm_pk_splits(M, n, k)
    tp = timeit(lambda: pack(a, 32))
    gf = 2.0 * M * n * k / 1e9
    print(f"{name:18s} M={M} N={n} K={k} ({gf:.2f} GF)  pick x{pick} | pack A {tp:4.1f} us | gemm_pk " +
          "  ".join(f"x{sp}:{t:5.1f}" for sp, t in res), flush=True)
    # the opt-in fp16-operand form (w2e_gemm_pk_h): fp16 weight pack, A rounded in registers
    wh = torch.empty((k // 8, np_, 8), device="cuda", dtype=torch.float16)
    call("w2e_pack_kq_h", ptr(w), wh.data_ptr(), n, np_, k, k, 0, stream_ptr())
    ref_h = a.half().double() @ w.half().double().t()
    res_h = []
    for sp in (1, 2, 3, 4, 6, 8, 12, 16):
        steps = k // 16
        per = -(-(-(-steps // sp)) // 4) * 4
        if sp > steps // 4 or (sp - 1) * per >= steps:
            continue
        c = torch.empty((sp, M, n), device="cuda")
        run = lambda: call("w2e_gemm_pk_h", ptr(ap), wh.data_ptr(), ptr(c), M, n, k, mp, np_, n, sp, stream_ptr())
        run()
        err = ((c.double().sum(0) - ref_h).abs().max() / ref_h.abs().max()).item()
        assert err < 1e-5, (name, sp, err)
        res_h.append((sp, timeit(run)))
    pick_h = _lib.load().w2e_gemm_pk_h_splits(M, n, k)
    print(f"{'':18s} fp16 operands: pick x{pick_h} | gemm_pk_h " + "  ".join(f"x{sp}:{t:5.1f}" for sp, t in res_h), flush=True)

#!/usr/bin/env python3
"""fp32 GEMM rate of the vendor library (hipBLASLt through torch.mm) on this device: the practical fp32-matrix ceiling
SURVEY 8(d) asks to be recorded next to the 157.3 TFLOP/s datasheet figure."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
for n in (4096, 8192):
    a, b = torch.randn(n, n, device="cuda"), torch.randn(n, n, device="cuda")
    for _ in range(3):
        a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"torch.mm fp32 {n}^3: {ms:.3f} ms  {2.0 * n ** 3 / ms / 1e9:.1f} TFLOP/s")

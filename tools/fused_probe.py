"""Skip-bit decomposition of w2e_wino_fused on a -DW2E_TUNING build (W2E_LIB_PATH): bit 0 no DMA after the prologue's, 1 no transform,
2 no MFMAs, 3 no output rounds.  Relative use only (the runtime skip branches cost a little themselves)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from where2edit_amd._lib import call, ptr, stream_ptr  # noqa: E402

if __name__ == "__main__":
    dev = "cuda"
    for (k, hw, b) in ((32, 1024, 8), (64, 512, 8), (128, 256, 8)):
        n = k
        x = torch.randn(b, k, hw, hw, device=dev)
        y = torch.empty(b, n, hw, hw, device=dev)
        uf = torch.randn(36, k // 8, 2, n, 4, device=dev) * 0.01
        for skip in (0, 8, 2, 10, 4, 12, 14, 6):
            wgs = skip << 16
            for _ in range(3):
                call("w2e_wino_fused", ptr(x), None, ptr(uf), None, ptr(y), b, k, n, hw, hw, 0, None, None, None, None, None, None, wgs, stream_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call("w2e_wino_fused", ptr(x), None, ptr(uf), None, ptr(y), b, k, n, hw, hw, 0, None, None, None, None, None, None, wgs, stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            what = ", ".join(w for bit, w in ((1, "no DMA"), (2, "no transform"), (4, "no MFMA"), (8, "no output")) if skip & bit) or "everything"
            print(f"K {k:3d} @ {hw:4d} b{b} skip {skip:2d} ({what}): {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)

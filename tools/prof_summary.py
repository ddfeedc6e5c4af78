#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (kernel_stats.csv) per benchmark step."""
import csv
import sys

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
if len(sys.argv) > 2 and sys.argv[2] != "auto":
    steps = float(sys.argv[2])
else:  # every mapper step launches the CLIP preprocessing backward exactly once (one CLIP loss per step, whatever the workload)
    steps = float(sum(int(r["Calls"]) for r in rows if "clip_preproc_bwd" in r["Name"])) or 1.0
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"# {path}\n# total kernel time {tot / steps / 1e6:.3f} ms/step over {steps:g} steps")
fam = {}
for r in rows:
    name = r["Name"]
    key = ("w2e::modconv_kernel<*>" if "modconv_kernel" in name
           else "Cijk_* (library fp32 GEMMs: stock-op mm of the [B,512]-sized tails)" if name.startswith("Cijk_") else name.split("(")[0][:70])
    c, t = fam.get(key, (0, 0))
    fam[key] = (c + int(r["Calls"]), t + int(r["TotalDurationNs"]))
# the 3x3 modulated convs as bench.py's roofline counts them: direct launches + Winograd-form calls (one wino4_fused3 launch, or one
# wino4_gemm launch with its packed input transform and, for a split K, its finish launch -- all kernels of this repo)
def is_conv(n):
    return "modconv_kernel" in n or "wino" in n


conv_t = sum(int(r["TotalDurationNs"]) for r in rows if is_conv(r["Name"]))
conv_c = sum(int(r["Calls"]) for r in rows if "modconv_kernel" in r["Name"] or "wino4_gemm_kernel" in r["Name"] or "wino4_fused" in r["Name"])
wino_t = sum(int(r["TotalDurationNs"]) for r in rows if "wino" in r["Name"])
if conv_c:
    print(f"# all 3x3 modulated conv calls: {conv_t / steps / 1e6:.3f} ms/step, {conv_c / steps:.1f} calls/step, {conv_t / conv_c / 1e3:.1f} us per call "
          f"(of which Winograd-form calls: {wino_t / steps / 1e6:.3f} ms/step)")
print("# by kernel family: ms/step, calls/step, avg us, share")
top = 18 if "--all" not in sys.argv else len(fam)  # --all: every kernel family (the launch tail)
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / steps / 1e6:9.3f} {c / steps:8.1f} {t / c / 1e3:10.1f} {100.0 * t / tot:6.2f}%  {k}")
print("# modconv instantiations <MODE,EPI,NOB,NPB,WO,WP,KC>")
for r in rows:
    if "modconv_kernel" in r["Name"]:
        print(f"{int(r['TotalDurationNs']) / steps / 1e6:9.3f} {int(r['Calls']) / steps:8.1f} {float(r['AverageNs']) / 1e3:10.1f}  {r['Name'][5:60]}")

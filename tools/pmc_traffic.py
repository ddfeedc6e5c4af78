#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes of bench.py (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace only)
into profiles/<tag>_pmc_hbm_traffic.txt and profiles/traffic.json.

    tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <tag> [out_dir]
(out_dir defaults to profiles/; on the GPU box pass gpurun_out/... and copy the two files into profiles/ afterwards)
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if "modconv_kernel" in name:
            key = "w2e::modconv_kernel<*>"
        else:
            key = name.split("(")[0][:60]
        d[key][0] += 1
        d[key][1] += float(r["Counter_Value"])
    return d


def main():
    f, w, tag = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of",
             "#   python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing   (batch 4, 1024^2)",
             "# per-launch averages.  HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950's FETCH_SIZE reads exactly 1/2 on wide",
             "# (16 B/lane) streaming reads (MI355X_MICROARCH.md, HBM); verified here on torgb_fwd (float4 loads: 2*FETCH = its",
             "# algorithmic 313 MB/launch).  The conv kernel stages activations with 4 B/lane loads, for which the factor is",
             "# uncalibrated: its corrected figure is an upper bound, the uncorrected one a lower bound.",
             "# kernel, launches, FETCH_SIZE KB/launch, WRITE_SIZE KB/launch, HBM MB/launch corrected, uncorrected"]
    out = {}
    # every 3x3 modulated conv call of the step = a direct launch, or a Winograd-form call (one wino4_fused3 launch, or one wino4_gemm
    # launch with its packed input transform and finish launch): the `traffic` of bench.py's roofline is their HBM bytes per call
    conv = [k for k in f if k in w and ("modconv_kernel" in k or "wino" in k)]
    calls = sum(f[k][0] for k in conv if "modconv_kernel" in k or "wino4_gemm_kernel" in k or "wino4_fused" in k)
    if calls:
        ft, wt = sum(f[k][1] for k in conv) / calls, sum(w[k][1] for k in conv) / calls
        lines.append(f"all 3x3 modulated conv calls (direct launches + Winograd-form calls), {calls}, {ft:.0f}, {wt:.0f}, {(2 * ft + wt) / 1024:.1f}, {(ft + wt) / 1024:.1f}")
        out["conv_calls"] = (2 * ft + wt) * 1024
    for k in sorted(f, key=lambda k: -f[k][1]):
        if k not in w:
            continue
        n = f[k][0]
        fk, wk = f[k][1] / n, w[k][1] / w[k][0]
        if fk + wk < 1000:
            continue
        lines.append(f"{k}, {n}, {fk:.0f}, {wk:.0f}, {(2 * fk + wk) / 1024:.1f}, {(fk + wk) / 1024:.1f}")
        out[k] = (2 * fk + wk) * 1024
    out_dir = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles")
    txt = os.path.join(out_dir, f"{tag}_pmc_hbm_traffic.txt")
    open(txt, "w").write("\n".join(lines) + "\n")
    json.dump({"source": f"profiles/{tag}_pmc_hbm_traffic.txt", "modconv_hbm_bytes_per_launch": out.get("conv_calls", out.get("w2e::modconv_kernel<*>"))},
              open(os.path.join(out_dir, "traffic.json"), "w"))
    print("\n".join(lines))


if __name__ == "__main__":
    main()

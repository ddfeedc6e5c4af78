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
             "# per-launch averages.  HBM-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  The factor 2 is CALIBRATED for every access",
             "# pattern of this library (profiles/r05_fetch_calibration.txt, tools/fetch_calib.hip): every L2 fill on gfx950 is a 128-byte",
             "# line that FETCH_SIZE tallies as 64 bytes -- 16 / 8 / 4 bytes per lane, global_load, raw_buffer_load and buffer_load ... lds",
             "# alike; a partially used line (the 16-byte column halo of a patch row) costs its whole 128 bytes.  Units: KiB / MiB (2^10, 2^20).",
             "# kernel, launches, FETCH_SIZE KiB/launch, WRITE_SIZE KiB/launch, HBM-side MiB/launch (2 x FETCH + WRITE), MB/launch (10^6)"]
    out = {}
    # every 3x3 modulated conv call of the step = a direct launch, or a Winograd-form call (one wino4_fused3 launch, or one wino4_gemm
    # launch with its packed input transform and finish launch): the `traffic` of bench.py's roofline is their HBM bytes per call
    conv = [k for k in f if k in w and ("modconv_kernel" in k or "wino" in k)]
    calls = sum(f[k][0] for k in conv if "modconv_kernel" in k or "wino4_gemm_kernel" in k or "wino4_fused" in k)
    if calls:
        ft, wt = sum(f[k][1] for k in conv) / calls, sum(w[k][1] for k in conv) / calls
        lines.append(f"all 3x3 modulated conv calls (direct launches + Winograd-form calls), {calls}, {ft:.0f}, {wt:.0f}, {(2 * ft + wt) / 1024:.1f}, {(2 * ft + wt) * 1024 / 1e6:.1f}")
        out["conv_calls"] = (2 * ft + wt) * 1024
    for k in sorted(f, key=lambda k: -f[k][1]):
        if k not in w:
            continue
        n = f[k][0]
        fk, wk = f[k][1] / n, w[k][1] / w[k][0]
        if fk + wk < 1000:
            continue
        lines.append(f"{k}, {n}, {fk:.0f}, {wk:.0f}, {(2 * fk + wk) / 1024:.1f}, {(2 * fk + wk) * 1024 / 1e6:.1f}")
        out[k] = (2 * fk + wk) * 1024
    out_dir = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles")
    txt = os.path.join(out_dir, f"{tag}_pmc_hbm_traffic.txt")
    open(txt, "w").write("\n".join(lines) + "\n")
    # (`source` names the file as it is kept under profiles/: profiles/<tag>_pmc_hbm_traffic.txt -- pass the tag the committed file carries)
    json.dump({"source": f"profiles/{tag}_pmc_hbm_traffic.txt", "unit": "bytes per 3x3 modulated conv call = (2 x FETCH_SIZE + WRITE_SIZE) x 1024",
               "calibration": "profiles/r05_fetch_calibration.txt", "modconv_hbm_bytes_per_launch": out.get("conv_calls", out.get("w2e::modconv_kernel<*>"))},
              open(os.path.join(out_dir, "traffic.json"), "w"))
    print("\n".join(lines))


if __name__ == "__main__":
    main()

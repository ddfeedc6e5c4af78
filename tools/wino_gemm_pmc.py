#!/usr/bin/env python3
"""Per-launch counters of the Winograd-domain contraction kernel (w2e::wino4_gemm_kernel) and its input transform from three
rocprofv3 --pmc passes of tools/layer_bench.py --batch 8 --only 8,10 (512 @ 64^2 and 256 @ 128^2):
    wino_gemm_pmc.py <TCC csv> <FETCH_SIZE csv> <SQ csv>
L2 hit rate = TCC_HIT / (HIT + MISS); HBM-side bytes = 2 * FETCH_SIZE KB (gfx950's FETCH_SIZE reads 1/2 on 16-B-per-lane streaming
reads: MI355X_MICROARCH.md, HBM); matrix-pipe occupancy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)."""
import collections
import csv
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "wino4_gemm" not in n and "pack_input" not in n:
            continue
        agg[(n.split("(")[0].replace("void ", "")[:48], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


tcc, fetch, sq = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
print("# kernel, grid threads | L2 requests (128 B), hit rate | fabric-side fetch MB (2 x FETCH_SIZE) | matrix-pipe occupancy, wait / issue-stall share of wave cycles")
for k in sorted(tcc):
    t, f, s = tcc[k], fetch.get(k, {}), sq.get(k, {})
    hit = t.get("TCC_HIT_sum", 0) / max(t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0), 1)
    occ = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1024 * s.get("GRBM_GUI_ACTIVE", 0) / 8, 1)
    wc = max(s.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{k[0]:50s} {k[1]:>8s} | {t.get('TCC_REQ_sum', 0) / 1e6:7.2f} M, {hit:.3f} | {2 * f.get('FETCH_SIZE', 0) / 1024:8.1f} | "
          f"{occ:.3f}, {s.get('SQ_WAIT_ANY', 0) / wc:.3f} / {s.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}")

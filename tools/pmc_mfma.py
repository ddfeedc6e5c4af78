#!/usr/bin/env python3
"""Matrix-pipe occupancy of the conv kernel from a rocprofv3 SQ counter pass:
   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 bench.py ...
usage: pmc_mfma.py <counter_collection.csv>   (per kernel family: launches, sums per launch, MFMA busy / CU busy)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
fam = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in rows:
    name = r["Kernel_Name"]
    key = ("w2e::modconv_kernel<*>" if "modconv_kernel" in name else "w2e::gemm_pk_kernel (ViT tower GEMM)" if "gemm_pk" in name
           else "w2e::wino4_gemm_kernel<*> (the Winograd-domain contraction)" if "wino4_gemm_kernel" in name
           else "w2e::wino4_fused3_kernel<*>" if "wino4_fused3" in name else None)
    if key is None:
        continue
    fam[key][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[key].add(r["Dispatch_Id"])
print("# kernel family, launches, then per-launch sums of each counter over the chip.  SQ_VALU_MFMA_BUSY_CYCLES = SIMD-cycles a matrix pipe is busy\n"
      "# (= FLOP / 64 for v_mfma_f32_32x32x2_f32: 4096 FLOP per 64 cycles); GRBM_GUI_ACTIVE = the sum over the 8 XCDs of the cycles the launch\n"
      "# is resident.  Matrix-pipe occupancy = MFMA busy / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).")
for key, c in fam.items():
    n = len(launches[key])
    print(f"{key}: {n} launches")
    for k in sorted(c):
        print(f"    {k:28s} {c[k] / n:16.0f}")
    if c.get("GRBM_GUI_ACTIVE"):
        print(f"    matrix-pipe occupancy        {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0):.3f}")
    if c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        print("    of wave cycles: " + ", ".join(f"{k} {c[k] / w:.3f}" for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in c))

"""LDS-array share of a kernel's time from a rocprofv3 --pmc pass (SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_LDS_ADDR_CONFLICT,
SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE): per kernel name, per-launch averages.   usage: pmc_lds.py counter_collection.csv [name filter]"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if flt not in name:
            continue
        key = name[:90] + " | grid " + r.get("Grid_Size", "?")
        rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[key].add(r["Dispatch_Id"])
for key, c in rows.items():
    n = max(len(launches[key]), 1)
    print(key, f"({n} launches)")
    for k, v in sorted(c.items()):
        print(f"    {k:28s} {v / n:16.0f}")
    if "SQ_LDS_IDX_ACTIVE" in c and "GRBM_GUI_ACTIVE" in c:
        cu_cycles = c["GRBM_GUI_ACTIVE"] / 8 * 256  # per-XCD active cycles summed over 8 XCDs -> CU-cycles of the launch
        print(f"    LDS array busy / CU-cycles of the launch: {c['SQ_LDS_IDX_ACTIVE'] / cu_cycles:.3f}   bank-conflict share of it: "
              f"{c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(c['SQ_LDS_IDX_ACTIVE'], 1.0):.3f}")

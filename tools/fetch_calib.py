#!/usr/bin/env python3
"""Merge tools/fetch_calib.hip's table (known bytes per probe) with the rocprofv3 PMC passes of the same binary:

    tools/fetch_calib.py <probe table .txt> <counter_collection.csv> [<counter_collection.csv> ...]  > profiles/rNN_fetch_calibration.txt

For every probe: FETCH_SIZE x 1024 against the bytes the lanes requested (U) and against the bytes of the distinct 128-B lines / 64-B
half-lines / 32-B sectors touched; the raw request counters when a pass collected them.  `factor` = U / (FETCH_SIZE x 1024) is what a
kernel with that access pattern must multiply its FETCH_SIZE by to get the bytes it pulled through the fabric-side of L2."""
import collections
import csv
import re
import sys


def main():
    table = {}
    for ln in open(sys.argv[1]):
        if ln.startswith("P"):
            f = [x.strip() for x in ln.split(",")]
            table[int(f[0][1:])] = {"what": f[1], "u": float(f[2]), "l128": float(f[3]), "l64": float(f[4]), "l32": float(f[5]), "ms": float(f[6]), "gbs": float(f[7])}
    counters = collections.defaultdict(dict)
    for path in sys.argv[2:]:
        for r in csv.DictReader(open(path)):
            m = re.search(r"probe<(\d+)>", r["Kernel_Name"])
            if m:
                counters[int(m.group(1))][r["Counter_Name"]] = counters[int(m.group(1))].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    print("# tools/fetch_calib.hip under rocprofv3 --pmc (separate passes): FETCH_SIZE against known byte counts, per access pattern")
    print("# probe | pattern | U = MB requested | FETCH_SIZE MB | U/FETCH (the correction factor) | FETCH / 128-B lines touched | / 64-B halves | / 32-B sectors | GB/s of U (unprofiled run) | raw counters")
    for p in sorted(table):
        t, c = table[p], counters.get(p, {})
        fetch = c.get("FETCH_SIZE")
        raw = ", ".join(f"{k}={v:.0f}" for k, v in sorted(c.items()) if k != "FETCH_SIZE")
        if fetch is None:
            print(f"P{p} | {t['what']} | {t['u'] / 1e6:.1f} | - | - | - | - | - | {t['gbs']:.0f} | {raw}")
            continue
        fb = fetch * 1024.0
        print(f"P{p} | {t['what']} | {t['u'] / 1e6:.1f} | {fb / 1e6:.1f} | {t['u'] / fb:.3f} | {fb / t['l128']:.3f} | {fb / t['l64']:.3f} | {fb / t['l32']:.3f} | {t['gbs']:.0f} | {raw}")


if __name__ == "__main__":
    main()

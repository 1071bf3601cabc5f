#!/usr/bin/env python3
"""Per-kernel summary CSV of a rocprofv3 --pmc output directory (the raw counter_collection.csv
has one row per dispatch and counter: too big to commit).
usage: pmc_summary_csv.py DIR OUT.csv
Columns: kernel, dispatches, then for every counter its total and its mean per dispatch.  With
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE present a `mfma_busy_frac` column is added:
SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), DESIGN.md section 3."""
import csv
import glob
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
counters = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
        if r["Counter_Name"] not in counters:
            counters.append(r["Counter_Name"])
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    mfma = "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters
    w.writerow(["kernel", "dispatches"] + [c + s for c in counters for s in ("_total", "_per_dispatch")]
               + (["mfma_busy_frac"] if mfma else []))
    for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
        n = len(calls[k])
        row = [k[:120], n]
        for c in counters:
            row += [round(tot[k][c], 1), round(tot[k][c] / n, 1)]
        if mfma:
            g = tot[k]["GRBM_GUI_ACTIVE"]
            row.append(round(tot[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (g / 8), 4) if g else "")
        w.writerow(row)
print("wrote", sys.argv[2], len(tot), "kernels")

#!/usr/bin/env python3
"""Prints the kernel timeline of the LAST rollout in a rocprofv3 kernel trace.
usage: timeline.py <kernel_trace.csv> [marker substring, default rollout_setup]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2] if len(sys.argv) > 2 else "rollout_setup"
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
i0, i1 = idx[which], idx[which + 1] if which + 1 < 0 or which + 1 < len(idx) else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
prev = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1000:9.1f} +{(e - s) / 1000:8.1f} gap {(s - prev) / 1000:6.1f}  "
          f"{r['Kernel_Name'][:64]}  grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']} "
          f"vgpr={r['VGPR_Count']}+{r['Accum_VGPR_Count']}")
    prev = e
print(f"total {(prev - t0) / 1000:.1f} us")

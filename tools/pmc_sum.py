#!/usr/bin/env python3
"""Per-kernel totals of the counters in a rocprofv3 --pmc output directory.
usage: pmc_sum.py DIR [name-substring]"""
import csv
import glob
import sys
from collections import defaultdict

want = sys.argv[2] if len(sys.argv) > 2 else ""
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            tot[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[r["Kernel_Name"][:60]].add(r["Dispatch_Id"])
for k, d in tot.items():
    n = len(calls[k])
    print(k, "calls", n, {c: round(v / n) for c, v in d.items()})

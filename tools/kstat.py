#!/usr/bin/env python3
"""Prints calls / average us of the kernels whose name contains a substring, from a rocprofv3
kernel_stats.csv.  usage: kstat.py <dir or csv> <substring> [...]"""
import csv, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    for root, _, files in os.walk(path):
        for f in files:
            if f.endswith("kernel_stats.csv"):
                path = os.path.join(root, f)
for r in csv.DictReader(open(path)):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us')

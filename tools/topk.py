#!/usr/bin/env python3
"""Prints the top kernels of a rocprofv3 --stats output directory. usage: topk.py DIR [K=12]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(f)))[:k]:
    print(r["Name"][:72], r["Calls"], r["AverageNs"], r["Percentage"])

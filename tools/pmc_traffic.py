#!/usr/bin/env python3
"""Turns the rocprofv3 PMC passes over tools/step_probe.py into profiles/r01_traffic.json.

usage: pmc_traffic.py <dir with fetch_<workload>/ and write_<workload>/ outputs> [out.json]
Each pass:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir>/fetch_<w> -o p
            -- python3 tools/step_probe.py <kind,N,B>      (WRITE_SIZE in its own pass)
FETCH_SIZE is doubled (MI355X_MICROARCH.md, gfx950 correction: the counter counts 32-B
beats of 64-B requests as one), both counters are KB -> bytes x1024; the figure is the mean
over all decode_step launches of the probe (whole episodes)."""
import csv
import glob
import json
import os
import sys


def mean_counter(d, name):
    vals = []
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "decode_step_rt_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


def main():
    root = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_traffic.json")
    res = {}
    for fd in sorted(glob.glob(os.path.join(root, "fetch_*"))):
        w = os.path.basename(fd)[len("fetch_"):]
        fetch, n = mean_counter(fd, "FETCH_SIZE")
        write, _ = mean_counter(os.path.join(root, "write_" + w), "WRITE_SIZE")
        res[w] = {"fetch_size_kb_raw": round(fetch, 1), "write_size_kb": round(write, 1),
                  "launches": n, "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace "
                   "only) around tools/step_probe.py; FETCH_SIZE doubled as MI355X_MICROARCH.md "
                   "prescribes for gfx950, KB -> bytes x1024; mean over all launches of the probe "
                   "(tools/pmc_traffic.py)",
           "kernel": "decode_step_rt_kernel", "workloads": res}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

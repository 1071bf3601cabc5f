#!/usr/bin/env python3
"""Turns the rocprofv3 PMC passes over tools/step_probe.py into profiles/r0N_traffic.json.

usage: pmc_traffic.py <dir with fetch_<workload>/ and write_<workload>/ outputs> [out.json]
Each pass:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir>/fetch_<w> -o p
            -- python3 tools/step_probe.py <kind,N,B>      (WRITE_SIZE in its own pass)
FETCH_SIZE is doubled (MI355X_MICROARCH.md, gfx950 correction: the counter counts 32-B
beats of 64-B requests as one), both counters are KB -> bytes x1024.  The figure is HBM bytes
per decode STEP: the sum over every decode kernel of an episode (per-step kernels, or the
step-0 launch plus the persistent multi-step kernel) divided by the episode's steps, mean over
the probe's episodes -- the unit of roofline.algorithmic_bytes_per_launch."""
import csv
import glob
import json
import os
import sys

DECODE = ("decode_step_rt_kernel", "decode_persistent_kernel", "decode_step_tile_mfma_kernel",
          "decode_step_tile_zmfma_kernel",
          "persistent_finalize_kernel")


def counter_sum(d, name, idle_below_kb=0.0):
    """(sum of the counter over all decode kernels, number of env steps they cover).
    idle_below_kb: per-step launches that fetched less than this are the no-op launches behind
    `done` (a fixed-length loop of 2(N-1) launches; they leave at their first instruction) and
    cover no env step."""
    total, launches, persistent = 0.0, 0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name or not any(k in r["Kernel_Name"] for k in DECODE):
                continue
            v = float(r["Counter_Value"])
            total += v
            if "decode_persistent_kernel" in r["Kernel_Name"]:
                persistent += 1
            elif "finalize" not in r["Kernel_Name"] and v >= idle_below_kb:
                launches += 1
    return total, launches, persistent


def main():
    root = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_traffic.json")
    res = {}
    for fd in sorted(glob.glob(os.path.join(root, "fetch_*"))):
        w = os.path.basename(fd)[len("fetch_"):]
        kind, N, B = int(w.split("_")[0][4:]), int(w.split("_")[1][1:]), int(w.split("_")[2][1:])
        # a live step fetches at least its graphs' score rows and masks (> 1 KB per graph); a no-op
        # launch a few bytes per wave: the line is drawn at 256 B per graph (raw counter: 128 B)
        fetch, launches, persistent = counter_sum(fd, "FETCH_SIZE", idle_below_kb=B * 128 / 1024.0)
        write, _, _ = counter_sum(os.path.join(root, "write_" + w), "WRITE_SIZE")
        # steps covered: one per per-step launch; a persistent launch covers the rest of an episode
        # (TSP: N-1 steps per episode in all; the probe prints T for the others)
        steps = launches
        if persistent:
            log = os.path.join(os.path.dirname(root.rstrip("/")), f"pmc_fetch_{w}.log")
            T = None
            if os.path.exists(log):
                for line in open(log):
                    if line.startswith("{") and "steps_per_episode" in line:
                        T = json.loads(line)["steps_per_episode"]
            T = T or N - 1
            steps = launches + persistent * (T - 1)
        res[w] = {"fetch_size_kb_raw": round(fetch, 1), "write_size_kb": round(write, 1),
                  "decode_launches": launches + persistent, "steps": steps,
                  "hbm_bytes_per_launch": int((2 * fetch + write) * 1024 / max(steps, 1))}
    sha = None
    sha_file = os.path.join(os.path.dirname(root.rstrip("/")), "source_hash.txt")
    if os.path.exists(sha_file):
        sha = open(sha_file).read().strip()
    doc = {"source_hash": sha,
           "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace "
                   "only) around tools/step_probe.py; FETCH_SIZE doubled as MI355X_MICROARCH.md "
                   "prescribes for gfx950, KB -> bytes x1024; HBM bytes of all decode kernels per "
                   "env step (tools/pmc_traffic.py)",
           "kernels": list(DECODE), "workloads": res}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

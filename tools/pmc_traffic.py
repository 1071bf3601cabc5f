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
          "persistent_finalize_kernel")


def counter_sum(d, name):
    """(sum of the counter over all decode kernels, number of env steps they cover)"""
    total, launches, persistent = 0.0, 0, 0
    tile_ids, rt_ids = set(), set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name or not any(k in r["Kernel_Name"] for k in DECODE):
                continue
            total += float(r["Counter_Value"])
            if "decode_persistent_kernel" in r["Kernel_Name"]:
                persistent += 1
            elif "finalize" not in r["Kernel_Name"]:
                launches += 1
                (tile_ids if "tile" in r["Kernel_Name"] else rt_ids).add(int(r["Dispatch_Id"]))
    # hybrid dispatch (N > 64) launches the raw-tile and the table kernel back to back for ONE
    # env step while the batch straddles the threshold: such a pair is one step
    launches -= sum(1 for i in tile_ids if i + 1 in rt_ids)
    return total, launches, persistent


def main():
    root = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_traffic.json")
    res = {}
    for fd in sorted(glob.glob(os.path.join(root, "fetch_*"))):
        w = os.path.basename(fd)[len("fetch_"):]
        fetch, launches, persistent = counter_sum(fd, "FETCH_SIZE")
        write, _, _ = counter_sum(os.path.join(root, "write_" + w), "WRITE_SIZE")
        kind, N = int(w.split("_")[0][4:]), int(w.split("_")[1][1:])
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
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace "
                   "only) around tools/step_probe.py; FETCH_SIZE doubled as MI355X_MICROARCH.md "
                   "prescribes for gfx950, KB -> bytes x1024; HBM bytes of all decode kernels per "
                   "env step (tools/pmc_traffic.py)",
           "kernels": list(DECODE), "workloads": res}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

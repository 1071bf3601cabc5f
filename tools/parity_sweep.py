#!/usr/bin/env python3
"""Randomised HIP-vs-oracle rollout parity sweep (tests/test_gpu_parity.py::_compare_rollout over
random kinds, sizes, seeds, modes and kernel variants).
usage: parity_sweep.py [seed] [seconds] [train]   ("train": train-mode cases only; with
VRPGYM_TRAIN_PARITY_LOG=<file> every case logs its error against the fp64 evaluation)"""
import sys, random, time, traceback
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch
import test_gpu_parity as T
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    kind = random.choice([0, 1, 2]); N = random.choice([5, 9, 16, 17, 20, 31, 32, 33, 40, 50, 63, 64, 65, 80, 84, 96, 100, 104, 108, 112, 128])
    B = random.choice([1, 2, 7, 8, 9, 33, 64, 100, 257])
    if N > 64: B = min(B, 33)
    greedy = random.random() < 0.5; train = random.random() < 0.3 or "train" in sys.argv[3:]
    es, ag, ts = random.randint(0, 999), random.choice([69, 1, 7]), random.randint(0, 999)
    mode = random.choice(["default", "wide", "table", "table_wide", "tile", "fused"])
    kw = dict(throughput_kernel=mode in ("wide", "table_wide"), table_kernel=mode in ("table", "table_wide"),
              tile_kernel=mode == "tile" and kind != 2 and N <= 104, fused=mode == "fused" and N <= 63)
    try:
        T._compare_rollout(kind, B, N, greedy, es, ag, ts, train=train, **kw)
    except AssertionError as e:
        bad += 1; print("FAIL", (kind, B, N, greedy, train, es, ag, ts, mode), str(e)[:200])
    n += 1
print("cases", n, "failures", bad)

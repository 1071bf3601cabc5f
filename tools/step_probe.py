#!/usr/bin/env python3
"""Times the decode_step launch (HIP events) over a sweep of shapes; also usable under
rocprofv3 --pmc to attribute counters to the step kernel.  GPU box only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import torch  # noqa: E402

import bench  # noqa: E402

shapes = [(0, 20, 512), (0, 20, 2048), (0, 20, 8192), (0, 40, 2048), (0, 40, 8192), (1, 40, 8192),
          (2, 40, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
dev = torch.device("cuda", 0)
for shp in shapes:
    kind, N, B = shp[:3]
    extra = shp[3] if len(shp) > 3 else 0   # 4th field: extra step flags (16 = throughput kernel, 4 = tile kernel, 32 = no persistent)
    greedy = not (len(shp) > 4 and shp[4])         # 5th field 1 = sampling
    if os.environ.get("STEP_PROBE_PER_STEP"):
        print(json.dumps({"workload": f"kind{kind}_N{N}_B{B}", "per_step_us":
                          bench.step_kernel_roofline(kind, N, B, greedy, dev, reps=3,
                                                     extra_flags=extra, per_step=True)}))
        continue
    r = bench.step_kernel_roofline(kind, N, B, greedy, dev, reps=3, extra_flags=extra)
    print(json.dumps({k: r[k] for k in ("workload", "kernel", "steps_per_episode", "avg_launch_us", "event_pair_per_launch_us", "loop_us_per_step", "loop_us", "rollout_us", "achieved", "frac", "loop_frac", "rollout_frac")}))

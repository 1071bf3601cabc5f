#!/usr/bin/env python3
"""Wall time per rollout the way bench.py times it (resident instances, reset inside the rollout's
set-up kernel, K rollouts between two synchronizes), for whichever library VRPGYM_HIP_LIB selects.
usage: rollout_wall.py kind N B [K=50] [blocks=5] [greedy=1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import torch  # noqa: E402

import bench  # noqa: E402
from agents import runtime  # noqa: E402

kind, N, B = (int(x) for x in sys.argv[1:4])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 50
blocks = int(sys.argv[5]) if len(sys.argv) > 5 else 5
greedy = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True
dev = torch.device("cuda", 0)
env, agent = bench.make(kind, N, B, 69, dev)
ts = []
with torch.no_grad():
    for _ in range(10):
        res = runtime.rollout(agent.model, env, greedy, reset_env=True)
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            res = runtime.rollout(agent.model, env, greedy, reset_env=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K)
    # how long the HOST needs to issue K rollouts (no waiting for the GPU in between): if this is
    # close to the figure above, the loop is bound by Python + launch calls, not by the kernels
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        res = runtime.rollout(agent.model, env, greedy, reset_env=True)
    issue = (time.perf_counter() - t0) / K
    torch.cuda.synchronize()
ts.sort()
print(f"kind={kind} N={N} B={B}: {ts[len(ts) // 2] * 1e3:.4f} ms per rollout (min {ts[0] * 1e3:.4f}, max {ts[-1] * 1e3:.4f}); "
      f"host issue {issue * 1e3:.4f} ms; T {res.T} cost {float(-res.acc_loss.mean()):.6f} lib {os.environ.get('VRPGYM_HIP_LIB', 'default')[-24:]}")

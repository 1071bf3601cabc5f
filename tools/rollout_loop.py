#!/usr/bin/env python3
"""Runs K greedy/sampled rollouts of one shape (for profilers).
usage: rollout_loop.py kind N B [K=5] [greedy=1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import torch  # noqa: E402

import bench  # noqa: E402
from agents import runtime  # noqa: E402

kind, N, B = (int(x) for x in sys.argv[1:4])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 5
greedy = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
dev = torch.device("cuda", 0)
env, agent = bench.make(kind, N, B, 69, dev)
with torch.no_grad():
    for _ in range(K):
        # the way bench.py runs it: the episode reset is part of the rollout's own set-up kernel
        res = runtime.rollout(agent.model, env, greedy, reset_env=True)
    torch.cuda.synchronize()
print("T", res.T, "cost", float(-res.acc_loss.mean()))

#!/usr/bin/env python3
"""Whole-rollout wall time, table-driven vs raw-tile step kernel.  usage: kind N B greedy(0/1)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import torch
import agents
from agents import runtime
from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
kind, N, B, greedy = (int(x) for x in sys.argv[1:5])
env = (TSPEnv, VRPEnv, IRPEnv)[kind](N, B, 1, 69, generator="device")
agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind](seed=69)
agent.model.eval()
for tile in (False, True):
    if tile and N > 104:
        continue
    for it in range(3):
        env.reset(return_state=False)
        torch.cuda.synchronize(); t0 = time.time()
        with torch.no_grad():
            res = runtime.rollout(agent.model, env, bool(greedy), tile_kernel=tile)
        torch.cuda.synchronize(); dt = time.time() - t0
    T = res.T
    print(f"kind={kind} N={N} B={B} {'tile ' if tile else 'table'}: {dt*1e3:8.2f} ms, T={T}, "
          f"{B*N*T/dt/1e9:.3f} G node-steps/s, cost {-res.acc_loss.mean().item():.4f}")

#!/usr/bin/env python3
"""Times REINFORCE epochs (agent.train) on the GPU.  usage: train_probe.py kind N B epochs"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import logging
logging.disable(logging.CRITICAL)
import torch
import agents
from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
kind, N, B, epochs = (int(x) for x in sys.argv[1:5])
Env = (TSPEnv, VRPEnv, IRPEnv)[kind]
Agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind]
d = tempfile.mkdtemp()
env = Env(num_nodes=N, batch_size=B, seed=69)
agent = Agent(seed=69, csv_path=os.path.join(d, "log.csv"))
agent.train(env, epochs=1, check_point_dir=d + "/")
torch.cuda.synchronize(); t0 = time.time()
agent.train(env, epochs=epochs, check_point_dir=d + "/")
torch.cuda.synchronize(); dt = (time.time() - t0) / epochs
rows = open(os.path.join(d, "log.csv")).read().strip().splitlines()
print(f"kind={kind} N={N} B={B}: {dt*1e3:.1f} ms/epoch (4 rollouts + backward + Adam + t-test); last row {rows[-1]}")

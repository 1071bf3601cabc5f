#!/usr/bin/env python3
"""cProfile of agent.train on the GPU box (host-side share of an epoch).  usage: kind N B epochs"""
import cProfile, os, pstats, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import logging
logging.disable(logging.CRITICAL)
import torch
import agents
from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
kind, N, B, epochs = (int(x) for x in sys.argv[1:5])
d = tempfile.mkdtemp()
env = (TSPEnv, VRPEnv, IRPEnv)[kind](num_nodes=N, batch_size=B, seed=69)
agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind](seed=69, csv_path=os.path.join(d, "log.csv"))
agent.train(env, epochs=2, check_point_dir=d + "/")
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
agent.train(env, epochs=epochs, check_point_dir=d + "/")
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)

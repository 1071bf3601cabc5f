#!/usr/bin/env python3
"""Sanity run (GPU box): 30 REINFORCE epochs of two non-default architectures (hidden_dim not a
multiple of 128, 16 / 4 encoder heads, 2 / 4 layers); prints the sampled cost of the first and the
last epoch.  Round 4: VRP-20 11.13 -> 5.80, IRP-20 13.21 -> 8.44."""
import os, sys, tempfile, logging
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vrp-gym_amd"), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
logging.disable(logging.CRITICAL)
import torch, agents
from gym_vrp.envs import VRPEnv, IRPEnv
d = tempfile.mkdtemp()
for cls, Env, kw in ((agents.VRPAgent, VRPEnv, dict(hidden_dim=200, num_heads=16, num_attention_layers=2)),
                     (agents.IRPAgent, IRPEnv, dict(hidden_dim=320, num_heads=4, num_attention_layers=4))):
    env = Env(num_nodes=20, batch_size=256, seed=69)
    a = cls(seed=69, csv_path=os.path.join(d, "l.csv"), **kw)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        a.train(env, epochs=30, check_point_dir=d + "/")
    rows = open(os.path.join(d, "l.csv")).read().strip().splitlines()
    print(cls.__name__, kw, rows[1].split(",")[2], "->", rows[-1].split(",")[2])
    os.remove(os.path.join(d, "l.csv"))

#!/usr/bin/env python3
"""Sanity: REINFORCE on N=20 instances (batch 256, the reference's train_models.py setting) for
a few hundred epochs; prints the mean sampled tour cost every 25 epochs.
usage: train_curve.py [epochs] [kind: 0 TSP, 1 VRP, 2 IRP]"""
import os, sys, tempfile, csv, time, logging
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
logging.disable(logging.CRITICAL)
import torch, agents
from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
Env = (TSPEnv, VRPEnv, IRPEnv)[kind]
Agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind]
d = tempfile.mkdtemp()
env = Env(num_nodes=20, batch_size=256, seed=69)
agent = Agent(seed=69, csv_path=os.path.join(d, "log.csv"))
import contextlib, io
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()) as out:
    agent.train(env, epochs=epochs, check_point_dir=d + "/")
dt = time.time() - t0
rows = list(csv.reader(open(os.path.join(d, "log.csv"))))[1:]
for r in rows[::25] + [rows[-1]]:
    print(f"epoch {int(r[0]):4d}  cost {-float(r[2]):7.4f}  advantage {float(r[3]):8.4f}")
print(f"{epochs} epochs in {dt:.1f} s ({dt/epochs*1e3:.1f} ms/epoch); baseline replaced "
      f"{out.getvalue().count('replacing')} times")
env_eval = Env(num_nodes=20, batch_size=256, seed=1234)
print("greedy eval cost after training:", -agent.evaluate(env_eval).mean().item())

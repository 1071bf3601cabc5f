#!/usr/bin/env python3
"""The reference's training sweep and evaluation, run with the package's own classes on one
MI355X: train_models.py:4-39 (seeds 69 / 123 x N 20 / 30 / 40 x TSP / VRP / IRP, batch 256,
851 epochs, Adam 1e-4, checkpoint every 50 epochs) followed by reproduction.py:17-57 /
reproduction.sh (the seed-69 model of epoch 850, loaded from its checkpoint file, greedy on
3 seeds x 256 fresh graphs, next to the RandomAgent; and the N = 20 models on N = 40 graphs).

    python tools/train_sweep.py OUTDIR [--epochs 851] [--seeds 69 123] [--nodes 20 30 40]

Writes OUTDIR/loss_log_{tsp,vrp,irp}_{N}_{seed}.csv (the reference's CSV schema),
OUTDIR/reproduction_results_{N}_nodes_model_{ENV}.csv, OUTDIR/reproduction_20_in_40_nodes_model_
{ENV}.csv and OUTDIR/summary.md: our numbers beside the reference's published ones (the constants
below are the means of /root/reference/reproduction_log/*.csv and the last rows of
/root/reference/train_logs/*.csv, GTX 1070 Ti)."""
import argparse
import contextlib
import csv
import io
import logging
import os
import sys
import time
from copy import deepcopy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
logging.disable(logging.CRITICAL)

import torch  # noqa: E402

import agents  # noqa: E402
from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv  # noqa: E402

ENVS = {"TSP": TSPEnv, "VRP": VRPEnv, "IRP": IRPEnv}
AGENTS = {"TSP": agents.TSPAgent, "VRP": agents.VRPAgent, "IRP": agents.IRPAgent}
# reference: mean "Mean Distance" over reproduction_log/reproduction_results_{N}_nodes_model_{ENV}.csv
# (3 seeds x 256 graphs; agent, random agent)
PUBLISHED_EVAL = {("TSP", 20): (4.164, 9.878), ("TSP", 30): (5.189, 15.105), ("TSP", 40): (6.146, 20.384),
                  ("VRP", 20): (4.293, 11.648), ("VRP", 30): (5.516, 17.105), ("VRP", 40): (6.521, 22.533),
                  ("IRP", 20): (7.019, 12.672), ("IRP", 30): (9.678, 18.779), ("IRP", 40): (11.711, 24.761)}
PUBLISHED_20_IN_40 = {"TSP": 6.120, "VRP": 6.300, "IRP": 11.879}
# reference: train_logs/loss_log_{env}_{N}_{seed}.csv rows 2 and 852 (Cost at epoch 0 / 850,
# seconds for the 851 epochs)
PUBLISHED_TRAIN = {
    ("TSP", 20, 69): (9.247, 4.323, 1394), ("TSP", 20, 123): (10.215, 4.391, 1623),
    ("TSP", 30, 69): (14.210, 5.498, 2714), ("TSP", 30, 123): (15.463, 5.509, 3145),
    ("TSP", 40, 69): (18.926, 6.648, 4622), ("TSP", 40, 123): (21.506, 6.681, 4783),
    ("VRP", 20, 69): (12.526, 4.496, 1479), ("VRP", 20, 123): (13.018, 4.545, 1795),
    ("VRP", 30, 69): (19.164, 5.966, 2881), ("VRP", 30, 123): (19.266, 5.900, 3389),
    ("VRP", 40, 69): (24.724, 7.049, 4892), ("VRP", 40, 123): (26.326, 7.122, 5079),
    ("IRP", 20, 69): (12.889, 7.241, 1785), ("IRP", 20, 123): (12.882, 7.299, 2257),
    ("IRP", 30, 69): (19.210, 9.859, 3720), ("IRP", 30, 123): (19.152, 9.896, 4086),
    ("IRP", 40, 69): (25.020, 12.152, 5706), ("IRP", 40, 123): (24.911, 12.145, 6425)}
EVAL_SEEDS = (1234, 2048, 2468)   # reproduction.py's default --seeds (reproduction_log/*.csv)


def train_one(out, name, N, seed, epochs):
    """train_models.py:10-21."""
    env = ENVS[name](num_nodes=N, batch_size=256, seed=seed)
    csv_path = os.path.join(out, f"loss_log_{name.lower()}_{N}_{seed}.csv")
    ckpt = os.path.join(out, "check_points", f"{name.lower()}_{N}_{seed}") + "/"
    agent = AGENTS[name](seed=seed, csv_path=csv_path)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()) as so:
        agent.train(env, epochs=epochs, check_point_dir=ckpt)
    torch.cuda.synchronize()
    dt = time.time() - t0
    rows = list(csv.reader(open(csv_path)))[1:]
    return {"first": -float(rows[0][2]), "last": -float(rows[-1][2]), "seconds": dt,
            "replaced": so.getvalue().count("replacing"), "ckpt": ckpt}


def reproduce(out, name, N, model_path, csv_name):
    """reproduction.py:17-57 (without the video)."""
    path = os.path.join(out, csv_name)
    with open(path, "w+", newline="") as fh:
        csv.writer(fh).writerow(["Model", "Seed", "Mean Distance"])
    tot_a = tot_r = n = 0
    for seed in EVAL_SEEDS:
        env = ENVS[name](num_nodes=N, batch_size=256, num_draw=6, seed=seed)
        env_r = deepcopy(env)
        agent = AGENTS[name](seed=seed)
        agent.model.load_state_dict(torch.load(model_path, map_location=agent.device))
        rnd = agents.RandomAgent(seed=seed)
        rnd.eval()
        loss_a = agent.evaluate(env)
        loss_r = rnd(env_r)
        with open(path, "a", newline="") as fh:
            w = csv.writer(fh)
            for a, r in zip(loss_a, loss_r):
                w.writerow([f"{name}-Agent", seed, -a.item()])
                w.writerow([f"{name}-Random-Agent", seed, -r.mean().item()])
        tot_a += -loss_a.sum().item()
        tot_r += -loss_r.sum().item()
        n += loss_a.numel()
    return tot_a / n, tot_r / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--epochs", type=int, default=851)
    ap.add_argument("--seeds", type=int, nargs="+", default=[69, 123])
    ap.add_argument("--nodes", type=int, nargs="+", default=[20, 30, 40])
    ap.add_argument("--envs", nargs="+", default=["TSP", "VRP", "IRP"])
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    last_epoch = (a.epochs - 1) // 50 * 50
    lines = ["| env | N | seed | cost epoch 0 (sampled) | cost last epoch (sampled) | reference log: first / last | "
             "seconds for the epochs (MI355X) | reference (GTX 1070 Ti) | baseline replaced |",
             "|---|---|---|---|---|---|---|---|---|"]
    trained = {}
    for seed in a.seeds:
        for N in a.nodes:
            for name in a.envs:
                r = train_one(a.out, name, N, seed, a.epochs)
                trained[(name, N, seed)] = r
                ref = PUBLISHED_TRAIN.get((name, N, seed))
                lines.append(f"| {name} | {N} | {seed} | {r['first']:.3f} | {r['last']:.3f} | "
                             f"{ref[0]:.3f} / {ref[1]:.3f} | {r['seconds']:.1f} | {ref[2]} | "
                             f"{r['replaced']} |" if ref else
                             f"| {name} | {N} | {seed} | {r['first']:.3f} | {r['last']:.3f} | - | "
                             f"{r['seconds']:.1f} | - | {r['replaced']} |")
                print(lines[-1], flush=True)
    ev = ["| env | N | greedy cost, model of epoch %d (3 seeds x 256 graphs) | reference | random agent | "
          "reference random |" % last_epoch, "|---|---|---|---|---|---|"]
    seed0 = a.seeds[0]
    for N in a.nodes:
        for name in a.envs:
            ck = trained[(name, N, seed0)]["ckpt"] + f"model_epoch_{last_epoch}.pt"
            if not os.path.exists(ck):
                continue
            ca, cr = reproduce(a.out, name, N, ck, f"reproduction_results_{N}_nodes_model_{name}.csv")
            pa, pr = PUBLISHED_EVAL[(name, N)]
            ev.append(f"| {name} | {N} | {ca:.3f} | {pa:.3f} | {cr:.3f} | {pr:.3f} |")
            print(ev[-1], flush=True)
    if 20 in a.nodes:
        for name in a.envs:
            ck = trained[(name, 20, seed0)]["ckpt"] + f"model_epoch_{last_epoch}.pt"
            if os.path.exists(ck):
                ca, _ = reproduce(a.out, name, 40, ck, f"reproduction_20_in_40_nodes_model_{name}.csv")
                ev.append(f"| {name} | 20 -> 40 | {ca:.3f} | {PUBLISHED_20_IN_40[name]:.3f} | | |")
                print(ev[-1], flush=True)
    with open(os.path.join(a.out, "summary.md"), "w") as fh:
        fh.write("# Training sweep (tools/train_sweep.py): train_models.py + reproduction.py settings\n\n"
                 "Costs are mean tour lengths (positive).  The reference's logs were written by an\n"
                 "earlier revision of its code (SURVEY.md section 6): magnitudes are comparable, the\n"
                 "instances and the weight-initialisation stream are the same as ours for a seed.\n\n"
                 "## Training\n\n" + "\n".join(lines) + "\n\n## Greedy evaluation (reproduction.py)\n\n"
                 + "\n".join(ev) + "\n")
    # the checkpoints are large (4.6 MB each): keep only what the evaluation used
    print("done")


if __name__ == "__main__":
    main()

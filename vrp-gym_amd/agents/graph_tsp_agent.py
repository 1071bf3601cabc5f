"""TSPModel / TSPAgent (reference: agents/graph_tsp_agent.py).

`TSPModel.forward(env, rollout)` is the hot path: ONE library call runs the encoder
and the whole decode/step loop on the GPU (agents/runtime.py -> vrp_rollout).  The
agent keeps the reference's REINFORCE-with-rollout-baseline training loop, CSV
schema and checkpoint cadence.
"""
import csv
import logging
import os
import time
from copy import deepcopy
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn
from scipy import stats

from . import runtime
from .graph_decoder import GraphDecoder
from .graph_encoder import GraphEncoder

logging.basicConfig(level=logging.INFO)


def default_device():
    return torch.device("cuda:%d" % torch.cuda.current_device()
                        if torch.cuda.is_available() else "cpu")


class TSPModel(nn.Module):
    ENV_KINDS = (0,)  # VRP_KIND_TSP

    def __init__(self, node_dim, emb_dim, hidden_dim, num_attention_layers, num_heads):
        super().__init__()
        self.device = default_device()
        self.encoder = GraphEncoder(node_input_dim=node_dim, embedding_dim=emb_dim,
                                    hidden_dim=hidden_dim,
                                    num_attention_layers=num_attention_layers,
                                    num_heads=num_heads)
        # the decoder always has 8 heads (graph_tsp_agent.py:53-55)
        self.decoder = GraphDecoder(emb_dim=emb_dim, num_heads=8, v_dim=emb_dim, k_dim=emb_dim)
        self.sampling_noise = "device"  # "host": draw Exp(1) on the CPU generator (parity)
        self.last_rollout = None

    def forward(self, env, rollout=False) -> Tuple[torch.Tensor, torch.Tensor]:
        """graph_tsp_agent.py:61-92 -> (acc_loss (B,), acc_log_prob (B,)) on the GPU.
        rollout=True: greedy; False: sampled (Categorical)."""
        if env.KIND not in self.ENV_KINDS:
            raise TypeError(f"{type(self).__name__} cannot drive a {type(env).__name__}")
        grad = self.training and torch.is_grad_enabled() and not rollout
        # somebody watches the tour (reproduction.py:37-47: enable_video_capturing, then
        # agent.evaluate; or sampler.graphs materialised for render()): the reference records
        # the edge and captures a frame inside every env.step (tsp.py:88-93).  The fused
        # rollout keeps the chosen nodes and replays that bookkeeping on the host afterwards.
        watched = env.video_save_path is not None or env.sampler._graphs is not None
        start = env.current_location if watched else None
        before = env.snapshot_state() if env.video_save_path is not None else None
        res = runtime.rollout(self, env, greedy=bool(rollout), train=self.training,
                              noise_mode=self.sampling_noise, record=grad, trace=watched)
        self.last_rollout = res
        self.decoder.reset()
        if watched:
            env.replay_tour(start, res.actions[: res.T].cpu().numpy(), before)
        logp = res.acc_logp
        if grad:
            logp = runtime.attach_grad(self, env, res)
        return res.acc_loss, logp


class TSPAgent:
    _MODEL = TSPModel

    def __init__(self, node_dim: int = 2, emb_dim: int = 128, hidden_dim: int = 512,
                 num_attention_layers: int = 3, num_heads: int = 8, lr: float = 1e-4,
                 csv_path: str = "loss_log.csv", seed=69, **model_kw):
        torch.manual_seed(seed)
        np.random.seed(seed)
        self.device = default_device()
        self.csv_path = csv_path
        arch = dict(node_dim=node_dim, emb_dim=emb_dim, hidden_dim=hidden_dim,
                    num_attention_layers=num_attention_layers, num_heads=num_heads)
        self._build(arch, model_kw, lr, first=TSPModel)

    def _build(self, arch, extra, lr, first=None):
        """Model + frozen baseline copy, built on the CPU then moved (so the initial
        weights depend only on the torch CPU stream, graph_tsp_agent.py:129-148)."""
        cls = first or self._MODEL
        kw = dict(arch, **extra) if cls is not TSPModel else arch
        self.model = cls(**kw).to(self.device)
        self.target_model = cls(**kw).to(self.device)
        self.target_model.load_state_dict(self.model.state_dict())
        self.target_model.eval()
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr)

    # ------------------------------------------------------------------ training
    def train(self, env, epochs: int = 100, eval_epochs: int = 1,
              check_point_dir: str = "./check_points/"):
        """REINFORCE with a rollout baseline (graph_tsp_agent.py:150-208).  Data parallel:
        every rank trains on its shard; the CSV, the log lines and the checkpoints are
        written by rank 0 only, with means taken over the whole (global) batch."""
        from . import distributed
        root = distributed.rank() == 0
        logging.info("Start Training")
        if root:
            with open(self.csv_path, "w+", newline="") as fh:
                csv.writer(fh).writerow(["Epoch", "Loss", "Cost", "Advantage", "Time"])
        t0 = time.time()
        for e in range(epochs):
            loss, cost, adv = distributed.global_means(*self.train_epoch(env, eval_epochs))
            if root:
                logging.info(f"Epoch {e} finished - Loss: {loss}, Advantage: {adv} Dist: {cost}")
                with open(self.csv_path, "a", newline="") as fh:
                    csv.writer(fh).writerow([e, loss, cost, adv, time.time() - t0])
            self.save_model(episode=e, check_point_dir=check_point_dir)

    def train_epoch(self, env, eval_epochs: int = 1):
        """One iteration of the reference's training loop (graph_tsp_agent.py:174-189):
        sampled model + baseline rollouts, REINFORCE loss, backward (HIP), gradient
        all-reduce, Adam, baseline t-test.  Returns this shard's (loss, mean cost, mean
        advantage) as 0-d device tensors (no host sync)."""
        self.model.train()
        loss_m, loss_b, log_prob = self.step(env, (False, True))
        advantage = (loss_m - loss_b) * -1
        loss = (advantage * log_prob).mean()
        self.opt.zero_grad()
        loss.backward()
        self.reduce_gradients()
        self.opt.step()
        self.baseline_update(env, eval_epochs)
        return loss.detach(), loss_m.mean(), advantage.mean()

    def sync_weights(self):
        """Call after writing parameters through `.data` (or any other route that bypasses
        autograd's version counters): drops the folded decoder matrices derived from the old
        values.  Optimizer steps, load_state_dict, `.to()` and in-place ops on the parameters
        themselves are noticed automatically."""
        for model in (self.model, self.target_model):
            for m in model.modules():
                runtime.invalidate(m)

    def reduce_gradients(self):
        """Data parallel: one flat RCCL all-reduce of the gradient."""
        from . import distributed
        distributed.allreduce_gradients(self.model)

    def save_model(self, episode: int, check_point_dir: str) -> None:
        """state_dict every 50 epochs (graph_tsp_agent.py:210-225)."""
        os.makedirs(check_point_dir, exist_ok=True)
        if episode % 50 == 0 and episode != 0:
            from . import distributed
            distributed.average_buffers(self.model)
            if distributed.is_distributed() and torch.distributed.get_rank() != 0:
                return
            torch.save(self.model.state_dict(), check_point_dir + f"model_epoch_{episode}.pt")

    def step(self, env, rollouts: Tuple[bool, bool]):
        """Reset, play model and baseline on identical instances
        (graph_tsp_agent.py:227-255).  QUIRK kept: the baseline also uses rollouts[0]."""
        if hasattr(env, "twin"):     # device-resident env: no host-side state needed
            env.reset(return_state=False)
            env_baseline = env.twin()
        else:
            env.reset()
            env_baseline = deepcopy(env)
        loss, log_prob = self.model(env, rollouts[0])
        with torch.no_grad():
            loss_b, _ = self.target_model(env_baseline, rollouts[0])
        return loss, loss_b, log_prob

    def evaluate(self, env):
        """Greedy rollout of the current model, eval-mode BN (graph_tsp_agent.py:257-273)."""
        self.model.eval()
        with torch.no_grad():
            loss, _ = self.model(env, rollout=True)
        return loss

    def baseline_update(self, env, batch_steps: int = 3):
        """Paired t-test of model vs baseline on fresh instances
        (graph_tsp_agent.py:275-306)."""
        logging.info("Update Baseline")
        self.model.eval()
        self.target_model.eval()
        cur, base = [], []
        with torch.no_grad():
            for _ in range(batch_steps):
                loss, loss_b, _ = self.step(env, [True, True])
                cur.append(loss)
                base.append(loss_b)
        cur, base = torch.cat(cur), torch.cat(base)
        from . import distributed
        cur, base = distributed.gather_costs(cur, base)
        advantage = ((cur - base) * -1).mean()
        _, p_value = stats.ttest_rel(cur.tolist(), base.tolist())
        if advantage.item() <= 0 and p_value <= 0.05:
            print("replacing baceline")
            distributed.average_buffers(self.model)
            self.target_model.load_state_dict(self.model.state_dict())

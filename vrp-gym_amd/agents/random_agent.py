"""RandomAgent (reference: agents/random_agent.py:15-41).

Host-driven on purpose: it is the reference's CPU plumbing case (BASELINE config 1) and
exercises the env's host-facing `get_state()` / `step()` surface.  Per step and per graph,
in graph order, one node is drawn uniformly among the unmasked ones with the GLOBAL numpy
stream (`np.random.choice(candidates, 1)`), which is what makes a seed reproduce the
reference's tours exactly.
"""
import numpy as np
import torch
import torch.nn as nn


def _graph_state(state):
    """IRPEnv hands back (graph_state, load); the mask is the last column either way."""
    return state[0] if isinstance(state, tuple) else state


def _draw_actions(mask_column):
    picks = np.empty(mask_column.shape[0], dtype=np.int64)
    for g, row in enumerate(mask_column):
        candidates = np.flatnonzero(row == 0)
        picks[g] = np.random.choice(candidates, 1)[0]
    return picks.reshape(-1, 1)


class RandomAgent(nn.Module):
    def __init__(self, seed: int = 69):
        super().__init__()
        np.random.seed(seed)

    def forward(self, env):
        graph_state = _graph_state(env.get_state())
        total = torch.zeros(graph_state.shape[0])
        finished = False
        while not finished:
            state, reward, finished, _ = env.step(_draw_actions(graph_state[:, :, -1]))
            total += torch.tensor(reward, dtype=torch.float)
            graph_state = _graph_state(state)
        return total

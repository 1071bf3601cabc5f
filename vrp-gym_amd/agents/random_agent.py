"""RandomAgent (reference: agents/random_agent.py): uniform choice among the
unmasked nodes, per graph, from numpy's global stream.  Host-driven on purpose —
it is the reference's CPU plumbing case (BASELINE config 1) and exercises the
env's host-facing step()/get_state() surface."""
import numpy as np
import torch
import torch.nn as nn


class RandomAgent(nn.Module):
    def __init__(self, seed: int = 69):
        super().__init__()
        np.random.seed(seed)

    def forward(self, env):
        state = env.get_state()
        if isinstance(state, tuple):  # IRPEnv returns (graph_state, load)
            state = state[0]
        acc_loss = torch.zeros(size=(state.shape[0],))
        done = False
        while not done:
            if isinstance(state, tuple):
                state = state[0]
            free = state[:, :, -1] == 0
            actions = np.array([np.random.choice(np.flatnonzero(row), 1)[0] for row in free])
            state, loss, done, _ = env.step(actions[:, None])
            acc_loss += torch.tensor(loss, dtype=torch.float)
        return acc_loss

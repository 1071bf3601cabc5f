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
    def __init__(self, seed: int = 69, on_device: bool = False):
        """on_device=True (an extension): the whole random rollout runs on
        the GPU (vrp_random_rollout: Philox draws, same distribution, NOT the reference's
        numpy stream) and returns a device tensor -- for throughput runs at large batch."""
        super().__init__()
        np.random.seed(seed)
        self._seed, self._episode, self._on_device = int(seed), 0, bool(on_device)

    def _forward_device(self, env):
        import ctypes as C
        import vrpgym_hip as hip
        from .runtime import max_steps_for
        lib = hip.require_gpu()
        B, N, dev = env.batch_size, env.num_nodes, env._device
        steps = max_steps_for(env.KIND, N)
        acc = torch.empty((B,), dtype=torch.float32, device=dev)
        notdone = torch.empty((steps + 1,), dtype=torch.int32, device=dev)
        env._sync_positions()
        env._parity = 0
        cenv = env._cenv()
        hip.check(lib.vrp_random_rollout(C.byref(cenv), self._seed, self._episode,
                                         env._slice.start, steps, acc.data_ptr(),
                                         notdone.data_ptr(), None, hip.current_stream(dev)))
        self._episode += 1
        env._mask_fresh = False
        nd = notdone[:steps].cpu()
        zero = (nd == 0).nonzero()
        env._step_count += int(zero[0].item()) + 1 if len(zero) else steps
        return acc

    def forward(self, env):
        if self._on_device:
            return self._forward_device(env)
        graph_state = _graph_state(env.get_state())
        total = torch.zeros(graph_state.shape[0])
        finished = False
        while not finished:
            state, reward, finished, _ = env.step(_draw_actions(graph_state[:, :, -1]))
            total += torch.tensor(reward, dtype=torch.float)
            graph_state = _graph_state(state)
        return total

"""IRPEnv — capacity/demand variant (reference: gym_vrp/envs/irp.py).
State (B,N,5) = [x, y, demand, is_depot, mask] plus the vehicle load (B,)."""
from typing import Tuple

import numpy as np

from .tsp import TSPEnv


class IRPEnv(TSPEnv):
    KIND = 2  # VRP_KIND_IRP
    _PLOT_DEMAND = True

    def __init__(self, num_nodes: int = 32, batch_size: int = 128, num_draw: int = 6,
                 seed: int = 69, device=None, shard=None, generator: str = "numpy"):
        super().__init__(num_nodes=num_nodes, batch_size=batch_size, num_draw=num_draw,
                         seed=seed, device=device, shard=shard, generator=generator)

    def generate_graphs(self):
        """irp.py:157-174."""
        super().generate_graphs()
        self._demands_host = None if self._generator == "device" else self.sampler.get_demands()

    @property
    def demands(self):
        if self._demands_host is None:
            self._demands_host = self.sampler.get_demands()
        return self._demands_host

    @demands.setter
    def demands(self, value):
        self._demands_host = value

    def reset(self, return_state: bool = True):
        """irp.py:176-185: load := 1 after the new instances are in place."""
        self.step_count = 0
        self.generate_graphs()
        return self.get_state() if return_state else None

    @property
    def load(self):
        return self._load.cpu().numpy()

    @load.setter
    def load(self, value):
        self._load.copy_(self._torch.from_numpy(np.asarray(value, dtype=np.float64)))
        self._mask_fresh = False

    def get_state(self) -> Tuple[np.ndarray, np.ndarray]:
        """irp.py:101-124."""
        mask = self.generate_mask()
        is_depot = np.zeros((self.batch_size, self.num_nodes))
        is_depot[np.arange(self.batch_size), self.depots[:, 0]] = 1
        state = np.dstack([self.sampler.get_graph_positions(), self.demands[:, :, 0], is_depot,
                           mask])
        return state, self.load

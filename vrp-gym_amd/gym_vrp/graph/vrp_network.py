"""Batch of instances (reference: gym_vrp/graph/vrp_network.py).

Owns the host copies of the instance arrays; `graphs[i]` are lazy views.  The
environment uploads the arrays to the GPU once per reset; nothing here is on the
per-step path.
"""
import numpy as np

from .instances import draw_instances
from .vrp_graph import VRPGraph


class VRPNetwork:
    def __init__(self, num_graphs, num_nodes, num_depots, plot_demand=False, _arrays=None):
        assert num_nodes >= num_depots, "Number of depots should be lower than number of depots"
        self.num_nodes, self.num_depots, self.num_graphs = num_nodes, num_depots, num_graphs
        self.plot_demand = plot_demand
        if _arrays is None:
            _arrays = draw_instances(num_graphs, num_nodes, num_depots)
        # a callable defers the host copies until somebody looks (instances drawn on the GPU)
        self._fetch = _arrays if callable(_arrays) else None
        self._host = None if callable(_arrays) else tuple(_arrays)
        self._graphs = None
        self._dirty = False  # set when a caller rewrites coordinates through a view
        self._on_change = None

    def _arrays(self):
        if self._host is None:
            self._host = tuple(self._fetch())
            self._fetch = None
        return self._host

    @property
    def _pos(self):
        return self._arrays()[0]

    @property
    def _depots(self):
        return self._arrays()[1]

    @property
    def _demands(self):
        return self._arrays()[2]

    @property
    def graphs(self):
        if self._graphs is None:
            self._graphs = [VRPGraph(self.num_nodes, self.num_depots, self.plot_demand,
                                     _owner=self, _index=i) for i in range(self.num_graphs)]
        return self._graphs

    def _positions_changed(self, index):
        self._dirty = True
        if self._on_change is not None:
            self._on_change()

    def get_distance(self, graph_idx, node_idx_1, node_idx_2):
        return np.linalg.norm(self._pos[graph_idx, node_idx_1] - self._pos[graph_idx, node_idx_2])

    def get_distances(self, paths):
        paths = np.asarray(paths)
        rows = np.arange(len(paths))
        d = self._pos[rows, paths[:, 0]] - self._pos[rows, paths[:, 1]]
        return np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])

    def get_depots(self):
        return np.array(self._depots, dtype=int)

    def get_demands(self):
        return np.array(self._demands)

    def get_graph_positions(self):
        return np.array(self._pos)

    def visit_edges(self, transition_matrix):
        if self._graphs is None:
            return  # nobody is looking at the per-graph views: nothing to record
        for i, row in enumerate(transition_matrix):
            self._graphs[i].visit_edge(int(row[0]), int(row[1]))

    def draw(self, graph_idxs):
        import matplotlib.pyplot as plt
        num_columns = min(len(graph_idxs), 3)
        num_rows = int(np.ceil(len(graph_idxs) / num_columns))
        plt.clf()
        fig = plt.figure(figsize=(5 * num_columns, 5 * num_rows))
        for n, graph_idx in enumerate(graph_idxs):
            ax = plt.subplot(num_rows, num_columns, n + 1)
            self.graphs[graph_idx].draw(ax=ax)
        fig.canvas.draw()
        image = np.asarray(fig.canvas.buffer_rgba())[..., :3].copy()
        plt.close(fig)
        return image

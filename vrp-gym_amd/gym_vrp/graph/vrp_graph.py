"""Host-side view of ONE instance, for rendering and for callers that poke at
`env.sampler.graphs[i]` (reference: gym_vrp/graph/vrp_graph.py).

The hot path never touches these objects; they are thin views over the arrays
owned by the environment (or over private arrays when built stand-alone).
"""
import numpy as np

from .instances import demand_scale


class _NodeData(dict):
    """Attribute dict of one node; writing "coordinates" updates the owner."""

    def __init__(self, graph, idx):
        super().__init__()
        self._g, self._i = graph, idx

    def __getitem__(self, key):
        g, i = self._g, self._i
        if key == "coordinates":
            return g._pos[i]
        if key == "depot":
            return float(i in set(np.atleast_1d(g.depots).tolist()))
        if key == "demand":
            return g._demand[i]
        if key == "node_color":
            return "red" if i in set(np.atleast_1d(g.depots).tolist()) else "black"
        raise KeyError(key)

    def __setitem__(self, key, value):
        if key == "coordinates":
            self._g._set_position(self._i, np.asarray(value, dtype=np.float64))
        elif key == "demand":
            self._g._demand[self._i] = value
        else:
            raise KeyError(f"attribute {key!r} is read-only in the device-resident graph")

    def keys(self):
        return ["coordinates", "depot", "demand", "node_color"]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return 4

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class _NodeView:
    """Enough of networkx's NodeView for `nx.set_node_attributes(graph, {...}, name)`
    and `len(graph.nodes)` (reference tests/test_env.py:31-41)."""

    def __init__(self, graph):
        self._g = graph

    def __len__(self):
        return self._g.num_nodes

    def __iter__(self):
        return iter(range(self._g.num_nodes))

    def __getitem__(self, idx):
        if not 0 <= idx < self._g.num_nodes:
            raise KeyError(idx)
        return _NodeData(self._g, int(idx))

    def __contains__(self, idx):
        return 0 <= idx < self._g.num_nodes

    def data(self):
        return [(i, dict(self[i].items())) for i in range(self._g.num_nodes)]


class VRPGraph:
    def __init__(self, num_nodes, num_depots, plot_demand=False, _owner=None, _index=None):
        self.num_nodes = num_nodes
        self.num_depots = num_depots
        self.plot_demand = plot_demand
        self.offset = np.array([0, 0.065])
        self._owner, self._index = _owner, _index
        if _owner is None:
            # stand-alone: same three draws as the reference (vrp_graph.py:28-43)
            self._pos = np.random.rand(num_nodes, 2)
            self.depots = np.random.choice(num_nodes, size=num_depots, replace=False)
            self._demand = np.random.uniform(1, 10, size=(num_nodes, 1)) / demand_scale(num_nodes)
            self._demand[self.depots] = 0
        else:
            self._pos = _owner._pos[_index]
            self.depots = _owner._depots[_index]
            self._demand = _owner._demands[_index]
        self.visited_edges = set()

    # --- views -------------------------------------------------------------
    @property
    def nodes(self):
        return _NodeView(self)

    # nx.set_node_attributes(G, values, name) dispatches on G.nodes via these:
    @property
    def _node(self):
        return self.nodes

    @property
    def graph(self):
        return self

    def is_multigraph(self):
        return False

    def is_directed(self):
        return False

    @property
    def edges(self):
        return [(i, j, {"visited": (min(i, j), max(i, j)) in self.visited_edges})
                for i in range(self.num_nodes) for j in range(i + 1, self.num_nodes)]

    @property
    def node_positions(self):
        return np.asarray(self._pos)

    @property
    def demand(self):
        return np.asarray(self._demand)

    def _set_position(self, idx, xy):
        self._pos[idx] = xy
        if self._owner is not None:
            self._owner._positions_changed(self._index)

    # --- reference API -------------------------------------------------------
    def visit_edge(self, source_node, target_node):
        if source_node != target_node:
            self.visited_edges.add((min(source_node, target_node), max(source_node, target_node)))

    def euclid_distance(self, node1_idx, node2_idx):
        return np.linalg.norm(self._pos[node1_idx] - self._pos[node2_idx])

    def draw(self, ax):
        """Matplotlib rendering of one instance (reference vrp_graph.py:62-96)."""
        pos = self.node_positions
        dep = set(np.atleast_1d(self.depots).tolist())
        colors = ["red" if i in dep else "black" for i in range(self.num_nodes)]
        ax.scatter(pos[:, 0], pos[:, 1], c=colors, s=100)
        for (i, j) in sorted(self.visited_edges):
            ax.plot([pos[i, 0], pos[j, 0]], [pos[i, 1], pos[j, 1]], color="red", alpha=0.5,
                    linewidth=1.5)
        if self.plot_demand:
            for i in range(self.num_nodes):
                ax.annotate(str(np.round(self._demand[i], 2)[0]), pos[i] + self.offset)

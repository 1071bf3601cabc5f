"""Oracle environments: vectorised numpy restatement of gym_vrp/envs + gym_vrp/graph.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Reference citations are relative
to /root/reference.
"""
import numpy as np

TSP, VRP, IRP = 0, 1, 2
KIND_NAMES = {TSP: "tsp", VRP: "vrp", IRP: "irp"}


def draw_instances(batch_size, num_nodes):
    """Instance sampling from the global legacy numpy stream.

    Follows VRPGraph.__init__ (gym_vrp/graph/vrp_graph.py:28-43) called B times
    by VRPNetwork.__init__ (gym_vrp/graph/vrp_network.py:41-42): per graph, in
    this order, rand(N,2) -> choice(N,1,replace=False) -> uniform(1,10,(N,1))/C,
    depot demand := 0.  Demand is drawn for every env kind (it advances the
    stream even when unused).
    """
    pos = np.empty((batch_size, num_nodes, 2), dtype=np.float64)
    depots = np.empty((batch_size, 1), dtype=np.int64)
    demands = np.empty((batch_size, num_nodes, 1), dtype=np.float64)
    scale = 0.2449 * num_nodes + 26.12  # vrp_graph.py:41
    for b in range(batch_size):
        pos[b] = np.random.rand(num_nodes, 2)  # vrp_graph.py:29
        dep = np.random.choice(num_nodes, size=1, replace=False)  # :34
        dem = np.random.uniform(low=1, high=10, size=(num_nodes, 1)) / scale  # :42
        dem[dep] = 0  # :43
        depots[b] = dep
        demands[b] = dem
    return pos, depots, demands


class OracleEnv:
    """One class for the three env kinds; `kind` selects the mask rule.

    State layout and call order follow TSPEnv (gym_vrp/envs/tsp.py:27-174),
    VRPEnv.generate_mask (gym_vrp/envs/vrp.py:13-37) and IRPEnv
    (gym_vrp/envs/irp.py:28-185).
    """

    def __init__(self, kind, num_nodes=20, batch_size=128, num_draw=6, seed=69):
        assert num_draw <= batch_size  # tsp.py:44-46
        np.random.seed(seed)  # tsp.py:48
        self.kind = kind
        self.step_count = 0
        self.num_nodes = num_nodes
        self.batch_size = batch_size
        self.draw_idxs = np.random.choice(batch_size, num_draw, replace=False)  # :55
        self.generate_graphs()
        if kind == IRP:
            self.load = np.ones(shape=(batch_size,))  # irp.py:47

    # -- E1 ---------------------------------------------------------------
    def generate_graphs(self):
        """tsp.py:162-174 / irp.py:157-174."""
        self.visited = np.zeros((self.batch_size, self.num_nodes))
        self.pos, self.depots, self.demands = draw_instances(
            self.batch_size, self.num_nodes
        )
        self.current_location = self.depots

    def reset(self):
        """tsp.py:150-160 / irp.py:176-185 (no reseed)."""
        self.step_count = 0
        self.generate_graphs()
        state = self.get_state()
        if self.kind == IRP:
            self.load = np.ones(shape=(self.batch_size,))
            state = self.get_state()
        return state

    # -- E4/E5/E6 ---------------------------------------------------------
    def step(self, actions):
        """tsp.py:60-101 / irp.py:49-99."""
        assert actions.shape[0] == self.batch_size
        self.step_count += 1
        rows = np.arange(self.batch_size)
        a = np.asarray(actions).reshape(self.batch_size).astype(np.int64)
        self.visited[rows, a] = 1  # tsp.py:86
        src = self.current_location.reshape(self.batch_size).astype(np.int64)
        if self.kind == IRP:
            self.load = self.load - self.demands[rows, a, 0]  # irp.py:80-85
            self.load[a == self.depots[:, 0]] = 1  # irp.py:86
        self.current_location = np.array(actions).reshape(self.batch_size, 1)
        done = self.is_done()  # evaluated BEFORE the mask fix-ups (tsp.py:95)
        # E6: vrp_graph.py:137-146, np.linalg.norm of a 2-vector in fp64
        d = self.pos[rows, src] - self.pos[rows, a]
        dist = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
        return self.get_state(), -dist, done, None

    def is_done(self):
        return bool(np.all(self.visited == 1))  # tsp.py:103-104

    # -- E7/E8 ------------------------------------------------------------
    def generate_mask(self):
        """tsp.py:131-148, vrp.py:13-37, irp.py:126-155.  Mutates self.visited."""
        rows = np.arange(self.batch_size)
        dep = self.depots[:, 0]
        at_depot = self.current_location[:, 0] == dep
        self.visited[rows[at_depot], dep[at_depot]] = 1
        if self.kind != TSP:
            self.visited[rows[~at_depot], dep[~at_depot]] = 0
        solved = np.all(self.visited, axis=1)
        self.visited[rows[solved], dep[solved]] = 0
        if self.kind != IRP:
            return self.visited
        mask = np.copy(self.visited)
        exceed = (self.demands[:, :, 0] - self.load[:, None]) > 0  # irp.py:152
        mask[exceed] = 1
        return mask

    # -- E3 ---------------------------------------------------------------
    def get_state(self):
        """tsp.py:106-129 / irp.py:101-124."""
        mask = self.generate_mask()
        is_depot = np.zeros((self.batch_size, self.num_nodes))
        is_depot[np.arange(self.batch_size), self.depots[:, 0]] = 1
        if self.kind == IRP:
            state = np.dstack([self.pos, self.demands[:, :, 0], is_depot, mask])
            return state, self.load
        return np.dstack([self.pos, is_depot, mask])


def make_env(kind, *args, **kw):
    return OracleEnv(kind, *args, **kw)

"""Host-side instance sampling in the reference's RNG order.

The reference draws every instance from numpy's *global legacy* stream, one graph
at a time: `rand(N,2)` -> `choice(N, num_depots, replace=False)` ->
`uniform(1,10,(N,1)) / (0.2449*N + 26.12)` with the depots' demand forced to 0
(gym_vrp/graph/vrp_graph.py:28-43, called B times by
gym_vrp/graph/vrp_network.py:41-42).  The three calls are kept verbatim so that a
seed reproduces the reference's instances bit for bit; only the networkx graph
objects around them are gone (the arrays go straight to device tensors).
"""
import numpy as np


def demand_scale(num_nodes):
    """vrp_graph.py:41 — linear fit of the capacities used by Kool et al."""
    return 0.2449 * num_nodes + 26.12


def _draw_native(num_graphs, num_nodes, keep=None):
    """The same stream replayed by libvrpgym_hip (csrc/instances.hip): numpy's generator
    state goes in, comes back advanced exactly as the three numpy calls per graph would
    leave it.  ~100x faster than the Python loop; falls back to it when the library is
    not built (host-only code path: no GPU needed either way).  keep = (first, count): only
    those graphs are stored (a rank's shard); the stream still advances over all of them."""
    import ctypes as C
    try:
        from vrpgym_hip import lib
        fn = lib().vrp_draw_instances_host_range
    except Exception:
        return None
    st = np.random.get_state()
    if st[0] != "MT19937":
        return None
    first, count = keep if keep is not None else (0, num_graphs)
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos_state = C.c_int32(int(st[2]))
    pos = np.empty((count, num_nodes, 2), dtype=np.float64)
    depots = np.empty((count, 1), dtype=np.int64)
    demands = np.empty((count, num_nodes, 1), dtype=np.float64)
    rc = fn(key.ctypes.data, C.addressof(pos_state), num_graphs, num_nodes, first, count,
            pos.ctypes.data, depots.ctypes.data, demands.ctypes.data)
    if rc != 0:
        return None
    np.random.set_state((st[0], key, int(pos_state.value), st[3], st[4]))
    return pos, depots, demands


def draw_instances(num_graphs, num_nodes, num_depots=1, native=True, keep=None):
    """Returns pos (B,N,2) f64, depots (B,num_depots) i64, demands (B,N,1) f64 -- of all
    `num_graphs` graphs, or with keep = (first, count) of that slice only (the global stream
    advances over all graphs either way, so every rank of a sharded run ends up with the same
    generator state the unsharded run would have)."""
    assert num_nodes >= num_depots, "Number of depots should be lower than number of depots"
    if native and num_depots == 1:
        out = _draw_native(num_graphs, num_nodes, keep)
        if out is not None:
            return out
    pos = np.empty((num_graphs, num_nodes, 2), dtype=np.float64)
    depots = np.empty((num_graphs, num_depots), dtype=np.int64)
    demands = np.empty((num_graphs, num_nodes, 1), dtype=np.float64)
    scale = demand_scale(num_nodes)
    rand, choice, uniform = np.random.rand, np.random.choice, np.random.uniform
    for g in range(num_graphs):
        pos[g] = rand(num_nodes, 2)
        depots[g] = choice(num_nodes, size=num_depots, replace=False)
        demands[g] = uniform(low=1, high=10, size=(num_nodes, 1)) / scale
        demands[g, depots[g]] = 0
    if keep is not None:
        s = slice(keep[0], keep[0] + keep[1])
        return pos[s], depots[s], demands[s]
    return pos, depots, demands


def shard_bounds(num_graphs, rank, world_size):
    """Rank r owns instances [r*B/R, (r+1)*B/R) of the seed-ordered set."""
    assert num_graphs % world_size == 0, "batch must divide evenly over ranks"
    per = num_graphs // world_size
    return rank * per, (rank + 1) * per

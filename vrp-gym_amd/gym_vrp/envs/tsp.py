"""TSPEnv — device-resident batched environment (reference: gym_vrp/envs/tsp.py).

Public surface identical to the reference (constructor, step/reset/get_state/
generate_mask/is_done/render/enable_video_capturing, attributes visited, depots,
current_location, sampler, step_count, num_nodes, batch_size, draw_idxs).  State
lives in contiguous device tensors; `step` is one launch of the fused HIP kernel
`vrp_env_step` (include/vrpgym_hip.h).  Host numpy views are materialised lazily
for foreign callers (RandomAgent, tests); the agents of this package never leave
the device (agents/runtime.py).
"""
import copy
from typing import Tuple, Union

import numpy as np

from ..graph.instances import draw_instances, shard_bounds
from ..graph.vrp_network import VRPNetwork
from .common import GymEnv, ObsType


class TSPEnv(GymEnv):
    metadata = {"render.modes": ["human", "rgb_array"]}
    KIND = 0  # VRP_KIND_TSP
    _PLOT_DEMAND = False

    def __init__(self, num_nodes: int = 20, batch_size: int = 128, num_draw: int = 6,
                 seed: int = 69, device=None, shard=None, generator: str = "numpy"):
        """Same arguments as the reference (tsp.py:27-58).  Extra, optional:
        device — torch device of the state (default: current CUDA device);
        shard  — (rank, world_size): keep only this rank's slice of the
                 seed-ordered instance stream;
        generator — "numpy": the reference's instances, bit for bit (global legacy numpy
                 stream, replayed natively on the host); "device": the same distributions
                 drawn on the GPU by a counter-based Philox stream (throughput runs at
                 large batch: no host work, no upload, but NOT the reference's instances)."""
        assert generator in ("numpy", "device"), generator
        self._generator = generator
        self._seed = int(seed)
        self._episode = 0
        assert (
            num_draw <= batch_size
        ), "Num_draw needs to be equal or lower than the number of generated graphs."
        import torch
        from vrpgym_hip import require_gpu
        self._lib = require_gpu()  # loud failure without GPU / library: no CPU fallback
        self._torch = torch
        self._device = torch.device(device) if device is not None else torch.device(
            "cuda", torch.cuda.current_device())
        np.random.seed(seed)  # tsp.py:48
        self._step_count = 0
        self._last_rollout = None
        self.num_nodes = num_nodes
        self._global_batch = batch_size
        self._shard = shard
        if shard is not None:
            lo, hi = shard_bounds(batch_size, *shard)
            self._slice = slice(lo, hi)
            self.batch_size = hi - lo
        else:
            self._slice = slice(0, batch_size)
            self.batch_size = batch_size
        self.draw_idxs = np.random.choice(batch_size, num_draw, replace=False)  # tsp.py:55
        if shard is not None:
            self.draw_idxs = self.draw_idxs[self.draw_idxs < self.batch_size]
        self.video_save_path = None
        self._alloc()
        self.generate_graphs()

    # ------------------------------------------------------------------ device state
    def _alloc(self):
        t, B, N, dev = self._torch, self.batch_size, self.num_nodes, self._device
        self._pos = t.empty((B, N, 2), dtype=t.float64, device=dev)
        self._demand = t.zeros((B, N), dtype=t.float64, device=dev)
        self._depot = t.empty((B,), dtype=t.int32, device=dev)
        self._visited = t.zeros((B, N), dtype=t.uint8, device=dev)
        self._mask = t.zeros((2, B, N), dtype=t.uint8, device=dev)
        self._cur = t.empty((B,), dtype=t.int32, device=dev)
        self._load = t.ones((B,), dtype=t.float64, device=dev)
        self._reward = t.empty((B,), dtype=t.float64, device=dev)
        self._notdone = t.zeros((1,), dtype=t.int32, device=dev)
        self._actions = t.empty((B,), dtype=t.int64, device=dev)
        self._parity = 0        # mask buffer holding the current state's mask column
        self._mask_fresh = False  # fix-ups of generate_mask applied since the last change?

    def _cenv(self):
        from vrpgym_hip import Env
        e = Env()
        e.kind, e.B, e.N = self.KIND, self.batch_size, self.num_nodes
        e.pos, e.demand, e.depot = self._pos.data_ptr(), self._demand.data_ptr(), self._depot.data_ptr()
        e.visited, e.mask = self._visited.data_ptr(), self._mask.data_ptr()
        e.cur, e.load = self._cur.data_ptr(), self._load.data_ptr()
        return e

    def _stream(self):
        return self._torch.cuda.current_stream(self._device).cuda_stream

    def _upload_instances(self):
        t, s = self._torch, self._slice
        net = self.sampler
        self._pos.copy_(t.from_numpy(np.ascontiguousarray(net._pos)))
        self._demand.copy_(t.from_numpy(np.ascontiguousarray(net._demands[:, :, 0])))
        self._depot.copy_(t.from_numpy(net._depots[:, 0].astype(np.int32)))
        self._pos_dirty = False

    def _sync_positions(self):
        """Callers may rewrite coordinates through sampler.graphs[i] (the reference reads
        them lazily every step, tests/test_env.py:31-36); push them before the next use."""
        if self.sampler._dirty:
            self._pos.copy_(self._torch.from_numpy(np.ascontiguousarray(self.sampler._pos)))
            self.sampler._dirty = False

    # ------------------------------------------------------------------ E1
    def generate_graphs(self):
        """tsp.py:162-174: new instances from the global numpy stream, visited := 0,
        current_location := depots."""
        if self._generator == "device":
            self._draw_on_device()
        else:
            # a shard keeps its own rows only; the rest of the global stream is drawn and
            # discarded natively (no (B_global, N, 2) arrays per rank)
            s = self._slice
            pos, depots, demands = draw_instances(self._global_batch, self.num_nodes, 1,
                                                  keep=(s.start, s.stop - s.start))
            self.sampler = VRPNetwork(self.batch_size, self.num_nodes, 1,
                                      plot_demand=self._PLOT_DEMAND,
                                      _arrays=(pos, depots, demands))
            self._depots_host = self.sampler.get_depots()
            self._upload_instances()
        self._reset_state()

    def _reset_state(self):
        """Start-of-episode state on the instances in place (one launch)."""
        from vrpgym_hip import check
        check(self._lib.vrp_env_reset(self._cenv(), self._stream()))
        self._parity = 0
        self._mask_fresh = False

    def _draw_on_device(self):
        """vrp_draw_instances_device straight into the state tensors; the host-side views
        (sampler.graphs, depots, demands) are fetched only if somebody reads them."""
        from vrpgym_hip import check
        check(self._lib.vrp_draw_instances_device(
            self._seed, self._episode, self._slice.start, self.batch_size, self.num_nodes,
            self._pos.data_ptr(), self._depot.data_ptr(), self._demand.data_ptr(),
            self._stream()))
        self._episode += 1
        pos_t, dep_t, dem_t = self._pos, self._depot, self._demand

        def fetch():
            return (pos_t.cpu().numpy(), dep_t.cpu().numpy().astype(np.int64)[:, None],
                    dem_t.cpu().numpy()[:, :, None])

        self.sampler = VRPNetwork(self.batch_size, self.num_nodes, 1,
                                  plot_demand=self._PLOT_DEMAND, _arrays=fetch)
        self._depots_host = None

    @property
    def depots(self):
        if self._depots_host is None:
            self._depots_host = self.sampler.get_depots()
        return self._depots_host

    @depots.setter
    def depots(self, value):
        self._depots_host = value

    def reset(self, return_state: bool = True) -> Union[ObsType, Tuple[ObsType, dict]]:
        """tsp.py:150-160 (no reseed).  return_state=False skips building the host-side
        state array (the device-resident agents never read it)."""
        self.step_count = 0
        self.generate_graphs()
        return self.get_state() if return_state else None

    @property
    def step_count(self):
        """Steps taken since the last reset (device rollouts report theirs lazily)."""
        extra = self._last_rollout.T if self._last_rollout is not None else 0
        return self._step_count + extra

    @step_count.setter
    def step_count(self, value):
        self._step_count = value
        self._last_rollout = None

    # ------------------------------------------------------------------ E4-E6
    def step(self, actions: np.ndarray) -> Tuple[ObsType, float, bool, dict]:
        """tsp.py:60-101.  `actions` (B,1) host integers (or a device int64 tensor)."""
        assert (
            actions.shape[0] == self.batch_size
        ), "Number of actions need to equal the number of generated graphs."
        from vrpgym_hip import check
        t = self._torch
        self._sync_positions()
        self._step_count += 1
        if isinstance(actions, t.Tensor):
            self._actions.copy_(actions.reshape(-1))
        else:
            # numpy indexing semantics of `self.visited[..., actions]` (tsp.py:86): negative
            # indices wrap, anything outside [-N, N) raises IndexError
            a = np.asarray(actions).reshape(-1).astype(np.int64)
            if a.size and (a.min() < -self.num_nodes or a.max() >= self.num_nodes):
                raise IndexError(f"index {int(a.max() if a.max() >= self.num_nodes else a.min())}"
                                 f" is out of bounds for axis 1 with size {self.num_nodes}")
            a = np.where(a < 0, a + self.num_nodes, a)
            actions = a.reshape(-1, 1)
            self._actions.copy_(t.from_numpy(np.ascontiguousarray(a)))
        if self.sampler._graphs is not None:  # rendering flags only (vrp_network.py:143-152)
            host = actions.cpu().numpy() if isinstance(actions, t.Tensor) else np.asarray(actions)
            edges = np.hstack([self.current_location, host.reshape(-1, 1)])
            self.sampler.visit_edges(edges.astype(int))
        self._notdone.zero_()
        e = self._cenv()
        check(self._lib.vrp_env_step(e, self._actions.data_ptr(), self._parity ^ 1,
                                     self._reward.data_ptr(), self._notdone.data_ptr(),
                                     self._stream()))
        self._parity ^= 1
        self._mask_fresh = True  # vrp_env_step applies generate_mask's fix-ups itself
        if self.video_save_path is not None:
            self.vid.capture_frame()
        done = int(self._notdone.item()) == 0
        return self.get_state(), self._reward.cpu().numpy(), done, None

    def snapshot_state(self):
        """Device-side episode state (what `step` mutates), for replay_tour."""
        return (self._visited.clone(), self._cur.clone(), self._load.clone(), self._mask.clone(),
                self._parity, self._mask_fresh, self._step_count)

    def replay_tour(self, start, actions, before=None):
        """Host-side part of T env.steps that ran fused on the device: the rendering flags of
        the traversed edges (vrp_network.py:143-152 via tsp.py:88-89) and one video frame per
        step (tsp.py:92-93).  start (B,1) = current_location before the first step, actions
        (T,B) the chosen nodes.  With a video recorder attached and `before` (the
        snapshot_state() taken before the rollout) the episode is re-stepped through `step`, so
        every frame is captured with that step's current_location / visited / load in place,
        like the reference's (the env step is bit-exact: the end state equals the fused one)."""
        actions = np.asarray(actions)
        if self.video_save_path is not None and before is not None:
            last, count = self._last_rollout, self._step_count
            vis, cur, load, mask, self._parity, self._mask_fresh, self._step_count = before
            self._last_rollout = None     # step_count reads t + 1 inside the t-th step's frame
            self._visited.copy_(vis); self._cur.copy_(cur); self._load.copy_(load)
            self._mask.copy_(mask)
            self._apply_mask()            # the model's initial env.get_state() (depot fix-up)
            for a in actions:
                self.step(a.reshape(-1, 1))   # visit_edges + capture_frame inside
            self._step_count, self._last_rollout = count, last   # counted once, by the rollout
            return
        cur = np.asarray(start).reshape(-1, 1)
        for a in actions:
            nxt = a.reshape(-1, 1)
            self.sampler.visit_edges(np.hstack([cur, nxt]).astype(int))
            if self.video_save_path is not None:
                self.vid.capture_frame()
            cur = nxt

    def is_done(self):
        """tsp.py:103-104."""
        return bool((self._visited == 1).all().item())

    # ------------------------------------------------------------------ E7
    def _apply_mask(self):
        if not self._mask_fresh:
            from vrpgym_hip import check
            self._sync_positions()
            check(self._lib.vrp_env_mask(self._cenv(), self._parity, self._stream()))
            self._mask_fresh = True

    def generate_mask(self):
        """tsp.py:131-148: applies the depot fix-ups to `visited` (device side) and
        returns the mask as a host float64 array like the reference."""
        self._apply_mask()
        return self._mask[self._parity].cpu().numpy().astype(np.float64)

    # ------------------------------------------------------------------ E3
    def get_state(self) -> np.ndarray:
        """tsp.py:106-129: (B,N,4) float64 = [x, y, is_depot, mask]."""
        mask = self.generate_mask()
        is_depot = np.zeros((self.batch_size, self.num_nodes))
        is_depot[np.arange(self.batch_size), self.depots[:, 0]] = 1
        return np.dstack([self.sampler.get_graph_positions(), is_depot, mask])

    # ------------------------------------------------------------------ host views
    @property
    def visited(self):
        return self._visited.cpu().numpy().astype(np.float64)

    @visited.setter
    def visited(self, value):
        self._visited.copy_(self._torch.from_numpy(np.asarray(value).astype(np.uint8)))
        self._mask_fresh = False

    @property
    def current_location(self):
        return self._cur.cpu().numpy().astype(np.int64).reshape(-1, 1)

    @current_location.setter
    def current_location(self, value):
        self._cur.copy_(self._torch.from_numpy(np.asarray(value).reshape(-1).astype(np.int32)))
        self._mask_fresh = False

    # ------------------------------------------------------------------ misc
    def __deepcopy__(self, memo):
        """agents copy the env for the baseline rollout (graph_tsp_agent.py:248)."""
        new = object.__new__(type(self))
        memo[id(self)] = new
        t = self._torch
        for k, v in self.__dict__.items():
            if isinstance(v, t.Tensor):
                new.__dict__[k] = v.clone()
            elif k in ("_lib", "_torch", "vid", "_last_rollout"):
                new.__dict__[k] = v
            elif k in ("_twin_env", "_graph_sightings", "_ws"):
                continue
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def twin(self):
        """A second env holding the same instances and state, for the baseline rollout
        (the reference deep-copies the env every step, graph_tsp_agent.py:248).  The twin
        object and its device tensors are reused across calls, so captured hipGraphs of
        the baseline rollout stay valid; only the contents are refreshed."""
        tw = self.__dict__.get("_twin_env")
        if tw is None:
            tw = copy.deepcopy(self)
            self.__dict__["_twin_env"] = tw
            return tw
        t = self._torch
        for k, v in self.__dict__.items():
            if k in ("_twin_env", "_graph_sightings", "_ws"):
                continue
            if isinstance(v, t.Tensor):
                tw.__dict__[k].copy_(v)
            elif k not in ("_lib", "_torch", "vid"):
                tw.__dict__[k] = v if k in ("sampler", "_depots_host", "_demands_host", "draw_idxs") \
                    else copy.copy(v)
        return tw

    def render(self, mode: str = "human"):
        return self.sampler.draw(self.draw_idxs)

    def enable_video_capturing(self, video_save_path: str):
        self.video_save_path = video_save_path
        if self.video_save_path is not None:
            try:
                from gym.wrappers.monitoring.video_recorder import VideoRecorder
            except Exception as exc:  # gym is an optional dependency
                raise RuntimeError("video capturing needs the `gym` package") from exc
            self.sampler.graphs  # materialise the per-graph views so edges get recorded
            self.vid = VideoRecorder(self, self.video_save_path)
            self.vid.frames_per_sec = 1

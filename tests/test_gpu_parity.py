"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the
golden vectors taken from the reference.  Run with `-m gpu` on an MI355X.

Bars (BASELINE.json north_star): visited/mask/step bookkeeping bit-exact; tour cost
and log-prob within 1e-5 (fp32) for the same seed; actions identical except where
the oracle's own top-2 logit gap is < 5e-5 (near-tie rule) —
such graphs are then checked teacher-forced.
"""
import glob
import os
from copy import deepcopy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
TOL = 1e-5          # north_star tolerance on cost / log-prob


def _margin(site, err, tol):
    """With VRPGYM_PARITY_MARGINS=<file> every eval-mode comparison appends `site err tol`: what
    profiles/r06_parity_margins.txt (the measured distance to each tolerance) was made from."""
    path = os.environ.get("VRPGYM_PARITY_MARGINS")
    if path:
        with open(path, "a") as fh:
            fh.write(f"{site} {err:.3e} {tol:.3e}\n")
TIE_GAP = 5e-5      # near-tie exemption threshold on the oracle's top-2 logit gap: twice the
                    # 2e-5 the logits may differ by; the largest slack seen over the 150-case
                    # GPU suite is 2.7e-7 (profiles/r03_parity_tie_statistics.csv, VRPGYM_PARITY_LOG)


def _load(pat):
    files = sorted(glob.glob(os.path.join(G, pat)))
    assert files, pat
    return [(os.path.basename(f)[:-4], f) for f in files]


def _envs():
    from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
    return {0: TSPEnv, 1: VRPEnv, 2: IRPEnv}


def _agents():
    import agents
    return {0: agents.TSPAgent, 1: agents.VRPAgent, 2: agents.IRPAgent}


def _arch_kw(z):
    """Agent kwargs of a fixture taken on a non-default architecture (tests/golden/arch*:
    hidden_dim not a multiple of the kernels' 128-wide slices, other layer counts)."""
    if "hidden" not in z.files:
        return {}
    kw = dict(hidden_dim=int(z["hidden"]), num_attention_layers=int(z["layers"]))
    if "heads" in z.files:
        kw["num_heads"] = int(z["heads"])
    return kw


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import vrpgym_hip
    vrpgym_hip.require_gpu()
    # exactly one HIP runtime mapped (torch's), see SURVEY 7.3 item 9
    libs = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
    assert len(libs) == 1, libs
    yield


# ------------------------------------------------------------------ E1-E8: environment
@pytest.mark.parametrize("name,path", _load("envtrace_*.npz"))
def test_env_trace_bit_exact(name, path):
    """Host-driven env.step through vrp_env_step: visited/mask/cur/load/done bit-exact
    against the reference trace, reward identical after the fp32 cast."""
    z = np.load(path)
    kind, B, N = int(z["kind"]), int(z["B"]), int(z["N"])
    env = _envs()[kind](N, B, 1, int(z["seed"]))
    assert np.array_equal(env.sampler.get_graph_positions(), z["pos"])
    assert np.array_equal(env.depots, z["depots"])
    assert np.array_equal(env.visited, np.zeros((B, N)))
    st = env.get_state()
    g = st[0] if kind == 2 else st
    assert g.shape == (B, N, 5 if kind == 2 else 4)
    assert np.array_equal(g[:, :, -1].astype(np.uint8), z["mask_init"])
    assert np.array_equal(env.visited.astype(np.uint8), z["visited_init"])
    done = False
    for t in range(int(z["T"])):
        st, r, done, info = env.step(z["actions"][t][:, None])
        g = st[0] if kind == 2 else st
        assert info is None and r.dtype == np.float64 and r.shape == (B,)
        assert np.array_equal(g[:, :, -1].astype(np.uint8), z["mask"][t]), t
        assert np.array_equal(env.visited.astype(np.uint8), z["visited"][t]), t
        assert np.array_equal(env.current_location[:, 0], z["cur"][t]), t
        assert done == bool(z["done"][t]), t
        assert np.max(np.abs(r - z["reward"][t])) <= 4.5e-16, t   # <= 1 ulp(fp64) near 1
        assert np.array_equal(r.astype(np.float32), z["reward"][t].astype(np.float32)), t
        if kind == 2:
            assert np.array_equal(st[1], z["load"][t]), t
            assert np.array_equal(g[:, :, 2], z["demands"])
        assert env.step_count == t + 1
    assert done and env.is_done() == bool(np.all(z["visited"][-1] == 1))


def test_env_reference_unit_tests():
    """The reference's tests/test_env.py:39-60 and tests/test_graph.py:26-42 run against
    the product env (coordinates injected through sampler.graphs, like the original)."""
    from gym_vrp.envs import VRPEnv
    from gym_vrp.graph.vrp_graph import VRPGraph
    env = VRPEnv(3, 2, 2, 69)
    y = np.sqrt(3) / 2
    for gi, coords in enumerate(([[0, 0], [1, 0], [0.5, y]], [[0, 0], [4, 0], [2, 4 * y]])):
        for n, c in enumerate(coords):
            env.sampler.graphs[gi].nodes[n]["coordinates"] = np.array(c, dtype=float)
    assert len(env.sampler.graphs) == 2 and len(env.sampler.graphs[0].nodes) == 3
    state = env.get_state()
    assert state.shape == (2, 3, 4) and np.sum(state[:, :, 2]) == 2
    state, reward, _, _ = env.step(np.array([2, 2])[:, None])
    assert np.allclose(reward, np.array([-1, 0]))
    assert state[0, 2, 3] == 1 and state[1, 2, 3] == 1
    np.random.seed(69)
    g = VRPGraph(2, 1)
    g.nodes[0]["coordinates"] = np.array([2, -1])
    g.nodes[1]["coordinates"] = np.array([-2, 2])
    assert g.euclid_distance(0, 1) == 5


def test_env_deepcopy_and_reset_keep_stream_order():
    from gym_vrp.envs import IRPEnv
    from oracle import envs as oenv
    # both draw from numpy's GLOBAL stream: run them one after the other
    o = oenv.OracleEnv(2, 9, 7, 3, 7)
    want = []
    for _ in range(2):
        o.reset()
        want.append((o.pos.copy(), o.demands.copy()))
    env = IRPEnv(9, 7, 3, 7)
    for r in range(2):
        env.reset()
        assert np.array_equal(env.sampler.get_graph_positions(), want[r][0])
        assert np.array_equal(env.demands, want[r][1])
    twin = deepcopy(env)
    a = np.array([int(np.flatnonzero(m == 0)[0]) for m in env.get_state()[0][:, :, -1]])
    env.step(a[:, None])
    assert not np.array_equal(env.visited, twin.visited)
    st, _, _, _ = twin.step(a[:, None])
    assert np.array_equal(env.visited, twin.visited) and np.array_equal(env.load, twin.load)
    with pytest.raises(AssertionError):
        env.step(np.zeros((3, 1), dtype=int))
    with pytest.raises(AssertionError):
        IRPEnv(5, 2, 3)


def test_random_agent_config1_pin():
    """BASELINE config 1: TSPEnv(20,64,seed=69) + RandomAgent(seed=69) -> T=19, mean
    cost 9.624367713928223 (measured on the reference); and the
    reference's own KAT tests/test_agent.py:57-69."""
    from agents import RandomAgent
    from gym_vrp.envs import TSPEnv, VRPEnv
    env = TSPEnv(20, 64, 6, 69)
    loss = RandomAgent(seed=69)(env)
    assert env.step_count == 19
    assert np.isclose(-loss.mean().item(), 9.624367713928223)
    np.random.seed(69)
    torch.manual_seed(69)
    env = VRPEnv(num_nodes=8, batch_size=2, num_draw=1)
    loss = RandomAgent()(env)
    assert np.isclose([loss.mean().item()], [-5.585874557495117])


# ------------------------------------------------------------------ N1-N3: encoder
@pytest.mark.parametrize("name,path", _load("encoder_*.npz"))
def test_encoder_against_reference(name, path):
    z = np.load(path)
    kind = int(z["kind"])
    for mode in ("eval", "train"):
        agent = _agents()[kind](seed=69)
        enc = agent.model.encoder
        enc.train(mode == "train")
        x = torch.tensor(z["x"])
        emb = enc(x) if kind == 0 else enc(x, torch.tensor(z["depot_mask"]))
        assert emb.is_cuda and emb.shape == z[f"emb_{mode}"].shape
        err = np.max(np.abs(emb.cpu().numpy() - z[f"emb_{mode}"]))
        assert err < TOL, (mode, err)
        if mode == "train":
            sd = enc.state_dict()
            for k in z.files:
                if k.startswith("bn_"):
                    d = (sd[k[3:]].float().cpu() - torch.tensor(z[k]).float()).abs().max().item()
                    assert d < TOL, (k, d)


def test_encoder_large_against_oracle():
    """B*N not a multiple of the 128-row GEMM tile, N > 64 attention path, and every row
    tiling of the fused eval-mode block kernel (16-row tiles x 2/3/4 up to 16384 rows, the
    64-row tile above); from 20480 rows on the persistent 80-row block kernel (ragged last
    tile) behind the fused in_proj+attention kernel (N = 37: two graphs per tile, N = 64: one,
    four key tiles) or behind the GEMM + attention pair (N = 50); N = 113 / 128: seven / eight key
    tiles of the attention kernel that reads q|k|v from global memory."""
    from oracle import policy as opol
    # (B >= 256 at 64 < N <= 105: encoder_qkv_attn_graph_kernel -- one graph per workgroup pass,
    # q|k|v in LDS -- with five, six and seven row tiles, ragged last tile, more graphs than CUs)
    for kind, B, N in [(0, 37, 20), (2, 5, 100), (1, 300, 40), (0, 350, 40), (1, 901, 20),
                       (0, 601, 37), (1, 450, 50), (2, 330, 64), (0, 3, 128), (1, 7, 113),
                       (1, 257, 65), (0, 300, 80), (2, 260, 81), (1, 520, 100), (0, 256, 105)]:
        agent = _agents()[kind](seed=69)
        sd, _ = opol.init_state_dicts(kind, 69)
        g = torch.Generator().manual_seed(B * N)
        x = torch.rand(B, N, 3, generator=g)
        dm = torch.zeros(B, N, dtype=torch.bool)
        dm[torch.arange(B), torch.randint(0, N, (B,), generator=g)] = True
        for train in (False, True):
            agent.model.encoder.train(train)
            emb = agent.model.encoder(x[:, :, :2] if kind != 2 else x, None if kind == 0 else dm)
            want = opol.encoder_forward(sd, x[:, :, :2] if kind != 2 else x,
                                        None if kind == 0 else dm, train=train)
            err = (emb.cpu() - want).abs().max().item()
            # eval mode: north_star's 1e-5 (measured over the 15 shapes: 1.4e-6 at most); batch-
            # statistics BatchNorm amplifies fp32 re-association noise: 2e-5 (measured 1.2e-5)
            tol = 2e-5 if train else TOL
            _margin("encoder_large_train" if train else "encoder_large_eval", err, tol)
            assert err < tol, (kind, B, N, train, err)


def test_gemm_mfma_against_torch():
    """fp32 MFMA GEMM vs a plain torch fp32 reference: ragged M, bias, residual, relu, and
    the small-problem kernel (<= 3M outputs, K = 128, no residual/relu: 16-row tiles)."""
    import vrpgym_hip as hip
    lib = hip.lib()
    g = torch.Generator().manual_seed(1)
    for M, N, K, relu, res, bias in [(1, 128, 128, 0, 1, 1), (200, 384, 128, 0, 1, 1),
                                     (777, 128, 512, 0, 1, 1), (1000, 512, 128, 1, 1, 1),
                                     (64, 1152, 128, 0, 1, 1),
                                     (1, 128, 128, 0, 0, 1), (17, 384, 128, 0, 0, 0),
                                     (512, 384, 128, 0, 0, 1), (2047, 1536, 128, 0, 0, 1),
                                     (8000, 384, 128, 0, 0, 1), (9000, 384, 128, 0, 0, 1)]:
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) * 0.1
        b = torch.randn(N, generator=g)
        R = torch.randn(M, N, generator=g)
        want = A.double() @ W.double().t()
        if bias:
            want = want + b.double()
        if res:
            want = want + R.double()
        if relu:
            want = want.clamp_min(0)
        Ad, Wd, bd, Rd = A.cuda(), W.cuda(), b.cuda(), R.cuda()
        Cd = torch.zeros(M, N, device="cuda")
        hip.check(lib.vrp_gemm_nt(Ad.data_ptr(), K, Wd.data_ptr(), K, bd.data_ptr() if bias else None,
                                  Rd.data_ptr() if res else None, N, Cd.data_ptr(), N, M, N, K, relu,
                                  hip.current_stream()))
        err = (Cd.cpu().double() - want).abs().max().item()
        assert err < 5e-5 * max(1.0, want.abs().max().item()), (M, N, K, res, err)


# ------------------------------------------------------------------ D1-D6: decoder
@pytest.mark.parametrize("step_flags", [0, 16, 4], ids=["table", "table_wide", "tile"])
@pytest.mark.parametrize("name,path", _load("decoder_*.npz"))
def test_decoder_teacher_forced(name, path, step_flags):
    """GraphDecoder.forward (decode-only kernel) on the reference's inputs: same
    action, log-prob within 1e-5, for greedy and (host-noise) sampled steps, with B
    not a multiple of 8 and B < 8 (the scrambled-mask indexing).  Every step kernel:
    decode_step_rt_kernel<NPL,1>, <NPL,4> and the raw-tile kernel; the N = 100 file runs
    their two-nodes-per-lane instances."""
    z = np.load(path)
    kind = int(z["kind"])
    agent = _agents()[kind](seed=69)
    dec = agent.model.decoder
    dec.step_flags = step_flags
    dec.reset()
    emb = torch.tensor(z["emb"])
    for t in range(z["mask"].shape[0]):
        greedy = t % 2 == 0
        torch.manual_seed(900 + t)  # the reference drew its noise from this state
        idx, logp = dec(emb, mask=torch.tensor(z["mask"][t]),
                        load=torch.tensor(z["load"][t]) if kind == 2 else None, rollout=greedy)
        assert idx.shape == (emb.shape[0], 1) and idx.dtype == torch.int64
        assert np.array_equal(idx[:, 0].cpu().numpy(), z["idx"][t]), (t, greedy)
        assert np.max(np.abs(logp.reshape(-1).cpu().numpy() - z["logp"][t])) < TOL, t
    dec.reset()
    assert dec.first_step


# ------------------------------------------------------------------ R1: rollouts
def _log_roots(*row):
    """Survey aid: with VRPGYM_PARITY_LOG=<file> every comparison appends its tie statistics
    (the bound asserted in _compare_rollout rests on these)."""
    path = os.environ.get("VRPGYM_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(",".join(str(v) for v in row) + "\n")


def _log_train(*row):
    """With VRPGYM_TRAIN_PARITY_LOG=<file> every train-mode comparison appends the HIP path's
    error against the fp64 evaluation, its bound and the fp32 oracle's error on the same inputs
    (profiles/r04_train_mode_parity.csv is such a log)."""
    path = os.environ.get("VRPGYM_TRAIN_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(",".join(f"{v:.3e}" if isinstance(v, float) else str(v) for v in row) + "\n")


def _train_mode_statistics():
    """tests/golden/train_mode_error.json (tools/make_golden.py trainrollouts): how far the
    REFERENCE's own fp32 train-mode rollouts sit from an fp64 evaluation of the same model on
    the same action path, and the same figure for the fp32 oracle -- for the seven committed
    trainrollout_* shapes ("cases") and over a random sweep of shapes / seeds ("sweep")."""
    import json
    with open(os.path.join(G, "train_mode_error.json")) as f:
        return json.load(f)


_TRAIN_KEYS = ("du_max", "dlogp_step_max", "dlogp_acc_max")
_TRAIN_FLOORS = (1e-5, 5e-6, 5e-6)   # half the eval-mode tolerances (2e-5 logits, 1e-5 log-prob)


def _reference_to_oracle_ratio(quantile=None):
    """Ratio (reference fp32 error) / (oracle fp32 error), both against fp64, over every measured
    case whose errors are above the floors (below them the ratio is rounding noise and the floors
    decide): the largest one, or the given quantile of the sample (the maxima are heavy-tailed:
    du_max 4.15 once and <= 1.65 otherwise, p95 1.43; dlogp_step_max max 1.55, p95 1.50;
    dlogp_acc_max max 2.55, p95 1.46)."""
    st = _train_mode_statistics()
    out = []
    for k, fl in zip(_TRAIN_KEYS, _TRAIN_FLOORS):
        rs = [c["reference_fp32_vs_fp64"][k] / c["oracle_fp32_vs_fp64"][k]
              for c in st["cases"] + st["sweep"]
              if c["oracle_fp32_vs_fp64"][k] > fl / 2 and c["reference_fp32_vs_fp64"][k] > fl / 2]
        out.append(max(rs) if quantile is None else float(np.percentile(rs, quantile)))
    return out


def _train_bounds(kind, B, N, o32):
    """Bounds for the HIP path's train-mode error AGAINST THE FP64 EVALUATION (per-step logits,
    per-step log-prob, accumulated log-prob) = 2 x the reference's own fp32 error.  Batch-
    statistics BatchNorm amplifies fp32 re-association noise with the batch (B*N rows) and the
    episode length, so north_star's flat 1e-5 is not what the reference itself achieves: at VRP
    33 x 100 its logits sit 4.2e-5 and its per-step log-probs 2.2e-5 from the fp64 values, and
    two legitimate fp32 evaluations (reference, oracle) 1.06e-4 from each other.

    The reference's error on THIS test's inputs is estimated from the fp32 oracle's (`o32`,
    measured by the caller against the same fp64 trace) times a reference/oracle ratio from the
    committed measurements:
      * a shape with a committed measurement of its own (the seven trainrollout_* fixtures): the
        95th percentile of the ratio over the sweep (1.43 / 1.50 / 1.46), and at least the
        reference's measured error on exactly that shape -- round 5; until then the sweep's single
        4.15x outlier widened every bound;
      * any other shape (tools/parity_sweep.py: random shapes and seeds, where a 1-in-20 tail event
        of the ratio would be a 1-in-20 spurious failure): the largest ratio observed."""
    own = [c for c in _train_mode_statistics()["cases"] if (c["kind"], c["B"], c["N"]) == (kind, B, N)]
    ratio = _reference_to_oracle_ratio(95 if own else None)
    ref = [r * o for r, o in zip(ratio, o32)]
    for c in own:
        ref = [max(a, c["reference_fp32_vs_fp64"][k]) for a, k in zip(ref, _TRAIN_KEYS)]
    return tuple(2.0 * max(r, f) for r, f in zip(ref, _TRAIN_FLOORS))


def _train_mean_bound(o32_mean):
    """The MEAN logit error is a stable statistic (the maxima above are heavy-tailed: over the
    sweep the reference's own maximum is 4.15x the oracle's once, <= 1.65x otherwise): a
    systematic loss of accuracy shows here first.  Bound: 2 x the reference's mean error,
    estimated from the fp32 oracle's mean on these inputs times the largest reference/oracle
    ratio of means over the measured cases (1.30; median 1.01).  Measured over 426 random
    train-mode cases (profiles/r04_train_mode_parity.csv): HIP mean / oracle mean has median
    1.03, p90 1.17, maximum 2.1 (two graphs of 17 nodes)."""
    st = _train_mode_statistics()
    r = max(c["reference_fp32_vs_fp64"]["du_mean"] / c["oracle_fp32_vs_fp64"]["du_mean"]
            for c in st["cases"] + st["sweep"] if c["oracle_fp32_vs_fp64"]["du_mean"] > 2e-7)
    return 2.0 * r * max(o32_mean, 2e-7)


def _compare_rollout(kind, B, N, greedy, env_seed, agent_seed, torch_seed, ref_actions=None,
                     ref_loss=None, ref_logp=None, ref_T=None, train=False, tile_kernel=False,
                     throughput_kernel=False, table_kernel=False, agent=None, fused=False):
    """HIP rollout vs oracle (and vs reference outputs when given).  `agent`: use this
    (e.g. trained) agent's weights on both sides instead of the seed's initial ones.

    Actions are compared for greedy AND sampled rollouts.  Along the HIP action path (the
    oracle teacher-forced on it, same host noise) every HIP choice must be the oracle's
    choice up to a near tie: greedy = top-2 logit gap < TIE_GAP; sampled = the chosen node's
    softmax(u)/q within a relative TIE_GAP of the maximum (Categorical.sample is
    argmax(p/q), graph_decoder.py:104-107; p ~ exp(u), so a relative gap in p/q is an
    absolute gap in u).  The number of graphs that took a near-tie runner-up is bounded, and
    without one the free-running action sequences must be identical.

    fused=True: no per-step logits trace (which forces one launch per step), so the episode
    kernel that runs steps 1..T-1 in ONE launch is the one compared (decode_persistent_kernel,
    N <= 63): actions, per-step log-probs, accumulators and T are checked, the logits only
    through them."""
    from oracle import envs as oenv
    from oracle import policy as opol
    from agents import runtime
    if agent is None:
        agent = _agents()[kind](seed=agent_seed)
        sd, _ = opol.init_state_dicts(kind, agent_seed)
    else:
        sd = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
    model = agent.model
    model.train(train)
    env = _envs()[kind](N, B, 1, env_seed)
    oe = oenv.OracleEnv(kind, N, B, 1, env_seed)
    heads = agent.model.encoder._dims[3]   # encoder heads (8 unless the fixture says otherwise)
    trace = []
    torch.manual_seed(torch_seed)
    with torch.no_grad():
        ol, olp, oT = opol.rollout(sd, deepcopy(oe), greedy, train=train, trace=trace, heads=heads)
    oacts = np.array([t["idx"].numpy() for t in trace])
    torch.manual_seed(torch_seed)
    with torch.no_grad():
        res = runtime.rollout(model, deepcopy(env), greedy, train=train, trace=not fused,
                              step_trace=fused, noise_mode="host", tile_kernel=tile_kernel,
                              throughput_kernel=throughput_kernel, table_kernel=table_kernel)
    T = res.T
    acts = res.actions[:T].cpu().numpy()

    def diverged(want):
        out = {}
        for b in range(B):
            for t in range(min(T, len(want))):
                if acts[t, b] != want[t, b]:
                    out[b] = t
                    break
        return out

    div_oracle = diverged(oacts)
    forced_trace = trace
    if div_oracle or T != oT:
        # the oracle follows the HIP actions; every divergence must sit on a near tie of the
        # oracle's own logits (greedy) / ratios p/q (sampled) at that step
        forced_trace = []
        torch.manual_seed(torch_seed)   # same noise stream: one (B,N) draw per step
        with torch.no_grad():
            ol, olp, oT2 = opol.rollout(sd, deepcopy(oe), greedy, train=train,
                                        trace=forced_trace, forced=acts, heads=heads)
        assert oT2 == T
    div_ref = diverged(ref_actions) if ref_actions is not None else {}
    # Along the HIP action path the oracle must agree with every HIP choice up to a near
    # tie.  (A graph may also leave the free-running oracle's path WITHOUT a tie of its own:
    # the scrambled glimpse mask couples it to graphs that flipped earlier.)
    U = torch.stack([st["u"] for st in forced_trace])                   # (T,B,N)
    A = torch.as_tensor(acts)[:, :, None]
    du_bound = dl_bound = acc_bound = None
    if train:
        # Train mode is judged against the fp64 evaluation of the same model on the same
        # action path, with bounds tied to the reference's own measured fp32 error
        # (_train_bounds).  The fp32 oracle's error on these very inputs is measured here too.
        trace64 = []
        torch.manual_seed(torch_seed)
        with torch.no_grad():
            _, olp64, oT64 = opol.rollout(opol.as_double(sd), deepcopy(oe), greedy, train=True,
                                          trace=trace64, forced=acts, heads=heads)
        assert oT64 == T
        U64 = torch.stack([st["u"] for st in trace64])
        LP64 = torch.stack([st["logp"] for st in trace64])
        fin64 = torch.isfinite(U64)
        assert torch.equal(fin64, torch.isfinite(U))
        o32_mean = (U.double()[fin64] - U64[fin64]).abs().mean().item()
        o32 = ((U.double()[fin64] - U64[fin64]).abs().max().item(),
               (torch.stack([st["logp"] for st in forced_trace]).double() - LP64).abs().max().item(),
               (olp.double() - olp64).abs().max().item())
        du_bound, dl_bound, acc_bound = _train_bounds(kind, B, N, o32)
        U = U64   # ties are judged on the exact logits
    if greedy:
        slack = U.max(dim=2).values - U.gather(2, A)[..., 0]
    else:
        Q = torch.stack([st["noise"] for st in forced_trace])
        ratio = torch.softmax(U - U.logsumexp(-1, keepdim=True), dim=-1) / Q
        best = ratio.max(dim=2).values
        slack = (best - ratio.gather(2, A)[..., 0]) / best              # relative
    # eval mode: TIE_GAP on the fp32 oracle's logits.  Train mode: the slack is measured on the
    # fp64 logits and every HIP logit may sit du_bound away from them, so a choice can trail the
    # exact maximum by twice that.
    gap = 2 * du_bound if train else TIE_GAP
    assert slack.max().item() < gap, (slack.max().item(), gap)
    # graphs with a tie flip: the HIP choice is not the oracle's own pick on the same state (an
    # EXACT tie of the oracle's logits counts -- slack 0, the oracle takes the lowest index, our
    # logits differ in the last bits: seen once in 640 sweep cases, VRP 100 x 63 step 112)
    pick = (U if greedy else ratio).argmax(dim=2)
    roots = int((pick != A[..., 0]).any(dim=0).sum())
    _log_roots(kind, B, N, greedy, train, tile_kernel, throughput_kernel, table_kernel, roots,
               slack.max().item(), len(div_oracle), len(div_ref))
    # seen: 2 of 200 graphs in one shape (IRP 200 x 40: tanh-saturated logits), 0 elsewhere
    assert roots <= max(2, B // 100), f"{roots} of {B} graphs chose a near-tie runner-up"
    if not roots:
        def _where():
            b, t = min(div_oracle.items(), key=lambda kv: kv[1])
            top = torch.topk(forced_trace[t]["u"][b], 2)
            return (f"graph {b} step {t}: HIP chose {int(acts[t, b])}, oracle {int(oacts[t, b])}, "
                    f"slack {slack[t, b].item():.3e}, oracle top-2 logits {top.values.tolist()} at {top.indices.tolist()}")
        if not train:
            assert not div_oracle, "diverged from the oracle without any near tie: " + _where()
            assert T == oT
        elif div_oracle:
            # HIP followed the fp64 pick at every step, so it is the free-running fp32 ORACLE
            # that left the exact path: its choice at the earliest divergence must be a near tie
            # of the exact logits / ratios
            t0 = min(div_oracle.values())
            score = U[t0] if greedy else ratio[t0]
            for b in [b for b, t in div_oracle.items() if t == t0]:
                top, theirs = score[b].max().item(), score[b, int(oacts[t0, b])].item()
                miss = top - theirs if greedy else (top - theirs) / top
                assert miss < gap, (b, t0, miss, _where())
    if ref_T is not None and not div_ref:
        assert T == ref_T
    if div_ref:
        # The reference's own actions (torch's fused CPU attention) may differ from ours only
        # through a near tie as well: up to the EARLIEST divergence both sides are in the same
        # state, so there the reference's choice must be within the gap of the best node on our
        # logits / ratios; later divergences can be coupled to it through the scrambled mask.
        t0 = min(div_ref.values())
        first = [b for b, t in div_ref.items() if t == t0]
        assert len(first) <= max(2, B // 100), (t0, first)
        score = U[t0] if greedy else ratio[t0]
        for b in first:
            top, theirs = score[b].max().item(), score[b, int(ref_actions[t0, b])].item()
            miss = top - theirs if greedy else (top - theirs) / top
            assert miss < gap, (b, t0, miss)
    exempt = np.zeros(B, bool)
    exempt[list(div_ref)] = True
    loss, logp = res.acc_loss.cpu(), res.acc_logp.cpu()
    _margin("rollout_cost_train" if train else "rollout_cost_eval", (loss - ol).abs().max().item(), TOL)
    assert (loss - ol).abs().max().item() < TOL, (loss - ol).abs().max().item()
    acc_tol = TOL * (1 if greedy else max(1, T / 4))
    if train:
        # against the fp64 sum, within twice the reference's own accumulated error -- and never
        # below what an fp32 accumulator of that magnitude can resolve: the reference sums the
        # step log-probs in fp32 (graph_tsp_agent.py:86), and next to a sum A one ulp is 2^-23 |A|
        # (7.6e-6 at |A| in [64, 128): two legitimate fp32 sums of a 100-step episode differ by
        # more than the 5e-6 floor of _TRAIN_FLOORS.  Round 5's random sweep, seed 12: VRP 1 x 32,
        # |A| = 137: 1.6e-5 against the floor's 1.0e-5 -- the same number with every kernel of the
        # round switched off, profiles/r05_parity_sweep_train.log)
        acc_bound = max(acc_bound, 2.0 * 2.0 ** -23 * olp64.abs().max().item())
        acc_err = (logp.double() - olp64).abs().max().item()
        assert acc_err < (TOL if greedy else acc_bound), (acc_err, acc_bound)
        # two fp32 evaluations may sit on opposite sides of the exact value
        acc_tol = max(acc_tol, 0.0 if greedy else acc_bound + o32[2])
    if not greedy:
        _margin("rollout_acc_logp_train" if train else "rollout_acc_logp_eval", (logp - olp).abs().max().item(), acc_tol)
    assert (logp - olp).abs().max().item() < acc_tol, (logp - olp).abs().max().item()
    if ref_loss is not None:
        ok = ~exempt
        assert np.max(np.abs(loss.numpy() - ref_loss)[ok]) < TOL
        assert np.max(np.abs(logp.numpy() - ref_logp)[ok]) < acc_tol
    # per-step logits along the same action path.  Logits live in [-10, 10].  Eval mode:
    # within 2e-5 of the fp32 oracle.  Train mode: within du_bound of the FP64 logits (twice
    # what the reference's own fp32 logits are off by, _train_bounds).
    worst_du, sum_du, n_du = 0.0, 0.0, 0
    for t in range(0 if fused else T):
        u = res.logits[t].cpu()
        ou = trace64[t]["u"] if train else forced_trace[t]["u"]
        fin = torch.isfinite(ou)
        assert torch.equal(fin, torch.isfinite(u)), t
        d = (u.to(ou.dtype)[fin] - ou[fin]).abs()
        err = d.max().item()
        worst_du = max(worst_du, err)
        sum_du += d.sum().item()
        n_du += d.numel()
        if not train:
            _margin("step_logits_eval", err, 2e-5)
        assert err < (du_bound if train else 2e-5), (t, err, du_bound)
    if train and n_du:
        mean_bound = _train_mean_bound(o32_mean)
        assert sum_du / n_du < mean_bound, (sum_du / n_du, mean_bound, o32_mean)
    worst_dl = 0.0
    if not greedy:
        # north_star: log-prob within 1e-5 -- held PER STEP (log p(a_t) of every sampled
        # action) against the fp32 oracle in eval mode; the accumulated sum of T such terms is
        # checked above at 1e-5 x max(1, T/4).  Train mode: per step within dl_bound of the fp64
        # log-prob.
        slp = res.step_logp[:T].cpu()
        if train:
            worst_dl = (slp.double() - LP64).abs().max().item()
            assert worst_dl < dl_bound, (worst_dl, dl_bound)
        else:
            olp_t = torch.stack([st["logp"] for st in forced_trace])
            worst_dl = (slp - olp_t).abs().max().item()
            assert worst_dl < TOL, worst_dl
    if train:
        _log_train(kind, B, N, greedy, T, tile_kernel, throughput_kernel, table_kernel, fused,
                   worst_du, du_bound, o32[0], worst_dl, dl_bound, o32[1],
                   (logp.double() - olp64).abs().max().item(), acc_bound, o32[2],
                   sum_du / max(n_du, 1), o32_mean)
    return res, exempt


@pytest.mark.parametrize("mode", ["default", "table", "table_wide", "tile"])
@pytest.mark.parametrize("name,path", _load("rollout_*.npz"))
def test_rollout_against_reference(name, path, mode):
    """Every step kernel against the reference's recorded rollouts, greedy and sampled: the
    default dispatch (persistent kernel at N <= 63; at 64 < N <= 104 each graph goes to the
    raw-tile or the table kernel by its number of selectable nodes), the table-driven kernel
    for every graph in its latency mode (`table`: decode_step_rt_kernel<NPL,1>) and in its
    large-batch mode (`table_wide`: <NPL,4>), and the raw-tile formulation for every graph."""
    z = np.load(path)
    _compare_rollout(int(z["kind"]), int(z["B"]), int(z["N"]), bool(z["greedy"]), 69, 69,
                     int(z["torch_seed"]), ref_actions=z["actions"], ref_loss=z["acc_loss"],
                     ref_logp=z["acc_logp"], ref_T=int(z["T"]), tile_kernel=mode == "tile",
                     throughput_kernel=mode == "table_wide",
                     table_kernel=mode in ("table", "table_wide"))


@pytest.mark.parametrize("mode", ["default", "table", "fused"])
@pytest.mark.parametrize("name,path", _load("archrollout_*.npz"))
def test_non_default_architecture_against_reference(name, path, mode):
    """The reference's agent kwargs `hidden_dim` / `num_attention_layers` / `num_heads`
    (graph_tsp_agent.py:96-106, graph_encoder.py:6-39) at values the kernels' 128-wide
    feed-forward slices do not divide (200, 64, 520, 300), at 1, 2 and 4 layers, and with four
    and sixteen encoder heads (GEMM + per-head attention kernels instead of the fused ones): the
    constructor yields the reference's weights, and the reference's recorded episode is
    reproduced through the stack kernel / per-layer kernels and every step kernel."""
    import hashlib
    z = np.load(path)
    kind = int(z["kind"])
    agent = _agents()[kind](seed=69, **_arch_kw(z))
    h = hashlib.sha256()
    for k, v in agent.model.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().numpy().tobytes())
    assert h.hexdigest()[:16] == str(z["sd_hash"])
    _compare_rollout(kind, int(z["B"]), int(z["N"]), bool(z["greedy"]), 69, 69,
                     int(z["torch_seed"]), ref_actions=z["actions"], ref_loss=z["acc_loss"],
                     ref_logp=z["acc_logp"], ref_T=int(z["T"]), table_kernel=mode == "table",
                     fused=mode == "fused", agent=agent)


@pytest.mark.parametrize("hidden,layers,B,N,heads", [(200, 2, 600, 40, 8), (72, 5, 300, 80, 8),
                                                     (256, 10, 600, 40, 8), (128, 9, 100, 20, 8),
                                                     (520, 1, 64, 20, 8), (512, 3, 600, 40, 4),
                                                     (256, 2, 300, 100, 16), (200, 2, 40, 128, 4)])
def test_non_default_architecture_encoder_large(hidden, layers, B, N, heads):
    """... and through the large-batch encoder kernels (persistent 80-row block kernel behind
    the fused in_proj + attention kernels, eval mode; GEMM + BatchNorm kernels, train mode)
    against the oracle."""
    from oracle import policy as opol
    import agents
    agent = agents.VRPAgent(seed=69, hidden_dim=hidden, num_attention_layers=layers, num_heads=heads)
    sd = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
    g = torch.Generator().manual_seed(B * N)
    x = torch.rand(B, N, 2, generator=g)
    dm = torch.zeros(B, N, dtype=torch.bool)
    dm[torch.arange(B), torch.randint(0, N, (B,), generator=g)] = True
    for train in (False, True):
        agent.model.encoder.train(train)
        emb = agent.model.encoder(x, dm)
        want = opol.encoder_forward(sd, x, dm, train=train, heads=heads)
        err = (emb.cpu() - want).abs().max().item()
        # two fp32 evaluations drift apart layer by layer (every layer renormalises and adds its own
        # rounding): 2e-5 up to five layers, proportionally more beyond (ten layers, train mode:
        # 2.1e-5 with the fp32-MFMA kernels and with the bf16-plane ones alike)
        # (eval mode: 1e-5 whatever the depth -- measured 3.3e-6 at most, profiles/r06_parity_margins.txt)
        tol = 2e-5 * max(1.0, layers / 5) if train else TOL
        _margin("encoder_arch_train" if train else "encoder_arch_eval", err, tol)
        assert err < tol, (hidden, layers, B, N, heads, train, err)


@pytest.mark.parametrize("mode", ["default", "table", "tile"])
@pytest.mark.parametrize("name,path", _load("trainrollout_*.npz"))
def test_train_mode_rollout_against_reference(name, path, mode):
    """The reference's own train-mode (batch-statistics BatchNorm) sampled rollouts: same
    actions (near-tie rule on the fp64 logits), cost within 1e-5, per-step logits / log-probs
    within twice the reference's measured distance from the fp64 evaluation, accumulated
    log-prob likewise, and directly against the reference's recorded sums."""
    z = np.load(path)
    N = int(z["N"])
    if mode == "tile" and N > 104:
        pytest.skip("raw-tile kernel: N <= 104")
    _compare_rollout(int(z["kind"]), int(z["B"]), N, False, 69, 69, int(z["torch_seed"]),
                     ref_actions=z["actions"], ref_loss=z["acc_loss"], ref_logp=z["acc_logp"],
                     ref_T=int(z["T"]), train=True, tile_kernel=mode == "tile",
                     table_kernel=mode == "table")


@pytest.mark.parametrize("kind,B,N,greedy,train", [
    (0, 512, 20, True, False),    # BASELINE config 2
    (1, 200, 40, True, False),    # config 3 shape (smaller batch), VRP quirks
    (2, 200, 40, True, False),    # config 4 shape
    (1, 24, 100, False, False),   # config 5 shape: N=100 sampling (two nodes per lane)
    (2, 31, 33, False, True),     # train-mode BN + sampling, odd sizes
    (0, 3, 5, True, True),        # B < 8: scrambled mask wraps around
    (1, 130, 64, True, False),    # N = 64 boundary
    (0, 77, 65, False, False),    # N = 65 boundary
])
def test_rollout_against_oracle(kind, B, N, greedy, train):
    _compare_rollout(kind, B, N, greedy, 11, 69, 5, train=train)
    _compare_rollout(kind, B, N, greedy, 11, 69, 5, train=train, table_kernel=True)
    _compare_rollout(kind, B, N, greedy, 11, 69, 5, train=train, throughput_kernel=True,
                     table_kernel=True)
    if N <= 104:  # the raw-tile kernel (opt-in flag) stays covered at every size it supports
        _compare_rollout(kind, B, N, greedy, 11, 69, 5, train=train, tile_kernel=True)


@pytest.mark.parametrize("kind,B,N,greedy,train", [
    (1, 200, 40, False, False), (2, 64, 21, False, False), (0, 512, 20, True, False),
    (1, 130, 63, True, False), (2, 31, 33, False, True),
])
def test_fused_episode_kernel_against_oracle(kind, B, N, greedy, train):
    """Steps 1..T-1 in one launch (decode_persistent_kernel, N <= 63) against the oracle."""
    import vrpgym_hip as hip
    name = hip.lib().vrp_step_kernel_name(kind, B, N, 0 if greedy else 1).decode()
    assert name == "decode_persistent_kernel", name
    _compare_rollout(kind, B, N, greedy, 11, 69, 5, train=train, fused=True)


@pytest.mark.parametrize("name,path", _load("rollout_*_N10_sample.npz") + _load("rollout_*_N20_greedy.npz"))
def test_fused_episode_kernel_against_reference(name, path):
    """The reference's recorded episodes through the one-launch episode kernel."""
    z = np.load(path)
    _compare_rollout(int(z["kind"]), int(z["B"]), int(z["N"]), bool(z["greedy"]), 69, 69,
                     int(z["torch_seed"]), ref_actions=z["actions"], ref_loss=z["acc_loss"],
                     ref_logp=z["acc_logp"], ref_T=int(z["T"]), fused=True)


@pytest.mark.parametrize("kind,B,N,greedy", [
    (0, 1, 2, True),      # smallest legal instance: one graph, depot + one customer
    (1, 1, 3, False),     # B = 1: every scrambled-mask row is the graph's own
    (2, 6, 9, False),     # B not a multiple of the 4 graphs per workgroup, sampling
    (1, 5, 128, True),    # VRP_MAX_NODES: two nodes per lane, up to 254 steps
    (0, 2051, 11, True),  # B just above the single-wave-workgroup threshold
])
def test_rollout_edge_shapes(kind, B, N, greedy):
    _compare_rollout(kind, B, N, greedy, 3, 69, 9)


def test_api_error_behaviour():
    """Errors mirror the reference's asserts; library misuse raises with vrp_last_error()."""
    import agents
    import vrpgym_hip as hip
    from gym_vrp.envs import IRPEnv, TSPEnv
    env = TSPEnv(5, 4, 1, 1)
    with pytest.raises(AssertionError, match="Number of actions"):
        env.step(np.zeros((3, 1), dtype=int))
    with pytest.raises(AssertionError, match="Num_draw"):
        TSPEnv(5, 2, 3)
    with pytest.raises(TypeError):
        agents.TSPAgent(seed=1).model(IRPEnv(5, 4, 1, 1), True)   # wrong env kind
    lib = hip.lib()
    rc = lib.vrp_gemm_nt(None, 128, None, 128, None, None, 0, None, 100, 4, 100, 128, 0, None)
    assert rc != 0 and b"multiple" in lib.vrp_last_error()
    with pytest.raises(RuntimeError, match="libvrpgym_hip error"):
        hip.check(rc)


def test_reference_agent_kats():
    """tests/test_agent.py:72-114 of the reference, unchanged but for the import root:
    greedy agent.step on B=2, N=4 must give the reference's published values."""
    import agents
    from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
    torch.manual_seed(69)
    np.random.seed(69)
    for Env, Agent, want in [(TSPEnv, agents.TSPAgent, -1.5130789279937744),
                             (VRPEnv, agents.VRPAgent, -1.952601671218872),
                             (IRPEnv, agents.IRPAgent, -2.9770922660827637)]:
        env = Env(num_nodes=4, batch_size=2, num_draw=1)
        agent = Agent()
        loss, loss_b, _ = agent.step(env, [True, True])
        assert np.isclose([loss.mean().item()], [want]), (Env.__name__, loss.mean().item())
    enc = agents.GraphEncoder(node_input_dim=2).to("cuda")
    env = VRPEnv(num_nodes=8, batch_size=2, num_draw=1)
    emb = enc(torch.from_numpy(env.reset()).float()[:, :, :2])
    assert emb.shape == (2, 8, 128)


def test_full_size_properties():
    """North-star shape 8192 x 40 (too big for the oracle to replay quickly): size-
    independent invariants of a greedy rollout."""
    import agents
    from gym_vrp.envs import TSPEnv, VRPEnv
    from agents import runtime
    for Env, Agent, kind in [(TSPEnv, agents.TSPAgent, 0), (VRPEnv, agents.VRPAgent, 1)]:
        B, N = 8192, 40
        env = Env(N, B, 1, 3)
        agent = Agent(seed=69)
        agent.model.eval()
        pos = env.sampler.get_graph_positions()
        dep = env.depots[:, 0]
        with torch.no_grad():
            res = runtime.rollout(agent.model, env, True, trace=True)
        T = res.T
        acts = res.actions[:T].cpu().numpy()
        assert T == N - 1 if kind == 0 else N <= T <= 2 * (N - 1)
        # every customer visited exactly once; depot never chosen by TSP
        rows = np.arange(B)
        cost = np.zeros(B)
        cur = dep.copy()
        seen = np.zeros((B, N), int)
        for t in range(T):
            a = acts[t]
            cost += np.linalg.norm(pos[rows, cur] - pos[rows, a], axis=1)
            seen[rows, a] += 1
            cur = a
        cust = np.ones((B, N), bool)
        cust[rows, dep] = False
        assert np.all(seen[cust] == 1)
        if kind == 0:
            assert np.all(seen[~cust] == 0)
        assert np.max(np.abs(-res.acc_loss.cpu().numpy() - cost)) < 2e-5 * T
        assert torch.all(res.acc_logp == 0)
        # idempotence: the same rollout twice gives identical actions (no atomics races)
        env2 = Env(N, B, 1, 3)
        with torch.no_grad():
            res2 = runtime.rollout(agent.model, env2, True, trace=True)
        assert res2.T == T and torch.equal(res2.actions[:T], res.actions[:T])
        assert torch.equal(res2.acc_loss, res.acc_loss)


def test_graph_replay_equals_eager():
    """The hipGraph-captured rollout (default path) gives bit-identical results to the
    eager launch sequence, across resets, for greedy and train-mode rollouts, and the
    agent's twin env keeps model/baseline on identical instances."""
    import agents
    from gym_vrp.envs import IRPEnv
    from agents import runtime
    env = IRPEnv(12, 33, 1, 5)
    agent = agents.IRPAgent(seed=69)
    agent.model.eval()
    for rep in range(4):  # second sighting captures, later ones replay
        env.reset()
        ref_env = deepcopy(env)
        with torch.no_grad():
            eager = runtime.rollout(agent.model, ref_env, True, use_graph=False)
            got = runtime.rollout(agent.model, env, True, use_graph=True)
        assert torch.equal(eager.acc_loss, got.acc_loss), rep
        assert eager.T == got.T
        assert np.array_equal(env.visited, ref_env.visited)
    loss, loss_b, _ = agent.step(env, [True, True])
    assert torch.equal(loss, loss_b)  # same weights, same instances
    # many replays, with and without a sync in between (a captured hipMemsetAsync node
    # used to race with the first step kernels: the accumulators are now zeroed by a kernel)
    from gym_vrp.envs import TSPEnv
    env2 = TSPEnv(20, 512, 1, 69)
    ag2 = agents.TSPAgent(seed=69)
    ag2.model.eval()
    with torch.no_grad():
        want = runtime.rollout(ag2.model, deepcopy(env2), True, use_graph=False).acc_loss
        for rep in range(40):
            env2._visited.zero_(); env2._cur.copy_(env2._depot); env2._mask_fresh = False
            got = runtime.rollout(ag2.model, env2, True, use_graph=True)
            if rep % 2:
                torch.cuda.synchronize()
            assert torch.equal(got.acc_loss, want), rep
    # train-mode replays keep updating the BN running statistics once per rollout
    agent.model.train()
    nb0 = int(agent.model.encoder.attention_layers[0].bn1.norm.num_batches_tracked.item())
    for rep in range(3):
        env.reset()
        with torch.no_grad():
            runtime.rollout(agent.model, env, True, train=True, use_graph=True)
    nb1 = int(agent.model.encoder.attention_layers[0].bn1.norm.num_batches_tracked.item())
    assert nb1 - nb0 == 3


@pytest.mark.parametrize("hidden", [512, 200])
def test_graph_replay_after_sync_weights(hidden):
    """A captured rollout keeps reading the encoder's device-side weight shadows (the bf16
    planes of vrp_encoder_prepare; the zero-padded feed-forward copies at hidden = 200) at the
    addresses baked into its kernel arguments.  `agent.sync_weights()` -> `runtime.invalidate`
    must therefore refresh those buffers IN PLACE: a rebuilt struct with a fresh allocation
    would leave the graph replaying against freed memory (round-5 advisor finding).  Weights
    written through `.data` (no version bump), sync_weights, then replay == eager."""
    import agents
    from gym_vrp.envs import TSPEnv
    from agents import runtime
    env = TSPEnv(20, 64, 1, 69)
    agent = agents.TSPAgent(seed=69, hidden_dim=hidden) if hidden != 512 else agents.TSPAgent(seed=69)
    agent.model.eval()

    def fresh():
        env._visited.zero_(); env._cur.copy_(env._depot); env._mask_fresh = False

    with torch.no_grad():
        for _ in range(3):   # second sighting captures, the third replays
            fresh()
            got0 = runtime.rollout(agent.model, env, True, use_graph=True)
        before = runtime.shadow_pointers(agent.model.encoder)
        fresh()
        base = runtime.rollout(agent.model, deepcopy(env), True, use_graph=False).acc_loss.clone()
        assert torch.equal(got0.acc_loss, base)
        g = torch.Generator(device="cpu").manual_seed(5)
        for prm in agent.model.encoder.parameters():
            prm.data.add_(0.05 * torch.randn(prm.shape, generator=g).to(prm.device))
        for prm in agent.model.decoder.parameters():
            prm.data.add_(0.05 * torch.randn(prm.shape, generator=g).to(prm.device))
        # churn the caching allocator: a freed shadow must not come back by luck
        junk = [torch.empty(1 << 20, dtype=torch.uint8, device=env._device) for _ in range(8)]
        agent.sync_weights()
        del junk
        assert runtime.shadow_pointers(agent.model.encoder) == before
        fresh()
        want = runtime.rollout(agent.model, deepcopy(env), True, use_graph=False).acc_loss.clone()
        assert not torch.equal(want, base)          # the weights did change
        fresh()
        got = runtime.rollout(agent.model, env, True, use_graph=True)
        assert torch.equal(got.acc_loss, want)


@pytest.mark.parametrize("name,path", _load("trainstep_*.npz") + _load("archstep_*.npz"))
def test_training_step_against_reference(name, path):
    """One REINFORCE step (agent.step(env,(False,True)) + backward) on the reference's
    inputs: sampled actions identical (host noise = the reference's CPU stream), loss, T and
    per-parameter gradient norms as the reference computed them (tests/golden/trainstep_*).
    The *_B64_N20 files are SURVEY.md 8a row A1's pins (N=20, B=64, env then agent, seed 69:
    loss 22.715494 / -34.453243 / 7.296117, T 19 / 35 / 30, grad-norm 54.92951 / 134.16078 /
    72.18533)."""
    z = np.load(path)
    kind, B, N = int(z["kind"]), int(z["B"]), int(z["N"])
    if int(z["env_first"]):
        env = _envs()[kind](N, B, 1, 69)
        agent = _agents()[kind](seed=69, **_arch_kw(z))
    else:
        agent = _agents()[kind](seed=69, **_arch_kw(z))
        env = _envs()[kind](N, B, 1, 69)
    agent.model.train()
    agent.model.sampling_noise = agent.target_model.sampling_noise = "host"
    if int(z["torch_seed"]) >= 0:
        torch.manual_seed(int(z["torch_seed"]))
    loss_m, loss_b, logp = agent.step(env, (False, True))
    assert env.step_count == int(z["T"])
    assert np.max(np.abs(loss_m.detach().cpu().numpy() - z["loss_m"])) < TOL
    assert np.max(np.abs(loss_b.cpu().numpy() - z["loss_b"])) < TOL
    # (a sum of T sampled log-probs, |sum| up to 160 on the untrained fixtures: one ulp of the fp32
    # accumulator is 1.5e-5 there and every step rounds it once -- the rule of the rollout tests)
    assert np.max(np.abs(logp.detach().cpu().numpy() - z["logp"])) < TOL * max(5, int(z["T"]) / 4)
    adv = (loss_m - loss_b) * -1
    loss = (adv * logp).mean()
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * max(1.0, abs(float(z["loss"])))
    agent.opt.zero_grad()
    loss.backward()
    got = {k: (p.grad.norm().item() if p.grad is not None else -1.0)
           for k, p in agent.model.named_parameters()}
    for k, want in zip(z["grad_keys"], z["grad_norms"]):
        g = got[str(k)]
        if want < 0:
            assert g < 0, f"{k} must not receive a gradient"
        else:
            # (biases feeding a train-mode BatchNorm have an exactly-zero gradient: both
            # sides hold rounding noise there, hence the floor tied to the total norm)
            assert abs(g - want) <= 2e-3 * want + 2e-6 * float(z["grad_total"]), \
                (str(k), g, float(want))
    tot = np.sqrt(sum(v * v for v in got.values() if v >= 0))
    assert abs(tot - float(z["grad_total"])) < 1e-3 * float(z["grad_total"])
    if name.endswith("_B64_N20"):
        pins = {0: (22.715494, 19, 54.92951), 1: (-34.453243, 35, 134.16078),
                2: (7.296117, 30, 72.18533)}[kind]
        assert abs(loss.item() - pins[0]) < 2e-4 * abs(pins[0]) and env.step_count == pins[1]
        assert abs(tot - pins[2]) < 1e-3 * pins[2]
    nb = agent.model.encoder.attention_layers[0].bn1.norm.num_batches_tracked.item()
    assert nb == 1  # exactly one train-mode encoder pass per step, like the reference
    if "ff0_grad_last" in z.files:
        # zero-padded feed-forward width: the real part of the padded gradients, element-wise
        layers = agent.model.encoder.attention_layers
        for got_g, want_g in ((layers[len(layers) - 1].ff[0].weight.grad, z["ff0_grad_last"]),
                              (layers[0].ff[2].weight.grad, z["ff2_grad_first"])):
            assert got_g.shape == want_g.shape
            err = np.max(np.abs(got_g.cpu().numpy() - want_g))
            assert err <= 2e-3 * np.max(np.abs(want_g)) + 2e-6 * float(z["grad_total"]), err
    agent.opt.step()
    if "ff0_grad_last" in z.files:
        # the shadows follow the optimizer step: a rollout on the updated weights equals the
        # oracle's on the same state dict
        from oracle import envs as oenv
        from oracle import policy as opol
        from agents import runtime
        sd = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}
        agent.model.eval()
        e2 = _envs()[kind](N, B, 1, 7)
        with torch.no_grad():
            res = runtime.rollout(agent.model, e2, True)
            ol, olp, T = opol.rollout(sd, oenv.OracleEnv(kind, N, B, 1, 7), True,
                                      heads=agent.model.encoder._dims[3])
        assert res.T == T
        assert (res.acc_loss.cpu() - ol).abs().max().item() < 1e-5
        assert (res.acc_logp.cpu() - olp).abs().max().item() < 1e-5 * max(1, T / 4)


def test_train_loop_csv_and_checkpoints(tmp_path):
    """TSPAgent.train end to end on the GPU: CSV schema, finite losses, baseline update,
    checkpoint cadence (graph_tsp_agent.py:150-225)."""
    import csv
    import agents
    from gym_vrp.envs import VRPEnv
    env = VRPEnv(num_nodes=10, batch_size=32, seed=3)
    agent = agents.VRPAgent(seed=3, csv_path=str(tmp_path / "log.csv"))
    w0 = agent.model.decoder._kp.weight.detach().clone()
    agent.train(env, epochs=3, check_point_dir=str(tmp_path / "ckpt") + "/")
    rows = list(csv.reader(open(tmp_path / "log.csv")))
    assert rows[0] == ["Epoch", "Loss", "Cost", "Advantage", "Time"] and len(rows) == 4
    assert all(np.isfinite(float(v)) for r in rows[1:] for v in r)
    assert not torch.equal(w0, agent.model.decoder._kp.weight.detach())
    assert os.path.isdir(tmp_path / "ckpt")
    agent.save_model(50, str(tmp_path / "ckpt") + "/")
    sd = torch.load(tmp_path / "ckpt" / "model_epoch_50.pt", map_location="cpu")
    assert list(sd.keys()) == list(agent.model.state_dict().keys())
    fresh = agents.VRPAgent(seed=4)
    fresh.model.load_state_dict(sd)
    loss = fresh.evaluate(VRPEnv(num_nodes=10, batch_size=8, seed=1))
    assert loss.shape == (8,) and torch.isfinite(loss).all()


def test_device_instance_generator():
    """generator="device" (SURVEY 8f rank 4): the reference's distributions from a Philox
    stream on the GPU -- deterministic in (seed, episode), shard-consistent, fresh on every
    reset, lazily visible through the host-side views, and playable."""
    import agents
    from gym_vrp.envs import IRPEnv, TSPEnv
    B, N = 4096, 40
    e1 = IRPEnv(N, B, 1, 123, generator="device")
    e2 = IRPEnv(N, B, 1, 123, generator="device")
    assert torch.equal(e1._pos, e2._pos) and torch.equal(e1._depot, e2._depot)
    assert torch.equal(e1._demand, e2._demand)
    e3 = IRPEnv(N, B, 1, 124, generator="device")
    assert not torch.equal(e1._pos, e3._pos)
    pos, dep, dem = e1._pos.cpu().numpy(), e1._depot.cpu().numpy(), e1._demand.cpu().numpy()
    assert pos.min() >= 0.0 and pos.max() < 1.0
    assert abs(pos.mean() - 0.5) < 5e-3 and abs(pos.var() - 1 / 12) < 2e-3
    assert dep.min() >= 0 and dep.max() == N - 1
    counts = np.bincount(dep, minlength=N)
    assert counts.min() > 0.5 * B / N and counts.max() < 1.6 * B / N
    scale = 0.2449 * N + 26.12
    assert (dem[np.arange(B), dep] == 0).all()
    others = dem[dem > 0]
    assert others.size == B * (N - 1)
    assert others.min() >= 1 / scale and others.max() < 10 / scale
    assert abs(others.mean() * scale - 5.5) < 0.05
    # all coordinates distinct draws (no counter reuse between x, y, demand, graphs)
    assert np.unique(pos.round(12)).size > 0.999 * pos.size
    # shards reproduce the rows of the unsharded batch
    half = IRPEnv(N, B, 1, 123, generator="device", shard=(1, 2))
    assert torch.equal(half._pos, e1._pos[B // 2:]) and torch.equal(half._depot, e1._depot[B // 2:])
    # reset draws new instances; host-side views follow lazily
    before = e1._pos.clone()
    e1.reset()
    assert not torch.equal(before, e1._pos)
    assert np.array_equal(e1.depots[:, 0], e1._depot.cpu().numpy())
    assert np.allclose(e1.sampler.get_graph_positions(), e1._pos.cpu().numpy())
    assert np.array_equal(e1.demands[:, :, 0], e1._demand.cpu().numpy())
    st, load = e1.get_state()
    assert st.shape == (B, N, 5) and load.shape == (B,)
    # and a rollout runs on them
    agent = agents.IRPAgent(seed=69)
    loss = agent.evaluate(e1)
    assert torch.isfinite(loss).all() and (loss < 0).all()
    t = TSPEnv(20, 64, 1, 5, generator="device")
    assert torch.isfinite(agents.TSPAgent(seed=69).evaluate(t)).all()


def test_device_random_agent():
    """RandomAgent(on_device=True): Philox draws on the GPU -- valid tours, the reference's
    cost level, deterministic in the seed."""
    import ctypes as C
    import agents
    import vrpgym_hip as hip
    from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
    B, N = 2048, 20
    env = TSPEnv(N, B, 1, 7, generator="device")
    cost = -agents.RandomAgent(seed=3, on_device=True)(env)
    assert cost.shape == (B,) and torch.isfinite(cost).all()
    # N-1 uniform-random edges in the unit square: E[d] = 0.5214 each
    assert abs(cost.mean().item() - (N - 1) * 0.5214) < 0.15
    assert env.step_count == N - 1
    # same seed and instances -> same tours; explicit action trace is a permutation
    env2 = TSPEnv(N, B, 1, 7, generator="device")
    assert torch.equal(-agents.RandomAgent(seed=3, on_device=True)(env2), cost)
    env3 = TSPEnv(N, B, 1, 7, generator="device")
    lib = hip.lib()
    acc = torch.empty(B, device="cuda")
    nd = torch.empty(N, dtype=torch.int32, device="cuda")
    acts = torch.full((N - 1, B), -1, dtype=torch.int64, device="cuda")
    cenv = env3._cenv()
    hip.check(lib.vrp_random_rollout(C.byref(cenv), 3, 0, 0, N - 1, acc.data_ptr(), nd.data_ptr(),
                                     acts.data_ptr(), hip.current_stream()))
    assert torch.equal(-acc, cost)
    a = acts.t().cpu().numpy()                                    # (B, N-1)
    dep = env3._depot.cpu().numpy()
    for bi in range(0, B, 97):
        assert sorted(a[bi].tolist()) == [n for n in range(N) if n != dep[bi]]
    # each first move is uniform over the N-1 customers
    first = np.bincount(a[:, 0], minlength=N)
    assert first.max() < 1.6 * B / (N - 1)
    for Env in (VRPEnv, IRPEnv):
        e = Env(N, 256, 1, 11, generator="device")
        c = -agents.RandomAgent(seed=5, on_device=True)(e)
        assert torch.isfinite(c).all() and (c > 0).all() and N <= e.step_count <= 2 * (N - 1)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_rollout_parity_with_trained_weights(kind, tmp_path):
    """After some REINFORCE epochs the weights (and the BatchNorm running statistics) have
    left their initial values and the glimpse scores are larger: the HIP rollout still
    matches the oracle run on the SAME trained state dict, greedy and sampled."""
    import logging
    logging.disable(logging.CRITICAL)
    agent = _agents()[kind](seed=69, csv_path=str(tmp_path / "log.csv"))
    env = _envs()[kind](20, 128, 1, 69)
    agent.train(env, epochs=40, check_point_dir=str(tmp_path) + "/")
    logging.disable(logging.NOTSET)
    moved = (agent.model.decoder._kp.weight - _agents()[kind](seed=69).model.decoder._kp.weight)
    assert moved.abs().max().item() > 1e-4
    for greedy in (True, False):
        _compare_rollout(kind, 64, 20, greedy, 321, 69, 17, agent=agent)
    _compare_rollout(kind, 33, 40, True, 322, 69, 18, agent=agent, throughput_kernel=True)


def test_video_frames_and_edges_under_fused_rollout(monkeypatch):
    """reproduction.py:37-47 flow: enable_video_capturing(...) then agent.evaluate(env).  The
    reference captures one frame and records one edge per env.step (tsp.py:88-93); the fused
    rollout replays exactly that on the host.  Also: render() on a watched env."""
    import sys
    import types
    frames, locs = [], []

    class CountingRecorder:
        def __init__(self, env=None, path=None, **kw):
            self.env, self.path, self.frames_per_sec = env, path, None

        def capture_frame(self):
            frames.append(set(self.env.sampler.graphs[0].visited_edges))
            # the reference captures INSIDE env.step (tsp.py:92-93): a recorder that looks at
            # the env sees that step's location and visited flags
            locs.append((int(self.env.current_location[0, 0]), int(self.env.visited[0].sum()),
                         int(self.env.step_count)))

        def close(self):
            pass

    gym = types.ModuleType("gym")
    wr, mon, vr = (types.ModuleType("gym.wrappers"), types.ModuleType("gym.wrappers.monitoring"),
                   types.ModuleType("gym.wrappers.monitoring.video_recorder"))
    vr.VideoRecorder = CountingRecorder
    gym.wrappers, wr.monitoring, mon.video_recorder = wr, mon, vr
    for name, m in (("gym", gym), ("gym.wrappers", wr), ("gym.wrappers.monitoring", mon),
                    ("gym.wrappers.monitoring.video_recorder", vr)):
        monkeypatch.setitem(sys.modules, name, m)
    import agents
    from gym_vrp.envs import IRPEnv, TSPEnv
    from oracle import envs as oenv
    from oracle import policy as opol
    for kind, Env, Agent in ((0, TSPEnv, agents.TSPAgent), (2, IRPEnv, agents.IRPAgent)):
        del frames[:], locs[:]
        env = Env(num_nodes=9, batch_size=6, num_draw=2, seed=11)
        agent = Agent(seed=69)
        env.enable_video_capturing("unused.mp4")
        assert env.vid.frames_per_sec == 1
        loss = agent.evaluate(env)
        T = env.step_count
        sd, _ = opol.init_state_dicts(kind, 69)
        oe = oenv.OracleEnv(kind, 9, 6, 2, 11)
        tr = []
        start0 = int(oe.depots[0, 0])
        with torch.no_grad():
            oloss, _, oT = opol.rollout(sd, oe, True, trace=tr)
        assert T == oT and len(frames) == T          # one frame per env.step
        assert float((loss.cpu() - oloss).abs().max()) < 1e-5
        # frame t shows the first t+1 edges of graph 0's tour (self-loops are not edges)
        tour0 = [start0] + [int(s_["idx"][0]) for s_ in tr]
        for t in range(T):
            want = {(min(a, b), max(a, b)) for a, b in zip(tour0[: t + 1], tour0[1: t + 2])
                    if a != b}
            assert frames[t] == want, (t, frames[t], want)
            assert locs[t][0] == tour0[t + 1] and locs[t][2] == t + 1, (t, locs[t])
        assert [l[1] for l in locs] == [int(s_["visited_after"][0].sum()) for s_ in tr]
        assert env.step_count == T
        img = env.render()
        assert img.ndim == 3 and img.shape[2] == 3 and img.shape[0] > 0
    # an env nobody watches records nothing and keeps the fused path free of host work
    env = TSPEnv(num_nodes=9, batch_size=6, num_draw=2, seed=11)
    agents.TSPAgent(seed=69).evaluate(env)
    assert env.sampler._graphs is None


def test_decoder_weight_changes_are_seen():
    """The folded decoder matrices are cached per parameter version.  In-place writes on the
    parameters (optimizer, load_state_dict, copy_ under no_grad) are noticed; writes through
    `.data` are not visible to autograd's counters and need agent.sync_weights()."""
    import agents
    from agents import runtime
    from gym_vrp.envs import TSPEnv
    agent = agents.TSPAgent(seed=69)
    env = TSPEnv(num_nodes=10, batch_size=8, num_draw=1, seed=3)

    def logits():
        e = deepcopy(env)
        agent.model.eval()
        with torch.no_grad():
            res = runtime.rollout(agent.model, e, True, trace=True)
        return res.logits[0].clone()

    base = logits()
    assert torch.equal(base, logits())
    w = agent.model.decoder._kp.weight
    with torch.no_grad():
        w.mul_(1.5)                       # bumps w._version
    l1 = logits()
    assert not torch.equal(base, l1)
    w.data.mul_(1.0 / 1.5)                # invisible to the version counter ...
    agent.sync_weights()                  # ... so the caller says so
    l2 = logits()
    fin = torch.isfinite(base)
    assert torch.equal(torch.isfinite(l2), fin)
    assert float((l2[fin] - base[fin]).abs().max()) < 1e-4
    sd = {k: v.clone() for k, v in agent.model.state_dict().items()}
    sd["decoder._kp.weight"] *= 2.0
    agent.model.load_state_dict(sd)       # copy_ under no_grad: noticed automatically
    assert not torch.equal(logits(), l2)


@pytest.mark.parametrize("kind,B,N", [(0, 257, 20), (1, 300, 33), (2, 300, 40), (2, 64, 100)])
def test_device_random_rollout_replays_through_oracle(kind, B, N):
    """the device-side RandomAgent (Philox draws + the shared env-step code)
    on device-generated instances.  Its RNG is not the reference's, but its BOOKKEEPING must
    be: the recorded actions, replayed through the oracle env on the fetched instances, are
    feasible at every step (mask 0 under the oracle's mask), end the episode at the same
    step (per-step `done`), leave bit-identical visited / mask / load / location state, and
    give the fp32-identical accumulated cost."""
    import ctypes as C
    import vrpgym_hip as hip
    from agents import runtime
    from oracle import envs as oenv
    env = _envs()[kind](N, B, 1, 21, generator="device")
    steps = runtime.max_steps_for(kind, N)
    acc = torch.empty(B, device="cuda")
    nd = torch.full((steps + 1,), -1, dtype=torch.int32, device="cuda")
    acts = torch.full((steps, B), -1, dtype=torch.int64, device="cuda")
    cenv = env._cenv()
    hip.check(hip.lib().vrp_random_rollout(C.byref(cenv), 9, 0, 0, steps, acc.data_ptr(),
                                           nd.data_ptr(), acts.data_ptr(), hip.current_stream()))
    torch.cuda.synchronize()
    nd, acts = nd.cpu().numpy(), acts.cpu().numpy()
    T = int(np.flatnonzero(nd[:steps] == 0)[0]) + 1
    o = oenv.OracleEnv(kind, N, B, 1, 0)
    o.pos = env._pos.cpu().numpy().copy()
    o.depots = env._depot.cpu().numpy().astype(np.int64)[:, None]
    o.demands = env._demand.cpu().numpy()[:, :, None].copy()
    o.visited = np.zeros((B, N))
    o.current_location = o.depots
    if kind == 2:
        o.load = np.ones(B)
    state = o.get_state()
    total = torch.zeros(B)
    rows = np.arange(B)
    for t in range(T):
        mask = (state[0] if kind == 2 else state)[:, :, -1]
        assert np.all(mask[rows, acts[t]] == 0), f"step {t}: a masked node was drawn"
        state, reward, done, _ = o.step(acts[t][:, None])
        total += torch.tensor(reward, dtype=torch.float)
        assert done == (t == T - 1), (t, T)
        assert (nd[t] == 0) == done
    fmask = (state[0] if kind == 2 else state)[:, :, -1]
    par = T & 1   # step t writes mask buffer (t+1)&1
    assert np.array_equal(env._visited.cpu().numpy(), o.visited.astype(np.uint8))
    assert np.array_equal(env._mask[par].cpu().numpy(), fmask.astype(np.uint8))
    assert np.array_equal(env._cur.cpu().numpy(), o.current_location[:, 0])
    if kind == 2:
        assert np.array_equal(env._load.cpu().numpy(), o.load)
    assert torch.equal(acc.cpu(), total)
    # steps after `done` are exact no-ops
    assert np.all(acts[T:] == -1)


@pytest.mark.parametrize("kind,B,N,greedy,train", [(2, 1024, 40, False, True),
                                                   (1, 2048, 100, False, False)],
                         ids=["config4_shard_irp40_b1024_train", "config5_shard_vrp100_b2048"])
def test_full_size_properties_configs_4_5(kind, B, N, greedy, train):
    """BASELINE configs[3] and [4] at their per-GPU shard sizes (IRP 1024 x 40 train-mode
    sampling, VRP 2048 x 100 sampling): too big for the oracle policy to replay quickly, so
    the HIP rollout is checked through size-independent properties -- the recorded tour,
    replayed through the ORACLE ENV (numpy, cheap), reproduces every mask the decoder saw,
    the per-step `done`, the load trace and the fp32 cost bit for bit / within 1e-5; the
    per-step log-probs are a valid distribution's (<= 0, sum = acc_logp); idempotence."""
    from agents import runtime
    from oracle import envs as oenv
    env = _envs()[kind](N, B, 1, 5)
    agent = _agents()[kind](seed=69)
    agent.model.train(train)
    steps = runtime.max_steps_for(kind, N)
    noise = torch.empty((steps, B, N)).exponential_(1, generator=torch.Generator().manual_seed(1))
    e1, e2 = deepcopy(env), deepcopy(env)
    with torch.no_grad():
        res = runtime.rollout(agent.model, e1, greedy, train=train, noise=noise, trace=True,
                              record=True)
        agent2 = _agents()[kind](seed=69)
        agent2.model.train(train)
        res2 = runtime.rollout(agent2.model, e2, greedy, train=train, noise=noise, trace=True)
    T = res.T
    assert res2.T == T and torch.equal(res.actions, res2.actions)
    assert torch.equal(res.acc_logp, res2.acc_logp) and torch.equal(res.acc_loss, res2.acc_loss)
    acts = res.actions[:T].cpu().numpy()
    masks = res.mask_trace[:T].cpu().numpy()
    o = oenv.OracleEnv(kind, N, B, 1, 5)          # same seed -> same numpy-stream instances
    assert np.array_equal(o.pos, env.sampler.get_graph_positions())
    state = o.get_state()
    total = torch.zeros(B)
    rows = np.arange(B)
    for t in range(T):
        mask = (state[0] if kind == 2 else state)[:, :, -1]
        assert np.array_equal(masks[t], mask.astype(np.uint8)), t
        if kind == 2:
            assert np.array_equal(res.load_trace[t].cpu().numpy(), o.load.astype(np.float32))
        assert np.all(mask[rows, acts[t]] == 0)
        state, reward, done, _ = o.step(acts[t][:, None])
        total += torch.tensor(reward, dtype=torch.float)
        assert done == (t == T - 1)
    assert (res.acc_loss.cpu() - total).abs().max().item() < TOL
    slp = res.step_logp[:T].cpu()
    assert (slp <= 0).all() and torch.isfinite(slp).all()
    acc = torch.zeros(B)
    for t in range(T):          # fp32 accumulation in step order (graph_tsp_agent.py:86)
        acc += slp[t]
    assert torch.equal(acc, res.acc_logp.cpu())
    # the logits the step kernel produced are finite exactly where the mask is 0, and the
    # log-prob it reports is log_softmax(u)[a] of those logits
    u = res.logits[:T].cpu()
    assert torch.equal(torch.isfinite(u), torch.as_tensor(masks == 0))
    lsm = torch.log_softmax(u.double(), dim=2).gather(2, torch.as_tensor(acts)[:, :, None])[..., 0]
    assert (lsm.float() - slp).abs().max().item() < TOL


def _persistent_equal(kind, B, steps, ref, got):
    (r0, e0), (r1, e1) = ref, got
    T = r0.T
    assert r1.T == T
    assert torch.equal(r0.notdone[:steps], r1.notdone[:steps])
    assert torch.equal(r0.actions[:T], r1.actions[:T])
    assert torch.equal(r0.step_logp[:T], r1.step_logp[:T])
    assert torch.equal(r0.mask_trace[:T], r1.mask_trace[:T])
    if kind == 2:
        assert torch.equal(r0.load_trace[:T], r1.load_trace[:T])
    assert torch.equal(r0.acc_loss, r1.acc_loss) and torch.equal(r0.acc_logp, r1.acc_logp)
    assert torch.equal(e0._visited, e1._visited) and torch.equal(e0._cur, e1._cur)
    assert torch.equal(e0._load, e1._load)


@pytest.mark.parametrize("kind,B,N,greedy,train", [
    (0, 512, 20, True, False), (0, 512, 20, False, True), (1, 300, 33, True, False),
    (1, 2048, 40, False, True), (2, 1024, 40, False, True), (2, 37, 63, True, False),
    (1, 5, 3, False, False), (2, 64, 21, False, False)])
def test_persistent_steps_equal_per_step_launches(kind, B, N, greedy, train):
    """The persistent multi-step kernel (one launch for steps 1..T-1: masks handed between
    graphs as published words, per-graph termination, the forced way back of a graph that
    finished by leaving the depot counted iff the batch ran on) against one launch per step:
    same arithmetic, so everything is bit-identical -- actions, traces the backward pass uses,
    accumulators, step count, final env state."""
    from agents import runtime
    env = _envs()[kind](N, B, 1, 13)
    agent = _agents()[kind](seed=69)
    agent.model.train(train)
    steps = runtime.max_steps_for(kind, N)
    noise = torch.empty((steps, B, N)).exponential_(1, generator=torch.Generator().manual_seed(2))
    import vrpgym_hip as hip
    out = []
    # one launch per step, then the persistent kernel one, two and four waves per graph wide
    # (VRP_PERSISTENT_WAVES is read at every call; a width that would not be resident falls back
    # to the one-wave kernel -- all of them must reproduce the per-step results bit for bit)
    for waves in (None, "1", "2", "4"):
        if waves is None:
            os.environ.pop("VRP_PERSISTENT_WAVES", None)
        else:
            os.environ["VRP_PERSISTENT_WAVES"] = waves
        try:
            e = deepcopy(env)
            a = _agents()[kind](seed=69)
            a.model.train(train)
            with torch.no_grad():
                r = runtime.rollout(a.model, e, greedy, train=train, noise=noise, record=True,
                                    step_trace=True, persistent=waves is not None)
            r.T
        finally:
            os.environ.pop("VRP_PERSISTENT_WAVES", None)
        out.append((r, e))
    for other in out[2:]:
        _persistent_equal(kind, B, steps, out[0], other)
    (r0, e0), (r1, e1) = out[0], out[1]
    T = r0.T
    assert r1.T == T
    assert torch.equal(r0.notdone[:steps], r1.notdone[:steps])
    assert torch.equal(r0.actions[:T], r1.actions[:T])
    assert torch.equal(r0.step_logp[:T], r1.step_logp[:T])
    assert torch.equal(r0.mask_trace[:T], r1.mask_trace[:T])
    if kind == 2:
        assert torch.equal(r0.load_trace[:T], r1.load_trace[:T])
    assert torch.equal(r0.acc_loss, r1.acc_loss) and torch.equal(r0.acc_logp, r1.acc_logp)
    assert torch.equal(e0._visited, e1._visited) and torch.equal(e0._cur, e1._cur)
    assert torch.equal(e0._load, e1._load)
    assert np.array_equal(e0.generate_mask(), e1.generate_mask())
    assert e0.step_count == e1.step_count == T
    if kind != 0 and not greedy and B >= 8:
        # the interesting case must occur: graphs that finish at different steps
        acts = r0.actions[:T].cpu().numpy()
        dep = e0._depot.cpu().numpy()
        last_move = np.array([np.flatnonzero(acts[:, b] != dep[b]).max() for b in range(B)])
        assert last_move.min() < last_move.max()


_FIRST_FOLD_CHILD = r"""
import hashlib, os, sys
sys.path[:0] = [os.path.join(sys.argv[1], "vrp-gym_amd"), sys.argv[1]]
import numpy as np, torch
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from agents import runtime
import agents
from gym_vrp.envs import TSPEnv, VRPEnv
out = {}
for kind, B, N, greedy in ((0, 512, 20, True), (1, 100, 21, False), (0, 33, 5, True), (1, 1024, 40, True), (1, 7, 63, False)):
    env = (TSPEnv, VRPEnv)[kind](num_nodes=N, batch_size=B, num_draw=1, seed=13)
    agent = (agents.TSPAgent, agents.VRPAgent)[kind](seed=69)
    agent.model.eval()
    steps = runtime.max_steps_for(kind, N)
    noise = torch.empty((steps, B, N)).exponential_(1, generator=torch.Generator().manual_seed(2))
    for persistent in (True, False):
        e = __import__("copy").deepcopy(env)
        with torch.no_grad():
            r = runtime.rollout(agent.model, e, greedy, noise=None if greedy else noise, record=True,
                                persistent=persistent)
        T = r.T
        out[f"{kind}_{B}_{N}_{int(persistent)}_loss"] = r.acc_loss.cpu().numpy()
        out[f"{kind}_{B}_{N}_{int(persistent)}_logp"] = r.acc_logp.cpu().numpy()
        out[f"{kind}_{B}_{N}_{int(persistent)}_act"] = r.actions[:T].cpu().numpy()
np.savez(sys.argv[2], **out)
"""


@pytest.mark.gpu
def test_first_node_fold_against_gemm_route(tmp_path):
    """Small batches (B <= 1024, N <= 63, TSP / VRP) take the first chosen node's part of the score
    rows from persist_first_base -- inside the persistent grid or as first_base_kernel -- instead of
    the first-node GEMM + score_base_kernel (graph_decoder.py:88-92).  Same rollouts with
    VRP_NO_KEEP_KEYS=1 (the GEMM route at every shape): costs and log-probs within the eval
    tolerance, the same tours except on near-ties; and within ONE build the persistent and the
    per-step path stay bit-identical (base is a function of the shape, not of the path)."""
    import sys
    import _proc
    res = {}
    for tag, extra in (("keys", {}), ("gemm", {"VRP_NO_KEEP_KEYS": "1"})):
        env = {k: v for k, v in os.environ.items() if k != "VRP_NO_KEEP_KEYS"}
        env.update(extra)
        out = str(tmp_path / f"{tag}.npz")
        r = _proc.run([sys.executable, "-c", _FIRST_FOLD_CHILD, ROOT, out], env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(out)
    a, g = res["keys"], res["gemm"]
    for key in a.files:
        if key.endswith("_1_loss") or key.endswith("_1_logp") or key.endswith("_1_act"):
            # persistent == per-step, bit for bit, in both builds
            other = key.replace("_1_", "_0_")
            assert np.array_equal(a[key], a[other]), key
            assert np.array_equal(g[key], g[other]), key
    for key in a.files:
        if key.endswith("_act"):
            same = (a[key] == g[key]).all(axis=0)     # graphs whose whole tour agrees
            assert same.mean() >= 0.97, (key, same.mean())
            stem = key[:-4]
            T = a[key].shape[0]
            assert np.max(np.abs(a[stem + "_loss"][same] - g[stem + "_loss"][same])) < TOL
            assert np.max(np.abs(a[stem + "_logp"][same] - g[stem + "_logp"][same])) < TOL * max(1, T / 4)


@pytest.mark.parametrize("kind,B,N,greedy", [(1, 64, 100, False), (2, 3000, 40, False),
                                             (1, 3000, 40, True), (2, 16, 30, False)])
def test_paced_step_loop_equals_fixed_length_loop(kind, B, N, greedy):
    """vrp_rollout_steps_range stops queueing per-step launches once the device reports the
    batch finished (a pinned flag word per chunk of eight steps, from step N - 1 on) instead of
    queueing all 2 (N - 1): everything a caller can read must equal the fixed-length loop's
    (VRP_NO_THROTTLE=1), including the step count and the per-step done flags."""
    from agents import runtime
    env = _envs()[kind](N, B, 1, 21)
    steps = runtime.max_steps_for(kind, N)
    noise = torch.empty((steps, B, N)).exponential_(1, generator=torch.Generator().manual_seed(5))
    out = []
    for off in (True, False):
        if off:
            os.environ["VRP_NO_THROTTLE"] = "1"
        else:
            os.environ.pop("VRP_NO_THROTTLE", None)
        try:
            e = deepcopy(env)
            a = _agents()[kind](seed=69)
            with torch.no_grad():
                r = runtime.rollout(a.model, e, greedy, noise=None if greedy else noise, record=True,
                                    step_trace=True, persistent=False)
            r.T
        finally:
            os.environ.pop("VRP_NO_THROTTLE", None)
        out.append((r, e))
    _persistent_equal(kind, B, steps, out[0], out[1])
    assert out[0][0].T < steps or greedy   # sampled tours end early: the case the pacing is for


def test_training_reaches_the_reference_cost_level(tmp_path):
    """Solution quality, not just throughput: train_models.py's TSP-20 setting (batch 256, seed
    69, Adam 1e-4, rollout baseline with the paired t-test) for 300 of its 851 epochs, then
    reproduction.py's greedy evaluation on fresh seed-1234 instances.  The reference's own log
    reads 4.62 at epoch 300 and 4.32 at epoch 850 (sampled cost,
    train_logs/loss_log_tsp_20_69.csv:302,852), its greedy evaluation 4.16
    (reproduction_log/reproduction_results_20_nodes_model_TSP.csv); the full 851-epoch sweep of
    this package is in profiles/r03_train/summary.md (greedy 4.175)."""
    import csv
    import logging
    import agents
    from gym_vrp.envs import TSPEnv
    logging.disable(logging.CRITICAL)
    try:
        env = TSPEnv(num_nodes=20, batch_size=256, seed=69)
        agent = agents.TSPAgent(seed=69, csv_path=str(tmp_path / "log.csv"))
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            agent.train(env, epochs=300, check_point_dir=str(tmp_path) + "/")
    finally:
        logging.disable(logging.NOTSET)
    rows = list(csv.reader(open(tmp_path / "log.csv")))[1:]
    first, last = -float(rows[0][2]), -float(rows[-1][2])
    assert 8.8 < first < 9.7, first           # the reference's epoch-0 log line: 9.25
    assert last < 5.1, last                   # the reference at epoch 300: 4.62
    greedy = -agent.evaluate(TSPEnv(num_nodes=20, batch_size=256, num_draw=6, seed=1234)).mean().item()
    assert greedy < 4.5, greedy


@pytest.mark.parametrize("kind,B,N", [(0, 512, 20), (1, 300, 40), (2, 64, 21), (1, 24, 100)])
def test_rollout_with_in_kernel_reset_equals_reset_then_rollout(kind, B, N):
    """runtime.rollout(reset_env=True) -- VRP_ENV_RESET_ON_ROLLOUT: the state part of env.reset()
    runs inside the rollout's set-up kernel -- equals _reset_state() followed by a rollout, bit
    for bit, on an env that has already been played."""
    from agents import runtime
    env = _envs()[kind](N, B, 1, 4)
    agent = _agents()[kind](seed=69)
    agent.model.eval()
    with torch.no_grad():
        first = runtime.rollout(agent.model, env, True, step_trace=True)
        T = first.T
        assert env.step_count == T and env._visited.any()
        again = runtime.rollout(agent.model, env, True, step_trace=True, reset_env=True)
        assert again.T == T and env.step_count == T
        env._reset_state()
        env._step_count, env._last_rollout = 0, None
        ref = runtime.rollout(agent.model, env, True, step_trace=True)
    for a, b in ((again, first), (again, ref)):
        assert torch.equal(a.acc_loss, b.acc_loss) and torch.equal(a.actions[:T], b.actions[:T])


@pytest.mark.parametrize("kind,B,N", [(0, 2048, 20), (1, 512, 100), (2, 1024, 40), (0, 4096, 40)])
def test_in_kernel_sampling_noise(kind, B, N):
    """Throughput mode draws the Exp(1) noise of Categorical.sample inside the step kernels
    (Philox, counter = graph / node / step, keyed by a per-rollout seed from the CPU generator)
    instead of shipping a (max_steps, B, N) tensor.  Not the reference's stream, so no
    action-level parity -- but: reproducible under torch.manual_seed, different across
    rollouts, the same draw in every step kernel (persistent, table, raw-tile: same counter),
    log-probs those of the chosen actions, and a cost distribution matching host-noise
    sampling."""
    from agents import runtime
    env = _envs()[kind](N, B, 1, 8)
    agent = _agents()[kind](seed=69)
    agent.model.eval()

    def run(seed, **kw):
        torch.manual_seed(seed)
        with torch.no_grad():
            return runtime.rollout(agent.model, deepcopy(env), False, step_trace=True, **kw)

    a, b, c = run(1), run(1), run(2)
    T = a.T
    assert b.T == T and torch.equal(a.actions[:T], b.actions[:T]) and torch.equal(a.acc_logp, b.acc_logp)
    assert not torch.equal(a.actions[:min(T, c.T)], c.actions[:min(T, c.T)])
    # the same noise whichever kernel consumes it
    d = run(1, persistent=False)
    assert d.T == T and torch.equal(a.actions[:T], d.actions[:T])
    if N <= 104 and kind != 2:
        e = run(1, persistent=False, tile_kernel=True)
        flips = (e.actions[:T] != a.actions[:T]).any(dim=0).float().mean().item()
        assert flips <= 0.01, flips      # other arithmetic for the logits: near ties only
    assert (a.step_logp[:T] <= 0).all() and torch.isfinite(a.acc_logp).all()
    acc = torch.zeros(B, device=a.acc_logp.device)
    for t in range(T):
        acc += a.step_logp[t]
    assert torch.equal(acc, a.acc_logp)
    # distribution: mean sampled cost vs host-noise sampling of the same policy
    h = run(3, noise_mode="host")
    ma, mh = -a.acc_loss.mean().item(), -h.acc_loss.mean().item()
    sd = a.acc_loss.std().item() / (B ** 0.5)
    assert abs(ma - mh) < 6 * sd + 0.01 * mh, (ma, mh, sd)


def test_in_kernel_noise_is_strictly_positive_and_finite():
    """The bits -> Exp(1) map of the in-kernel sampler: q in (0, inf) for EVERY 32-bit draw.
    (q = -0.0 at the largest draw made p/q = -inf on the one selectable node of a forced move
    and let a masked node win the argmax: advisor finding, round 3.)"""
    import vrpgym_hip as hip
    lib = hip.lib()
    edge = torch.tensor([0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFE00, 0xFFFFFF00,
                         0xFFFFFFFE, 0xFFFFFFFF], dtype=torch.int64)
    rnd = torch.randint(0, 2 ** 32, (1 << 20,), dtype=torch.int64)
    bits = torch.cat([edge, rnd]).numpy().astype(np.uint32)
    d_bits = torch.from_numpy(bits.view(np.int32)).cuda()
    out = torch.empty(len(bits), dtype=torch.float32, device="cuda")
    hip.check(lib.vrp_debug_exp1_from_bits(d_bits.data_ptr(), out.data_ptr(), len(bits),
                                           hip.current_stream()))
    q = out.cpu().numpy()
    assert np.all(q > 0) and np.all(np.isfinite(q)) and not np.any(np.signbit(q))
    u = ((bits >> 9).astype(np.float64) + 0.5) / 2.0 ** 23
    assert np.max(np.abs(q - (-np.log(u))) / np.maximum(-np.log(u), 1e-6)) < 1e-3
    assert abs(q[len(edge):].mean() - 1.0) < 5e-3      # Exp(1)


@pytest.mark.parametrize("C", [5.0, 10.0, 20.0])
def test_decoder_clipping_constant(C):
    """GraphDecoder.forward(..., C) (graph_decoder.py:56,97): u = C tanh(.), so against the
    reference's recorded C = 10 logits u_C = (C / 10) u_10 on the unmasked nodes: the greedy
    node is the same, the sampled one is argmax(softmax(u_C) / q) with the CPU noise stream.
    (First step of an episode: the later recorded steps assume the C = 10 choices.)"""
    z = np.load(os.path.join(G, "decoder_k1_B64_N20.npz"))
    agent = _agents()[1](seed=69)
    dec = agent.model.decoder
    emb = torch.tensor(z["emb"])
    B, N = emb.shape[:2]
    mask = torch.tensor(z["mask"][0])
    u = torch.tensor(z["u"][0]) * (C / 10.0)
    dec.reset()
    idx, _ = dec(emb, mask=mask, C=C, rollout=True)
    assert torch.equal(idx[:, 0].cpu(), u.argmax(-1))
    dec.reset()
    torch.manual_seed(1234)
    idx, logp = dec(emb, mask=mask, C=C, rollout=False)
    torch.manual_seed(1234)
    noise = torch.empty(B, N).exponential_(1)
    lsm = u - u.logsumexp(-1, keepdim=True)
    ratio = torch.softmax(lsm, -1) / noise
    want, got = ratio.argmax(-1), idx[:, 0].cpu()
    flip = got != want
    if flip.any():   # only on a near tie of the ratios
        best = ratio.max(-1).values
        mine = ratio.gather(1, got[:, None])[:, 0]
        assert (((best - mine) / best)[flip] < TIE_GAP).all()
    want_lp = lsm.gather(1, got[:, None])[:, 0]
    assert (logp.reshape(-1).cpu() - want_lp).abs().max().item() < 2 * TOL * max(1.0, C / 10.0)
    dec.reset()


@pytest.mark.parametrize("kind,B,N,greedy", [
    (1, 2048, 40, True),     # BASELINE config 3's batch (the rollout its training steps run)
    (2, 1024, 40, True),     # config 4's per-GPU shard
    (1, 2048, 100, False),   # config 5's per-GPU shard: sampling, raw-tile + table kernels
])
def test_full_size_rollout_against_oracle(kind, B, N, greedy):
    """The BASELINE configs' per-GPU batches against the oracle itself (not only through
    size-independent properties): every action, per-step logits and log-probs, cost and
    accumulated log-prob, with the near-tie rule's allowance of B / 100 graphs."""
    _compare_rollout(kind, B, N, greedy, 21, 69, 9)

"""The persistent multi-step kernel is fast only while its whole grid is resident (its waves
wait for each other's mask words) -- and must be CORRECT whether or not it is.  These tests run
the product in child processes: with the HIP runtime restricted to 32 compute units
(ROC_GLOBAL_CU_MASK) eligibility must follow the usable CUs (one launch per step, decided before
the episode); a grid that is forced through anyway must drain within tens of milliseconds and
fall back inside persistent_finalize_kernel -- same actions, same accumulators, bit for bit, no
NaN, no exception; and two INDEPENDENT processes on one GPU (the situation that hung round 4's
driver run) finish quickly with exactly their solo results, with the cross-process lease and
without it."""
import json
import os
import sys
import time

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _proc  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path[:0] = [os.path.join(%(root)r, "vrp-gym_amd"), %(root)r]
import torch
import agents, vrpgym_hip as hip
from copy import deepcopy
from agents import runtime
from gym_vrp.envs import TSPEnv, VRPEnv, IRPEnv
cap = hip.lib().vrp_persistent_capacity()
B, N = int(os.environ.get("GUARD_B", "2048")), 20
KIND = int(os.environ.get("GUARD_KIND", "1"))
env = (TSPEnv, VRPEnv, IRPEnv)[KIND](N, B, 1, 13)
agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[KIND](seed=69)
agent.model.eval()
steps = runtime.max_steps_for(KIND, N)
noise = torch.empty((steps, B, N)).exponential_(1, generator=torch.Generator().manual_seed(2))
out = {"capacity": cap, "kernel": hip.lib().vrp_step_kernel_name(KIND, B, N, 0).decode()}
res = []
for persistent in (False, True):
    with torch.no_grad():
        r = runtime.rollout(agent.model, deepcopy(env), False, noise=noise, step_trace=True,
                            persistent=persistent)
    torch.cuda.synchronize()
    res.append(r)
# the traces a training step records (masks, IRP loads) through both paths
rec = []
for persistent in (False, True):
    with torch.no_grad():
        r = runtime.rollout(agent.model, deepcopy(env), False, noise=noise, record=True,
                            persistent=persistent)
    torch.cuda.synchronize()
    rec.append(r)
Tr = rec[0].T
out["traces_equal"] = bool(rec[1].T == Tr and torch.equal(rec[0].actions[:Tr], rec[1].actions[:Tr])
                           and torch.equal(rec[0].mask_trace[:Tr], rec[1].mask_trace[:Tr])
                           and (rec[0].load_trace is None
                                or torch.equal(rec[0].load_trace[:Tr], rec[1].load_trace[:Tr]))
                           and torch.equal(rec[0].acc_logp, rec[1].acc_logp))
final = [deepcopy(env) for _ in range(2)]
for e, persistent in zip(final, (False, True)):
    with torch.no_grad():
        runtime.rollout(agent.model, e, True, persistent=persistent)
torch.cuda.synchronize()
out["env_state_equal"] = bool(torch.equal(final[0]._visited, final[1]._visited)
                              and torch.equal(final[0]._cur, final[1]._cur)
                              and torch.equal(final[0]._load, final[1]._load))
out["nan"] = bool(torch.isnan(res[1].acc_loss).any())
out["T"] = [res[0].T, res[1].T]
out["failures"] = hip.lib().vrp_persistent_failures()
T = min(out["T"])
out["equal"] = bool(torch.equal(res[0].acc_loss, res[1].acc_loss)
                    and torch.equal(res[0].acc_logp, res[1].acc_logp)
                    and torch.equal(res[0].actions[:T], res[1].actions[:T]))
print("RESULT " + json.dumps(out))
"""


def _child(extra_env, timeout=_proc.SUBPROCESS_TIMEOUT):
    env = dict(os.environ)
    env.update(extra_env)
    t0 = time.time()
    p = _proc.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[7:])
    out["seconds"] = time.time() - t0
    return out


def test_capacity_on_the_whole_device():
    import vrpgym_hip as hip
    cap = hip.lib().vrp_persistent_capacity()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    # measured residency (a census launch of the kernel itself) x usable CUs; the B <= 2048
    # regime of the persistent kernel must fit a whole MI355X
    assert 2048 <= cap <= 32 * cus and cap % cus == 0, (cap, cus)
    full = _child({})
    assert full["capacity"] == cap and full["kernel"] == "decode_persistent_kernel"
    assert full["equal"] and not full["nan"] and full["T"][0] == full["T"][1]
    assert full["traces_equal"] and full["env_state_equal"] and full["failures"] == 0, full


def test_cu_mask_falls_back_before_the_episode():
    full_cus = torch.cuda.get_device_properties(0).multi_processor_count
    r = _child({"ROC_GLOBAL_CU_MASK": "0xffffffff"})
    if r["capacity"] >= 2048:
        pytest.skip(f"this runtime ignores ROC_GLOBAL_CU_MASK (capacity {r['capacity']})")
    assert r["capacity"] < 2048, r
    # B = 2048 no longer fits: one launch per step, decided up front -- same results
    assert r["kernel"] != "decode_persistent_kernel", r
    assert r["equal"] and not r["nan"] and r["failures"] == 0, r


@pytest.mark.parametrize("kind,batch,waves", [
    (1, 2048, None), (2, 2048, None),      # one wave per graph, VRP / IRP
    (0, 2046, None),                       # TSP; a batch that is not a multiple of four: the
                                           # fallback walks four graphs at a time, two idle at the end
    (1, 1022, 2), (2, 510, 4),             # the several-waves-per-graph kernels: their own
                                           # state-save code (wave 0) and finalize path
    (0, 510, 4), (0, 766, 4),              # a four-wave TSP grid finalizes itself: the workgroup that
                                           # raised the flag first waits for the others and runs the
                                           # fallback inside the grid (no finalize launch)
])
def test_forced_non_resident_grid_falls_back_transparently(kind, batch, waves):
    """Workgroups forced onto 32 compute units that cannot hold them all: most of the grid waits
    for words of workgroups that have no slot.  The waits give up (20 ms by default; 5 ms here), the grid drains, and the
    finalize kernel reruns the steps: exactly the per-step path's results, in seconds."""
    probe = _child({"ROC_GLOBAL_CU_MASK": "0xffffffff"})
    if probe["capacity"] >= 2048:
        pytest.skip("this runtime ignores ROC_GLOBAL_CU_MASK")
    extra = {"ROC_GLOBAL_CU_MASK": "0xffffffff", "VRP_PERSISTENT_FORCE": "1",
             "VRP_PERSISTENT_LEASE": "0", "GUARD_KIND": str(kind), "GUARD_B": str(batch),
             "VRP_PERSISTENT_SPIN_MS": "5"}
    if waves:
        extra["VRP_PERSISTENT_WAVES"] = str(waves)
    r = _child(extra, timeout=120)
    print("forced non-resident episodes:", r)
    assert r["equal"] and not r["nan"] and r["T"][0] == r["T"][1], r
    assert r["traces_equal"] and r["env_state_equal"], r
    assert r["failures"] >= 1, r      # the first forced episode cannot have been resident
    assert r["seconds"] < 90, r


# Two processes that know nothing of each other on ONE GPU, default settings: sampled rollouts
# and whole training epochs.  Each prints a digest of everything it computed.
PAIR_CHILD = r"""
import hashlib, json, os, sys, time
sys.path[:0] = [os.path.join(%(root)r, "vrp-gym_amd"), %(root)r]
import torch
import agents, vrpgym_hip as hip
from agents import runtime
from gym_vrp.envs import VRPEnv
tag, start_file = sys.argv[1], sys.argv[2]
B, N, ROLLOUTS = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
torch.cuda.set_device(0)
env = VRPEnv(N, B, 1, 17)
agent = agents.VRPAgent(seed=69)
agent.model.eval()
torch.manual_seed(5)
with torch.no_grad():                      # warm-up, then wait for the partner
    runtime.rollout(agent.model, env, False, reset_env=True)
torch.cuda.synchronize()
open(start_file + "." + tag, "w").close()
deadline = time.time() + 120
while not all(os.path.exists(start_file + "." + x) for x in "ab") and time.time() < deadline:
    time.sleep(0.01)
t0 = time.time()
h = hashlib.sha256()
torch.manual_seed(11)
with torch.no_grad():
    for i in range(ROLLOUTS):
        r = runtime.rollout(agent.model, env, False, reset_env=True)
        h.update(r.acc_loss.cpu().numpy().tobytes())
        h.update(r.acc_logp.cpu().numpy().tobytes())
tenv = VRPEnv(num_nodes=N, batch_size=B, num_draw=1, seed=69)
for e in range(5):
    out = agent.train_epoch(tenv, 1)
    h.update(repr([float(x) for x in out]).encode())
for p in agent.model.parameters():
    h.update(p.detach().cpu().numpy().tobytes())
torch.cuda.synchronize()
print("RESULT " + json.dumps({"digest": h.hexdigest(), "seconds": time.time() - t0,
                              "failures": hip.lib().vrp_persistent_failures(),
                              "kernel": hip.lib().vrp_step_kernel_name(1, B, N, 0).decode()}))
"""


def _pair(tmp_path, extra_env, concurrent, shape=(1024, 40, 50)):
    env = dict(os.environ)
    env.update(extra_env)
    script = tmp_path / "pair_child.py"
    script.write_text(PAIR_CHILD % {"root": ROOT})
    import threading
    outs = {}

    def one(tag, start):
        p = _proc.run([sys.executable, str(script), tag, start] + [str(x) for x in shape], env=env,
                      timeout=200)
        outs[tag] = p

    if concurrent:
        start = str(tmp_path / f"go{len(os.listdir(tmp_path))}")
        th = [threading.Thread(target=one, args=(t, start)) for t in "ab"]
        [t.start() for t in th]
        [t.join() for t in th]
    else:
        for t in "ab":
            start = str(tmp_path / f"solo_{t}_{len(os.listdir(tmp_path))}")
            for x in "ab":
                open(start + "." + x, "w").close()   # nobody to wait for
            one(t, start)
    res = {}
    for t, p in outs.items():
        assert p.returncode == 0, (t, p.stderr[-3000:])
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
        res[t] = json.loads(line[7:])
    return res


@pytest.mark.parametrize("lease,shape", [("1", (1024, 40, 50)), ("0", (1024, 40, 50)),
                                         ("0", (2048, 20, 600))])
def test_two_independent_processes_share_one_gpu(tmp_path, lease, shape):
    """50 sampled VRP-40 x 1024 rollouts + 5 REINFORCE epochs in each of two unrelated processes
    on cuda:0 at the same time, default settings.  lease=1: they take turns on the persistent
    grid through the shared lease word.  lease=0: nothing coordinates them -- both launch grids
    sized against the whole device, hand-off waits time out, episodes fall back in-kernel.
    Either way: every number equals the solo run's, and both finish in well under a minute.
    (2048, 20, 600): two one-wave grids of 2048 workgroups do not fit the device together, and 600
    back-to-back episodes per process overlap for certain.)"""
    extra = {"VRP_PERSISTENT_LEASE": lease}
    solo = _pair(tmp_path, extra, concurrent=False, shape=shape)
    assert solo["a"]["digest"] == solo["b"]["digest"]          # deterministic to begin with
    assert solo["a"]["failures"] == 0, solo
    both = _pair(tmp_path, extra, concurrent=True, shape=shape)
    print("solo", solo, "\nconcurrent", both)
    for t in "ab":
        assert both[t]["digest"] == solo["a"]["digest"], (t, both, solo)
        assert both[t]["seconds"] < 60, both
    if lease == "1":
        assert both["a"]["failures"] + both["b"]["failures"] <= 2, both


def test_persistent_rollouts_from_two_streams():
    """runtime.workspaces() gives every (model, env) pair private scratch so that two rollouts
    may be in flight on two streams -- but two persistent grids of one device must not overlap
    (each is sized against the whole device).  The library serialises them (a one-time host
    wait when the second stream shows up, events afterwards): interleaved launches from two
    streams give exactly the results of running them one after the other."""
    import sys
    sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
    from copy import deepcopy
    import agents
    from agents import runtime
    from gym_vrp.envs import VRPEnv
    B, N = 2048, 20
    agent = agents.VRPAgent(seed=69)
    agent.model.eval()
    envs = [VRPEnv(N, B, 1, 3), VRPEnv(N, B, 1, 4)]
    with torch.no_grad():
        want = [runtime.rollout(agent.model, deepcopy(e), True).acc_loss.clone() for e in envs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        got = [[], []]
        for rep in range(4):
            for i, (e, s) in enumerate(zip(envs, streams)):
                with torch.cuda.stream(s):
                    got[i].append(runtime.rollout(agent.model, e, True, reset_env=True).acc_loss)
        torch.cuda.synchronize()
    for i in range(2):
        for g in got[i]:
            assert torch.equal(g, want[i])

"""The persistent multi-step kernel needs its whole grid resident (its waves wait for each
other's mask words).  These tests run the product in child processes whose HIP runtime is
restricted to 32 compute units (ROC_GLOBAL_CU_MASK): eligibility must follow the usable CUs
(fall back to one launch per step BEFORE the episode, same results, no exception), and a grid
that is forced through anyway must fail loudly -- NaN accumulators for every consumer and an
exception when the step count is read -- instead of hanging or returning wrong costs."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path[:0] = [os.path.join(%(root)r, "vrp-gym_amd"), %(root)r]
import torch
import agents, vrpgym_hip as hip
from copy import deepcopy
from agents import runtime
from gym_vrp.envs import VRPEnv
cap = hip.lib().vrp_persistent_capacity()
B, N = 2048, 20
env = VRPEnv(N, B, 1, 13)
agent = agents.VRPAgent(seed=69)
agent.model.eval()
steps = runtime.max_steps_for(1, N)
noise = torch.empty((steps, B, N)).exponential_(1, generator=torch.Generator().manual_seed(2))
out = {"capacity": cap, "kernel": hip.lib().vrp_step_kernel_name(1, B, N, 0).decode()}
res = []
for persistent in (False, True):
    with torch.no_grad():
        r = runtime.rollout(agent.model, deepcopy(env), False, noise=noise, step_trace=True,
                            persistent=persistent)
    torch.cuda.synchronize()
    res.append(r)
out["nan"] = bool(torch.isnan(res[1].acc_loss).any())
try:
    out["T"] = [res[0].T, res[1].T]
    out["raised"] = False
except RuntimeError as e:
    out["raised"] = "timed out" in str(e)
T = min(out.get("T", [1, 1]))
out["equal"] = bool(torch.equal(res[0].acc_loss, res[1].acc_loss)
                    and torch.equal(res[0].acc_logp, res[1].acc_logp)
                    and torch.equal(res[0].actions[:T], res[1].actions[:T]))
print("RESULT " + json.dumps(out))
"""


def _child(extra_env, timeout=600):
    env = dict(os.environ)
    env.update(extra_env)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, timeout=timeout,
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_capacity_on_the_whole_device():
    import vrpgym_hip as hip
    cap = hip.lib().vrp_persistent_capacity()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    # measured residency (a census launch of the kernel itself) x usable CUs; the B <= 2048
    # regime of the persistent kernel must fit a whole MI355X
    assert 2048 <= cap <= 32 * cus and cap % cus == 0, (cap, cus)
    full = _child({})
    assert full["capacity"] == cap and full["kernel"] == "decode_persistent_kernel"
    assert full["equal"] and not full["nan"] and not full["raised"]


def test_cu_mask_falls_back_before_the_episode():
    full_cus = torch.cuda.get_device_properties(0).multi_processor_count
    r = _child({"ROC_GLOBAL_CU_MASK": "0xffffffff"})
    if r["capacity"] >= 2048:
        pytest.skip(f"this runtime ignores ROC_GLOBAL_CU_MASK (capacity {r['capacity']})")
    assert r["capacity"] < 2048, r
    # B = 2048 no longer fits: one launch per step, decided up front -- same results
    assert r["kernel"] != "decode_persistent_kernel", r
    assert r["equal"] and not r["nan"] and not r["raised"], r


def test_forced_non_resident_grid_fails_loudly():
    full_cus = torch.cuda.get_device_properties(0).multi_processor_count
    probe = _child({"ROC_GLOBAL_CU_MASK": "0xffffffff"})
    if probe["capacity"] >= 2048:
        pytest.skip("this runtime ignores ROC_GLOBAL_CU_MASK")
    r = _child({"ROC_GLOBAL_CU_MASK": "0xffffffff", "VRP_PERSISTENT_FORCE": "1"}, timeout=900)
    # either the grid happened to drain (then it must be correct) or it failed loudly
    assert (r["equal"] and not r["nan"]) or (r["nan"] and r["raised"]), r


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

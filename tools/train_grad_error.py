#!/usr/bin/env python3
"""How far fp32 autograd sits from an exact evaluation at the FULL training sizes (BASELINE
configs 3 and 4): the oracle (oracle/policy.py, torch CPU) runs one sampled train-mode rollout in
fp32 with autograd, then the same model in fp64 is teacher-forced along the same actions; both get
the REINFORCE-shaped loss of tests/test_gpu_backward.py::test_full_size_training_step_against_
oracle_autograd ((w * sum log p).mean(), w = linspace(-1, 1, B)).  Written to
tests/golden/train_grad_error.json: max |d sum log p|, |d loss| and, per parameter, the gradient's
relative max-norm / Frobenius error with the test's own normalisation.  The HIP path's bounds in
that test are derived from these figures (2 x the oracle's own fp32 error), not from a guess.
CPU only; a few minutes per case on eight cores.
usage: train_grad_error.py [kind B N ...]   (default: 1 2048 40  2 1024 40)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import envs as oenv  # noqa: E402
from oracle import policy as opol  # noqa: E402


def with_grad(sd, dtype):
    return {k: (v.detach().to(dtype).requires_grad_("running" not in k) if v.is_floating_point() else v.detach().clone())
            for k, v in sd.items()}


def run(kind, B, N):
    t0 = time.time()
    okind = (oenv.TSP, oenv.VRP, oenv.IRP)[kind]
    sd, _ = opol.init_state_dicts(okind, 69)
    wgt = torch.linspace(-1.0, 1.0, B, dtype=torch.float64)
    # fp32: free-running sampled rollout, train mode, autograd
    p32 = with_grad(sd, torch.float32)
    env = oenv.OracleEnv(okind, N, B, 1, 69)
    torch.manual_seed(3)
    trace = []
    _, logp32, T = opol.rollout(p32, env, greedy=False, train=True, trace=trace)
    loss32 = (wgt.float() * logp32).mean()
    loss32.backward()
    acts = torch.stack([t["idx"] for t in trace])
    del trace
    # fp64: the same model, the same inputs, teacher-forced along the same actions
    p64 = with_grad(sd, torch.float64)
    env = oenv.OracleEnv(okind, N, B, 1, 69)
    _, logp64, T2 = opol.rollout(p64, env, greedy=False, train=True, forced=acts,
                                 noise_fn=lambda t, u: torch.ones_like(u))
    assert T2 == T
    loss64 = (wgt * logp64).mean()
    loss64.backward()
    dlogp = (logp32.detach().double() - logp64.detach()).abs().max().item()
    gmax = max(v.grad.abs().max().item() for v in p64.values() if torch.is_tensor(v) and v.grad is not None)
    worst, worst_name, worst_fro, worst_fro_name = 0.0, "", 0.0, ""
    for name, v in p64.items():
        if not torch.is_tensor(v) or v.grad is None:
            continue
        want, got = v.grad, p32[name].grad.double()
        diff = got - want
        rel = diff.abs().max().item() / (want.abs().max().item() + 1e-3 * gmax)
        fro = diff.norm().item() / (want.norm().item() + 1e-3 * gmax * want.numel() ** 0.5)
        if rel > worst:
            worst, worst_name = rel, name
        if fro > worst_fro:
            worst_fro, worst_fro_name = fro, name
    out = {"kind": kind, "B": B, "N": N, "T": T, "max_abs_dlogp": dlogp,
           "abs_dloss": abs(loss32.item() - loss64.item()), "loss": loss64.item(),
           "grad_rel_maxnorm": worst, "grad_rel_maxnorm_param": worst_name,
           "grad_rel_frobenius": worst_fro, "grad_rel_frobenius_param": worst_fro_name,
           "seconds": round(time.time() - t0, 1), "torch_threads": torch.get_num_threads()}
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]] or [1, 2048, 40, 2, 1024, 40]
    cases = [run(*a[i:i + 3]) for i in range(0, len(a), 3)]
    path = os.path.join(ROOT, "tests", "golden", "train_grad_error.json")
    with open(path, "w") as fh:
        json.dump({"what": "fp32 oracle autograd vs the fp64 oracle teacher-forced on the same sampled actions "
                           "(tools/train_grad_error.py); normalisation as in tests/test_gpu_backward.py",
                   "cases": cases}, fh, indent=1)
    print("wrote", path)

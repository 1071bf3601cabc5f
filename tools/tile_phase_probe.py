#!/usr/bin/env python3
"""Phase times of the raw-tile step kernel: the same launch (step t of a prepared episode)
stopped after phase 1 (loads + glimpse scores), 2 (glimpse weights + z), 3 (weight folds),
4 (pointer logits) and run whole (VRP_TILE_DBG, read by the library at every call).
usage: tile_phase_probe.py kind N B [t=3] [sample=0]    (VRP_TILE_V1=1: first-generation kernel)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import vrpgym_hip as hip  # noqa: E402
from agents import runtime  # noqa: E402

kind, N, B = (int(x) for x in sys.argv[1:4])
t = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sample = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
dev = torch.device("cuda", 0)
env, agent = bench.make(kind, N, B, 69, dev)
lib = hip.lib()
model = agent.model
with torch.no_grad():
    res = runtime.rollout(model, env, not sample, tile_kernel=True)
    T = res.T
dw = runtime.decoder_struct(model.decoder)
derived = runtime.decoder_derived(model.decoder, kind)
_, dec_ws = runtime.workspaces(model, env)
io = hip.RolloutIO()
io.acc_loss, io.acc_logp, io.notdone = res.acc_loss.data_ptr(), res.acc_logp.data_ptr(), res.notdone.data_ptr()
noise = None
if sample:
    noise = torch.empty((res.max_steps, B, N), device=dev).exponential_(1)
    io.noise = noise.data_ptr()
stream = hip.current_stream(dev)
flags = (1 if sample else 0) | 4 | 8
# a live episode state at step t: replay the first t steps
bench.rewind(env)
cenv = env._cenv()
hip.check(lib.vrp_env_mask(C.byref(cenv), 0, stream))
hip.check(lib.vrp_decode_prologue(kind, derived.data_ptr(), B, N, res.emb.data_ptr(), dec_ws.data_ptr(), stream))
res.acc_loss.zero_(); res.acc_logp.zero_(); res.notdone.zero_()
for s in range(t):
    hip.check(lib.vrp_decode_step(kind, derived.data_ptr(), C.byref(dw), C.byref(cenv), res.emb.data_ptr(),
                                  dec_ws.data_ptr(), C.byref(io), s, res.max_steps, flags, stream))
    if s == 0:
        hip.check(lib.vrp_decode_first_row(kind, derived.data_ptr(), B, N, res.emb.data_ptr(), dec_ws.data_ptr(), stream))
torch.cuda.synchronize()
name = lib.vrp_step_kernel_name(kind, B, N, flags).decode()
out = {}
for dbg in (1, 2, 3, 4, 0):
    os.environ["VRP_TILE_DBG"] = str(dbg)
    ts = []
    for rep in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        # dbg != 0 leaves the state untouched (the kernel returns before the env step); the whole
        # kernel (dbg 0) advances the episode: re-run step t on whatever state that leaves -- same cost
        hip.check(lib.vrp_decode_step(kind, derived.data_ptr(), C.byref(dw), C.byref(cenv), res.emb.data_ptr(),
                                      dec_ws.data_ptr(), C.byref(io), t if dbg else t + rep % 2, res.max_steps, flags, stream))
        e1.record()
        torch.cuda.synchronize()
        if rep >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    out[dbg] = round(float(np.median(ts)), 2)
print(f"{name} kind={kind} N={N} B={B} t={t} sample={sample}: us through phase "
      f"1 loads {out[1]} | 2 z {out[2]} | 3 folds {out[3]} | 4 logits {out[4]} | whole {out[0]}")

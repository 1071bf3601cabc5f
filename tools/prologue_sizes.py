#!/usr/bin/env python3
"""Times vrp_decode_prologue (the fused projections + tables kernel) at a list of sizes, for
whichever arithmetic the environment selects (default: bf16 planes where the dispatch rule takes
them; VRP_PROLOGUE_FP32=1: the fp32-MFMA instances).  One line per (N, B).
usage: prologue_sizes.py kind B N [N ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import torch  # noqa: E402

import agents  # noqa: E402
import vrpgym_hip as hip  # noqa: E402
from agents import runtime  # noqa: E402

kind, B = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind](seed=69)
agent.model.eval()
lib = hip.lib()
derived = runtime.decoder_derived(agent.model.decoder, kind)
stream = hip.current_stream(dev)
tag = "fp32" if os.environ.get("VRP_PROLOGUE_FP32") else "x3"
for N in (int(x) for x in sys.argv[3:]):
    torch.manual_seed(N)
    emb = torch.randn((B, N, 128), device=dev)
    ws = torch.empty(int(lib.vrp_decoder_workspace_bytes(kind, B, N)), dtype=torch.uint8, device=dev)
    for _ in range(3):
        hip.check(lib.vrp_decode_prologue(kind, derived.data_ptr(), B, N, emb.data_ptr(), ws.data_ptr(), stream))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        hip.check(lib.vrp_decode_prologue(kind, derived.data_ptr(), B, N, emb.data_ptr(), ws.data_ptr(), stream))
    e1.record()
    torch.cuda.synchronize()
    print(f"prologue {tag:4s} kind={kind} B={B} N={N:3d}: {e0.elapsed_time(e1) / reps * 1e3:9.1f} us "
          f"(graph mean + query GEMM + tables)")

#!/usr/bin/env python3
"""Times vrp_gemm_tn (C = X^T Y, the weight-gradient products of the backward pass) on the shapes
of a VRP-40 x 2048 training epoch and checks them against torch fp64.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import torch  # noqa: E402

import vrpgym_hip as hip  # noqa: E402

lib = hip.lib()
st = hip.current_stream()
g = torch.Generator(device="cuda").manual_seed(0)
for R, N1, N2 in [(81920, 128, 128), (81920, 128, 128), (81920, 384, 128), (81920, 128, 512), (81920, 512, 128),
                  (102400, 384, 384), (102400, 128, 384), (40960, 128, 128), (40960, 384, 128),
                  (20480, 512, 128), (5000, 384, 128)]:
    X = torch.randn(R, N1, device="cuda", generator=g)
    Y = torch.randn(R, N2, device="cuda", generator=g)
    ws = torch.empty(int(lib.vrp_gemm_tn_workspace_bytes(R, N1, N2)), dtype=torch.uint8, device="cuda")
    C = torch.empty(N1, N2, device="cuda")

    def run():
        hip.check(lib.vrp_gemm_tn(X.data_ptr(), N1, Y.data_ptr(), N2, C.data_ptr(), R, N1, N2, 0,
                                  ws.data_ptr(), st))
    run()
    torch.cuda.synchronize()
    want = X[:, :64].double().t() @ Y.double()
    err = (C[:64].double() - want).abs().max().item() / max(1.0, want.abs().max().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print(f"R={R:7d} N1={N1:4d} N2={N2:4d}: {us:8.1f} us (product + slab sum) {2*R*N1*N2/us/1e6:7.1f} TFLOP/s  rel err {err:.1e}")

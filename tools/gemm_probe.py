#!/usr/bin/env python3
"""fp32 MFMA GEMM micro-benchmark (HIP events, 20 launches per shape).  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import torch  # noqa: E402

import vrpgym_hip as hip  # noqa: E402

lib = hip.lib()
shapes = [(10240, 384, 128), (10240, 128, 128), (10240, 512, 128), (10240, 128, 512),
          (10240, 1920, 128), (327680, 384, 128), (327680, 128, 128), (327680, 512, 128),
          (327680, 128, 512), (327680, 1920, 128)]
for M, N, K in shapes:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.1
    b = torch.randn(N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    st = hip.current_stream()
    for _ in range(3):
        hip.check(lib.vrp_gemm_nt(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), None, 0,
                                  C.data_ptr(), N, M, N, K, 0, st))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):  # 20 launches replayed without host launch latency
        for _ in range(20):
            hip.check(lib.vrp_gemm_nt(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), None, 0,
                                      C.data_ptr(), N, M, N, K, 0, hip.current_stream()))
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    byts = 4 * (M * K + N * K + M * N)
    print(f"M={M:7d} N={N:5d} K={K:4d}: {us:9.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s  "
          f"{byts/us/1e3:7.1f} GB/s")

#!/usr/bin/env python3
"""fp32 GEMM timings at small/medium M with K = 128 (20 launches replayed from a hipGraph per
shape).  usage: gemm_small_probe.py [M1,M2,...]; VRP_GEMM_VARIANT=64x32 forces the tiled kernel."""
import os, sys
sys.path[:0] = ["/root/repo/vrp-gym_amd", "/root/repo"]
import torch
import vrpgym_hip as hip
lib = hip.lib()
Ms = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (256, 512, 1024, 2048, 4096)
for M in Ms:
    for N in (128, 384, 512, 1536):
        K = 128
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1
        b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda")
        st = hip.current_stream()
        def run():
            hip.check(lib.vrp_gemm_nt(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), None, 0,
                                      C.data_ptr(), N, M, N, K, 0, hip.current_stream()))
        for _ in range(3): run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"M={M} N={N}: {us:.1f} us  {2*M*N*K/us/1e6:.1f} TFLOP/s")

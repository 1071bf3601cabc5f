#!/usr/bin/env python3
"""One GEMM shape a few times (for rocprofv3 --pmc).  usage: gemm_one.py M N K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import torch
import vrpgym_hip as hip
lib = hip.lib()
M, N, K = (int(x) for x in sys.argv[1:4])
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1
b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda")
for _ in range(5):
    hip.check(lib.vrp_gemm_nt(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), None, 0,
                              C.data_ptr(), N, M, N, K, 0, hip.current_stream()))
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Times vrp_gemm_nt on the tall shapes of the training path with the LDS-tiled kernels
(VRP_GEMM_VARIANT=default) and with the persistent 80-row kernel (VRP_GEMM_VARIANT=rows), and checks both against torch fp64.
usage: gemm_rows_probe.py          (spawns itself once per variant; GPU box only)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for v in ("default", "rows"):
        env = dict(os.environ)
        env["VRP_GEMM_VARIANT"] = v
        print("== variant:", v, flush=True)
        subprocess.run([sys.executable, __file__, "run"], env=env, check=False)
    sys.exit(0)

sys.path[:0] = [os.path.join(ROOT, "vrp-gym_amd"), ROOT]
import torch  # noqa: E402

import vrpgym_hip as hip  # noqa: E402

lib = hip.lib()
plain = bool(os.environ.get("GEMM_PROBE_PLAIN"))   # bias only: no residual, no ReLU
shapes = [(204800, 384, 128), (81920, 384, 128), (81920, 128, 128), (81920, 512, 128), (81920, 128, 512),
          (81920, 384, 384), (40960, 384, 128), (40960, 512, 128), (40960, 128, 512),
          (204800, 1536, 128), (327680, 384, 128), (327680, 512, 128), (327680, 128, 512),
          (20479, 128, 128), (30003, 256, 256),
          (50001, 384, 128), (30003, 256, 128), (102400, 384, 128)]
g = torch.Generator(device="cuda").manual_seed(0)
for M, N, K in shapes:
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) * 0.1
    b = torch.randn(N, device="cuda", generator=g)
    R = torch.randn(M, N, device="cuda", generator=g)
    C = torch.empty(M, N, device="cuda")
    st = hip.current_stream()
    Rp, relu = (None, 0) if plain else (R.data_ptr(), 1)
    hip.check(lib.vrp_gemm_nt(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), Rp, N,
                              C.data_ptr(), N, M, N, K, relu, st))
    torch.cuda.synchronize()
    idx = torch.randint(0, M, (2048,), device="cuda")
    def ref(rows):
        v = A[rows].double() @ W.double().t() + b.double()
        return v if plain else torch.relu(v + R[rows].double())
    err = (C[idx].double() - ref(idx)).abs().max().item()
    err = max(err, (C[-1:].double() - ref(slice(M - 1, M))).abs().max().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        hip.check(lib.vrp_gemm_nt(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), Rp, N,
                                  C.data_ptr(), N, M, N, K, relu, st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print(f"M={M:7d} N={N:5d} K={K:4d}: {us:9.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s  max err {err:.2e}")

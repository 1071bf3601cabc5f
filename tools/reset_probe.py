#!/usr/bin/env python3
"""Times env.reset(return_state=False) with the host (numpy-stream) and device (Philox) generators."""
import sys, time
sys.path[:0] = ["/root/repo/vrp-gym_amd", "/root/repo"]
import torch
from gym_vrp.envs import VRPEnv
for gen in ("numpy", "device"):
    env = VRPEnv(40, 8192, 1, 69, generator=gen)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(10):
        env.reset(return_state=False)
    torch.cuda.synchronize()
    print(gen, "reset ms:", (time.time() - t0) / 10 * 1e3)

"""GPU tests of the hand-written backward pass (K4).  Building blocks are compared with
torch (fp64 matmul references / torch autograd of the same fp32 op on the GPU); the
assembled backward is compared with autograd through the CPU oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    import vrpgym_hip as hip
    return hip, hip.lib(), hip.current_stream()


def test_gemm_tn_and_colsum():
    hip, lib, st = _lib()
    g = torch.Generator().manual_seed(3)
    for R, N1, N2 in [(1, 128, 128), (100, 128, 128), (5000, 384, 128), (70001, 1536, 128),
                      (4097, 128, 512)]:
        X = torch.randn(R, N1, generator=g).cuda()
        Y = torch.randn(R, N2, generator=g).cuda()
        ws = torch.empty(int(lib.vrp_gemm_tn_workspace_bytes(R, N1, N2)), dtype=torch.uint8,
                         device="cuda")
        C = torch.full((N1, N2), 7.0, device="cuda")
        hip.check(lib.vrp_gemm_tn(X.data_ptr(), N1, Y.data_ptr(), N2, C.data_ptr(), R, N1, N2, 0,
                                  ws.data_ptr(), st))
        want = X.double().t() @ Y.double()
        scale = max(1.0, want.abs().max().item())
        assert (C.double() - want).abs().max().item() < 2e-5 * scale, (R, N1, N2)
        C2 = C.clone()
        hip.check(lib.vrp_gemm_tn(X.data_ptr(), N1, Y.data_ptr(), N2, C2.data_ptr(), R, N1, N2, 1,
                                  ws.data_ptr(), st))
        assert (C2.double() - 2 * want).abs().max().item() < 4e-5 * scale
        out = torch.full((N1,), 3.0, device="cuda")
        hip.check(lib.vrp_colsum(X.data_ptr(), N1, R, N1, out.data_ptr(), 1, st))
        assert (out.double() - (3.0 + X.double().sum(0))).abs().max().item() < 1e-4 * max(1, R ** 0.5)
        # bitwise reproducible
        C3 = torch.empty_like(C)
        hip.check(lib.vrp_gemm_tn(X.data_ptr(), N1, Y.data_ptr(), N2, C3.data_ptr(), R, N1, N2, 0,
                                  ws.data_ptr(), st))
        assert torch.equal(C3, C)


def test_bn_backward():
    hip, lib, st = _lib()
    g = torch.Generator().manual_seed(4)
    for R in (8, 777, 40000):
        z = (torch.randn(R, 128, generator=g) * 2 + 0.5).cuda().requires_grad_(True)
        gamma = (torch.rand(128, generator=g) + 0.5).cuda().requires_grad_(True)
        beta = torch.randn(128, generator=g).cuda().requires_grad_(True)
        dy = torch.randn(R, 128, generator=g).cuda()
        y = torch.nn.functional.batch_norm(z, None, None, gamma, beta, True, 0.0, 1e-5)
        y.backward(dy)
        mean = z.detach().mean(0)
        invstd = 1.0 / torch.sqrt(z.detach().var(0, unbiased=False) + 1e-5)
        stats = torch.cat([mean, invstd]).contiguous()
        dz = torch.empty(R, 128, device="cuda")
        dgamma = torch.zeros(128, device="cuda")
        dbeta = torch.zeros(128, device="cuda")
        ws = torch.empty(int(lib.vrp_bn_bwd_workspace_bytes()), dtype=torch.uint8, device="cuda")
        hip.check(lib.vrp_bn_bwd(dy.data_ptr(), z.detach().data_ptr(), stats.data_ptr(),
                                 gamma.detach().data_ptr(), R, dz.data_ptr(), dgamma.data_ptr(),
                                 dbeta.data_ptr(), 0, ws.data_ptr(), st))
        tol = 2e-5 * max(1.0, R ** 0.5)
        assert (dz - z.grad).abs().max().item() < 2e-5, R
        assert (dgamma - gamma.grad).abs().max().item() < tol
        assert (dbeta - beta.grad).abs().max().item() < tol


def test_attention_backward():
    hip, lib, st = _lib()
    g = torch.Generator().manual_seed(5)
    for B, N in [(3, 5), (7, 40), (2, 100), (1, 128)]:
        qkv = torch.randn(B * N, 384, generator=g).cuda().requires_grad_(True)
        dO = torch.randn(B * N, 128, generator=g).cuda()
        q, k, v = qkv.view(B, N, 3, 8, 16).permute(2, 0, 3, 1, 4)
        att = torch.softmax(q @ k.transpose(-1, -2) * 0.25, -1)
        o = (att @ v).permute(0, 2, 1, 3).reshape(B * N, 128)
        o.backward(dO)
        dqkv = torch.empty(B * N, 384, device="cuda")
        hip.check(lib.vrp_attention_bwd(qkv.detach().data_ptr(), dO.data_ptr(), dqkv.data_ptr(),
                                        B, N, st))
        assert (dqkv - qkv.grad).abs().max().item() < 5e-5, (B, N)

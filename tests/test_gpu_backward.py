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
        cws = torch.empty(int(lib.vrp_colsum_workspace_bytes(R, N1)), dtype=torch.uint8,
                          device="cuda")
        hip.check(lib.vrp_colsum(X.data_ptr(), N1, R, N1, out.data_ptr(), 1, cws.data_ptr(), st))
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


@pytest.mark.parametrize("kind,B,N", [(0, 7, 9), (1, 16, 20), (2, 5, 33), (1, 40, 40), (0, 5, 33), (2, 16, 20), (2, 5, 32), (1, 5, 33)])
def test_encoder_backward_against_oracle_autograd(kind, B, N):
    """Taped train-mode forward == regular train forward; encoder parameter gradients of
    sum(emb * G) == autograd through the CPU oracle's explicit-math encoder."""
    import agents
    from agents import runtime
    from oracle import policy as opol
    Agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind]
    agent = Agent(seed=69)
    enc = agent.model.encoder
    enc.train()
    g = torch.Generator().manual_seed(B * N + kind)
    x = torch.rand(B, N, 3, generator=g)
    dm = torch.zeros(B, N, dtype=torch.bool)
    dm[torch.arange(B), torch.randint(0, N, (B,), generator=g)] = True
    G = torch.randn(B, N, 128, generator=g)
    # oracle (CPU autograd), once in fp64 (the reference value) and once in fp32: the ReLU
    # gates make the gradient discontinuous, a pre-activation within rounding of zero flips a
    # whole row of ff.0 — the fp32-vs-fp64 gap of the oracle itself measures that noise.
    def oracle_grads(dtype):
        sd, _ = opol.init_state_dicts(kind, 69)
        psd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        psd = {k: v.requires_grad_(v.is_floating_point() and "running" not in k)
               for k, v in psd.items()}
        xin = (x if kind == 2 else x[:, :, :2]).to(dtype)
        o = opol.encoder_forward(psd, xin, None if kind == 0 else dm, train=True)
        (o * G.to(dtype)).sum().backward()
        return o.detach(), psd
    oemb, psd = oracle_grads(torch.float32)
    _, psd64 = oracle_grads(torch.float64)
    # HIP
    x3 = x.clone()
    if kind != 2:
        x3[:, :, 2] = 0
    x3 = x3.cuda().contiguous()
    dmu = None if kind == 0 else dm.to(torch.uint8).cuda().contiguous()
    emb, tape = runtime.encoder_forward_tape(enc, x3, dmu, update_running=True)
    assert (emb.cpu() - oemb.detach()).abs().max().item() < 2e-5
    params, grads = runtime.encoder_backward(enc, x3, dmu, tape, G.cuda())
    names = {id(p): n for n, p in enc.named_parameters()}
    # Biases added right before a train-mode BatchNorm (out_proj.bias, ff.2.bias) and the key
    # bias have an exactly-zero true gradient: both sides only produce rounding noise there,
    # so errors are judged against the tensor's own scale PLUS a floor tied to the typical
    # gradient magnitude of the layer stack.
    wants = [None if p is None else psd64["encoder." + names[id(p)]].grad for p in params]
    noise = [None if p is None else
             (psd["encoder." + names[id(p)]].grad.double() - w).abs().max().item()
             for p, w in zip(params, wants)]
    floor = 1e-5 * float(np.max([w.abs().max().item() for w in wants if w is not None]))
    checked, report = 0, []
    for p, gr, want, nz in zip(params, grads, wants, noise):
        if p is None:
            continue
        err = (gr.cpu().double() - want).abs().max().item()
        tol = 2e-4 * want.abs().max().item() + max(floor, 2e-5) + 4.0 * nz
        report.append((err / tol, names[id(p)], err, want.abs().max().item()))
        checked += 1
    report.sort(reverse=True)
    assert report[0][0] < 1.0, report[:5]
    assert checked == (2 if kind == 0 else 4) + 12 * 3
    nb = enc.attention_layers[0].bn1.norm.num_batches_tracked.item()
    assert nb == 1


def test_encoder_backward_wide_gemm_paths():
    """hidden_dim = 384 at 20800 rows: the feed-forward products take the W-resident tall GEMM
    (gemm_rows_wide_kernel: N = 384, K = 128), forward with ReLU and backward with the ReLU gate;
    in_proj takes it in every train-mode pass of that size.  Against the oracle's autograd."""
    import agents
    from agents import runtime
    from oracle import policy as opol
    B, N = 520, 40
    agent = agents.VRPAgent(seed=69, hidden_dim=384, num_attention_layers=2)
    enc = agent.model.encoder
    enc.train()
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, N, 3, generator=g)
    x[:, :, 2] = 0
    dm = torch.zeros(B, N, dtype=torch.bool)
    dm[torch.arange(B), torch.randint(0, N, (B,), generator=g)] = True
    G = torch.randn(B, N, 128, generator=g)
    sd0 = {k: v.detach().cpu().clone() for k, v in agent.model.state_dict().items()}

    def oracle_grads(dtype):
        psd = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone())
               for k, v in sd0.items()}
        psd = {k: v.requires_grad_(v.is_floating_point() and "running" not in k) for k, v in psd.items()}
        o = opol.encoder_forward(psd, x[:, :, :2].to(dtype), dm, train=True)
        (o * G.to(dtype)).sum().backward()
        return o.detach(), psd
    oemb, psd = oracle_grads(torch.float32)
    _, psd64 = oracle_grads(torch.float64)
    x3 = x.cuda().contiguous()
    dmu = dm.to(torch.uint8).cuda().contiguous()
    emb, tape = runtime.encoder_forward_tape(enc, x3, dmu, update_running=True)
    assert (emb.cpu() - oemb).abs().max().item() < 2e-5
    params, grads = runtime.encoder_backward(enc, x3, dmu, tape, G.cuda())
    names = {id(p): n for n, p in enc.named_parameters()}
    wants = [None if p is None else psd64["encoder." + names[id(p)]].grad for p in params]
    floor = 1e-5 * float(np.max([w.abs().max().item() for w in wants if w is not None]))
    report = []
    for p, gr, want in zip(params, grads, wants):
        if p is None:
            continue
        nz = (psd["encoder." + names[id(p)]].grad.double() - want).abs().max().item()
        err = (gr.cpu().double() - want).abs().max().item()
        tol = 2e-4 * want.abs().max().item() + max(floor, 2e-5) + 4.0 * nz
        report.append((err / tol, names[id(p)], err, want.abs().max().item()))
    report.sort(reverse=True)
    assert report[0][0] < 1.0, report[:5]


def _decoder_oracle_grads(kind, dtype, sd, emb, acts, masks, loads, wgt):
    """Autograd through the CPU oracle's DecoderEpisode, teacher-forced on the recorded
    actions / masks / loads: gradient of sum_b w_b sum_t log p(a_t)."""
    from oracle import policy as opol
    psd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    psd = {k: v.requires_grad_(v.is_floating_point() and k.startswith("decoder."))
           for k, v in psd.items()}
    e = emb.to(dtype).clone().requires_grad_(True)
    ep = opol.DecoderEpisode(psd, e)
    total, lps = 0.0, []
    for t in range(acts.shape[0]):
        load = None if loads is None else loads[t].to(dtype)
        u = ep.logits(masks[t].to(dtype), load)
        lp = (u - u.logsumexp(-1, keepdim=True)).gather(1, acts[t][:, None])[:, 0]
        lps.append(lp.detach())
        total = total + (wgt.to(dtype) * lp).sum()
        ep.advance(acts[t])
    total.backward()
    return psd, e.grad, torch.stack(lps)


@pytest.mark.parametrize("kind,B,N", [(0, 8, 10), (1, 8, 10), (2, 8, 10), (0, 64, 20), (1, 33, 21),
                                      (2, 16, 40), (1, 5, 100), (0, 3, 70), (2, 1, 7)])
def test_decoder_backward_against_oracle_autograd(kind, B, N):
    """vrp_decoder_backward on an episode recorded by the HIP rollout == autograd through
    the oracle decoder (fp64), for parameters and node embeddings; the re-run per-step
    log-probabilities equal the rollout's own."""
    import agents
    from agents import runtime
    from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
    Agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind]
    agent = Agent(seed=69)
    model = agent.model
    model.eval()
    env = (TSPEnv, VRPEnv, IRPEnv)[kind](N, B, 1, 11 + kind)
    torch.manual_seed(5)
    with torch.no_grad():
        res = runtime.rollout(model, env, greedy=False, train=False, trace=True, record=True)
    T = res.T
    acts = res.actions[:T].contiguous()
    masks = res.mask_trace[:T].contiguous()
    loads = None if kind != 2 else res.load_trace[:T].contiguous()
    # recorded masks == the -inf pattern of the recorded logits
    assert torch.equal(masks.bool(), torch.isinf(res.logits[:T]))
    g = torch.Generator().manual_seed(B * N)
    wgt = torch.randn(B, generator=g)
    params, grads, d_emb, step_logp = runtime.decoder_backward(
        model.decoder, kind, res.emb, acts, masks, loads, wgt.cuda(), T, want_logp=True)
    assert (step_logp - res.step_logp[:T]).abs().max().item() < 2e-5
    # (a column sum against the rollout's step-order fp32 accumulation: a few ulp of |sum|,
    # which reaches ~380 at N = 100)
    assert (step_logp.sum(0) - res.acc_logp).abs().max().item() < \
        1e-4 + 1e-6 * res.acc_logp.abs().max().item()

    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    args = (sd, res.emb.cpu(), acts.cpu(), masks.cpu(), None if loads is None else loads.cpu(), wgt)
    p64, e64, lp64 = _decoder_oracle_grads(kind, torch.float64, *args)
    p32, e32, _ = _decoder_oracle_grads(kind, torch.float32, *args)
    assert (step_logp.cpu().double() - lp64).abs().max().item() < 2e-5
    names = {id(p): n for n, p in model.decoder.named_parameters()}
    report = []

    def judge(name, got, want, noise):
        err = (got.double() - want).abs().max().item()
        tol = 2e-4 * want.abs().max().item() + 2e-5 + 4.0 * noise
        report.append((err / tol, name, err, want.abs().max().item()))

    judge("d_emb", d_emb.cpu(), e64, (e32.double() - e64).abs().max().item())
    checked = 0
    for p, gr in zip(params, grads):
        if p is None:
            continue
        key = "decoder." + names[id(p)]
        want = p64[key].grad
        assert want is not None, key
        judge(key, gr.cpu(), want, (p32[key].grad.double() - want).abs().max().item())
        checked += 1
    assert checked == 10
    report.sort(reverse=True)
    assert report[0][0] < 1.0, report[:5]


def test_backward_full_size_properties():
    """BASELINE config 3 shape (VRP N=40, B=2048, sampled): the oracle cannot run this in
    seconds, so the HIP backward is checked through size-independent properties --
    bitwise reproducibility (no float atomics anywhere) and linearity in d_logp."""
    import agents
    from agents import runtime
    from gym_vrp.envs import VRPEnv
    agent = agents.VRPAgent(seed=69)
    model = agent.model
    model.train()
    env = VRPEnv(40, 2048, 1, 69)
    torch.manual_seed(1)
    with torch.no_grad():
        res = runtime.rollout(model, env, greedy=False, train=True, record=True)
    T = res.T
    assert 40 <= T <= 78
    g = torch.Generator().manual_seed(7)
    w1 = torch.randn(2048, generator=g).cuda()
    w2 = torch.randn(2048, generator=g).cuda()

    def grads(w):
        _, dg, d_emb, lp = runtime.decoder_backward(model.decoder, 1, res.emb, res.actions[:T],
                                                    res.mask_trace[:T], None, w, T, want_logp=True)
        _, eg = runtime.encoder_backward(model.encoder, res.x3, res.depot_mask, res.tape, d_emb)
        return [t for t in dg + eg if t is not None] + [d_emb], lp

    ga, lp = grads(w1)
    ga2, _ = grads(w1)
    for x, y in zip(ga, ga2):
        assert torch.equal(x, y)
    # the re-run of the episode reproduces the rollout's accumulated log-probability
    assert (lp.sum(0) - res.acc_logp).abs().max().item() < 2e-4
    gb, _ = grads(w2)
    gs, _ = grads(w1 + w2)
    for x, y, z in zip(ga, gb, gs):
        scale = max(1.0, z.abs().max().item())
        assert (x + y - z).abs().max().item() < 2e-4 * scale


@pytest.mark.parametrize("kind,B,N", [(0, 512, 20), (2, 1024, 40)])
def test_training_step_is_bitwise_reproducible(kind, B, N):
    """Two identical training steps (train-mode rollout with batch-statistics BatchNorm, HIP
    backward) give torch.equal embeddings, BN running statistics and gradients: the
    batch statistics are a fixed-order two-stage reduction, not float atomics.  (2, 1024, 40)
    is BASELINE config 4's per-GPU shard."""
    import agents
    from agents import runtime
    from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
    Env = (TSPEnv, VRPEnv, IRPEnv)[kind]
    Agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind]

    def one():
        agent = Agent(seed=69)
        model = agent.model
        model.train()
        env = Env(N, B, 1, 69)
        noise = torch.empty((runtime.max_steps_for(kind, N), B, N)).exponential_(
            1, generator=torch.Generator().manual_seed(5))
        res = runtime.rollout(model, env, greedy=False, train=True, record=True, noise=noise)
        logp = runtime.attach_grad(model, env, res)
        wgt = torch.linspace(-1.0, 1.0, B, device=logp.device)
        model.zero_grad()
        (wgt * logp).mean().backward()
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        bufs = {n: b.clone() for n, b in model.named_buffers()}
        return res, grads, bufs

    r1, g1, b1 = one()
    r2, g2, b2 = one()
    assert r1.T == r2.T
    assert torch.equal(r1.emb, r2.emb)  # (the tape has uninitialised padding: not compared)
    assert torch.equal(r1.acc_logp, r2.acc_logp) and torch.equal(r1.acc_loss, r2.acc_loss)
    assert torch.equal(r1.actions, r2.actions)
    assert g1.keys() == g2.keys() and len(g1) > 30
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    for n in b1:
        assert torch.equal(b1[n], b2[n]), n


@pytest.mark.parametrize("kind,B,N", [(1, 2048, 40), (2, 1024, 40)])
def test_full_size_training_step_against_oracle_autograd(kind, B, N):
    """BASELINE configs 3 and 4 (per-GPU shard) at FULL size: one sampled train-mode rollout (batch
    statistics BatchNorm, running-stat update) + the HIP backward of a REINFORCE-shaped loss,
    against fp64 autograd through the oracle teacher-forced on the HIP path's own actions and
    masks (what __graft_entry__.smoke() does at B = 16; graph_tsp_agent.py:178-186,
    graph_encoder.py:141-154).  Checked: the accumulated log-probability of every graph, the
    loss scalar, and every parameter's gradient relative to its own scale (+ a floor tied to the
    largest gradient: biases in front of a train-mode BatchNorm have an exactly-zero gradient)."""
    import time
    import agents
    from agents import runtime
    from gym_vrp.envs import IRPEnv, VRPEnv
    from oracle import envs as oenv
    from oracle import policy as opol
    t0 = time.time()
    Agent = (None, agents.VRPAgent, agents.IRPAgent)[kind]
    Env = (None, VRPEnv, IRPEnv)[kind]
    agent = Agent(seed=69)
    model = agent.model
    model.train()
    env = Env(N, B, 1, 69)
    torch.manual_seed(3)
    res = runtime.rollout(model, env, greedy=False, train=True, record=True)
    logp = runtime.attach_grad(model, env, res)
    wgt = torch.linspace(-1.0, 1.0, B, device=logp.device)
    model.zero_grad()
    loss = (wgt * logp).mean()
    loss.backward()
    T = res.T
    sd, _ = opol.init_state_dicts((None, oenv.VRP, oenv.IRP)[kind], 69)
    psd = {k: (v.detach().double().requires_grad_("running" not in k) if v.is_floating_point() else v.detach())
           for k, v in sd.items()}
    x = res.x3.cpu().double()     # [x, y, demand]: the embeddings read the columns they are built for
    dmask = res.depot_mask.cpu().bool() if res.depot_mask is not None else None
    emb = opol.encoder_forward(psd, x, dmask, train=True)
    assert (emb.detach() - res.emb.cpu().double()).abs().max().item() < 2e-4
    ep = opol.DecoderEpisode(psd, emb)
    acts, masks = res.actions[:T].cpu(), res.mask_trace[:T].cpu().double()
    loads = res.load_trace[:T].cpu().double() if res.load_trace is not None else None
    total = torch.zeros(B, dtype=torch.float64)
    for t in range(T):
        u = ep.logits(masks[t], None if loads is None else loads[t])
        total = total + (u - u.logsumexp(-1, keepdim=True)).gather(1, acts[t][:, None])[:, 0]
        ep.advance(acts[t])
    want_loss = (wgt.cpu().double() * total).mean()
    want_loss.backward()
    dlogp = (logp.detach().cpu().double() - total.detach()).abs().max().item()
    dloss = abs(loss.item() - want_loss.item())
    gmax = max(v.grad.abs().max().item() for v in psd.values() if torch.is_tensor(v) and v.grad is not None)
    worst, worst_name, worst_fro, worst_fro_name = 0.0, "", 0.0, ""
    for name, p in model.named_parameters():
        want = psd[name].grad
        if want is None:
            assert p.grad is None, name
            continue
        diff = p.grad.cpu().double() - want
        rel = diff.abs().max().item() / (want.abs().max().item() + 1e-3 * gmax)
        fro = diff.norm().item() / (want.norm().item() + 1e-3 * gmax * want.numel() ** 0.5)
        if rel > worst:
            worst, worst_name = rel, name
        if fro > worst_fro:
            worst_fro, worst_fro_name = fro, name
    print(f"kind {kind} B {B} N {N} T {T}: max |d logp| {dlogp:.2e}, |d loss| {dloss:.2e} "
          f"(loss {want_loss.item():.6f}), worst relative gradient error: max-norm {worst:.2e} "
          f"({worst_name}), Frobenius {worst_fro:.2e} ({worst_fro_name}), {time.time() - t0:.1f} s")
    # Bounds DERIVED from what fp32 autograd itself achieves at this size (round 6):
    # tools/train_grad_error.py runs the oracle's sampled train-mode rollout in fp32 with torch
    # autograd and the same model in fp64 along the same actions, same loss, same normalisation
    # (tests/golden/train_grad_error.json).  VRP 2048 x 40: |d sum log p| 5.1e-5, gradient max-norm
    # 1.13e-2, Frobenius 7.9e-4; IRP 1024 x 40: 4.2e-5, 2.48e-2, 3.8e-3.  The HIP path must stay
    # within TWICE the oracle's own fp32 error of its case -- and never looser than round 5's flat
    # bounds (5e-4, 3e-2, 5e-3).  (A gradient here is an fp32 sum over 41 - 82 thousand rows x up to
    # 78 steps; B = 16, smoke(): 1.7e-4.)
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_grad_error.json")) as fh:
        ref = next(c for c in json.load(fh)["cases"] if (c["kind"], c["B"], c["N"]) == (kind, B, N))
    b_logp = min(5e-4, 2 * ref["max_abs_dlogp"])
    b_max = min(3e-2, 2 * ref["grad_rel_maxnorm"])
    b_fro = min(5e-3, 2 * ref["grad_rel_frobenius"])
    print(f"  bounds (2 x the fp32 oracle's own error): d logp {b_logp:.2e}, max-norm {b_max:.2e}, Frobenius {b_fro:.2e}")
    assert dlogp < b_logp, (dlogp, b_logp)           # a sum of T <= 78 fp32 step log-probabilities
    assert dloss < 1e-5 * max(1.0, abs(want_loss.item())) + 2e-5, dloss
    assert worst < b_max, (worst, worst_name, b_max)
    assert worst_fro < b_fro, (worst_fro, worst_fro_name, b_fro)


def test_step_accounting_does_not_keep_rollouts_alive():
    """runtime.ROLLOUT_LOG (bench.py's step accounting) keeps a rollout's done flags, not the
    RolloutResult: a logged result held the 0.7 GB tape of every training rollout of the timed
    epochs (round 5: eight ranks x 100 epochs ran one GPU out of memory)."""
    import gc
    import agents
    from agents import runtime
    from gym_vrp.envs import IRPEnv
    agent = agents.IRPAgent(seed=69)
    agent.model.train()
    env = IRPEnv(20, 64, 1, 69)
    runtime.ROLLOUT_LOG = log = []
    try:
        res = runtime.rollout(agent.model, env, greedy=False, train=True, record=True)
    finally:
        runtime.ROLLOUT_LOG = None
    assert len(log) == 1 and isinstance(log[0], runtime.RolloutSteps)
    assert not hasattr(log[0], "tape") and not hasattr(log[0], "emb")
    assert res.tape is not None
    T = res.T
    env._last_rollout = None
    del res
    gc.collect()
    assert log[0].T == T     # the flags outlive the result, and they are all the entry holds
    assert all(not isinstance(v, torch.Tensor) or v.numel() <= 4 * 20 for v in vars(log[0]).values())

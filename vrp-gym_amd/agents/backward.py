"""Gradient of the sampled tour's log-probability (REINFORCE, graph_tsp_agent.py:178-186).

INTERIM (round 1): the rollout itself — actions, costs, baseline, BatchNorm running
statistics — comes from the HIP path; only d(sum_t log p(a_t))/d(theta) is obtained by
replaying the recorded actions through a differentiable torch-on-GPU restatement of the
policy (teacher-forced, all T steps batched: the decoder has no recurrent state, a step
depends on earlier ones only through the recorded indices).  It runs on the same device
tensors, never on the CPU.  The hand-written HIP backward (K4 of SURVEY.md 7.1) replaces
this module; the forward value returned to the caller is always the HIP one.
"""
import math

import torch
import torch.nn.functional as F

H_DEC = 8


def _linear(x, w, b=None):
    return F.linear(x, w, b)


def _bn_train(x2d, bn):
    # batch statistics, no running-stat update (the HIP rollout already did it once)
    return F.batch_norm(x2d, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)


def _bn(x, bn, train):
    B, N, E = x.shape
    x2d = x.reshape(B * N, E)
    y = _bn_train(x2d, bn) if train else F.batch_norm(x2d, bn.running_mean, bn.running_var,
                                                     bn.weight, bn.bias, False, 0.0, bn.eps)
    return y.view(B, N, E)


def encoder(enc, x, depot_mask, train):
    """agents/graph_encoder.py:41-58,95-138,183-198 in explicit torch ops."""
    nd = enc.node_embed.weight.shape[1]
    out = _linear(x[..., :nd], enc.node_embed.weight, enc.node_embed.bias)
    dep = getattr(enc, "depot_embed", None)
    if dep is not None and depot_mask is not None:
        dd = dep.weight.shape[1]
        out = torch.where(depot_mask.unsqueeze(-1), _linear(x[..., :dd], dep.weight, dep.bias), out)
    for layer in enc.attention_layers:
        att = layer.attention_layer
        B, N, E = out.shape
        heads = att.num_heads
        hd = E // heads
        qkv = _linear(out, att.in_proj_weight, att.in_proj_bias)
        q, k, v = qkv.split(E, dim=-1)
        q = q.view(B, N, heads, hd).transpose(1, 2)
        k = k.view(B, N, heads, hd).transpose(1, 2)
        v = v.view(B, N, heads, hd).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, N, E)
        o = _linear(o, att.out_proj.weight, att.out_proj.bias)
        y = _bn(out + o, layer.bn1.norm, train)
        f = _linear(torch.relu(_linear(y, layer.ff[0].weight, layer.ff[0].bias)),
                    layer.ff[2].weight, layer.ff[2].bias)
        out = _bn(y + f, layer.bn2.norm, train)
    return out


def step_logits(dec, emb, masks, first_idx, last_idx, loads):
    """All T decoder steps at once.  masks (T,B,N) float 0/1 (the state column fed to
    step t), first_idx/last_idx (T,B) node indices (-1 = learned placeholder),
    loads (T,B) or None.  Returns u (T,B,N) with own-mask -inf (graph_decoder.py:75-98)."""
    T, B, N = masks.shape
    E = emb.shape[-1]
    D, hd = 3 * E, 3 * E // H_DEC
    att = dec.attention
    bq, bk, bv = att.in_proj_bias[:D], att.in_proj_bias[D:2 * D], att.in_proj_bias[2 * D:]
    g = emb.mean(dim=1)                                      # (B,E)
    K = _linear(emb, att.k_proj_weight, bk).view(B, N, H_DEC, hd)
    V = _linear(emb, att.v_proj_weight, bv).view(B, N, H_DEC, hd)
    kp = _linear(emb, dec._kp.weight)                        # (B,N,E)

    def gather(idx, placeholder):
        safe = idx.clamp(min=0)
        rows = emb[torch.arange(B, device=emb.device).unsqueeze(0), safe]      # (T,B,E)
        return torch.where((idx >= 0).unsqueeze(-1), rows, placeholder.view(1, 1, E))

    last = gather(last_idx, dec._last_node)
    gg = g.unsqueeze(0).expand(T, B, E)
    if loads is None:
        first = gather(first_idx, dec._first_node)
        ctx = torch.cat([gg, first, last], -1)               # graph_decoder.py:88
    else:
        ctx = _linear(torch.cat([gg, last, loads.unsqueeze(-1)], -1), dec._context_proj.weight)
    q = _linear(ctx, att.q_proj_weight, bq).view(T, B, H_DEC, hd)
    s = torch.einsum("tbhd,bnhd->tbhn", q, K) / math.sqrt(hd)
    # QUIRK D3: additive float mask, head h of graph b reads row (8b+h) mod B
    scr = (torch.arange(B, device=emb.device).unsqueeze(1) * H_DEC
           + torch.arange(H_DEC, device=emb.device).unsqueeze(0)) % B          # (B,H)
    s = s + masks[:, scr]                                                      # (T,B,H,N)
    a = torch.softmax(s, dim=-1)
    o = torch.einsum("tbhn,bnhd->tbhd", a, V).reshape(T, B, D)
    o = _linear(o, att.out_proj.weight, att.out_proj.bias)
    q2 = _linear(o, dec._att_output.weight)                                    # (T,B,E)
    u = torch.tanh(torch.einsum("tbe,bne->tbn", q2, kp) / math.sqrt(E)) * 10
    return u.masked_fill(masks.bool(), float("-inf"))


def episode_inputs(env, res):
    """Per-step decoder inputs reconstructed from the HIP trace: masks from the -inf
    pattern of the recorded logits, first/last indices from the actions, IRP loads from
    the demands (irp.py:80-86)."""
    T = res.T
    acts = res.actions[:T]                                   # (T,B) int64
    masks = torch.isinf(res.logits[:T]).float()              # (T,B,N)
    B = acts.shape[1]
    neg = torch.full((1, B), -1, dtype=torch.int64, device=acts.device)
    last_idx = torch.cat([neg, acts[:-1]], 0)
    first_idx = torch.cat([neg, acts[0:1].expand(T - 1, B)], 0) if T > 1 else neg
    loads = None
    if env.KIND == 2:
        dem = env._demand                                     # (B,N) f64
        dep = env._depot.long()
        load = torch.ones(B, dtype=torch.float64, device=acts.device)
        ls = []
        for t in range(T):
            ls.append(load.float())
            a = acts[t]
            load = load - dem.gather(1, a[:, None])[:, 0]
            load = torch.where(a == dep, torch.ones_like(load), load)
        loads = torch.stack(ls)
    return acts, masks, first_idx, last_idx, loads


def logp_with_grad(model, env, res):
    """acc_log_prob (B,) whose value is the HIP result and whose gradient flows to the
    model parameters."""
    kind = env.KIND
    dev = res.acc_logp.device
    x = torch.cat([env._pos.float(), env._demand.float().unsqueeze(-1)], -1)   # (B,N,3)
    depot_mask = None
    if kind != 0:
        depot_mask = torch.zeros(x.shape[:2], dtype=torch.bool, device=dev)
        depot_mask[torch.arange(x.shape[0], device=dev), env._depot.long()] = True
    acts, masks, first_idx, last_idx, loads = episode_inputs(env, res)
    with torch.enable_grad():
        emb = encoder(model.encoder, x, depot_mask, model.training)
        u = step_logits(model.decoder, emb, masks, first_idx, last_idx, loads)
        lp = torch.log_softmax(u, dim=-1).gather(2, acts.unsqueeze(-1))[..., 0].sum(0)
    return res.acc_logp + (lp - lp.detach())

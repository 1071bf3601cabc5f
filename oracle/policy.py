"""Oracle policy: explicit-math torch-CPU fp32 restatement of the attention model.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference expresses the
network through ``nn.MultiheadAttention`` / ``nn.BatchNorm1d`` / ``Categorical``;
here every projection, softmax and the glimpse-mask indexing quirk is written
out so that the HIP kernels can be checked stage by stage.  Weights travel as a
plain ``state_dict`` with the reference's key names.
Reference citations are relative to /root/reference.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .envs import IRP, TSP, VRP

H_DEC = 8  # decoder head count is fixed by the caller, agents/graph_tsp_agent.py:53-55


# ---------------------------------------------------------------------------
# A1: weight initialisation in the reference's construction order
# ---------------------------------------------------------------------------
def _encoder_modules(node_dim, emb, hidden, layers, heads, depot_dim=None):
    """Module creation order of GraphEncoder / GraphDemandEncoder
    (agents/graph_encoder.py:26-39, 80-93, 168-181)."""
    mods = OrderedDict()
    mods["node_embed"] = nn.Linear(node_dim, emb)
    for i in range(layers):
        p = f"attention_layers.{i}."
        mods[p + "attention_layer"] = nn.MultiheadAttention(emb, heads, batch_first=True)
        mods[p + "bn1.norm"] = nn.BatchNorm1d(emb)
        mods[p + "bn2.norm"] = nn.BatchNorm1d(emb)
        mods[p + "ff.0"] = nn.Linear(emb, hidden)
        mods[p + "ff.2"] = nn.Linear(hidden, emb)
    if depot_dim is not None:
        mods["depot_embed"] = nn.Linear(depot_dim, emb)
    return mods


def _decoder_modules(emb):
    """agents/graph_decoder.py:29-44."""
    mods = OrderedDict()
    mods["_first_node"] = nn.Parameter(torch.rand(1, 1, emb))
    mods["_last_node"] = nn.Parameter(torch.rand(1, 1, emb))
    mods["attention"] = nn.MultiheadAttention(
        3 * emb, H_DEC, kdim=emb, vdim=emb, batch_first=True
    )
    mods["_kp"] = nn.Linear(emb, emb, bias=False)
    mods["_att_output"] = nn.Linear(3 * emb, emb, bias=False)
    mods["_context_proj"] = nn.Linear(2 * emb + 1, 3 * emb, bias=False)
    return mods


def _flatten(prefix, mods, out):
    for name, m in mods.items():
        if isinstance(m, nn.Parameter):
            out[prefix + name] = m.detach().clone()
        else:
            for k, v in m.state_dict().items():
                out[prefix + name + "." + k] = v.detach().clone()


def _one_model(kind, node_dim, emb, hidden, layers, heads, depot_dim, base_only):
    """One TSPModel/VRPModel/IRPModel worth of RNG draws.

    VRPModel/IRPModel first run TSPModel.__init__ (GraphEncoder + decoder) and
    then *replace* the encoder by a fresh GraphDemandEncoder
    (agents/graph_vrp_agent.py:35-50, graph_irp_agent.py:37-52)."""
    enc = _encoder_modules(node_dim, emb, hidden, layers, heads)
    dec = _decoder_modules(emb)
    if not base_only and kind != TSP:
        enc = _encoder_modules(node_dim, emb, hidden, layers, heads, depot_dim=depot_dim)
    sd = OrderedDict()
    # state_dict order of the reference: encoder.* then decoder.* with the
    # depot_embed entry last inside the encoder block.
    _flatten("encoder.", enc, sd)
    _flatten("decoder.", dec, sd)
    return sd


def init_state_dicts(kind, seed=69, node_dim=None, emb=128, hidden=512, layers=3,
                     heads=8, depot_dim=2):
    """(model_sd, target_sd) exactly as the agent constructors leave them.

    TSPAgent.__init__ agents/graph_tsp_agent.py:124-146; VRPAgent/IRPAgent call it
    first (two throw-away TSPModels consume the torch stream), then build their own
    model pair (graph_vrp_agent.py:118-146, graph_irp_agent.py:140-168).  The
    target model is overwritten with the model's weights.  Also reseeds numpy like
    the reference does."""
    if node_dim is None:
        node_dim = 3 if kind == IRP else 2
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = _one_model(TSP, node_dim, emb, hidden, layers, heads, depot_dim, True)
    _one_model(TSP, node_dim, emb, hidden, layers, heads, depot_dim, True)  # target
    if kind != TSP:
        model = _one_model(kind, node_dim, emb, hidden, layers, heads, depot_dim, False)
        _one_model(kind, node_dim, emb, hidden, layers, heads, depot_dim, False)
    target = OrderedDict((k, v.clone()) for k, v in model.items())
    return model, target


# ---------------------------------------------------------------------------
# N1-N3: encoder
# ---------------------------------------------------------------------------
def _linear(x, w, b=None):
    y = x @ w.t()
    return y if b is None else y + b


def _batchnorm(sd, p, x2d, train):
    """agents/graph_encoder.py:141-154 -> nn.BatchNorm1d(128) over (B*N, 128):
    train = whole-batch statistics (biased var, eps 1e-5) + running-stat update
    (momentum 0.1, unbiased var); eval = running statistics."""
    w, b = sd[p + "weight"], sd[p + "bias"]
    if train:
        n = x2d.shape[0]
        mean = x2d.mean(0)
        var = x2d.var(0, unbiased=False)
        with torch.no_grad():
            sd[p + "running_mean"].mul_(0.9).add_(0.1 * mean)
            sd[p + "running_var"].mul_(0.9).add_(0.1 * var * (n / max(n - 1, 1)))
            sd[p + "num_batches_tracked"] += 1
    else:
        mean, var = sd[p + "running_mean"], sd[p + "running_var"]
    return (x2d - mean) / torch.sqrt(var + 1e-5) * w + b


def _encoder_layer(sd, p, x, heads, train):
    """MultiHeadAttentionLayer.forward agents/graph_encoder.py:183-198."""
    B, N, E = x.shape
    hd = E // heads
    qkv = _linear(x, sd[p + "attention_layer.in_proj_weight"],
                  sd[p + "attention_layer.in_proj_bias"])
    q, k, v = qkv.split(E, dim=-1)
    q = q.view(B, N, heads, hd).transpose(1, 2)
    k = k.view(B, N, heads, hd).transpose(1, 2)
    v = v.view(B, N, heads, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, E)
    o = _linear(o, sd[p + "attention_layer.out_proj.weight"],
                sd[p + "attention_layer.out_proj.bias"])
    y = _batchnorm(sd, p + "bn1.norm.", (x + o).reshape(B * N, E), train).view(B, N, E)
    f = torch.relu(_linear(y, sd[p + "ff.0.weight"], sd[p + "ff.0.bias"]))
    f = _linear(f, sd[p + "ff.2.weight"], sd[p + "ff.2.bias"])
    return _batchnorm(sd, p + "bn2.norm.", (y + f).reshape(B * N, E), train).view(B, N, E)


def encoder_forward(sd, x, depot_mask=None, train=False, heads=8, prefix="encoder."):
    """GraphEncoder.forward agents/graph_encoder.py:41-58 (depot_mask None) or
    GraphDemandEncoder.forward :95-138: the depot row goes through depot_embed
    (x,y only), every other row through node_embed, rows stay in place."""
    w, b = sd[prefix + "node_embed.weight"], sd[prefix + "node_embed.bias"]
    out = _linear(x[..., : w.shape[1]], w, b)
    if depot_mask is not None:
        dw, db = sd[prefix + "depot_embed.weight"], sd[prefix + "depot_embed.bias"]
        dep = _linear(x[..., : dw.shape[1]], dw, db)
        out = torch.where(depot_mask.unsqueeze(-1), dep, out)
    i = 0
    while f"{prefix}attention_layers.{i}.ff.0.weight" in sd:
        out = _encoder_layer(sd, f"{prefix}attention_layers.{i}.", out, heads, train)
        i += 1
    return out


# ---------------------------------------------------------------------------
# D1-D6: decoder
# ---------------------------------------------------------------------------
class DecoderEpisode:
    """Per-episode constants + the first_/last_ state of GraphDecoder
    (agents/graph_decoder.py:46-48,75-83,108-124).  K/V/_kp projections are
    hoisted out of the step: their inputs do not change during an episode."""

    def __init__(self, sd, emb, prefix="decoder."):
        p = prefix
        self.sd, self.p, self.emb = sd, p, emb
        B, N, E = emb.shape
        self.B, self.N, self.E = B, N, E
        D = 3 * E
        bias = sd[p + "attention.in_proj_bias"]
        self.bq, bk, bv = bias[:D], bias[D:2 * D], bias[2 * D:]
        self.graph_emb = emb.mean(dim=1, keepdim=True)  # :75-77
        self.K = _linear(emb, sd[p + "attention.k_proj_weight"], bk)  # (B,N,384)
        self.V = _linear(emb, sd[p + "attention.v_proj_weight"], bv)
        self.kp = _linear(emb, sd[p + "_kp.weight"])  # :83
        self.first = sd[p + "_first_node"].expand(B, 1, E)  # :79-81
        self.last = sd[p + "_last_node"].expand(B, 1, E)
        self.first_step = True
        # head h of graph b reads mask row (b*H + h) mod B  (graph_decoder.py:93:
        # mask.repeat(H,1) lays rows out as h*B+b, torch indexes them as b*H+h)
        b_idx = torch.arange(B).unsqueeze(1) * H_DEC + torch.arange(H_DEC).unsqueeze(0)
        self.scramble = b_idx % B  # (B,H)

    def logits(self, mask, load=None):
        """graph_decoder.py:85-98 -> u (B,N) with own-mask -inf applied.
        `mask` is the float 0/1 state column; it is ADDED to the glimpse scores."""
        sd, p, B, N, E = self.sd, self.p, self.B, self.N, self.E
        D, hd = 3 * E, 3 * E // H_DEC
        if load is None:
            ctx = torch.cat([self.graph_emb, self.first, self.last], -1)  # :88
        else:
            ctx = torch.cat([self.graph_emb, self.last, load[:, None, None]], -1)
            ctx = _linear(ctx, sd[p + "_context_proj.weight"])  # :90-91
        q = _linear(ctx, sd[p + "attention.q_proj_weight"], self.bq)  # (B,1,384)
        qh = q.view(B, H_DEC, hd)
        Kh = self.K.view(B, N, H_DEC, hd).permute(0, 2, 1, 3)  # (B,H,N,hd)
        Vh = self.V.view(B, N, H_DEC, hd).permute(0, 2, 1, 3)
        s = torch.einsum("bhd,bhnd->bhn", qh, Kh) / math.sqrt(hd)
        s = s + mask[self.scramble]  # additive, other graphs' rows (QUIRK D3)
        a = torch.softmax(s, dim=-1)
        o = torch.einsum("bhn,bhnd->bhd", a, Vh).reshape(B, D)
        o = _linear(o, sd[p + "attention.out_proj.weight"], sd[p + "attention.out_proj.bias"])
        q2 = _linear(o, sd[p + "_att_output.weight"])  # (B,128)  :95
        u = torch.tanh(torch.einsum("be,bne->bn", q2, self.kp) / math.sqrt(E)) * 10  # :97
        return u.masked_fill(mask.bool(), float("-inf"))  # :98

    def choose(self, u, greedy, noise=None):
        """graph_decoder.py:100-107.  Sampling = argmax(softmax(u)/q), q~Exp(1)
        (torch.multinomial's single-draw path); `noise` lets a test inject q."""
        if greedy:
            return u.argmax(-1), torch.zeros(self.B, dtype=u.dtype)
        logits = u - u.logsumexp(-1, keepdim=True)
        probs = torch.softmax(logits, dim=-1)
        if noise is None:
            noise = torch.empty_like(probs).exponential_(1)
        idx = (probs / noise).argmax(-1)
        return idx, logits.gather(1, idx[:, None])[:, 0]

    def advance(self, idx):
        """graph_decoder.py:108-113."""
        self.last = self.emb.gather(1, idx[:, None, None].expand(-1, 1, self.E))
        if self.first_step:
            self.first = self.last
            self.first_step = False


# ---------------------------------------------------------------------------
# R1-R3: rollouts
# ---------------------------------------------------------------------------
def split_state(kind, raw, dtype=torch.float):
    """State columns the models read: TSP graph_tsp_agent.py:72-81, VRP
    graph_vrp_agent.py:63-70 (QUIRK: depot_mask := visited column 3),
    IRP graph_irp_agent.py:68-91.  The reference casts the fp64 state to fp32; `dtype`
    (fp64 evaluations of the same model) widens AFTER that rounding, so every precision
    sees the same network inputs."""
    if kind == IRP:
        st = torch.tensor(raw[0], dtype=torch.float).to(dtype)
        load = torch.tensor(raw[1], dtype=torch.float).to(dtype)
        return st[:, :, :3], st[:, :, 3].bool(), st[:, :, -1], load
    st = torch.tensor(raw, dtype=torch.float).to(dtype)
    if kind == VRP:
        return st[:, :, :2], st[:, :, 3].bool(), st[:, :, -1], None
    return st[:, :, :2], None, st[:, :, 3], None


def as_double(sd):
    """fp64 copy of a state dict (integer buffers unchanged)."""
    return OrderedDict((k, v.double() if v.is_floating_point() else v.clone())
                       for k, v in sd.items())


def rollout(sd, env, greedy, train=False, heads=8, noise_fn=None, trace=None,
            forced=None):
    """TSPModel/VRPModel/IRPModel.forward (graph_tsp_agent.py:61-92,
    graph_vrp_agent.py:52-83, graph_irp_agent.py:54-105).
    Returns (acc_loss (B,), acc_log_prob (B,), T).

    `forced` (T,B) replaces the chosen actions (teacher forcing; the log-prob is
    then that of the forced action) so that two implementations can be compared
    past a near-tie where their argmax legitimately differs."""
    kind = env.kind
    # the precision of the weights decides the precision of the evaluation (fp64 state dicts:
    # the "exact" value of the same model on the same fp32 inputs, what fp32 results are
    # measured against in tools/make_golden.py and the train-mode parity tests)
    dtype = sd["decoder._kp.weight"].dtype
    x, depot_mask, mask, load = split_state(kind, env.get_state(), dtype)
    B = x.shape[0]
    acc_loss = torch.zeros(B)
    acc_logp = torch.zeros(B, dtype=dtype)
    emb = encoder_forward(sd, x, depot_mask, train=train, heads=heads)
    ep = DecoderEpisode(sd, emb)
    done, T = False, 0
    while not done:
        u = ep.logits(mask, load)
        noise = None
        if not greedy:
            # same draw (shape, dtype, stream position) as Categorical.sample's
            # multinomial, graph_decoder.py:105-106; kept in the trace for the tests
            noise = noise_fn(T, u) if noise_fn is not None else torch.empty_like(u).exponential_(1)
        idx, logp = ep.choose(u, greedy, noise)
        if forced is not None:
            idx = torch.as_tensor(forced[T], dtype=torch.long)
            if not greedy:
                logp = (u - u.logsumexp(-1, keepdim=True)).gather(1, idx[:, None])[:, 0]
        ep.advance(idx)
        if trace is not None:
            trace.append({"u": u.detach().clone(), "idx": idx.clone(),
                          "logp": logp.detach().clone(), "mask": mask.clone(),
                          "noise": noise})
        _, reward, done, _ = env.step(idx[:, None].numpy())
        if trace is not None:
            trace[-1]["visited_after"] = np.array(env.visited, dtype=np.float64)
        acc_loss = acc_loss + torch.tensor(reward, dtype=torch.float)
        acc_logp = acc_logp + logp
        _, _, mask, load = split_state(kind, env.get_state(), dtype)
        T += 1
    return acc_loss, acc_logp, T


def random_rollout(env):
    """RandomAgent.forward agents/random_agent.py:15-41 (global numpy stream)."""
    state = env.get_state()
    if isinstance(state, tuple):
        state = state[0]
    acc = torch.zeros(state.shape[0])
    done, T = False, 0
    while not done:
        if isinstance(state, tuple):
            state = state[0]
        acts = [np.random.choice(np.argwhere(state[i, :, -1] == 0).flatten(), 1)[0]
                for i in range(state.shape[0])]
        state, loss, done, _ = env.step(np.array(acts)[:, None])
        acc += torch.tensor(loss, dtype=torch.float)
        T += 1
    return acc, T

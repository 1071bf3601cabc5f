"""Host-side plumbing between the torch parameter containers and libvrpgym_hip.so.

Nothing in here computes: it builds the pointer structs of include/vrpgym_hip.h
from the modules' parameters, owns the device scratch buffers, draws the sampling
noise and launches the library on torch's current stream.  There is no CPU path.
"""
import ctypes as C
import os
import weakref

import torch

import vrpgym_hip as hip

EMB, HEADS = 128, 8

_struct_cache = weakref.WeakKeyDictionary()  # module -> (struct, keepalive tensors)
_derived_cache = weakref.WeakKeyDictionary()  # decoder -> (version key, device buffer)
_scratch = weakref.WeakKeyDictionary()  # owner (model / module) -> {tag: scratch tensor}
# encoder -> {"split": tensor, "pad": [_PaddedFF, ...]}: the device-side SHADOWS of an encoder's
# weights (bf16 planes, zero-padded feed-forward copies).  They outlive invalidate(): a captured
# hipGraph has their addresses baked into its kernel arguments, so a rebuilt struct must find
# the same buffers again (refreshed in place), not fresh allocations next to freed ones.
_shadows = weakref.WeakKeyDictionary()
ROLLOUT_LOG = None    # bench.py: a list collecting a RolloutSteps per rollout (step accounting)


def check_supported_dims(emb_dim, num_heads, hidden_dim, decoder=False):
    """The kernels are specialised for the reference's embedding width (128) and, in the decoder,
    its eight heads (fixed by the reference too: graph_tsp_agent.py:53-55).  The encoder runs 8
    heads on every fused kernel and 4 or 16 heads (head width 32 / 8) on a plain GEMM + per-head
    attention path, forward and backward; any `hidden_dim` >= 1 and up to sixteen attention layers
    run (a feed-forward width that is not a multiple of the kernels' 128-wide slices is
    zero-padded to the next one, see `_PaddedFF`; more than eight layers run layer by layer instead
    of through the one-launch stack kernel).  Other sizes can be constructed (state_dict
    compatibility) but not run."""
    if decoder:
        return emb_dim == EMB and num_heads == HEADS
    return emb_dim == EMB and num_heads in (4, 8, 16) and (hidden_dim is None or hidden_dim >= 1)


class _PaddedFF:
    """Feed-forward weights of one encoder layer, zero-padded from `hidden` to the next multiple
    of 128 (graph_encoder.py:177-181 allows any width; the kernels walk the hidden dimension in
    128-wide slices).  Exact, not approximate: a padded hidden unit is relu(0 . y + 0) = 0, adds
    0 . W2 = +0.0 to every output, and receives a zero gradient.  The shadows are refreshed when
    the parameters' version counters move (optimizer step, load_state_dict); the backward
    kernels write padded gradients, of which the real part is copied out."""

    def __init__(self, layer, hp, dev):
        self.ff0, self.ff2 = layer.ff[0], layer.ff[2]
        h = self.ff0.weight.shape[0]
        self.h, self.hp = h, hp
        self.w0 = torch.zeros((hp, EMB), dtype=torch.float32, device=dev)
        self.b0 = torch.zeros((hp,), dtype=torch.float32, device=dev)
        self.w2 = torch.zeros((EMB, hp), dtype=torch.float32, device=dev)
        self.g_w0 = torch.zeros((hp, EMB), dtype=torch.float32, device=dev)
        self.g_b0 = torch.zeros((hp,), dtype=torch.float32, device=dev)
        self.g_w2 = torch.zeros((EMB, hp), dtype=torch.float32, device=dev)
        self.version = None

    def matches(self, layer, hp):
        return (self.ff0 is layer.ff[0] and self.ff2 is layer.ff[2] and self.hp == hp
                and self.h == layer.ff[0].weight.shape[0]
                and self.w0.device == layer.ff[0].weight.device)

    def sync(self):
        ver = (self.ff0.weight._version, self.ff0.bias._version, self.ff2.weight._version,
               self.ff0.weight.data_ptr(), self.ff2.weight.data_ptr())
        if ver != self.version:
            with torch.no_grad():
                self.w0[: self.h].copy_(self.ff0.weight)
                self.b0[: self.h].copy_(self.ff0.bias)
                self.w2[:, : self.h].copy_(self.ff2.weight)
            self.version = ver


class _SplitWeights:
    """The encoder's dense weights as three bf16 planes in MFMA fragment order
    (vrp_encoder_prepare; csrc/encoder_x3.h): what lets the eval-mode kernels run their fp32
    products on the bf16 matrix cores.  One device buffer per encoder, refreshed in place when a
    weight's version counter moves (optimizer step, load_state_dict, the zero-padded feed-forward
    shadows being re-synced), so captured hipGraphs keep pointing at it."""

    def __init__(self, w, mats, dev, reuse=None):
        lib = hip.lib()
        self.mats = mats
        nbytes = int(lib.vrp_encoder_split_bytes(w.hidden, w.num_layers))
        if reuse is not None and reuse.numel() == nbytes and reuse.device == torch.device(dev):
            self.buf = reuse       # same address as before invalidate(): see _shadows
        else:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.version = None        # forces a re-prepare into the (possibly reused) buffer
        w.split = self.buf.data_ptr()

    def sync(self, w):
        ver = tuple((t._version, t.data_ptr()) for t in self.mats)
        if ver != self.version:
            dev = self.buf.device
            hip.check(hip.lib().vrp_encoder_prepare(C.byref(w), self.buf.data_ptr(),
                                                    hip.current_stream(dev)))
            self.version = ver


def invalidate(module):
    """Forget the cached pointer structs of a module (its parameters were re-created or moved).
    The device-side shadows stay (`_shadows`, and `decoder_derived` keeps its buffer through
    `_derived_cache` being refreshed in place): the next `encoder_struct` re-fills them at the
    same addresses, which is what a captured hipGraph replays against."""
    _struct_cache.pop(module, None)
    state = _derived_cache.get(module)
    if state is not None:
        _derived_cache[module] = (None, state[1])   # version key dropped, buffer kept


def _dev(module):
    return next(module.parameters()).device


def _require_cuda(module):
    dev = _dev(module)
    if dev.type != "cuda":
        raise RuntimeError(
            "this model lives on %s: the MI355X-native path has no CPU fallback "
            "(move the model to the GPU; a GPU-less run cannot execute it)" % dev)
    hip.require_gpu()
    return dev


def _buf(owner, tag, device, nbytes):
    """Scratch owned by `owner` (a model, or an (env, model) pair through env._ws): nothing
    is shared between models, so two agents may run on two streams at the same time."""
    bufs = owner if isinstance(owner, dict) else _scratch.setdefault(owner, {})
    key = (tag, str(device))
    t = bufs.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        bufs[key] = t
    return t


def workspaces(model, env):
    """(encoder scratch, decoder per-episode workspace) private to this (model, env) pair:
    the decoder workspace holds the episode's tables, so the model and the baseline model
    (or two envs of one model) must not share it when they run on different streams.  The
    env owns the tensors (they die with it); the model is referenced weakly."""
    lib = hip.lib()
    dev = _dev(model)
    ew = encoder_struct(model.encoder)
    B, N = env.batch_size, env.num_nodes
    per_env = env.__dict__.setdefault("_ws", {})
    key = id(model)
    slot = per_env.get(key)
    if slot is None or slot[0]() is not model:
        if len(per_env) > 8:  # ids of dead models
            for k in [k for k, v in per_env.items() if v[0]() is None]:
                del per_env[k]
        slot = per_env[key] = (weakref.ref(model), {})
    enc = _buf(slot[1], "enc", dev, lib.vrp_encoder_workspace_bytes(B, N, ew.hidden))
    dec = _buf(slot[1], "dec", dev, lib.vrp_decoder_workspace_bytes(env.KIND, B, N))
    return enc, dec


# ------------------------------------------------------------------ weight structs
def encoder_struct(enc):
    hit = _struct_cache.get(enc)
    if hit is not None:
        for pad in hit[2]:
            pad.sync()
        if hit[3] is not None:
            hit[3].sync(hit[0])
        return hit[0]
    node_dim, emb, hidden, heads = enc._dims
    if not check_supported_dims(emb, heads, hidden):
        raise NotImplementedError(
            f"HIP encoder is built for emb_dim=128 and 4, 8 or 16 heads (got {emb}, {heads})")
    hp = (hidden + 127) // 128 * 128
    w = hip.EncoderWeights()
    w.node_dim, w.hidden, w.num_layers = node_dim, hp, len(enc.attention_layers)
    w.heads = heads
    if w.num_layers > hip.MAX_LAYERS:
        raise NotImplementedError(f"at most {hip.MAX_LAYERS} attention layers")
    keep, padded = [], []

    def P(t):
        assert t.dtype in (torch.float32, torch.int64) and t.is_contiguous()
        keep.append(t)
        return t.data_ptr()

    w.node_embed_weight, w.node_embed_bias = P(enc.node_embed.weight), P(enc.node_embed.bias)
    dep = getattr(enc, "depot_embed", None)
    if dep is not None:
        w.depot_dim = dep.weight.shape[1]
        w.depot_embed_weight, w.depot_embed_bias = P(dep.weight), P(dep.bias)
    for i, layer in enumerate(enc.attention_layers):
        L, att = w.layer[i], layer.attention_layer
        L.in_proj_weight, L.in_proj_bias = P(att.in_proj_weight), P(att.in_proj_bias)
        L.out_proj_weight, L.out_proj_bias = P(att.out_proj.weight), P(att.out_proj.bias)
        for tag, bn in (("bn1", layer.bn1.norm), ("bn2", layer.bn2.norm)):
            setattr(L, tag + "_weight", P(bn.weight))
            setattr(L, tag + "_bias", P(bn.bias))
            setattr(L, tag + "_running_mean", P(bn.running_mean))
            setattr(L, tag + "_running_var", P(bn.running_var))
            setattr(L, tag + "_num_batches_tracked", P(bn.num_batches_tracked))
        if hp != hidden:
            old_pads = _shadows.get(enc, {}).get("pad", [])
            pad = None
            if i < len(old_pads) and old_pads[i].matches(layer, hp):
                pad = old_pads[i]          # same shadow tensors as before invalidate()
                pad.version = None
            if pad is None:
                pad = _PaddedFF(layer, hp, layer.ff[0].weight.device)
            pad.sync()
            padded.append(pad)
            L.ff0_weight, L.ff0_bias, L.ff2_weight = P(pad.w0), P(pad.b0), P(pad.w2)
        else:
            L.ff0_weight, L.ff0_bias = P(layer.ff[0].weight), P(layer.ff[0].bias)
            L.ff2_weight = P(layer.ff[2].weight)
        L.ff2_bias = P(layer.ff[2].bias)
    split = None
    if enc.node_embed.weight.is_cuda and heads == HEADS and os.environ.get("VRP_ENCODER_FP32") is None:
        mats = []
        for i, layer in enumerate(enc.attention_layers):
            att = layer.attention_layer
            pad = padded[i] if padded else None
            mats += [att.in_proj_weight, att.out_proj.weight,
                     pad.w0 if pad else layer.ff[0].weight, pad.w2 if pad else layer.ff[2].weight]
        split = _SplitWeights(w, mats, enc.node_embed.weight.device,
                              reuse=_shadows.get(enc, {}).get("split"))
        split.sync(w)
    _shadows[enc] = {"split": split.buf if split is not None else None, "pad": padded}
    _struct_cache[enc] = (w, keep, padded, split)
    return w


def shadow_pointers(enc):
    """Addresses of the encoder's device-side shadows (part of a captured graph's identity)."""
    encoder_struct(enc)
    sh = _shadows.get(enc, {})
    ptrs = [sh["split"].data_ptr() if sh.get("split") is not None else 0]
    for pad in sh.get("pad", []):
        ptrs += [pad.w0.data_ptr(), pad.b0.data_ptr(), pad.w2.data_ptr()]
    return tuple(ptrs)


def padded_ff(enc):
    """The `_PaddedFF` shadows of an encoder whose hidden_dim is not a multiple of 128 ([] else)."""
    encoder_struct(enc)
    return _struct_cache[enc][2]


def decoder_struct(dec):
    hit = _struct_cache.get(dec)
    if hit is not None:
        return hit[0]
    att = dec.attention
    if att.embed_dim != 3 * EMB or att.num_heads != HEADS:
        raise NotImplementedError("HIP decoder is built for emb_dim=128 and 8 heads")
    w = hip.DecoderWeights()
    keep = []

    def P(t):
        assert t.dtype == torch.float32 and t.is_contiguous()
        keep.append(t)
        return t.data_ptr()

    w.first_node, w.last_node = P(dec._first_node), P(dec._last_node)
    w.q_proj_weight, w.k_proj_weight = P(att.q_proj_weight), P(att.k_proj_weight)
    w.v_proj_weight, w.in_proj_bias = P(att.v_proj_weight), P(att.in_proj_bias)
    w.out_proj_weight, w.out_proj_bias = P(att.out_proj.weight), P(att.out_proj.bias)
    w.kp_weight, w.att_output_weight = P(dec._kp.weight), P(dec._att_output.weight)
    w.context_proj_weight = P(dec._context_proj.weight)
    _struct_cache[dec] = (w, keep, [], None)
    return w


def _decoder_version(dec):
    return tuple(p._version for p in dec.parameters())


def decoder_derived(dec, kind):
    """Folded decoder matrices on the device; refreshed when a parameter changed
    (optimizer step, load_state_dict) or the env kind differs."""
    dev = _require_cuda(dec)
    lib = hip.lib()
    state = _derived_cache.get(dec)
    ver = (kind, _decoder_version(dec), str(dev), dec.attention.q_proj_weight.data_ptr())
    if state is None or state[0] != ver:
        if state is not None and state[1].device == dev:
            buf = state[1]  # refresh in place: captured hipGraphs keep pointing at it
        else:
            buf = torch.empty(int(lib.vrp_decoder_derived_bytes()), dtype=torch.uint8,
                              device=dev)
        w = decoder_struct(dec)
        hip.check(lib.vrp_decoder_prepare(kind, C.byref(w), buf.data_ptr(),
                                          hip.current_stream(dev)))
        state = (ver, buf)
        _derived_cache[dec] = state
    return state[1]


# ------------------------------------------------------------------ stand-alone encoder
def encoder_forward(enc, x, depot_mask, train):
    """GraphEncoder/GraphDemandEncoder.forward on arbitrary input tensors."""
    dev = _require_cuda(enc)
    lib = hip.lib()
    w = encoder_struct(enc)
    x = x.detach().to(device=dev, dtype=torch.float32)
    B, N, F = x.shape
    x3 = torch.zeros((B, N, 3), dtype=torch.float32, device=dev)
    x3[:, :, :min(F, 3)] = x[:, :, :3]
    dm = None
    if depot_mask is not None:
        dm = depot_mask.detach().to(device=dev).to(torch.uint8).contiguous()
    emb = torch.empty((B, N, EMB), dtype=torch.float32, device=dev)
    ws = _buf(enc, "enc", dev, lib.vrp_encoder_workspace_bytes(B, N, w.hidden))
    hip.check(lib.vrp_encoder_forward(C.byref(w), int(bool(train)), B, N, x3.data_ptr(),
                                      hip.ptr(dm), emb.data_ptr(), ws.data_ptr(),
                                      hip.current_stream(dev)))
    return emb


# ------------------------------------------------------------------ stand-alone decoder
class _Episode:
    def __init__(self):
        self.t = 0


def decoder_step(dec, node_embs, mask, load, greedy, clip=10.0):
    """GraphDecoder.forward as ONE decode-only kernel launch (no env).  clip: the C of
    `u = C tanh(...)` (graph_decoder.py:56,97; the models never pass anything but 10)."""
    dev = _require_cuda(dec)
    lib = hip.lib()
    kind = hip.KIND_IRP if load is not None else hip.KIND_TSP
    derived = decoder_derived(dec, kind)
    emb = node_embs.detach().to(device=dev, dtype=torch.float32).contiguous()
    B, N, _ = emb.shape
    ep = dec._episode
    stream = hip.current_stream(dev)
    if ep is None:
        ep = _Episode()
        ep.emb = emb
        ep.ws = torch.empty(int(lib.vrp_decoder_workspace_bytes(kind, B, N)), dtype=torch.uint8,
                            device=dev)
        ep.mask = torch.zeros((2, B, N), dtype=torch.uint8, device=dev)
        ep.load = torch.ones((B,), dtype=torch.float64, device=dev)
        hip.check(lib.vrp_decode_prologue(kind, derived.data_ptr(), B, N, emb.data_ptr(),
                                          ep.ws.data_ptr(), stream))
        dec._episode = ep
    if mask is None:
        mask = torch.zeros((B, N))
    ep.mask[ep.t & 1].copy_(mask.detach().to(dev).ne(0).to(torch.uint8))
    if load is not None:
        ep.load.copy_(load.detach().to(dev).to(torch.float64).reshape(B))
    e = hip.Env()
    e.kind, e.B, e.N = kind, B, N
    e.mask, e.load = ep.mask.data_ptr(), ep.load.data_ptr()
    actions = torch.empty((1, B), dtype=torch.int64, device=dev)
    logp = torch.zeros((1, B), dtype=torch.float32, device=dev)
    io = hip.RolloutIO()
    # VRP_STEP_DECODE_ONLY | kernel selection bits (tests: dec.step_flags = 4 raw-tile kernel,
    # 16 large-batch table kernel; default 0 = the dispatch's own choice)
    flags = 2 | (int(getattr(dec, "step_flags", 0)) & (4 | 16 | 64))
    noise = None
    if not greedy:
        noise = torch.empty((B, N)).exponential_(1).to(dev)  # default CPU generator, like
        flags |= 1                                           # Categorical.sample on CPU
    # io arrays are indexed by t inside the kernel: offset the base pointers instead
    t = ep.t
    io.actions = actions.data_ptr() - t * B * 8
    io.step_logp = logp.data_ptr() - t * B * 4
    io.logit_clip = float(clip)
    if noise is not None:
        io.noise = noise.data_ptr() - t * B * N * 4
    w = decoder_struct(dec)
    hip.check(lib.vrp_decode_step(kind, derived.data_ptr(), C.byref(w), C.byref(e),
                                  ep.emb.data_ptr(), ep.ws.data_ptr(), C.byref(io), t, t + 1,
                                  flags, stream))
    ep.t += 1
    lp = torch.zeros(B) if greedy else logp.reshape(B, 1)
    return actions.reshape(B, 1), lp


# ------------------------------------------------------------------ full rollouts
class RolloutResult:
    """Device tensors of one episode.  `T` (number of env steps until the batch-wide
    `done`, tsp.py:95) is read back lazily because it needs a stream sync."""

    def __init__(self, acc_loss, acc_logp, notdone, actions, logits, step_logp, emb, max_steps,
                 mask_trace=None, load_trace=None):
        self.acc_loss, self.acc_logp, self.notdone = acc_loss, acc_logp, notdone
        self.actions, self.logits, self.step_logp, self.emb = actions, logits, step_logp, emb
        self.mask_trace, self.load_trace = mask_trace, load_trace
        self.tape = self.x3 = self.depot_mask = None  # encoder tape of a recording train rollout
        self.max_steps = max_steps
        self._T = None

    @property
    def T(self):
        if self._T is None:
            nd = self.notdone[: self.max_steps].cpu()
            zero = (nd == 0).nonzero()
            self._T = int(zero[0].item()) + 1 if len(zero) else self.max_steps
        return self._T


class RolloutSteps:
    """What the step accounting keeps of a rollout: the done flags only (a logged RolloutResult
    would keep the episode's tape, traces and embeddings alive -- 0.7 GB per training rollout at
    IRP 1024 x 40)."""

    def __init__(self, res):
        self.notdone, self.max_steps, self._T = res.notdone, res.max_steps, res._T

    T = RolloutResult.T


def max_steps_for(kind, N):
    """TSP ends after exactly N-1 steps; VRP/IRP after at most 2(N-1)."""
    return N - 1 if kind == hip.KIND_TSP else 2 * (N - 1)


def host_noise(max_steps, B, N):
    """Parity-mode sampling noise: the reference's Categorical.sample draws
    `empty(B,N).exponential_(1)` from the default CPU generator once per step."""
    return torch.stack([torch.empty((B, N)).exponential_(1) for _ in range(max_steps)])


_graphs = {}  # key -> _CapturedRollout
# hipGraph replay of whole rollouts: opt-in (VRPGYM_GRAPHS=1).  On MI355X the rollout is
# GPU-bound even at B=512 (0.885 ms replayed vs 0.857 ms eager), so eager is the default.
USE_GRAPHS = os.environ.get("VRPGYM_GRAPHS", "0") == "1"


class _CapturedRollout:
    """One hipGraph holding a complete vrp_rollout (mask init, features, encoder,
    prologue, max_steps fused steps): the ~90 launches of a small-batch rollout replay
    without host launch latency.  Everything the graph touches is pointer-stable: env
    tensors, parameters, the in-place derived buffer, private scratch and outputs."""

    def __init__(self, model, env, greedy, train, tile_kernel, dev):
        lib = hip.lib()
        kind, B, N = env.KIND, env.batch_size, env.num_nodes
        self.max_steps = max_steps_for(kind, N)
        ew, dw = encoder_struct(model.encoder), decoder_struct(model.decoder)
        self.derived = decoder_derived(model.decoder, kind)
        self.enc_ws = torch.empty(int(lib.vrp_encoder_workspace_bytes(B, N, ew.hidden)),
                                  dtype=torch.uint8, device=dev)
        self.dec_ws = torch.empty(int(lib.vrp_decoder_workspace_bytes(kind, B, N)),
                                  dtype=torch.uint8, device=dev)
        self.emb = torch.empty((B, N, EMB), dtype=torch.float32, device=dev)
        self.acc_loss = torch.empty((B,), dtype=torch.float32, device=dev)
        self.acc_logp = torch.empty((B,), dtype=torch.float32, device=dev)
        self.notdone = torch.empty((self.max_steps + 1,), dtype=torch.int32, device=dev)
        self.noise = None
        io = hip.RolloutIO()
        io.acc_loss, io.acc_logp = self.acc_loss.data_ptr(), self.acc_logp.data_ptr()
        io.notdone = self.notdone.data_ptr()
        if not greedy:
            self.noise = torch.ones((self.max_steps, B, N), dtype=torch.float32, device=dev)
            io.noise = self.noise.data_ptr()
        self.ptrs = self._pointer_key(model, env)
        cenv = env._cenv()
        flags = int(not greedy) | (4 if tile_kernel else 0)

        def launch():
            hip.check(lib.vrp_rollout(kind, C.byref(ew), C.byref(dw), self.derived.data_ptr(),
                                      C.byref(cenv), int(bool(train)), flags,
                                      self.emb.data_ptr(), self.enc_ws.data_ptr(),
                                      self.dec_ws.data_ptr(), C.byref(io), self.max_steps,
                                      hip.current_stream(dev)))

        # warm-up outside the capture (first-use attribute calls, lazy module load)
        snap = [t.clone() for t in _bn_buffers(model)] if train else None
        vis, cur, load = env._visited.clone(), env._cur.clone(), env._load.clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            launch()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        env._visited.copy_(vis); env._cur.copy_(cur); env._load.copy_(load)
        if snap is not None:  # the warm-up must not count as a training step
            for t, s0 in zip(_bn_buffers(model), snap):
                t.copy_(s0)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            launch()
        # capture only records; state is untouched
        # The choice between the persistent step grid and one launch per step
        # (vrp_persistent_width: residency, failure back-off, cross-process lease) was made ONCE,
        # during the capture, and is frozen into the graph.  The failure counter at that moment:
        # if it has moved by the time of a replay, somebody else is on this device and the frozen
        # grid would pay the bounded spin + in-kernel fallback on every episode -- _graph_rollout
        # then drops the graph and the episode runs eagerly (where the library decides afresh).
        self.failures0 = int(lib.vrp_persistent_failures())

    @staticmethod
    def _pointer_key(model, env):
        # (the shadows keep their addresses across invalidate(); should one ever move -- another
        # device, another hidden width -- the key changes and the rollout is captured again)
        return (env._pos.data_ptr(), env._visited.data_ptr(), env._mask.data_ptr(),
                model.encoder.node_embed.weight.data_ptr(),
                model.decoder.attention.q_proj_weight.data_ptr(),
                decoder_derived(model.decoder, env.KIND).data_ptr()) + shadow_pointers(model.encoder)

    def valid_for(self, model, env):
        return self.ptrs == self._pointer_key(model, env)


def _bn_buffers(model):
    return [b for n, b in model.encoder.named_buffers()]


def _graph_rollout(model, env, greedy, train, tile_kernel, dev):
    key = (id(model), id(env), bool(greedy), bool(train), bool(tile_kernel))
    cap = _graphs.get(key)
    if cap is None:
        # capture on the second sighting only: throw-away envs (deepcopies) stay eager
        seen = env.__dict__.setdefault("_graph_sightings", set())
        if key not in seen:
            seen.add(key)
            return None
    if cap is None or not cap.valid_for(model, env):
        if len(_graphs) > 64:
            _graphs.clear()
        cap = _CapturedRollout(model, env, greedy, train, tile_kernel, dev)
        _graphs[key] = cap
        cap.model_ref, cap.env_ref = weakref.ref(model), weakref.ref(env)
    elif cap.model_ref() is not model or cap.env_ref() is not env:
        _graphs.pop(key)  # id() reuse after garbage collection
        return _graph_rollout(model, env, greedy, train, tile_kernel, dev)
    if int(hip.lib().vrp_persistent_failures()) != cap.failures0:
        _graphs.pop(key, None)   # the device is shared (see _CapturedRollout): eager from here,
        env.__dict__.get("_graph_sightings", set()).discard(key)   # captured again on a later sighting
        return None
    decoder_derived(model.decoder, env.KIND)  # in-place refresh if the weights changed
    encoder_struct(model.encoder)             # likewise the zero-padded feed-forward shadows
    if cap.noise is not None:
        cap.noise.exponential_(1)
    cap.graph.replay()
    res = RolloutResult(cap.acc_loss.clone(), cap.acc_logp.clone(), cap.notdone.clone(), None,
                        None, None, cap.emb, cap.max_steps)
    return res


def rollout(model, env, greedy, train=False, forced=None, noise=None, trace=False,
            noise_mode="device", tile_kernel=False, use_graph=None, record=False,
            throughput_kernel=False, persistent=True, step_trace=False, table_kernel=False,
            reset_env=False):
    """TSPModel/VRPModel/IRPModel.forward: encoder + T x (decode, env.step) on the GPU.
    trace: keep actions, per-step logits and log-probs (the logits trace needs one launch per
    step); step_trace: actions and per-step log-probs only; persistent=False: one launch per
    step even where the persistent multi-step kernel applies (A/B and tests); table_kernel:
    the table-driven step kernel at every step (the raw-tile one takes no first steps);
    reset_env: start a fresh episode on the instances in place (the state part of env.reset():
    visited, current_location, load, step_count) inside the rollout's own set-up kernel."""
    dev = _require_cuda(model)
    if use_graph is None:
        use_graph = USE_GRAPHS
    if (use_graph and forced is None and noise is None and not trace and not record
            and not throughput_kernel and not table_kernel and not reset_env
            and (greedy or noise_mode == "device")):
        if str(env._device) != str(dev):
            raise RuntimeError(f"env is on {env._device} but the model on {dev}")
        env._sync_positions()
        env._parity = 0
        res = _graph_rollout(model, env, greedy, train, tile_kernel, dev)
        if res is not None:
            env._mask_fresh = False
            env._last_rollout = res
            return res
    lib = hip.lib()
    kind = env.KIND
    if str(env._device) != str(dev):
        raise RuntimeError(f"env is on {env._device} but the model on {dev}")
    B, N = env.batch_size, env.num_nodes
    enc, dec = model.encoder, model.decoder
    ew, dw = encoder_struct(enc), decoder_struct(dec)
    derived = decoder_derived(dec, kind)
    max_steps = max_steps_for(kind, N)
    stream = hip.current_stream(dev)

    enc_ws, dec_ws = workspaces(model, env)
    emb = torch.empty((B, N, EMB), dtype=torch.float32, device=dev)
    acc_loss = torch.empty((B,), dtype=torch.float32, device=dev)
    acc_logp = torch.empty((B,), dtype=torch.float32, device=dev)
    notdone = torch.empty((max_steps + 1,), dtype=torch.int32, device=dev)
    io = hip.RolloutIO()
    io.acc_loss, io.acc_logp, io.notdone = acc_loss.data_ptr(), acc_logp.data_ptr(), notdone.data_ptr()
    actions = logits = step_logp = mask_trace = load_trace = None
    if record:
        # what the backward pass needs to re-run the episode: actions, masks, IRP loads
        mask_trace = torch.empty((max_steps, B, N), dtype=torch.uint8, device=dev)
        io.mask_trace = mask_trace.data_ptr()
        if kind == hip.KIND_IRP:
            load_trace = torch.empty((max_steps, B), dtype=torch.float32, device=dev)
            io.load_trace = load_trace.data_ptr()
    if trace or record or step_trace or forced is not None:
        actions = torch.zeros((max_steps, B), dtype=torch.int64, device=dev)
        io.actions = actions.data_ptr()
    if step_trace and not trace:
        step_logp = torch.zeros((max_steps, B), dtype=torch.float32, device=dev)
        io.step_logp = step_logp.data_ptr()
    if trace:
        logits = torch.zeros((max_steps, B, N), dtype=torch.float32, device=dev)
        step_logp = torch.zeros((max_steps, B), dtype=torch.float32, device=dev)
        io.logits, io.step_logp = logits.data_ptr(), step_logp.data_ptr()
    keep = []
    if forced is not None:
        f = torch.zeros((max_steps, B), dtype=torch.int64)
        ft = torch.as_tensor(forced, dtype=torch.int64)
        f[: ft.shape[0]] = ft
        f[ft.shape[0]:] = ft[-1] if ft.shape[0] else 0
        f = f.to(dev)
        keep.append(f)
        io.forced = f.data_ptr()
    gen_state = None
    if not greedy:
        if noise is None and noise_mode != "host":
            # throughput mode: the step kernels draw their Exp(1) noise themselves (Philox,
            # counter = graph / node / step); one seed per rollout from the CPU generator, so
            # torch.manual_seed still makes a run reproducible.  No (max_steps, B, N) tensor.
            # Data-parallel ranks seed torch identically and index their graphs locally: mix
            # the shard's first global graph index into the key, or every rank would explore
            # with the same noise.
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            first = int(getattr(env, "_slice", slice(0, 0)).start or 0)
            seed ^= (first * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF
            io.noise_seed = seed | 1
        else:
            if noise is None:
                gen_state = torch.get_rng_state()
                noise = host_noise(max_steps, B, N)
            noise = torch.as_tensor(noise, dtype=torch.float32).to(dev).contiguous()
            if noise.shape[0] < max_steps:
                pad = torch.ones((max_steps - noise.shape[0], B, N), device=dev)
                noise = torch.cat([noise, pad]).contiguous()
            keep.append(noise)
            io.noise = noise.data_ptr()

    env._sync_positions()
    env._parity = 0
    cenv = env._cenv()
    if reset_env:
        if record and train:
            env._reset_state()          # the taped path runs vrp_env_mask itself
        else:
            cenv.flags = 1              # VRP_ENV_RESET_ON_ROLLOUT
        env._step_count = 0
        env._last_rollout = None
    flags = (int(not greedy) | (4 if tile_kernel else 0) | (16 if throughput_kernel else 0) |
             (0 if persistent else 32) | (64 if table_kernel else 0))
    tape = x3 = dmask = None
    if record and train:
        # the pieces of vrp_rollout with the taped encoder: the backward pass reuses the
        # intermediates instead of re-running the encoder
        hip.check(lib.vrp_env_mask(C.byref(cenv), 0, stream))
        x3 = torch.empty((B, N, 3), dtype=torch.float32, device=dev)
        isd = torch.empty((B, N), dtype=torch.uint8, device=dev)
        hip.check(lib.vrp_env_features(C.byref(cenv), x3.data_ptr(), isd.data_ptr(), stream))
        if kind == hip.KIND_VRP:      # QUIRK graph_vrp_agent.py:67: depot_mask := mask column
            dmask = env._mask[0].clone()
        elif kind == hip.KIND_IRP:    # graph_irp_agent.py:77-79
            dmask = isd
        tape = torch.empty(int(lib.vrp_encoder_tape_bytes(B, N, ew.hidden, ew.num_layers)),
                           dtype=torch.uint8, device=dev)
        hip.check(lib.vrp_encoder_forward_tape(C.byref(ew), B, N, x3.data_ptr(), hip.ptr(dmask),
                                               emb.data_ptr(), tape.data_ptr(), 1, stream))
        hip.check(lib.vrp_decode_prologue(kind, derived.data_ptr(), B, N, emb.data_ptr(),
                                          dec_ws.data_ptr(), stream))
        hip.check(lib.vrp_rollout_steps(kind, derived.data_ptr(), C.byref(dw), C.byref(cenv),
                                        emb.data_ptr(), dec_ws.data_ptr(), C.byref(io),
                                        max_steps, flags, stream))
    else:
        hip.check(lib.vrp_rollout(kind, C.byref(ew), C.byref(dw), derived.data_ptr(),
                                  C.byref(cenv), int(bool(train)), flags, emb.data_ptr(),
                                  enc_ws.data_ptr(), dec_ws.data_ptr(), C.byref(io), max_steps,
                                  stream))
    env._mask_fresh = False  # final mask sits in buffer T&1; recompute lazily into buffer 0
    res = RolloutResult(acc_loss, acc_logp, notdone, actions, logits, step_logp, emb, max_steps,
                        mask_trace, load_trace)
    res._keep = keep
    res.tape, res.x3, res.depot_mask = tape, x3, dmask
    if gen_state is not None:
        # leave the CPU generator where the reference would: it draws only T steps
        T = res.T
        torch.set_rng_state(gen_state)
        host_noise(T, B, N)
    env._last_rollout = res  # env.step_count adds its T lazily
    if ROLLOUT_LOG is not None:
        ROLLOUT_LOG.append(RolloutSteps(res))
    return res


def attach_grad(model, env, res):
    """Differentiable log-probability of the sampled tour (REINFORCE needs
    d sum_t log p(a_t) / d theta): an autograd node over the HIP backward pass
    (agents/backward.py -> vrp_decoder_backward, vrp_encoder_backward)."""
    from . import backward
    return backward.logp_with_grad(model, env, res)


# ------------------------------------------------------------------ backward pass (K4)
def encoder_param_list(enc):
    """Encoder parameters in the order of the vrp_encoder_grads struct."""
    out = [enc.node_embed.weight, enc.node_embed.bias]
    dep = getattr(enc, "depot_embed", None)
    out += [dep.weight, dep.bias] if dep is not None else [None, None]
    for layer in enc.attention_layers:
        att = layer.attention_layer
        out += [att.in_proj_weight, att.in_proj_bias, att.out_proj.weight, att.out_proj.bias,
                layer.bn1.norm.weight, layer.bn1.norm.bias, layer.ff[0].weight, layer.ff[0].bias,
                layer.ff[2].weight, layer.ff[2].bias, layer.bn2.norm.weight, layer.bn2.norm.bias]
    return out


def encoder_forward_tape(enc, x3, depot_mask_u8, update_running):
    """Train-mode encoder forward keeping every intermediate (vrp_encoder_forward_tape)."""
    dev = _require_cuda(enc)
    lib = hip.lib()
    w = encoder_struct(enc)
    B, N, _ = x3.shape
    emb = torch.empty((B, N, EMB), dtype=torch.float32, device=dev)
    tape = torch.empty(int(lib.vrp_encoder_tape_bytes(B, N, w.hidden, w.num_layers)),
                       dtype=torch.uint8, device=dev)
    hip.check(lib.vrp_encoder_forward_tape(C.byref(w), B, N, x3.data_ptr(), hip.ptr(depot_mask_u8),
                                           emb.data_ptr(), tape.data_ptr(),
                                           int(bool(update_running)), hip.current_stream(dev)))
    return emb, tape


def grad_bucket(model, kind):
    """ONE persistent flat fp32 buffer per model holding the gradient of every parameter the
    HIP backward writes (model.parameters() order; decoder._context_proj for TSP/VRP and
    decoder._first_node for IRP never get one and are left out), plus the per-parameter views
    into it.  The backward kernels write straight into the views, `.grad` IS the view, and the
    data-parallel all-reduce runs on the flat buffer in place: no cat, no copy-back
    (one collective on one bucket).

    ALIASING CONTRACT (differs from torch autograd, which allocates fresh gradient tensors):
    the next backward overwrites the bucket in place, so a gradient tensor kept across
    `zero_grad()` (`old = [p.grad for p in ...]`, gradient-surgery helpers) changes with it --
    clone what must survive a step, or set `model.grad_bucket_enabled = False` to get ordinary
    freshly allocated gradients (the all-reduce then packs and unpacks like any DDP bucket)."""
    hit = getattr(model, "_grad_bucket", None)
    dev = _dev(model)
    if hit is not None and hit[0].device == dev and hit[2] == kind:
        return hit
    # exactly the parameters the backward kernels write (the two *_param_list orders are the
    # grads structs'); anything else keeps .grad = None, like torch autograd
    written = {id(p) for p in decoder_param_list(model.decoder, kind) + encoder_param_list(model.encoder)
               if p is not None}
    params = [p for p in model.parameters() if p.requires_grad and id(p) in written]
    flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
    views, off = {}, 0
    for p in params:
        views[id(p)] = flat[off:off + p.numel()].view(p.shape)
        off += p.numel()
    hit = (flat, views, kind)
    model._grad_bucket = hit
    return hit


def encoder_backward(enc, x3, depot_mask_u8, tape, d_emb, out=None):
    """Gradients of every encoder parameter given d_emb; list aligned with encoder_param_list.
    out: {id(param): preallocated gradient tensor} (grad_bucket views) or None."""
    dev = _require_cuda(enc)
    lib = hip.lib()
    w = encoder_struct(enc)
    B, N, _ = x3.shape
    params = encoder_param_list(enc)
    grads = [None if p is None else (out[id(p)] if out is not None else torch.empty_like(p))
             for p in params]
    g = hip.EncoderGrads()
    ptrs = [hip.ptr(t) for t in grads]
    pads = padded_ff(enc)
    for l, pad in enumerate(pads):   # ff.0.weight, ff.0.bias, ff.2.weight: padded gradients
        ptrs[4 + 12 * l + 6], ptrs[4 + 12 * l + 7], ptrs[4 + 12 * l + 8] = (
            pad.g_w0.data_ptr(), pad.g_b0.data_ptr(), pad.g_w2.data_ptr())
    g.node_embed_weight, g.node_embed_bias, g.depot_embed_weight, g.depot_embed_bias = ptrs[:4]
    names = [n for n, _ in hip._lib.EncoderLayerGrads._fields_]
    for l in range(w.num_layers):
        for k, n in enumerate(names):
            setattr(g.layer[l], n, ptrs[4 + 12 * l + k])
    ws = torch.empty(int(lib.vrp_encoder_backward_workspace_bytes(B, N, w.hidden)),
                     dtype=torch.uint8, device=dev)
    d_emb = d_emb.contiguous()
    hip.check(lib.vrp_encoder_backward(C.byref(w), C.byref(g), B, N, x3.data_ptr(),
                                       hip.ptr(depot_mask_u8), tape.data_ptr(), d_emb.data_ptr(),
                                       ws.data_ptr(), hip.current_stream(dev)))
    for l, pad in enumerate(pads):
        grads[4 + 12 * l + 6].copy_(pad.g_w0[: pad.h])
        grads[4 + 12 * l + 7].copy_(pad.g_b0[: pad.h])
        grads[4 + 12 * l + 8].copy_(pad.g_w2[:, : pad.h])
    return params, grads


def decoder_param_list(dec, kind):
    """Decoder parameters in the order of the vrp_decoder_grads struct (None = not used by
    this env kind: no gradient, like torch autograd's None)."""
    att = dec.attention
    irp = kind == hip.KIND_IRP
    return [None if irp else dec._first_node, dec._last_node, att.q_proj_weight, att.k_proj_weight,
            att.v_proj_weight, att.in_proj_bias, att.out_proj.weight, att.out_proj.bias,
            dec._kp.weight, dec._att_output.weight, dec._context_proj.weight if irp else None]


def decoder_backward(dec, kind, emb, actions, masks, loads, d_logp, T, want_logp=False, out=None):
    """vrp_decoder_backward over the first T recorded steps.  Returns (params, grads, d_emb
    [, step_logp]); grads aligned with decoder_param_list.  out: as for encoder_backward."""
    dev = _require_cuda(dec)
    lib = hip.lib()
    w = decoder_struct(dec)
    B, N, _ = emb.shape
    params = decoder_param_list(dec, kind)
    grads = [None if p is None else (out[id(p)] if out is not None else torch.empty_like(p))
             for p in params]
    g = hip.DecoderGrads()
    for (name, _), t in zip(hip.DecoderGrads._fields_, grads):
        setattr(g, name, hip.ptr(t))
    d_emb = torch.empty((B, N, EMB), dtype=torch.float32, device=dev)
    step_logp = torch.empty((T, B), dtype=torch.float32, device=dev) if want_logp else None
    ws = _buf(dec, "dec_bwd", dev, lib.vrp_decoder_backward_workspace_bytes(kind, B, N, T))
    assert actions.dtype == torch.int64 and masks.dtype == torch.uint8
    assert actions.is_contiguous() and masks.is_contiguous() and emb.is_contiguous()
    d_logp = d_logp.to(torch.float32).contiguous()
    hip.check(lib.vrp_decoder_backward(kind, C.byref(w), C.byref(g), B, N, T, emb.data_ptr(),
                                       actions.data_ptr(), masks.data_ptr(), hip.ptr(loads),
                                       d_logp.data_ptr(), d_emb.data_ptr(), hip.ptr(step_logp),
                                       ws.data_ptr(), hip.current_stream(dev)))
    out = (params, grads, d_emb)
    return out + (step_logp,) if want_logp else out

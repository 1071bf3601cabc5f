"""CPU-only tests of the product's host side: instance RNG order, weight containers,
C-ABI exports, sharding and the gradient bucket (gloo, world_size 2)."""
import glob
import hashlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_instances_match_reference_rng_order():
    from gym_vrp.graph.instances import draw_instances
    for f in sorted(glob.glob(os.path.join(G, "instances_*.npz"))):
        z = np.load(f)
        B, N = int(z["B"]), int(z["N"])
        np.random.seed(int(z["seed"]))
        idx = np.random.choice(B, int(z["num_draw"]), replace=False)
        assert np.array_equal(idx, z["draw_idxs"])
        for r in range(3):
            pos, dep, dem = draw_instances(B, N, 1)
            assert np.array_equal(pos, z[f"pos{r}"])
            assert np.array_equal(dep, z[f"depots{r}"])
            assert np.array_equal(dem, z[f"demands{r}"])


def test_native_instance_sampler_is_bit_exact():
    """csrc/instances.hip replays numpy's legacy stream: same arrays AND same generator
    state afterwards as the three-numpy-calls-per-graph loop."""
    from gym_vrp.graph.instances import draw_instances
    import vrpgym_hip
    assert hasattr(vrpgym_hip.lib(), "vrp_draw_instances_host")
    for B, N, seed in [(1, 2, 0), (5, 6, 69), (64, 20, 123), (17, 100, 7), (300, 33, 2 ** 31 - 1)]:
        np.random.seed(seed)
        np.random.rand(seed % 5)  # arbitrary stream position
        want = draw_instances(B, N, native=False)
        tail_want = np.random.rand(3)
        np.random.seed(seed)
        np.random.rand(seed % 5)
        got = draw_instances(B, N, native=True)
        tail_got = np.random.rand(3)
        for a, b in zip(want, got):
            assert np.array_equal(a, b)
        assert np.array_equal(tail_want, tail_got)
    for f in sorted(glob.glob(os.path.join(G, "instances_*.npz"))):  # and the reference's own
        z = np.load(f)
        np.random.seed(int(z["seed"]))
        np.random.choice(int(z["B"]), int(z["num_draw"]), replace=False)
        pos, dep, dem = draw_instances(int(z["B"]), int(z["N"]), native=True)
        assert np.array_equal(pos, z["pos0"]) and np.array_equal(dep, z["depots0"])
        assert np.array_equal(dem, z["demands0"])


def test_native_sampler_keeps_a_shard_and_advances_the_whole_stream():
    """rank r owns rows [rB/R, (r+1)B/R) of the seed-ordered stream.  The range
    sampler stores only those rows, equal to the same rows of the full draw, and leaves the
    generator where the full draw leaves it (so every rank's next reset() stays aligned)."""
    import time
    from gym_vrp.graph.instances import draw_instances, shard_bounds
    for B, N, world, seed in [(8, 5, 2, 1), (64, 20, 8, 69), (96, 41, 3, 5), (16, 100, 16, 9)]:
        np.random.seed(seed)
        full = draw_instances(B, N, native=False)
        tail = np.random.rand(4)
        for r in range(world):
            lo, hi = shard_bounds(B, r, world)
            for native in (True, False):
                np.random.seed(seed)
                part = draw_instances(B, N, native=native, keep=(lo, hi - lo))
                assert np.array_equal(np.random.rand(4), tail)
                for a, b in zip(full, part):
                    assert b.shape[0] == hi - lo and np.array_equal(a[lo:hi], b)
    # per-reset host cost of one rank of an 8-rank IRP-40 x 8192 run (config 4): replaying the
    # other ranks' graphs must stay a small fraction of a training epoch
    np.random.seed(0)
    t0 = time.perf_counter()
    draw_instances(8192, 40, keep=(1024, 1024))
    dt = time.perf_counter() - t0
    print(f"shard of 1024 out of 8192 x 40: {dt * 1e3:.1f} ms per reset per rank")
    assert dt < 0.5


def test_reference_checkpoint_loads():
    """A state_dict with the reference's keys/shapes (here: the oracle's restatement of the
    reference's modules) loads into the product models, and back."""
    import agents
    from oracle import policy as opol
    for kind, cls in enumerate((agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)):
        sd, _ = opol.init_state_dicts(kind, seed=5)
        a = cls(seed=1)
        missing = a.model.load_state_dict(sd, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        for k, v in a.model.state_dict().items():
            assert torch.equal(v.cpu(), sd[k]), k


def test_shard_bounds_partition():
    from gym_vrp.graph.instances import shard_bounds
    spans = [shard_bounds(8192, r, 8) for r in range(8)]
    assert spans[0] == (0, 1024) and spans[-1] == (7168, 8192)
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    with pytest.raises(AssertionError):
        shard_bounds(10, 0, 3)


def test_agent_weight_init_matches_reference():
    """Same construction order => same initial weights as the reference for a seed
    (hashes taken from the reference, tests/golden/weights.npz)."""
    import agents
    z = np.load(os.path.join(G, "weights.npz"))
    for kind, cls in enumerate((agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)):
        a = cls(seed=69)
        sd = a.model.state_dict()
        h = hashlib.sha256()
        for k, v in sd.items():
            h.update(k.encode())
            h.update(v.detach().cpu().contiguous().numpy().tobytes())
        assert h.hexdigest()[:16] == str(z[f"sha_k{kind}"])
        assert list(sd.keys()) == list(z[f"keys_k{kind}"])
        assert [str(tuple(v.shape)) for v in sd.values()] == list(z[f"shapes_k{kind}"])
        assert sum(p.numel() for p in a.model.parameters()) == int(z[f"nparam_k{kind}"])
        assert not a.target_model.training and a.model.training
        for (k1, v1), (k2, v2) in zip(sd.items(), a.target_model.state_dict().items()):
            assert k1 == k2 and torch.equal(v1, v2)


def test_non_default_architecture_weights_and_padded_shadows():
    """Agent kwargs hidden_dim / num_attention_layers (graph_tsp_agent.py:96-106): the constructor
    yields the reference's weights (hash from tests/golden/arch*), and the weight struct handed to
    the kernels carries zero-padded shadows of the feed-forward weights (next multiple of 128)
    that follow the parameters' version counters."""
    import agents
    from agents import runtime
    for path in sorted(glob.glob(os.path.join(G, "archrollout_*.npz"))):
        z = np.load(path)
        kind, hidden, layers = int(z["kind"]), int(z["hidden"]), int(z["layers"])
        cls = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind]
        heads = int(z["heads"]) if "heads" in z.files else 8
        a = cls(seed=69, hidden_dim=hidden, num_attention_layers=layers, num_heads=heads)
        h = hashlib.sha256()
        for k, v in a.model.state_dict().items():
            h.update(k.encode())
            h.update(v.detach().cpu().contiguous().numpy().tobytes())
        assert h.hexdigest()[:16] == str(z["sd_hash"]), path
        enc = a.model.encoder
        w = runtime.encoder_struct(enc)
        hp = (hidden + 127) // 128 * 128
        assert w.hidden == hp and w.num_layers == layers and w.heads == heads
        pads = runtime.padded_ff(enc)
        assert len(pads) == (layers if hp != hidden else 0)
        for layer, pad in zip(enc.attention_layers, pads):
            assert pad.w0.shape == (hp, 128) and pad.w2.shape == (128, hp) and pad.b0.shape == (hp,)
            assert torch.equal(pad.w0[:hidden], layer.ff[0].weight) and not pad.w0[hidden:].any()
            assert torch.equal(pad.w2[:, :hidden], layer.ff[2].weight) and not pad.w2[:, hidden:].any()
            assert torch.equal(pad.b0[:hidden], layer.ff[0].bias) and not pad.b0[hidden:].any()
            with torch.no_grad():
                layer.ff[0].weight.add_(1.0)      # an optimizer step moves the version counter
        runtime.encoder_struct(enc)               # ... and the next use refreshes the shadows
        for layer, pad in zip(enc.attention_layers, pads):
            assert torch.equal(pad.w0[:hidden], layer.ff[0].weight) and not pad.w0[hidden:].any()


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads and exports every function include/vrpgym_hip.h
    declares (no compute call: there is no GPU here)."""
    import vrpgym_hip
    lib = vrpgym_hip.lib()
    header = open(os.path.join(ROOT, "include", "vrpgym_hip.h")).read()
    names = set(re.findall(r"\b(vrp_[a-z_0-9]+)\s*\(", header))
    assert len(names) >= 15
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    # one version in three places: the header's macro, the library, the binding (which refuses
    # to load a library of another ABI)
    want = int(re.search(r"#define\s+VRP_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.vrp_abi_version() == want == vrpgym_hip.ABI_VERSION
    assert len(lib.vrp_source_hash()) == 16
    assert lib.vrp_decoder_derived_bytes() > 0
    assert lib.vrp_encoder_workspace_bytes(512, 20, 512) > 512 * 20 * 128 * 4
    assert lib.vrp_decoder_workspace_bytes(0, 512, 20) > 2 * 512 * 20 * 8 * 20 * 4
    # the structs mirrored in Python have the C sizes
    import ctypes
    assert ctypes.sizeof(vrpgym_hip.Env) == 16 + 7 * 8
    assert ctypes.sizeof(vrpgym_hip.DecoderWeights) == 11 * 8
    assert ctypes.sizeof(vrpgym_hip.RolloutIO) == 12 * 8   # + logit_clip (float, padded)
    assert ctypes.sizeof(vrpgym_hip.DecoderGrads) == 11 * 8
    assert ctypes.sizeof(vrpgym_hip.EncoderWeights) == 24 + 4 * 8 + 16 * 18 * 8 + 8   # + heads, reserved_; 16 layers; split


def test_graft_entry_build():
    """The driver's build entry point on a checkout whose library is already built: make is a
    no-op, the package imports, the ABI matches (round 4 shipped a build() that asserted a stale
    version and nothing ran it)."""
    p = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); print('built ok')"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=1200)
    assert p.returncode == 0 and "built ok" in p.stdout, p.stdout[-3000:]


def test_stale_library_is_refused(tmp_path):
    """A library of another ABI (an old build left in the tree) must not be called into."""
    src = tmp_path / "stale.c"
    src.write_text("int vrp_abi_version(void) { return 5; }\n")
    so = tmp_path / "libstale.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    code = ("import sys; sys.path.insert(0, %r); import vrpgym_hip\n"
            "try:\n    vrpgym_hip.lib()\nexcept RuntimeError as e:\n    print('refused:', e)\n"
            % os.path.join(ROOT, "vrp-gym_amd"))
    env = dict(os.environ, VRPGYM_HIP_LIB=str(so))
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "refused:" in p.stdout and "ABI 5" in p.stdout, p.stdout[-2000:]


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gym_vrp.envs import TSPEnv
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TSPEnv(num_nodes=5, batch_size=2, num_draw=1)
    import agents
    a = agents.TSPAgent()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        a.model.encoder(torch.zeros(2, 5, 2))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vrp-gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "vrp-gym_amd"))
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2],
                        rank=int(sys.argv[3]), world_size=2)
import agents
from agents import distributed as D
rank = dist.get_rank()
a = agents.TSPAgent(seed=69)
torch.manual_seed(100 + rank)
for n, p in a.model.named_parameters():
    if "_context_proj" in n:      # never receives a gradient for TSP (SURVEY 8e)
        continue
    p.grad = torch.randn_like(p)
params = D.grad_parameters(a.model)
assert sum(p.numel() for p in params) == 1154432 - 98688
mine = D.flatten_grads(params).clone()
D.allreduce_gradients(a.model)
got = D.flatten_grads(params)
other = [torch.empty_like(mine) for _ in range(2)]
dist.all_gather(other, mine)
want = (other[0] + other[1]) / 2
assert torch.allclose(got, want, atol=1e-6), (got - want).abs().max()
c, b = D.gather_costs(torch.full((4,), float(rank)), torch.full((4,), 10.0 + rank))
assert c.tolist() == [0.0] * 4 + [1.0] * 4 and b.tolist() == [10.0] * 4 + [11.0] * 4
if rank == 1:
    with torch.no_grad():
        for p in a.model.parameters():
            p.add_(1.0)
ver = [p._version for p in a.model.decoder.parameters()]
D.broadcast_model(a.model)
# the broadcast must be visible to the version-keyed cache of the folded decoder matrices
assert all(p._version > v for p, v in zip(a.model.decoder.parameters(), ver))
m = D.global_means(torch.tensor(float(rank)), torch.tensor(2.0 * rank + 1.0))
assert m == (0.5, 2.0), m
assert D.rank() == rank and D.world_size() == 2
bn = a.model.encoder.attention_layers[0].bn1.norm
bn.running_mean.fill_(float(rank))
bn.num_batches_tracked.fill_(3 + rank)
D.average_buffers(a.model)
assert torch.allclose(bn.running_mean, torch.full_like(bn.running_mean, 0.5))
assert int(bn.num_batches_tracked) == 3
chk = torch.stack([p.detach().sum() for p in a.model.parameters()]).sum()
both = [torch.empty_like(chk) for _ in range(2)]
dist.all_gather(both, chk)
assert torch.equal(both[0], both[1])
dist.destroy_process_group()
print("ok", rank)
"""


def test_gradient_allreduce_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"ok {r}" in o


_SAN_CHILD = r"""
import ctypes as C, os, re, sys
import numpy as np
sys.path[:0] = [os.path.join(%(root)r, "vrp-gym_amd"), %(root)r]
import vrpgym_hip
lib = vrpgym_hip.lib()
assert vrpgym_hip.library_path().endswith("_asan.so")
# (1) every declared symbol, the size functions, the ABI version
header = open(os.path.join(%(root)r, "include", "vrpgym_hip.h")).read()
for n in sorted(set(re.findall(r"\b(vrp_[a-z_0-9]+)\s*\(", header))):
    assert hasattr(lib, n), n
assert lib.vrp_abi_version() == vrpgym_hip.ABI_VERSION and lib.vrp_decoder_derived_bytes() > 0
for B, N in ((1, 2), (512, 20), (8192, 40), (2048, 100), (5, 128)):
    assert lib.vrp_encoder_workspace_bytes(B, N, 512) > 0
    for kind in (0, 1, 2):
        assert lib.vrp_decoder_workspace_bytes(kind, B, N) > 0
    assert lib.vrp_encoder_tape_bytes(B, N, 512, 3) > 0
    assert lib.vrp_decoder_backward_workspace_bytes(1, B, N, 2 * N) > 0
# (2) the native MT19937 instance sampler against numpy's legacy stream, ragged sizes
for seed, B, N in ((69, 7, 9), (5, 33, 2), (123, 64, 100), (1, 1, 128)):
    np.random.seed(seed)
    st = np.random.get_state()
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(st[2]))
    xy = np.empty((B, N, 2)); dep = np.empty((B,), np.int64); dem = np.empty((B, N))
    rc = lib.vrp_draw_instances_host(key.ctypes.data, C.addressof(pos), B, N, xy.ctypes.data,
                                     dep.ctypes.data, dem.ctypes.data)
    assert rc == 0, lib.vrp_last_error()
    for b in range(B):
        want = np.random.rand(N, 2)
        d = np.random.choice(N, size=1, replace=False)
        w = np.random.uniform(1, 10, size=(N, 1)) / (0.2449 * N + 26.12)
        w[d] = 0
        assert np.array_equal(xy[b], want) and dep[b] == d[0] and np.array_equal(dem[b], w[:, 0])
    st2 = np.random.get_state()
    assert np.array_equal(key, st2[1]) and pos.value == st2[2]
# (3) argument checking returns errors (never touches the pointers)
rc = lib.vrp_gemm_nt(None, 128, None, 128, None, None, 0, None, 100, 4, 100, 128, 0, None)
assert rc != 0 and b"multiple" in lib.vrp_last_error()
env = vrpgym_hip.Env(); env.kind, env.B, env.N = 1, 4, 5
io = vrpgym_hip.RolloutIO()
dw = vrpgym_hip.DecoderWeights()
assert lib.vrp_decode_step(0, None, C.byref(dw), C.byref(env), None, None, C.byref(io), 0, 4, 0, None) != 0
assert lib.vrp_rollout_steps_range(1, None, C.byref(dw), C.byref(env), None, None, C.byref(io), 3, 2, 8, 0, None) != 0
assert b"outside" in lib.vrp_last_error()
assert lib.vrp_draw_instances_host(None, None, 1, 1, None, None, None) != 0
# (4) the fold task tables of vrp_decoder_prepare are built on the host before the two
# launches; without a GPU the launch itself fails (and reports), nothing is dereferenced
import torch
if not torch.cuda.is_available():
    fake = 0x10000000
    for name, _ in vrpgym_hip.DecoderWeights._fields_:
        setattr(dw, name, fake)
    for kind in (0, 1, 2):
        rc = lib.vrp_decoder_prepare(kind, C.byref(dw), fake, None)
        assert rc != 0 and b"launch failed" in lib.vrp_last_error(), lib.vrp_last_error()
    assert lib.vrp_decoder_prepare(5, C.byref(dw), fake, None) != 0
print("SANITIZED-OK")
"""


def test_host_side_under_sanitizers():
    """`make asan` builds the host side of every translation unit with
    -fsanitize=address,undefined (CPU build; GPU sanitizers are unavailable on the pool): the
    instrumented library is driven through its host-only entry points -- symbol table, size
    functions, the MT19937 replay, argument checking, the fold task tables -- in a child
    process running under the ASan runtime.  Any report aborts the child."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "vrp-gym_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asan", "-j4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rt = subprocess.run(["hipcc", "-print-file-name=libclang_rt.asan-x86_64.so"],
                        capture_output=True, text=True).stdout.strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ)
    env.update(LD_PRELOAD=rt, VRPGYM_HIP_LIB=os.path.join(ROOT, "vrp-gym_amd", "vrpgym_hip",
                                                         "libvrpgym_hip_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, "-c", _SAN_CHILD % {"root": ROOT}], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "SANITIZED-OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr


def test_gemm_tn_x3_lds_slot_map_is_a_conflict_free_permutation():
    """The LDS image of csrc/backward_blocks.hip:gemm_tn_x3_kernel puts column `col` of a 32-row slab
    at 16-byte slot (col & 3) * 32 + ((col >> 2) + 4 (col & 3)) % 32 of its (plane, 8-row chunk): a
    permutation of the 128 columns in which (a) the store instruction of component c (lane cg writes
    column 4 cg + c) covers 32 CONSECUTIVE slots and (b) the 16 columns an MFMA operand read touches
    (16 t .. 16 t + 15) fall on 16 different 16-byte groups of the 256-byte bank row."""
    slot = lambda col: (col & 3) * 32 + (((col >> 2) + 4 * (col & 3)) & 31)
    assert sorted(slot(c) for c in range(128)) == list(range(128))
    for c in range(4):
        assert sorted(slot(4 * cg + c) for cg in range(32)) == list(range(32 * c, 32 * c + 32))
    for t in range(8):
        assert len({slot(16 * t + i) % 16 for i in range(16)}) == 16


def test_three_bf16_planes_carry_an_fp32_product():
    """The arithmetic of csrc/x3_common.h restated in numpy (no GPU): x = h + m + l with h = bf16(x),
    m = bf16(x - h), l = bf16(x - h - m) (round to nearest even, the subtractions exact), and
    x y ~ mm' + hl' + lh' + hm' + mh' + hh' accumulated in fp32 (small terms first).  Against fp64
    the six-product sum is as accurate as an fp32 fma chain; two planes (three products) are not."""
    import numpy as np

    def bf16(x):  # RNE to bfloat16, returned as float32
        b = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
        b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
        return b.astype(np.uint32).view(np.float32)

    rng = np.random.default_rng(0)
    K = 128
    a = rng.standard_normal((256, K)).astype(np.float32)
    b = rng.standard_normal((256, K)).astype(np.float32)
    ah = bf16(a); am = bf16(a - ah); al = bf16(a - ah - am)
    bh = bf16(b); bm = bf16(b - bh); bl = bf16(b - bh - bm)
    # the split is exact up to 2^-24 |x| (three 8-bit mantissas cover fp32's 24 bits)
    assert np.all(np.abs(a.astype(np.float64) - (ah.astype(np.float64) + am + al)) <= 2.0 ** -23 * np.abs(a))
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(1)

    def acc(terms):  # one MFMA = the 32 products of a k-chunk summed, then ONE fp32 add to the accumulator
        s = np.zeros(256, np.float32)
        for k0 in range(0, K, 32):
            for x, y in terms:   # bf16 x bf16 is exact in fp32; the chunk sum in fp64 stands for the
                d = (x[:, k0:k0 + 32].astype(np.float64) * y[:, k0:k0 + 32]).sum(1)   # pipe's wide adder
                s = (s.astype(np.float64) + d).astype(np.float32)
        return s

    six = acc([(am, bm), (ah, bl), (al, bh), (ah, bm), (am, bh), (ah, bh)])
    three = acc([(ah, bm), (am, bh), (ah, bh)])
    chain = np.zeros(256, np.float32)
    for k in range(K):
        chain = (chain + (a[:, k].astype(np.float64) * b[:, k]).astype(np.float32)).astype(np.float32)
    rms = lambda v: float(np.sqrt(np.mean((v.astype(np.float64) - ref) ** 2)))
    assert rms(six) <= 1.5 * rms(chain)          # fp32 accuracy
    assert rms(three) > 5 * rms(six)             # two planes are not enough


# ---- spill audit (round 6): registers / scratch of the kernels the BASELINE shapes dispatch, read
# from the shipped library's code objects (AMDGPU metadata notes; tools/kernel_resources.py) ----------
# name (demangled, without the argument list) -> largest .vgpr_spill_count this instance may have.
# 0 everywhere except where a comment says why: a spill reload is a scratch (vector-memory) load
# whose s_waitcnt also waits for every prefetched operand and table store issued before it.
HOT_KERNEL_SPILL_BUDGET = {
    # headline: TSP-20 x 512 greedy rollout
    # (four dwords -- two addresses and an index of the final store and of the per-layer BN
    #  constants -- stored once before the layer loop and read back once per layer / at the end:
    #  outside every MFMA stage; the kernel sits at exactly 256 registers with two weight
    #  fragments of 48 live across the attention)
    "encoder_stack_x3_kernel<3>": 4,
    "gemm_nt_m16_k128_kernel": 0,
    # (eight waves per workgroup on one LDS copy of the head's weight fragments: 23 registers over
    #  the 256 a wave may address, and still 7 % faster than four unspilled waves -- DESIGN.md 3.2;
    #  the reloads sit at pack boundaries, not inside stage 1)
    "prologue_tables_kernel<3, true, false, true, false>": 23,
    # (the small-batch instance, which also leaves the glimpse keys in memory: two more)
    "prologue_tables_kernel<3, true, false, true, true>": 25,
    "first_base_kernel": 0,
    "decode_step_rt_kernel<1, 1>": 0,
    "score_base_kernel<2>": 0,
    "decode_persistent4_kernel<4>": 0,
    "persistent_finalize_kernel": 0,
    # north star: TSP / VRP-40 x 8192
    "encoder_qkv_attn8_x3_kernel<5>": 0,
    "encoder_block8_x3_kernel<4>": 0,
    "decode_step_tile_zmfma_kernel<40, 2, false>": 0,
    "decode_step_rt_kernel<1, 4>": 0,
    "score_base_kernel<4>": 0,
    "rollout_setup_kernel": 0,
    "graph_mean_cvec_kernel": 0,
    # config 5: VRP-100 x 2048 sampling
    # (the seven-tile RING instance on bf16 planes: 18 values parked in accumulator registers --
    #  the code object counts them as spills -- and no scratch)
    "prologue_tables_kernel<7, true, true, true, false>": 18,
    "decode_step_tile_zmfma_kernel<100, 1, false>": 0,
    # (19 registers around the attention phase, as the fp32 kernel it replaces (26); the next
    #  graph's prefetched rows additionally wait in scratch across the attention, once per graph)
    "encoder_qkv_attn_graph_x3_kernel<7>": 19,
    "decode_step_rt_kernel<2, 1>": 0,
    # configs 3 / 4: training epochs
    "gemm_rows_x3_kernel<4>": 0,
    "gemm_tn_x3_kernel": 0,
    "decode_persistent_kernel": 0,
}


# decode_persistent4_kernel<4>: no spilled register in the kernel's own code (the episode loop has no
# scratch instruction: checked in the ISA); the private segment is the stack of
# persistent_fallback_call, the out-of-line fallback a self-finalizing TSP grid runs after a failed
# hand-off -- a real call precisely so that its 200 registers stay out of the loop's allocation.
HOT_KERNEL_CALL_STACK = {"decode_persistent4_kernel<4>": 640}


def test_decoder_workspace_keeps_keys_for_small_batches_only():
    """DecWs::KK4 (the glimpse keys the small-batch prologue instances leave for
    persist_first_base, csrc/decoder_ws.h: kk_floats): B x 8 x N x 48 floats for B <= 1024 and
    N <= 63, nothing beyond, nothing with VRP_NO_KEEP_KEYS (host-side layout only: no GPU)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import vrpgym_hip as hip; lib = hip.lib();"
            "print(*[lib.vrp_decoder_workspace_bytes(k, B, N) for k, B, N in "
            "((0, 1024, 20), (0, 1025, 20), (1, 512, 63), (1, 512, 64))])" % os.path.join(ROOT, "vrp-gym_amd"))

    def sizes(extra):
        env = {k: v for k, v in os.environ.items() if k != "VRP_NO_KEEP_KEYS"}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, text=True, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        return [int(x) for x in r.stdout.split()[-4:]]

    keys, none = sizes({}), sizes({"VRP_NO_KEEP_KEYS": "1"})

    def up(x):
        return (x + 255) // 256 * 256
    assert keys[0] - none[0] == up(1024 * 8 * 20 * 48 * 4)      # kept
    assert keys[1] == none[1]                                    # B = 1025: not kept
    assert keys[2] - none[2] == up(512 * 8 * 63 * 48 * 4)        # N = 63: kept
    assert keys[3] == none[3]                                    # N = 64: not kept


def test_hot_kernels_do_not_spill():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    table = {name.split("(")[0].replace("void ", "").strip(): r
             for name, r in kernel_resources.resources().items()}
    missing = [k for k in HOT_KERNEL_SPILL_BUDGET if k not in table]
    assert not missing, f"hot kernels not found in the library (renamed?): {missing}"
    over = {k: (table[k]["vgpr_spill"], table[k]["scratch"]) for k, budget in HOT_KERNEL_SPILL_BUDGET.items()
            if table[k]["vgpr_spill"] > budget}
    assert not over, f"(spilled registers, scratch bytes) above the audited budget: {over}"
    # an instance with no spilled register has no scratch at all -- except the stack of a function
    # it CALLS on a path that never runs in a healthy episode
    for k, budget in HOT_KERNEL_SPILL_BUDGET.items():
        if budget == 0:
            assert table[k]["scratch"] <= HOT_KERNEL_CALL_STACK.get(k, 0), (k, table[k])
    # every kernel fits the register file of its launch bounds (a sanity check of the reader)
    assert all(0 < r["vgpr"] <= 512 and r["agpr"] <= r["vgpr"] for r in table.values())   # (.vgpr_count includes the AGPRs)

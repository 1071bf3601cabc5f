#!/usr/bin/env python3
"""Benchmark of the hot path: batched env.step() + attention-policy rollout.

Metric (BASELINE.json): env-steps/sec in node-steps (batch x nodes x steps / s) on
TSP-20, batch 512 per GPU, greedy rollout of the untrained seed-69 attention agent
(configs[1]).  One bench "step" = one complete rollout of one batch: encoder + T
fused decode/env steps, instances already resident in HBM.  Weak scaling: rank r owns
rows [512 r, 512 (r+1)) of ONE seed-ordered instance stream of 512 x N graphs
(`Env(shard=(rank, world))`); the value is the whole-job aggregate.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

`--gpus N` with N > 1 launches the N ranks itself (child processes with torchrun's environment
variables, started before this process touches the GPU; they die with the launcher) unless it
already runs under torchrun (RANK/WORLD_SIZE set).  Workloads:

    tsp20_b512         BASELINE configs[1], greedy rollout (default: the metric's config)
    tsp40_b8192        north-star shape, greedy rollout
    vrp40_b2048        greedy rollout
    vrp100_b2048       configs[4] per-GPU shard, sampling rollout
    vrp40_b2048_train  configs[2]: REINFORCE epochs (2 sampled rollouts, HIP backward, Adam,
                       2 greedy rollouts + paired t-test), step = one epoch
    irp40_b1024_train  configs[3] per-GPU shard: the same with the RCCL all-reduce of the
                       flat gradient; its time is reported separately

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline            the decode_step kernel of the benched workload vs the HBM peak, plus
                      the loop-level (prologue + table builds + steps) and whole-rollout
                      fractions on the same algorithmic bytes
  roofline_north_star the same at 8192 x 40 TSP (the north-star target shape);
  roofline_north_star_vrp / roofline_cfg5: VRP 8192 x 40 greedy, VRP 2048 x 100 sampling
  other_configs       short runs of the training / sampling configs (ms per step; the training
                      ones carry roofline_train: algorithmic flops per epoch vs the ceiling of the
                      bf16-plane arithmetic, the fp32-MFMA yardstick kept as vs_fp32_mfma_peak)
  cpu_baseline        the CPU oracle (oracle/, a port of the reference) on this host
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "vrp-gym_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import logging  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

logging.getLogger().setLevel(logging.WARNING)  # the agents log every epoch at INFO

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6.3 TB/s achievable

WORKLOADS = {
    # name: (kind, nodes, batch per GPU, mode)
    "tsp20_b512": (0, 20, 512, "greedy"),       # BASELINE configs[1] (the metric's config)
    "tsp40_b8192": (0, 40, 8192, "greedy"),     # north-star shape
    "vrp40_b2048": (1, 40, 2048, "greedy"),
    "vrp100_b2048": (1, 100, 2048, "sample"),   # configs[4] per-GPU shard
    "vrp40_b2048_train": (1, 40, 2048, "train"),  # configs[2]
    "irp40_b1024_train": (2, 40, 1024, "train"),  # configs[3] per-GPU shard
}
KIND_NAMES = "TSP VRP IRP".split()


# ---------------------------------------------------------------------- multi-rank launch
def visible_gpu_count():
    """GPUs this process would see, counted WITHOUT calling into HIP: the KFD topology in sysfs
    lists every agent (GPU nodes have simd_count > 0, CPU nodes 0); *_VISIBLE_DEVICES narrows it.
    Only where sysfs has no KFD topology (not an amdgpu host) does torch get asked."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    n = None
    if nodes:
        n = 0
        for path in nodes:
            try:
                with open(path) as fh:
                    props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
                n += int(props.get("simd_count", "0")) > 0
            except OSError:
                continue
    if n is None:
        return torch.cuda.device_count()
    render = glob.glob("/dev/dri/renderD*")  # a container is handed only its GPUs' render nodes
    if render:
        n = min(n, len(render))
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def _die_with_parent():
    """preexec_fn of a rank: SIGKILL it when this launcher exits, however that happens (a test
    harness that SIGKILLs the launcher on a timeout must not leave ranks on the GPU)."""
    import ctypes
    import signal
    try:
        ctypes.CDLL(None).prctl(1, signal.SIGKILL, 0, 0, 0)   # PR_SET_PDEATHSIG
    except Exception:
        pass


# ---- N > 1: fail fast and loud (first contact with RCCL happens on the driver's 8-GPU box) ------
# Every rank keeps RCCL's own warnings in <faildir>/rccl_rank<r>.log (NCCL_DEBUG=WARN) and, when it
# dies of an exception, leaves <faildir>/rank<r>.json = {"rank", "where", "error", "rccl_tail"}.
# Whoever owns stdout then prints ONE JSON line with "value": null and an "error" field made of
# those records: the launcher below, or rank 0 itself under torchrun.  Nothing hangs longer than
# DIST_TIMEOUT_S in a collective.
DIST_TIMEOUT_S = 120


def fail_dir():
    d = os.environ.get("VRPGYM_BENCH_FAILDIR")
    if not d:
        d = os.path.join("/tmp", "vrpgym_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.getuid()))
        os.environ["VRPGYM_BENCH_FAILDIR"] = d
    os.makedirs(d, exist_ok=True)
    return d


def _tail(path, nbytes=1500):
    try:
        with open(path, "rb") as fh:
            fh.seek(0, 2)
            size = fh.tell()
            fh.seek(max(0, size - nbytes))
            return fh.read().decode("utf-8", "replace")
    except OSError:
        return ""


def record_failure(rank, where, exc):
    """This rank's failure record (never raises)."""
    import traceback
    d = fail_dir()
    rec = {"rank": rank, "where": where,
           "error": "".join(traceback.format_exception_only(type(exc), exc)).strip()[:800],
           "rccl_tail": _tail(os.path.join(d, f"rccl_rank{rank}.log"))}
    try:
        with open(os.path.join(d, f"rank{rank}.json"), "w") as fh:
            json.dump(rec, fh)
    except OSError:
        pass
    return rec


def failure_line(a, world, extra=None):
    """The ONE JSON line of a failed N > 1 run: the contract keys with value null + `error`."""
    import glob
    recs = [extra] if extra else []
    d = os.environ.get("VRPGYM_BENCH_FAILDIR")
    if d:
        for path in sorted(glob.glob(os.path.join(d, "rank*.json"))):
            try:
                with open(path) as fh:
                    r = json.load(fh)
                if not extra or r.get("rank") != extra.get("rank"):
                    recs.append(r)
            except (OSError, ValueError):
                continue
        seen = {r.get("rank") for r in recs if r}
        for path in sorted(glob.glob(os.path.join(d, "rccl_rank*.log"))):   # warnings of ranks that did not raise
            r = int(os.path.basename(path)[len("rccl_rank"):-4])
            t = _tail(path, 600)
            if r not in seen and t.strip():
                recs.append({"rank": r, "where": "rccl log", "error": "", "rccl_tail": t})
    recs = [r for r in recs if r]
    msg = "; ".join(f"rank {r['rank']} [{r['where']}]: {r['error']}" +
                    (f" | RCCL: {r['rccl_tail'][-500:].strip()}" if r.get("rccl_tail", "").strip() else "")
                    for r in recs) or "a rank exited without leaving a record"
    kind, N, B, mode = WORKLOADS[a.workload]
    return json.dumps({"metric": "env-steps/sec (batch x nodes), " +
                       ("REINFORCE training" if mode == "train" else f"{mode} attention rollout"),
                       "value": None, "unit": "node-steps/s", "n_gpus": world, "steps": a.steps,
                       "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
                       "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                       "config": {"workload": a.workload, "name": a.workload},
                       "error": msg[:4000], "failed_ranks": sorted({r["rank"] for r in recs})})


def spawn_ranks(a, argv):
    """--gpus N outside torchrun: start the N ranks as CHILD processes (env-var rendezvous:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, what torchrun would set) and
    relay the first failure.  Nothing in this process touches the GPU runtime: the GPUs are
    counted in sysfs.  The ranks stay in this process's group and session and carry
    PR_SET_PDEATHSIG: SIGTERM / SIGINT are forwarded, the first rank to fail takes the others
    down, and no rank survives the launcher."""
    import signal
    ndev = visible_gpu_count()
    env = dict(os.environ)
    if ndev < a.gpus and env.get("VRPGYM_BENCH_ONE_GPU") != "1":
        sys.exit(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) visible "
                 "(VRPGYM_BENCH_ONE_GPU=1 maps every rank to cuda:0 over gloo: a test aid, "
                 "flagged in the JSON line)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(a.gpus),
               LOCAL_WORLD_SIZE=str(a.gpus))
    env.setdefault("OMP_NUM_THREADS", "1")      # torchrun's default for its workers
    import tempfile
    faildir = tempfile.mkdtemp(prefix="vrpgym_bench_")
    env.update(VRPGYM_BENCH_FAILDIR=faildir, VRPGYM_BENCH_LAUNCHER="1")   # the launcher owns stdout's error line
    os.environ["VRPGYM_BENCH_FAILDIR"] = faildir
    procs = []
    for r in range(a.gpus):
        renv = dict(env, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=renv,
                                      preexec_fn=_die_with_parent))

    def stop(sig=signal.SIGTERM):
        for q in procs:
            if q.poll() is None:
                try:
                    q.send_signal(sig)
                except ProcessLookupError:
                    pass

    def on_signal(signum, _frame):
        stop(signal.SIGTERM)
        deadline = time.time() + 5
        while time.time() < deadline and any(q.poll() is None for q in procs):
            time.sleep(0.05)
        stop(signal.SIGKILL)
        sys.exit(128 + signum)

    signal.signal(signal.SIGTERM, on_signal)
    signal.signal(signal.SIGINT, on_signal)
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for q in list(live):
            code = q.poll()
            if code is None:
                continue
            live.remove(q)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                # one rank failed: the others would wait in a collective forever
                stop(signal.SIGTERM)
                deadline = time.time() + 10
                while time.time() < deadline and any(x.poll() is None for x in procs):
                    time.sleep(0.05)
                stop(signal.SIGKILL)
    if rc != 0:
        print(failure_line(a, a.gpus), flush=True)
    import shutil
    shutil.rmtree(faildir, ignore_errors=True)
    sys.exit(rc)


# ---------------------------------------------------------------------- helpers
def pmc_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/r0N_traffic.json:
    separate FETCH_SIZE / WRITE_SIZE runs, gfx950 corrections applied) -- but ONLY if they were
    collected on the kernels this library was built from (the file carries vrp_source_hash() of
    the build it measured; tools/collect_profiles.sh).  Counters cannot be read inside this
    run (they need rocprofv3 around the process), so a stale or missing file gives None: the
    line never carries bytes of other kernels."""
    import glob
    import vrpgym_hip as hip
    mine = hip.lib().vrp_source_hash().decode()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as fh:
                doc = json.load(fh)
            if doc.get("source_hash") != mine:
                continue
            return doc["workloads"][workload]["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def algorithmic_bytes_per_step(B, N):
    """bytes(B,N) = B*(523*N + 72) per decode+env step."""
    return B * (523 * N + 72)


def make(kind, N, B, seed, device, shard=None, generator="numpy"):
    import agents
    from gym_vrp.envs import IRPEnv, TSPEnv, VRPEnv
    logging.getLogger().setLevel(logging.WARNING)  # the agents' module sets INFO on import
    Env = (TSPEnv, VRPEnv, IRPEnv)[kind]
    Agent = (agents.TSPAgent, agents.VRPAgent, agents.IRPAgent)[kind]
    gb = B * (shard[1] if shard else 1)
    env = Env(num_nodes=N, batch_size=gb, num_draw=1, seed=seed, device=device, shard=shard,
              generator=generator)
    agent = Agent(seed=69, csv_path=os.devnull)
    agent.model.eval()
    return env, agent


def rewind(env):
    """Start-of-episode state on the SAME resident instances (one device launch)."""
    env._reset_state()
    env._step_count = 0
    env._last_rollout = None


def sync_barrier(dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def timed_rollouts(env, agent, greedy, steps, warmup, dist, reset=None, blocks=1):
    """`blocks` blocks of EXACTLY K rollouts, each between barrier + synchronize brackets (a
    block of 20 TSP-20 rollouts is 6 ms: one clock ramp moves it by 8 %, so the headline is the
    median block and the spread is reported).  reset=None: instances stay resident (the
    headline); "env": env.reset() (new instances) before every rollout.
    Returns (list of block seconds, T, mean tour cost)."""
    from agents import runtime

    def one():
        if reset == "env":
            env.reset(return_state=False)
            return runtime.rollout(agent.model, env, greedy)
        # same resident instances, a fresh episode: the state reset (visited, location, load) is
        # part of the rollout's set-up kernel (VRP_ENV_RESET_ON_ROLLOUT)
        return runtime.rollout(agent.model, env, greedy, reset_env=True)

    res, dts = None, []
    with torch.no_grad():
        for _ in range(max(warmup, 1)):
            res = one()
        for _ in range(blocks):
            sync_barrier(dist)
            t0 = time.perf_counter()
            for _ in range(steps):
                res = one()
            sync_barrier(dist)
            dts.append(time.perf_counter() - t0)
    return dts, res.T, float(-res.acc_loss.mean().item())


def timed_training(env, agent, steps, warmup, dist, blocks=1):
    """`blocks` blocks of EXACTLY K training epochs (TSPAgent.train_epoch: graph_tsp_agent.py:174-189
    of the reference), each between the same brackets.  (One block of six epochs read 56.9 ms per
    epoch instead of 17.3 on a freshly started box once: a first-touch page-in inside the timed
    region; the median block is reported and the spread kept.)  Returns (list of block seconds,
    env steps taken by all rollouts of the LAST block's epochs on this rank, rollouts per epoch,
    mean sampled cost of the last epoch, all-reduce ms/epoch, mean steps per rollout)."""
    from agents import distributed, runtime
    from scipy import stats  # noqa: F401  (first import outside the timed region)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):  # "replacing baceline" must not hit the JSON
        for _ in range(max(warmup, 1)):
            agent.train_epoch(env, 1)
        dts = []
        for _ in range(max(blocks, 1)):
            sync_barrier(dist)
            runtime.ROLLOUT_LOG = log = []
            distributed.ALLREDUCE_EVENTS = evs = []
            t0 = time.perf_counter()
            for _ in range(steps):
                _, cost, _ = agent.train_epoch(env, 1)
            sync_barrier(dist)
            dts.append(time.perf_counter() - t0)
    runtime.ROLLOUT_LOG = distributed.ALLREDUCE_EVENTS = None
    sum_T = sum(r.T for r in log)  # env steps per graph over every rollout of the timed epochs
    ar_ms = sum(a.elapsed_time(b) for a, b in evs) / max(steps, 1)
    # an epoch's rollouts in order: sampled model, sampled baseline, greedy model, greedy baseline
    per = max(len(log) // max(steps, 1), 1)
    Ts = [r.T for r in log]
    t_sampled = float(np.mean([t for i, t in enumerate(Ts) if i % per < 2])) if Ts else 0.0
    t_greedy = float(np.mean([t for i, t in enumerate(Ts) if i % per >= 2])) if per > 2 else t_sampled
    return dts, sum_T, per, float(-cost.item()), ar_ms, (t_sampled, t_greedy)


def _event():
    return torch.cuda.Event(enable_timing=True)


def step_kernel_roofline(kind, N, B, greedy, device, reps=5, extra_flags=0, per_step=False):
    """Timing of the decode+env step on the library's launch stream (torch's current one),
    HIP events only:

    avg_launch_us     the step kernel itself: the launches the product's own C loop issues
                      (vrp_rollout_steps_range), step 0 in one event pair and steps 1..T-1
                      back to back in another, the once-per-episode table build between
                      them untimed; = kernel time + the ~1 us kernel-to-kernel boundary.
                      This is the figure `achieved`/`frac` use; rocprofv3's average duration
                      of the same kernel (profiles/) must agree with it.
    loop_us_per_step  (prologue + step loop incl. the first-node table build and the no-op
                      launches after `done`) / T, one event pair: what a step costs the
                      decode loop once the work hoisted out of it is charged to it.
    rollout_us        the whole vrp_rollout (encoder + prologue + loop), one event pair.
    event_pair_per_launch_us  an event pair around EVERY launch, driven from Python (adds
                      ~2.5 us of record/launch/record per launch; kept as a cross-check)."""
    import ctypes as C
    import vrpgym_hip as hip
    from agents import runtime
    env, agent = make(kind, N, B, 69, device)
    lib = hip.lib()
    model = agent.model
    with torch.no_grad():
        res = runtime.rollout(model, env, greedy, trace=False)  # fills emb + tables
        T = res.T
    max_steps = res.max_steps
    ew, dw = runtime.encoder_struct(model.encoder), runtime.decoder_struct(model.decoder)
    derived = runtime.decoder_derived(model.decoder, kind)
    enc_ws, dec_ws = runtime.workspaces(model, env)
    io = hip.RolloutIO()
    io.acc_loss, io.acc_logp = res.acc_loss.data_ptr(), res.acc_logp.data_ptr()
    io.notdone = res.notdone.data_ptr()
    noise = None
    if not greedy:
        noise = torch.empty((max_steps, B, N), device=device).exponential_(1)
        io.noise = noise.data_ptr()
    stream = hip.current_stream(device)
    sflags = (0 if greedy else 1) | extra_flags
    args = (kind, derived.data_ptr(), C.byref(dw))
    tail = (res.emb.data_ptr(), dec_ws.data_ptr(), C.byref(io))

    def start_episode():
        rewind(env)
        cenv = env._cenv()
        hip.check(lib.vrp_env_mask(C.byref(cenv), 0, stream))
        hip.check(lib.vrp_decode_prologue(kind, derived.data_ptr(), B, N, res.emb.data_ptr(),
                                          dec_ws.data_ptr(), stream))
        res.acc_loss.zero_(); res.acc_logp.zero_(); res.notdone.zero_()
        return cenv

    pairs = []
    for _ in range(reps):
        cenv = start_episode()
        evs = [(_event(), _event()) for _ in range(T)]
        for t in range(T):
            evs[t][0].record()
            hip.check(lib.vrp_decode_step(*args, C.byref(cenv), *tail, t, max_steps,
                                          sflags | 8, stream))
            evs[t][1].record()
            if t == 0:  # the once-per-episode first-node fold: other kernels, outside the pair
                hip.check(lib.vrp_decode_first_row(kind, derived.data_ptr(), B, N,
                                                   res.emb.data_ptr(), dec_ws.data_ptr(), stream))
        torch.cuda.synchronize()
        pairs += [a.elapsed_time(b) * 1e-3 for a, b in evs]
    if per_step:  # tuning aid: mean duration of step t over the repetitions
        return [round(float(np.mean(pairs[t::T])) * 1e6, 2) for t in range(T)]

    name = lib.vrp_step_kernel_name(kind, B, N, sflags)
    fused = b"persistent" in name   # steps 1 .. T-1 run as ONE launch
    kern = []
    for _ in range(reps):
        cenv = start_episode()
        ev = [_event() for _ in range(4)]
        ev[0].record()
        hip.check(lib.vrp_rollout_steps_range(*args, C.byref(cenv), *tail, 0, 1, max_steps,
                                              sflags | 8, stream))
        ev[1].record()
        hip.check(lib.vrp_decode_first_row(kind, derived.data_ptr(), B, N, res.emb.data_ptr(),
                                           dec_ws.data_ptr(), stream))
        ev[2].record()
        # (a range that ends the episode may run as ONE launch, the persistent kernel: then up
        # to max_steps; per-step kernels: the T real steps, not the no-op launches after `done`)
        hip.check(lib.vrp_rollout_steps_range(*args, C.byref(cenv), *tail, 1,
                                              max_steps if fused else T, max_steps,
                                              sflags | 8, stream))
        ev[3].record()
        torch.cuda.synchronize()
        kern.append((ev[0].elapsed_time(ev[1]) + ev[2].elapsed_time(ev[3])) * 1e-3 / T)

    loops = []
    for _ in range(reps):
        rewind(env)
        cenv = env._cenv()
        hip.check(lib.vrp_env_mask(C.byref(cenv), 0, stream))
        e0, e1 = _event(), _event()
        e0.record()
        hip.check(lib.vrp_decode_prologue(kind, derived.data_ptr(), B, N, res.emb.data_ptr(),
                                          dec_ws.data_ptr(), stream))
        hip.check(lib.vrp_rollout_steps(*args, C.byref(cenv), *tail, max_steps, sflags, stream))
        e1.record()
        torch.cuda.synchronize()
        loops.append(e0.elapsed_time(e1) * 1e-3)

    rolls = []
    for _ in range(reps):
        rewind(env)
        cenv = env._cenv()
        e0, e1 = _event(), _event()
        e0.record()
        hip.check(lib.vrp_rollout(kind, C.byref(ew), C.byref(dw), derived.data_ptr(),
                                  C.byref(cenv), 0, sflags, res.emb.data_ptr(), enc_ws.data_ptr(),
                                  dec_ws.data_ptr(), C.byref(io), max_steps, stream))
        e1.record()
        torch.cuda.synchronize()
        rolls.append(e0.elapsed_time(e1) * 1e-3)

    avg, pair = float(np.mean(kern)), float(np.mean(pairs))
    loop, roll = float(np.mean(loops)), float(np.mean(rolls))
    byts = algorithmic_bytes_per_step(B, N)
    achieved = byts / avg / 1e9
    wl = f"kind{kind}_N{N}_B{B}"
    traffic = pmc_traffic(wl)
    out = {"bound": "hbm", "kernel": name.decode() if name else "?", "workload": wl,
           "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
           "algorithmic_bytes_per_launch": byts, "avg_launch_us": round(avg * 1e6, 3),
           "timer": "HIP events around the C loop's launches (vrp_rollout_steps_range)",
           "launches_timed": reps * T, "steps_per_episode": T,
           "event_pair_per_launch_us": round(pair * 1e6, 3),
           "loop_us_per_step": round(loop / T * 1e6, 3),
           "loop_frac": round(byts * T / loop / 1e9 / HBM_PEAK_GBS, 4),
           "loop_us": round(loop * 1e6, 1),
           "rollout_us": round(roll * 1e6, 1),
           "rollout_frac": round(byts * T / roll / 1e9 / HBM_PEAK_GBS, 4)}
    if traffic:
        out["frac_on_measured_traffic"] = round(traffic / avg / 1e9 / HBM_PEAK_GBS, 4)
    return out


MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (MI355X_MICROARCH.md; = vector peak)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak (same table): 16 x the fp32 MFMA rate


def encoder_roofline(kind, N, B, device, reps=20):
    """The encoder phase of the rollout (vrp_rollout_encode: set-up + all layers; ONE launch,
    encoder_stack_x3_kernel<3>, for small eval-mode batches) against the matrix-core ceiling of
    its arithmetic (see below; the fp32 MFMA peak for the fp32 kernels, VRP_ENCODER_FP32=1).
    Flops = B*N*(1 180 160 + 1 536 N) (projections, attention, feed-forward of
    three layers); time = HIP events around the launches on the library's stream."""
    import ctypes as C
    import vrpgym_hip as hip
    from agents import runtime
    env, agent = make(kind, N, B, 69, device)
    lib = hip.lib()
    model = agent.model
    with torch.no_grad():
        res = runtime.rollout(model, env, True)
    ew = runtime.encoder_struct(model.encoder)
    derived = runtime.decoder_derived(model.decoder, kind)
    enc_ws, dec_ws = runtime.workspaces(model, env)
    io = hip.RolloutIO()
    io.acc_loss, io.acc_logp = res.acc_loss.data_ptr(), res.acc_logp.data_ptr()
    io.notdone = res.notdone.data_ptr()
    stream = hip.current_stream(device)
    ts = []
    for i in range(reps + 3):
        rewind(env)
        cenv = env._cenv()
        e0, e1 = _event(), _event()
        e0.record()
        hip.check(lib.vrp_rollout_encode(kind, C.byref(ew), derived.data_ptr(), C.byref(cenv), 0,
                                         res.emb.data_ptr(), enc_ws.data_ptr(), dec_ws.data_ptr(),
                                         C.byref(io), res.max_steps, stream))
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e-3)
    avg = float(np.mean(ts))
    dense, attn = B * N * 1180160, B * N * 1536 * N
    flops = dense + attn
    name = lib.vrp_encoder_kernel_name(C.byref(ew), 0, B, N)
    name = name.decode() if name else "?"
    out = {"bound": "mfma", "kernel": name, "workload": f"kind{kind}_N{N}_B{B}",
           "achieved": round(flops / avg / 1e12, 2), "unit": "TFLOP/s", "traffic": None,
           "algorithmic_flops_per_launch": flops, "avg_launch_us": round(avg * 1e6, 2),
           "timer": "HIP events around vrp_rollout_encode (one launch for small eval-mode "
                    "batches; the per-layer kernels otherwise)",
           "launches_timed": reps}
    if "_x3" in name:
        # Round 5: the dense products (projections, feed-forward: 1 180 160 of the flops per row)
        # run as SIX bf16 MFMAs per fp32 product on three-plane operands (csrc/encoder_x3.h), the
        # attention (1 536 N per row, K = 16 per head) on the fp32 MFMA.  The ceiling of that mix,
        # in algorithmic (fp32-equivalent) flops: dense at 2 500 / 6 TFLOP/s + attention at 157.3.
        bound_s = 6.0 * dense / (MFMA_BF16_PEAK_TFLOPS * 1e12) + attn / (MFMA_F32_PEAK_TFLOPS * 1e12)
        peak = flops / bound_s / 1e12
        out.update({
            "peak": round(peak, 1), "frac": round(flops / avg / 1e12 / peak, 4),
            "arithmetic": "fp32 operands as three bf16 planes, six v_mfma_f32_16x16x32_bf16 per "
                          "product, fp32 accumulation (error = that of the fp32 MFMA: "
                          "profiles/r05_bf16x3_probe.txt); attention on v_mfma_f32_16x16x4_f32",
            "peak_is": "algorithmic flops / (6 x dense flops / 2500 TFLOP/s bf16 + attention flops "
                       "/ 157.3 TFLOP/s fp32)",
            "issued_bf16_mfma_tflops": round(6.0 * dense / avg / 1e12, 1),
            "bf16_mfma_peak": MFMA_BF16_PEAK_TFLOPS,
            "vs_fp32_mfma_peak": round(flops / avg / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)})
    else:
        out.update({"peak": MFMA_F32_PEAK_TFLOPS,
                    "frac": round(flops / avg / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)})
    return out


def training_flops_per_epoch(kind, N, B, T_sampled, T_greedy):
    """Algorithmic fp32 flops of one REINFORCE epoch (TSPAgent.train's loop body,
    graph_tsp_agent.py:174-189 + baseline_update :275-306), per GPU:

      encoder forward  x4   model (train mode, taped) + baseline for the sampled pair, model +
                            baseline for the greedy pair: B N (1 180 160 + 1 536 N) each
                            (projections, attention, feed-forward of 3 layers)
      encoder backward x1   2x a forward (input- and weight-gradient products)
      prologue         x4   per-episode decoder tables: B N 2 (4*384*128 + 2*8*48 N)
      decode steps          sum over the four rollouts of B T 2*8*N*N: the table formulation the
                            training batches run (glimpse weights x logit-table rows; the 196 608
                            flop of per-step weight folds in SURVEY 8d belong to the raw-tile
                            formulation, which these batch sizes never take)
      decoder backward x1   the T sampled steps re-run un-folded (graph_decoder.py:75-107):
                            forward = B T (2*384*384 [query] + 2*2*8*48 N [scores, values] +
                            2*384*384 [out_proj] + 2*128*384 [_att_output] + 2*128 N [logits])
                            + B N 2*128*(384+384+128) [K, V, _kp projections]; backward = 2x that,
                            plus the forward recompute = 3x
    Adam, the t-test and BatchNorm reductions are not matrix work and are left out."""
    enc = B * N * (1180160 + 1536 * N)
    pro = B * N * 2 * (4 * 384 * 128 + 2 * 8 * 48 * N)
    steps = B * (2 * T_sampled + 2 * T_greedy) * (2 * 8 * N * N)
    dec_fwd = (B * T_sampled * (2 * 384 * 384 + 2 * 2 * 8 * 48 * N + 2 * 384 * 384 + 2 * 128 * 384
                                + 2 * 128 * N) + B * N * 2 * 128 * (384 + 384 + 128))
    parts = {"encoder_forward_x4": 4 * enc, "encoder_backward": 2 * enc, "prologue_x4": 4 * pro,
             "decode_steps": steps, "decoder_backward": 3 * dec_fwd}
    return sum(parts.values()), parts


def training_dense_flops(kind, N, B, T_sampled):
    """The part of training_flops_per_epoch that is DENSE products with a 128- or 384-long inner
    dimension (projections, feed-forward, the prologue's stage 1, the decoder backward's
    projections): what the library issues -- or could issue -- as six bf16 MFMAs per product.
    Everything else (per-head attention with inner dimension 16 / 48, table steps) is priced at
    the fp32 rate.  Used for roofline_train's ceiling, so counting a product here that still
    runs on the fp32 MFMA (gemm_rows_wide_kernel) only LOWERS the reported fraction."""
    enc_dense = B * N * 1180160
    pro_dense = B * N * 2 * 4 * 384 * 128
    dec_dense = (B * T_sampled * (2 * 384 * 384 + 2 * 384 * 384 + 2 * 128 * 384)
                 + B * N * 2 * 128 * (384 + 384 + 128))
    return 4 * enc_dense + 2 * enc_dense + 4 * pro_dense + 3 * dec_dense


def gemm_kernel_roofline(device, M=81920, N=384, K=128, reps=20):
    """The training path's most frequent tall GEMM (the train-mode in_proj of VRP-40 x 2048:
    M = B N rows, bias only; vrp_gemm_nt -> gemm_rows_wide_kernel: all of W in registers) timed
    with HIP events."""
    import vrpgym_hip as hip
    lib = hip.lib()
    A = torch.randn(M, K, device=device)
    W = torch.randn(N, K, device=device) * 0.1
    b = torch.randn(N, device=device)
    C = torch.empty(M, N, device=device)
    st = hip.current_stream(device)

    def run():
        hip.check(lib.vrp_gemm_nt(A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), None, 0,
                                  C.data_ptr(), N, M, N, K, 0, st))
    for _ in range(3):
        run()
    e0, e1 = _event(), _event()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tf = 2.0 * M * N * K / us / 1e6
    return {"kernel": "gemm_rows_wide_kernel<3, false>", "shape": [M, N, K], "avg_launch_us": round(us, 2),
            "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
            "hbm_bytes": 4 * (M * K + M * N + N * K),
            "hbm_frac_at_this_time": round(4 * (M * K + M * N + N * K) / us / 1e3 / HBM_PEAK_GBS, 4)}


def device_identity(device):
    """What tells two GPUs apart: uuid / PCI bus id where torch exposes them."""
    p = torch.cuda.get_device_properties(device)
    ident = {"index": device.index, "name": p.name}
    for key in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        v = getattr(p, key, None)
        if v is not None:
            ident[key] = str(v)
    return ident


def cpu_baseline(kind, N, B, greedy, budget_s=15.0):
    """The CPU oracle (a port of the reference's algorithm: vectorised numpy env +
    torch-CPU fp32 policy) on this host's cores, same workload, bounded sample.  The ops of a
    rollout at this size are small: more threads than a socket's worth make it SLOWER, so a
    few thread counts are tried briefly and the sample runs at the best one (`cores`)."""
    from copy import deepcopy
    from oracle import envs as oenv
    from oracle import policy as opol
    sd, _ = opol.init_state_dicts(kind, 69)
    env = oenv.OracleEnv(kind, N, B, 1, 69)
    default_threads = torch.get_num_threads()
    cands = sorted({t for t in (8, 16, 32, default_threads) if t <= (os.cpu_count() or 1)})
    best_t, best_dt = default_threads, None
    with torch.no_grad():
        for t in cands:
            torch.set_num_threads(t)
            opol.rollout(sd, deepcopy(env), greedy)  # warm
            t0 = time.perf_counter()
            opol.rollout(sd, deepcopy(env), greedy)
            dt = time.perf_counter() - t0
            if best_dt is None or dt < best_dt:
                best_t, best_dt = t, dt
        torch.set_num_threads(best_t)
        n, node_steps, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            _, _, T = opol.rollout(sd, deepcopy(env), greedy)
            node_steps += B * N * T
            n += 1
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    return {"value": round(node_steps / dt, 1), "unit": "node-steps/s",
            "cores": best_t, "host_cpus": os.cpu_count(), "kind": "port",
            "threads_tried": cands,
            "sample": f"{n} {'greedy' if greedy else 'sampled'} rollouts of kind{kind} N={N} B={B} "
                      f"in {dt:.1f}s (oracle/: numpy env + torch-CPU fp32 policy; measured 4.8x "
                      f"(TSP-20) to 7.6x (VRP-40) faster than the reference itself on 8 cores, "
                      f"BASELINE.md section 3)"}


def run_workload(name, steps, warmup, device, dist, rank, world, blocks=1):
    """One workload on this rank -> dict of whole-job figures (rank-reduced).  Per block the
    time is the MAX over ranks; `seconds` is the median block."""
    kind, N, B, mode = WORKLOADS[name]
    shard = (rank, world) if world > 1 else None
    train = mode == "train"
    env, agent = make(kind, N, B, 69, device, shard=shard,
                      generator="device" if train else "numpy")
    extra = {}
    if train:
        from agents import distributed
        distributed.broadcast_model(agent.model)
        distributed.broadcast_model(agent.target_model)
        dts, sum_T, rollouts, cost, ar_ms, (t_s, t_g) = timed_training(env, agent, steps, warmup,
                                                                        dist, blocks=blocks)
        dt = sorted(dts)[len(dts) // 2]   # (this rank's median block: the flop rate below)
        graph_steps = sum_T * B
        node_steps = graph_steps * N
        T = round(sum_T / max(steps * rollouts, 1), 2)
        extra = {"rollouts_per_step": rollouts, "allreduce_ms_per_step": round(ar_ms, 4),
                 "grad_bucket_bytes": 4 * sum(p.numel() for p in agent.model.parameters()
                                              if p.grad is not None)}
        flops, parts = training_flops_per_epoch(kind, N, B, t_s, t_g)
        tf = flops / (dt / steps) / 1e12
        # ceiling of the arithmetic an epoch issues: dense products as six bf16 MFMAs (2500 / 6
        # TFLOP/s), the rest at the fp32 rate -- the same accounting as encoder_roofline
        dense = training_dense_flops(kind, N, B, t_s)
        bound_s = 6.0 * dense / (MFMA_BF16_PEAK_TFLOPS * 1e12) + (flops - dense) / (MFMA_F32_PEAK_TFLOPS * 1e12)
        peak_mixed = flops / bound_s / 1e12
        extra["roofline_train"] = {
            "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak_mixed, 1),
            "unit": "TFLOP/s", "frac": round(tf / peak_mixed, 4),
            "peak_is": "algorithmic flops / (6 x dense flops / 2500 TFLOP/s bf16 + other flops / "
                       "157.3 TFLOP/s fp32): the ceiling of the bf16-plane arithmetic "
                       "(bench.training_dense_flops)",
            "vs_fp32_mfma_peak": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
            "dense_share_of_flops": round(dense / flops, 3),
            "algorithmic_gflop_per_epoch": round(flops / 1e9, 2),
            "gflop_by_part": {k: round(v / 1e9, 2) for k, v in parts.items()},
            "steps_per_rollout": {"sampled": round(t_s, 1), "greedy": round(t_g, 1)},
            "ms_per_epoch": round(dt / steps * 1e3, 3),
            "note": "whole epoch (rollouts incl. their HBM-bound decode loops, backward, Adam, "
                    "t-test) against the ceiling of the matrix arithmetic it issues; flop model: "
                    "bench.training_flops_per_epoch"}
    else:
        dts, T, cost = timed_rollouts(env, agent, mode == "greedy", steps, warmup, dist,
                                      blocks=blocks)
        graph_steps = steps * B * T
        node_steps = graph_steps * N
    red = torch.tensor([cost, float(node_steps), float(graph_steps)], device=device,
                       dtype=torch.float64)
    mine = torch.tensor(dts, device=device, dtype=torch.float64)
    tmax, tmin = mine.clone(), mine.clone()
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
    per_block = sorted(tmax.tolist())
    med = per_block[len(per_block) // 2] if len(per_block) % 2 else \
        0.5 * (per_block[len(per_block) // 2 - 1] + per_block[len(per_block) // 2])
    ms = lambda sec: round(sec / steps * 1e3, 4)  # noqa: E731
    out = {"name": name, "kind": kind, "N": N, "B": B, "mode": mode, "T": T,
           "seconds": med, "ms_per_step": ms(med),
           "node_steps_per_s": round(float(red[1].item()) / med, 1),
           "graph_steps_per_s": round(float(red[2].item()) / med, 1),
           "mean_tour_cost": round(float(red[0].item()) / world, 6)}
    if len(per_block) > 1 or world > 1:
        # blocks: max-over-ranks time of every block; ranks: fastest / slowest rank's own time
        # of its median block (ms per step) -- the spread the aggregate hides
        mid = int(torch.argsort(tmax)[len(per_block) // 2].item())
        out["dispersion"] = {"blocks": len(per_block), "steps_per_block": steps,
                             "ms_per_step_min": ms(per_block[0]),
                             "ms_per_step_median": ms(med),
                             "ms_per_step_max": ms(per_block[-1]),
                             "rank_ms_per_step_min": ms(float(tmin[mid].item())),
                             "rank_ms_per_step_max": ms(float(tmax[mid].item()))}
    out.update(extra)
    return out, env, agent


def bind_device(local):
    """This rank's GPU, in the order the library depends on: make it torch's (and HIP's) current
    device, check that it is, and only then let the library look at it -- its residency census and
    its per-device state (failure counter, cross-process lease keyed by PCI bus id) belong to the
    device that is current at that moment (vrpgym_hip.require_gpu).  Before any communicator
    exists and before anything is enqueued."""
    import vrpgym_hip as hip
    torch.cuda.set_device(local)
    assert torch.cuda.current_device() == local, (torch.cuda.current_device(), local)
    hip.require_gpu()
    return torch.device("cuda", local)


def describe(kind, N, B, mode, T, world):
    what = {"greedy": "attention agent greedy rollout", "sample": "attention agent sampling rollout",
            "train": "REINFORCE training epoch (2 sampled rollouts, HIP backward, gradient "
                     "all-reduce, Adam, 2 greedy rollouts + paired t-test)"}[mode]
    return (f"{KIND_NAMES[kind]}Env num_nodes={N} batch_size={B} per GPU, {what} "
            f"(encoder + {T} fused decode/env steps per rollout), untrained seed-69 weights")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="tsp20_b512", choices=sorted(WORKLOADS))
    ap.add_argument("--blocks", type=int, default=5,
                    help="timed blocks of --steps rollouts each; ms_per_step is the median block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-north-star", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the short runs of the other BASELINE configs")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(a, sys.argv[1:])  # does not return

    if a.gpus == 1:
        return rank_main(a)          # (the N = 1 path: nothing between the driver and the run)
    rank = int(os.environ.get("RANK", "0"))
    stage = ["start-up"]
    try:
        return rank_main(a, stage)
    except BaseException as exc:     # SystemExit included: a refused start-up is a failure too
        if isinstance(exc, SystemExit) and exc.code in (0, None):
            raise
        rec = record_failure(rank, stage[0], exc)
        sys.stderr.write(f"bench.py rank {rank} failed in [{stage[0]}]: {rec['error']}\n")
        if rank == 0 and os.environ.get("VRPGYM_BENCH_LAUNCHER") != "1":
            time.sleep(1.0)          # torchrun: give the rank that failed first a moment to write its record
            print(failure_line(a, int(os.environ.get("WORLD_SIZE", a.gpus)), rec), flush=True)
        if isinstance(exc, (SystemExit, KeyboardInterrupt)):
            raise
        import traceback
        traceback.print_exc()
        sys.exit(1)


def rank_main(a, stage=None):
    stage = stage if stage is not None else ["run"]
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"WORLD_SIZE={world} but --gpus {a.gpus}"
    # test aid (one-GPU boxes): VRPGYM_BENCH_ONE_GPU=1 maps every rank to cuda:0 and the
    # collectives run over gloo; the driver's real runs never set it
    one_gpu = os.environ.get("VRPGYM_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
        # (ranks that share a GPU need no special care: the library's lease word lets one of them
        # use the persistent step grid at a time, the others run one launch per step)
    if world > 1 and not one_gpu and torch.cuda.device_count() < world:
        # (the parent counted in sysfs; a rank sees what the runtime really offers)
        sys.exit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
    if world > 1 and os.environ.get("VRPGYM_BENCH_TEST_FAIL") == f"startup:{rank}":
        raise RuntimeError("injected start-up failure (VRPGYM_BENCH_TEST_FAIL)")   # CPU test of the failure path
    stage[0] = "bind device"
    device = bind_device(local)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL's own diagnostics, per rank, where the failure line can find them
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", os.path.join(fail_dir(), f"rccl_rank{rank}.log"))
        stage[0] = "init_process_group"
        tmo = datetime.timedelta(seconds=DIST_TIMEOUT_S)
        if one_gpu:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=device, timeout=tmo)  # nccl == RCCL on ROCm
        # first contact: one tiny all-reduce on the device (RCCL builds its rings / xGMI links
        # here, not in init_process_group), checked, before anything is timed
        stage[0] = "first all-reduce"
        one = torch.ones(1, device=device if not one_gpu else "cpu")
        dist.all_reduce(one)
        if not one_gpu:
            torch.cuda.synchronize()
        assert int(one.item()) == world, f"all-reduce of ones over {world} ranks gave {one.item()}"
        stage[0] = "device identities"

    # every rank on its own GPU: gather what identifies the device and insist on N distinct ones
    backend, idents = None, [device_identity(device)]
    if dist is not None:
        backend = dist.get_backend()
        idents = [None] * world
        dist.all_gather_object(idents, device_identity(device))
        keys = {json.dumps({k: v for k, v in i.items() if k != "index"}, sort_keys=True) + (
            "" if any(k in i for k in ("uuid", "pci_bus_id")) else str(i["index"])) for i in idents}
        if not one_gpu:
            assert len(keys) == world, f"{world} ranks but {len(keys)} distinct GPUs: {idents}"

    kind, N, B, mode = WORKLOADS[a.workload]
    stage[0] = "timed workload " + a.workload
    r, env, agent = run_workload(a.workload, a.steps, a.warmup, device, dist, rank, world,
                                 blocks=max(a.blocks, 1))
    stage[0] = "report"
    T = r["T"]
    out = {
        "metric": "env-steps/sec (batch x nodes), " + ("REINFORCE training" if mode == "train"
                                                       else f"{mode} attention rollout"),
        "value": r["node_steps_per_s"], "unit": "node-steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": r["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "dtype_note": "fp32 results throughout (tour cost / log-prob within 1e-5 of the reference's "
                      "CPU fp32 path; env bookkeeping fp64 / integer, bit-exact).  The eval-mode "
                      "encoder's and the decoder prologue's dense products run on the bf16 matrix "
                      "cores with every fp32 operand carried as three bf16 planes (six MFMAs per "
                      "product, fp32 accumulation): measured error equal to the fp32 MFMA's "
                      "(profiles/r05_bf16x3_probe.txt), same parity tests, same tolerances",
        "data": "synthetic" + (" (instances drawn on the device, Philox stream, reference "
                               "distributions)" if mode == "train" else
                               " (reference-exact numpy-stream instances, resident in HBM)"),
        "config": {"workload": describe(kind, N, B, mode, T, world),
                   "name": a.workload, "global_batch": B * world, "num_nodes": N,
                   "steps_per_rollout": T,
                   "parallelism": f"dp{world} (rank r = rows [r*{B},(r+1)*{B}) of one "
                                  f"seed-ordered stream; " +
                                  ("one RCCL all-reduce of the flat gradient per epoch)"
                                   if mode == "train" else "no collective on the rollout path)")},
        "graph_steps_per_s": r["graph_steps_per_s"],
        "mean_tour_cost": r["mean_tour_cost"],
    }
    if "dispersion" in r:
        out["dispersion"] = r["dispersion"]
    if world > 1:
        out["backend"] = backend + (" (RCCL)" if backend == "nccl" else "")
        out["rccl_ranks"] = world if backend == "nccl" else 0
        out["devices"] = idents
    if one_gpu and world > 1:
        out["one_gpu_test_mode"] = True  # all ranks share cuda:0: not a scaling measurement
    if mode == "train":
        out["training"] = {k: r[k] for k in ("rollouts_per_step", "allreduce_ms_per_step",
                                             "grad_bucket_bytes")}
        out["roofline"] = r["roofline_train"]
        if rank == 0:
            out["roofline"]["dominant_kernel"] = gemm_kernel_roofline(device, B * N, 384, 128)

    # ---- short runs of the other BASELINE configs (bounded; failures are reported, not fatal)
    if not a.no_extras and a.workload == "tsp20_b512":
        extras = {}
        todo = [("irp40_b1024_train", 6, 2)] if world > 1 else \
               [("vrp40_b2048_train", 6, 2), ("irp40_b1024_train", 6, 2), ("vrp100_b2048", 4, 1)]
        for name, k, w in todo:
            try:
                e, _, _ = run_workload(name, k, w, device, dist, rank, world, blocks=3)
                extras[name] = {kk: e[kk] for kk in e if kk not in ("name", "kind", "seconds")}
            except Exception as exc:  # pragma: no cover
                extras[name] = {"error": repr(exc)[:300]}
        out["other_configs"] = extras

    if rank == 0 and mode != "train":
        # `roofline` = the DOMINANT kernel of the benched rollout: whichever of the encoder
        # phase (fp32 MFMA-bound) and the decode/env step loop (HBM-bound) takes longer; both
        # are reported (roofline_encoder, roofline_step)
        step_r = step_kernel_roofline(kind, N, B, mode == "greedy", device)
        enc_r = encoder_roofline(kind, N, B, device)
        steps_us = step_r["avg_launch_us"] * step_r["steps_per_episode"]
        out["roofline"] = dict(enc_r if enc_r["avg_launch_us"] >= steps_us else step_r)
        out["roofline"]["share_of_rollout"] = round(
            max(enc_r["avg_launch_us"], steps_us) / step_r["rollout_us"], 3)
        out["roofline_encoder"], out["roofline_step"] = enc_r, step_r
        if not a.no_north_star and world == 1 and a.workload != "tsp40_b8192":
            out["roofline_north_star"] = step_kernel_roofline(0, 40, 8192, True, device)
            out["roofline_north_star_vrp"] = step_kernel_roofline(1, 40, 8192, True, device)
            if a.workload != "vrp100_b2048":
                out["roofline_cfg5"] = step_kernel_roofline(1, 100, 2048, False, device, reps=2)
    if rank == 0 and world == 1 and mode != "train":
        # the same rollouts with env.reset() (new instances: host numpy stream + upload, or
        # the device generator) inside the timed region
        (dt,), T2, _ = timed_rollouts(env, agent, mode == "greedy", max(a.steps // 2, 1), 1, None,
                                      reset="env")
        out["incl_env_reset"] = {"generator": "numpy (host MT19937 replay + upload)",
                                 "ms_per_step": round(dt / max(a.steps // 2, 1) * 1e3, 4),
                                 "value": round(max(a.steps // 2, 1) * B * N * T2 / dt, 1)}
        env_d, _ = make(kind, N, B, 69, device, generator="device")
        (dt,), T2, _ = timed_rollouts(env_d, agent, mode == "greedy", a.steps, 1, None,
                                      reset="env")
        out["incl_env_reset_device_generator"] = {
            "ms_per_step": round(dt / a.steps * 1e3, 4),
            "value": round(a.steps * B * N * T2 / dt, 1)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(kind, N, B, mode != "sample")
        if mode != "train":
            out["cpu_baseline"]["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

// Steps 1 .. T-1 of an episode in ONE launch -- and, for batches of up to 1024 graphs, the first
// chosen node's part of the score rows (persist_first_base) -- for the latency-bound regime
// (B <= 2048 graphs, N <= 63): one wave per graph keeps the whole per-graph state of the rollout loop
// (agents/graph_tsp_agent.py:78-88: GraphDecoder.forward agents/graph_decoder.py:51-115 +
// env.step gym_vrp/envs/tsp.py:60-101) in registers across steps -- coordinates, visited row,
// demand, load, current/last node, accumulators -- and only streams the step's score row and
// the rows of the logit table.  What a kernel boundary per step used to provide is replaced by
// the two things the algorithm really needs from other graphs:
//   * QUIRK D3 (graph_decoder.py:93-94): head h of graph b adds the mask row of graph
//     (8b + h) mod B.  Every graph publishes its mask for step t as ONE 8-byte word
//     hist[t][b] = bits | 1 << 63, written by one agent-scope (sc1) store; a reader polls the
//     eight words it needs with agent-scope loads until bit 63 shows -- the data is its own
//     flag (the naturally aligned 8-byte granule of MI355X_MICROARCH.md, "handoff-1to1"), no
//     barrier, no fence, no second round trip.  (All-masked rows do not exist: at least one
//     node is always selectable, and bit 63 is never a node: N <= 63.)
//   * the batch-wide `done` (tsp.py:95,103-104): `visited` all ones is monotone per graph
//     (once reached it is re-established by every later step), so T - 1 = max over graphs of
//     the step ta_b at which a graph first reaches it.  A graph stops at ta_b when it stands on
//     the depot (or is a TSP graph: the whole batch ends there); a VRP/IRP graph that reached
//     it by LEAVING the depot for its last customer also computes its forced
//     way back (the only unmasked node, log-prob exactly 0) and keeps that edge's reward
//     aside: vrp_persistent_finalize adds it iff the batch ran on (T - 1 > ta_b).  Later steps
//     of a finished graph are self-loops on the depot with reward 0 and log-prob 0: nothing to
//     compute; it publishes its (constant) mask for all remaining steps at once.
// FAST only when every workgroup of the grid is resident -- CORRECT whether or not it is:
//   * vrp_persistent_eligible admits only B <= vrp_persistent_capacity() (occupancy of THIS
//     kernel x the compute units a census kernel finds usable), persistent launches of one
//     process are serialised across streams, and processes that share a device take turns
//     through a lease word in shared memory (device_lease_claim below): whoever does not hold
//     it runs one launch per step;
//   * none of that is relied upon.  A wait lasts at most `spin_ticks` of the wall clock
//     (20 ms); the first wave that gives up raises `err`, every other wave sees it at its next
//     hand-off (lane 8 reads it together with the eight mask words; a spinning lane polls it)
//     and leaves: a grid that cannot make progress drains in a few tens of milliseconds;
//   * every graph saves the state it was launched with (visited row, location, load,
//     accumulators) before it overwrites anything, and persistent_finalize_kernel -- one
//     workgroup, launched behind the grid (a four-wave TSP grid does without the launch: the
//     workgroup that raised `err` first runs the same code once every other one has left, see
//     the end of decode_persistent4_kernel) -- on `err` puts that state back and
//     walks the remaining steps itself with the per-step kernel's body (decoder_rt_body.h),
//     one workgroup barrier per step instead of hand-off words.  The step kernels are
//     bit-identical (tests/test_gpu_parity.py), the in-kernel noise is counter-based: the
//     episode's results are those of an undisturbed run.  No host round trip, no NaN, no
//     exception; the host learns about it from a pinned counter (vrp_persistent_failures) and
//     backs off from the persistent path for a while.
#include <atomic>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include "decoder_rt_body.h"

struct PersistParams {
  StepParams s;               // s.t = first step of the launch (>= 1)
  unsigned long long *hist;   // (2N, B) >= (max_steps + 1, B), zeroed by vrp_decode_prologue
  int32_t *ta;                // (B)
  float *ret;                 // (B) reward of the way back after ta (0 if none)
  int32_t *wb_cur;            // (B) node the graph stood on before its way back (-1: none)
  double *wb_load;            // (B) its load there (IRP)
  int32_t *err;               // raised by the first wave that gives up waiting
  int32_t *census;            // residency census mode (vrp_persistent_capacity): {arrived, saw all}
  long long spin_ticks;       // longest wait for a hand-off word, in wall_clock64() ticks
  int32_t *fail_host;         // pinned host counter: episodes of this device that fell back
  // the first chosen node's part of the score rows, computed by the grid itself (persist_first_base)
  int fold_final;             // four-wave TSP grid: no persistent_finalize_kernel behind it (see the
                              // end of decode_persistent4_kernel); hist row 0 = per-graph "left" flags
  int fold_first;             // 1: base is not there yet (vrp_decode_first_row was skipped)
  const float *WqfT;          // (128,384)  Derived::WqfT
  const float *KK4;           // (B,8,12,N,4)  DecWs::KK4
  const float *SG;            // (B,8,N)
  float *base_out;            // (B,8,N)  = StepParams::base, writable
  // the state this launch started from (saved by every graph before it overwrites anything)
  uint8_t *sv_visited;        // (B,N)
  int32_t *sv_cur, *sv_last;  // (B)
  double *sv_load;            // (B)
  float *sv_accl, *sv_accp;   // (B)
};

// The workgroup is ONE wave: its LDS operations execute in program order, so a barrier between
// writing and reading a_s / u_s only has to keep the compiler from reordering them -- s_barrier
// would also drain every global load in flight (__syncthreads waits for vmcnt(0)), among them the
// prefetched rows of the logit table and the next step's score row.
#define PERSIST_WAVE_SYNC()                                  \
  do {                                                       \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0) */     \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)
#define PERSIST_VALID (1ull << 63)

// number of set bits of a wave-wide ballot below this lane (v_mbcnt: no lane-mask pair to keep alive)
__device__ __forceinline__ int lanes_below(unsigned long long bits) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bits >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bits, 0u));
}

// agent-scope (write-through) store: what a four-wave grid that finalizes itself writes must not
// linger dirty in one XCD's L2 (see the end of decode_persistent4_kernel)
template <typename T>
__device__ __forceinline__ void st_agent(T *p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One hand-off wait.  Lanes 0..7 poll the mask word of "their" graph until bit 63 shows; lane 8
// reads the error flag in the same round trip (and reports "not valid" when it is raised), so a
// wave whose words are all there still leaves a failed episode at its next step.  A legitimate
// wait is microseconds; a lane gives up after `limit` wall-clock ticks (20 ms) or as soon as it
// sees the flag (polled every 16 spins).  Returns the word (bit 63 clear = give up).
__device__ __forceinline__ unsigned long long persist_wait(const unsigned long long *hist_row, int B,
                                                           int b, int lane, const int32_t *err,
                                                           long long limit) {
  unsigned long long w = PERSIST_VALID;
  if (lane < 8) {
    const unsigned long long *src = hist_row + (b * 8 + lane) % B;
    w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!(w & PERSIST_VALID)) {
      const long long start = wall_clock64();
      int spins = 0;
      do {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 15) == 0 &&
            (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
             wall_clock64() - start > limit))
          break;
        w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } while (!(w & PERSIST_VALID));
    }
  } else if (lane == 8) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) w = 0ull;
  }
  return w;
}

__global__ __launch_bounds__(64, 3) void decode_persistent_kernel(PersistParams pp) {
  const StepParams &p = pp.s;
  __shared__ __attribute__((aligned(16))) float a_s[8 * 64];  // a[h][n], hn order
  __shared__ __attribute__((aligned(16))) float u_s[64];
  __shared__ int sel_s[64];  // compacted list of selectable nodes

  const int lane = threadIdx.x;
  if (pp.census) {
    // Residency census: is a grid of this size, of THIS kernel (same registers, same LDS),
    // resident all at once?  Every workgroup checks in and waits (bounded) for the others; a grid
    // that runs in two shifts leaves the first shift without the full count.
    if (lane == 0) {
      __hip_atomic_fetch_add(&pp.census[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int seen = 0;
      for (int spins = 0; spins < 4000; ++spins) {
        seen = __hip_atomic_load(&pp.census[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen >= (int)gridDim.x) break;
        __builtin_amdgcn_s_sleep(8);
      }
      if (seen >= (int)gridDim.x)
        __hip_atomic_fetch_add(&pp.census[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const int N = p.N, B = p.B;
  const int b = blockIdx.x;
  const int t0 = p.t;
  if (p.io.notdone[t0 - 1] == 0) return;  // the batch was done before this launch
  const int n4 = 2 * N;  // float4 per RT row (8N floats)
  const int rsl = lane >> 3, part = lane & 7;
  const bool inN = lane < N;
  const int ln = inN ? lane : 0;
  const size_t row = (size_t)b * 8 * N;

  // ---- per-graph state, loaded once ----------------------------------------------------
  const uint8_t *mask0 = p.env.mask + (size_t)(t0 & 1) * B * N;
  int own_mask = mask0[(size_t)b * N + ln];
  int msk[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) msk[h] = mask0[(size_t)((b * 8 + h) % B) * N + ln];  // QUIRK D3
  float sld[8], bs[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    sld[h] = (p.kind == VRP_KIND_IRP) ? p.SLD[row + h * N + ln] : 0.f;
    bs[h] = p.base ? p.base[row + h * N + ln] : 0.f;
  }
  const float cv = p.cvec[(size_t)b * N + ln];
  const double2 xy = reinterpret_cast<const double2 *>(p.env.pos)[(size_t)b * N + ln];
  int vis = inN ? p.env.visited[(size_t)b * N + ln] : 1;
  const double dem = (p.kind == VRP_KIND_IRP) ? p.env.demand[(size_t)b * N + ln] : 0.0;
  int cur = p.env.cur[b];
  const int dep = p.env.depot[b];
  double load0 = (p.kind == VRP_KIND_IRP) ? p.env.load[b] : 1.0;
  float accl = p.io.acc_loss[b], accp = p.io.acc_logp[b];
  int last = __builtin_amdgcn_readfirstlane(p.last[b]);
  float sc[8];  // this step's score row SL[b][last] (requested at the end of the step before)
  {
    const float *srow = p.SL + ((size_t)b * N + last) * 8 * N;
#pragma unroll
    for (int h = 0; h < 8; ++h) sc[h] = srow[h * N + ln];
  }
  // what the fallback (persistent_finalize_kernel) restarts from, should this launch fail
  if (inN) pp.sv_visited[(size_t)b * N + lane] = (uint8_t)vis;
  if (lane == 0) {
    pp.sv_cur[b] = cur;
    pp.sv_last[b] = last;
    pp.sv_load[b] = load0;
    pp.sv_accl[b] = accl;
    pp.sv_accp[b] = accp;
  }
  const int cnt = (n4 - part + 7) >> 3;          // float4 of a row owned by this lane
  const int nchunk = (((n4 + 7) >> 3) + RT_U - 1) / RT_U;
  const float4 *rtb = reinterpret_cast<const float4 *>(p.RT) + (size_t)b * N * n4 + part;
  constexpr int NB = 2;  // work items in flight
  int ta = -1, wb_cur = -1;
  float ret = 0.f;
  double wb_load = 1.0;
  int t = t0;

  for (; t < p.max_steps; ++t) {
    // ---- loads that do not depend on other graphs: noise, first rows of the logit table
    const float q_noise = !p.sample ? 1.f
                          : p.io.noise ? p.io.noise[((size_t)t * B + b) * N + ln]
                                       : vrp_exp1_noise(p.io.noise_seed, t, b, ln);
    const bool s_i = inN && !own_mask;
    const unsigned long long sel = __ballot(s_i);
    const int nsel = __popcll(sel);
    if (s_i) sel_s[__popcll(sel & ((1ull << lane) - 1ull))] = lane;
    const int total = ((nsel + 7) >> 3) * nchunk;  // work items (pass, chunk), wave-uniform
    float4 rbuf[NB][RT_U];
    int mrow[NB];
    const int m_first = (rsl < nsel) ? kth_set_bit(sel, rsl) : -1;  // pass 0 rows
    auto load_item = [&](float4 (&r)[RT_U], int w, int &m_out) {
      const int pass = w / nchunk, ch = w - pass * nchunk;
      const int k = 8 * pass + rsl;
      const int m = (pass == 0) ? m_first : (k < nsel ? sel_s[k] : -1);
      m_out = m;
      rt_load(r, rtb + (size_t)(m < 0 ? 0 : m) * n4, ch * RT_U, cnt, m >= 0);
    };
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      mrow[j] = -1;
      if (j < total) load_item(rbuf[j], j, mrow[j]);
    }
    // ---- the eight other graphs' masks of this step (first step: the byte rows in memory)
    if (t > t0) {
      const unsigned long long w = persist_wait(pp.hist + (size_t)t * B, B, b, lane, pp.err,
                                                pp.spin_ticks);
      if (__any(!(w & PERSIST_VALID))) {
        // gave up (the grid is not fully resident), or somebody else did: the episode is void --
        // raise the flag (everybody who waits for THIS graph sees it within a few polls) and
        // leave; persistent_finalize_kernel reruns the steps from the saved state
        if (lane == 0) __hip_atomic_store(pp.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const unsigned long long wh =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, h) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)lo, h);
        msk[h] = (int)((wh >> lane) & 1ull);
      }
    }
    // ---- glimpse attention weights (lane = n): one wave-wide shift for all eight heads ----
    {
      const float loadf = (float)load0;
      float s[8], mx = -INFINITY;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        float v = sc[h] + bs[h];
        if (p.kind == VRP_KIND_IRP) v = fmaf(loadf, sld[h], v);
        v = inN ? v + (float)msk[h] : -INFINITY;
        s[h] = v;
        mx = fmaxf(mx, v);
      }
      const float M = wave_max(mx);
      float e[8], sum[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) { e[h] = inN ? exp_nonpos(s[h] - M) : 0.f; sum[h] = e[h]; }
      wave_sum8(sum);  // eight interleaved reductions
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        if (!(sum[h] > 1e-30f)) {  // wave-uniform, practically never: per-head maximum
          const float hm = wave_max(s[h]);
          e[h] = inN ? exp_nonpos(s[h] - hm) : 0.f;
          sum[h] = wave_sum(e[h]);
        }
        float r = __builtin_amdgcn_rcpf(sum[h]);
        r = fmaf(fmaf(-sum[h], r, 1.f), r, r);
        if (inN) a_s[h * N + lane] = e[h] * r;
      }
    }
    PERSIST_WAVE_SYNC();
    // ---- u_m = sum_{h,n} a[h][n] * RT[m][h][n] + cvec[m]  for selectable m ---------------
    {
      const float4 *aw = reinterpret_cast<const float4 *>(a_s) + part;
      float acc = 0.f;
      auto consume = [&](const float4 (&r)[RT_U], int w, int m) {
        const int ch = w % nchunk;
        acc = rt_dot(acc, r, aw, ch * RT_U, m >= 0 ? cnt : 0);
        if (ch == nchunk - 1) {
          acc = group8_sum(acc);
          if (part == 0 && m >= 0) u_s[m] = acc;
          acc = 0.f;
        }
      };
      for (int w = 0; w < total; w += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if (w + j < total) {
            consume(rbuf[j], w + j, mrow[j]);
            if (w + j + NB < total) load_item(rbuf[j], w + j + NB, mrow[j]);
          }
        }
      }
    }
    PERSIST_WAVE_SYNC();
    float u = -INFINITY;
    if (inN && !own_mask) u = p.clip * tanhf(u_s[lane] + cv);  // graph_decoder.py:97-98
    if (p.io.mask_trace && inN) p.io.mask_trace[((size_t)t * B + b) * N + lane] = (uint8_t)own_mask;
    if (p.io.load_trace && lane == 0) p.io.load_trace[(size_t)t * B + b] = (float)load0;

    // ---- action: lowest index among the maxima (torch CPU argmax) -------------------------
    int idx;
    float logp = 0.f;
    if (!p.sample) {
      idx = wave_argmax_lane(u);
    } else {
      // Categorical(logits=u): logits - logsumexp, probs = softmax, sample = argmax(p/q)
      const float m = wave_max(u);
      const float se = wave_sum(expf(u - m));
      const float lse = m + logf(se);
      const float l = u - lse;
      const float lm = wave_max(l);
      const float pe = expf(l - lm);
      const float ps = wave_sum(pe);
      const float ratio = inN ? (pe / ps) / q_noise : -1.f;
      idx = wave_argmax_lane(ratio);
      logp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l), idx));
    }
    idx = __builtin_amdgcn_readfirstlane(idx);
    // next step's score row: requested now, consumed after the next hand-off
    {
      const float *srow = p.SL + ((size_t)b * N + idx) * 8 * N;
#pragma unroll
      for (int h = 0; h < 8; ++h) sc[h] = srow[h * N + ln];
    }
    // ---- env.step on registers (same operation order as env_device.h) -------------------
    if (lane == idx) vis = 1;  // tsp.py:86
    const double dx = readlane_f64(xy.x, cur) - readlane_f64(xy.x, idx);
    const double dy = readlane_f64(xy.y, cur) - readlane_f64(xy.y, idx);
    const double dist = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
    double load = 1.0;
    if (p.kind == VRP_KIND_IRP) {                               // irp.py:80-86
      load = load0 - readlane_f64(dem, idx);
      if (idx == dep) load = 1.0;
    }
    const bool done = __all(vis);                               // before the fix-ups, tsp.py:95
    if (lane == dep) {
      if (idx == dep) vis = 1;                                  // tsp.py:141-142
      else if (p.kind != VRP_KIND_TSP) vis = 0;                 // vrp.py:28-31
    }
    if (__all(vis) && lane == dep) vis = 0;                     // tsp.py:145-146
    int mk = vis;
    if (p.kind == VRP_KIND_IRP && inN && dem - load > 0.0) mk = 1;  // irp.py:151-153
    const unsigned long long word = (__ballot(inN && mk) & ~PERSIST_VALID) | PERSIST_VALID;
    const bool way_back = ta >= 0;  // this step is the forced return after `done`
    const bool finish = done && (way_back || p.kind == VRP_KIND_TSP || idx == dep);
    if (lane == 0) {
      if (p.io.actions) p.io.actions[(size_t)t * B + b] = idx;
      if (p.io.step_logp) p.io.step_logp[(size_t)t * B + b] = logp;
      if (!finish)
        __hip_atomic_store(pp.hist + (size_t)(t + 1) * B + b, word, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    if (way_back) {
      ret = (float)(-dist);
      wb_cur = cur;        // where the episode ends if the batch was done at ta
      wb_load = load0;
    } else {
      accl += (float)(-dist);  // fp32 accumulate in step order, tsp_agent:85
      accp += logp;
      if (done) ta = t;
    }
    own_mask = mk;
    cur = idx;
    last = idx;
    load0 = load;
    if (finish) {
      // the mask is constant from here on: publish it for every remaining step, fill the traces
      // the way the reference's self-loops on the depot would
      for (int tt = t + 1 + lane; tt <= p.max_steps; tt += 64)
        __hip_atomic_store(pp.hist + (size_t)tt * B + b, word, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      for (int tt = t + 1; tt < p.max_steps; ++tt) {
        if (p.io.mask_trace && inN) p.io.mask_trace[((size_t)tt * B + b) * N + lane] = (uint8_t)mk;
        if (lane == 0) {
          if (p.io.load_trace) p.io.load_trace[(size_t)tt * B + b] = (float)load;
          if (p.io.actions) p.io.actions[(size_t)tt * B + b] = idx;
          if (p.io.step_logp) p.io.step_logp[(size_t)tt * B + b] = 0.f;
        }
      }
      break;
    }
  }
  // ---- state back to memory ---------------------------------------------------------------
  if (inN) p.env.visited[(size_t)b * N + lane] = (uint8_t)vis;
  if (lane == 0) {
    p.env.cur[b] = cur;
    if (p.kind == VRP_KIND_IRP) p.env.load[b] = load0;
    p.io.acc_loss[b] = accl;
    p.io.acc_logp[b] = accp;
    p.last[b] = last;
    pp.ta[b] = ta < 0 ? p.max_steps - 1 : ta;
    pp.ret[b] = ret;
    pp.wb_cur[b] = wb_cur;
    pp.wb_load[b] = wb_load;
  }
}

// ---- the first chosen node's constant part of every later score row, per graph ----------------
// base[b][h][n] = SG[b][h][n] + (Wq_first e_first)_h . (Wk e_n + bk)_h / sqrt(48)
// (graph_decoder.py:88-92: the first-node block of the context projection times the glimpse keys).
// Until round 6 two launches after step 0 -- a (B x 1024 x 128) GEMM with a row gather (the query
// folded through the keys: sixteen graphs per workgroup) and score_base_kernel -- 12.7 us at
// 512 x 20 for a few hundred thousand MACs per graph.  For the small-batch shapes (kk_floats) the
// graph's own workgroup does it: the query part first (work item = four of the 384 columns x one of
// TWO halves of k, WqfT rows coalesced, two batches of eight loads in flight: 192 KB through the
// CU's L1 per graph), then lane n takes the 48-long dot products of its wave's heads against the
// keys the prologue left in KK4 (coalesced 16-byte loads).  Called by the two- and four-wave
// persistent kernels ahead of their first step (no launch at all) and by first_base_kernel (the
// per-step path, the one-wave kernel): one function, one operation order -- `base` does not depend
// on the path an episode takes.  Also stored to memory: the in-kernel fallback reads it from there.
template <int NW, int HPW>
__device__ __forceinline__ void persist_first_base(const PersistParams &pp, int b, int h0, float (&bs)[HPW],
                                                   float *e_s, float *qp_s, float *curs_out) {
  constexpr int NTH = 64 * NW;
  const StepParams &p = pp.s;
  const int tid = threadIdx.x, lane = tid & 63, N = p.N;
  const int fb = p.first[b];
  // (the weights do not depend on the node: their first batch is in flight while the node's
  // embedding row makes its two round trips; then always one batch ahead of the one consumed)
  // work item w < 192: column quad w % 96, k half w / 96; NPASS items per thread
  constexpr int NPASS = (192 + NTH - 1) / NTH;
  const bool on0 = tid < 192;
  const int cq0 = on0 ? tid % 96 : 0, kp0 = on0 ? tid / 96 : 0;
  const float4 *wsrc0 = reinterpret_cast<const float4 *>(pp.WqfT + (size_t)(kp0 * 64) * 384) + cq0;
  float4 wv[2][8];
#pragma unroll
  for (int u = 0; u < 8; ++u) wv[0][u] = wsrc0[(size_t)u * 96];
  if (tid < 32)
    reinterpret_cast<float4 *>(e_s)[tid] =
        reinterpret_cast<const float4 *>(p.emb + ((size_t)b * N + fb) * VRP_EMB)[tid];
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int w = tid + ps * NTH;
    const bool on = w < 192;
    const int cq = on ? w % 96 : 0, kp = on ? w / 96 : 0;
    const float4 *wsrc = ps == 0 ? wsrc0 : reinterpret_cast<const float4 *>(pp.WqfT + (size_t)(kp * 64) * 384) + cq;
    if (ps > 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[0][u] = wsrc[(size_t)u * 96];
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + 1 < 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[(i + 1) & 1][u] = wsrc[(size_t)(8 * (i + 1) + u) * 96];
      }
      const float4 ea = *reinterpret_cast<const float4 *>(e_s + kp * 64 + 8 * i);
      const float4 eb = *reinterpret_cast<const float4 *>(e_s + kp * 64 + 8 * i + 4);
      const float ev[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 wq = wv[i & 1][u];
        acc.x = fmaf(ev[u], wq.x, acc.x); acc.y = fmaf(ev[u], wq.y, acc.y);
        acc.z = fmaf(ev[u], wq.z, acc.z); acc.w = fmaf(ev[u], wq.w, acc.w);
      }
    }
    if (on) reinterpret_cast<float4 *>(qp_s)[kp * 96 + cq] = acc;   // [half][column quad]
  }
  // the keys of this wave's heads: requested before the barrier the query part ends with
  const bool inN = lane < N;
  const int ln = inN ? lane : 0;
  const float c48 = 0.14433756729740643f;  // 1/sqrt(48)
  float4 kv[2][6];
  auto load_keys = [&](float4 (&dst)[6], int item) {   // item = 2 * head + half
    const float4 *kk = reinterpret_cast<const float4 *>(pp.KK4) +
                       (((size_t)b * 8 + h0 + (item >> 1)) * 12 + 6 * (item & 1)) * N + ln;
#pragma unroll
    for (int d = 0; d < 6; ++d) dst[d] = kk[(size_t)d * N];
  };
  load_keys(kv[0], 0);
  float sg[HPW], sl[HPW];
#pragma unroll
  for (int j = 0; j < HPW; ++j) {
    sg[j] = pp.SG[((size_t)b * 8 + h0 + j) * N + ln];
    sl[j] = curs_out ? p.SL[(((size_t)b * N + fb) * 8 + h0 + j) * N + ln] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < HPW; ++j) {
    const int h = h0 + j;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int item = 2 * j + half;
      if (item + 1 < 2 * HPW) load_keys(kv[(item + 1) & 1], item + 1);
#pragma unroll
      for (int d = 0; d < 6; ++d) {
        const float4 q0 = reinterpret_cast<const float4 *>(qp_s)[h * 12 + 6 * half + d];
        const float4 q1 = reinterpret_cast<const float4 *>(qp_s)[96 + h * 12 + 6 * half + d];
        const float4 k4 = kv[item & 1][d];
        float &a = (d & 1) ? a1 : a0;
        a = fmaf(q0.x + q1.x, k4.x, a); a = fmaf(q0.y + q1.y, k4.y, a);
        a = fmaf(q0.z + q1.z, k4.z, a); a = fmaf(q0.w + q1.w, k4.w, a);
      }
    }
    const float v = fmaf(a0 + a1, c48, sg[j]);
    bs[j] = v;
    const size_t o = ((size_t)b * 8 + h) * N + ln;
    if (inN) st_agent(&pp.base_out[o], v);
    if (inN && curs_out) curs_out[o] = sl[j] + v;   // step 1's complete row (last = first), latency mode
  }
}

// the same for the paths that do not run it inside a persistent grid: one workgroup per graph
__global__ __launch_bounds__(256) void first_base_kernel(PersistParams pp) {
  __shared__ __attribute__((aligned(16))) float fe_s[128];
  __shared__ __attribute__((aligned(16))) float fq_s[2 * 384];
  float bs[2];
  persist_first_base<4, 2>(pp, blockIdx.x, 2 * (threadIdx.x >> 6), bs, fe_s, fq_s, pp.s.curs);
}

// ---- the same episode loop with FOUR waves per graph (round 4) -----------------------------
// At B = 512 the kernel above puts two lone waves on every CU: a wave that has a SIMD to itself
// issues one vector instruction per ~4 cycles, and a step is ~635 of them (2 us) + the hand-off
// (PMC).  Here a graph is a 256-thread workgroup, one wave per SIMD:
//   wave w    glimpse weights of heads 2w, 2w+1 (score rows, scrambled masks, softmax);
//             the wave-wide shift M is still ONE maximum over all eight heads (exchanged
//             through LDS): results stay bit-identical to the one-wave and per-step kernels
//   wave w    the logit-table rows of passes w, w+4 (a pass = eight selectable nodes)
//   wave 0    polls the eight hand-off words, picks the action, steps the env, publishes
// Five workgroup barriers per step; everything another wave needs goes through LDS.
#define P4_BARRIER() __syncthreads()
// NW = waves per graph: 4 (heads in pairs; at most four such workgroups per CU) or 2 (heads in
// fours; for batches that would not be resident four waves wide)
#ifndef P2_WAVES
#define P2_WAVES 4   // waves per SIMD the two-wave instance is compiled for (A/B: 5 = 96 registers)
#endif
#ifndef P2_NB
#define P2_NB 2      // its table-row work items in flight
#endif
// (a real call: the fallback -- 200 registers wide -- must not take part in the episode loop's
// register allocation; it reads the parameters from the kernel argument segment)
__device__ __attribute__((noinline)) void persistent_fallback_call(const PersistParams *pp,
                                                                   float (*a_s)[8 * 64], float (*u_s)[64],
                                                                   int (*sel_s)[64]);
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 5 : P2_WAVES) void decode_persistent4_kernel(PersistParams pp) {
  constexpr int HPW = 8 / NW;   // heads per wave
  constexpr bool FOLD = NW == 4;   // a 256-thread workgroup: can run the fallback itself (fold_final)
  const StepParams &p = pp.s;
  // (FOLD: four times the rows the episode loop needs -- the fallback walks four graphs at a time)
  __shared__ __attribute__((aligned(16))) float fa_s[FOLD ? 4 : 1][8 * 64];
  __shared__ __attribute__((aligned(16))) float fu_s[FOLD ? 4 : 1][64];
  __shared__ int fsel_s[FOLD ? 4 : 1][64];
  float (&a_s)[8 * 64] = fa_s[0];           // a[h][n], hn order
  float (&u_s)[64] = fu_s[0];
  int (&sel_s)[64] = fsel_s[0];             // compacted list of selectable nodes of the step
  __shared__ unsigned long long wrd_s[8];   // the eight other graphs' mask words of the step
  __shared__ float mx_s[NW];
  __shared__ int ctl_s[6];                  // [0] chosen node, [1] 1 = finished / gave up, [2] nsel,
                                            // [3] bits of the vehicle load as fp32 (IRP), [4] 1 = this
                                            // workgroup raised `err` first, [5] 1 = it left on `err`
  __shared__ __attribute__((aligned(16))) float fe_s[128];          // persist_first_base: e_first,
  __shared__ __attribute__((aligned(16))) float fq_s[2 * 384];      // the query part's two k halves

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (pp.census) {
    // residency census (vrp_persistent_capacity4): see decode_persistent_kernel
    if (tid == 0) {
      __hip_atomic_fetch_add(&pp.census[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int seen = 0;
      for (int spins = 0; spins < 4000; ++spins) {
        seen = __hip_atomic_load(&pp.census[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen >= (int)gridDim.x) break;
        __builtin_amdgcn_s_sleep(8);
      }
      if (seen >= (int)gridDim.x)
        __hip_atomic_fetch_add(&pp.census[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const int N = p.N, B = p.B;
  const int b = blockIdx.x;
  const int t0 = p.t;
  if (p.io.notdone[t0 - 1] == 0) return;  // the batch was done before this launch
  const int n4 = 2 * N;  // float4 per RT row (8N floats)
  const int rsl = lane >> 3, part = lane & 7;
  const bool inN = lane < N;
  const int ln = inN ? lane : 0;
  const size_t row = (size_t)b * 8 * N;
  const int h0 = HPW * wave;   // this wave's heads

  // (first, with nothing else alive: its 40 registers of operands in flight would push the
  // per-graph state below into scratch for the whole episode)
  float bs[HPW];
  if (pp.fold_first) persist_first_base<NW, HPW>(pp, b, h0, bs, fe_s, fq_s, nullptr);   // (uniform over the grid)
  // ---- per-graph state.  Wave 0: the env row (lane = node); every wave: its heads' constants
  const uint8_t *mask0 = p.env.mask + (size_t)(t0 & 1) * B * N;
  int own_mask = mask0[(size_t)b * N + ln];
  int msk[HPW];
#pragma unroll
  for (int j = 0; j < HPW; ++j) msk[j] = mask0[(size_t)((b * 8 + h0 + j) % B) * N + ln];  // QUIRK D3
  float sld[HPW];
#pragma unroll
  for (int j = 0; j < HPW; ++j) {
    sld[j] = (p.kind == VRP_KIND_IRP) ? p.SLD[row + (h0 + j) * N + ln] : 0.f;
    if (!pp.fold_first) bs[j] = p.base ? p.base[row + (h0 + j) * N + ln] : 0.f;
  }
  const float cv = p.cvec[(size_t)b * N + ln];
  const double2 xy = reinterpret_cast<const double2 *>(p.env.pos)[(size_t)b * N + ln];
  int vis = inN ? p.env.visited[(size_t)b * N + ln] : 1;
  const double dem = (p.kind == VRP_KIND_IRP) ? p.env.demand[(size_t)b * N + ln] : 0.0;
  int cur = p.env.cur[b];
  const int dep = p.env.depot[b];
  double load0 = (p.kind == VRP_KIND_IRP) ? p.env.load[b] : 1.0;
  float accl = p.io.acc_loss[b], accp = p.io.acc_logp[b];
  int last = __builtin_amdgcn_readfirstlane(p.last[b]);
  float sc[HPW];  // this step's score rows SL[b][last][h] (requested at the end of the step before)
  {
    const float *srow = p.SL + ((size_t)b * N + last) * 8 * N;
#pragma unroll
    for (int j = 0; j < HPW; ++j) sc[j] = srow[(h0 + j) * N + ln];
  }
  const int cnt = (n4 - part + 7) >> 3;          // float4 of a row owned by this lane
  const int nchunk = (((n4 + 7) >> 3) + RT_U - 1) / RT_U;
  const float4 *rtb = reinterpret_cast<const float4 *>(p.RT) + (size_t)b * N * n4 + part;
  // work items in flight.  Four waves wide: one (a wave has at most two passes), which keeps the
  // kernel at 96 registers = five workgroups per CU: B = 1024 resident with a workgroup per CU to spare
  constexpr int NB = NW == 4 ? 1 : P2_NB;
  int ta = -1, wb_cur = -1;
  float ret = 0.f;
  double wb_load = 1.0;
  float loadf = (float)load0;   // the softmax's load factor (IRP); later steps: from wave 0
  // what the fallback (persistent_finalize_kernel) restarts from, should this launch fail
  if (wave == 0) {
    if (inN) st_agent(&pp.sv_visited[(size_t)b * N + lane], (uint8_t)vis);
    if (lane == 0) {
      st_agent(&pp.sv_cur[b], cur);
      st_agent(&pp.sv_last[b], last);
      st_agent(&pp.sv_load[b], load0);
      st_agent(&pp.sv_accl[b], accl);
      st_agent(&pp.sv_accp[b], accp);
    }
  }
  // the selectable list of the first step
  {
    const bool s_i = inN && !own_mask;
    const unsigned long long sel = __ballot(s_i);
    if (wave == 0) {
      if (s_i) sel_s[lanes_below(sel)] = lane;
      if (lane == 0) { ctl_s[1] = 0; ctl_s[2] = __popcll(sel); ctl_s[4] = 0; ctl_s[5] = 0; }
    }
  }
  P4_BARRIER();
  int t = t0;

  for (; t < p.max_steps; ++t) {
    const int nsel = ctl_s[2];
    const int npass = (nsel + 7) >> 3;
    // ---- loads that do not depend on other graphs: noise (wave 0), this wave's table rows
    const float q_noise = (!p.sample || wave != 0) ? 1.f
                          : p.io.noise ? p.io.noise[((size_t)t * B + b) * N + ln]
                                       : vrp_exp1_noise(p.io.noise_seed, t, b, ln);
    // this wave's work items: passes wave, wave + NW, ... (N <= 63: at most eight), chunk-major
    const int mypass = wave < npass ? (npass - wave + NW - 1) / NW : 0;
    const int total = mypass * nchunk;
    float4 rbuf[NB][RT_U];
    int mrow[NB];
    auto load_item = [&](float4 (&r)[RT_U], int w, int &m_out) {
      const int pi = w / nchunk, ch = w - pi * nchunk;
      const int k = 8 * (wave + NW * pi) + rsl;
      const int m = k < nsel ? sel_s[k] : -1;
      m_out = m;
      rt_load(r, rtb + (size_t)(m < 0 ? 0 : m) * n4, ch * RT_U, cnt, m >= 0);
    };
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      mrow[j] = -1;
      if (j < total) load_item(rbuf[j], j, mrow[j]);
    }
    // ---- the eight other graphs' masks of this step (first step: the byte rows in memory)
    if (t > t0) {
      if (wave == 0) {
        const unsigned long long w = persist_wait(pp.hist + (size_t)t * B, B, b, lane, pp.err,
                                                  pp.spin_ticks);
        if (lane < 8) wrd_s[lane] = w;
        if (__any(!(w & PERSIST_VALID))) {
          // gave up, or somebody else did: raise the flag and leave (see the one-wave kernel)
          if (lane == 0) {
            // (whoever raises it FIRST also runs the fallback when the grid finalizes itself)
            ctl_s[4] = __hip_atomic_exchange(pp.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
            ctl_s[1] = 1;
            ctl_s[5] = 1;
          }
        }
      }
      P4_BARRIER();
      if (ctl_s[1]) break;   // uniform over the workgroup
#pragma unroll
      for (int j = 0; j < HPW; ++j) msk[j] = (int)((wrd_s[h0 + j] >> lane) & 1ull);
    }
    // ---- glimpse attention weights of this wave's heads; ONE shift M for all eight heads ----
    float e[HPW];
    {
      float s[HPW], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < HPW; ++j) {
        float v = sc[j] + bs[j];
        if (p.kind == VRP_KIND_IRP) v = fmaf(loadf, sld[j], v);
        v = inN ? v + (float)msk[j] : -INFINITY;
        s[j] = v;
        mx = fmaxf(mx, v);
      }
      const float mw = wave_max(mx);
      if (lane == 0) mx_s[wave] = mw;
      P4_BARRIER();
      float M = mx_s[0];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) M = fmaxf(M, mx_s[w2]);
#pragma unroll
      for (int j = 0; j < HPW; ++j) {
        e[j] = inN ? exp_nonpos(s[j] - M) : 0.f;
        float sum = wave_sum(e[j]);
        if (!(sum > 1e-30f)) {  // wave-uniform, practically never: per-head maximum
          const float hm = wave_max(s[j]);
          e[j] = inN ? exp_nonpos(s[j] - hm) : 0.f;
          sum = wave_sum(e[j]);
        }
        float r = __builtin_amdgcn_rcpf(sum);
        r = fmaf(fmaf(-sum, r, 1.f), r, r);
        if (inN) a_s[(h0 + j) * N + lane] = e[j] * r;
      }
    }
    P4_BARRIER();
    // ---- u_m = sum_{h,n} a[h][n] * RT[m][h][n] for this wave's selectable rows ------------
    {
      const float4 *aw = reinterpret_cast<const float4 *>(a_s) + part;
      float acc = 0.f;
      auto consume = [&](const float4 (&r)[RT_U], int w, int m) {
        const int ch = w % nchunk;
        acc = rt_dot(acc, r, aw, ch * RT_U, m >= 0 ? cnt : 0);
        if (ch == nchunk - 1) {
          acc = group8_sum(acc);
          if (part == 0 && m >= 0) u_s[m] = acc;
          acc = 0.f;
        }
      };
      for (int w = 0; w < total; w += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if (w + j < total) {
            consume(rbuf[j], w + j, mrow[j]);
            if (w + j + NB < total) load_item(rbuf[j], w + j + NB, mrow[j]);
          }
        }
      }
    }
    P4_BARRIER();
    // ---- wave 0: action, the next mask (eight other graphs wait for that word), hand-over ----
    int idx = 0;
    float logp = 0.f;
    double load = 1.0;
    bool done = false, finish = false;
    int mk = 0;
    unsigned long long word = 0ull;
    if (wave == 0) {
      float u = -INFINITY;
      if (inN && !own_mask) u = p.clip * tanhf(u_s[lane] + cv);  // graph_decoder.py:97-98
      if (p.io.mask_trace && inN) st_agent(&p.io.mask_trace[((size_t)t * B + b) * N + lane], (uint8_t)own_mask);
      if (p.io.load_trace && lane == 0) st_agent(&p.io.load_trace[(size_t)t * B + b], (float)load0);
      if (!p.sample) {
        idx = wave_argmax_lane(u);
      } else {
        const float m = wave_max(u);
        const float se = wave_sum(expf(u - m));
        const float lse = m + logf(se);
        const float l = u - lse;
        const float lm = wave_max(l);
        const float pe = expf(l - lm);
        const float ps = wave_sum(pe);
        const float ratio = inN ? (pe / ps) / q_noise : -1.f;
        idx = wave_argmax_lane(ratio);
        logp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l), idx));
      }
      idx = __builtin_amdgcn_readfirstlane(idx);
      if (lane == idx) vis = 1;  // tsp.py:86
      if (p.kind == VRP_KIND_IRP) {                               // irp.py:80-86
        load = load0 - readlane_f64(dem, idx);
        if (idx == dep) load = 1.0;
      }
      done = __all(vis);                                          // before the fix-ups, tsp.py:95
      if (lane == dep) {
        if (idx == dep) vis = 1;                                  // tsp.py:141-142
        else if (p.kind != VRP_KIND_TSP) vis = 0;                 // vrp.py:28-31
      }
      if (__all(vis) && lane == dep) vis = 0;                     // tsp.py:145-146
      mk = vis;
      if (p.kind == VRP_KIND_IRP && inN && dem - load > 0.0) mk = 1;  // irp.py:151-153
      word = (__ballot(inN && mk) & ~PERSIST_VALID) | PERSIST_VALID;
      finish = done && (ta >= 0 || p.kind == VRP_KIND_TSP || idx == dep);
      if (lane == 0 && !finish)
        __hip_atomic_store(pp.hist + (size_t)(t + 1) * B + b, word, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      // what the other waves need for the next step
      const bool s_i = inN && !mk;
      const unsigned long long sel = __ballot(s_i);
      if (s_i) sel_s[lanes_below(sel)] = lane;
      if (lane == 0) {
        ctl_s[0] = idx; ctl_s[1] = finish ? 1 : 0; ctl_s[2] = __popcll(sel);
        ctl_s[3] = __builtin_bit_cast(int, (float)load);
      }
    }
    P4_BARRIER();
    last = ctl_s[0];
    const bool leave = ctl_s[1] != 0;   // finished (uniform)
    loadf = __builtin_bit_cast(float, ctl_s[3]);
    if (!leave) {
      // next step's score rows of this wave's heads: requested now, consumed after the hand-off
      const float *srow = p.SL + ((size_t)b * N + last) * 8 * N;
#pragma unroll
      for (int j = 0; j < HPW; ++j) sc[j] = srow[(h0 + j) * N + ln];
    }
    // ---- wave 0 alone: the edge's length (an fp64 square root), accumulators, traces -- while
    // the other three waves already wait at the next step's first barrier
    if (wave == 0) {
      const double dx = readlane_f64(xy.x, cur) - readlane_f64(xy.x, idx);
      const double dy = readlane_f64(xy.y, cur) - readlane_f64(xy.y, idx);
      const double dist = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
      if (lane == 0) {
        if (p.io.actions) st_agent(&p.io.actions[(size_t)t * B + b], (int64_t)idx);
        if (p.io.step_logp) st_agent(&p.io.step_logp[(size_t)t * B + b], logp);
      }
      if (ta >= 0) {           // this step was the forced return after `done`
        ret = (float)(-dist);
        wb_cur = cur;          // where the episode ends if the batch was done at ta
        wb_load = load0;
      } else {
        accl += (float)(-dist);  // fp32 accumulate in step order, tsp_agent:85
        accp += logp;
        if (done) ta = t;
      }
      own_mask = mk;
      cur = idx;
      load0 = load;
      if (finish) {
        // the mask is constant from here on: publish it for every remaining step, fill the
        // traces the way the reference's self-loops on the depot would
        for (int tt = t + 1 + lane; tt <= p.max_steps; tt += 64)
          __hip_atomic_store(pp.hist + (size_t)tt * B + b, word, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        for (int tt = t + 1; tt < p.max_steps; ++tt) {
          if (p.io.mask_trace && inN) st_agent(&p.io.mask_trace[((size_t)tt * B + b) * N + lane], (uint8_t)mk);
          if (lane == 0) {
            if (p.io.load_trace) st_agent(&p.io.load_trace[(size_t)tt * B + b], (float)load);
            if (p.io.actions) st_agent(&p.io.actions[(size_t)tt * B + b], (int64_t)idx);
            if (p.io.step_logp) st_agent(&p.io.step_logp[(size_t)tt * B + b], 0.f);
          }
        }
      }
    }
    if (leave) break;
  }
  // ---- state back to memory (wave 0 holds it) ------------------------------------------------
  if (wave == 0) {
    if (inN) st_agent(&p.env.visited[(size_t)b * N + lane], (uint8_t)vis);
    if (lane == 0) {
      st_agent(&p.env.cur[b], cur);
      if (p.kind == VRP_KIND_IRP) st_agent(&p.env.load[b], load0);
      st_agent(&p.io.acc_loss[b], accl);
      st_agent(&p.io.acc_logp[b], accp);
      st_agent(&p.last[b], cur);
      st_agent(&pp.ta[b], ta < 0 ? p.max_steps - 1 : ta);
      st_agent(&pp.ret[b], ret);
      st_agent(&pp.wb_cur[b], wb_cur);
      st_agent(&pp.wb_load[b], wb_load);
    }
  }
  // ---- a four-wave TSP grid finalizes itself: no launch behind it --------------------------------
  // What persistent_finalize_kernel does for a TSP batch that ran through is nothing but the
  // notdone flags (every graph finishes at the same step, there is no way back to add or take
  // back): graph 0's workgroup writes them.  What it does for a FAILED episode -- the fallback --
  // needs every workgroup of the grid to have left: each one raises its "left" word (hist row 0,
  // unused otherwise) once its stores are through, and the workgroup that raised `err` FIRST
  // waits for all of them and then walks the episode itself, exactly as the finalize kernel would.
  // Everything this kernel writes to memory is written by wave 0 with agent-scope (write-through)
  // stores: nothing of a workgroup that left stays dirty in its XCD's L2 to land on top of what
  // the fallback writes later from another one.  (Folding the normal path of VRP / IRP as well
  // -- "the last workgroup to leave takes the maximum" -- was measured: a release / acquire
  // fence per workgroup costs the grid 11 / 45 us, a same-address atomic with return per
  // workgroup 5.6 us at the end of an episode whose graphs finish together.)
  if constexpr (FOLD) {
    if (pp.fold_final) {
      if (wave == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);   // the stores above are through
        if (lane == 0) st_agent(pp.hist + b, 1ull);
        if (b == 0 && !ctl_s[5]) {       // ran through: T - 1 = ta (the same for every TSP graph)
          const int last_step = ta < 0 ? p.max_steps - 1 : ta;
          for (int tt = t0 + lane; tt < p.max_steps; tt += 64) st_agent(&p.io.notdone[tt], tt < last_step ? 1 : 0);
        }
      }
      if (ctl_s[4]) {   // (uniform: written before the barrier the loop was left behind)
        // bounded like every other wait of this kernel: 64 x the hand-off limit
        const long long start = wall_clock64();
        for (int g = tid; g < B; g += 64 * NW)
          while (__hip_atomic_load(pp.hist + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull &&
                 wall_clock64() - start < 64 * pp.spin_ticks)
            __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#if defined(__HIP_DEVICE_COMPILE__)
        persistent_fallback_call((const PersistParams *)__builtin_amdgcn_kernarg_segment_ptr(), fa_s, fu_s,
                                 fsel_s);
#endif
      }
    }
  }
}

// T - 1 = max_b ta_b; the way back of a graph counts iff the batch ran on after its ta
// (otherwise the episode ended with the graph on its last customer: location and load are
// put back); notdone[t] as the per-step launches would have left it.  One workgroup.
//
// If the grid raised `err` (a wave waited 20 ms for a mask word: the grid was not resident, e.g.
// another process' grid held the compute units), this workgroup IS the per-step path: it puts
// the saved state back and runs steps t0 .. of all B graphs itself, four graphs at a time, with
// the body of decode_step_rt_kernel<1, 4> (same arithmetic, same operation order: results are
// bit-identical to an undisturbed launch) and a workgroup barrier where that path has a kernel
// boundary.  Slow (one CU) and rare; never wrong, never waiting for anybody.
__device__ int32_t g_fail_sink;   // fail_host stand-in when no pinned word could be allocated
// the fallback: 256 threads, after every workgroup of the failed grid has left
__device__ __forceinline__ void persistent_fallback_body(const PersistParams &pp, float (&a_s)[4][8 * 64],
                                                         float (&u_s)[4][64], int (&sel_s)[4][64]) {
  const StepParams &p = pp.s;
  const int B = p.B, N = p.N, t0 = p.t, max_steps = p.max_steps;
  const int tid = threadIdx.x;
  // (every workgroup saved its graph's state before anything else)
  for (int i = tid; i < B * N; i += 256) p.env.visited[i] = pp.sv_visited[i];
  for (int b = tid; b < B; b += 256) {
    p.env.cur[b] = pp.sv_cur[b];
    p.last[b] = pp.sv_last[b];
    if (p.kind == VRP_KIND_IRP) p.env.load[b] = pp.sv_load[b];
    p.io.acc_loss[b] = pp.sv_accl[b];
    p.io.acc_logp[b] = pp.sv_accp[b];
  }
  __threadfence();
  __syncthreads();
  for (int t = t0; t < max_steps; ++t) {
    // batch-wide done (tsp.py:95): every thread reads the flag the step before left in L2
    if (__hip_atomic_load(&p.io.notdone[t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
      break;
    for (int g0 = 0; g0 < B; g0 += 4)
      step_rt_body<1, 4, false>(p, t, g0 + (tid >> 6), a_s, u_s, sel_s);
    __threadfence();
    __syncthreads();
  }
  if (tid == 0)
    __hip_atomic_fetch_add(pp.fail_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __attribute__((noinline)) void persistent_fallback_call(const PersistParams *pp,
                                                                   float (*a_s)[8 * 64], float (*u_s)[64],
                                                                   int (*sel_s)[64]) {
  persistent_fallback_body(*pp, *reinterpret_cast<float (*)[4][8 * 64]>(a_s),
                           *reinterpret_cast<float (*)[4][64]>(u_s), *reinterpret_cast<int (*)[4][64]>(sel_s));
}
__global__ __launch_bounds__(256) void persistent_finalize_kernel(PersistParams pp) {
  const StepParams &p = pp.s;
  const int B = p.B, t0 = p.t, max_steps = p.max_steps;
  const int tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float a_s[4][8 * 64];
  __shared__ __attribute__((aligned(16))) float u_s[4][64];
  __shared__ int sel_s[4][64];
  __shared__ int smax[4];
  if (__hip_atomic_load(pp.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
    // (every workgroup of the grid has run by now -- kernels of a stream do not overlap)
    persistent_fallback_body(pp, a_s, u_s, sel_s);
    return;
  }
  if (p.io.notdone[t0 - 1] == 0) return;
  const int32_t *__restrict__ ta = pp.ta;
  int m = 0;
  for (int b = tid; b < B; b += 256) m = max(m, ta[b]);
  m = (int)wave_max((float)m);  // step indices are small integers: exact in fp32
  if ((tid & 63) == 0) smax[tid >> 6] = m;
  __syncthreads();
  const int last_step = max(max(smax[0], smax[1]), max(smax[2], smax[3]));  // T - 1
  for (int b = tid; b < B; b += 256) {
    if (last_step > ta[b]) {
      p.io.acc_loss[b] += pp.ret[b];
    } else if (pp.wb_cur[b] >= 0) {
      p.env.cur[b] = pp.wb_cur[b];
      if (p.kind == VRP_KIND_IRP) p.env.load[b] = pp.wb_load[b];
    }
  }
  for (int t = t0 + tid; t < max_steps; t += 256) p.io.notdone[t] = t < last_step ? 1 : 0;
}

// ---- residency: how many single-wave workgroups of decode_persistent_kernel run at once ----
// Compute units a launch can actually use (a CU mask -- ROC_GLOBAL_CU_MASK, HSA_CU_MASK -- or
// a partition mode leaves fewer than hipDeviceProp_t.multiProcessorCount): a census kernel
// marks the (XCC, shader engine, CU) id every workgroup ran on.
__device__ unsigned g_census_bits[64];   // 2048 ids: XCC (4 bits) | SE (3 bits) | CU (4 bits)
__device__ int32_t g_residency[2];       // decode_persistent_kernel's census: {arrived, saw all}
__global__ __launch_bounds__(64) void cu_census_kernel() {
  if (threadIdx.x == 0) {
    const unsigned id = __smid() & 2047u;
    atomicOr(&g_census_bits[id >> 5], 1u << (id & 31));
  }
  __builtin_amdgcn_s_sleep(64);  // stay a moment: the dispatcher moves on to the other CUs
}

struct PersistDevice {
  int capacity = -1;             // resident single-wave workgroups, -1 = not measured yet
  int cus = 0;
  int retries = 0;               // censuses thrown away because they looked disturbed
  int capacity4 = -1, capacity2 = -1;   // resident workgroups of decode_persistent4_kernel<4> / <2>
  int retries4 = 0, retries2 = 0;
  hipEvent_t last = nullptr;     // end of the device's most recent persistent launch
  hipStream_t last_stream = nullptr;  // identity of that launch's stream (compared, never used)
  // ---- failures (episodes that fell back inside persistent_finalize_kernel) and the back-off
  int32_t *fail_host = nullptr;  // pinned; incremented by the device
  bool fail_tried = false;
  int fail_seen = 0;             // value of *fail_host the host last acted upon
  long long off_until_ms = 0;    // no persistent launch before this time (CLOCK_MONOTONIC)
  long long backoff_ms = 0;      // current back-off (doubles per failure, forgotten after a quiet minute)
  long long last_fail_ms = 0;
  long long spin_ticks = 0;      // 20 ms in wall_clock64() ticks
  // ---- the lease word shared by the processes that use this device
  std::atomic<unsigned long long> *lease = nullptr;
  bool lease_tried = false;
};
static PersistDevice g_pdev[VRP_MAX_DEVICES];

static long long monotonic_ms() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}

// ---- processes that share a device take turns -------------------------------------------------
// Each persistent grid is sized against the WHOLE device; two of them from two processes (two
// ranks mapped to one GPU, a notebook next to a training job) are not resident together, and the
// in-process serialiser below cannot see the other process.  The abort + fallback above keeps
// that correct; this keeps it fast: one 8-byte word per device in shared memory,
//   /dev/shm/vrpgym_hip.<uid>.<pci bus id>  =  owner pid << 40 | time of its last launch (ms),
// claimed with a compare-and-swap before every persistent launch.  A process may launch if the
// word is free, its own, or older than the lease (500 ms: the owner went idle or died -- nothing
// to clean up after a SIGKILL); otherwise it takes the one-launch-per-step path for this
// episode.  A lease that expires while the owner's launch still sits in a long queue merely costs
// that episode a fallback.  VRP_PERSISTENT_LEASE=0 turns the coordination off (tests of the
// fallback); a file that cannot be created or mapped does the same.
#define LEASE_MS 500
#define LEASE_TIME_MASK ((1ull << 40) - 1ull)
static void device_lease_open(PersistDevice &pd, int dev) {
  pd.lease_tried = true;
  const char *off = getenv("VRP_PERSISTENT_LEASE");
  if (off && off[0] == '0') return;
  char bus[64] = "";
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) != hipSuccess) {
    (void)hipGetLastError();
    snprintf(bus, sizeof(bus), "dev%d", dev);
  }
  for (char *c = bus; *c; ++c)
    if (*c == '/' || *c == ' ') *c = '_';
  char path[160];
  snprintf(path, sizeof(path), "/dev/shm/vrpgym_hip.%u.%s", (unsigned)getuid(), bus);
  // /dev/shm is world-writable and the name is predictable: never follow a link somebody planted
  // there, accept only a plain file of our own with one name, and grow it -- never shrink it
  const int fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC | O_NOFOLLOW, 0600);
  if (fd < 0) return;
  struct stat sb;
  void *m = MAP_FAILED;
  if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_uid == getuid() && sb.st_nlink == 1 &&
      (sb.st_size >= 8 || ftruncate(fd, 8) == 0))
    m = mmap(nullptr, 8, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return;
  pd.lease = reinterpret_cast<std::atomic<unsigned long long> *>(m);
}
// true: this process may launch a persistent grid on the device now (and has renewed the lease)
static bool device_lease_claim(PersistDevice &pd, int dev) {
  if (!pd.lease_tried) device_lease_open(pd, dev);
  if (!pd.lease) return true;
  const unsigned long long me = (unsigned long long)(unsigned)getpid() & 0xFFFFFFull;
  const unsigned long long now = (unsigned long long)monotonic_ms() & LEASE_TIME_MASK;
  unsigned long long w = pd.lease->load(std::memory_order_relaxed);
  for (int tries = 0; tries < 4; ++tries) {
    const unsigned long long owner = w >> 40, stamp = w & LEASE_TIME_MASK;
    const bool mine = owner == me, free_ = owner == 0 || stamp > now || now - stamp > LEASE_MS;
    if (!mine && !free_) return false;
    if (pd.lease->compare_exchange_weak(w, me << 40 | now, std::memory_order_acq_rel)) return true;
  }
  return false;
}
// a clean exit hands the device over at once
__attribute__((destructor)) static void device_lease_release_all() {
  const unsigned long long me = (unsigned long long)(unsigned)getpid() & 0xFFFFFFull;
  for (int d = 0; d < VRP_MAX_DEVICES; ++d) {
    std::atomic<unsigned long long> *l = g_pdev[d].lease;
    if (!l) continue;
    unsigned long long w = l->load(std::memory_order_relaxed);
    if ((w >> 40) == me) (void)l->compare_exchange_strong(w, 0ull);
  }
}

// the pinned failure counter and the spin limit of a device (first use: outside any capture)
static void persist_device_init(PersistDevice &pd, int dev) {
  if (pd.fail_tried) return;
  pd.fail_tried = true;
  if (hipHostMalloc((void **)&pd.fail_host, 64, hipHostMallocDefault) == hipSuccess) {
    *pd.fail_host = 0;
  } else {
    (void)hipGetLastError();
    pd.fail_host = nullptr;
  }
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
    (void)hipGetLastError();
    khz = 100000;   // gfx9: s_memrealtime counts at 100 MHz
  }
  const char *ms = getenv("VRP_PERSISTENT_SPIN_MS");   // tests
  const int limit_ms = ms && atoi(ms) > 0 ? atoi(ms) : 20;
  pd.spin_ticks = (long long)khz * limit_ms;
}

// Failures seen since the last look -> stay off the persistent path for a while: 100 ms after the
// first, doubling up to 10 s while they keep coming, forgotten after a minute without one.
// (Locked by the caller.)
static bool persist_backed_off(PersistDevice &pd) {
  if (!pd.fail_host) return false;
  const long long now = monotonic_ms();
  const int seen = *(volatile int32_t *)pd.fail_host;
  if (seen != pd.fail_seen) {
    pd.fail_seen = seen;
    if (now - pd.last_fail_ms > 60000) pd.backoff_ms = 0;
    pd.backoff_ms = pd.backoff_ms ? (pd.backoff_ms * 2 > 10000 ? 10000 : pd.backoff_ms * 2) : 100;
    pd.last_fail_ms = now;
    pd.off_until_ms = now + pd.backoff_ms;
  }
  return now < pd.off_until_ms;
}
// (process-wide state besides the thread-local error string: the measured capacity per device and
// the stream of its last persistent launch; launches from several host threads take this lock)
#include <mutex>
static std::mutex g_pdev_lock;

static int persistent_capacity_of(int dev, hipStream_t capturing_guard) {
  if (dev < 0 || dev >= VRP_MAX_DEVICES) return 0;
  std::lock_guard<std::mutex> guard(g_pdev_lock);
  PersistDevice &pd = g_pdev[dev];
  if (pd.capacity >= 0) return pd.capacity;
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_persistent_kernel, 64, 0) !=
      hipSuccess) { (void)hipGetLastError(); return 0; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  cus = prop.multiProcessorCount;
  // (asked for the legacy stream too -- capturing_guard == nullptr: it reports a capture that is
  // active on another stream of this thread as an error, which counts as "capturing" here)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(capturing_guard, &cs) != hipSuccess) {
    (void)hipGetLastError();
    cs = hipStreamCaptureStatusActive;
  }
  if (cs == hipStreamCaptureStatusNone) {
    persist_device_init(pd, dev);
    // the census synchronises: never inside a stream capture (the figure of the device
    // properties serves until an eager call gets here).  It must not compete for CU slots with
    // work this process has in flight (a rollout's encoder enqueued just before the first
    // eligibility check): drain the device first.  vrp-gym_amd measures when it loads the library
    // (vrpgym_hip.require_gpu), before anything is enqueued.
    if (capturing_guard) (void)hipStreamSynchronize(capturing_guard);
    else (void)hipDeviceSynchronize();
    unsigned zero[64] = {0}, bits[64];
    hipStream_t st = nullptr;
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMemcpyToSymbolAsync(HIP_SYMBOL(g_census_bits), zero, sizeof(zero), 0,
                                      hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok) {
      hipLaunchKernelGGL(cu_census_kernel, dim3(16 * cus), dim3(64), 0, st);
      ok = hipGetLastError() == hipSuccess;
    }
    ok = ok && hipMemcpyFromSymbolAsync(bits, HIP_SYMBOL(g_census_bits), sizeof(bits), 0,
                                        hipMemcpyDeviceToHost, st) == hipSuccess;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    if (st) (void)hipStreamDestroy(st);
    if (!ok) { (void)hipGetLastError(); return 0; }
    int seen = 0;
    for (int i = 0; i < 64; ++i) seen += __builtin_popcount(bits[i]);
    if (seen > 0 && seen < cus) cus = seen;
    // The occupancy query can be one block per CU high (MI355X_MICROARCH.md, residency): the
    // kernel itself is the judge.  A census launch of cus x per_cu workgroups in which every
    // workgroup saw all the others proves that such a grid is resident at once; otherwise one
    // workgroup per CU less, and so on.
    per_cu = per_cu > 32 ? 32 : per_cu;
    pd.cus = cus;
    int32_t *res = nullptr;
    hipStream_t st2 = nullptr;
    if (hipGetSymbolAddress((void **)&res, HIP_SYMBOL(g_residency)) != hipSuccess ||
        hipStreamCreateWithFlags(&st2, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      pd.capacity = cus * (per_cu > 1 ? per_cu - 1 : 0);
      return pd.capacity;
    }
    const int query_per_cu = per_cu;
    int measured = 0;
    for (int k = per_cu; k >= 1 && measured == 0; --k) {
      int32_t zero2[2] = {0, 0}, got[2] = {0, 0};
      PersistParams cp = {};
      cp.census = res;
      bool ok2 = hipMemcpyAsync(res, zero2, sizeof(zero2), hipMemcpyHostToDevice, st2) == hipSuccess;
      if (ok2) {
        hipLaunchKernelGGL(decode_persistent_kernel, dim3(cus * k), dim3(64), 0, st2, cp);
        ok2 = hipGetLastError() == hipSuccess;
      }
      ok2 = ok2 && hipMemcpyAsync(got, res, sizeof(got), hipMemcpyDeviceToHost, st2) == hipSuccess;
      ok2 = ok2 && hipStreamSynchronize(st2) == hipSuccess;
      if (!ok2) { (void)hipGetLastError(); break; }
      if (got[1] == cus * k) measured = cus * k;
    }
    (void)hipStreamDestroy(st2);
    // The query is at most one workgroup per CU high.  A census that lost more than that was
    // disturbed (another process or stream held CU slots while it ran): use the figure for this
    // call but do not keep it -- the next call measures again (at most a few times).
    if (measured < cus * (query_per_cu - 1) && pd.retries < 4) {
      ++pd.retries;
      pd.capacity = -1;
      return measured;
    }
    pd.capacity = measured;
    return pd.capacity;
  }
  per_cu = per_cu > 32 ? 32 : per_cu;
  return cus * (per_cu > 1 ? per_cu - 1 : 0);
}

// compute units a launch can use on the current device (the census' first half); 0 if unknown
int vrp_usable_cus(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  (void)persistent_capacity_of(dev, st);
  if (dev < 0 || dev >= VRP_MAX_DEVICES) return 0;
  std::lock_guard<std::mutex> guard(g_pdev_lock);
  return g_pdev[dev].cus;
}

// Workgroups of decode_persistent4_kernel<NW> the device keeps resident at once: the same census
// (every workgroup must see all the others).
template <int NW>
static int persistent_wide_capacity_of(int dev, hipStream_t capturing_guard) {
  if (dev < 0 || dev >= VRP_MAX_DEVICES) return 0;
  (void)persistent_capacity_of(dev, capturing_guard);   // the usable CUs
  std::lock_guard<std::mutex> guard(g_pdev_lock);
  PersistDevice &pd = g_pdev[dev];
  int &cap = NW == 4 ? pd.capacity4 : pd.capacity2;
  int &retries = NW == 4 ? pd.retries4 : pd.retries2;
  if (cap >= 0) return cap;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(capturing_guard, &cs) != hipSuccess) {
    (void)hipGetLastError();
    cs = hipStreamCaptureStatusActive;
  }
  int per_cu = 0;
  if (pd.cus <= 0 || cs != hipStreamCaptureStatusNone ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_persistent4_kernel<NW>, 64 * NW,
                                                   0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  per_cu = per_cu > 16 ? 16 : per_cu;
  int32_t *res = nullptr;
  hipStream_t st2 = nullptr;
  if (hipGetSymbolAddress((void **)&res, HIP_SYMBOL(g_residency)) != hipSuccess ||
      hipStreamCreateWithFlags(&st2, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (capturing_guard) (void)hipStreamSynchronize(capturing_guard);
  else (void)hipDeviceSynchronize();
  const int query = per_cu;
  int measured = 0;
  for (int k = per_cu; k >= 1 && measured == 0; --k) {
    int32_t zero2[2] = {0, 0}, got[2] = {0, 0};
    PersistParams cp = {};
    cp.census = res;
    bool ok2 = hipMemcpyAsync(res, zero2, sizeof(zero2), hipMemcpyHostToDevice, st2) == hipSuccess;
    if (ok2) {
      hipLaunchKernelGGL(decode_persistent4_kernel<NW>, dim3(pd.cus * k), dim3(64 * NW), 0, st2, cp);
      ok2 = hipGetLastError() == hipSuccess;
    }
    ok2 = ok2 && hipMemcpyAsync(got, res, sizeof(got), hipMemcpyDeviceToHost, st2) == hipSuccess;
    ok2 = ok2 && hipStreamSynchronize(st2) == hipSuccess;
    if (!ok2) { (void)hipGetLastError(); break; }
    if (got[1] == pd.cus * k) measured = pd.cus * k;
  }
  (void)hipStreamDestroy(st2);
  if (measured < pd.cus * (query - 1) && retries < 4) { ++retries; return measured; }
  cap = measured;
  return measured;
}

// waves per graph of the persistent launch: 4 where every four-wave workgroup of the batch is
// resident, else 2, else 1 (decode_persistent_kernel)
int vrp_persistent_width(int kind, int B, int N, int max_steps, int flags, const vrp_rollout_io *io,
                         hipStream_t st) {
  if (!vrp_persistent_eligible(kind, B, N, max_steps, flags, io, st)) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VRP_MAX_DEVICES) {
    (void)hipGetLastError();
    return 0;
  }
  {
    // recent episodes fell back (something else holds the compute units), or another process
    // holds the device's lease: one launch per step for this episode
    std::lock_guard<std::mutex> guard(g_pdev_lock);
    PersistDevice &pd = g_pdev[dev];
    if (persist_backed_off(pd)) return 0;
    if (!device_lease_claim(pd, dev)) return 0;
  }
  const char *forced = getenv("VRP_PERSISTENT_WAVES");   // tests, A/B: "1", "2" or "4" (read per call)
  if (forced && forced[0] == '1') return 1;
  static const bool force = getenv("VRP_PERSISTENT_FORCE") != nullptr;
  // One workgroup per CU of slack below the measured capacity: at 128 registers four waves fill a
  // SIMD's register file exactly, and a grid sized to the last slot (B = 2048 two waves wide:
  // 8 workgroups on every CU) stopped being resident as soon as another stream's kernels had
  // fragmented a register file -- tests/test_gpu_persistent_guard.py, two streams: timeouts.
  const int cus = force ? 0 : vrp_usable_cus(st);
  if (!(forced && forced[0] == '2'))
    if (force ? B <= 768 : B <= persistent_wide_capacity_of<4>(dev, st) - cus) return 4;
  if (forced && forced[0] == '4') return 1;   // asked for four, not resident: the one-wave kernel
  if (force ? B <= 1792 : B <= persistent_wide_capacity_of<2>(dev, st) - cus) return 2;
  return 1;
}

// episodes of the current device that a persistent launch of this process could not finish and
// persistent_finalize_kernel reran (results unaffected): diagnostics and tests
extern "C" int vrp_persistent_failures(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VRP_MAX_DEVICES) {
    (void)hipGetLastError();
    return 0;
  }
  std::lock_guard<std::mutex> guard(g_pdev_lock);
  return g_pdev[dev].fail_host ? *(volatile int32_t *)g_pdev[dev].fail_host : 0;
}

extern "C" int vrp_persistent_capacity(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  const int c = persistent_capacity_of(dev, nullptr);
  (void)persistent_wide_capacity_of<4>(dev, nullptr);   // measured at the same quiet moment
  (void)persistent_wide_capacity_of<2>(dev, nullptr);
  return c;
}

bool vrp_persistent_eligible(int kind, int B, int N, int max_steps, int flags,
                             const vrp_rollout_io *io, hipStream_t st) {
  (void)kind;
  static const bool off = getenv("VRP_NO_PERSISTENT") != nullptr;  // A/B aid
  if (off || N < 3 || N > 63 || max_steps < 2 || io->logits || io->forced ||
      (flags & (VRP_STEP_TILE_KERNEL | VRP_STEP_THROUGHPUT_KERNEL | VRP_STEP_DECODE_ONLY |
                VRP_STEP_TABLE_KERNEL | VRP_STEP_NO_PERSISTENT)))
    return false;
  if (B > 2048) return false;  // beyond this the one-launch-per-step kernel is the faster one
  if (max_steps + 1 > 2 * N) return false;  // hist holds 2N rows (decoder_ws.h: hist_rows)
  static const bool force = getenv("VRP_PERSISTENT_FORCE") != nullptr;  // tests: skip the
  if (force) return true;                                               // residency check
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  // decided BEFORE the episode starts: a grid that might not be resident takes the per-step path
  return B <= persistent_capacity_of(dev, st);
}

// Two persistent grids of one device must not overlap (each is sized against the whole
// device: together they might not be resident, and the resident waves of both would wait
// for words of workgroups that were never scheduled).  As long as every persistent launch of
// the device comes from ONE stream, stream order is the guarantee and nothing is added to the
// queue.  The first launch from a second stream waits on the host for the first stream (once per
// process); from then on every launch records an event and a launch on another stream than the
// previous one waits for it.  (Captured streams are left alone: a hipGraph replays on one
// stream; do not replay persistent rollouts of one device on two streams at once.)
void vrp_persistent_serialize_begin(hipStream_t st, void **token) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cs);
  PersistDevice *pd = (dev >= 0 && dev < VRP_MAX_DEVICES && cs == hipStreamCaptureStatusNone)
                          ? &g_pdev[dev] : nullptr;
  std::lock_guard<std::mutex> guard(g_pdev_lock);
  // the previous persistent launch of this device came from another stream: wait for the event
  // recorded behind it (an event outlives its stream; no handle of a possibly destroyed stream
  // is ever used)
  if (pd && pd->last && pd->last_stream != st) (void)hipStreamWaitEvent(st, pd->last, 0);
  *token = pd;
}
void vrp_persistent_serialize_end(hipStream_t st, void *token) {
  PersistDevice *pd = (PersistDevice *)token;
  if (!pd) return;
  std::lock_guard<std::mutex> guard(g_pdev_lock);
  pd->last_stream = st;  // compared only, never dereferenced
  if (!pd->last && hipEventCreateWithFlags(&pd->last, hipEventDisableTiming) != hipSuccess) {
    pd->last = nullptr;
    (void)hipGetLastError();
  }
  if (pd->last) (void)hipEventRecord(pd->last, st);
}

static PersistParams make_persist_params(const StepParams &sp, void *workspace) {
  DecWs ws = carve_decws(workspace, sp.B, sp.N);
  PersistParams pp;
  pp.s = sp;
  pp.hist = ws.hist;
  pp.ta = ws.ta;
  pp.ret = ws.ret;
  pp.wb_cur = ws.wb_cur;
  pp.wb_load = ws.wb_load;
  pp.err = ws.err;
  pp.census = nullptr;
  pp.sv_visited = ws.sv_visited;
  pp.sv_cur = ws.sv_cur;
  pp.sv_last = ws.sv_last;
  pp.sv_load = ws.sv_load;
  pp.sv_accl = ws.sv_accl;
  pp.sv_accp = ws.sv_accp;
  pp.spin_ticks = 2000000;   // 20 ms at 100 MHz
  pp.fail_host = nullptr;
  pp.fold_first = 0;
  pp.fold_final = 0;
  pp.WqfT = nullptr; pp.KK4 = ws.KK4; pp.SG = ws.SG; pp.base_out = ws.base;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < VRP_MAX_DEVICES) {
    std::lock_guard<std::mutex> guard(g_pdev_lock);
    if (g_pdev[dev].spin_ticks > 0) pp.spin_ticks = g_pdev[dev].spin_ticks;
    pp.fail_host = g_pdev[dev].fail_host;
  } else {
    (void)hipGetLastError();
  }
  if (!pp.fail_host && hipGetSymbolAddress((void **)&pp.fail_host, HIP_SYMBOL(g_fail_sink)) != hipSuccess) {
    (void)hipGetLastError();
    pp.fail_host = ws.err + 1;   // (the flag's 256-byte slot)
  }
  return pp;
}

// steps sp.t .. max_steps-1 (sp.t >= 1: step 0 and the first-node fold have run)
// vrp_decode_first_row for the shapes whose keys the prologue kept (kk_floats): one launch
int vrp_launch_first_base(int kind, const void *derived, int B, int N, const float *emb,
                          void *workspace, hipStream_t st) {
  DecWs ws = carve_decws(workspace, B, N);
  StepParams sp = {};
  sp.kind = kind; sp.B = B; sp.N = N; sp.emb = emb;
  sp.SL = ws.SL; sp.curs = ws.curs; sp.first = ws.first; sp.last = ws.last;
  PersistParams pp = make_persist_params(sp, workspace);
  if (!pp.KK4) { vrp_set_error("first_base: no keys kept for B=%d N=%d", B, N); return 1; }
  pp.fold_first = 1;
  pp.WqfT = carve_derived(const_cast<void *>(derived)).WqfT;
  hipLaunchKernelGGL(first_base_kernel, dim3(sp.B), dim3(256), 0, st, pp);
  VRP_CHECK_LAUNCH("first_base");
  return 0;
}

// may the persistent launch that follows step 0 fold the first node into the score rows itself
// (then step 0 runs with VRP_STEP_NO_FIRST_ROW)?  Two- and four-wave grids, TSP / VRP.
bool vrp_persistent_folds_first(int kind, int B, int N, int waves, int flags) {
  static const bool off = getenv("VRP_NO_FOLD_FIRST") != nullptr;   // A/B aid
  return !off && waves >= 2 && kind != VRP_KIND_IRP && kk_floats(B, N) > 0 &&
         !(flags & VRP_STEP_NO_FIRST_ROW);
}

int vrp_launch_persistent_steps(const StepParams &sp, void *workspace, hipStream_t st,
                                int waves, const void *derived_for_first) {
  PersistParams pp = make_persist_params(sp, workspace);
  if (derived_for_first) {
    if (waves < 2 || !pp.KK4 || sp.t != 1) {
      vrp_set_error("persistent_steps: first-node fold asked for waves=%d t=%d", waves, sp.t);
      return 1;
    }
    pp.fold_first = 1;
    pp.WqfT = carve_derived(const_cast<void *>(derived_for_first)).WqfT;
  }
  // (the hand-off words and the error flag were cleared by vrp_decode_prologue: one persistent
  // launch per episode)
  // a four-wave TSP grid finalizes itself (VRP_PERSISTENT_FINALIZE_LAUNCH=1: A/B aid, keeps the launch)
  static const bool keep_launch = getenv("VRP_PERSISTENT_FINALIZE_LAUNCH") != nullptr;
  pp.fold_final = (waves == 4 && sp.kind == VRP_KIND_TSP && !keep_launch) ? 1 : 0;
  void *token = nullptr;
  vrp_persistent_serialize_begin(st, &token);
  if (waves == 4) hipLaunchKernelGGL(decode_persistent4_kernel<4>, dim3(sp.B), dim3(256), 0, st, pp);
  else if (waves == 2) hipLaunchKernelGGL(decode_persistent4_kernel<2>, dim3(sp.B), dim3(128), 0, st, pp);
  else hipLaunchKernelGGL(decode_persistent_kernel, dim3(sp.B), dim3(64), 0, st, pp);
  VRP_CHECK_LAUNCH("decode_persistent");
  if (!pp.fold_final) {
    hipLaunchKernelGGL(persistent_finalize_kernel, dim3(1), dim3(256), 0, st, pp);
    VRP_CHECK_LAUNCH("persistent_finalize");
  }
  vrp_persistent_serialize_end(st, token);
  return 0;
}

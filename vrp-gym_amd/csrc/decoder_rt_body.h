// The table-driven step (decode_step_rt_kernel) as a device function: shared by the per-step
// kernels of decoder.hip and the persistent path's in-kernel fallback (decoder_persistent.hip).
#pragma once
#include "decoder_step.h"

// Batch-wide "somebody is not done" flag.  Only zero / non-zero matters, the readers are
// the NEXT launch and the host, and every writer stores the same value, so an unfinished
// graph issues one plain store that the L2 merges with everybody else's.  Measured
// alternatives at B = 8192: a same-address atomic RMW per graph 90 us per launch, an
// agent-scope atomic store 240 us (each one serialises at the memory side), reading the flag
// first to skip the store +2..4 us (a round trip at the very end of every wave).
__device__ __forceinline__ void flag_notdone(int32_t *flag) { *flag = 1; }

// ------------------------------------------------------------------ the step kernel

// ---------------------------------------------------------------- table-driven step (N <= 64)
// One wave per graph, no weight matrix and no embedding tile.  Per step a graph streams
//   ONE (8,N) glimpse score row, 9 mask rows, its coordinate/visited/demand rows and the
//   rows RT[b][m][:][:] (8N floats each) of the pointer-logit table -- ONLY for nodes m that
//   are still selectable: masked logits are -inf whatever their value (graph_decoder.py:98),
//   so their rows are never read.  Averaged over a TSP episode that halves the traffic.
// Latency structure (the kernel is launch- and latency-bound at small batch):
//   * every load that does not depend on the chosen action is issued at kernel entry
//     (incl. the first RT rows and the whole env row: lane n holds node n's
//     coordinates/visited/demand; the action's and the current node's values are then
//     fetched with readlane instead of dependent loads);
//   * the score row is the table row SL[b][last[b]]: one dependent load at entry (last[b]
//     comes from the previous launch), issued together with the independent RT rows;
//     nothing is read or written after the action is known except the env commits;
//   * the batch-wide done flag is read with everything else and only gates the commits.
// RT pass: lane = (row slot r = lane>>3, part q = lane&7).  The k-th selectable node
// (k = 8*pass + r) is found by ballot/prefix; the 8 parts split its row of 2N float4;
// every wave-level load is 8 x 128 contiguous bytes.
// WPG = waves (= graphs) per workgroup: 4 for large batches; 1 for small ones, where the
// kernel is latency-bound and single-wave workgroups spread over more CUs and never wait
// for a sibling wave at the two barriers.
// Barrier between a wave's LDS writes (a_s, u_s) and their reads.  a_s / u_s / sel_s are private
// to a wave, and the LDS operations of one wave execute in program order: all that is needed is to
// keep the compiler from reordering them.  Single-wave workgroups therefore skip s_barrier --
// __syncthreads would also drain every global load in flight (vmcnt(0)); multi-wave workgroups
// keep it (their waves share nothing either, but the round-1 tuning was done with it in place).
template <int WPG>
__device__ __forceinline__ void step_sync() {
#ifdef VRP_RT_NO_BARRIER   // experiment: no workgroup barrier in the four-wave instances either
  constexpr bool wave_only = true;
#else
  constexpr bool wave_only = WPG == 1;
#endif
  if constexpr (wave_only) {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
  } else {
    __syncthreads();
  }
}

// The body of one step for one graph per wave: the per-step kernel below runs it once; the
// persistent path's in-kernel fallback (decoder_persistent.hip: one workgroup walks the whole
// episode after a failed hand-off) calls it for every (step, graph).  `t` = step, `braw` = this
// wave's graph (>= B: idle wave), CHECK_DONE: leave at once if the batch was done before step t
// (the fallback's loop checks that itself, with an agent-scope load).
template <int NPL, int WPG, bool CHECK_DONE>
__device__ __forceinline__ void step_rt_body(const StepParams &p, const int t, const int braw,
                                             float (&a_s)[WPG][8 * 64 * NPL],
                                             float (&u_s)[WPG][64 * NPL],
                                             int (&sel_s)[WPG][64 * NPL]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = p.N, B = p.B;
  const bool active = braw < B;
  const int b = __builtin_amdgcn_readfirstlane(active ? braw : B - 1);
  const int par = t & 1;
  const uint8_t *mask_in = p.env.mask + (size_t)par * B * N;
  uint8_t *mask_out = p.env.mask + (size_t)(par ^ 1) * B * N;
  const int n4 = 2 * N;  // float4 per RT row (8N floats)
  const int rsl = lane >> 3, part = lane & 7;
  bool inN[NPL];
  int ln[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) { inN[i] = lane + 64 * i < N; ln[i] = inN[i] ? lane + 64 * i : 0; }


  // the batch was done before this launch (tsp.py:95): nothing to commit, and a fixed-length
  // loop of 2(N-1) launches spends its tail here (VRP-100 x 2048: 86 of 198 launches, 10.7 us
  // each when they decoded first, 4.3 us now).  One scalar round trip ahead of the loads below
  // (checked behind them, a no-op launch still fetched its 9 KB of score rows per graph).
  // Workgroup-uniform, ahead of any barrier.
  if (CHECK_DONE && !p.decode_only && t > 0 && p.io.notdone[t - 1] == 0) return;
  // ---- entry: issue every action-independent load --------------------------------
  const size_t row = (size_t)b * 8 * N;
  // this step's complete glimpse score row: row0 at t = 0, else table row SL[b][last]
  // (one dependent load: last[b] was written by the previous launch)
  // WPG == 1 is the latency mode of small batches: there the previous launch already copied
  // its table row into `curs`, so no launch starts with a pointer chase.
  // (N > 80 only: the constant part lives in a second row, base[b]; the latency mode gets
  // the sum pre-added)
  const float *srow = p.row0 + row;
  const bool add_base = p.base && WPG != 1 && t > 0;
  if (t > 0) {
    if (WPG == 1) {
      srow = p.curs + row;
    } else {
      const int last = __builtin_amdgcn_readfirstlane(p.last[b]);
      srow = p.SL + ((size_t)b * N + last) * 8 * N;
    }
  }
  int own_mask[NPL];
  float sc[NPL][8], bs[NPL][8], sld[NPL][8], cv[NPL], q_noise[NPL];
  int msk[NPL][8];
  double2 xy[NPL];
  int vis[NPL];
  double dem[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    own_mask[i] = mask_in[(size_t)b * N + ln[i]];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      sc[i][h] = srow[h * N + ln[i]];
      // (added at the softmax, not here: an add at this point would wait for this load and,
      // vmcnt being in order, for every load issued before it)
      bs[i][h] = add_base ? p.base[row + h * N + ln[i]] : 0.f;
      sld[i][h] = (p.kind == VRP_KIND_IRP) ? p.SLD[row + h * N + ln[i]] : 0.f;
      msk[i][h] = mask_in[(size_t)((b * 8 + h) % B) * N + ln[i]];  // QUIRK D3: other graphs
    }
    cv[i] = p.cvec[(size_t)b * N + ln[i]];
    xy[i] = make_double2(0.0, 0.0);
    vis[i] = 1;
    dem[i] = 0.0;
    if (!p.decode_only) {  // env row (lane = node)
      xy[i] = reinterpret_cast<const double2 *>(p.env.pos)[(size_t)b * N + ln[i]];
      if (inN[i]) vis[i] = p.env.visited[(size_t)b * N + ln[i]];
      if (p.kind == VRP_KIND_IRP) dem[i] = p.env.demand[(size_t)b * N + ln[i]];
    }
    q_noise[i] = !p.sample ? 1.f
                 : p.io.noise ? p.io.noise[((size_t)t * B + b) * N + ln[i]]
                              : vrp_exp1_noise(p.io.noise_seed, t, b, ln[i]);
#ifdef VRP_MUTATION_NOISE_SHIFT  // test-the-tests build: off-by-one noise index (NPL = 2 kernels)
    if (NPL > 1 && p.sample && p.io.noise) q_noise[i] = p.io.noise[((size_t)t * B + b) * N + (ln[i] + 1) % N];
#endif
  }
  const int cur = p.decode_only ? 0 : p.env.cur[b];
  const int dep = p.decode_only ? 0 : p.env.depot[b];
  const double load0 = (p.kind == VRP_KIND_IRP) ? p.env.load[b] : 1.0;
  float accl = 0.f, accp = 0.f;
  if (!p.decode_only) { accl = p.io.acc_loss[b]; accp = p.io.acc_logp[b]; }
  // selectable nodes (own mask == 0); their RT rows are the only ones fetched
  unsigned long long sel[NPL];
  int nsel = 0;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const bool s_i = inN[i] && !own_mask[i];
    sel[i] = __ballot(s_i);
    if (s_i) sel_s[wave][nsel + __popcll(sel[i] & ((1ull << lane) - 1ull))] = lane + 64 * i;
    nsel += __popcll(sel[i]);
  }
  const int cnt = (n4 - part + 7) >> 3;          // float4 of a row owned by this lane
  const int nchunk = (((n4 + 7) >> 3) + RT_U - 1) / RT_U;
  const int total = ((nsel + 7) >> 3) * nchunk;  // work items (pass, chunk), wave-uniform
  const float4 *rtb = reinterpret_cast<const float4 *>(p.RT) + (size_t)b * N * n4 + part;
  // measured (tools/step_probe.py): three items in flight are best at 8192 graphs, two in
  // the latency mode (512..2048 graphs)
  constexpr int NB = (WPG == 1) ? 2 : RT_NB;
  float4 rbuf[NB][RT_U];
  int mrow[NB];
  int m_first = -1;  // pass 0 rows (k = rsl < 8)
  if (rsl < nsel) {
    const int c0 = __popcll(sel[0]);
    m_first = (NPL == 1 || rsl < c0) ? kth_set_bit(sel[0], rsl)
                                     : 64 + kth_set_bit(sel[NPL - 1], rsl - c0);
  }
  // the first NB work items (up to 12 selectable rows at N = 40) are in flight while the wave
  // computes the glimpse softmax below: a wave's life is a chain of memory round trips,
  // and the kernel's duration is that chain times the number of wave rounds per SIMD
  auto load_item = [&](float4 (&r)[RT_U], int w, int &m_out) {
    const int pass = w / nchunk, ch = w - pass * nchunk;
    const int k = 8 * pass + rsl;
    const int m = (pass == 0) ? m_first : (k < nsel ? sel_s[wave][k] : -1);
    m_out = m;
    rt_load(r, rtb + (size_t)(m < 0 ? 0 : m) * n4, ch * RT_U, cnt, m >= 0);
  };
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    mrow[j] = -1;
    if (j < total) load_item(rbuf[j], j, mrow[j]);
  }

  // ---- glimpse attention weights (lane = n) -----------------------------------------
  // softmax_n(s_h) is invariant to the shift, so ONE wave-wide maximum over all eight
  // heads replaces eight per-head ones (exp arguments stay <= 0); exp is a compensated
  // exp2 and the normalisation a multiplication by a Newton-refined reciprocal.  This
  // phase is VALU-bound (every wave of the chip runs it at the same time, before any
  // row of the logit table can be consumed): ~23 instead of ~92 instructions per head.
  // Should a head lie so far below the global maximum that its sum underflows, the wave
  // redoes that head with its own maximum.
  {
    const float loadf = (float)load0;
    float s[NPL][8], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPL; ++i)
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        float v = sc[i][h] + bs[i][h];
        if (p.kind == VRP_KIND_IRP) v = fmaf(loadf, sld[i][h], v);
        v = inN[i] ? v + (float)msk[i][h] : -INFINITY;
        s[i][h] = v;
        mx = fmaxf(mx, v);
      }
    const float M = wave_max(mx);
    float e[NPL][8], sum[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      sum[h] = 0.f;
#pragma unroll
      for (int i = 0; i < NPL; ++i) { e[i][h] = inN[i] ? exp_nonpos(s[i][h] - M) : 0.f; sum[h] += e[i][h]; }
    }
    wave_sum8(sum);  // eight interleaved reductions
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      if (!(sum[h] > 1e-30f)) {  // wave-uniform, practically never: per-head maximum
        float hm = -INFINITY;
#pragma unroll
        for (int i = 0; i < NPL; ++i) hm = fmaxf(hm, s[i][h]);
        hm = wave_max(hm);
        float es = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) { e[i][h] = inN[i] ? exp_nonpos(s[i][h] - hm) : 0.f; es += e[i][h]; }
        sum[h] = wave_sum(es);
      }
      float r = __builtin_amdgcn_rcpf(sum[h]);
      r = fmaf(fmaf(-sum[h], r, 1.f), r, r);
#pragma unroll
      for (int i = 0; i < NPL; ++i)
        if (inN[i]) a_s[wave][h * N + lane + 64 * i] = e[i][h] * r;
    }
  }
  step_sync<WPG>();

  // ---- u_m = sum_{h,n} a[h][n] * RT[m][h][n] + cvec[m]  for selectable m ---------------
  {
    const float4 *aw = reinterpret_cast<const float4 *>(a_s[wave]) + part;
    float acc = 0.f;
    auto consume = [&](const float4 (&r)[RT_U], int w, int m) {
      const int ch = w % nchunk;
      acc = rt_dot(acc, r, aw, ch * RT_U, m >= 0 ? cnt : 0);
      if (ch == nchunk - 1) {
        acc = group8_sum(acc);
        if (part == 0 && m >= 0) u_s[wave][m] = acc;
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
  step_sync<WPG>();

  float u[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    u[i] = -INFINITY;
    if (inN[i] && !own_mask[i])
      u[i] = p.clip * tanhf(u_s[wave][lane + 64 * i] + cv[i]);  // graph_decoder.py:97-98
    if (active && p.io.logits && inN[i])
      p.io.logits[((size_t)t * B + b) * N + lane + 64 * i] = u[i];
    if (active && p.io.mask_trace && inN[i])
      p.io.mask_trace[((size_t)t * B + b) * N + lane + 64 * i] = (uint8_t)own_mask[i];
  }
  if (active && p.io.load_trace && lane == 0)
    p.io.load_trace[(size_t)t * B + b] = (float)load0;

  // lowest node index among the maxima (torch CPU argmax): slot 0 holds nodes < 64
  auto argmax_nodes = [&](const float (&v)[NPL]) {
    float mx = v[0];
#pragma unroll
    for (int i = 1; i < NPL; ++i) mx = fmaxf(mx, v[i]);
    const float m = wave_max(mx);
    int res = 0;
    bool found = false;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const unsigned long long hit = __ballot(v[i] == m);
      if (!found && hit) { res = 64 * i + __ffsll((long long)hit) - 1; found = true; }
    }
    return res;
  };

  int idx;
  float logp = 0.f;
  if (!p.sample) {
    idx = argmax_nodes(u);
    if (p.io.forced) idx = (int)p.io.forced[(size_t)t * B + b];
  } else {
    // Categorical(logits=u): logits - logsumexp, probs = softmax, sample = argmax(p/q)
    float mx = u[0];
#pragma unroll
    for (int i = 1; i < NPL; ++i) mx = fmaxf(mx, u[i]);
    const float m = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) se += expf(u[i] - m);
    se = wave_sum(se);
    const float lse = m + logf(se);
    float l[NPL], lmx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { l[i] = u[i] - lse; lmx = fmaxf(lmx, l[i]); }
    const float lm = wave_max(lmx);
    float pe[NPL], ps = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { pe[i] = expf(l[i] - lm); ps += pe[i]; }
    ps = wave_sum(ps);
    float ratio[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) ratio[i] = inN[i] ? (pe[i] / ps) / q_noise[i] : -1.f;
    idx = argmax_nodes(ratio);
    if (p.io.forced) idx = (int)p.io.forced[(size_t)t * B + b];
    const float lsel = (NPL > 1 && idx >= 64) ? l[NPL - 1] : l[0];
    logp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lsel),
                                                               idx & 63));
  }
  idx = __builtin_amdgcn_readfirstlane(idx);
  if (!active) return;  // wave-uniform; no barriers below

  // latency mode: next step's score row = SL[b][idx], copied while the env step runs (after
  // step 0 of TSP/VRP the table does not exist yet: its builder writes the row itself)
  if (WPG == 1 && !(t == 0 && p.kind != VRP_KIND_IRP)) {
    const float *arow = p.SL + ((size_t)b * N + idx) * 8 * N;
    float sl[NPL][8];
#pragma unroll
    for (int i = 0; i < NPL; ++i)
#pragma unroll
      for (int h = 0; h < 8; ++h)
        sl[i][h] = arow[h * N + ln[i]] + (p.base ? p.base[row + h * N + ln[i]] : 0.f);
#pragma unroll
    for (int i = 0; i < NPL; ++i)
#pragma unroll
      for (int h = 0; h < 8; ++h)
        if (inN[i]) p.curs[row + h * N + lane + 64 * i] = sl[i][h];
  }

  if (p.decode_only) {
    if (lane == 0) {
      p.last[b] = idx;
      if (t == 0) p.first[b] = idx;
      if (p.io.actions) p.io.actions[(size_t)t * B + b] = idx;
      if (p.io.step_logp) p.io.step_logp[(size_t)t * B + b] = logp;
    }
    return;
  }

  // ---- env.step on registers (same operation order as env_device.h) -------------------
  auto node_f64 = [&](const double (&v)[NPL], int n) {
    return (NPL > 1 && n >= 64) ? readlane_f64(v[NPL - 1], n - 64) : readlane_f64(v[0], n);
  };
  double px[NPL], py[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) { px[i] = xy[i].x; py[i] = xy[i].y; }
#pragma unroll
  for (int i = 0; i < NPL; ++i) if (lane + 64 * i == idx) vis[i] = 1;  // tsp.py:86
  const double dx = node_f64(px, cur) - node_f64(px, idx);
  const double dy = node_f64(py, cur) - node_f64(py, idx);
  const double dist = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
  double load = 1.0;
  if (p.kind == VRP_KIND_IRP) {                               // irp.py:80-86
    load = load0 - node_f64(dem, idx);
    if (idx == dep) load = 1.0;
  }
  auto all_visited = [&]() {
    int ok = 1;
#pragma unroll
    for (int i = 0; i < NPL; ++i) ok &= vis[i];
    return __all(ok);
  };
  const bool done = all_visited();                            // before the fix-ups, tsp.py:95
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    if (lane + 64 * i == dep) {
      if (idx == dep) vis[i] = 1;                             // tsp.py:141-142
      else if (p.kind != VRP_KIND_TSP) vis[i] = 0;            // vrp.py:28-31
    }
  }
  if (all_visited()) {                                        // tsp.py:145-146
#pragma unroll
    for (int i = 0; i < NPL; ++i) if (lane + 64 * i == dep) vis[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    int mk = vis[i];
    if (p.kind == VRP_KIND_IRP && inN[i] && dem[i] - load > 0.0) mk = 1;  // irp.py:151-153
    if (inN[i]) {
      p.env.visited[(size_t)b * N + lane + 64 * i] = (uint8_t)vis[i];
      mask_out[(size_t)b * N + lane + 64 * i] = (uint8_t)mk;
    }
  }
  if (lane == 0) {
    p.env.cur[b] = idx;
    if (p.kind == VRP_KIND_IRP) p.env.load[b] = load;
    p.io.acc_loss[b] = accl + (float)(-dist);  // fp32 accumulate in step order, tsp_agent:85
    p.io.acc_logp[b] = accp + logp;
    p.last[b] = idx;
    if (t == 0) p.first[b] = idx;
    if (!done) flag_notdone(&p.io.notdone[t]);
    if (p.io.actions) p.io.actions[(size_t)t * B + b] = idx;
    if (p.io.step_logp) p.io.step_logp[(size_t)t * B + b] = logp;
  }
}

template <int NPL, int WPG>  // nodes per lane: 1 (N <= 64) or 2 (N <= 128); node = lane + 64*i
__global__ __launch_bounds__(64 * WPG, (NPL == 1 ? RT_MINW : 2)) void decode_step_rt_kernel(StepParams p) {
  constexpr int NMAXL = 64 * NPL;
  __shared__ __attribute__((aligned(16))) float a_s[WPG][8 * NMAXL];  // a[h][n], hn order
  __shared__ __attribute__((aligned(16))) float u_s[WPG][NMAXL];
  __shared__ int sel_s[WPG][NMAXL];  // compacted list of selectable nodes
  step_rt_body<NPL, WPG, true>(p, p.t, blockIdx.x * WPG + (threadIdx.x >> 6), a_s, u_s, sel_s);
}


// Pointer-attention decoder (D1-D6 of SURVEY.md 8a) fused with the environment
// step (E4-E8): one launch = one iteration of the rollout loop
// agents/graph_tsp_agent.py:78-88, i.e. GraphDecoder.forward
// (agents/graph_decoder.py:51-115) followed by env.step (gym_vrp/envs/tsp.py:60-101).
//
// Algebra (all exact up to fp32 re-association):
//   * glimpse query q = Wq*ctx + bq is linear in [graph_emb, first_, last_(, load)],
//     the keys K_n = Wk*e_n + bk are per-episode constants, so the glimpse score
//     q_h.K_{h,n}/sqrt(48) splits into per-episode tables
//        SG[b][h][n]        graph-embedding + bias part
//        C0[b][h][n]        step-0 placeholders _first_node/_last_node;  row0 = SG + C0
//        SL[b][m][h][n]     "last node = m" part
//        base[b][h][n]      everything constant after step 0: SG, and for TSP/VRP the "first
//                           node" part (known after step 0)
//        SLD[b][h][n]       load coefficient           (IRP)
//     built once per episode (decoder_prologue.hip: vrp_decode_prologue /
//     vrp_decode_first_row); a step reads row0 at t = 0, SL[b][last] + base[b] afterwards
//     (pre-added into `curs` by the previous launch in the latency mode).
//   * sum_n a_n (Wv e_n + bv) = Wv (sum_n a_n e_n) + bv, and _kp/_att_output/out_proj
//     fold into one 128x384 matrix M, so a step reads only the RAW (N,128)
//     embedding tile, once, and keeps it in registers for both the glimpse
//     (z = A*E) and the pointer logits (u = E*w).
//   * QUIRK D3 reproduced: the float 0/1 mask is ADDED to the glimpse scores and
//     head h of graph b reads mask row (8b+h) mod B (graph_decoder.py:93-94).
#include <stdlib.h>
#include "decoder_step.h"
#include "x3_common.h"

// ------------------------------------------------------------------ derived weights
extern "C" int64_t vrp_decoder_derived_bytes(void) { return (int64_t)sizeof(float) * derived_floats(); }

// All folds are small strided matrix products (<= 19 M MAC each) and copies.  They run as
// TWO launches (the second consumes products of the first): a launch executes a table of
// tasks, one thread per output element,
//   C[i*scr + j*scc] = alpha * sum_k A[i*sar + k*sac] * Bm[k*sbr + j*sbc] + (add ? add[i*adr + j*adc] : 0)
// (K = 0: a strided copy, or a zero fill when add is NULL).  The fmaf chain keeps the k order.
struct FoldTask {
  float *C;
  const float *A, *Bm, *add;
  int scr, scc, sar, sac, sbr, sbc, adr, adc;
  int M, N, K, first_block;
  float alpha;
};
#define FOLD_MAX_TASKS 28
struct FoldTable {
  FoldTask t[FOLD_MAX_TASKS];
  int n, blocks;
};

__global__ __launch_bounds__(256) void fold_tasks_kernel(FoldTable tab) {
  int ti = 0;
#pragma unroll 1
  for (int i = 1; i < tab.n; ++i)
    if ((int)blockIdx.x >= tab.t[i].first_block) ti = i;
  const FoldTask &k = tab.t[ti];
  const int idx = (blockIdx.x - k.first_block) * 256 + threadIdx.x;
  if (idx >= k.M * k.N) return;
  const int i = idx / k.N, j = idx - i * k.N;
  float acc = 0.f;
#pragma unroll 16  // the loads of 16 k-steps in flight
  for (int kk = 0; kk < k.K; ++kk)
    acc = fmaf(k.A[(size_t)i * k.sar + (size_t)kk * k.sac],
               k.Bm[(size_t)kk * k.sbr + (size_t)j * k.sbc], acc);
  float v = k.alpha * acc;
  if (k.add) v += k.add[(size_t)i * k.adr + (size_t)j * k.adc];
  k.C[(size_t)i * k.scr + (size_t)j * k.scc] = v;
}

struct FoldBuilder {
  FoldTable tab;
  FoldBuilder() { tab.n = 0; tab.blocks = 0; }
  // C = alpha * A Bm (+ add)
  void mm(float *C, int scr, int scc, const float *A, int sar, int sac, const float *Bm, int sbr,
          int sbc, int M, int N, int K, float alpha, const float *add = nullptr, int adr = 0,
          int adc = 0) {
    if (tab.n >= FOLD_MAX_TASKS) { ++tab.n; return; }  // reported by launch()
    FoldTask &k = tab.t[tab.n++];
    k.C = C; k.A = A; k.Bm = Bm; k.add = add;
    k.scr = scr; k.scc = scc; k.sar = sar; k.sac = sac; k.sbr = sbr; k.sbc = sbc;
    k.adr = adr; k.adc = adc; k.M = M; k.N = N; k.K = K; k.alpha = alpha;
    k.first_block = tab.blocks;
    tab.blocks += (M * N + 255) / 256;
  }
  // dst[r*sdr + c*sdc] = src ? src[r*ssr + c*ssc] : 0
  void cp(float *dst, int sdr, int sdc, const float *src, int ssr, int ssc, int rows, int cols) {
    mm(dst, sdr, sdc, nullptr, 0, 0, nullptr, 0, 0, rows, cols, 0, 0.f, src, ssr, ssc);
  }
  int launch(hipStream_t st) {
    if (tab.n > FOLD_MAX_TASKS) { vrp_set_error("decoder_prepare: task table overflow"); return 2; }
    hipLaunchKernelGGL(fold_tasks_kernel, dim3(tab.blocks), dim3(256), 0, st, tab);
    VRP_CHECK_LAUNCH("fold_tasks");
    return 0;
  }
};

// v_proj and M once more, in the order the fold MFMAs consume them: a wave's fragment load is
// then 64 consecutive float4 (1 KB, eight whole cache lines) instead of sixteen 64-byte pieces
// of sixteen rows.  Runs after the fold tasks (it reads M).
__global__ __launch_bounds__(256) void pack_fold_weights_kernel(const float *__restrict__ Wv,
                                                                const float *__restrict__ M,
                                                                float *__restrict__ WvP,
                                                                float *__restrict__ MP) {
  const int i = blockIdx.x * 256 + threadIdx.x;   // float4 index, 2 x 12288
  if (i < 12288) {            // WvP[head][k4][tile][lane]
    const int lane = i & 63, f = i >> 6, c = f % 3, k4 = (f / 3) & 7, h = f / 24;
    const int i16 = lane & 15, q = lane >> 4;
    reinterpret_cast<float4 *>(WvP)[i] = *reinterpret_cast<const float4 *>(
        Wv + (size_t)(h * VRP_HD + 16 * c + i16) * VRP_EMB + 16 * k4 + 4 * q);
  } else if (i < 24576) {     // MP[tile][k4][lane]
    const int j = i - 12288, lane = j & 63, f = j >> 6, k4 = f % 24, ct = f / 24;
    const int i16 = lane & 15, q = lane >> 4;
    reinterpret_cast<float4 *>(MP)[j] = *reinterpret_cast<const float4 *>(
        M + (size_t)(16 * ct + i16) * VRP_D + 16 * k4 + 4 * q);
  }
}

// Wproj = [Wq_last | Wk | M^T | Wv] (1536,128) as three bf16 planes in the fragment order of the
// fused prologue's stage 1 (decoder_ws.h: WprojX3).  One thread per (fragment, chunk, lane).
__global__ __launch_bounds__(256) void pack_proj_x3_kernel(const float *__restrict__ Wproj,
                                                           __bf16 *__restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;   // 96 fragments x 4 chunks x 64 lanes
  if (idx >= 96 * 256) return;
  const int lane = idx & 63, j = (idx >> 6) & 3, f = idx >> 8;
  const int c = f % 3, X = (f / 3) & 3, h = f / 12;
  const int j16 = lane & 15, q = lane >> 4;
  const float *src = Wproj + (size_t)(X * 384 + h * VRP_HD + 16 * c + j16) * VRP_EMB +
                     64 * (q & 1) + 32 * (q >> 1) + 8 * j;
  bf16x8 hp, mp, lp;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const Bf3 s = x3_split(src[e]);
    hp[e] = s.h; mp[e] = s.m; lp[e] = s.l;
  }
  bf16x8 *dst = reinterpret_cast<bf16x8 *>(out + (size_t)f * X3_FRAG) + lane;
  dst[(0 * 4 + j) * 64] = hp;
  dst[(1 * 4 + j) * 64] = mp;
  dst[(2 * 4 + j) * 64] = lp;
}

extern "C" int vrp_decoder_prepare(int kind, const vrp_decoder_weights *w, void *derived,
                                   void *stream) {
  VRP_REQUIRE(w && derived, "decoder_prepare: NULL argument");
  VRP_REQUIRE(kind >= 0 && kind <= 2, "decoder_prepare: kind=%d", kind);
  VRP_REQUIRE(kind != VRP_KIND_IRP || w->context_proj_weight, "IRP needs _context_proj");
  hipStream_t st = (hipStream_t)stream;
  Derived d = carve_derived(derived);
  const float *Wq = w->q_proj_weight, *bias = w->in_proj_bias;
  const float s = 0.08838834764831845f;   // 1/sqrt(128)   graph_decoder.py:97
  const float c48 = 0.14433756729740643f; // 1/sqrt(48)    head dim of the glimpse attention
  FoldBuilder a, b;
  // ---- launch 1: everything that reads the raw parameters only ---------------------------
  if (kind != VRP_KIND_IRP) {
    // ctx = [graph_emb | first_ | last_]   graph_decoder.py:88
    a.cp(d.Wqf, 128, 1, Wq + 128, 384, 1, 384, 128);
    a.cp(d.Wproj, 128, 1, Wq + 256, 384, 1, 384, 128);
    a.cp(d.Wqg, 128, 1, Wq, 384, 1, 384, 128);
    // qc0 = Wq_first * _first_node + Wq_last * _last_node   graph_decoder.py:79-81
    a.mm(d.qc0, 1, 0, Wq + 128, 384, 1, w->first_node, 1, 0, 384, 1, 128, 1.f);
    b.mm(d.qc0, 1, 0, Wq + 256, 384, 1, w->last_node, 1, 0, 384, 1, 128, 1.f, d.qc0, 1, 0);
    a.cp(d.wload, 0, 1, nullptr, 0, 0, 1, 384);
    // first-node query folded through the keys:
    //   AfT[h*128+k][i] = sum_d Wk[48h+d][k] Wq_first[48h+d][i] / sqrt(48)
    // (the bias part (Wq_first e_first) . bk is constant over the nodes of a head's score row:
    // softmax-invariant, dropped)
    for (int h = 0; h < 8; ++h)
      a.mm(d.AfT + (size_t)h * 128 * 128, 128, 1, w->k_proj_weight + (size_t)h * 48 * 128, 1, 128,
           Wq + (size_t)h * 48 * 384 + 128, 384, 1, 128, 128, 48, c48);
  } else {
    // ctx = _context_proj([graph_emb | last_ | load])   graph_decoder.py:90-91
    const float *Wc = w->context_proj_weight;  // (384,257)
    a.cp(d.Wqf, 0, 1, nullptr, 0, 0, 1, 384 * 128);
    a.mm(d.Wqg, 128, 1, Wq, 384, 1, Wc, 257, 1, 384, 128, 384, 1.f);
    a.mm(d.Wproj, 128, 1, Wq, 384, 1, Wc + 128, 257, 1, 384, 128, 384, 1.f);
    a.mm(d.wload, 1, 0, Wq, 384, 1, Wc + 256, 257, 0, 384, 1, 384, 1.f);
    b.mm(d.qc0, 1, 0, d.Wproj, 128, 1, w->last_node, 1, 0, 384, 1, 128, 1.f);
    a.cp(d.AfT, 0, 1, nullptr, 0, 0, 1, 1024 * 128);
  }
  a.cp(d.Wproj + 384 * 128, 128, 1, w->k_proj_weight, 128, 1, 384, 128);
  a.cp(d.Wproj + (size_t)1152 * 128, 128, 1, w->v_proj_weight, 128, 1, 384, 128);
  // bproj = 0 | bk | 0 | bv;  bq;  bv
  a.cp(d.bproj, 0, 1, nullptr, 0, 0, 1, 384);
  a.cp(d.bproj + 384, 0, 1, bias + 384, 0, 1, 1, 384);
  a.cp(d.bproj + 768, 0, 1, nullptr, 0, 0, 1, 384);
  a.cp(d.bproj + 1152, 0, 1, bias + 768, 0, 1, 1, 384);
  a.cp(d.bq, 0, 1, bias, 0, 1, 1, 384);
  a.cp(d.bv, 0, 1, bias + 768, 0, 1, 1, 384);
  a.cp(d.WvT, 1, 384, w->v_proj_weight, 128, 1, 384, 128);  // WvT[k][j] = Wv[j][k]
  // tmpA = Watt (128,384) * Wo (384,384);  tmpv = Watt * bo
  a.mm(d.tmpA, 384, 1, w->att_output_weight, 384, 1, w->out_proj_weight, 384, 1, 128, 384, 384, 1.f);
  a.mm(d.tmpv, 1, 0, w->att_output_weight, 384, 1, w->out_proj_bias, 1, 0, 128, 1, 384, 1.f);
  // ---- launch 2: M = s * Wkp^T tmpA (stored transposed, twice: MT and the KM block of
  // Wproj);  mb = s * Wkp^T tmpv
  b.mm(d.MT, 1, 128, w->kp_weight, 1, 128, d.tmpA, 384, 1, 128, 384, 128, s);
  b.mm(d.Wproj + (size_t)768 * 128, 1, 128, w->kp_weight, 1, 128, d.tmpA, 384, 1, 128, 384, 128, s);
  b.mm(d.M, 384, 1, w->kp_weight, 1, 128, d.tmpA, 384, 1, 128, 384, 128, s);
  b.mm(d.mb, 1, 0, w->kp_weight, 1, 128, d.tmpv, 1, 0, 128, 1, 128, s);
  b.cp(d.WqgT, 1, 384, d.Wqg, 128, 1, 384, 128);   // WqgT[k][j] = Wqg[j][k] (Wqg: launch 1)
  b.cp(d.WqfT, 1, 384, d.Wqf, 128, 1, 384, 128);   // WqfT[k][j] = Wqf[j][k]
  if (int r = a.launch(st)) return r;
  if (int r = b.launch(st)) return r;
  hipLaunchKernelGGL(pack_fold_weights_kernel, dim3(96), dim3(256), 0, st,
                     d.Wproj + (size_t)1152 * 128, d.M, d.WvP, d.MP);
  VRP_CHECK_LAUNCH("pack_fold_weights");
  hipLaunchKernelGGL(pack_proj_x3_kernel, dim3(96), dim3(256), 0, st, d.Wproj,
                     reinterpret_cast<__bf16 *>(d.WprojX3));
  VRP_CHECK_LAUNCH("pack_proj_x3");
  return 0;
}

#include "decoder_rt_body.h"

static int launch_step_any(const StepParams &p, int flags, hipStream_t st);

StepParams vrp_make_step_params(int kind, const void *derived, const vrp_env *env, const float *emb,
                                void *workspace, const vrp_rollout_io *io, int t, int max_steps,
                                int flags) {
  const int B = env->B, N = env->N;
  Derived d = carve_derived(const_cast<void *>(derived));
  DecWs ws = carve_decws(workspace, B, N);
  StepParams p;
  p.kind = kind; p.B = B; p.N = N; p.t = t; p.max_steps = max_steps;
  p.sample = flags & VRP_STEP_SAMPLE;
  p.decode_only = (flags & VRP_STEP_DECODE_ONLY) ? 1 : 0;
  p.emb = emb;
  p.embP = (kind != VRP_KIND_IRP && tile_pairs_shape(B, N)) ? ws.embP : nullptr;
  p.row0 = ws.row0; p.SLD = ws.SLD; p.SL = ws.SL; p.curs = ws.curs;
  p.base = (kind == VRP_KIND_IRP) ? nullptr : ws.base;  // IRP: the constant row is inside SL
  p.last = ws.last; p.first = ws.first;
  p.WvT = d.WvT; p.bv = d.bv; p.MT = d.MT; p.mb = d.mb;
  p.Wv = d.Wproj + (size_t)1152 * 128; p.M = d.M;
  p.WvP = d.WvP; p.MP = d.MP;
  p.RT = ws.RT; p.cvec = ws.cvec;
  p.skip_curs = 0;
  p.clip = io->logit_clip > 0.f ? io->logit_clip : 10.f;
  // tuning aid, read at every call (tools/tile_phase_probe.py changes it between launches)
  const char *dbg = getenv("VRP_TILE_DBG");
  p.dbg = dbg ? atoi(dbg) : 0;
  static const int stagger = getenv("VRP_TILE_STAGGER") ? atoi(getenv("VRP_TILE_STAGGER")) : 0;
  p.stagger = stagger;
  p.env = *env;
  p.io = *io;
  return p;
}

extern "C" int vrp_decode_step(int kind, const void *derived, const vrp_decoder_weights *w,
                               const vrp_env *env, const float *emb, void *workspace,
                               const vrp_rollout_io *io, int t, int max_steps, int flags,
                               void *stream) {
  (void)w;
  const int sample = flags & VRP_STEP_SAMPLE, decode_only = (flags & VRP_STEP_DECODE_ONLY) ? 1 : 0;
  VRP_REQUIRE(derived && env && emb && workspace && io, "decode_step: NULL argument");
  VRP_REQUIRE(env->kind == kind, "decode_step: env.kind=%d but kind=%d", env->kind, kind);
  VRP_REQUIRE(decode_only || (io->acc_loss && io->acc_logp && io->notdone),
              "decode_step: io accumulators NULL");
  VRP_REQUIRE(env->mask && (kind != VRP_KIND_IRP || env->load), "decode_step: env.mask/load NULL");
  VRP_REQUIRE(!sample || io->noise || io->noise_seed,
              "decode_step: sampling needs io.noise or io.noise_seed");
  VRP_REQUIRE(t >= 0 && t < max_steps, "decode_step: t=%d outside [0,%d)", t, max_steps);
  const int B = env->B, N = env->N;
  VRP_REQUIRE(N >= 2 && N <= VRP_MAX_NODES, "decode_step: N=%d unsupported (2..%d)", N, VRP_MAX_NODES);
  VRP_REQUIRE(use_rtable(N) || N <= 104, "decode_step: tile kernel supports N <= 104");
  VRP_REQUIRE(!(flags & VRP_STEP_TILE_KERNEL) || N <= 104, "decode_step: tile kernel supports N <= 104");
  const StepParams p = vrp_make_step_params(kind, derived, env, emb, workspace, io, t, max_steps, flags);
  hipStream_t st = (hipStream_t)stream;
  if (int r = launch_step_any(p, flags, st)) return r;
  if (t == 0 && kind != VRP_KIND_IRP && !(flags & VRP_STEP_NO_FIRST_ROW))
    return vrp_decode_first_row(kind, derived, B, N, emb, workspace, stream);
  return 0;
}

// Which kernel takes which graphs at step t.  The raw-tile kernel costs the same whatever the
// mask, a table step costs one 32 N-byte row per selectable node: tile for graphs with at least
// `thresh` selectable nodes (measured crossover, large batches), table below.
static int tile_threshold(int kind, int N) {
  static const int v = getenv("VRP_TILE_MIN_SEL") ? atoi(getenv("VRP_TILE_MIN_SEL")) : 0;
  if (v > 0) return v;
  // measured crossovers (tools/step_probe.py): N = 100: a table step costs 1.21 us per
  // selectable node at B = 2048 against 38 us flat for the raw-tile kernel; with one kernel per
  // step (launch_step_any) the mean step of a VRP-100 x 2048 sampling episode is 33.9 / 33.8 /
  // 33.6 / 33.7 / 33.5 / 34.1 / 34.4 us for thresholds of 25 / 28 / 31 / 34 / 37 / 40 / 44
  // expected selectable nodes (per-graph routing at 40: 37.0); greedy, untrained (long tours):
  // 36.0 / 37.1 / 38.3 at 28 / 34 / 40.  N = 40 (large batches):
  // 102 / 83 / 77 / 77 / 77 / 74 / 71 us for the first seven steps of a TSP episode, 66 us at
  // the eighth, against 65.7 us flat for the raw-tile kernel (round 3: fold weights in fragment
  // order): threshold 33 of 39 selectable nodes measured best (42.8 us per step, against 43.9 for
  // the table kernel alone and 43.1 / 43.2 for thresholds 30 / 36)
  // VRP at N <= 40 (the tile kernel runs only while EVERY graph is above the threshold, see
  // launch_step_any): 28 of 39 measured best (38.5 us per step; 30 / 33 / 35: 38.7 / 39.3 / 39.5;
  // table kernel alone 40.0)
  if (N > 64) return (31 * N + 50) / 100;
  return kind == VRP_KIND_VRP ? (28 * N + 20) / 40 : (33 * N + 20) / 40;
}
// N > 64: ON by default -- a table row is 32 N bytes per selectable node, the raw tile 512 N
// bytes whatever the mask, and since its weight folds stream their fragments line by line
// (decoder_tile.hip) the tile kernel wins the first two thirds of an episode: VRP-100 x 2048
// sampling 63 -> 33.6 us per step.
// IRP stays with the table kernel: its capacity overlay leaves few nodes selectable from the
// first step on (64 vs 57 us).
static bool hybrid_shape(int kind, int B, int N) {
  static const bool off = getenv("VRP_TILE_NO_HYBRID") != nullptr;   // A/B aid
  if (N > 64) return !off && kind != VRP_KIND_IRP && vrp_tile_mfma_supported(N);
  // 32 < N <= 40, large batches, TSP and VRP: the first steps of an episode -- while EVERY graph
  // still has at least 33 of 39 nodes selectable -- go to the raw-tile kernel, the rest to the table
  // kernel: exactly one kernel runs per step (a VRP batch split per graph ran both kernels for a
  // dozen steps: 41.0 - 46.2 us against 39.4).  IRP's capacity overlay closes nodes from the first
  // step on: table kernel.
  return !off && B > 2048 && N > 32 && N <= 40 && kind != VRP_KIND_IRP;
}

// name of the kernel vrp_decode_step dispatches for this shape (profiles, bench line)
extern "C" const char *vrp_step_kernel_name(int kind, int B, int N, int flags) {
  vrp_rollout_io none = {};
  if (vrp_persistent_eligible(kind, B, N, 2, flags, &none, nullptr)) return "decode_persistent_kernel";
  const bool v2 = !tile_v1_forced() && vrp_tile2_supported(N);
  const char *tile = v2 ? (kind == VRP_KIND_IRP
                               ? (N <= 40 ? "decode_step_tile_zmfma_kernel<40, 2, true>" : "decode_step_tile_zmfma_kernel<100, 1, true>")
                               : (N <= 40 ? "decode_step_tile_zmfma_kernel<40, 2, false>" : "decode_step_tile_zmfma_kernel<100, 1, false>"))
                        : N <= 40    ? "decode_step_tile_mfma_kernel<40, 2, 8>"
                          : N <= 100 ? "decode_step_tile_mfma_kernel<100, 1, 8>"
                                     : "decode_step_tile_mfma_kernel<104, 1, 8>";
  if (flags & VRP_STEP_TILE_KERNEL) return tile;
  if (N > 64 && vrp_tile_mfma_supported(N) && !(flags & VRP_STEP_THROUGHPUT_KERNEL) &&
      getenv("VRP_TILE_LARGE_N")) return tile;
  if (hybrid_shape(kind, B, N) && !(flags & VRP_STEP_TABLE_KERNEL)) {
    static thread_local char name[160];
    const char *rt = N <= 40 ? "decode_step_rt_kernel<1, 4>"
                             : (B <= 2048 && !(flags & VRP_STEP_THROUGHPUT_KERNEL)
                                    ? "decode_step_rt_kernel<2, 1>" : "decode_step_rt_kernel<2, 4>");
    snprintf(name, sizeof(name), "%s (first steps) | %s", tile, rt);
    return name;
  }
  const bool small = B <= 2048 && !(flags & VRP_STEP_THROUGHPUT_KERNEL);
  if (N <= 64) return small ? "decode_step_rt_kernel<1, 1>" : "decode_step_rt_kernel<1, 4>";
  return small ? "decode_step_rt_kernel<2, 1>" : "decode_step_rt_kernel<2, 4>";
}

static int launch_rt(const StepParams &p, int flags, hipStream_t st) {
  const int B = p.B, N = p.N;
  const bool small = B <= 2048 && !(flags & VRP_STEP_THROUGHPUT_KERNEL);
  if (N <= 64 && small)
    hipLaunchKernelGGL((decode_step_rt_kernel<1, 1>), dim3(B), dim3(64), 0, st, p);
  else if (N <= 64)
    hipLaunchKernelGGL((decode_step_rt_kernel<1, 4>), dim3((B + 3) / 4), dim3(256), 0, st, p);
  else if (small)
    hipLaunchKernelGGL((decode_step_rt_kernel<2, 1>), dim3(B), dim3(64), 0, st, p);
  else
    hipLaunchKernelGGL((decode_step_rt_kernel<2, 4>), dim3((B + 3) / 4), dim3(256), 0, st, p);
  VRP_CHECK_LAUNCH("decode_step_rt");
  return 0;
}

static int launch_step_any(const StepParams &p, int flags, hipStream_t st) {
  const int B = p.B, N = p.N;
  if (flags & VRP_STEP_TILE_KERNEL) return vrp_launch_tile_mfma_step(p, st);  // every graph
  if (N > 64 && vrp_tile_mfma_supported(N) && !(flags & VRP_STEP_THROUGHPUT_KERNEL)) {
    static const bool on = getenv("VRP_TILE_LARGE_N") != nullptr;  // A/B aid (67 vs 62 us at 2048 x 100)
    if (on) return vrp_launch_tile_mfma_step(p, st);
  }
  if (hybrid_shape(p.kind, B, N) && !p.decode_only && !(flags & VRP_STEP_TABLE_KERNEL)) {
    const int th = tile_threshold(p.kind, N);
    // most selectable nodes any graph can have at step t (TSP: exactly N-1-t; VRP/IRP: a customer
    // is served at least every other step, the depot may be open), and fewest (VRP: the mask is
    // the visited row and a step visits at most one customer; IRP: the capacity overlay can
    // close any number of them)
    auto most_at = [&](int t) { return (p.kind == VRP_KIND_TSP) ? N - 1 - t : N - (t + 1) / 2; };
    auto least_at = [&](int t) {
      return (p.kind == VRP_KIND_TSP) ? N - 1 - t : (p.kind == VRP_KIND_VRP ? max(0, N - 2 - t) : 0);
    };
    // ONE kernel per step, chosen by the step number alone (the raw-tile kernel is correct for
    // any mask; the choice is a cost estimate).  N <= 64: the tile kernel while EVERY graph is
    // above the threshold.  N > 64: while the expected count -- a quarter of the way from the
    // fewest to the most (depot returns are the minority of a tour's steps) -- is.  (Per-graph
    // routing, until round 3, ran BOTH kernels at full cost while a batch straddled the threshold
    // -- VRP-100 x 2048: six steps of 34 + 44 us -- and an empty tile launch, 4.7 us, for fifty
    // steps after: 37.0 against 33.6 us per step.)
    auto tile_step = [&](int t) {
      const int lo = least_at(t), hi = most_at(t);
      return (N <= 64 ? lo : lo + (hi - lo) / 4) >= th;
    };
    if (!tile_step(p.t)) return launch_rt(p, flags, st);
    StepParams pt = p;
    pt.skip_curs = (p.t + 1 < p.max_steps && tile_step(p.t + 1)) ? 1 : 0;
    return vrp_launch_tile_mfma_step(pt, st);
  }
  return launch_rt(p, flags, st);
}

// Decode + env step from the RAW embedding tile, second generation (round 4): the same step as
// decoder_tile.hip -- a graph-step reads its (N,128) fp32 tile ONCE, exactly the algorithmic
// 512 N bytes of SURVEY.md 8d -- with the tile held in the operand layout of the matrix cores,
// so that the glimpse sums run on them WHILE the rest of the tile still streams in, and the
// pointer logits need a 16-lane (one DPP row) reduction instead of a 64-lane one:
//
//   lane (r, c) = (lane >> 4, lane & 15) holds, for k-step j and half hf, the float4
//       T[j][hf] = e[node 4 j + r][columns 64 hf + 4 c .. 4 c + 3]
//   (one load instruction = four rows x 256 contiguous bytes, straight from `emb`: no paired
//   copy of the embeddings, no pair_rows_kernel).
//
//   a[h][n]   glimpse weights from the score rows + scrambled masks (graph_decoder.py:93-94),
//             lane = node, through LDS                                        VALU, as before
//   z_h       = sum_n a[h][n] e_n  =  v_mfma_f32_16x16x4_f32 with A[m = head][k = r] =
//             a[h][4 j + r] (LDS, one dword per k-step) and B[k = r][n = c] = T[j][hf].{x,y,z,w}:
//             eight accumulators (hf, element), 2 NMAX MFMAs per graph, issued in the order the
//             rows land -- the matrix pipe works under the tile loads instead of 16 N packed
//             FMAs per lane running behind them (round 3: 19.5 us of loads, THEN 12.9 us of
//             vector issue, THEN 4.2 us of folds per step at 2048 x 100)
//   o = Wv z + bv, w = M o + mb   the two weight folds, unchanged (decoder_tile.hip)
//   u_n       = 10 tanh(e_n . w + cvec_n): eight FMAs per node in the lane (its 8 columns), then
//             a reduce-scatter over the 16 lanes of the row (DPP only: row_ror:8, row_half_mirror,
//             two quad_perms; 45 instructions per 16 nodes) and one trip through LDS to the
//             lane = node layout the action / env code wants (round 3: 64-lane butterfly, 252
//             instructions per 64 nodes + 2 per node of products)
//
// Everything after the logits (argmax / Categorical sample, env.step on registers, traces) is
// the code of decoder_tile.hip.
#include "decoder_step.h"

#define T2_ZG 1028  // zs: floats between graphs (8 heads x 128 + 4: conflict-free b128 reads)
#define T2_OS 388   // os: floats per graph row (384 + 4)
#define T2_WS 132   // ws: floats per graph row
#define T2_US 128   // us: pointer-logit pre-activations of a graph, lane = node order
#ifndef T2_PF
#define T2_PF 2     // first weight fold, N > 64: k-steps of fragments requested ahead
#endif
#ifndef T2_PF2
#define T2_PF2 4    // second weight fold, N > 64
#endif
#ifndef T2_KQ
#define T2_KQ 8     // k-steps of the tile requested before the glimpse weights are computed
#endif
#ifndef T2_TILE_AUX
#define T2_TILE_AUX 2  // cache policy of the tile loads: 2 = nt (once-read stream), 0 = default
#endif

#define T2_DPP_ADD(x, ctrl) ((x) + VRP_DPP(0.f, (x), (ctrl), 0xF))

// v[i] (i = 0..15) summed over the 16 lanes of each DPP row; lane c of a row returns the total of
// v[c].  Recursive halving; partners c^8 (row_ror:8), c^7 (row_half_mirror), c^2, c^1 (quad_perm):
// the lanes a value has been summed over double at every level and stay disjoint, the kept half
// follows bit 3, 2, 1, 0 of c.
__device__ __forceinline__ float row_reduce_scatter16(const float (&v)[16], int c) {
  float a[8], b[4], d[2];
  const bool u8 = (c & 8) != 0, u4 = (c & 4) != 0, u2 = (c & 2) != 0, u1 = (c & 1) != 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float keep = u8 ? v[i + 8] : v[i], send = u8 ? v[i] : v[i + 8];
    a[i] = keep + VRP_DPP(0.f, send, 0x128, 0xF);   // row_ror:8
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float keep = u4 ? a[i + 4] : a[i], send = u4 ? a[i] : a[i + 4];
    b[i] = keep + VRP_DPP(0.f, send, 0x141, 0xF);   // row_half_mirror
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float keep = u2 ? b[i + 2] : b[i], send = u2 ? b[i] : b[i + 2];
    d[i] = keep + VRP_DPP(0.f, send, 0x4E, 0xF);    // quad_perm [2,3,0,1]
  }
  const float keep = u1 ? d[1] : d[0], send = u1 ? d[0] : d[1];
  return keep + VRP_DPP(0.f, send, 0xB1, 0xF);      // quad_perm [1,0,3,2]
}

typedef float t2_v4f __attribute__((vector_size(16)));

// IRP: compile-time env kind split (IRP reads the load-dependent score row and the demands; as a
// run-time test every one of those loads sat behind a branch of its own).  NT: nontemporal tile
// loads (tools/micro/stream_rate.hip: a once-read stream of this geometry moves 6.4-6.5 TB/s with
// `nt`, 6.0 without).
template <int NMAX, int GPW, bool IRP>
__global__ __launch_bounds__(512, 2) void decode_step_tile_zmfma_kernel(StepParams p) {
  constexpr int NW = 8;
  constexpr int KS = NMAX / 4;             // k-steps of four nodes
  constexpr int NPL = (NMAX + 63) / 64;
  constexpr int GPB = NW * GPW;            // graphs per workgroup (<= 16 = rows of one MFMA tile)
  constexpr int ROWS = GPB;                // rows of the LDS operand images (8 or 16: MFMA rows
                                           // beyond the workgroup's graphs repeat them)
  // Two nodes per lane (N > 64): 25 k-steps are 200 registers, and with the 32 accumulators of the
  // glimpse sums the allocator spills tile registers -- each reload a scratch load, the youngest
  // entry of the in-order vmcnt queue, whose wait drains every tile load in flight.  The last
  // KL k-steps therefore live in LDS (2 KB per graph and k-step, in lane order: the image IS the
  // register layout); they are requested FIRST, land with the score rows and are written to LDS
  // before the bulk of the tile is requested.
  constexpr int KL = NMAX > 64 ? 2 : 0;    // k-steps held in LDS
  constexpr int KR = KS - KL;              // k-steps held in registers
  static_assert(NMAX % 4 == 0 && GPB <= 16 && (ROWS & (ROWS - 1)) == 0, "tile shape");
  if (!p.decode_only && p.t > 0 && p.io.notdone[p.t - 1] == 0) return;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *a_s = smem;                      // [GPB][NMAX*8]  a[g][n][h]
  float *zs = a_s + GPB * NMAX * 8;       // [ROWS][T2_ZG]  z[g][h][128]
  float *os = zs + ROWS * T2_ZG;          // [ROWS][T2_OS]  o[g][384]
  float *ws = os + ROWS * T2_OS;          // [ROWS][T2_WS]  w[g][128]
  float *us = ws + ROWS * T2_WS;          // [GPB][T2_US]
  float4 *tl = reinterpret_cast<float4 *>(us + GPB * T2_US);   // [GPB][KL][2][64] tile k-steps in LDS

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, B = p.B;
  const int par = p.t & 1;
  const uint8_t *mask_in = p.env.mask + (size_t)par * B * N;
  uint8_t *mask_out = p.env.mask + (size_t)(par ^ 1) * B * N;
  const int i16 = lane & 15, q = lane >> 4;   // (c, r) of the header comment

  // ---- per-graph state kept across the matrix phase ---------------------------------------
  float4 T[GPW][KR > 0 ? KR : 1][2];
  int own_mask[GPW][NPL], vis[GPW][NPL];
  double2 xy[GPW][NPL];
  double dem[GPW][NPL];
  float cv[GPW][NPL], q_noise[GPW][NPL];
  int cur[GPW], dep[GPW], bg[GPW];
  double load0[GPW];
  float accl[GPW], accp[GPW];
  bool proc[GPW];
  // env row, noise, accumulators of a graph (lane = node).  Two nodes per lane (N > 64): the
  // tile alone takes 200 registers, so these are fetched after the matrix phase (in flight
  // during the logit sums) instead of being carried through it.
  constexpr bool DEFER = NPL > 1;
  auto load_env = [&](int gi, int b) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const bool in = lane + 64 * i < N;
      const int l = in ? lane + 64 * i : 0;
      cv[gi][i] = p.cvec[(size_t)b * N + l];
      xy[gi][i] = make_double2(0.0, 0.0);
      vis[gi][i] = 1;
      dem[gi][i] = 0.0;
      if (!p.decode_only) {
        xy[gi][i] = reinterpret_cast<const double2 *>(p.env.pos)[(size_t)b * N + l];
        if (in) vis[gi][i] = p.env.visited[(size_t)b * N + l];
        if (IRP) dem[gi][i] = p.env.demand[(size_t)b * N + l];
      }
      q_noise[gi][i] = !p.sample ? 1.f
                       : p.io.noise ? p.io.noise[((size_t)p.t * B + b) * N + l]
                                    : vrp_exp1_noise(p.io.noise_seed, p.t, b, l);
#ifdef VRP_MUTATION_NOISE_SHIFT  // test-the-tests build: off-by-one noise index
      if (p.sample && p.io.noise) q_noise[gi][i] = p.io.noise[((size_t)p.t * B + b) * N + (l + 1) % N];
#endif
      if (DEFER) own_mask[gi][i] = mask_in[(size_t)b * N + l];
    }
    cur[gi] = p.decode_only ? 0 : p.env.cur[b];
    dep[gi] = p.decode_only ? 0 : p.env.depot[b];
    accl[gi] = accp[gi] = 0.f;
    if (!p.decode_only) { accl[gi] = p.io.acc_loss[b]; accp[gi] = p.io.acc_logp[b]; }
  };

  int lastn[GPW];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int braw = blockIdx.x * GPB + wave * GPW + gi;
    const bool active = braw < B;
    const int b = __builtin_amdgcn_readfirstlane(active ? braw : B - 1);
    bg[gi] = b;
    lastn[gi] = p.t > 0 ? p.last[b] : 0;   // requested with the mask rows, not behind them
    if (!DEFER) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const bool in = lane + 64 * i < N;
        own_mask[gi][i] = mask_in[(size_t)b * N + (in ? lane + 64 * i : 0)];
      }
    }
    proc[gi] = active;
  }
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave * GPW + gi;
    const int b = bg[gi];
    bool inN[NPL];
    int ln[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) { inN[i] = lane + 64 * i < N; ln[i] = inN[i] ? lane + 64 * i : 0; }
    if (!proc[gi]) continue;  // wave-uniform
    // ---- loads.  Order matters: they return in order, and so does every wait.  (1) the score
    // rows and masks the glimpse weights need; (2) the first k-steps of the tile, enough to keep
    // the memory pipe busy while the weights are computed; (3) -- behind the weights -- the
    // rest, which the MFMAs below consume k-step by k-step as it lands.  sched_barriers pin the
    // three groups: left alone the compiler sinks score loads behind tile loads, and the wait for
    // them drains the tile (round 4 first cut: the whole first half landed before the softmax
    // started, 25 us of loads for 17 us worth of bytes).
    constexpr int KQ = KR < T2_KQ ? KR : T2_KQ;
    const size_t row = (size_t)b * 8 * N;
    const float *srow = p.row0 + row;
    if (p.t > 0) {
      const int last = __builtin_amdgcn_readfirstlane(lastn[gi]);
      srow = p.SL + ((size_t)b * N + last) * 8 * N;
    }
    const bool add_base = p.base && p.t > 0;
    load0[gi] = IRP ? p.env.load[b] : 1.0;
    // the tile, in MFMA operand order: k-step j, half hf = rows 4j..4j+3, 256 bytes of each.
    // Buffer loads: the descriptor covers exactly this graph's N rows, a row beyond N reads as
    // zeros (no predicate, no zero-fill, one scalar offset per k-step)
    const __amdgpu_buffer_rsrc_t tile_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.emb + (size_t)b * N * VRP_EMB), 0, N * VRP_EMB * 4, 0x00020000);
    const int tile_voff = q * (VRP_EMB * 4) + i16 * 16;
    auto tile_load = [&](int j, int hf) {
      // (whole-vector bit cast: __builtin_bit_cast of ONE element of a vector value reads
      // element 0 whatever the element named -- hipcc 7.2)
      const t2_v4f v = __builtin_bit_cast(t2_v4f, __builtin_amdgcn_raw_buffer_load_b128(
          tile_rsrc, tile_voff + 256 * hf, j * (4 * VRP_EMB * 4), T2_TILE_AUX));
      return make_float4(v[0], v[1], v[2], v[3]);
    };
    float4 tlv[KL > 0 ? KL : 1][2];
#pragma unroll
    for (int j = 0; j < KL; ++j)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) tlv[j][hf] = tile_load(KR + j, hf);
    float sv[NPL][8], bv_[NPL][8], sl_[NPL][8];
    int mo[NPL][8];
#pragma unroll
    for (int i = 0; i < NPL; ++i)
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        sv[i][h] = srow[h * N + ln[i]];
        bv_[i][h] = add_base ? p.base[row + h * N + ln[i]] : 0.f;
        sl_[i][h] = IRP ? p.SLD[row + h * N + ln[i]] : 0.f;
        mo[i][h] = mask_in[(size_t)((b * 8 + h) % B) * N + ln[i]];  // QUIRK D3: other graphs
      }
    __builtin_amdgcn_sched_barrier(0);
    auto load_kstep = [&](int j) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) T[gi][j][hf] = tile_load(j, hf);
    };
#pragma unroll
    for (int j = 0; j < KQ; ++j) load_kstep(j);
    __builtin_amdgcn_sched_barrier(0);
    float sc[NPL][8];  // score + additive scrambled mask
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        float v = sv[i][h];
        if (add_base) v += bv_[i][h];
        if (IRP) v = fmaf((float)load0[gi], sl_[i][h], v);
        sc[i][h] = v + (float)mo[i][h];
      }
    }
    // (the LDS k-steps were requested before the score rows: they have landed with them)
    float4 *tlg = tl + (size_t)g * KL * 128 + lane;
#pragma unroll
    for (int j = 0; j < KL; ++j)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) tlg[(2 * j + hf) * 64] = tlv[j][hf];
    if (!DEFER) load_env(gi, b);
    if (p.dbg == 1) { if (T[gi][0][0].x + sc[0][0] == 123.f) p.curs[0] = 1.f; continue; }

    // ---- glimpse attention weights (lane = n), one wave-wide shift for all eight heads -----
    {
      float s[NPL][8], mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < NPL; ++i)
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const float v = inN[i] ? sc[i][h] : -INFINITY;
          s[i][h] = v;
          mx = fmaxf(mx, v);
        }
      const float M = wave_max(mx);
      float *ag = a_s + (size_t)g * NMAX * 8;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        float ev[NPL], es = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) { ev[i] = inN[i] ? exp_nonpos(s[i][h] - M) : 0.f; es += ev[i]; }
        float sum = wave_sum(es);
        if (!(sum > 1e-30f)) {  // wave-uniform, practically never: per-head maximum
          float hm = -INFINITY;
#pragma unroll
          for (int i = 0; i < NPL; ++i) hm = fmaxf(hm, s[i][h]);
          hm = wave_max(hm);
          es = 0.f;
#pragma unroll
          for (int i = 0; i < NPL; ++i) { ev[i] = inN[i] ? exp_nonpos(s[i][h] - hm) : 0.f; es += ev[i]; }
          sum = wave_sum(es);
        }
        float r = __builtin_amdgcn_rcpf(sum);
        r = fmaf(fmaf(-sum, r, 1.f), r, r);
#pragma unroll
        for (int i = 0; i < NPL; ++i)
          if (lane + 64 * i < NMAX) ag[(lane + 64 * i) * 8 + h] = ev[i] * r;  // 0 beyond N
      }
    }
#pragma unroll
    for (int j = KQ; j < KR; ++j) load_kstep(j);
    __builtin_amdgcn_sched_barrier(0);
    // a_s of this graph is written and read by this wave only: LDS ops of one wave are ordered
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    // ---- z_h = sum_n a[h][n] e_n on the matrix cores, k-steps in the order they land --------
    {
      f32x4 zacc[2][4];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int el = 0; el < 4; ++el) zacc[hf][el] = f32x4{0.f, 0.f, 0.f, 0.f};
      // A[m = lane & 15][k = lane >> 4] = a[head m & 7][node 4 j + k]  (rows 8..15 repeat the heads)
      const float *ap = a_s + (size_t)g * NMAX * 8 + q * 8 + (i16 & 7);
      float an = ap[0];
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const float ac = an;
        if (j + 1 < KS) an = ap[(j + 1) * 32];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const float4 t = j < KR ? T[gi][j < KR ? j : 0][hf] : tlg[(2 * (j - KR) + hf) * 64];
          zacc[hf][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, t.x, zacc[hf][0], 0, 0, 0);
          zacc[hf][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, t.y, zacc[hf][1], 0, 0, 0);
          zacc[hf][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, t.z, zacc[hf][2], 0, 0, 0);
          zacc[hf][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac, t.w, zacc[hf][3], 0, 0, 0);
        }
      }
      // D[m = 4 q + i][n = c] = z[head 4 q + i][column 64 hf + 4 c + element]: lanes q < 2 hold
      // the eight heads; one 16-byte store per (hf, i)
      if (q < 2) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4 *>(zs + g * T2_ZG + (4 * q + i) * 128 + 64 * hf + 4 * i16) =
                make_float4(zacc[hf][0][i], zacc[hf][1][i], zacc[hf][2][i], zacc[hf][3][i]);
      }
    }
  }
  if (p.dbg == 1 || p.dbg == 2) return;
  __syncthreads();

  // ---- matrix phase: o = Wv z + bv (wave = head), then w = M o + mb (wave = 16 columns) ----
  // (decoder_tile.hip: fragments of the folded weights in MFMA order from L2, requested PF
  // k-steps ahead; inner dimensions spread over the lane groups as k = 16 S + 4 q + e)
  // (N > 64: the tile alone holds 200 of the 256 registers; one fragment set less in flight than
  // decoder_tile.hip keeps the kernel free of spills -- a spilled tile register is reloaded by a
  // scratch load, the youngest entry of the in-order vmcnt queue, and the wait for it drains
  // every tile load in flight)
  constexpr int PF = NMAX > 64 ? T2_PF : 3;
  constexpr int PF2 = NMAX > 64 ? T2_PF2 : 8;
  const int koff = 4 * q;
  float4 mw[PF2];
  {
    const int h = wave;
    const float4 *wbase = reinterpret_cast<const float4 *>(p.WvP) + (size_t)h * 24 * 64 + lane;
    float4 wq[PF][3];
#pragma unroll
    for (int j = 0; j < PF; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) wq[j][c] = wbase[(3 * j + c) * 64];
    f32x4 acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float bb = p.bv[h * VRP_HD + 16 * c + i16];  // D column = lane & 15
      acc[c] = f32x4{bb, bb, bb, bb};
    }
    const float *arow = zs + (i16 & (ROWS - 1)) * T2_ZG + h * 128 + koff;   // A row = graph
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      const float4 a = *reinterpret_cast<const float4 *>(arow + 16 * k4);
      const float av[4] = {a.x, a.y, a.z, a.w};
      float4 w[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        w[c] = wq[k4 % PF][c];
        if (k4 + PF < 8) wq[k4 % PF][c] = wbase[(3 * (k4 + PF) + c) * 64];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], w[c].x, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], w[c].y, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], w[c].z, acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], w[c].w, acc[c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)  // D: row = graph 4q + r4, column = lane & 15
        if (4 * q + r4 < ROWS) os[(4 * q + r4) * T2_OS + h * VRP_HD + 16 * c + i16] = acc[c][r4];
  }
  const float4 *mrow = reinterpret_cast<const float4 *>(p.MP) + (size_t)wave * 24 * 64 + lane;
#pragma unroll
  for (int j = 0; j < PF2; ++j) mw[j] = mrow[j * 64];  // travel across the barrier
  __syncthreads();
  {
    const int ct = wave;
    const float mbv = p.mb[ct * 16 + i16];
    f32x4 acc0 = {mbv, mbv, mbv, mbv}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: k4 even / odd
    const float *arow = os + (i16 & (ROWS - 1)) * T2_OS + koff;
#pragma unroll
    for (int k4 = 0; k4 < 24; ++k4) {
      const float4 a = *reinterpret_cast<const float4 *>(arow + 16 * k4);
      const float4 w = mw[k4 % PF2];
      if (k4 + PF2 < 24) mw[k4 % PF2] = mrow[(k4 + PF2) * 64];
      f32x4 &acc = (k4 & 1) ? acc1 : acc0;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
      if (4 * q + r4 < ROWS) ws[(4 * q + r4) * T2_WS + ct * 16 + i16] = acc0[r4] + acc1[r4];
  }
  __syncthreads();

  if (p.dbg == 3) return;
  // ---- per graph: pointer logits, action, env step -----------------------------------------
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    if (!proc[gi]) continue;
    const int g = wave * GPW + gi;
    const int b = bg[gi];
    bool inN[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) inN[i] = lane + 64 * i < N;
    const float4 *tlg = tl + (size_t)g * KL * 128 + lane;
    const float4 w0 = *reinterpret_cast<const float4 *>(ws + g * T2_WS + 4 * i16);
    const float4 w1 = *reinterpret_cast<const float4 *>(ws + g * T2_WS + 64 + 4 * i16);
    // e_n . w over the lane's eight columns, then over the 16 lanes of the row: lane c of row r
    // ends up with node 4 (c + 16 k) + r of batch k; through LDS to lane = node
    float *ug = us + g * T2_US;
#pragma unroll
    for (int k = 0; k < (KS + 15) / 16; ++k) {
      float pv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int j = 16 * k + i;
        if (j < KS) {
          const float4 t0 = j < KR ? T[gi][j < KR ? j : 0][0] : tlg[(2 * (j - KR)) * 64];
          const float4 t1 = j < KR ? T[gi][j < KR ? j : 0][1] : tlg[(2 * (j - KR) + 1) * 64];
          float s = t0.x * w0.x;
          s = fmaf(t0.y, w0.y, s); s = fmaf(t0.z, w0.z, s); s = fmaf(t0.w, w0.w, s);
          s = fmaf(t1.x, w1.x, s); s = fmaf(t1.y, w1.y, s); s = fmaf(t1.z, w1.z, s);
          pv[i] = fmaf(t1.w, w1.w, s);
        } else {
          pv[i] = 0.f;
        }
      }
      const float tot = row_reduce_scatter16(pv, i16);
      if (16 * k + i16 < KS) ug[4 * (16 * k + i16) + q] = tot;
      // the env row / noise / accumulators of a two-nodes-per-lane graph: requested once the
      // first sixteen k-steps of the tile are dead (their registers are what these land in),
      // in flight during the second batch of logit sums
      if (DEFER && k == 0) load_env(gi, b);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): us of this graph is this wave's own
    __builtin_amdgcn_wave_barrier();
    float u[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const float x = (lane + 64 * i < NMAX) ? ug[lane + 64 * i] : 0.f;
      if (p.dbg == 4) { if (x == 123.f) p.curs[0] = x; continue; }
      u[i] = (inN[i] && !own_mask[gi][i]) ? p.clip * tanhf(x + cv[gi][i]) : -INFINITY;  // graph_decoder.py:97-98
      if (p.io.logits && inN[i]) p.io.logits[((size_t)p.t * B + b) * N + lane + 64 * i] = u[i];
      if (p.io.mask_trace && inN[i])
        p.io.mask_trace[((size_t)p.t * B + b) * N + lane + 64 * i] = (uint8_t)own_mask[gi][i];
    }
    if (p.dbg == 4) continue;
    if (p.io.load_trace && lane == 0) p.io.load_trace[(size_t)p.t * B + b] = (float)load0[gi];

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
      if (p.io.forced) idx = (int)p.io.forced[(size_t)p.t * B + b];
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
      for (int i = 0; i < NPL; ++i) ratio[i] = inN[i] ? (pe[i] / ps) / q_noise[gi][i] : -1.f;
      idx = argmax_nodes(ratio);
      if (p.io.forced) idx = (int)p.io.forced[(size_t)p.t * B + b];
      const float lsel = (NPL > 1 && idx >= 64) ? l[NPL - 1] : l[0];
      logp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lsel),
                                                                 idx & 63));
    }
    idx = __builtin_amdgcn_readfirstlane(idx);

    if (p.decode_only) {
      if (lane == 0) {
        p.last[b] = idx;
        if (p.t == 0) p.first[b] = idx;
        if (p.io.actions) p.io.actions[(size_t)p.t * B + b] = idx;
        if (p.io.step_logp) p.io.step_logp[(size_t)p.t * B + b] = logp;
      }
      continue;
    }
    // ---- env.step on registers (same operation order as env_device.h) -------------------
    auto node_f64 = [&](const double (&v)[NPL], int n) {
      return (NPL > 1 && n >= 64) ? readlane_f64(v[NPL - 1], n - 64) : readlane_f64(v[0], n);
    };
    double px[NPL], py[NPL], dm[NPL];
    int vs[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) { px[i] = xy[gi][i].x; py[i] = xy[gi][i].y; dm[i] = dem[gi][i]; vs[i] = vis[gi][i]; }
#pragma unroll
    for (int i = 0; i < NPL; ++i) if (lane + 64 * i == idx) vs[i] = 1;  // tsp.py:86
    const int cu = cur[gi], de = dep[gi];
    const double dx = node_f64(px, cu) - node_f64(px, idx);
    const double dy = node_f64(py, cu) - node_f64(py, idx);
    const double dist = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
    double load = 1.0;
    if (IRP) {                                                  // irp.py:80-86
      load = load0[gi] - node_f64(dm, idx);
      if (idx == de) load = 1.0;
    }
    auto all_visited = [&]() {
      int ok = 1;
#pragma unroll
      for (int i = 0; i < NPL; ++i) ok &= vs[i];
      return __all(ok);
    };
    const bool done = all_visited();                            // before the fix-ups, tsp.py:95
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      if (lane + 64 * i == de) {
        if (idx == de) vs[i] = 1;                               // tsp.py:141-142
        else if (p.kind != VRP_KIND_TSP) vs[i] = 0;             // vrp.py:28-31
      }
    }
    if (all_visited()) {                                        // tsp.py:145-146
#pragma unroll
      for (int i = 0; i < NPL; ++i) if (lane + 64 * i == de) vs[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      int mk = vs[i];
      if (IRP && inN[i] && dm[i] - load > 0.0) mk = 1;            // irp.py:151-153
      if (inN[i]) {
        p.env.visited[(size_t)b * N + lane + 64 * i] = (uint8_t)vs[i];
        mask_out[(size_t)b * N + lane + 64 * i] = (uint8_t)mk;
      }
    }
    if (lane == 0) {
      p.env.cur[b] = idx;
      if (IRP) p.env.load[b] = load;
      p.io.acc_loss[b] = accl[gi] + (float)(-dist);  // fp32 accumulate in step order, tsp_agent:85
      p.io.acc_logp[b] = accp[gi] + logp;
      p.last[b] = idx;
      if (p.t == 0) p.first[b] = idx;
      if (!done) p.io.notdone[p.t] = 1;
      if (p.io.actions) p.io.actions[(size_t)p.t * B + b] = idx;
      if (p.io.step_logp) p.io.step_logp[(size_t)p.t * B + b] = logp;
    }
    // latency mode of the table kernel reads next step's row from `curs`: keep it current
    // -- unless the host's schedule gives the next step to this kernel as well
    if (B <= 2048 && !p.skip_curs &&
        !(p.t == 0 && p.kind != VRP_KIND_IRP)) {
      const size_t row = (size_t)b * 8 * N;
      const float *arow = p.SL + ((size_t)b * N + idx) * 8 * N;
#pragma unroll
      for (int i = 0; i < NPL; ++i)
#pragma unroll
        for (int h = 0; h < 8; ++h)
          if (inN[i])
            p.curs[row + h * N + lane + 64 * i] =
                arow[h * N + lane + 64 * i] + (p.base ? p.base[row + h * N + lane + 64 * i] : 0.f);
    }
  }
}


// (Round 4, measured and removed: the tile RESIDENT for a whole episode -- steps 1..T-1 of VRP-100 x
// 2048 in ONE launch, the tile in registers across steps, masks exchanged as hand-off words as in
// decoder_persistent.hip: 3.52 ms for 111 steps = 31.7 us per step against 31.5 for the per-step
// schedule; hipcc spilled 92 registers inside the step loop.  docs/rounds/DESIGN_rounds_1-5.md 3.5.1; the kernel is in the
// history of this file, commit 2458ff7.)

template <int NMAX, int GPW, bool IRP>
static int launch_tile2(const StepParams &p, hipStream_t st) {
  constexpr int GPB = 8 * GPW;
  constexpr int KL = NMAX > 64 ? 2 : 0;
  const size_t lds = sizeof(float) * ((size_t)GPB * NMAX * 8 + GPB * (T2_ZG + T2_OS + T2_WS) +
                                      (size_t)GPB * T2_US + (size_t)GPB * KL * 512);
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_step_tile_zmfma_kernel<NMAX, GPW, IRP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("decode_step_tile2: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  hipLaunchKernelGGL((decode_step_tile_zmfma_kernel<NMAX, GPW, IRP>), dim3((p.B + GPB - 1) / GPB),
                     dim3(512), lds, st, p);
  VRP_CHECK_LAUNCH("decode_step_tile_zmfma");
  return 0;
}

bool vrp_tile2_supported(int N) { return N <= 100; }

int vrp_launch_tile2_step(const StepParams &p, hipStream_t st) {
  if (p.kind == VRP_KIND_IRP)
    return p.N <= 40 ? launch_tile2<40, 2, true>(p, st) : launch_tile2<100, 1, true>(p, st);
  return p.N <= 40 ? launch_tile2<40, 2, false>(p, st) : launch_tile2<100, 1, false>(p, st);
}

// Decode + env step from the RAW embedding tile (the "streaming formulation" of SURVEY.md
// 8d): a graph-step reads its (N,128) fp32 tile ONCE -- exactly the algorithmic 512 N bytes --
// instead of one 32 N-byte row of the pointer-logit table per selectable node.  It wins
// wherever many nodes are still selectable: vrp_decode_step launches it for the first steps of
// an episode (64 < N <= 104: about two thirds of them; N <= 40, large batches: while every graph
// keeps most of its nodes) and the table kernel (decoder.hip) afterwards.
//
// Same algebra as the table kernel:
//   a[h][n]   glimpse weights from the score rows (row0 | SL[last] + base) + scrambled masks
//   z_h       = sum_n a[h][n] e_n                             (8 x 128)   VALU, lane = 2 dims
//   o_h       = Wv_h z_h + bv_h,  w = M o + mb                matrix cores, 16 graphs per tile
//   u_n       = 10 tanh(e_n . w + cvec_n)                     VALU + cross-lane reduce-scatter
// Workgroup = 8 waves = 8*GPW graphs.  A wave keeps its graphs' tiles in registers (lane =
// two embedding columns: e[n] is a float2; filled two rows per 16-byte load from the row-paired
// copy DecWs::embP where the prologue built one, else row by row) through both passes over them.
// The two weight folds are batched over the workgroup's graphs as the 16 rows of
// v_mfma_f32_16x16x4_f32 (A = z / o rows from LDS, B = weight fragments streamed from L2 in
// MFMA fragment order, Derived::WvP/MP: the 384 KB of folded weights are shared by every
// workgroup and step).  The partial dot products
// e_n . w of a lane's two columns are summed over the 64 lanes by a butterfly reduce-scatter
// that leaves node n's total in lane n -- the layout the action/env code wants.
#include "decoder_step.h"

#define TL_ZG 1028  // zs: floats between graphs (8 heads x 128 + 4: conflict-free b128 reads)
#define TL_OS 388   // os: floats per graph row (384 + 4)
#define TL_WS 132   // ws: floats per graph row

// sum over the 64 lanes of v[i] (i = node 0..63 of this block) -> lane n returns node n's total
template <int LEN>
__device__ __forceinline__ float reduce_scatter64(float (&v)[LEN], int lane) {
  if constexpr (LEN == 1) {
    return v[0];
  } else {
    constexpr int H = LEN / 2;           // lanes with bit H set keep the upper half
    const bool up = (lane & H) != 0;
    float nv[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const float keep = up ? v[i + H] : v[i];
      const float send = up ? v[i] : v[i + H];
      nv[i] = keep + __shfl_xor(send, H, 64);
    }
    return reduce_scatter64<H>(nv, lane);
  }
}

// NW = waves per workgroup.  8: one 16-graph (GPW = 2) or 8-graph workgroup fills a CU, every
// phase of its graphs runs in lockstep.  4: half the graphs per workgroup (the MFMA tiles run with
// 8 or 4 of their 16 rows in use) and TWO workgroups per CU that drift apart, so the tile loads of
// one run under the folds and logits of the other.
template <int NMAX, int GPW, int NW>
__global__ __launch_bounds__(64 * NW, 2) void decode_step_tile_mfma_kernel(StepParams p) {
  constexpr int NPL = (NMAX + 63) / 64;
  constexpr int GPB = NW * GPW;  // graphs per workgroup (<= 16 = rows of one MFMA tile)
  constexpr int ROWS = NW == 8 ? 16 : GPB;   // rows of the LDS operand images (power of two)
  constexpr int HPW = 8 / NW;                // heads (fold 1) / 16-column tiles (fold 2) per wave
  if (!p.decode_only && p.t > 0 && p.io.notdone[p.t - 1] == 0) return;
  // Every workgroup alternates between a phase that only loads (the tile) and phases that only
  // compute; launched together they do so in lockstep and the memory system idles while the
  // CUs compute.  Every other workgroup of an XCD (blocks b and b + 8 share one) starts late, so
  // that one half of the chip loads at up to twice its share while the other half computes.
  if (p.stagger > 0 && ((blockIdx.x >> 3) & 1))
    for (int i = 0; i < p.stagger; i += 64) __builtin_amdgcn_s_sleep(64);

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *a_s = smem;                      // [GPB][NMAX*8]  a[g][n][h]
  float *zs = a_s + GPB * NMAX * 8;       // [ROWS][TL_ZG]  z[g][h][128]
  float *os = zs + ROWS * TL_ZG;          // [ROWS][TL_OS]  o[g][384]
  float *ws = os + ROWS * TL_OS;          // [ROWS][TL_WS]  w[g][128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, B = p.B;
  const int par = p.t & 1;
  const uint8_t *mask_in = p.env.mask + (size_t)par * B * N;
  uint8_t *mask_out = p.env.mask + (size_t)(par ^ 1) * B * N;
  const int i16 = lane & 15, q = lane >> 4;

  // ---- per-graph state kept across the matrix phase ---------------------------------------
  float2 e[GPW][NMAX];
  int own_mask[GPW][NPL], vis[GPW][NPL];
  double2 xy[GPW][NPL];
  double dem[GPW][NPL];
  float cv[GPW][NPL], q_noise[GPW][NPL];
  int cur[GPW], dep[GPW], bg[GPW];
  double load0[GPW];
  float accl[GPW], accp[GPW];
  bool proc[GPW];
  // env row, noise, accumulators of a graph (lane = node).  Two nodes per lane (N > 64): the
  // tile alone takes 208 registers, so these are fetched after the matrix phase (in flight
  // during the reduce-scatter) instead of being carried through it.
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
        if (p.kind == VRP_KIND_IRP) dem[gi][i] = p.env.demand[(size_t)b * N + l];
      }
      q_noise[gi][i] = !p.sample ? 1.f
                       : p.io.noise ? p.io.noise[((size_t)p.t * B + b) * N + l]
                                    : vrp_exp1_noise(p.io.noise_seed, p.t, b, l);
#ifdef VRP_MUTATION_NOISE_SHIFT  // test-the-tests build: off-by-one noise index
      if (p.sample && p.io.noise) q_noise[gi][i] = p.io.noise[((size_t)p.t * B + b) * N + (l + 1) % N];
#endif
    }
    cur[gi] = p.decode_only ? 0 : p.env.cur[b];
    dep[gi] = p.decode_only ? 0 : p.env.depot[b];
    accl[gi] = accp[gi] = 0.f;
    if (!p.decode_only) { accl[gi] = p.io.acc_loss[b]; accp[gi] = p.io.acc_logp[b]; }
  };

  // (the host picks ONE kernel per step from the step number, launch_step_any: nothing here
  // waits for the mask rows -- the score rows and the tile are requested in the same round trip)
  int lastn[GPW];
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int braw = blockIdx.x * GPB + wave * GPW + gi;
    const bool active = braw < B;
    const int b = __builtin_amdgcn_readfirstlane(active ? braw : B - 1);
    bg[gi] = b;
    lastn[gi] = p.t > 0 ? p.last[b] : 0;   // requested with the mask rows, not behind them
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const bool in = lane + 64 * i < N;
      own_mask[gi][i] = mask_in[(size_t)b * N + (in ? lane + 64 * i : 0)];
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
    // ---- loads.  Order matters (they return in order): the score rows and masks first, then
    // the first half of the tile; the glimpse weights are computed while the tile streams in,
    // the second half is requested behind them and the sums below consume rows as they land.
    constexpr int NH = NMAX / 2;
    const size_t row = (size_t)b * 8 * N;
    const float *srow = p.row0 + row;
    if (p.t > 0) {
      const int last = __builtin_amdgcn_readfirstlane(lastn[gi]);
      srow = p.SL + ((size_t)b * N + last) * 8 * N;
    }
    const bool add_base = p.base && p.t > 0;
    load0[gi] = (p.kind == VRP_KIND_IRP) ? p.env.load[b] : 1.0;
    float sv[NPL][8], bv_[NPL][8], sl_[NPL][8];
    int mo[NPL][8];
#pragma unroll
    for (int i = 0; i < NPL; ++i)
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        sv[i][h] = srow[h * N + ln[i]];
        bv_[i][h] = add_base ? p.base[row + h * N + ln[i]] : 0.f;
        sl_[i][h] = (p.kind == VRP_KIND_IRP) ? p.SLD[row + h * N + ln[i]] : 0.f;
        mo[i][h] = mask_in[(size_t)((b * 8 + h) % B) * N + ln[i]];  // QUIRK D3: other graphs
      }
    // the tile: from the row-paired copy two rows per 16-byte load (prologue: pair_rows_kernel),
    // else row by row
    const float2 *src = reinterpret_cast<const float2 *>(p.emb + (size_t)b * N * VRP_EMB) + lane;
    const float4 *srcP = p.embP ? reinterpret_cast<const float4 *>(p.embP) +
                                      (size_t)b * ((N + 1) / 2) * 64 + lane : nullptr;
    static_assert(NH % 2 == 0, "the first half of the tile is a whole number of row pairs");
    if (srcP) {
#pragma unroll
      for (int i = 0; i < NH / 2; ++i) {
        const float4 v = (2 * i < N) ? srcP[(size_t)i * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        e[gi][2 * i] = make_float2(v.x, v.y);
        e[gi][2 * i + 1] = make_float2(v.z, v.w);
      }
    } else {
#pragma unroll
      for (int n = 0; n < NH; ++n)
        e[gi][n] = (n < N) ? src[(size_t)n * 64] : make_float2(0.f, 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
    float sc[NPL][8];  // score + additive scrambled mask
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        float v = sv[i][h];
        if (add_base) v += bv_[i][h];
        if (p.kind == VRP_KIND_IRP) v = fmaf((float)load0[gi], sl_[i][h], v);
        sc[i][h] = v + (float)mo[i][h];
      }
    }
    if (!DEFER) load_env(gi, b);
    if (p.dbg == 1) { if (e[gi][0].x + sc[0][0] == 123.f) p.curs[0] = 1.f; continue; }

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
    if (srcP) {
#pragma unroll
      for (int i = NH / 2; i < NMAX / 2; ++i) {
        const float4 v = (2 * i < N) ? srcP[(size_t)i * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        e[gi][2 * i] = make_float2(v.x, v.y);
        e[gi][2 * i + 1] = make_float2(v.z, v.w);
      }
    } else {
#pragma unroll
      for (int n = NH; n < NMAX; ++n)
        e[gi][n] = (n < N) ? src[(size_t)n * 64] : make_float2(0.f, 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
    // a_s of this graph is written and read by this wave only: LDS ops of one wave are ordered
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    // ---- z_h[2l..2l+1] = sum_n a[h][n] * e[n][2l..2l+1] -------------------------------------
    {
      float2 z[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) z[h] = make_float2(0.f, 0.f);
      const float4 *ap = reinterpret_cast<const float4 *>(a_s + (size_t)g * NMAX * 8);
      // the weights of node n + 1 are read (LDS broadcast) before node n's sixteen FMAs are
      // issued: left to the compiler, every node started with an LDS round trip
      float4 a0 = ap[0], a1 = ap[1];
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        const float4 c0 = a0, c1 = a1;
        if (n + 1 < NMAX) { a0 = ap[2 * n + 2]; a1 = ap[2 * n + 3]; }
        __builtin_amdgcn_sched_barrier(0);
        if (n < N) {
          const float av[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            z[h].x = fmaf(av[h], e[gi][n].x, z[h].x);
            z[h].y = fmaf(av[h], e[gi][n].y, z[h].y);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int h = 0; h < 8; ++h)
        *reinterpret_cast<float2 *>(zs + g * TL_ZG + h * 128 + 2 * lane) = z[h];
    }
  }
  if (p.dbg == 1 || p.dbg == 2) return;
  __syncthreads();

  // ---- matrix phase: o = Wv z + bv (wave = head), then w = M o + mb (wave = 16 columns) ----
  // The weight fragments come from L2 (~600 cycles): they are requested PF k-steps ahead of
  // their MFMAs, and the first ones of the second product before the barrier in between.
  constexpr int PF = 3;    // first product: a k-step is 12 MFMAs (384 cycles of matrix pipe)
  constexpr int PF2 = NMAX > 64 ? 5 : 8;   // second product: 4 MFMAs per k-step (128 cycles) against the same ~600-
                           // cycle fragment latency; the ring reuses the first product's registers
  // Inner dimensions are spread over the four 16-lane groups as k = 16 S + 4 q + e (S = k-step,
  // e = element of the lane's float4): the four lanes that read one weight row in one
  // instruction cover 64 consecutive bytes, so a fragment load touches 16 cache lines instead
  // of 64 (the vector memory pipe looks lines up one by one, and eight waves stream 384 KB of
  // weights through it in this phase).  Same permutation on the LDS operand.
  const int koff2 = 4 * q;
  const int arow_g = i16 & (ROWS - 1);  // MFMA row -> graph (rows beyond the workgroup's graphs
                                        // repeat them; their results are dropped)
  float4 mw[PF2];
#pragma unroll 1
  for (int hh = 0; hh < HPW; ++hh) {
    const int koff = 4 * q;
    const int h = wave * HPW + hh;
    // fragment (k4, c) of head h, in MFMA operand order (Derived::WvP): 64 consecutive float4
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
    const float *arow = zs + arow_g * TL_ZG + h * 128 + koff;   // A row = graph
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
        if (4 * q + r4 < ROWS) os[(4 * q + r4) * TL_OS + h * VRP_HD + 16 * c + i16] = acc[c][r4];
  }
  {  // the second product's first fragments travel across the barrier
    const float4 *mrow0 = reinterpret_cast<const float4 *>(p.MP) + (size_t)(wave * HPW) * 24 * 64 + lane;
#pragma unroll
    for (int j = 0; j < PF2; ++j) mw[j] = mrow0[j * 64];
  }
  __syncthreads();
#pragma unroll 1
  for (int cc = 0; cc < HPW; ++cc) {
    const int ct = wave * HPW + cc;
    const float4 *mrow = reinterpret_cast<const float4 *>(p.MP) + (size_t)ct * 24 * 64 + lane;
    if (cc > 0) {
#pragma unroll
      for (int j = 0; j < PF2; ++j) mw[j] = mrow[j * 64];
    }
    const float mbv = p.mb[ct * 16 + i16];
    f32x4 acc0 = {mbv, mbv, mbv, mbv}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: k4 even / odd
    const float *arow = os + arow_g * TL_OS + koff2;
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
      if (4 * q + r4 < ROWS) ws[(4 * q + r4) * TL_WS + ct * 16 + i16] = acc0[r4] + acc1[r4];
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
    const float2 wv = *reinterpret_cast<const float2 *>(ws + g * TL_WS + 2 * lane);
    if (DEFER) load_env(gi, b);  // in flight during the reduce-scatter
    float u[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      float pv[64];
#pragma unroll
      for (int k = 0; k < 64; ++k) {
        const int n = 64 * i + k;
        pv[k] = (n < NMAX) ? fmaf(wv.x, e[gi][n < NMAX ? n : 0].x, wv.y * e[gi][n < NMAX ? n : 0].y) : 0.f;
      }
      const float x = reduce_scatter64<64>(pv, lane);
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
    if (p.kind == VRP_KIND_IRP) {                               // irp.py:80-86
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
      if (p.kind == VRP_KIND_IRP && inN[i] && dm[i] - load > 0.0) mk = 1;  // irp.py:151-153
      if (inN[i]) {
        p.env.visited[(size_t)b * N + lane + 64 * i] = (uint8_t)vs[i];
        mask_out[(size_t)b * N + lane + 64 * i] = (uint8_t)mk;
      }
    }
    if (lane == 0) {
      p.env.cur[b] = idx;
      if (p.kind == VRP_KIND_IRP) p.env.load[b] = load;
      p.io.acc_loss[b] = accl[gi] + (float)(-dist);  // fp32 accumulate in step order, tsp_agent:85
      p.io.acc_logp[b] = accp[gi] + logp;
      p.last[b] = idx;
      if (p.t == 0) p.first[b] = idx;
      if (!done) p.io.notdone[p.t] = 1;
      if (p.io.actions) p.io.actions[(size_t)p.t * B + b] = idx;
      if (p.io.step_logp) p.io.step_logp[(size_t)p.t * B + b] = logp;
    }
    // (last: a register reload from scratch behind these stores would wait for them to land)
    // latency mode of the table kernel reads next step's row from `curs`: keep it current
    // -- unless the host's schedule gives the next step to this kernel as well, which reads SL
    // and base itself (the update costs 7 us per step at 2048 x 100: 3.2 KB of table row per
    // graph from HBM, at the very end of the workgroup)
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

template <int NMAX, int GPW, int NW>
static int launch_tile(const StepParams &p, hipStream_t st) {
  constexpr int GPB = NW * GPW;
  constexpr int ROWS = NW == 8 ? 16 : GPB;
  const size_t lds = sizeof(float) * ((size_t)GPB * NMAX * 8 + ROWS * (TL_ZG + TL_OS + TL_WS));
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_step_tile_mfma_kernel<NMAX, GPW, NW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("decode_step_tile: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  hipLaunchKernelGGL((decode_step_tile_mfma_kernel<NMAX, GPW, NW>), dim3((p.B + GPB - 1) / GPB),
                     dim3(64 * NW), lds, st, p);
  VRP_CHECK_LAUNCH("decode_step_tile_mfma");
  return 0;
}

bool vrp_tile_mfma_supported(int N) { return N <= 104; }

// VRP_TILE_WAVES=4: two 4-wave workgroups per CU instead of one of eight (A/B aid; measured
// SLOWER -- 132 vs 72 us at 8192 x 40, 87 vs 55 us at 2048 x 100: with half the MFMA rows in
// use the weight folds, which stream 384 KB of weights per workgroup from L2, cost twice as much)
static int tile_waves() {
  static const int v = getenv("VRP_TILE_WAVES") ? atoi(getenv("VRP_TILE_WAVES")) : 8;
  return v;
}

int vrp_launch_tile_mfma_step(const StepParams &p, hipStream_t st) {
  // round 4: the tile in MFMA operand order, glimpse sums on the matrix cores (decoder_tile2.hip)
  if (!tile_v1_forced() && vrp_tile2_supported(p.N)) return vrp_launch_tile2_step(p, st);
  if (tile_waves() == 8)   // N <= 100 (configs[4]) has its own instance: 8 registers less of tile
    return p.N <= 40    ? launch_tile<40, 2, 8>(p, st)
           : p.N <= 100 ? launch_tile<100, 1, 8>(p, st)
                        : launch_tile<104, 1, 8>(p, st);
  return p.N <= 40 ? launch_tile<40, 2, 4>(p, st) : launch_tile<104, 1, 4>(p, st);
}

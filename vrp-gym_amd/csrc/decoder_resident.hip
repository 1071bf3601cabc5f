// Steps 1 .. T-1 of an episode in ONE launch with the node embeddings RESIDENT IN REGISTERS, for
// 64 < N <= 104 and B <= 8 x (usable CUs) -- BASELINE configs[4]'s per-GPU shard (VRP, N = 100,
// B = 2048): eight graphs per CU x 51 KB of embeddings = 410 KB of the CU's 512 KB vector
// register file.  One iteration of the rollout loop agents/graph_tsp_agent.py:78-88
// (GraphDecoder.forward agents/graph_decoder.py:51-115 + env.step gym_vrp/envs/tsp.py:60-101)
// then touches HBM only for one 32 N-byte glimpse score row per graph (the table row of the
// node just chosen) and the eight mask words it needs from other graphs; the (N,128) tile, the
// coordinates, the visited row and the accumulators never leave the chip between steps.
//
// Workgroup = 4 waves (one per SIMD, up to 512 registers each) = 8 graphs, ONE workgroup per CU,
// every workgroup of the grid resident (vrp_resident_eligible: B <= 8 x CUs found by the census
// kernel of decoder_persistent.hip).  Per step and graph, same algebra as decoder_tile.hip:
//   a[h][n]  glimpse weights: score row SL[b][last] + base[b] + scrambled masks, softmax over n
//   z_h      = sum_n a[h][n] e_n                  VALU; lane = (row parity, 4 columns): the tile
//                                                 is held as float4 = 16-byte loads at set-up
//   o = Wv z + bv,  w = M o + mb                  matrix cores, v_mfma_f32_16x16x4_f32, rows =
//                                                 the workgroup's 8 graphs; the folded weights
//                                                 stream from L2 in fragment order (Derived::WvP/MP:
//                                                 every load is 1 KB contiguous)
//   u_n      = 10 tanh(e_n . w + cvec_n)          VALU + reduce-scatter inside each 32-lane half
// Between graphs: the hand-off of decoder_persistent.hip (QUIRK D3, graph_decoder.py:93-94: head
// h of graph b adds the mask row of graph (8b + h) mod B), two 8-byte words per graph and step
// (63 nodes + the valid bit each), and its per-graph termination / forced-way-back rule
// (persistent_finalize_kernel closes the episode).  TSP and VRP (IRP's
// capacity overlay keeps the table kernel).  Bounded spins; a wave that gives up flags the
// episode (NaN accumulators, decoder_persistent.hip).
#include "decoder_step.h"

struct ResidentParams {
  StepParams s;               // s.t = first step of the launch (>= 1)
  unsigned long long *hist;   // (hist_rows(N), B): row 2 t + w = word w of every graph's step-t mask
  int32_t *ta;                // (B) step at which a graph's visited row became all ones
  float *ret;                 // (B) reward of the forced way back after ta (0 if none)
  int32_t *wb_cur;            // (B) node the graph stood on before its way back (-1: none)
  double *wb_load;            // (B) unused here (IRP), written as 1
  int32_t *err;
};

#define RS_VALID (1ull << 63)
#define RS_SPIN_LIMIT (1 << 20)
#define RS_ZG 1028  // zs: floats between graphs (8 heads x 128 + 4: conflict-free b128 reads)
#define RS_OS 388   // os: floats per graph row (384 + 4)
#define RS_WS 132   // ws: floats per graph row

// sum over the 32 lanes of a half of v[j] (slot j = 0..LEN-1) -> lane j of the half returns slot
// j's total
template <int LEN>
__device__ __forceinline__ float reduce_scatter32(float (&v)[LEN], int c4) {
  if constexpr (LEN == 1) {
    return v[0];
  } else {
    constexpr int H = LEN / 2;           // lanes with bit H set keep the upper half of the slots
    const bool up = (c4 & H) != 0;
    float nv[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const float keep = up ? v[i + H] : v[i];
      const float send = up ? v[i] : v[i + H];
      nv[i] = keep + __shfl_xor(send, H, 64);
    }
    return reduce_scatter32<H>(nv, c4);
  }
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

template <int NR>  // row pairs held per graph: 2 NR >= N rows
__global__ __launch_bounds__(256, 1) void decode_resident_kernel(ResidentParams pp) {
  constexpr int NMAX = 2 * NR;
  constexpr int GPW = 2;                       // graphs per wave
  const StepParams &p = pp.s;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *a_s = smem;                           // [8][NMAX*8]  a[g][n][h]
  float *base_s = a_s + 8 * NMAX * 8;          // [8][8*NMAX]  base[g][h][n] (constant after step 0)
  float *cv_s = base_s + 8 * NMAX * 8;         // [8][128]     cvec[g][n]
  double2 *xy_s = reinterpret_cast<double2 *>(cv_s + 8 * 128);  // [8][128] coordinates
  float *zs = reinterpret_cast<float *>(xy_s + 8 * 128);        // [8][RS_ZG]  z[g][h][128]
  float *os = zs + 8 * RS_ZG;                  // [8][RS_OS]   o[g][384]
  float *ws = os + 8 * RS_OS;                  // [8][RS_WS]   w[g][128]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c4 = lane & 31;
  const int i16 = lane & 15, q = lane >> 4;
  const int N = p.N, B = p.B, t0 = p.t;
  if (p.io.notdone[t0 - 1] == 0) return;  // the batch was done before this launch (grid-uniform)

  // ---- per-graph state, loaded once -----------------------------------------------------------
  float4 e[GPW][NR];
  float sv[GPW][2][8];       // the coming step's score row SL[b][last]
  int own_mask[GPW][2], vis[GPW][2], msk[GPW][2];  // msk: bit h = mask of head h's graph at my node
  int bg[GPW], cur[GPW], dep[GPW], last[GPW], ta[GPW], wb_cur[GPW];
  float accl[GPW], accp[GPW], ret[GPW];
  bool act[GPW], fin[GPW];
  const uint8_t *mask0 = p.env.mask + (size_t)(t0 & 1) * B * N;
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    const int g = wave * GPW + gi;
    const int braw = blockIdx.x * 8 + g;
    act[gi] = braw < B;
    const int b = __builtin_amdgcn_readfirstlane(act[gi] ? braw : B - 1);
    bg[gi] = b;
    fin[gi] = !act[gi];
    const float4 *src = reinterpret_cast<const float4 *>(p.emb + (size_t)b * N * VRP_EMB) + c4;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = 2 * i + half;
      e[gi][i] = (r < N) ? src[(size_t)r * 32] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    last[gi] = __builtin_amdgcn_readfirstlane(p.last[b]);
    cur[gi] = p.env.cur[b];
    dep[gi] = p.env.depot[b];
    accl[gi] = p.io.acc_loss[b];
    accp[gi] = p.io.acc_logp[b];
    ta[gi] = -1;
    wb_cur[gi] = -1;
    ret[gi] = 0.f;
    const size_t row = (size_t)b * 8 * N;
    const float *srow = p.SL + ((size_t)b * N + last[gi]) * 8 * N;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      const int n = lane + 64 * i2;
      const bool in = n < N;
      const int ln = in ? n : 0;
      own_mask[gi][i2] = mask0[(size_t)b * N + ln];
      vis[gi][i2] = in ? p.env.visited[(size_t)b * N + ln] : 1;
      int m = 0;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        sv[gi][i2][h] = srow[h * N + ln];
        m |= (int)mask0[(size_t)((b * 8 + h) % B) * N + ln] << h;  // QUIRK D3: other graphs
        if (in) base_s[g * NMAX * 8 + h * N + n] = p.base[row + h * N + n];
      }
      msk[gi][i2] = m;
      cv_s[g * 128 + n] = in ? p.cvec[(size_t)b * N + ln] : 0.f;
      xy_s[g * 128 + n] = reinterpret_cast<const double2 *>(p.env.pos)[(size_t)b * N + ln];
      // glimpse weights of the padding rows stay zero for the whole episode
      if (!in && n < NMAX) {
        *reinterpret_cast<float4 *>(a_s + g * NMAX * 8 + n * 8) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(a_s + g * NMAX * 8 + n * 8 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  // (LDS rows of a graph are written and read by its own wave only, except zs/os/ws)

  constexpr int PF = 3;  // weight fragments requested this many k-steps ahead of their MFMAs
  const float4 *WvP4 = reinterpret_cast<const float4 *>(p.WvP) + lane;
  const float4 *MP4 = reinterpret_cast<const float4 *>(p.MP) + lane;

  for (int t = t0; t < p.max_steps; ++t) {
    // ---- the eight other graphs' masks of this step (first step: the byte rows read above) ----
    if (t > t0) {
      unsigned long long w = RS_VALID;
      const int gsel = (lane >> 4) & 1;
      const bool polls = lane < 32 && !(gsel ? fin[1] : fin[0]);
      if (polls) {
        const int b = gsel ? bg[1] : bg[0];
        const int h = (lane & 15) >> 1;
        const unsigned long long *src =
            pp.hist + (size_t)(2 * t + (lane & 1)) * B + (b * 8 + h) % B;
        int spins = 0;
        w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (!(w & RS_VALID)) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > RS_SPIN_LIMIT) break;
          w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (__any(!(w & RS_VALID))) {
        // gave up (grid not fully resident, or a graph we depend on gave up): flag the episode,
        // let everybody who waits for OUR graphs go on; the wave keeps joining the barriers
        if (lane == 0) __hip_atomic_store(pp.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int gi = 0; gi < GPW; ++gi) {
          if (fin[gi]) continue;
          for (int r = 2 * (t + 1) + lane; r <= 2 * p.max_steps + 1; r += 64)
            __hip_atomic_store(pp.hist + (size_t)r * B + bg[gi], RS_VALID | 1ull, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          fin[gi] = true;
        }
        w |= RS_VALID;
      }
#pragma unroll
      for (int gi = 0; gi < GPW; ++gi) {
        int m0 = 0, m1 = 0;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const unsigned long long w0 = readlane_u64(w, 16 * gi + 2 * h);
          const unsigned long long w1 = readlane_u64(w, 16 * gi + 2 * h + 1);
          // nodes 0..62 sit in word 0, nodes 63..125 in word 1 (bit 63 of each = valid)
          m0 |= (int)(((lane < 63 ? w0 >> lane : w1) & 1ull) << h);
          m1 |= (int)(((w1 >> ((lane + 1) & 63)) & 1ull) << h);   // node 64 + lane (< 126)
        }
        if (!fin[gi]) { msk[gi][0] = m0; msk[gi][1] = m1; }
      }
    }

    // ---- per graph: glimpse weights -> a_s, z = A E -> zs ---------------------------------------
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      if (fin[gi]) continue;  // wave-uniform
      const int g = wave * GPW + gi;
      const int b = bg[gi];
      bool inN[2];
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) inN[i2] = lane + 64 * i2 < N;
      if (p.io.mask_trace) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          if (inN[i2])
            p.io.mask_trace[((size_t)t * B + b) * N + lane + 64 * i2] = (uint8_t)own_mask[gi][i2];
      }
      {
        float s[2][8], mx = -INFINITY;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int h = 0; h < 8; ++h) {
            float v = sv[gi][i2][h] + base_s[g * NMAX * 8 + h * N + (inN[i2] ? lane + 64 * i2 : 0)];
            v = inN[i2] ? v + (float)((msk[gi][i2] >> h) & 1) : -INFINITY;
            s[i2][h] = v;
            mx = fmaxf(mx, v);
          }
        const float M = wave_max(mx);
        float ev[2][8], sum[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          sum[h] = 0.f;
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) {
            ev[i2][h] = inN[i2] ? exp_nonpos(s[i2][h] - M) : 0.f;
            sum[h] += ev[i2][h];
          }
        }
        wave_sum8(sum);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          if (!(sum[h] > 1e-30f)) {  // wave-uniform, practically never: per-head maximum
            float hm = fmaxf(s[0][h], s[1][h]);
            hm = wave_max(hm);
            float es = 0.f;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) { ev[i2][h] = inN[i2] ? exp_nonpos(s[i2][h] - hm) : 0.f; es += ev[i2][h]; }
            sum[h] = wave_sum(es);
          }
          float r = __builtin_amdgcn_rcpf(sum[h]);
          r = fmaf(fmaf(-sum[h], r, 1.f), r, r);
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2)
            if (inN[i2]) a_s[g * NMAX * 8 + (lane + 64 * i2) * 8 + h] = ev[i2][h] * r;
        }
      }
      // a_s of this graph is written and read by this wave only: LDS ops of one wave are ordered
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
      // z_h[4 c4 .. +3] = sum_n a[h][n] e[n][4 c4 .. +3]: each half sums its own row parity, four
      // heads at a time (register budget), the two halves are added across lanes
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        float4 z[4];
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) z[hh] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float *ap = a_s + g * NMAX * 8 + half * 8 + 4 * hp;   // row 2 i + half
        float4 an = *reinterpret_cast<const float4 *>(ap);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const float4 ac = an;
          if (i + 1 < NR) an = *reinterpret_cast<const float4 *>(ap + (i + 1) * 16);
          const float av[4] = {ac.x, ac.y, ac.z, ac.w};
#pragma unroll
          for (int hh = 0; hh < 4; ++hh) {
            z[hh].x = fmaf(av[hh], e[gi][i].x, z[hh].x);
            z[hh].y = fmaf(av[hh], e[gi][i].y, z[hh].y);
            z[hh].z = fmaf(av[hh], e[gi][i].z, z[hh].z);
            z[hh].w = fmaf(av[hh], e[gi][i].w, z[hh].w);
          }
        }
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {
          z[hh].x += __shfl_xor(z[hh].x, 32, 64);
          z[hh].y += __shfl_xor(z[hh].y, 32, 64);
          z[hh].z += __shfl_xor(z[hh].z, 32, 64);
          z[hh].w += __shfl_xor(z[hh].w, 32, 64);
          if (half == 0)
            *reinterpret_cast<float4 *>(zs + g * RS_ZG + (4 * hp + hh) * 128 + 4 * c4) = z[hh];
        }
      }
    }
    __syncthreads();

    // ---- matrix phase: o = Wv z + bv (wave = two heads), w = M o + mb (wave = two 16-column
    // tiles); rows 8..15 of the MFMA tiles repeat the eight graphs and are dropped ---------------
    const int arow_g = i16 & 7;
#pragma unroll 1
    for (int hh = 0; hh < 2; ++hh) {
      const int h = wave * 2 + hh;
      const float4 *wb = WvP4 + (size_t)h * 24 * 64;   // fragment (k4, c) = wb[(3 k4 + c) * 64]
      float4 wq[PF][3];
#pragma unroll
      for (int j = 0; j < PF; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) wq[j][c] = wb[(3 * j + c) * 64];
      f32x4 acc[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float bb = p.bv[h * VRP_HD + 16 * c + i16];  // D column = lane & 15
        acc[c] = f32x4{bb, bb, bb, bb};
      }
      const float *arow = zs + arow_g * RS_ZG + h * 128 + 4 * q;   // A row = graph, k = 16 k4 + 4 q + e
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        const float4 a = *reinterpret_cast<const float4 *>(arow + 16 * k4);
        const float av[4] = {a.x, a.y, a.z, a.w};
        float4 w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          w[c] = wq[k4 % PF][c];
          if (k4 + PF < 8) wq[k4 % PF][c] = wb[(3 * (k4 + PF) + c) * 64];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], w[c].x, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], w[c].y, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], w[c].z, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], w[c].w, acc[c], 0, 0, 0);
        }
      }
      if (q < 2) {  // D: row = graph 4 q + r4, column = lane & 15
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
            os[(4 * q + r4) * RS_OS + h * VRP_HD + 16 * c + i16] = acc[c][r4];
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int cc = 0; cc < 2; ++cc) {
      const int ct = wave * 2 + cc;
      const float4 *mbp = MP4 + (size_t)ct * 24 * 64;   // fragment k4 = mbp[k4 * 64]
      float4 mw[PF];
#pragma unroll
      for (int j = 0; j < PF; ++j) mw[j] = mbp[j * 64];
      const float mbv = p.mb[ct * 16 + i16];
      f32x4 acc0 = {mbv, mbv, mbv, mbv}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains: k4 even / odd
      const float *arow = os + arow_g * RS_OS + 4 * q;
#pragma unroll
      for (int k4 = 0; k4 < 24; ++k4) {
        const float4 a = *reinterpret_cast<const float4 *>(arow + 16 * k4);
        const float4 w = mw[k4 % PF];
        if (k4 + PF < 24) mw[k4 % PF] = mbp[(k4 + PF) * 64];
        f32x4 &acc = (k4 & 1) ? acc1 : acc0;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
      }
      if (q < 2) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) ws[(4 * q + r4) * RS_WS + ct * 16 + i16] = acc0[r4] + acc1[r4];
      }
    }
    __syncthreads();

    // ---- per graph: pointer logits, action, env step, hand-off ----------------------------------
#pragma unroll
    for (int gi = 0; gi < GPW; ++gi) {
      if (fin[gi]) continue;  // wave-uniform
      const int g = wave * GPW + gi;
      const int b = bg[gi];
      bool inN[2];
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) inN[i2] = lane + 64 * i2 < N;
      const float4 wv = *reinterpret_cast<const float4 *>(ws + g * RS_WS + 4 * c4);
      float u[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        float pv[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int i = 32 * blk + j;
          if (i < NR) {
            const float4 ev = e[gi][i < NR ? i : 0];
            pv[j] = fmaf(wv.x, ev.x, fmaf(wv.y, ev.y, fmaf(wv.z, ev.z, wv.w * ev.w)));
          } else {
            pv[j] = 0.f;
          }
        }
        // lane (half, j) now holds e_n . w of node n = 64 blk + 2 j + half; bring node 64 blk + L
        // to lane L (the layout of the masks, the noise and the env code)
        const float x = reduce_scatter32<32>(pv, c4);
        const float xn = __shfl(x, (lane & 1) * 32 + (lane >> 1), 64);
        u[blk] = (inN[blk] && !own_mask[gi][blk])
                     ? 10.f * tanhf(xn + cv_s[g * 128 + lane + 64 * blk]) : -INFINITY;  // graph_decoder.py:97-98
      }
      float q_noise[2] = {1.f, 1.f};
      if (p.sample) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int ln = inN[i2] ? lane + 64 * i2 : 0;
          q_noise[i2] = p.io.noise ? p.io.noise[((size_t)t * B + b) * N + ln]
                                   : vrp_exp1_noise(p.io.noise_seed, t, b, ln);
        }
      }
      // lowest node index among the maxima (torch CPU argmax): slot 0 holds nodes < 64
      auto argmax_nodes = [&](const float (&v)[2]) {
        const float m = wave_max(fmaxf(v[0], v[1]));
        const unsigned long long h0 = __ballot(v[0] == m), h1 = __ballot(v[1] == m);
        return h0 ? __ffsll((long long)h0) - 1 : 64 + __ffsll((long long)h1) - 1;
      };
      int idx;
      float logp = 0.f;
      if (!p.sample) {
        idx = argmax_nodes(u);
      } else {
        // Categorical(logits=u): logits - logsumexp, probs = softmax, sample = argmax(p/q)
        const float m = wave_max(fmaxf(u[0], u[1]));
        const float se = wave_sum(expf(u[0] - m) + expf(u[1] - m));
        const float lse = m + logf(se);
        const float l[2] = {u[0] - lse, u[1] - lse};
        const float lm = wave_max(fmaxf(l[0], l[1]));
        const float pe[2] = {expf(l[0] - lm), expf(l[1] - lm)};
        const float ps = wave_sum(pe[0] + pe[1]);
        float ratio[2];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) ratio[i2] = inN[i2] ? (pe[i2] / ps) / q_noise[i2] : -1.f;
        idx = argmax_nodes(ratio);
        const float lsel = idx >= 64 ? l[1] : l[0];
        logp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lsel),
                                                                   idx & 63));
      }
      idx = __builtin_amdgcn_readfirstlane(idx);
      // next step's score row: requested now, consumed after the next hand-off
      {
        const float *srow = p.SL + ((size_t)b * N + idx) * 8 * N;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int h = 0; h < 8; ++h) sv[gi][i2][h] = srow[h * N + (inN[i2] ? lane + 64 * i2 : 0)];
      }
      // ---- env.step on registers (same operation order as env_device.h) -------------------
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) if (lane + 64 * i2 == idx) vis[gi][i2] = 1;  // tsp.py:86
      const double2 pc = xy_s[g * 128 + cur[gi]], pi = xy_s[g * 128 + idx];
      const double dx = pc.x - pi.x, dy = pc.y - pi.y;
      const double dist = sqrt(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
      const bool done = __all(vis[gi][0] && vis[gi][1]);         // before the fix-ups, tsp.py:95
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        if (lane + 64 * i2 == dep[gi]) {
          if (idx == dep[gi]) vis[gi][i2] = 1;                    // tsp.py:141-142
          else if (p.kind != VRP_KIND_TSP) vis[gi][i2] = 0;       // vrp.py:28-31
        }
      }
      if (__all(vis[gi][0] && vis[gi][1])) {                      // tsp.py:145-146
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) if (lane + 64 * i2 == dep[gi]) vis[gi][i2] = 0;
      }
      const unsigned long long bal0 = __ballot(inN[0] && vis[gi][0]);
      const unsigned long long bal1 = __ballot(inN[1] && vis[gi][1]);
      const unsigned long long word0 = (bal0 & ~RS_VALID) | RS_VALID;
      const unsigned long long word1 = (((bal1 << 1) | (bal0 >> 63)) & ~RS_VALID) | RS_VALID;
      const bool way_back = ta[gi] >= 0;  // this step is the forced return after `done`
      const bool finish = done && (way_back || p.kind == VRP_KIND_TSP || idx == dep[gi]);
      if (lane == 0) {
        if (p.io.actions) p.io.actions[(size_t)t * B + b] = idx;
        if (p.io.step_logp) p.io.step_logp[(size_t)t * B + b] = logp;
      }
      if (!finish && lane < 2)
        __hip_atomic_store(pp.hist + (size_t)(2 * (t + 1) + lane) * B + b, lane ? word1 : word0,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (way_back) {
        ret[gi] = (float)(-dist);
        wb_cur[gi] = cur[gi];   // where the episode ends if the batch was done at ta
      } else {
        accl[gi] += (float)(-dist);  // fp32 accumulate in step order, tsp_agent:85
        accp[gi] += logp;
        if (done) ta[gi] = t;
      }
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) own_mask[gi][i2] = vis[gi][i2];
      cur[gi] = idx;
      last[gi] = idx;
      if (finish) {
        // the mask is constant from here on: publish it for every remaining step, fill the traces
        // the way the reference's self-loops on the depot would
        for (int r = 2 * (t + 1) + lane; r <= 2 * p.max_steps + 1; r += 64)
          __hip_atomic_store(pp.hist + (size_t)r * B + b, (r & 1) ? word1 : word0, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        for (int tt = t + 1; tt < p.max_steps; ++tt) {
          if (p.io.mask_trace) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
              if (inN[i2])
                p.io.mask_trace[((size_t)tt * B + b) * N + lane + 64 * i2] = (uint8_t)vis[gi][i2];
          }
          if (lane == 0) {
            if (p.io.actions) p.io.actions[(size_t)tt * B + b] = idx;
            if (p.io.step_logp) p.io.step_logp[(size_t)tt * B + b] = 0.f;
          }
        }
        fin[gi] = true;
      }
    }
    // every wave keeps joining the barriers until the workgroup's eight graphs are finished
    if (__syncthreads_and(fin[0] && fin[1])) break;
  }

  // ---- state back to memory -----------------------------------------------------------------
#pragma unroll
  for (int gi = 0; gi < GPW; ++gi) {
    if (!act[gi]) continue;
    const int b = bg[gi];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
      if (lane + 64 * i2 < N) p.env.visited[(size_t)b * N + lane + 64 * i2] = (uint8_t)vis[gi][i2];
    if (lane == 0) {
      p.env.cur[b] = cur[gi];
      p.io.acc_loss[b] = accl[gi];
      p.io.acc_logp[b] = accp[gi];
      p.last[b] = last[gi];
      pp.ta[b] = ta[gi] < 0 ? p.max_steps - 1 : ta[gi];
      pp.ret[b] = ret[gi];
      pp.wb_cur[b] = wb_cur[gi];
      pp.wb_load[b] = 1.0;
    }
  }
}

static size_t resident_lds_bytes(int NR) {
  const size_t NMAX = 2 * (size_t)NR;
  return sizeof(float) * (2 * 8 * NMAX * 8 + 8 * 128 + 8 * (RS_ZG + RS_OS + RS_WS)) +
         sizeof(double2) * 8 * 128;
}

bool vrp_resident_eligible(int kind, int B, int N, int max_steps, int flags,
                           const vrp_rollout_io *io, hipStream_t st) {
  static const bool off = getenv("VRP_NO_RESIDENT") != nullptr;  // A/B aid
  if (off || kind == VRP_KIND_IRP || N <= 63 || N > 104 || max_steps < 2 || io->logits ||
      io->forced || io->load_trace ||
      (flags & (VRP_STEP_TILE_KERNEL | VRP_STEP_THROUGHPUT_KERNEL | VRP_STEP_DECODE_ONLY |
                VRP_STEP_TABLE_KERNEL | VRP_STEP_NO_PERSISTENT)))
    return false;
  static const bool force = getenv("VRP_PERSISTENT_FORCE") != nullptr;  // tests
  if (force) return true;
  // one 256-thread workgroup (all 512 registers of each SIMD lane) per CU, decided BEFORE the
  // episode starts: a grid that might not be resident takes the per-step path
  return (B + 7) / 8 <= vrp_usable_cus(st);
}

template <int NR>
static int launch_resident(const ResidentParams &rp, hipStream_t st) {
  const size_t lds = resident_lds_bytes(NR);
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_resident_kernel<NR>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("decode_resident: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  hipLaunchKernelGGL((decode_resident_kernel<NR>), dim3((rp.s.B + 7) / 8), dim3(256), lds, st, rp);
  VRP_CHECK_LAUNCH("decode_resident");
  return 0;
}

// steps sp.t .. max_steps-1 (sp.t >= 1: step 0 and the first-node fold have run)
int vrp_launch_resident_steps(const StepParams &sp, void *workspace, hipStream_t st) {
  DecWs ws = carve_decws(workspace, sp.B, sp.N);
  ResidentParams rp;
  rp.s = sp;
  rp.hist = ws.hist;
  rp.ta = ws.ta;
  rp.ret = ws.ret;
  rp.wb_cur = ws.wb_cur;
  rp.wb_load = ws.wb_load;
  rp.err = ws.err;
  void *token = nullptr;
  vrp_persistent_serialize_begin(st, &token);
  if (int r = sp.N <= 100 ? launch_resident<50>(rp, st) : launch_resident<52>(rp, st)) return r;
  if (int r = vrp_launch_persistent_finalize(sp, workspace, st)) return r;
  vrp_persistent_serialize_end(st, token);
  return 0;
}

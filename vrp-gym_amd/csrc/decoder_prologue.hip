// Per-episode constants of the decoder (D2/D3 hoisting): everything in
// GraphDecoder.forward (agents/graph_decoder.py:75-98) that does not depend on the step.
//
//   SL[b][m][h][n]  = (Wq_last e_m)_h . (Wk e_n + bk)_h / sqrt(48)    last-node part of the score
//   RT[b][m][h][n]  = (M^T e_m)_h . (Wv e_n + bv)_h                   pointer-logit table
//   SG, C0, SLD     = [Wq_g g + bq ; qc0 ; wload]_h . (Wk e_n + bk)_h / sqrt(48)
//   row0 = SG + C0 (the step-0 score row);  base = SG (IRP) or, after step 0 (TSP/VRP,
//   first_ known: graph_decoder.py:111-113), SG + (Wq_first e_first)_h . (Wk e_n)_h / sqrt(48)
//
// N <= 80: ONE kernel projects the embedding rows AND builds the tables: the projected rows
// ([QL | KK | KM | VV], 1536 floats per node: 2 GB at 8192 x 40) never exist in memory.
// N > 80: projection GEMM into PROJ, then pair_tables_kernel.
#include "decoder_ws.h"
#include "x3_common.h"

// graph embedding = mean over nodes (sum, then divide; graph_decoder.py:75-77) and
// cvec[b][m] = e_m . mb in one launch: one workgroup per graph, threads 0..127 own an
// embedding column of the mean, each wave owns every fourth node row of cvec.
// Also clears the persistent step kernel's hand-off words of this graph (hist[t][b], all t)
// and its error flag: every episode starts with "nothing published".
__global__ __launch_bounds__(256) void graph_mean_cvec_kernel(const float *__restrict__ emb,
                                                              const float *__restrict__ mb, int N,
                                                              float *__restrict__ g,
                                                              float *__restrict__ cvec,
                                                              unsigned long long *__restrict__ hist,
                                                              int32_t *__restrict__ err) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int r = tid; r < hist_rows(N); r += 256) hist[(size_t)r * gridDim.x + b] = 0ull;
  if (b == 0 && tid == 0) *err = 0;
  const float *eb = emb + (size_t)b * N * VRP_EMB;
  if (tid < VRP_EMB) {
    float s = 0.f;
#pragma unroll 8  // eight rows in flight; the adds stay in node order
    for (int n = 0; n < N; ++n) s += eb[(size_t)n * VRP_EMB + tid];
    g[(size_t)b * VRP_EMB + tid] = s / (float)N;
  }
  const float2 m = reinterpret_cast<const float2 *>(mb)[lane];
  for (int n = wave; n < N; n += 4) {
    const float2 e = reinterpret_cast<const float2 *>(eb + (size_t)n * VRP_EMB)[lane];
    const float s = wave_sum(fmaf(e.x, m.x, e.y * m.y));
    if (lane == 0) cvec[(size_t)b * N + n] = s;
  }
}

// ------------------------------------------------------------------ fused projection + tables
// Work unit = (pack of G consecutive graphs, head h).  A pack's G*N <= 80 embedding rows are
// RT <= 5 tiles of 16 rows of the flat (B*N,128) matrix (N = 20: four graphs in five tiles
// instead of eight; N = 40: two graphs in five instead of six).  One WAVE owns a unit from
// the embedding rows to the finished tables -- no workgroup barrier after the weights are
// staged:
//   stage 1  P_X^T (48 x 16RT) = W_{X,h} (48 x 128) E^T for X in {QL, KK, KM, VV}:
//            v_mfma_f32_16x16x4_f32 with the head's weight slices in LDS (A operand, one
//            ds_read_b128 per 4 k-steps, reused by all node tiles) and the embedding rows in
//            registers (B operand, lane = node: 32 contiguous floats of its row).  The
//            result sits transposed in the accumulators: lane (node j, group q) holds
//            projection columns 16c + 4q + reg of node j --
//   stage 2  -- which IS the operand layout of the next product: the tiles of SL^T = KK QL^T
//            and RT^T = VV KM^T run as 12 MFMAs per 16 x 16 tile straight from those
//            registers (k = projection column; both operands use the same column ->
//            (group, step) map).  Transposed, so that a lane ends up with four consecutive
//            key columns n of one query row m: one 16-byte store per tile and lane (N % 4 == 0;
//            element-wise stores otherwise).  Two accumulation chains per tile keep the
//            matrix pipe issuing back to back, and the store of a tile is issued under the
//            MFMAs of the next one.
// Two halves -- (QL, KK) -> score tables, (KM, VV) -> logit table -- keep at most 2 x 3 x RT
// accumulator tiles live.  Every global load of a pack is issued before the ~15 k cycles of
// stage-1 MFMAs that separate it from its first use (a SIMD runs ONE wave of this kernel: an
// exposed memory round trip is a stalled matrix pipe); the next pack's embedding rows are
// requested before the last stage 2 starts.
// IRP (no first-node term, graph_decoder.py:90-91): the constant score row SG is folded into
// SL here; TSP/VRP: the steps add base[b] (vrp_decode_first_row) to SL[b][last].
// The 128-long inner dimension of stage 1 is split over the four 16-lane groups as
// k = koff(q) + s, koff = {0, 64, 32, 96}: with LDS rows of 132 floats the two lane groups
// that one ds_read_b128 pass serves together (q = 0 with 1, 2 with 3) are 64 floats apart,
// which makes the 16-byte reads conflict-free.
static bool prologue_x3_enabled();
#define PT_LD 132
#define PT_MAXROWS 112
struct PrologueParams {
  int kind, B, N, G, npacks;
  const float *emb, *Wproj, *bproj, *QG, *qc0, *wload;
  const __bf16 *WprojX3;   // Derived::WprojX3 (the X3 instances' stage-1 weights)
  float *SG, *C0, *SLD, *row0, *SL, *RT;
  float *KK4;              // DecWs::KK4 or NULL
};

// RING (more than five row tiles: one graph of 80 < N <= 112 rows): the rows' inner dimension
// travels through a ring of two QUARTERS (k-steps 2 qi, 2 qi + 1 of the eight) instead of sitting
// in registers whole -- 8 RT_ instead of 32 RT_ registers next to the 24 RT_ accumulators of a half
// (seven tiles: 392 registers otherwise, plus the stage-2 temporaries, of the 512 a lone wave of
// a SIMD has).  Quarter qi + 1 is requested ahead of quarter qi's 48 RT_ MFMAs; both halves read
// the rows (the second time from L2).  Same k order per accumulator.
// X3 (round 5, up to five row tiles): stage 1 -- 86 % of the kernel's flops -- on the bf16 matrix
// cores.  The head's weight slices sit in LDS as pre-split fragments (Derived::WprojX3: a wave's
// operand read is 1 KB contiguous), the embedding rows are split once per pack in registers
// (three planes of the 32 k-values a lane holds per row tile: chunk j = values 8 j .. 8 j + 7)
// and reused by all twelve (projection, column tile) products: six v_mfma_f32_16x16x32_bf16 per
// 32 k instead of eight v_mfma_f32_16x16x4_f32 at a sixteenth of the rate.  The accumulator
// layout is the same, so stage 2 (fp32 MFMA straight from the accumulators) is untouched.
// Waves per workgroup of the X3 instances: EIGHT, two per SIMD, on the one copy of the head's weight
// fragments in LDS (147 KB: one workgroup per CU either way).  The fp32 instances need up to 392
// registers and run one wave per SIMD; an X3 instance of up to four row tiles takes 184 + its 24 RT_
// accumulators, so two waves fit the 512-entry file with 24 (three tiles) to 104 (four) registers
// spilled -- and still the second wave fills more of the first one's stalls than the spills cost:
// 1246 -> 1157 us at TSP 8192 x 40, 49.3 -> 45.4 us at 512 x 20 (one box, rocprofv3).
#ifndef PT_X3_WAVES
#define PT_X3_WAVES 8
#endif
// Developer aid (make EXTRA=-DVRP_PRO_TRACE): shader-clock stamps of one workgroup's waves at the
// phase boundaries of their first six packs, printed after the sixth launch of an instance
// (profiles/r06_prologue_trace.txt).
#ifdef VRP_PRO_TRACE
#define PTT_SLOTS 64
__device__ unsigned long long g_pro_trace[8 * PTT_SLOTS];
#define PT_MARK(i)                                                                         \
  if (blockIdx.x == 100 && lane == 0 && (i) < PTT_SLOTS) g_pro_trace[wave * PTT_SLOTS + (i)] = __builtin_amdgcn_s_memtime()
#else
#define PT_MARK(i)
#endif
// KEEPK (round 6): the small-batch instances also leave the glimpse keys in memory (DecWs::KK4,
// decoder_persistent.hip: persist_first_base).  A template parameter, not a run-time branch: the
// store's address arithmetic costs the three-tile bf16 instance eleven more spilled registers,
// which the large-batch launches (no keys kept) should not pay.
template <int RT_, bool VEC, bool RING = (RT_ > 5), bool X3 = false, bool KEEPK = false>
__global__ __launch_bounds__((X3 && !RING) ? 64 * PT_X3_WAVES : 256, 1) void prologue_tables_kernel(PrologueParams p) {
  // (RING + X3, round 6: the six / seven-tile graphs' projections on the bf16 matrix cores too -- a
  // ring quarter IS one 32-deep bf16 k-chunk of the fragment order; one wave per SIMD, as the fp32
  // RING instances: 7 x 24 accumulator registers + the rows' ring leave no room for a second)
  constexpr int NWV = (X3 && !RING) ? PT_X3_WAVES : 4, NTH = 64 * NWV;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *wl = lds;                              // [4][48][PT_LD] weight slices of this head, or
  const __bf16 *wl3 = reinterpret_cast<const __bf16 *>(lds);   // X3: [4][3] fragments of X3_FRAG bf16
  constexpr int WL_FLOATS = X3 ? 12 * X3_FRAG / 2 : 4 * 48 * PT_LD;
  float *bl = wl + WL_FLOATS;                   // [4][48] bias slices
  int *rowinfo = reinterpret_cast<int *>(bl + 4 * 48);  // [80] row of a pack -> graph << 8 | node

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j16 = lane & 15, q = lane >> 4;
  const int N = p.N, G = p.G;
  // blocks b and b + 8 share an XCD (round-robin dispatch): the eight heads of a pack run on
  // one XCD at about the same time and share its embedding rows through that L2
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int h = jj & 7, sub = jj >> 3, nsub = gridDim.x >> 6;
  const int stride = 8 * nsub * NWV;
  const int first = (xcd * nsub + sub) * NWV + wave;
  const bool fold = p.kind == VRP_KIND_IRP;

  const int koff = 64 * (q & 1) + 32 * (q >> 1);
  const float c48 = 0.14433756729740643f;  // 1/sqrt(48)
  float eq[RING ? 2 : 1][RT_][8];
  auto load_quarter = [&](int pack, int qi, int buf) {
    const size_t R0 = (size_t)pack * G * N;
    const int valid = min(G, p.B - pack * G) * N;
#pragma unroll
    for (int r = 0; r < RT_; ++r) {
      const int row = 16 * r + j16;
      const float4 *src = reinterpret_cast<const float4 *>(
          p.emb + (R0 + (row < valid ? row : 0)) * VRP_EMB + koff) + 2 * qi;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (row < valid) { v0 = src[0]; v1 = src[1]; }
      float *d = eq[RING ? buf : 0][r];
      d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w;
      d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
    }
  };
  float ef[RING ? 1 : RT_][32];
  auto load_pack = [&](int pack) {
    if constexpr (RING) { load_quarter(pack, 0, 0); return; }
    const size_t R0 = (size_t)pack * G * N;
    const int valid = min(G, p.B - pack * G) * N;
#pragma unroll
    for (int r = 0; r < RT_; ++r) {
      const int row = 16 * r + j16;
      const float4 *src = reinterpret_cast<const float4 *>(
          p.emb + (R0 + (row < valid ? row : 0)) * VRP_EMB + koff);
      // UNCONDITIONAL loads (a padding row reads the pack's row 0: finite values whose products are
      // never stored -- rows and columns of the tables do not mix): behind a branch the compiler
      // cannot count the requests in flight and makes the first use wait for everything, i.e. for
      // the table stores issued after these loads
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        const float4 v = src[k4];
        ef[r][4 * k4] = v.x; ef[r][4 * k4 + 1] = v.y; ef[r][4 * k4 + 2] = v.z; ef[r][4 * k4 + 3] = v.w;
      }
    }
  };

  if constexpr (X3) {
    // (nine 16-byte loads in flight per thread, then their LDS stores: written as a plain loop the
    // compiler kept ONE load in flight -- load, s_waitcnt vmcnt(0), ds_write, 18 times: eighteen L2
    // round trips in a row, ~10 us at the head of every launch; round 6, found in the ISA)
    const float4 *src = reinterpret_cast<const float4 *>(p.WprojX3 + (size_t)h * 12 * X3_FRAG);
    constexpr int NV = 12 * X3_FRAG / 8 / NTH;   // 18 (eight waves) or 36 (four)
    static_assert(NV % 9 == 0 && NV * NTH * 8 == 12 * X3_FRAG, "staging loop geometry");
#pragma unroll
    for (int b0 = 0; b0 < NV; b0 += 9) {
      float4 t[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) t[u] = src[tid + (b0 + u) * NTH];
#pragma unroll
      for (int u = 0; u < 9; ++u) reinterpret_cast<float4 *>(lds)[tid + (b0 + u) * NTH] = t[u];
    }
  } else {
    // (eight loads in flight per thread: see the bf16-plane branch above)
    constexpr int NV = 4 * 48 * 32 / NTH;   // 24
    static_assert(NV % 8 == 0, "staging loop geometry");
#pragma unroll
    for (int b0 = 0; b0 < NV; b0 += 8) {
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = tid + (b0 + u) * NTH, c4 = i & 31, row = (i >> 5) % 48, X = i / (48 * 32);
        t[u] = *reinterpret_cast<const float4 *>(p.Wproj + ((size_t)(X * 384 + h * 48 + row)) * 128 + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = tid + (b0 + u) * NTH, c4 = i & 31, row = (i >> 5) % 48, X = i / (48 * 32);
        *reinterpret_cast<float4 *>(wl + (X * 48 + row) * PT_LD + 4 * c4) = t[u];
      }
    }
  }
  // the first pack's rows: requested right behind the weights' loads (their LDS stores above have
  // retired the registers; hoisted in front of the staging the 96 row registers made the compiler
  // serialise the staging loads again)
  if (first < p.npacks) load_pack(first);
  if (tid < 4 * 48) bl[tid] = p.bproj[(tid / 48) * 384 + h * 48 + tid % 48];
  if (tid < PT_MAXROWS) rowinfo[tid] = ((tid / N) << 8) | (tid % N);
  // tile pairs (tm, tn) that contain two nodes of one graph (wave-uniform bit mask)
  unsigned long long needmask = 0;
  {
    const int rows = G * N;
#pragma unroll
    for (int tm = 0; tm < RT_; ++tm) {
      const int lo_m = (16 * tm) / N, hi_m = min(16 * tm + 15, rows - 1) / N;
#pragma unroll
      for (int tn = 0; tn < RT_; ++tn) {
        const int lo_n = (16 * tn) / N, hi_n = min(16 * tn + 15, rows - 1) / N;
        if (16 * tm < rows && 16 * tn < rows && lo_m <= hi_n && lo_n <= hi_m)
          needmask |= 1ull << (tm * RT_ + tn);
      }
    }
  }
  __syncthreads();
  PT_MARK(0);
  [[maybe_unused]] int pk_ = 0;

  for (int pack = first; pack < p.npacks; pack += stride, ++pk_) {
    PT_MARK(8 * pk_ + 1);
    // (the weight slices are re-read from LDS for every pack: keeping them in registers
    // across the loop would cost up to 384 VGPRs)
    asm volatile("" ::: "memory");
    const int g0 = pack * G;                       // first graph of the pack
    const int valid = min(G, p.B - g0) * N;        // rows of the pack that exist
    // D[n][m] tiles: lane (j, q) holds query row m = 16tm + j and the four consecutive key
    // columns n = 16tn + 4q + {0..3}
    int mg[RT_], moff[RT_];   // row m: graph of the pack (-2: none), element offset of its table row
    int ng[RT_], nn0[RT_];    // column 16tn + 4q: graph (-1: none) and node
#pragma unroll
    for (int t = 0; t < RT_; ++t) {
      const int row = 16 * t + j16;
      const int ri = rowinfo[row < PT_MAXROWS ? row : 0];
      mg[t] = (row < valid) ? (ri >> 8) : -2;
      moff[t] = (((ri >> 8) * N + (ri & 255)) * 8 + h) * N;
      const int col = 16 * t + 4 * q;
      const int ci = rowinfo[col < PT_MAXROWS ? col : 0];
      ng[t] = (col < valid) ? (ci >> 8) : -1;
      nn0[t] = ci & 255;
    }
    // extra query rows per graph g of the pack, as B columns 4g + x: x = 0: Wq_g g + bq,
    // 1: step-0 placeholders, 2: load coefficient
    const int xg = j16 >> 2, xx = j16 & 3;
    const bool xon = xx < 3 && xg < G && g0 + xg < p.B;
    float ex[3][4];
    {
      const float *src = (xx == 0) ? p.QG + (size_t)(g0 + (xon ? xg : 0)) * VRP_D
                                   : (xx == 1 ? p.qc0 : p.wload);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xon) v = *reinterpret_cast<const float4 *>(src + h * VRP_HD + 16 * c + 4 * q);
        ex[c][0] = v.x; ex[c][1] = v.y; ex[c][2] = v.z; ex[c][3] = v.w;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    PT_MARK(8 * pk_ + 2);
#pragma unroll
    for (int H = 0; H < 2; ++H) {
      // ---- stage 1: transposed projections ----------------------------------------------
      f32x4 acc[2][3][RT_];
      if constexpr (RING) {
        // key side only (Y = 1: KK / VV of every row tile); the query side is projected tile by
        // tile in stage 2
#pragma unroll
        for (int Y = 1; Y < 2; ++Y)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float4 bb = *reinterpret_cast<const float4 *>(bl + (2 * H + Y) * 48 + 16 * c + 4 * q);
#pragma unroll
            for (int r = 0; r < RT_; ++r) acc[Y][c][r] = f32x4{bb.x, bb.y, bb.z, bb.w};
          }
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
          if (qi + 1 < 4) load_quarter(pack, qi + 1, (qi + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (X3) {
            // quarter qi = k-chunk qi of the weight fragments: split the quarter's rows once, run the
            // three key-side column tiles against them (six MFMAs per product, small terms first)
            bf16x8 ep[RT_][3];
#pragma unroll
            for (int r = 0; r < RT_; ++r) x3_split8(eq[qi & 1][r], ep[r][0], ep[r][1], ep[r][2]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const bf16x8 *wf =
                  reinterpret_cast<const bf16x8 *>(wl3 + (size_t)((2 * H + 1) * 3 + c) * X3_FRAG) + lane;
              const bf16x8 wh = wf[(0 * 4 + qi) * 64], wm = wf[(1 * 4 + qi) * 64], wlo = wf[(2 * 4 + qi) * 64];
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[1][c][r] = X3_MFMA(wm, ep[r][1], acc[1][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[1][c][r] = X3_MFMA(wh, ep[r][2], acc[1][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[1][c][r] = X3_MFMA(wlo, ep[r][0], acc[1][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[1][c][r] = X3_MFMA(wh, ep[r][1], acc[1][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[1][c][r] = X3_MFMA(wm, ep[r][0], acc[1][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[1][c][r] = X3_MFMA(wh, ep[r][0], acc[1][c][r]);
            }
          } else
#pragma unroll
          for (int Y = 1; Y < 2; ++Y)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const float *wrow = wl + ((2 * H + Y) * 48 + 16 * c + j16) * PT_LD + koff;
#pragma unroll
              for (int kl = 0; kl < 2; ++kl) {
                const float4 a = *reinterpret_cast<const float4 *>(wrow + 4 * (2 * qi + kl));
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                  for (int r = 0; r < RT_; ++r)
                    acc[Y][c][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        av[e], eq[qi & 1][r][4 * kl + e], acc[Y][c][r], 0, 0, 0);
              }
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        // quarter 0 for whoever runs stage 1 next: this pack's second half, or the next pack
        if (H == 0) load_quarter(pack, 0, 0);
        else if (pack + stride < p.npacks) load_quarter(pack + stride, 0, 0);
      } else if constexpr (X3) {
#pragma unroll
        for (int Y = 0; Y < 2; ++Y)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float4 bb = *reinterpret_cast<const float4 *>(bl + (2 * H + Y) * 48 + 16 * c + 4 * q);
#pragma unroll
            for (int r = 0; r < RT_; ++r) acc[Y][c][r] = f32x4{bb.x, bb.y, bb.z, bb.w};
          }
        // k-chunk by k-chunk: the rows' 8 values of the chunk are split into planes (RT_ x 3
        // operands alive, not 4 RT_ x 3: the fp32 rows stay for the second half), then all six
        // (projection, column tile) products of the chunk run against them
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x8 ep[RT_][3];
#pragma unroll
          for (int r = 0; r < RT_; ++r) {
            float x8[8] = {ef[r][8 * j], ef[r][8 * j + 1], ef[r][8 * j + 2], ef[r][8 * j + 3],
                           ef[r][8 * j + 4], ef[r][8 * j + 5], ef[r][8 * j + 6], ef[r][8 * j + 7]};
            // (opaque to the optimiser: the second half must split again, not keep the first
            // half's 12 RT_ operand registers alive across stage 2)
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(x8[e]));
            x3_split8(x8, ep[r][0], ep[r][1], ep[r][2]);
          }
#ifdef VRP_PRO_X3_FENCE
          __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
          for (int Y = 0; Y < 2; ++Y)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const bf16x8 *wf =
                  reinterpret_cast<const bf16x8 *>(wl3 + (size_t)((2 * H + Y) * 3 + c) * X3_FRAG) + lane;
              const bf16x8 wh = wf[(0 * 4 + j) * 64], wm = wf[(1 * 4 + j) * 64], wlo = wf[(2 * 4 + j) * 64];
              // weights x rows, small terms first: m m', h l', l h', h m', m h', h h'
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[Y][c][r] = X3_MFMA(wm, ep[r][1], acc[Y][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[Y][c][r] = X3_MFMA(wh, ep[r][2], acc[Y][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[Y][c][r] = X3_MFMA(wlo, ep[r][0], acc[Y][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[Y][c][r] = X3_MFMA(wh, ep[r][1], acc[Y][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[Y][c][r] = X3_MFMA(wm, ep[r][0], acc[Y][c][r]);
#pragma unroll
              for (int r = 0; r < RT_; ++r) acc[Y][c][r] = X3_MFMA(wh, ep[r][0], acc[Y][c][r]);
            }
#ifdef VRP_PRO_X3_FENCE
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      } else
#pragma unroll
      for (int Y = 0; Y < 2; ++Y)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int X = 2 * H + Y;
          // accumulators start at the bias of their projection columns 16c + 4q + reg (bk
          // for the keys, bv for the values, zero otherwise)
          const float4 bb = *reinterpret_cast<const float4 *>(bl + X * 48 + 16 * c + 4 * q);
#pragma unroll
          for (int r = 0; r < RT_; ++r) acc[Y][c][r] = f32x4{bb.x, bb.y, bb.z, bb.w};
          const float *wrow = wl + (X * 48 + 16 * c + j16) * PT_LD + koff;
          __builtin_amdgcn_sched_barrier(0);  // no hoisting of every slice's LDS reads
#pragma unroll
          for (int k4 = 0; k4 < 8; ++k4) {
            const float4 a = *reinterpret_cast<const float4 *>(wrow + 4 * k4);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int r = 0; r < RT_; ++r)
                acc[Y][c][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], ef[r][4 * k4 + e],
                                                                   acc[Y][c][r], 0, 0, 0);
          }
        }
      if (!RING && H == 1) {  // the next pack's rows: requested before the last stage 2
        const int nxt = pack + stride;
        if (nxt < p.npacks) load_pack(nxt);
      }
      if (KEEPK && H == 0) {
        // the keys of this head leave as they sit in the accumulators: lane (node j, q), column
        // tile c = columns 16 c + 4 q .. + 3 of the 48 -> element (4 c + q) of KK4[graph][h]
#pragma unroll
        for (int r = 0; r < RT_; ++r) {
          // (row 16 r + j of the pack = node (row - graph N) of graph mg[r]; 32-bit offsets inside the
          // pack's slice, as for the tables)
          const int off = ((mg[r] * 8 + h) * 12 + q) * N + (16 * r + j16 - mg[r] * N);
          float4 *dst = reinterpret_cast<float4 *>(p.KK4) + (size_t)g0 * 96 * N + off;
#pragma unroll
          for (int c = 0; c < 3; ++c)
            if (mg[r] >= 0)
              dst[4 * c * N] = make_float4(acc[1][c][r][0], acc[1][c][r][1], acc[1][c][r][2], acc[1][c][r][3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);

      PT_MARK(8 * pk_ + (H ? 6 : 3));
      // constant part of the score rows for this lane's four columns of every tile (IRP)
      float bcol[RT_][4];
#pragma unroll
      for (int tn = 0; tn < RT_; ++tn)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) bcol[tn][r4] = 0.f;
      if (H == 0) {
        // ---- extra rows: D[n][i], lane (i = 4g + x, q), register r4 = column n = 16tn + 4q + r4
#pragma unroll
        for (int tn = 0; tn < RT_; ++tn) {
          f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4 += 2) {
              da = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[1][c][tn][r4], ex[c][r4], da, 0, 0, 0);
              db = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[1][c][tn][r4 + 1], ex[c][r4 + 1], db, 0, 0, 0);
            }
          float v[4], c0[4];
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) v[r4] = (da[r4] + db[r4]) * c48;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) c0[r4] = __shfl_down(v[r4], 1, 64);  // the x = 1 row
          if (VEC) {
            const size_t o = ((size_t)(g0 + xg) * 8 + h) * N + nn0[tn];
            if (xon && ng[tn] == xg) {
              const float4 v4 = make_float4(v[0], v[1], v[2], v[3]);
              if (xx == 0) {
                *reinterpret_cast<float4 *>(p.SG + o) = v4;
                *reinterpret_cast<float4 *>(p.row0 + o) =   // step-0 score row
                    make_float4(v[0] + c0[0], v[1] + c0[1], v[2] + c0[2], v[3] + c0[3]);
              }
              if (xx == 1) *reinterpret_cast<float4 *>(p.C0 + o) = v4;
              if (xx == 2) *reinterpret_cast<float4 *>(p.SLD + o) = v4;
            }
            if (fold) {  // the lane that holds these columns of graph gn is (4 gn, q)
              const int srcl = (q << 4) | (4 * (ng[tn] < 0 ? 0 : ng[tn]));
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) bcol[tn][r4] = __shfl(v[r4], srcl, 64);
            }
          } else {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const int col = 16 * tn + 4 * q + r4;
              const int ci = rowinfo[col < PT_MAXROWS ? col : 0];
              const int gn = (col < valid) ? (ci >> 8) : -1;
              const size_t o = ((size_t)(g0 + xg) * 8 + h) * N + (ci & 255);
              if (gn == xg && xon) {
                if (xx == 0) { p.SG[o] = v[r4]; p.row0[o] = v[r4] + c0[r4]; }  // row0: step-0 row
                if (xx == 1) p.C0[o] = v[r4];
                if (xx == 2) p.SLD[o] = v[r4];
              }
              if (fold) bcol[tn][r4] = __shfl(v[r4], (q << 4) | (4 * (gn < 0 ? 0 : gn)), 64);
            }
          }
        }
      }
      PT_MARK(8 * pk_ + (H ? 8 * PTT_SLOTS : 4));
      // ---- stage 2: table tiles straight from the accumulators ---------------------------
      // Addresses: one 64-bit base per pack, 32-bit element offsets inside it (a pack's
      // slice of a table is at most 4 x 80 x 8 x 80 floats).
      float *tab = (H ? p.RT : p.SL) + (size_t)g0 * N * 8 * N;
      const float scale = H ? 1.f : c48;
      // the finished tile whose store is still to be issued (under the next tile's MFMAs)
      f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
      float pc[4] = {0.f, 0.f, 0.f, 0.f};
      int poff = 0, ptm = -2, ptn = -3;   // table offset; row graph, column graph of the tile
      auto flush = [&]() {
        if (VEC) {
          if (ptm == ptn)
            *reinterpret_cast<float4 *>(tab + poff) =
                make_float4(fmaf(pa[0] + pb[0], scale, pc[0]), fmaf(pa[1] + pb[1], scale, pc[1]),
                            fmaf(pa[2] + pb[2], scale, pc[2]), fmaf(pa[3] + pb[3], scale, pc[3]));
        }
      };
      // RING: the rows of query tile tm (whole inner dimension, 32 registers), tile tm + 1
      // requested ahead of tile tm's 96 + 12 RT_ MFMAs
      float et[RING ? 2 : 1][32];
      auto load_tile = [&](int tm, int buf) {
        const size_t R0 = (size_t)pack * G * N;
        const int row = 16 * tm + j16;
        const float4 *src = reinterpret_cast<const float4 *>(
            p.emb + (R0 + (row < valid ? row : 0)) * VRP_EMB + koff);
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (row < valid) v = src[k4];
          float *d = et[RING ? buf : 0] + 4 * k4;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      };
      if constexpr (RING) load_tile(0, 0);
#pragma unroll
      for (int tm = 0; tm < RT_; ++tm) {
        if constexpr (RING) {
          if (tm + 1 < RT_) load_tile(tm + 1, (tm + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
          // query-side projection of this tile (QL / KM, X = 2 H), same k order as stage 1
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float4 bb = *reinterpret_cast<const float4 *>(bl + (2 * H) * 48 + 16 * c + 4 * q);
            acc[0][c][0] = f32x4{bb.x, bb.y, bb.z, bb.w};
          }
          if constexpr (X3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float x8[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) x8[e] = et[tm & 1][8 * j + e];
              bf16x8 eh, em, el;
              x3_split8(x8, eh, em, el);
              bf16x8 wh[3], wm[3], wlo[3];
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const bf16x8 *wf =
                    reinterpret_cast<const bf16x8 *>(wl3 + (size_t)((2 * H) * 3 + c) * X3_FRAG) + lane;
                wh[c] = wf[(0 * 4 + j) * 64]; wm[c] = wf[(1 * 4 + j) * 64]; wlo[c] = wf[(2 * 4 + j) * 64];
              }
              // the three column tiles take turns: consecutive MFMAs never share an accumulator
#pragma unroll
              for (int c = 0; c < 3; ++c) acc[0][c][0] = X3_MFMA(wm[c], em, acc[0][c][0]);
#pragma unroll
              for (int c = 0; c < 3; ++c) acc[0][c][0] = X3_MFMA(wh[c], el, acc[0][c][0]);
#pragma unroll
              for (int c = 0; c < 3; ++c) acc[0][c][0] = X3_MFMA(wlo[c], eh, acc[0][c][0]);
#pragma unroll
              for (int c = 0; c < 3; ++c) acc[0][c][0] = X3_MFMA(wh[c], em, acc[0][c][0]);
#pragma unroll
              for (int c = 0; c < 3; ++c) acc[0][c][0] = X3_MFMA(wm[c], eh, acc[0][c][0]);
#pragma unroll
              for (int c = 0; c < 3; ++c) acc[0][c][0] = X3_MFMA(wh[c], eh, acc[0][c][0]);
            }
          } else {
          // the three column tiles take turns: consecutive MFMAs never share an accumulator
          const float *wrow = wl + ((2 * H) * 48 + j16) * PT_LD + koff;
          float4 a[3], an[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) a[c] = *reinterpret_cast<const float4 *>(wrow + 16 * c * PT_LD);
#pragma unroll
          for (int k4 = 0; k4 < 8; ++k4) {
            if (k4 + 1 < 8) {
#pragma unroll
              for (int c = 0; c < 3; ++c)
                an[c] = *reinterpret_cast<const float4 *>(wrow + 16 * c * PT_LD + 4 * (k4 + 1));
            }
            const float av[3][4] = {{a[0].x, a[0].y, a[0].z, a[0].w}, {a[1].x, a[1].y, a[1].z, a[1].w},
                                    {a[2].x, a[2].y, a[2].z, a[2].w}};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int c = 0; c < 3; ++c)
                acc[0][c][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][e], et[tm & 1][4 * k4 + e],
                                                                   acc[0][c][0], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 3; ++c) a[c] = an[c];
          }
          }
        }
#pragma unroll
        for (int tn = 0; tn < RT_; ++tn) {
          if (!((needmask >> (tm * RT_ + tn)) & 1ull)) continue;
          f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4 += 2) {
              da = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[1][c][tn][r4], acc[0][c][RING ? 0 : tm][r4], da, 0, 0, 0);
              db = __builtin_amdgcn_mfma_f32_16x16x4f32(acc[1][c][tn][r4 + 1], acc[0][c][RING ? 0 : tm][r4 + 1], db, 0, 0, 0);
            }
          if (VEC) {
            flush();  // previous tile
            pa = da; pb = db; poff = moff[tm] + nn0[tn]; ptm = mg[tm]; ptn = ng[tn];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) pc[r4] = bcol[tn][r4];
          } else {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const int col = 16 * tn + 4 * q + r4;
              const int ci = rowinfo[col < PT_MAXROWS ? col : 0];
              if (col < valid && (ci >> 8) == mg[tm])
                tab[moff[tm] + (ci & 255)] = fmaf(da[r4] + db[r4], scale, bcol[tn][r4]);
            }
          }
        }
      }
      flush();
      __builtin_amdgcn_sched_barrier(0);
      PT_MARK(8 * pk_ + (H ? 7 : 5));
    }
  }
}

// VRP_PROLOGUE_FP32=1: stage 1 on the fp32 MFMA at every size (A/B aid)
static bool prologue_x3_enabled() {
  static const bool off = getenv("VRP_PROLOGUE_FP32") != nullptr;
  return !off;
}
template <int RT_, bool VEC, bool X3, bool KEEPK>
static int launch_prologue_tables_as(const PrologueParams &p, hipStream_t st) {
  const size_t lds = sizeof(float) * ((X3 ? 12 * X3_FRAG / 2 : 4 * 48 * PT_LD) + 4 * 48) + sizeof(int) * PT_MAXROWS;
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&prologue_tables_kernel<RT_, VEC, (RT_ > 5), X3, KEEPK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("prologue_tables: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  // one workgroup per CU (100 - 150 KB of LDS each): 8 heads x nsub x 8 XCD slots; fewer when the
  // batch has fewer packs than wave slots
  constexpr int NWV = (X3 && RT_ <= 5) ? PT_X3_WAVES : 4;
  int nsub = 4;
  while (nsub > 1 && 8 * (nsub / 2) * NWV >= p.npacks) nsub /= 2;
  hipLaunchKernelGGL((prologue_tables_kernel<RT_, VEC, (RT_ > 5), X3, KEEPK>), dim3(64 * nsub), dim3(64 * NWV), lds, st, p);
  VRP_CHECK_LAUNCH("prologue_tables");
#ifdef VRP_PRO_TRACE
  {
    static int calls = 0;
    if (++calls == 6) {
      (void)hipDeviceSynchronize();
      static unsigned long long h[8 * PTT_SLOTS];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pro_trace), sizeof(h));
      for (int wv = 0; wv < NWV; ++wv) {
        const unsigned long long *t = h + wv * PTT_SLOTS;
        fprintf(stderr, "[prologue trace] wave %d:", wv);
        for (int pk = 0; pk < 6; ++pk) {
          const unsigned long long *u = t + 8 * pk;
          fprintf(stderr, " | top+%llu setup %llu s1a %llu extra %llu s2a %llu s1b %llu s2b %llu = %llu",
                  u[1] - t[0], u[2] - u[1], u[3] - u[2], u[4] - u[3], u[5] - u[4], u[6] - u[5], u[7] - u[6],
                  u[7] - u[1]);
        }
        fprintf(stderr, "\n");
      }
    }
  }
#endif
  return 0;
}
template <int RT_, bool VEC>
static int launch_prologue_tables(const PrologueParams &p, hipStream_t st) {
  if constexpr (RT_ <= 5) {   // (keys are kept for N <= 63 only: never a RING instance)
    if (p.KK4) {
      if (prologue_x3_enabled() && p.WprojX3) return launch_prologue_tables_as<RT_, VEC, true, true>(p, st);
      return launch_prologue_tables_as<RT_, VEC, false, true>(p, st);
    }
  }
  if (prologue_x3_enabled() && p.WprojX3) return launch_prologue_tables_as<RT_, VEC, true, false>(p, st);
  return launch_prologue_tables_as<RT_, VEC, false, false>(p, st);
}

template <bool VEC>
static int launch_prologue_vec(const PrologueParams &p, hipStream_t st) {
  switch ((p.G * p.N + 15) / 16) {
    case 1: return launch_prologue_tables<1, VEC>(p, st);
    case 2: return launch_prologue_tables<2, VEC>(p, st);
    case 3: return launch_prologue_tables<3, VEC>(p, st);
    case 4: return launch_prologue_tables<4, VEC>(p, st);
    case 5: return launch_prologue_tables<5, VEC>(p, st);
    case 6: return launch_prologue_tables<6, VEC>(p, st);
    default: return launch_prologue_tables<7, VEC>(p, st);
  }
}

static PrologueParams prologue_params(int kind, int B, int N, const float *emb, const Derived &d,
                                      const DecWs &w) {
  PrologueParams p;
  p.kind = kind; p.B = B; p.N = N;
  int G = fused_max_rows(N) / N;
  if (G > 4) G = 4;
  // The x3 stage 1 holds the rows' fp32 values AND a chunk's planes.  Eight waves per workgroup
  // leave a wave 256 registers: three row tiles fit with 23 of them spilled at pack boundaries,
  // FOUR spill 250 and FIVE 543 inside the products, and their reloads drain the table stores.
  // Measured (round 6, tools/prologue_sizes.py, profiles/r06_prologue_sizes.txt; B = 8192):
  //   N = 20 / 30 / 32 in packs of <= 3 tiles: 576 / 719 / 634 us, in packs of <= 4: 671 / 1225 / 922;
  //   ONE graph of four tiles (N = 50 / 64): bf16 planes 2620 / 2054 us, fp32 MFMA 2679 / 2306;
  //   ONE graph of five tiles (N = 72 / 80): bf16 planes 4136 / 4544 us, fp32 MFMA 3017 / 3058.
  // So: packs of at most three tiles; a single graph may take four; five-tile graphs stay on the
  // fp32 MFMA instances (p.WprojX3 = nullptr below).  Among the admissible pack sizes the one with
  // the fewest (rounds of the 2048 wave slots) x (row tiles per unit) -- stage 1 is the bulk of a
  // unit: TSP-20 x 512: G = 2 (2048 units, one per wave).
  static const int max_tiles = getenv("VRP_PRO_X3_TILES") ? atoi(getenv("VRP_PRO_X3_TILES")) : 3;
  if (G > B) G = B;
  bool x3 = prologue_x3_enabled() && fused_max_rows(N) <= 80;
  if (x3 && (N + 15) / 16 >= 5) x3 = false;            // one graph = five tiles: fp32 is faster
  if (x3) {
    int best = 0;
    long best_cost = 0;
    for (int g = G; g >= 1; --g) {
      const int tiles = (g * N + 15) / 16;
      if (tiles > max_tiles && g > 1) continue;
      const long slots = 256 * PT_X3_WAVES;   // waves of one launch
      const long units = (long)((B + g - 1) / g) * 8, cost = ((units + slots - 1) / slots) * tiles;
      if (!best || cost < best_cost) { best = g; best_cost = cost; }
    }
    G = best;
  }
  p.G = G;
  p.npacks = (B + G - 1) / G;
  p.emb = emb; p.Wproj = d.Wproj; p.bproj = d.bproj; p.QG = w.QG; p.qc0 = d.qc0; p.wload = d.wload;
  // (one graph of six / seven tiles, 80 < N <= 112: the RING instances, on bf16 planes since round 6;
  // VRP_PROLOGUE_RING_FP32=1 keeps them on the fp32 MFMA: A/B aid)
  static const bool ring_fp32 = getenv("VRP_PROLOGUE_RING_FP32") != nullptr;
  const bool ring_x3 = prologue_x3_enabled() && fused_max_rows(N) > 80 && !ring_fp32;
  p.WprojX3 = (x3 || ring_x3) ? reinterpret_cast<const __bf16 *>(d.WprojX3) : nullptr;
  p.SG = w.SG; p.C0 = w.C0; p.SLD = w.SLD; p.row0 = w.row0; p.SL = w.SL; p.RT = w.RT;
  p.KK4 = (kind != VRP_KIND_IRP) ? w.KK4 : nullptr;
  return p;
}

// ------------------------------------------------------------------ unfused path (N > 80)
// Per-(graph, head) products of two (N x 48) row blocks of PROJ on v_mfma_f32_16x16x4_f32,
// one wave per (graph, head).  The 48-long inner dimension is split over the four 16-lane
// groups: group q holds k in [12q, 12q+12) of its row (three float4 loads), MFMA step s
// consumes element s of every group -- a fixed permutation of k applied to both operands.
template <int NTMAX>
__device__ __forceinline__ void load_rows12(float (&dst)[12], const float *row_ptr, bool on) {
  if (on) {
#pragma unroll
    for (int d = 0; d < 12; d += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(row_ptr + d);
      dst[d] = t.x; dst[d + 1] = t.y; dst[d + 2] = t.z; dst[d + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int d = 0; d < 12; ++d) dst[d] = 0.f;
  }
}

template <int NTMAX>  // 16-column tiles per row: ceil(N/16) <= NTMAX
__global__ __launch_bounds__(256) void pair_tables_kernel(
    int kind, int N, int P, const float *__restrict__ PROJ, const float *__restrict__ QG,
    const float *__restrict__ qc0, const float *__restrict__ wload, float *__restrict__ SG,
    float *__restrict__ C0, float *__restrict__ SLD, float *__restrict__ SL,
    float *__restrict__ row0, float *__restrict__ RT) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x;
  const int h = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + (threadIdx.x >> 6));
  const int i16 = lane & 15, q = lane >> 4;
  const int NT = (N + 15) >> 4;
  const int qloff = 0, kkoff = 384, kmoff = 768, vvoff = 1152;
  const int hq = h * VRP_HD + 12 * q;
  const float c = 0.14433756729740643f;  // 1/sqrt(48)
  const float *rows = PROJ + (size_t)b * N * P;
  const size_t hn = ((size_t)b * 8 + h) * N;

  float bf[NTMAX][12];
  auto load_b = [&](int off) {
#pragma unroll
    for (int nt = 0; nt < NTMAX; ++nt) {
      const int n = nt * 16 + i16;
      load_rows12<NTMAX>(bf[nt], rows + (size_t)(n < N ? n : 0) * P + off + hq, nt < NT && n < N);
    }
  };
  // one 16-row tile of A against every column tile; store(row, column tile, column, value)
  auto tile_rows = [&](const float (&af)[12], auto &&store) {
#pragma unroll
    for (int nt = 0; nt < NTMAX; ++nt) {
      if (nt < NT) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < 12; ++s_)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s_], bf[nt][s_], acc, 0, 0, 0);
        const int n = nt * 16 + i16;  // D: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
        for (int r = 0; r < 4; ++r) store(q * 4 + r, nt, n, acc[r]);
      }
    }
  };

  // ---- glimpse score tables: B operand = projected keys -----------------------------
  load_b(kkoff);
  float cbase[NTMAX];  // IRP: the constant row SG, folded into every row of SL
#pragma unroll
  for (int nt = 0; nt < NTMAX; ++nt) cbase[nt] = 0.f;
  {
    float af[12];
    // three extra query rows: 0 = graph embedding + bq, 1 = step-0 placeholders, 2 = load
    const float *src = (i16 == 0) ? QG + (size_t)b * VRP_D : (i16 == 1 ? qc0 : wload);
    load_rows12<NTMAX>(af, src + hq, i16 < 3);
    float keep[NTMAX];  // row 0 of the tile lives in lanes q == 0, reg 0
#pragma unroll
    for (int nt = 0; nt < NTMAX; ++nt) keep[nt] = 0.f;
    tile_rows(af, [&](int r, int nt, int n, float v) {
      v *= c;
      if (r == 0) keep[nt] = v;
      if (n < N) {
        if (r == 0) SG[hn + n] = v;
        if (r == 1) { C0[hn + n] = v; row0[hn + n] = keep[nt] + v; }  // step-0 score row
        if (r == 2) SLD[hn + n] = v;
      }
    });
    if (kind == VRP_KIND_IRP) {  // no first-node term (graph_decoder.py:90-91)
#pragma unroll
      for (int nt = 0; nt < NTMAX; ++nt)
        if (nt < NT) cbase[nt] = __shfl(keep[nt], i16, 64);  // lane i16 (q == 0) holds column i16
    }
  }
  for (int mt = 0; mt < NT; ++mt) {
    const int m = mt * 16 + i16;
    float af[12];
    load_rows12<NTMAX>(af, rows + (size_t)(m < N ? m : 0) * P + qloff + hq, m < N);
    tile_rows(af, [&](int r, int nt, int n, float v) {
      const int mm = mt * 16 + r;
      if (mm < N && n < N) SL[(((size_t)b * N + mm) * 8 + h) * N + n] = fmaf(v, c, cbase[nt]);
    });
  }
  // ---- pointer-logit table: B operand = projected values ------------------------------
  load_b(vvoff);
  for (int mt = 0; mt < NT; ++mt) {
    const int m = mt * 16 + i16;
    float af[12];
    load_rows12<NTMAX>(af, rows + (size_t)(m < N ? m : 0) * P + kmoff + hq, m < N);
    tile_rows(af, [&](int r, int nt, int n, float v) {
      const int mm = mt * 16 + r;
      if (mm < N && n < N) RT[(((size_t)b * N + mm) * 8 + h) * N + n] = v;
    });
  }
}

// ------------------------------------------------------------------ score base after step 0
// TSP/VRP: base[b][h][n] = SG[b][h][n] + FK[b][h][:] . e_n, FK = AfT e_first (the first chosen
// node's query part folded through the keys).  One wave per graph: D (16 x 16 per node tile)
// = FK rows (8 heads, zero-padded to 16) x E^T on v_mfma_f32_16x16x4_f32, k split as in the
// prologue kernel.  Also hands step 1 its complete row (last = first) for the latency-mode
// step kernel: curs = SL[b][first] + base.
template <int NTMAX>
__global__ __launch_bounds__(256) void score_base_kernel(int B, int N, const float *__restrict__ emb,
                                                         const float *__restrict__ FK,
                                                         const float *__restrict__ SG,
                                                         const float *__restrict__ SL,
                                                         const int32_t *__restrict__ first,
                                                         float *__restrict__ base,
                                                         float *__restrict__ curs) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int j16 = lane & 15, q = lane >> 4;
  const int koff = 64 * (q & 1) + 32 * (q >> 1);
  const int NT = (N + 15) >> 4;
  float af[32];  // A row j16 = head (rows 8..15 are zero)
  {
    const float4 *src = reinterpret_cast<const float4 *>(FK + (size_t)b * 1024 + (j16 & 7) * 128 + koff);
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j16 < 8) v = src[k4];
      af[4 * k4] = v.x; af[4 * k4 + 1] = v.y; af[4 * k4 + 2] = v.z; af[4 * k4 + 3] = v.w;
    }
  }
  const int fb = first[b];
  // every load of the graph (embedding rows as B fragments, SG, the first node's SL row) is
  // issued before the first MFMA: one wave per graph has nothing else to hide a round trip
  float4 ev[NTMAX][8];
  float sgv[NTMAX][4], slv[NTMAX][4];
#pragma unroll
  for (int nt = 0; nt < NTMAX; ++nt) {
    const int n = 16 * nt + j16;
    const bool on = nt < NT && n < N;
    const float4 *src = reinterpret_cast<const float4 *>(
        emb + ((size_t)b * N + (on ? n : 0)) * VRP_EMB + koff);
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) ev[nt][k4] = on ? src[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int h = 4 * q + r4;
      const bool ok = on && q < 2;
      const size_t o = ((size_t)b * 8 + (ok ? h : 0)) * N + (ok ? n : 0);
      sgv[nt][r4] = ok ? SG[o] : 0.f;
      slv[nt][r4] = ok ? SL[(((size_t)b * N + fb) * 8 + h) * N + n] : 0.f;
    }
  }
#pragma unroll
  for (int nt = 0; nt < NTMAX; ++nt) {
    if (nt >= NT) break;
    const int n = 16 * nt + j16;
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};  // two chains: k4 even / odd
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      const float4 v = ev[nt][k4];
      f32x4 &d = (k4 & 1) ? d1 : d0;
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * k4], v.x, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * k4 + 1], v.y, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * k4 + 2], v.z, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * k4 + 3], v.w, d, 0, 0, 0);
    }
    // D: lane (node j, q), register r4 = head 4q + r4 (q < 2)
    if (q < 2 && n < N) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int h = 4 * q + r4;
        const size_t o = ((size_t)b * 8 + h) * N + n;
        const float v = sgv[nt][r4] + (d0[r4] + d1[r4]);
        base[o] = v;
        curs[o] = slv[nt][r4] + v;
      }
    }
  }
}

__global__ __launch_bounds__(128) void gather_first_kernel(const float *__restrict__ emb,
                                                           const int32_t *__restrict__ first,
                                                           int N, float *__restrict__ out) {
  const int b = blockIdx.x, c = threadIdx.x;
  out[(size_t)b * VRP_EMB + c] = emb[((size_t)b * N + first[b]) * VRP_EMB + c];
}

extern "C" int64_t vrp_decoder_workspace_bytes(int kind, int B, int N) {
  (void)kind;
  return decws_bytes(B, N);
}

// The embeddings once more with the rows paired (DecWs::embP): float4 number (pair i, lane l) =
// {e[2i][2l], e[2i][2l+1], e[2i+1][2l], e[2i+1][2l+1]}, zeros for a row beyond N.  The raw-tile
// step kernel keeps lane l's two columns of every row in registers; from this copy it fills two
// rows with ONE 16-byte load per lane (8-byte loads reach 0.5-0.7 of the 16-byte rate).  Written
// once per episode, read once per tile step.
__global__ __launch_bounds__(256) void pair_rows_kernel(const float *__restrict__ emb, int N,
                                                        float *__restrict__ embP, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // float4 index
  if (i >= total) return;
  const int NP = (N + 1) / 2;
  const int l = (int)(i & 63);
  const size_t pr = i >> 6;                 // b * NP + pair
  const size_t b = pr / NP;
  const int pi = (int)(pr - b * NP);
  const float2 *r0 = reinterpret_cast<const float2 *>(emb + (b * N + 2 * pi) * VRP_EMB) + l;
  const float2 lo = *r0;
  const float2 hi = (2 * pi + 1 < N) ? r0[64] : make_float2(0.f, 0.f);
  reinterpret_cast<float4 *>(embP)[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
}

int vrp_decode_prologue_ex(int kind, const void *derived, int B, int N, const float *emb,
                           void *workspace, int constants_done, void *stream);
extern "C" int vrp_decode_prologue(int kind, const void *derived, int B, int N, const float *emb,
                                   void *workspace, void *stream) {
  return vrp_decode_prologue_ex(kind, derived, B, N, emb, workspace, 0, stream);
}

// constants_done: bit 0: the graph mean, cvec and the cleared hand-off words were already produced
// by the encoder's stack kernel (vrp_rollout, small batches); bit 1: QG as well
int vrp_decode_prologue_ex(int kind, const void *derived, int B, int N, const float *emb,
                           void *workspace, int constants_done, void *stream) {
  VRP_REQUIRE(derived && emb && workspace, "decode_prologue: NULL argument");
  VRP_REQUIRE(B > 0 && N > 0 && N <= VRP_MAX_NODES, "decode_prologue: bad shape B=%d N=%d", B, N);
  VRP_REQUIRE(use_rtable(N), "decode_prologue: N=%d above the table limit %d", N, VRP_RT_MAX_N);
  hipStream_t st = (hipStream_t)stream;
  Derived d = carve_derived(const_cast<void *>(derived));
  DecWs w = carve_decws(workspace, B, N);
  if (!(constants_done & 1)) {
    hipLaunchKernelGGL(graph_mean_cvec_kernel, dim3(B), dim3(256), 0, st, emb, d.mb, N, w.g, w.cvec,
                       w.hist, w.err);
    VRP_CHECK_LAUNCH("graph_mean_cvec");
  }
  if (!(constants_done & 2))   // (the x3 stack kernel's epilogue leaves QG too)
    if (int r = vrp_launch_gemm_nt(w.g, 128, d.Wqg, 128, d.bq, nullptr, 0, w.QG, 384, B, 384, 128, 0,
                                   st)) return r;
  if (use_fused_prologue(N)) {
    const PrologueParams p = prologue_params(kind, B, N, emb, d, w);
    if (int r = (N & 3) == 0 ? launch_prologue_vec<true>(p, st) : launch_prologue_vec<false>(p, st))
      return r;
  } else {
    const int P = proj_width(N);
    float *PROJ = w.PROJ;
    if (int r = vrp_launch_gemm_nt(emb, 128, d.Wproj, 128, d.bproj, nullptr, 0, PROJ, P, B * N, P,
                                   128, 0, st)) return r;
    hipLaunchKernelGGL((pair_tables_kernel<8>), dim3(B, 2), dim3(256), 0, st, kind, N, P, PROJ, w.QG,
                       d.qc0, d.wload, w.SG, w.C0, w.SLD, w.SL, w.row0, w.RT);
    VRP_CHECK_LAUNCH("pair_tables");
  }
  // last, so that step 0 finds the copy it reads in the caches (the tables above stream a
  // gigabyte through them)
  if (kind != VRP_KIND_IRP && tile_pairs_shape(B, N)) {
    const size_t total = (size_t)B * ((N + 1) / 2) * 64;
    hipLaunchKernelGGL(pair_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, emb,
                       N, w.embP, total);
    VRP_CHECK_LAUNCH("pair_rows");
  }
  return 0;
}

// first_ is known after step 0 (graph_decoder.py:111-113): its query part, folded through the
// keys, completes the constant part of every later score row
extern "C" int vrp_decode_first_row(int kind, const void *derived, int B, int N, const float *emb,
                                    void *workspace, void *stream) {
  VRP_REQUIRE(derived && emb && workspace, "decode_first_row: NULL argument");
  if (kind == VRP_KIND_IRP) return 0;  // no first-node term: the prologue folded SG into SL
  hipStream_t st = (hipStream_t)stream;
  // small batches: the prologue kept the keys, one workgroup per graph does it in one launch
  // (the arithmetic the wider persistent kernels run ahead of their first step)
  if (kk_floats(B, N) > 0) return vrp_launch_first_base(kind, derived, B, N, emb, workspace, st);
  Derived d = carve_derived(const_cast<void *>(derived));
  DecWs ws = carve_decws(workspace, B, N);
  // base = SG + FK . e_n with FK = e_first AfT^T, read by the steps as a second row
  const int gr = vrp_launch_gemm_gather_k128(emb, 128, ws.first, N, d.AfT, 128, ws.FK, 1024, B, 1024, st);
  if (gr > 0) return gr;
  if (gr < 0) {
    hipLaunchKernelGGL(gather_first_kernel, dim3(B), dim3(128), 0, st, emb, ws.first, N, ws.Efirst);
    VRP_CHECK_LAUNCH("gather_first");
    if (int r = vrp_launch_gemm_nt(ws.Efirst, 128, d.AfT, 128, nullptr, nullptr, 0, ws.FK, 1024, B,
                                   1024, 128, 0, st)) return r;
  }
  if (N <= 32)
    hipLaunchKernelGGL((score_base_kernel<2>), dim3((B + 3) / 4), dim3(256), 0, st, B, N, emb, ws.FK,
                       ws.SG, ws.SL, ws.first, ws.base, ws.curs);
  else if (N <= 64)
    hipLaunchKernelGGL((score_base_kernel<4>), dim3((B + 3) / 4), dim3(256), 0, st, B, N, emb, ws.FK,
                       ws.SG, ws.SL, ws.first, ws.base, ws.curs);
  else
    hipLaunchKernelGGL((score_base_kernel<8>), dim3((B + 3) / 4), dim3(256), 0, st, B, N, emb, ws.FK,
                       ws.SG, ws.SL, ws.first, ws.base, ws.curs);
  VRP_CHECK_LAUNCH("score_base");
  return 0;
}

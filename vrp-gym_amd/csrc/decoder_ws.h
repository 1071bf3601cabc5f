// Layout of the decoder's two device buffers, shared by decoder.hip (weight folds, step
// kernels) and decoder_prologue.hip (per-episode tables):
//   Derived  - folded decoder matrices, rebuilt when the parameters change (vrp_decoder_prepare)
//   DecWs    - per-episode workspace (vrp_decode_prologue / vrp_decode_first_row / steps)
#pragma once
#include <stdlib.h>
#include "common.h"

#define VRP_WPROJ_X3_FLOATS (8 * 12 * 6144 / 2)   // 96 fragments of 6144 bf16
struct Derived {
  float *Wproj;  // (1536,128) = [Wq_last | Wk | M^T | Wv]
  float *bproj;  // (1536)       0 | bk | 0 | bv
  float *Wqf;    // (384,128)  first-node block of the query projection (TSP/VRP)
  float *Wqg;    // (384,128)  graph-embedding block of the query projection
  float *bq;     // (384)
  float *qc0;    // (384)      query contribution of the step-0 placeholders
  float *wload;  // (384)      query coefficient of the vehicle load (IRP)
  float *WvT;    // (128,384)  v_proj_weight transposed
  float *bv;     // (384)
  float *MT;     // (384,128)  transpose of M = Wkp^T Watt Wo / sqrt(128)
  float *mb;     // (128)      Wkp^T Watt bo / sqrt(128)
  float *tmpA;   // (128,384)  Watt Wo
  float *tmpv;   // (128)      Watt bo
  float *M;      // (128,384)  M itself (row-major: the tile kernel's B operand)
  float *AfT;    // (1024,128) row h*128+k: sum_d Wk[48h+d][k] Wq_first[48h+d][:] / sqrt(48): the
                 //            first-node query folded through the keys (TSP/VRP)
  float *WvP;    // (384*128)  v_proj rows in MFMA fragment order: [head][k4][col tile][lane][4],
                 //            lane (i16, q) = Wv[48 head + 16 tile + i16][16 k4 + 4 q ..+3]
  float *MP;     // (128*384)  M in fragment order: [col tile][k4][lane][4] = M[16 tile + i16][16 k4 + 4 q ..+3]
  float *WprojX3;  // Wproj once more for the fused prologue's projections on the bf16 matrix cores
                   // (x3_common.h): [head 8][X 4][column tile 3] fragments of X3_FRAG bf16 = three
                   // planes x four k-chunks x 64 lanes x 8; lane (j16, q), chunk j holds row
                   // X*384 + 48 head + 16 tile + j16, k = 64 (q & 1) + 32 (q >> 1) + 8 j .. + 7
  float *WqgT;     // (128,384)  Wqg transposed: the stack kernel's epilogue computes QG = Wq_g g + bq
                   // for its own graphs, lane = output column (coalesced), instead of a GEMM launch
  float *WqfT;     // (128,384)  Wqf transposed: the persistent kernels project the first chosen node
                   // themselves (decoder_persistent.hip: persist_first_base)
};

static inline Derived carve_derived(void *base) {
  float *p = (float *)base;
  Derived d;
  d.Wproj = p; p += 1536 * 128;
  d.bproj = p; p += 1536;
  d.Wqf = p;   p += 384 * 128;
  d.Wqg = p;   p += 384 * 128;
  d.bq = p;    p += 384;
  d.qc0 = p;   p += 384;
  d.wload = p; p += 384;
  d.WvT = p;   p += 128 * 384;
  d.bv = p;    p += 384;
  d.MT = p;    p += 384 * 128;
  d.mb = p;    p += 128;
  d.tmpA = p;  p += 128 * 384;
  d.tmpv = p;  p += 128;
  d.M = p;     p += 128 * 384;
  d.AfT = p;   p += 1024 * 128;
  d.WvP = p;   p += 384 * 128;
  d.MP = p;    p += 128 * 384;
  d.WprojX3 = p; p += VRP_WPROJ_X3_FLOATS;
  d.WqgT = p;  p += 128 * 384;
  d.WqfT = p;  p += 128 * 384;
  return d;
}

static inline int64_t derived_floats() {
  return 1536 * 128 + 1536 + 384 * 128 + 384 * 128 + 384 * 3 + 128 * 384 + 384 + 384 * 128 + 128 +
         128 * 384 + 128 + 128 * 384 + 1024 * 128 + 384 * 128 + 128 * 384 + VRP_WPROJ_X3_FLOATS +
         128 * 384 + 128 * 384;
}

// ------------------------------------------------------------------ per-episode workspace
// Glimpse score of step t > 0 (graph b, head h, node n) = SL[b][last][h][n] + base[b][h][n]
// (TSP/VRP; IRP: the constant part is folded into SL, + load * SLD[b][h][n]); step 0 reads
// row0.
struct DecWs {
  float *g;      // (B,128)     graph embedding            graph_decoder.py:75-77
  float *QG;     // (B,384)     Wq_g g + bq
  float *PROJ;   // (B*N,1536)  [QL | KK | KM | VV] rows; only the unfused path (N > 80)
  float *SG, *C0, *SLD, *row0, *curs;  // (B,8,N) each; curs = next step's row (latency mode)
  float *base;                   // (B,8,N)  constant part of every score row after step 0
  float *SL;                     // (B,N,8,N)  last-node part of the score: QL_m . KK_n / sqrt(48)
  float *Efirst;                 // (B,128)  first chosen node (general-GEMM fallback)
  float *FK;                     // (B,1024) first-node query folded through the keys (N > 80)

  float *embP;                   // (B, ceil(N/2), 64, 4)  the embeddings once more, rows paired: lane l of
                                 // pair i holds {e[2i][2l], e[2i][2l+1], e[2i+1][2l], e[2i+1][2l+1]} -- the
                                 // raw-tile kernel's register layout as ONE 16-byte load per lane and
                                 // row pair (only for the shapes that kernel serves by default)
  float *RT;                     // (B,N,8,N)  pointer-logit table, row m = RT[b][m][:][:]
  float *KK4;                    // (B,8,12,N,4)  the glimpse keys Wk e_n + bk of head h, four of the 48
                                 // columns per element: written by the fused prologue for the shapes the
                                 // persistent kernels serve (kk_floats), read once, after step 0
  float *cvec;                   // (B,N)                e_m . mb
  int32_t *last, *first;         // (B)
  // persistent multi-step kernel (decoder_persistent.hip)
  unsigned long long *hist;      // (hist_rows(N), B) published mask words, one row per step
  int32_t *ta;                   // (B) step at which a graph's visited row became all ones
  float *ret;                    // (B) reward of the forced way back after ta
  int32_t *wb_cur;               // (B) location before the way back
  double *wb_load;               // (B) load before the way back
  int32_t *err;                  // raised by the first wave that gives up waiting
  // the state a persistent launch started from (its in-kernel fallback restarts there)
  uint8_t *sv_visited;           // (B,N)
  int32_t *sv_cur, *sv_last;     // (B)
  double *sv_load;               // (B)
  float *sv_accl, *sv_accp;      // (B)
};

#define VRP_RT_MAX_N 128    // above this the tile kernel (one raw-tile read per step) is used
// the fused projection+table prologue packs <= 80 rows (five 16-row tiles) per wave when
// N % 4 == 0 (16-byte table stores), <= 64 rows otherwise
#define VRP_FUSED_MAX_N 80
// rows of the hand-off words (one row = one 8-byte word per graph): a step's mask is one word
// up to 63 nodes (decoder_persistent.hip), for up to 2N steps (max_steps + 1 <= 2N); larger
// graphs never take the persistent path and have none
__host__ __device__ static inline int hist_rows(int N) { return N > 63 ? 0 : 2 * N; }
// bytes of the saved-state rows (persistent path only, N <= 63)
static inline size_t persist_save_bytes(int B, int N) {
  return N > 63 ? 0 : vrp_align_up((size_t)B * N) + 4 * vrp_align_up((size_t)B * 4) +
                          vrp_align_up((size_t)B * 8);
}
// shapes on which the default dispatch sends steps to the raw-tile kernel (decoder.hip,
// hybrid_shape; IRP excepted there): they get the row-paired copy of the embeddings
// Round 4: the second-generation tile kernel (decoder_tile2.hip, N <= 100) reads `emb` itself in
// the matrix cores' operand order and needs no copy; only the first-generation kernel does
// (100 < N <= 104, or everywhere with the A/B aid VRP_TILE_V1=1).
static inline bool tile_v1_forced() {
  static const bool v = getenv("VRP_TILE_V1") != nullptr;
  return v;
}
static inline bool tile_pairs_shape(int B, int N) {
  if (!tile_v1_forced()) return N > 100 && N <= 104;
  return (N > 64 && N <= 104) || (B > 2048 && N > 32 && N <= 40);
}
static inline size_t pairs_floats(int B, int N) {
  return tile_pairs_shape(B, N) ? (size_t)B * ((N + 1) / 2) * 256 : 0;
}
// rows of a (pack of graphs, head) unit of the fused prologue kernel: 80 (five 16-row tiles, packs
// of up to four graphs), 64 when N is not a multiple of four (element-wise table stores), and
// ONE graph of up to 112 rows (seven tiles) for 80 < N <= 112
static inline int fused_max_rows(int N) { return (N & 3) == 0 ? (N > 80 ? 112 : 80) : 64; }
static inline bool use_rtable(int N) { return N <= VRP_RT_MAX_N; }
// A/B aid: VRP_PROLOGUE_UNFUSED=1 forces the projection GEMM + pair_tables path at every N
static inline bool prologue_unfused() {
  static const bool v = getenv("VRP_PROLOGUE_UNFUSED") != nullptr;
  return v;
}
static inline bool use_fused_prologue(int N) { return N <= fused_max_rows(N) && !prologue_unfused(); }
static inline int proj_width(int N) { return use_rtable(N) ? 1536 : 768; }
static inline size_t proj_floats(int B, int N) {
  return use_fused_prologue(N) ? 0 : (size_t)B * N * proj_width(N);
}
static inline size_t rtable_floats(int B, int N) {
  return use_rtable(N) ? (size_t)B * N * 8 * N : 0;
}
// (the shapes vrp_persistent_eligible admits: the persistent kernels fold the first chosen node
// into the score rows themselves and need the keys for it)
// (B <= 1024: where the four-wave grid is resident on a whole MI355X.  For these shapes EVERY
// path takes the first node's part of the score rows from persist_first_base -- `base` is a
// function of the shape, not of the path; beyond, the keys' stores cost the prologue more than
// the two launches they replace: +5..9 % at 2048 x 40)
// A/B aid and the parity test: VRP_NO_KEEP_KEYS=1 sends every shape through the GEMM + score_base route
static inline bool keep_keys_off() {
  static const bool v = getenv("VRP_NO_KEEP_KEYS") != nullptr;
  return v;
}
static inline size_t kk_floats(int B, int N) {
  return (N <= 63 && B <= 1024 && use_fused_prologue(N) && !keep_keys_off()) ? (size_t)B * 8 * N * 48 : 0;
}

static inline DecWs carve_decws(void *ws, int B, int N) {
  char *p = (char *)ws;
  DecWs w;
  const size_t R = (size_t)B * N, hn = (size_t)B * 8 * N * 4, tb = R * 8 * N * 4;
  w.g = (float *)p;     p += vrp_align_up((size_t)B * 128 * 4);
  w.QG = (float *)p;    p += vrp_align_up((size_t)B * 384 * 4);
  w.PROJ = (float *)p;  p += vrp_align_up(proj_floats(B, N) * 4);
  w.SG = (float *)p;    p += vrp_align_up(hn);
  w.C0 = (float *)p;    p += vrp_align_up(hn);
  w.SLD = (float *)p;   p += vrp_align_up(hn);
  w.row0 = (float *)p;  p += vrp_align_up(hn);
  w.curs = (float *)p;  p += vrp_align_up(hn);
  w.base = (float *)p;  p += vrp_align_up(hn);
  w.Efirst = (float *)p; p += vrp_align_up((size_t)B * 128 * 4);
  w.FK = (float *)p;     p += vrp_align_up((size_t)B * 1024 * 4);
  w.SL = (float *)p;    p += vrp_align_up(tb);
  w.RT = (float *)p;    p += vrp_align_up(rtable_floats(B, N) * 4);
  w.KK4 = kk_floats(B, N) ? (float *)p : nullptr; p += vrp_align_up(kk_floats(B, N) * 4);
  w.embP = (float *)p;  p += vrp_align_up(pairs_floats(B, N) * 4);
  w.cvec = (float *)p;  p += vrp_align_up(R * 4);
  w.last = (int32_t *)p;  p += vrp_align_up((size_t)B * 4);
  w.first = (int32_t *)p; p += vrp_align_up((size_t)B * 4);
  w.hist = (unsigned long long *)p; p += vrp_align_up((size_t)hist_rows(N) * B * 8);
  w.ta = (int32_t *)p;    p += vrp_align_up((size_t)B * 4);
  w.ret = (float *)p;     p += vrp_align_up((size_t)B * 4);
  w.wb_cur = (int32_t *)p; p += vrp_align_up((size_t)B * 4);
  w.wb_load = (double *)p; p += vrp_align_up((size_t)B * 8);
  w.err = (int32_t *)p;   p += vrp_align_up(4);
  const bool sv = N <= 63;
  w.sv_visited = (uint8_t *)p; p += sv ? vrp_align_up((size_t)B * N) : 0;
  w.sv_cur = (int32_t *)p;     p += sv ? vrp_align_up((size_t)B * 4) : 0;
  w.sv_last = (int32_t *)p;    p += sv ? vrp_align_up((size_t)B * 4) : 0;
  w.sv_accl = (float *)p;      p += sv ? vrp_align_up((size_t)B * 4) : 0;
  w.sv_accp = (float *)p;      p += sv ? vrp_align_up((size_t)B * 4) : 0;
  w.sv_load = (double *)p;     p += sv ? vrp_align_up((size_t)B * 8) : 0;
  return w;
}

static inline int64_t decws_bytes(int B, int N) {
  const size_t R = (size_t)B * N, hn = (size_t)B * 8 * N * 4, tb = R * 8 * N * 4;
  return (int64_t)(vrp_align_up((size_t)B * 128 * 4) + vrp_align_up((size_t)B * 384 * 4) +
                   vrp_align_up(proj_floats(B, N) * 4) + 6 * vrp_align_up(hn) +
                   vrp_align_up((size_t)B * 128 * 4) + vrp_align_up((size_t)B * 1024 * 4) +
                   vrp_align_up(tb) + vrp_align_up(rtable_floats(B, N) * 4) +
                   vrp_align_up(kk_floats(B, N) * 4) +
                   vrp_align_up(pairs_floats(B, N) * 4) +
                   vrp_align_up(R * 4) + 5 * vrp_align_up((size_t)B * 4) +
                   vrp_align_up((size_t)B * 8) +
                   vrp_align_up((size_t)hist_rows(N) * B * 8) + vrp_align_up(4) +
                   persist_save_bytes(B, N));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

int vrp_launch_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                       const float *R, int ldr, float *C, int ldc, int M, int N, int K,
                       int relu, hipStream_t stream);
int vrp_launch_first_base(int kind, const void *derived, int B, int N, const float *emb,
                          void *workspace, hipStream_t st);   // decoder_persistent.hip
int vrp_launch_gemm_gather_k128(const float *A, int lda, const int32_t *gidx, int gstride,
                                const float *W, int ldw, float *C, int ldc, int M, int N,
                                hipStream_t stream);

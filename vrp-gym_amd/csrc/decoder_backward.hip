// Backward of the pointer-attention decoder over a whole recorded episode (K4 of
// SURVEY.md 7.1): d/d(theta, emb) of  sum_b w_b * sum_t log p(a_{t,b})  -- the REINFORCE
// surrogate of agents/graph_tsp_agent.py:178-186 -- for GraphDecoder.forward
// (agents/graph_decoder.py:51-115) applied T times.
//
// The decoder has no recurrent state: step t depends on earlier steps only through the
// recorded indices (first / last node, mask, load).  So all T steps are re-run at once in
// the reference's own (un-folded) form as batched fp32 MFMA GEMMs over the T*B "step rows"
// plus three per-graph kernels, every intermediate stays in HBM, and the backward walks the
// same chain in reverse:
//
//   K|V|KP = emb Wk^T+bk | emb Wv^T+bv | emb Wkp^T          (B*N rows)   graph_decoder.py:82-83
//   ctx    = [g | first | last]   (IRP: [g | last | load] Wc^T)           :75-91
//   Q      = ctx Wq^T + bq                                   (T*B rows)   :93 (in_proj)
//   a,O    = softmax(Q_h K_h^T/sqrt(48) + scrambled mask) V_h             :93-94 (QUIRK D3)
//   O2     = O Wo^T + bo ;  Q2 = O2 Watt^T                                :93,95
//   u      = 10 tanh(Q2 KP^T / sqrt(128)), own mask -> -inf, log-softmax  :96-100
//
// Reductions over steps run in a fixed order (no float atomics): gradients are bitwise
// reproducible.  Layouts: step row r = t*B + b; node row = b*N + n.
#include "common.h"

int vrp_launch_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                       const float *R, int ldr, float *C, int ldc, int M, int N, int K,
                       int relu, hipStream_t stream);
int vrp_launch_gemm_tn(const float *X, int ldx, const float *Y, int ldy, float *C, int R, int N1,
                       int N2, int accumulate, void *slab_ws, hipStream_t st);
extern "C" int64_t vrp_colsum_workspace_bytes(int R, int N);
int vrp_launch_colsum(const float *Y, int ldy, int R, int N, float *out, int accumulate, void *ws,
                      hipStream_t st);
int vrp_launch_transpose(const float *src, int rows, int cols, int lds, float *dst, hipStream_t st);

#define DB_E 128
#define DB_C48 0.14433756729740643f   // 1/sqrt(48)   head dim of the 8-head glimpse
#define DB_C128 0.08838834764831845f  // 1/sqrt(128)  graph_decoder.py:97
#define DBL_LD 132  // LDS row stride of the (N x 128) _kp(emb) tile in the logit kernel
#define DBK_LD 52   // LDS row stride of the (N x 48) K_h / V_h tiles in the backward kernel

// ------------------------------------------------------------------ small helpers
__global__ __launch_bounds__(128) void db_graph_mean_kernel(const float *__restrict__ emb, int N,
                                                            float *__restrict__ g) {
  const int b = blockIdx.x, c = threadIdx.x;
  const float *e = emb + (size_t)b * N * DB_E + c;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += e[(size_t)n * DB_E];
  g[(size_t)b * DB_E + c] = s / (float)N;
}

// dst (rows, ldd) := src (rows, lds)[:, :cols], columns >= cols zeroed up to `width`
__global__ void db_pad_copy_kernel(const float *__restrict__ src, int lds, int cols,
                                   float *__restrict__ dst, int ldd, int width, int rows) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * width) return;
  const int r = idx / width, c = idx - r * width;
  dst[(size_t)r * ldd + c] = c < cols ? src[(size_t)r * lds + c] : 0.f;
}

// Context rows (graph_decoder.py:75-91).  TSP/VRP: [g | first | last]; IRP: [g | last |
// load | 0...] (the input of _context_proj, zero-padded from 257 to 384 columns).
__global__ __launch_bounds__(128) void db_ctx_kernel(int kind, int B, int N, int T,
                                                     const float *__restrict__ emb,
                                                     const float *__restrict__ g,
                                                     const float *__restrict__ first_node,
                                                     const float *__restrict__ last_node,
                                                     const int64_t *__restrict__ actions,
                                                     const float *__restrict__ loads,
                                                     float *__restrict__ ctx) {
  const int r = blockIdx.x, c = threadIdx.x;
  const int t = r / B, b = r - t * B;
  float *row = ctx + (size_t)r * VRP_D;
  const float gv = g[(size_t)b * DB_E + c];
  float lastv;
  if (t == 0) lastv = last_node[c];
  else lastv = emb[((size_t)b * N + (int)actions[(size_t)(t - 1) * B + b]) * DB_E + c];
  row[c] = gv;
  if (kind != VRP_KIND_IRP) {
    float firstv;
    if (t == 0) firstv = first_node[c];
    else firstv = emb[((size_t)b * N + (int)actions[b]) * DB_E + c];
    row[128 + c] = firstv;
    row[256 + c] = lastv;
  } else {
    row[128 + c] = lastv;
    row[256 + c] = (c == 0) ? loads[r] : 0.f;
  }
}

// ------------------------------------------------------------------ glimpse attention, forward
// One workgroup per (graph, head); K_h and V_h (N x 48) staged once in LDS and reused by all
// T steps; wave w handles steps w, w+4, ...  lane = node for the scores, lane = d for o.
template <int NPL>
__global__ __launch_bounds__(256) void db_attn_fwd_kernel(int B, int N, int T,
                                                          const float *__restrict__ Q,
                                                          const float *__restrict__ Kb,
                                                          const float *__restrict__ Vb,
                                                          const uint8_t *__restrict__ masks,
                                                          float *__restrict__ A,
                                                          float *__restrict__ O) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *K_s = smem;                 // [N][52]: 16-byte aligned rows, conflict-free b128 reads
  float *V_s = K_s + N * DBK_LD;     // [N][48]
  float *q_s = V_s + N * 48;         // [4][48]
  float *a_s = q_s + 4 * 48;         // [4][64*NPL]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, h = blockIdx.y;
  for (int idx = tid; idx < N * 48; idx += 256) {
    const int n = idx / 48, d = idx - n * 48;
    const size_t src = ((size_t)b * N + n) * VRP_D + h * VRP_HD + d;
    K_s[n * DBK_LD + d] = Kb[src];
    V_s[n * 48 + d] = Vb[src];
  }
  __syncthreads();
  const int mrow = (b * 8 + h) % B;  // QUIRK D3: head h of graph b reads this graph's mask
  for (int t0 = 0; t0 < T; t0 += 4) {
    const int t = t0 + wave;
    const bool on = t < T;
    const size_t r = (size_t)(on ? t : 0) * B + b;
    if (lane < 48) q_s[wave * 48 + lane] = Q[r * VRP_D + h * VRP_HD + lane];
    __syncthreads();
    float s[NPL], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int n = lane + 64 * i;
      s[i] = -INFINITY;
      if (n < N) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < 48; d += 4) {
          const float4 q4 = *reinterpret_cast<const float4 *>(q_s + wave * 48 + d);
          const float4 k4 = *reinterpret_cast<const float4 *>(K_s + n * DBK_LD + d);
          acc = fmaf(q4.x, k4.x, acc); acc = fmaf(q4.y, k4.y, acc);
          acc = fmaf(q4.z, k4.z, acc); acc = fmaf(q4.w, k4.w, acc);
        }
        s[i] = acc * DB_C48 + (float)masks[((size_t)(on ? t : 0) * B + mrow) * N + n];
      }
      mx = fmaxf(mx, s[i]);
    }
    const float m = wave_max(mx);
    float e[NPL], es = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { e[i] = (lane + 64 * i < N) ? expf(s[i] - m) : 0.f; es += e[i]; }
    const float sum = wave_sum(es);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int n = lane + 64 * i;
      if (n < N) {
        const float a = e[i] / sum;
        a_s[wave * 64 * NPL + n] = a;
        if (on) A[(r * 8 + h) * N + n] = a;
      }
    }
    __syncthreads();
    if (lane < 48 && on) {
      float o = 0.f;
      for (int n = 0; n < N; ++n) o = fmaf(a_s[wave * 64 * NPL + n], V_s[n * 48 + lane], o);
      O[r * VRP_D + h * VRP_HD + lane] = o;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ glimpse attention, backward
// Same decomposition.  Per step: da = dO_h V_h^T, ds = a (da - <a,da>), dQ_h = ds K_h / sqrt(48).
// dK_h += ds^T Q_h / sqrt(48) and dV_h += a^T dO_h are accumulated over the wave's steps in
// registers (lane = node, 48 values each) and the four waves are summed in order at the end.
template <int NPL>
__global__ __launch_bounds__(256) void db_attn_bwd_kernel(int B, int N, int T,
                                                          const float *__restrict__ Q,
                                                          const float *__restrict__ Kb,
                                                          const float *__restrict__ Vb,
                                                          const float *__restrict__ A,
                                                          const float *__restrict__ dO,
                                                          float *__restrict__ dQ,
                                                          float *__restrict__ dK,
                                                          float *__restrict__ dV) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *K_s = smem;                  // [N][52]: rows 16-byte aligned, b128 reads conflict-free
  float *V_s = K_s + N * DBK_LD;      // [N][52]
  float *ds_s = V_s + N * DBK_LD;         // [4][64*NPL]  (each wave touches only its own row)
  float *qd_s = ds_s + 4 * 64 * NPL;  // [4][96]      this step's q | dO rows, per wave
  float *red_s = qd_s + 4 * 96;       // [N][49]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, h = blockIdx.y;
  for (int idx = tid; idx < N * 48; idx += 256) {
    const int n = idx / 48, d = idx - n * 48;
    const size_t src = ((size_t)b * N + n) * VRP_D + h * VRP_HD + d;
    K_s[n * DBK_LD + d] = Kb[src];
    V_s[n * DBK_LD + d] = Vb[src];
  }
  float dKa[NPL][48], dVa[NPL][48];
#pragma unroll
  for (int i = 0; i < NPL; ++i)
#pragma unroll
    for (int d = 0; d < 48; ++d) { dKa[i][d] = 0.f; dVa[i][d] = 0.f; }
  __syncthreads();
  // Each wave walks its own steps (t = wave, wave+4, ...) without workgroup barriers: its
  // query / output-gradient rows pass through a wave-private LDS slot (broadcast reads), and
  // the rows and attention weights of the NEXT step are loaded while this one is computed --
  // with two workgroups per CU the global-load latency of a step would otherwise be exposed.
  float *dsw = ds_s + wave * 64 * NPL;
  float *qw = qd_s + wave * 96, *dow = qw + 48;
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  float qn = 0.f, don = 0.f, an[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) an[i] = 0.f;
  auto prefetch = [&](int t) {
    if (t < T) {
      const size_t r = (size_t)t * B + b;
      if (lane < 48) {
        qn = Q[r * VRP_D + h * VRP_HD + lane];
        don = dO[r * VRP_D + h * VRP_HD + lane];
      }
#pragma unroll
      for (int i = 0; i < NPL; ++i)
        if (lane + 64 * i < N) an[i] = A[(r * 8 + h) * N + lane + 64 * i];
    }
  };
  prefetch(wave);
  for (int t = wave; t < T; t += 4) {
    const size_t r = (size_t)t * B + b;
    if (lane < 48) { qw[lane] = qn; dow[lane] = don; }
    float a[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) a[i] = an[i];
    wave_sync();
    prefetch(t + 4);
    float da[NPL], part = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int n = lane + 64 * i;
      da[i] = 0.f;
      if (n < N) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < 48; d += 4) {
          const float4 g4 = *reinterpret_cast<const float4 *>(dow + d);
          const float4 v4 = *reinterpret_cast<const float4 *>(V_s + n * DBK_LD + d);
          acc = fmaf(g4.x, v4.x, acc); acc = fmaf(g4.y, v4.y, acc);
          acc = fmaf(g4.z, v4.z, acc); acc = fmaf(g4.w, v4.w, acc);
        }
        da[i] = acc;
      } else {
        a[i] = 0.f;
      }
      part = fmaf(a[i], da[i], part);
    }
    const float dot = wave_sum(part);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int n = lane + 64 * i;
      const float ds = a[i] * (da[i] - dot);
      if (n < N) dsw[n] = ds;
      const float dsc = ds * DB_C48;
#pragma unroll
      for (int d = 0; d < 48; d += 4) {
        const float4 q4 = *reinterpret_cast<const float4 *>(qw + d);
        const float4 g4 = *reinterpret_cast<const float4 *>(dow + d);
        dKa[i][d] = fmaf(dsc, q4.x, dKa[i][d]);         dVa[i][d] = fmaf(a[i], g4.x, dVa[i][d]);
        dKa[i][d + 1] = fmaf(dsc, q4.y, dKa[i][d + 1]); dVa[i][d + 1] = fmaf(a[i], g4.y, dVa[i][d + 1]);
        dKa[i][d + 2] = fmaf(dsc, q4.z, dKa[i][d + 2]); dVa[i][d + 2] = fmaf(a[i], g4.z, dVa[i][d + 2]);
        dKa[i][d + 3] = fmaf(dsc, q4.w, dKa[i][d + 3]); dVa[i][d + 3] = fmaf(a[i], g4.w, dVa[i][d + 3]);
      }
    }
    wave_sync();
    if (lane < 48) {
      float dq = 0.f;
      for (int n = 0; n < N; ++n) dq = fmaf(dsw[n], K_s[n * DBK_LD + lane], dq);
      dQ[r * VRP_D + h * VRP_HD + lane] = dq * DB_C48;
    }
    wave_sync();
  }
  __syncthreads();
  // ordered sum over the four waves: wave 0 + wave 1 + wave 2 + wave 3
  for (int pass = 0; pass < 2; ++pass) {
    float (&acc)[NPL][48] = pass == 0 ? dKa : dVa;
    for (int w = 1; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          const int n = lane + 64 * i;
          if (n < N) {
#pragma unroll
            for (int d = 0; d < 48; ++d) red_s[n * DBK_LD + d] = acc[i][d];
          }
        }
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
          const int n = lane + 64 * i;
          if (n < N) {
#pragma unroll
            for (int d = 0; d < 48; ++d) acc[i][d] += red_s[n * DBK_LD + d];
          }
        }
      }
      __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int n = lane + 64 * i;
        if (n < N) {
#pragma unroll
          for (int d = 0; d < 48; ++d) red_s[n * DBK_LD + d] = acc[i][d];
        }
      }
    }
    __syncthreads();
    {
      float *dst = pass == 0 ? dK : dV;
      for (int idx = tid; idx < N * 48; idx += 256) {
        const int n = idx / 48, d = idx - n * 48;
        dst[((size_t)b * N + n) * VRP_D + h * VRP_HD + d] = red_s[n * DBK_LD + d];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ glimpse attention on the matrix cores
// N <= 48 (three 16-node tiles; the training configurations run N = 40): ONE WAVE per (graph,
// head), no LDS, no barrier -- every product of the forward and of the backward is a chain of
// v_mfma_f32_16x16x4_f32 over 16-step tiles of the episode.  The VALU kernels above are bound by
// LDS bandwidth (per step and wave ~50 ds_read_b128 of K_h / V_h rows and broadcast q / dO rows:
// 1.2 ms of a 18.5 ms VRP-40 x 2048 epoch); here K_h and V_h sit in registers as MFMA operands for
// the whole episode.  Lane (c, g) = (lane & 15, lane >> 4).  The inner index of every product is
// permuted the same way for both operands (an MFMA k-step hands lane group g ONE inner index; any
// bijection works), chosen so that the accumulator layout of one product IS an operand of the
// next:
//   "row" fragments   X[16 tile + c][12 g + s],   s = 0..11   three 16-byte loads of a row piece
//   "col" fragments   X[16 tile + 4 g + i][16 dt + c], i < 4  one dword each
//   D of a product    D[4 g + i][c]
// Forward:  S^T(tn)  = K(row) Q(row)^T       -> lane (t = c) holds n = 16 tn + 4 g + i: softmax over
//                                               n in the lane + two xor steps; a^T IS the B operand of
//           O^T(dt)  = V(col) a^T            -> lane (t = c) holds d = 16 dt + 4 g + i: 16-byte store.
// Backward: dA^T(tn) = V(row) dO(row)^T      -> ds^T in the layout of a^T above (a re-read so),
//           dQ^T(dt) = K(col) ds^T           -> 16-byte stores;
//           dA(tn)   = dO(row) V(row)^T      -> the same registers, operands swapped: lane (n = c)
//                                               holds t = 4 g + i, the B operand (inner index t) of
//           dK^T(dt,tn) += Q(col) ds,  dV^T(dt,tn) += dO(col) a     (a re-read in this layout)
// accumulated over the episode's tiles in registers: lane (n = c) ends with four consecutive d.
// Steps beyond T and nodes beyond N are zero operands.  Same mathematics as the VALU kernels,
// another summation order; deterministic (no atomics).
typedef float db_f4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ void db_load_row_frag(float (&f)[NT][12], const float *base, int b, int N,
                                                 int h, int c, int g) {
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) {
    const int n = 16 * tn + c;
    const bool on = n < N;
    const float4 *src = reinterpret_cast<const float4 *>(
        base + ((size_t)b * N + (on ? n : 0)) * VRP_D + h * VRP_HD + 12 * g);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 v = on ? src[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      f[tn][4 * k] = v.x; f[tn][4 * k + 1] = v.y; f[tn][4 * k + 2] = v.z; f[tn][4 * k + 3] = v.w;
    }
  }
}
template <int NT>
__device__ __forceinline__ void db_load_col_frag(float (&f)[NT][4][3], const float *base, int b, int N,
                                                 int h, int c, int g) {
#pragma unroll
  for (int tn = 0; tn < NT; ++tn)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 16 * tn + 4 * g + i;
      const bool on = n < N;
      const float *src = base + ((size_t)b * N + (on ? n : 0)) * VRP_D + h * VRP_HD + c;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) f[tn][i][dt] = on ? src[16 * dt] : 0.f;
    }
}
// attention weights of step tile tt: lane (t = c) x nodes 16 tn + 4 g + i  (the layout of S^T)
template <int NT, bool VEC>
__device__ __forceinline__ void db_load_a_t(float (&a)[NT][4], const float *A, size_t r, bool on_t,
                                            int h, int N, int g) {
  const float *row = A + (r * 8 + h) * N;
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) {
    const int n0 = 16 * tn + 4 * g;
    if (VEC) {
      const float4 v = (on_t && n0 < N) ? *reinterpret_cast<const float4 *>(row + n0)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
      a[tn][0] = v.x; a[tn][1] = v.y; a[tn][2] = v.z; a[tn][3] = v.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[tn][i] = (on_t && n0 + i < N) ? row[n0 + i] : 0.f;
    }
  }
}

template <int NT, bool VEC>
__global__ __launch_bounds__(256, 2) void db_attn_fwd_mfma_kernel(int B, int N, int T,
                                                                  const float *__restrict__ Q,
                                                                  const float *__restrict__ Kb,
                                                                  const float *__restrict__ Vb,
                                                                  const uint8_t *__restrict__ masks,
                                                                  float *__restrict__ A,
                                                                  float *__restrict__ O) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, h = blockIdx.y * 4 + wave;
  const int mrow = (b * 8 + h) % B;  // QUIRK D3: head h of graph b reads this graph's mask
  float KR[NT][12], VC[NT][4][3];
  db_load_row_frag<NT>(KR, Kb, b, N, h, c, g);
  db_load_col_frag<NT>(VC, Vb, b, N, h, c, g);
  for (int t0 = 0; t0 < T; t0 += 16) {
    const int t = t0 + c;
    const bool on = t < T;
    const size_t r = (size_t)(on ? t : 0) * B + b;
    float QR[12];
    {
      const float4 *src = reinterpret_cast<const float4 *>(Q + r * VRP_D + h * VRP_HD + 12 * g);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 v = on ? src[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        QR[4 * k] = v.x; QR[4 * k + 1] = v.y; QR[4 * k + 2] = v.z; QR[4 * k + 3] = v.w;
      }
    }
    float mk[NT][4];
    {
      const uint8_t *mr = masks + ((size_t)(on ? t : 0) * B + mrow) * N;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int n0 = 16 * tn + 4 * g;
        if (VEC) {
          const uint32_t m4 = (n0 < N) ? *reinterpret_cast<const uint32_t *>(mr + n0) : 0u;
#pragma unroll
          for (int i = 0; i < 4; ++i) mk[tn][i] = (float)((m4 >> (8 * i)) & 0xffu);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) mk[tn][i] = (n0 + i < N) ? (float)mr[n0 + i] : 0.f;
        }
      }
    }
    float s[NT][4], mx = -INFINITY;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      db_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 12; ++k) d = __builtin_amdgcn_mfma_f32_16x16x4f32(KR[tn][k], QR[k], d, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool nv = 16 * tn + 4 * g + i < N;
        s[tn][i] = nv ? d[i] * DB_C48 + mk[tn][i] : -INFINITY;
        mx = fmaxf(mx, s[tn][i]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float es = 0.f;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[tn][i] = (16 * tn + 4 * g + i < N) ? expf(s[tn][i] - mx) : 0.f;
        es += s[tn][i];
      }
    es += __shfl_xor(es, 16, 64);
    es += __shfl_xor(es, 32, 64);
    float *arow = A + (r * 8 + h) * N;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
      for (int i = 0; i < 4; ++i) s[tn][i] = s[tn][i] / es;
      const int n0 = 16 * tn + 4 * g;
      if (on) {
        if (VEC) {
          if (n0 < N) *reinterpret_cast<float4 *>(arow + n0) = make_float4(s[tn][0], s[tn][1], s[tn][2], s[tn][3]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (n0 + i < N) arow[n0 + i] = s[tn][i];
        }
      }
    }
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      db_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(VC[tn][i][dt], s[tn][i], d, 0, 0, 0);
      if (on)
        *reinterpret_cast<float4 *>(O + r * VRP_D + h * VRP_HD + 16 * dt + 4 * g) =
            make_float4(d[0], d[1], d[2], d[3]);
    }
  }
}

template <int NT, bool VEC>
__global__ __launch_bounds__(256, 2) void db_attn_bwd_mfma_kernel(int B, int N, int T,
                                                                  const float *__restrict__ Q,
                                                                  const float *__restrict__ Kb,
                                                                  const float *__restrict__ Vb,
                                                                  const float *__restrict__ A,
                                                                  const float *__restrict__ dO,
                                                                  float *__restrict__ dQ,
                                                                  float *__restrict__ dK,
                                                                  float *__restrict__ dV) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, h = blockIdx.y * 4 + wave;
  float VR[NT][12], KC[NT][4][3];
  db_load_row_frag<NT>(VR, Vb, b, N, h, c, g);
  db_load_col_frag<NT>(KC, Kb, b, N, h, c, g);
  db_f4 accK[3][NT], accV[3][NT];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      accK[dt][tn] = db_f4{0.f, 0.f, 0.f, 0.f};
      accV[dt][tn] = db_f4{0.f, 0.f, 0.f, 0.f};
    }
  for (int t0 = 0; t0 < T; t0 += 16) {
    // ---- this tile's rows, in both layouts ------------------------------------------------
    const int t1 = t0 + c;
    const bool on1 = t1 < T;
    const size_t r1 = (size_t)(on1 ? t1 : 0) * B + b;
    float DR[12];   // dO[t = c][12 g + s]
    {
      const float4 *src = reinterpret_cast<const float4 *>(dO + r1 * VRP_D + h * VRP_HD + 12 * g);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 v = on1 ? src[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        DR[4 * k] = v.x; DR[4 * k + 1] = v.y; DR[4 * k + 2] = v.z; DR[4 * k + 3] = v.w;
      }
    }
    float a1[NT][4];
    db_load_a_t<NT, VEC>(a1, A, r1, on1, h, N, g);
    float QC[4][3], DC[4][3], a2[NT][4];   // Q / dO [t = 4 g + i][16 dt + c];  a[t = 4 g + i][n = c]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t2 = t0 + 4 * g + i;
      const bool on2 = t2 < T;
      const size_t r2 = (size_t)(on2 ? t2 : 0) * B + b;
      const float *qs = Q + r2 * VRP_D + h * VRP_HD + c, *ds = dO + r2 * VRP_D + h * VRP_HD + c;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        QC[i][dt] = on2 ? qs[16 * dt] : 0.f;
        DC[i][dt] = on2 ? ds[16 * dt] : 0.f;
      }
      const float *ar = A + (r2 * 8 + h) * N;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) a2[tn][i] = (on2 && 16 * tn + c < N) ? ar[16 * tn + c] : 0.f;
    }
    // ---- dA^T = V dO^T, ds^T = a (dA - <a, dA>), dQ^T = K ds^T ------------------------------
    float ds1[NT][4], part = 0.f;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      db_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 12; ++k) d = __builtin_amdgcn_mfma_f32_16x16x4f32(VR[tn][k], DR[k], d, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) { ds1[tn][i] = d[i]; part = fmaf(a1[tn][i], d[i], part); }
    }
    part += __shfl_xor(part, 16, 64);
    const float dot = part + __shfl_xor(part, 32, 64);   // <a_t, dA_t> of step t = c, in every group
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int i = 0; i < 4; ++i) ds1[tn][i] = a1[tn][i] * (ds1[tn][i] - dot);
#pragma unroll
    for (int dt = 0; dt < 3; ++dt) {
      db_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(KC[tn][i][dt], ds1[tn][i], d, 0, 0, 0);
      if (on1)
        *reinterpret_cast<float4 *>(dQ + r1 * VRP_D + h * VRP_HD + 16 * dt + 4 * g) =
            make_float4(d[0] * DB_C48, d[1] * DB_C48, d[2] * DB_C48, d[3] * DB_C48);
    }
    // ---- dA = dO V^T (lane = node), ds, dK^T += Q^T ds, dV^T += dO^T a ----------------------
    float dot2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dot2[i] = __shfl(dot, 4 * g + i, 64);   // step t = 4 g + i
    float ds2[NT][4];
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      db_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 12; ++k) d = __builtin_amdgcn_mfma_f32_16x16x4f32(DR[k], VR[tn][k], d, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) ds2[tn][i] = a2[tn][i] * (d[i] - dot2[i]);
    }
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          accK[dt][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(QC[i][dt], ds2[tn][i], accK[dt][tn], 0, 0, 0);
          accV[dt][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(DC[i][dt], a2[tn][i], accV[dt][tn], 0, 0, 0);
        }
  }
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) {
    const int n = 16 * tn + c;
    if (n < N) {
      const size_t o = ((size_t)b * N + n) * VRP_D + h * VRP_HD + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const db_f4 k4 = accK[dt][tn], v4 = accV[dt][tn];
        *reinterpret_cast<float4 *>(dK + o + 16 * dt) =
            make_float4(k4[0] * DB_C48, k4[1] * DB_C48, k4[2] * DB_C48, k4[3] * DB_C48);
        *reinterpret_cast<float4 *>(dV + o + 16 * dt) = make_float4(v4[0], v4[1], v4[2], v4[3]);
      }
    }
  }
}

// N <= 48: the matrix-core kernels (A/B: VRP_DB_ATTN_VALU=1 keeps the VALU ones)
static bool db_attn_mfma_applies(int N) {
  static const bool off = getenv("VRP_DB_ATTN_VALU") != nullptr;
  return !off && N <= 48;
}
template <int NT, bool VEC>
static void db_launch_attn_fwd_mfma(int B, int N, int T, const float *Q, const float *Kb, const float *Vb,
                                    const uint8_t *masks, float *A, float *O, hipStream_t st) {
  hipLaunchKernelGGL((db_attn_fwd_mfma_kernel<NT, VEC>), dim3(B, 2), dim3(256), 0, st, B, N, T, Q, Kb,
                     Vb, masks, A, O);
}
template <int NT, bool VEC>
static void db_launch_attn_bwd_mfma(int B, int N, int T, const float *Q, const float *Kb, const float *Vb,
                                    const float *A, const float *dO, float *dQ, float *dK, float *dV,
                                    hipStream_t st) {
  hipLaunchKernelGGL((db_attn_bwd_mfma_kernel<NT, VEC>), dim3(B, 2), dim3(256), 0, st, B, N, T, Q, Kb,
                     Vb, A, dO, dQ, dK, dV);
}
#define DB_ATTN_DISPATCH(FN, ...)                                        \
  do {                                                                   \
    const int nt_ = (N + 15) / 16;                                       \
    const bool vec_ = (N & 3) == 0;                                      \
    if (nt_ == 1) { if (vec_) FN<1, true>(__VA_ARGS__); else FN<1, false>(__VA_ARGS__); }      \
    else if (nt_ == 2) { if (vec_) FN<2, true>(__VA_ARGS__); else FN<2, false>(__VA_ARGS__); } \
    else { if (vec_) FN<3, true>(__VA_ARGS__); else FN<3, false>(__VA_ARGS__); }               \
  } while (0)

// ------------------------------------------------------------------ pointer logits + log-softmax
// One workgroup per graph, KP_b (N x 128) in LDS.  Forward u = 10 tanh(Q2.KP_n/sqrt(128)),
// own mask -> -inf, log p = u - logsumexp(u) (graph_decoder.py:96-100, graph_tsp_agent.py:86);
// backward of  w_b * log p(a_t):  du = w_b (onehot(a_t) - p), dz = du 10 (1 - tanh^2)/sqrt(128),
// dQ2 = dz KP,  dKP += dz^T Q2 (registers, lane = node; waves summed in order).
template <int NPL>
__global__ __launch_bounds__(256) void db_logit_kernel(int B, int N, int T,
                                                       const float *__restrict__ Q2,
                                                       const float *__restrict__ KP,
                                                       const uint8_t *__restrict__ masks,
                                                       const int64_t *__restrict__ actions,
                                                       const float *__restrict__ d_logp,
                                                       float *__restrict__ dQ2,
                                                       float *__restrict__ dKP,
                                                       float *__restrict__ step_logp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *KP_s = smem;                  // [N][132]
  float *q_s = KP_s + N * DBL_LD;         // [4][128]
  float *dz_s = q_s + 4 * 128;         // [4][64*NPL]
  float *red_s = dz_s + 4 * 64 * NPL;  // [N][132]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  for (int idx = tid; idx < N * 128; idx += 256) {
    const int n = idx >> 7, k = idx & 127;
    KP_s[n * DBL_LD + k] = KP[((size_t)b * N + n) * DB_E + k];
  }
  float acc[NPL][128];
#pragma unroll
  for (int i = 0; i < NPL; ++i)
#pragma unroll
    for (int k = 0; k < 128; ++k) acc[i][k] = 0.f;
  const float wgt = d_logp[b];
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 4) {
    const int t = t0 + wave;
    const bool on = t < T;
    const size_t r = (size_t)(on ? t : 0) * B + b;
    q_s[wave * 128 + lane] = Q2[r * DB_E + lane];
    q_s[wave * 128 + 64 + lane] = Q2[r * DB_E + 64 + lane];
    __syncthreads();
    const int act = (int)actions[r];
    float u[NPL], th[NPL], mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int n = lane + 64 * i;
      u[i] = -INFINITY; th[i] = 0.f;
      if (n < N && !masks[r * N + n]) {
        float z = 0.f;
#pragma unroll 8
        for (int k = 0; k < 128; k += 4) {
          const float4 q4 = *reinterpret_cast<const float4 *>(q_s + wave * 128 + k);
          const float4 p4 = *reinterpret_cast<const float4 *>(KP_s + n * DBL_LD + k);
          z = fmaf(q4.x, p4.x, z); z = fmaf(q4.y, p4.y, z);
          z = fmaf(q4.z, p4.z, z); z = fmaf(q4.w, p4.w, z);
        }
        th[i] = tanhf(z * DB_C128);
        u[i] = 10.f * th[i];
      }
      mx = fmaxf(mx, u[i]);
    }
    const float m = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) se += expf(u[i] - m);
    se = wave_sum(se);
    const float lse = m + logf(se);
    float lp_part = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int n = lane + 64 * i;
      const float p = expf(u[i] - lse);  // 0 for masked nodes
      float dz = 0.f;
      if (n < N && u[i] > -INFINITY && on) {
        const float du = wgt * ((n == act ? 1.f : 0.f) - p);
        dz = du * 10.f * (1.f - th[i] * th[i]) * DB_C128;
      }
      if (n == act) lp_part += u[i] - lse;
      if (n < N) dz_s[wave * 64 * NPL + n] = dz;
#pragma unroll
      for (int k = 0; k < 128; k += 4) {
        const float4 q4 = *reinterpret_cast<const float4 *>(q_s + wave * 128 + k);
        acc[i][k] = fmaf(dz, q4.x, acc[i][k]);
        acc[i][k + 1] = fmaf(dz, q4.y, acc[i][k + 1]);
        acc[i][k + 2] = fmaf(dz, q4.z, acc[i][k + 2]);
        acc[i][k + 3] = fmaf(dz, q4.w, acc[i][k + 3]);
      }
    }
    const float lp = wave_sum(lp_part);
    if (step_logp && on && lane == 0) step_logp[r] = lp;
    __syncthreads();
    if (on) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = lane + 64 * j;
        float dq = 0.f;
        for (int n = 0; n < N; ++n) dq = fmaf(dz_s[wave * 64 * NPL + n], KP_s[n * DBL_LD + k], dq);
        dQ2[r * DB_E + k] = dq;
      }
    }
    __syncthreads();
  }
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int n = lane + 64 * i;
        if (n < N) {
#pragma unroll
          for (int k = 0; k < 128; ++k) red_s[n * DBL_LD + k] = acc[i][k];
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int n = lane + 64 * i;
        if (n < N) {
#pragma unroll
          for (int k = 0; k < 128; ++k) acc[i][k] += red_s[n * DBL_LD + k];
        }
      }
    }
    __syncthreads();
  }
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int n = lane + 64 * i;
      if (n < N) {
#pragma unroll
        for (int k = 0; k < 128; ++k) red_s[n * DBL_LD + k] = acc[i][k];
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < N * 128; idx += 256) {
    const int n = idx >> 7, k = idx & 127;
    dKP[((size_t)b * N + n) * DB_E + k] = red_s[n * DBL_LD + k];
  }
}

// The same step on the matrix cores (N <= 48): one wave per graph, KP_b (N x 128) in LDS, 16-step
// tiles.  Z^T(tn) = KP Q2^T puts step t on the lane (c) and nodes 16 tn + 4 g + i in its registers:
// tanh, mask, log-softmax and dz are in-lane plus two xor steps, and dz^T IS the B operand of
// dQ2^T(dt) = KP^T dz^T (A operand: one dword of KP_s per MFMA).  dz goes through a 16 x 48 LDS
// tile once to put the node on the lane (t = 4 g + i in the registers): the B operand of
// dKP^T(dt, tn) += Q2^T dz, accumulated over the episode in registers (96 of them).
#define DBZ_LD 49
template <int NT, bool VEC>
__global__ __launch_bounds__(64) void db_logit_mfma_kernel(int B, int N, int T,
                                                           const float *__restrict__ Q2,
                                                           const float *__restrict__ KP,
                                                           const uint8_t *__restrict__ masks,
                                                           const int64_t *__restrict__ actions,
                                                           const float *__restrict__ d_logp,
                                                           float *__restrict__ dQ2,
                                                           float *__restrict__ dKP,
                                                           float *__restrict__ step_logp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *KP_s = smem;                       // [16 NT][132], rows >= N zero
  float *dz_s = KP_s + 16 * NT * DBL_LD;    // [16][DBZ_LD]
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  for (int idx = lane; idx < 16 * NT * 32; idx += 64) {
    const int n = idx >> 5, k4 = idx & 31;
    const float4 v = n < N ? reinterpret_cast<const float4 *>(KP + ((size_t)b * N + n) * DB_E)[k4]
                           : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4 *>(KP_s + n * DBL_LD + 4 * k4) = v;
  }
  db_f4 acc[8][NT];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) acc[dt][tn] = db_f4{0.f, 0.f, 0.f, 0.f};
  const float wgt = d_logp[b];
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 16) {
    const int t1 = t0 + c;
    const bool on1 = t1 < T;
    const size_t r1 = (size_t)(on1 ? t1 : 0) * B + b;
    float QR[32];   // Q2[t = c][32 g + s]
    {
      const float4 *src = reinterpret_cast<const float4 *>(Q2 + r1 * DB_E + 32 * g);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 v = on1 ? src[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        QR[4 * k] = v.x; QR[4 * k + 1] = v.y; QR[4 * k + 2] = v.z; QR[4 * k + 3] = v.w;
      }
    }
    float QC[4][8];  // Q2[t = 4 g + i][16 dt + c]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t2 = t0 + 4 * g + i;
      const bool on2 = t2 < T;
      const float *src = Q2 + ((size_t)(on2 ? t2 : 0) * B + b) * DB_E + c;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) QC[i][dt] = on2 ? src[16 * dt] : 0.f;
    }
    const int act = (int)actions[r1];
    bool open[NT][4];   // selectable: inside the graph and not masked (graph_decoder.py:98)
    {
      const uint8_t *mr = masks + r1 * N;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        const int n0 = 16 * tn + 4 * g;
        if (VEC) {
          const uint32_t m4 = (n0 < N) ? *reinterpret_cast<const uint32_t *>(mr + n0) : 0xffffffffu;
#pragma unroll
          for (int i = 0; i < 4; ++i) open[tn][i] = ((m4 >> (8 * i)) & 0xffu) == 0;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) open[tn][i] = (n0 + i < N) && mr[n0 + i] == 0;
        }
      }
    }
    // ---- Z^T, u = 10 tanh(z / sqrt(128)), log-softmax over the open nodes ----------------------
    float u[NT][4], th[NT][4], mx = -INFINITY;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      db_f4 d = {0.f, 0.f, 0.f, 0.f};
      const float *kr = KP_s + (16 * tn + c) * DBL_LD + 32 * g;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 a = *reinterpret_cast<const float4 *>(kr + 4 * k);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, QR[4 * k], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, QR[4 * k + 1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, QR[4 * k + 2], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, QR[4 * k + 3], d, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        th[tn][i] = open[tn][i] ? tanhf(d[i] * DB_C128) : 0.f;
        u[tn][i] = open[tn][i] ? 10.f * th[tn][i] : -INFINITY;
        mx = fmaxf(mx, u[tn][i]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float se = 0.f;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int i = 0; i < 4; ++i) se += expf(u[tn][i] - mx);
    se += __shfl_xor(se, 16, 64);
    se += __shfl_xor(se, 32, 64);
    const float lse = mx + logf(se);
    float dz[NT][4], lp = 0.f;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = 16 * tn + 4 * g + i;
        const float pn = expf(u[tn][i] - lse);   // 0 for closed nodes
        const float du = wgt * ((n == act ? 1.f : 0.f) - pn);
        dz[tn][i] = (open[tn][i] && on1) ? du * 10.f * (1.f - th[tn][i] * th[tn][i]) * DB_C128 : 0.f;
        if (n == act) lp += u[tn][i] - lse;
      }
    lp += __shfl_xor(lp, 16, 64);
    lp += __shfl_xor(lp, 32, 64);
    if (step_logp && on1 && g == 0) step_logp[r1] = lp;
    // ---- dQ2^T = KP^T dz^T ----------------------------------------------------------------
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      db_f4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(KP_s[(16 * tn + 4 * g + i) * DBL_LD + 16 * dt + c],
                                                   dz[tn][i], d, 0, 0, 0);
      if (on1)
        *reinterpret_cast<float4 *>(dQ2 + r1 * DB_E + 16 * dt + 4 * g) = make_float4(d[0], d[1], d[2], d[3]);
    }
    // ---- dz with the node on the lane, dKP^T += Q2^T dz -----------------------------------
    __syncthreads();   // (one wave: orders the previous tile's reads of dz_s before these writes)
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int i = 0; i < 4; ++i) dz_s[c * DBZ_LD + 16 * tn + 4 * g + i] = dz[tn][i];
    __syncthreads();
    float dz2[NT][4];
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int i = 0; i < 4; ++i) dz2[tn][i] = dz_s[(4 * g + i) * DBZ_LD + 16 * tn + c];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[dt][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(QC[i][dt], dz2[tn][i], acc[dt][tn], 0, 0, 0);
  }
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) {
    const int n = 16 * tn + c;
    if (n < N) {
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const db_f4 v = acc[dt][tn];
        *reinterpret_cast<float4 *>(dKP + ((size_t)b * N + n) * DB_E + 16 * dt + 4 * g) =
            make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}
template <int NT, bool VEC>
static void db_launch_logit_mfma(int B, int N, int T, const float *Q2, const float *KP,
                                 const uint8_t *masks, const int64_t *actions, const float *d_logp,
                                 float *dQ2, float *dKP, float *step_logp, hipStream_t st) {
  const size_t lds = sizeof(float) * ((size_t)16 * NT * DBL_LD + 16 * DBZ_LD);
  hipLaunchKernelGGL((db_logit_mfma_kernel<NT, VEC>), dim3(B), dim3(64), lds, st, B, N, T, Q2, KP, masks,
                     actions, d_logp, dQ2, dKP, step_logp);
}

// ------------------------------------------------------------------ context rows -> d_emb
// D (T*B,384) = gradient of the context rows ([dg | dfirst | dlast], IRP: [dg | dlast | ..]).
// One workgroup per graph, thread = embedding column, steps visited in order:
//   d_emb[b][n] = (sum_t dg)/N ;  d_emb[b][first] += sum_{t>=1} dfirst ;  d_emb[b][last_t] += dlast_t
__global__ __launch_bounds__(128) void db_scatter_kernel(int kind, int B, int N, int T,
                                                         const float *__restrict__ D,
                                                         const int64_t *__restrict__ actions,
                                                         float *__restrict__ d_emb) {
  const int b = blockIdx.x, c = threadIdx.x;
  const int lastoff = (kind == VRP_KIND_IRP) ? 128 : 256;
  float sg = 0.f, sf = 0.f;
  for (int t = 0; t < T; ++t) {
    const float *row = D + ((size_t)t * B + b) * VRP_D;
    sg += row[c];
    if (kind != VRP_KIND_IRP && t >= 1) sf += row[128 + c];
  }
  float *e = d_emb + (size_t)b * N * DB_E + c;
  const float gm = sg / (float)N;
  for (int n = 0; n < N; ++n) e[(size_t)n * DB_E] = gm;
  if (kind != VRP_KIND_IRP && T > 1) e[(size_t)(int)actions[b] * DB_E] += sf;
  for (int t = 1; t < T; ++t) {
    const int last = (int)actions[(size_t)(t - 1) * B + b];
    e[(size_t)last * DB_E] += D[((size_t)t * B + b) * VRP_D + lastoff + c];
  }
}

// ------------------------------------------------------------------ workspace
struct DecBwdWs {
  float *g, *Kb, *Vb, *KPb, *dKb, *dVb, *dKPb;        // node-row buffers
  float *ctx, *cin, *Q, *O, *O2, *Q2, *A;             // step-row tape
  float *dQ2, *dO2, *dO, *dQ, *dctx, *dcin;           // step-row gradients
  float *WT, *Wcp, *dWcp, *tmp;                       // transposed weight, padded Wc, its grad
  void *slab, *csws;
};

static size_t db_slab_bytes(int R, int RB) {
  const int big = R > RB ? R : RB;
  return (size_t)vrp_gemm_tn_workspace_bytes(big, VRP_D, VRP_D);
}

static DecBwdWs carve_dec_bwd(int kind, void *ws, int B, int N, int T, size_t *total) {
  // (offsets are computed on integers: the size query carves from a null base, and pointer
  // arithmetic on a null pointer is undefined behaviour)
  uintptr_t p = (uintptr_t)ws;
  DecBwdWs w;
  const size_t R = (size_t)T * B, RB = (size_t)B * N;
  auto take = [&](size_t floats) { float *q = (float *)p; p += vrp_align_up(floats * 4); return q; };
  w.g = take((size_t)B * DB_E);
  w.Kb = take(RB * VRP_D); w.Vb = take(RB * VRP_D); w.KPb = take(RB * DB_E);
  w.dKb = take(RB * VRP_D); w.dVb = take(RB * VRP_D); w.dKPb = take(RB * DB_E);
  w.ctx = take(R * VRP_D);
  w.cin = (kind == VRP_KIND_IRP) ? take(R * VRP_D) : nullptr;
  w.Q = take(R * VRP_D); w.O = take(R * VRP_D); w.O2 = take(R * VRP_D);
  w.Q2 = take(R * DB_E); w.A = take(R * 8 * N);
  w.dQ2 = take(R * DB_E); w.dO2 = take(R * VRP_D); w.dO = take(R * VRP_D);
  w.dQ = take(R * VRP_D); w.dctx = take(R * VRP_D);
  w.dcin = (kind == VRP_KIND_IRP) ? take(R * VRP_D) : nullptr;
  w.WT = take((size_t)VRP_D * VRP_D);
  w.Wcp = take((size_t)VRP_D * VRP_D);
  w.dWcp = take((size_t)VRP_D * VRP_D);
  w.tmp = take(512);
  w.slab = (void *)p;
  p += vrp_align_up(db_slab_bytes((int)R, (int)RB));
  w.csws = (void *)p;
  p += vrp_align_up((size_t)vrp_colsum_workspace_bytes((int)(R > RB ? R : RB), VRP_D));
  if (total) *total = (size_t)(p - (uintptr_t)ws);
  return w;
}

extern "C" int64_t vrp_decoder_backward_workspace_bytes(int kind, int B, int N, int T) {
  size_t total = 0;
  carve_dec_bwd(kind, nullptr, B, N, T, &total);
  return (int64_t)total;
}

template <typename Kern>
static int db_raise_lds(Kern kern, size_t lds, const char *name) {
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    vrp_set_error("%s: cannot raise dynamic LDS to %zu bytes", name, lds);
    return 1;
  }
  return 0;
}

extern "C" int vrp_decoder_backward(int kind, const vrp_decoder_weights *w,
                                    const vrp_decoder_grads *gr, int B, int N, int T,
                                    const float *emb, const int64_t *actions,
                                    const uint8_t *masks, const float *loads, const float *d_logp,
                                    float *d_emb, float *step_logp, void *workspace,
                                    void *stream) {
  VRP_REQUIRE(w && gr && emb && actions && masks && d_logp && d_emb && workspace,
              "decoder_backward: NULL argument");
  VRP_REQUIRE(kind >= 0 && kind <= 2, "decoder_backward: kind=%d", kind);
  VRP_REQUIRE(B > 0 && N >= 2 && N <= 128 && T > 0, "decoder_backward: bad shape B=%d N=%d T=%d", B,
              N, T);
  VRP_REQUIRE(kind != VRP_KIND_IRP || (loads && w->context_proj_weight && gr->context_proj_weight),
              "decoder_backward: IRP needs loads and _context_proj");
  hipStream_t st = (hipStream_t)stream;
  DecBwdWs s = carve_dec_bwd(kind, workspace, B, N, T, nullptr);
  const int R = T * B, RB = B * N;
  const float *bias = w->in_proj_bias;
  const int npl = N <= 64 ? 1 : 2;

  // ---- forward re-run, everything kept ----------------------------------------------
  hipLaunchKernelGGL(db_graph_mean_kernel, dim3(B), dim3(128), 0, st, emb, N, s.g);
  VRP_CHECK_LAUNCH("db_graph_mean");
  if (int r = vrp_launch_gemm_nt(emb, 128, w->k_proj_weight, 128, bias + 384, nullptr, 0, s.Kb, 384,
                                 RB, 384, 128, 0, st)) return r;
  if (int r = vrp_launch_gemm_nt(emb, 128, w->v_proj_weight, 128, bias + 768, nullptr, 0, s.Vb, 384,
                                 RB, 384, 128, 0, st)) return r;
  if (int r = vrp_launch_gemm_nt(emb, 128, w->kp_weight, 128, nullptr, nullptr, 0, s.KPb, 128, RB,
                                 128, 128, 0, st)) return r;
  float *ctx_in = (kind == VRP_KIND_IRP) ? s.cin : s.ctx;
  hipLaunchKernelGGL(db_ctx_kernel, dim3(R), dim3(128), 0, st, kind, B, N, T, emb, s.g,
                     w->first_node, w->last_node, actions, loads, ctx_in);
  VRP_CHECK_LAUNCH("db_ctx");
  if (kind == VRP_KIND_IRP) {
    hipLaunchKernelGGL(db_pad_copy_kernel, dim3((384 * 384 + 255) / 256), dim3(256), 0, st,
                       w->context_proj_weight, 257, 257, s.Wcp, 384, 384, 384);
    VRP_CHECK_LAUNCH("db_pad_copy");
    if (int r = vrp_launch_gemm_nt(s.cin, 384, s.Wcp, 384, nullptr, nullptr, 0, s.ctx, 384, R, 384,
                                   384, 0, st)) return r;
  }
  if (int r = vrp_launch_gemm_nt(s.ctx, 384, w->q_proj_weight, 384, bias, nullptr, 0, s.Q, 384, R,
                                 384, 384, 0, st)) return r;
  if (db_attn_mfma_applies(N)) {
    DB_ATTN_DISPATCH(db_launch_attn_fwd_mfma, B, N, T, s.Q, s.Kb, s.Vb, masks, s.A, s.O, st);
    VRP_CHECK_LAUNCH("db_attn_fwd_mfma");
  } else {
    const size_t lds = sizeof(float) * ((size_t)N * DBK_LD + N * 48 + 4 * 48 + 4 * 64 * npl);
    if (npl == 1) {
      if (db_raise_lds(db_attn_fwd_kernel<1>, lds, "db_attn_fwd")) return 1;
      hipLaunchKernelGGL(db_attn_fwd_kernel<1>, dim3(B, 8), dim3(256), lds, st, B, N, T, s.Q, s.Kb,
                         s.Vb, masks, s.A, s.O);
    } else {
      if (db_raise_lds(db_attn_fwd_kernel<2>, lds, "db_attn_fwd")) return 1;
      hipLaunchKernelGGL(db_attn_fwd_kernel<2>, dim3(B, 8), dim3(256), lds, st, B, N, T, s.Q, s.Kb,
                         s.Vb, masks, s.A, s.O);
    }
    VRP_CHECK_LAUNCH("db_attn_fwd");
  }
  if (int r = vrp_launch_gemm_nt(s.O, 384, w->out_proj_weight, 384, w->out_proj_bias, nullptr, 0,
                                 s.O2, 384, R, 384, 384, 0, st)) return r;
  if (int r = vrp_launch_gemm_nt(s.O2, 384, w->att_output_weight, 384, nullptr, nullptr, 0, s.Q2,
                                 128, R, 128, 384, 0, st)) return r;

  // ---- backward ------------------------------------------------------------------------
  if (db_attn_mfma_applies(N)) {
    DB_ATTN_DISPATCH(db_launch_logit_mfma, B, N, T, s.Q2, s.KPb, masks, actions, d_logp, s.dQ2, s.dKPb,
                     step_logp, st);
    VRP_CHECK_LAUNCH("db_logit_mfma");
  } else {
    const size_t lds = sizeof(float) * ((size_t)2 * N * DBL_LD + 4 * 128 + 4 * 64 * npl);
    if (npl == 1) {
      if (db_raise_lds(db_logit_kernel<1>, lds, "db_logit")) return 1;
      hipLaunchKernelGGL(db_logit_kernel<1>, dim3(B), dim3(256), lds, st, B, N, T, s.Q2, s.KPb,
                         masks, actions, d_logp, s.dQ2, s.dKPb, step_logp);
    } else {
      if (db_raise_lds(db_logit_kernel<2>, lds, "db_logit")) return 1;
      hipLaunchKernelGGL(db_logit_kernel<2>, dim3(B), dim3(256), lds, st, B, N, T, s.Q2, s.KPb,
                         masks, actions, d_logp, s.dQ2, s.dKPb, step_logp);
    }
    VRP_CHECK_LAUNCH("db_logit");
  }
  // Q2 = O2 Watt^T
  if (int r = vrp_launch_gemm_tn(s.dQ2, 128, s.O2, 384, gr->att_output_weight, R, 128, 384, 0,
                                 s.slab, st)) return r;
  if (int r = vrp_launch_transpose(w->att_output_weight, 128, 384, 384, s.WT, st)) return r;
  if (int r = vrp_launch_gemm_nt(s.dQ2, 128, s.WT, 128, nullptr, nullptr, 0, s.dO2, 384, R, 384, 128,
                                 0, st)) return r;
  // O2 = O Wo^T + bo
  if (int r = vrp_launch_colsum(s.dO2, 384, R, 384, gr->out_proj_bias, 0, s.csws, st)) return r;
  if (int r = vrp_launch_gemm_tn(s.dO2, 384, s.O, 384, gr->out_proj_weight, R, 384, 384, 0, s.slab,
                                 st)) return r;
  if (int r = vrp_launch_transpose(w->out_proj_weight, 384, 384, 384, s.WT, st)) return r;
  if (int r = vrp_launch_gemm_nt(s.dO2, 384, s.WT, 384, nullptr, nullptr, 0, s.dO, 384, R, 384, 384,
                                 0, st)) return r;
  if (db_attn_mfma_applies(N)) {
    DB_ATTN_DISPATCH(db_launch_attn_bwd_mfma, B, N, T, s.Q, s.Kb, s.Vb, s.A, s.dO, s.dQ, s.dKb, s.dVb, st);
    VRP_CHECK_LAUNCH("db_attn_bwd_mfma");
  } else {
    const size_t lds = sizeof(float) * ((size_t)3 * N * DBK_LD + 4 * 64 * npl + 4 * 96);
    if (npl == 1) {
      if (db_raise_lds(db_attn_bwd_kernel<1>, lds, "db_attn_bwd")) return 1;
      hipLaunchKernelGGL(db_attn_bwd_kernel<1>, dim3(B, 8), dim3(256), lds, st, B, N, T, s.Q, s.Kb,
                         s.Vb, s.A, s.dO, s.dQ, s.dKb, s.dVb);
    } else {
      if (db_raise_lds(db_attn_bwd_kernel<2>, lds, "db_attn_bwd")) return 1;
      hipLaunchKernelGGL(db_attn_bwd_kernel<2>, dim3(B, 8), dim3(256), lds, st, B, N, T, s.Q, s.Kb,
                         s.Vb, s.A, s.dO, s.dQ, s.dKb, s.dVb);
    }
    VRP_CHECK_LAUNCH("db_attn_bwd");
  }
  // Q = ctx Wq^T + bq;  K = emb Wk^T + bk;  V = emb Wv^T + bv
  if (int r = vrp_launch_colsum(s.dQ, 384, R, 384, gr->in_proj_bias, 0, s.csws, st)) return r;
  if (int r = vrp_launch_colsum(s.dKb, 384, RB, 384, gr->in_proj_bias + 384, 0, s.csws, st)) return r;
  if (int r = vrp_launch_colsum(s.dVb, 384, RB, 384, gr->in_proj_bias + 768, 0, s.csws, st)) return r;
  if (int r = vrp_launch_gemm_tn(s.dQ, 384, s.ctx, 384, gr->q_proj_weight, R, 384, 384, 0, s.slab,
                                 st)) return r;
  if (int r = vrp_launch_transpose(w->q_proj_weight, 384, 384, 384, s.WT, st)) return r;
  if (int r = vrp_launch_gemm_nt(s.dQ, 384, s.WT, 384, nullptr, nullptr, 0, s.dctx, 384, R, 384, 384,
                                 0, st)) return r;
  const float *D = s.dctx;
  if (kind == VRP_KIND_IRP) {
    // ctx = cin Wc^T  (Wc zero-padded to 384 columns)
    if (int r = vrp_launch_gemm_tn(s.dctx, 384, s.cin, 384, s.dWcp, R, 384, 384, 0, s.slab, st))
      return r;
    hipLaunchKernelGGL(db_pad_copy_kernel, dim3((384 * 257 + 255) / 256), dim3(256), 0, st, s.dWcp,
                       384, 257, gr->context_proj_weight, 257, 257, 384);
    VRP_CHECK_LAUNCH("db_pad_copy");
    if (int r = vrp_launch_transpose(s.Wcp, 384, 384, 384, s.WT, st)) return r;
    if (int r = vrp_launch_gemm_nt(s.dctx, 384, s.WT, 384, nullptr, nullptr, 0, s.dcin, 384, R, 384,
                                   384, 0, st)) return r;
    D = s.dcin;
  }
  // placeholders: the step-0 rows of the context gradient (graph_decoder.py:79-81)
  if (kind != VRP_KIND_IRP) {
    if (int r = vrp_launch_colsum(D + 128, 384, B, 128, gr->first_node, 0, s.csws, st)) return r;
    if (int r = vrp_launch_colsum(D + 256, 384, B, 128, gr->last_node, 0, s.csws, st)) return r;
  } else {
    if (int r = vrp_launch_colsum(D + 128, 384, B, 128, gr->last_node, 0, s.csws, st)) return r;
    if (gr->first_node) {
      hipLaunchKernelGGL(db_pad_copy_kernel, dim3(1), dim3(256), 0, st, D, 0, 0, gr->first_node, 128,
                         128, 1);
      VRP_CHECK_LAUNCH("db_pad_copy");
    }
  }
  hipLaunchKernelGGL(db_scatter_kernel, dim3(B), dim3(128), 0, st, kind, B, N, T, D, actions, d_emb);
  VRP_CHECK_LAUNCH("db_scatter");
  // node-row projections: weight gradients, then d_emb += dK Wk + dV Wv + dKP Wkp
  if (int r = vrp_launch_gemm_tn(s.dKb, 384, emb, 128, gr->k_proj_weight, RB, 384, 128, 0, s.slab,
                                 st)) return r;
  if (int r = vrp_launch_gemm_tn(s.dVb, 384, emb, 128, gr->v_proj_weight, RB, 384, 128, 0, s.slab,
                                 st)) return r;
  if (int r = vrp_launch_gemm_tn(s.dKPb, 128, emb, 128, gr->kp_weight, RB, 128, 128, 0, s.slab, st))
    return r;
  if (int r = vrp_launch_transpose(w->k_proj_weight, 384, 128, 128, s.WT, st)) return r;
  if (int r = vrp_launch_gemm_nt(s.dKb, 384, s.WT, 384, nullptr, d_emb, 128, d_emb, 128, RB, 128, 384,
                                 0, st)) return r;
  if (int r = vrp_launch_transpose(w->v_proj_weight, 384, 128, 128, s.WT, st)) return r;
  if (int r = vrp_launch_gemm_nt(s.dVb, 384, s.WT, 384, nullptr, d_emb, 128, d_emb, 128, RB, 128, 384,
                                 0, st)) return r;
  if (int r = vrp_launch_transpose(w->kp_weight, 128, 128, 128, s.WT, st)) return r;
  if (int r = vrp_launch_gemm_nt(s.dKPb, 128, s.WT, 128, nullptr, d_emb, 128, d_emb, 128, RB, 128, 128,
                                 0, st)) return r;
  return 0;
}

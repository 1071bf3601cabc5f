// Graph encoder forward (N1-N3 of SURVEY.md 8a): node/depot embedding followed by
// L x [MHA(128, 8 heads) + residual + BatchNorm, FF(128->hidden->128) + residual +
// BatchNorm].  agents/graph_encoder.py:41-58, 95-138, 141-154, 183-198.
// Dense projections run on the matrix cores (gemm.hip); the per-graph N x N
// attention, the batch statistics and the normalisation are small HBM-bound kernels.
#include <stdlib.h>
#include "env_device.h"

int vrp_launch_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                       const float *R, int ldr, float *C, int ldc, int M, int N, int K,
                       int relu, hipStream_t stream);
int vrp_launch_gemm_nt_ex(const float *A, int lda, const float *W, int ldw, const float *bias,
                          const float *R, int ldr, const float *norm, float *C, int ldc, int M,
                          int N, int K, int relu, hipStream_t stream);
int vrp_launch_gemm_nt_full(const float *A, int lda, const float *W, int ldw, const float *bias,
                            const float *R, int ldr, const float *norm, const float *gate, float *C,
                            int ldc, int M, int N, int K, int relu, hipStream_t stream);
int vrp_launch_gemm_tn(const float *X, int ldx, const float *Y, int ldy, float *C, int R, int N1,
                       int N2, int accumulate, void *slab_ws, hipStream_t st);
extern "C" int64_t vrp_colsum_workspace_bytes(int R, int N);
int vrp_launch_colsum(const float *Y, int ldy, int R, int N, float *out, int accumulate, void *ws,
                      hipStream_t st);
int vrp_launch_transpose(const float *src, int rows, int cols, int lds, float *dst, hipStream_t st);
int vrp_launch_bn_bwd(const float *dy, const float *z, const float *stats, const float *gamma,
                      int R, float *dz, float *dgamma, float *dbeta, int accumulate, void *ws,
                      hipStream_t st);
int vrp_launch_attention_bwd(const float *qkv, const float *dO, float *dqkv, int B, int N,
                             hipStream_t st, int heads = 8);
extern "C" int64_t vrp_gemm_tn_workspace_bytes(int R, int N1, int N2);
extern "C" int64_t vrp_bn_bwd_workspace_bytes(void);

// ---- embedding: node_embed / depot_embed select (graph_encoder.py:54, 110-132) -----
__global__ __launch_bounds__(256) void embed_kernel(const float *__restrict__ x,
                                                    const uint8_t *__restrict__ depot_mask,
                                                    const float *__restrict__ Wn,
                                                    const float *__restrict__ bn, int node_dim,
                                                    const float *__restrict__ Wd,
                                                    const float *__restrict__ bd, int depot_dim,
                                                    float *__restrict__ out, int rows) {
  const int c = threadIdx.x & 127;
  int r = blockIdx.x * 2 + (threadIdx.x >> 7);
  if (r >= rows) return;
  const float *xr = x + (size_t)r * 3;
  float v;
  if (depot_mask && depot_mask[r]) {
    v = bd[c];
    for (int d = 0; d < depot_dim; ++d) v = fmaf(xr[d], Wd[c * depot_dim + d], v);
  } else {
    v = bn[c];
    for (int d = 0; d < node_dim; ++d) v = fmaf(xr[d], Wn[c * node_dim + d], v);
  }
  out[(size_t)r * VRP_EMB + c] = v;
}

// ---- per-graph multi-head self-attention on the projected QKV --------------------
// One wave per (graph, head); lane = query node (two per lane when N > 64).  K_h and
// V_h (N x 16 each) sit in LDS and are read as broadcasts.  One pass over the keys with a
// running maximum (flash-style); equal to torch's softmax up to fp32 rounding.
// graph_encoder.py:170-172,196.
// D = head width: 16 (eight heads; the A/B arm of the matrix-core kernels) or 32 / 8 (four / sixteen
// heads, graph_encoder.py:170-172 with a non-default num_heads: this kernel is their only path).
template <int D>
__global__ __launch_bounds__(256) void encoder_attention_kernel(const float *__restrict__ qkv,
                                                                float *__restrict__ out, int N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x, h = blockIdx.y * 4 + wave;
  const float scale = D == 16 ? 0.25f : (D == 32 ? 0.17677669529663687f : 0.35355339059327373f);
  float *Ks = smem + (size_t)wave * N * 2 * D;
  float *Vs = Ks + N * D;
  const float *base = qkv + (size_t)b * N * 384;
  for (int idx = lane; idx < N * (D / 4); idx += 64) {
    const int j = idx / (D / 4), q4 = (idx % (D / 4)) * 4;
    *reinterpret_cast<float4 *>(Ks + j * D + q4) =
        *reinterpret_cast<const float4 *>(base + (size_t)j * 384 + 128 + h * D + q4);
    *reinterpret_cast<float4 *>(Vs + j * D + q4) =
        *reinterpret_cast<const float4 *>(base + (size_t)j * 384 + 256 + h * D + q4);
  }
  __syncthreads();
  for (int i = lane; i < N; i += 64) {
    float q[D];
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      float4 t = *reinterpret_cast<const float4 *>(base + (size_t)i * 384 + h * D + d);
      q[d] = t.x * scale; q[d + 1] = t.y * scale; q[d + 2] = t.z * scale; q[d + 3] = t.w * scale;
    }
    // single pass over the keys with a running maximum (scores are computed once); the
    // accumulators are rescaled only when the maximum moves
    float m = -INFINITY, l = 0.f, o[D];
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
    for (int j = 0; j < N; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d) s = fmaf(q[d], Ks[j * D + d], s);
      if (s > m) {
        const float corr = expf(m - s);  // exp(-inf) = 0 on the first key
        l *= corr;
#pragma unroll
        for (int d = 0; d < D; ++d) o[d] *= corr;
        m = s;
      }
      const float p = expf(s - m);
      l += p;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] = fmaf(p, Vs[j * D + d], o[d]);
    }
    const float inv = 1.f / l;
    float *dst = out + ((size_t)b * N + i) * VRP_EMB + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4)
      *reinterpret_cast<float4 *>(dst + d) =
          make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
  }
}
// heads = 128 / D waves per graph, four per workgroup; LDS: K_h | V_h of every wave
static int launch_attention_valu(const float *qkv, float *att, int B, int N, int heads, hipStream_t st) {
  const size_t lds = (size_t)4 * N * 2 * (128 / heads) * sizeof(float);
  VRP_REQUIRE(lds <= 160 * 1024, "encoder attention: N=%d too large for %d heads", N, heads);
#define VRP_ATT_VALU(D_)                                                                          \
  do {                                                                                            \
    static VrpAttrOnce attr_set;                                                                  \
    if (!attr_set.done() && lds > 64 * 1024) {                                                    \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_attention_kernel<D_>),      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { \
        vrp_set_error("encoder_attention: cannot raise dynamic LDS");                             \
        return 1;                                                                                 \
      }                                                                                           \
      attr_set.mark();                                                                            \
    }                                                                                             \
    hipLaunchKernelGGL(encoder_attention_kernel<D_>, dim3(B, heads / 4), dim3(256), lds, st, qkv, att, N); \
  } while (0)
  if (heads == 8) VRP_ATT_VALU(16);
  else if (heads == 4) VRP_ATT_VALU(32);
  else VRP_ATT_VALU(8);
#undef VRP_ATT_VALU
  VRP_CHECK_LAUNCH("encoder_attention");
  return 0;
}
// encoder heads of a weight struct (0 = the reference's default)
static inline int enc_heads(const vrp_encoder_weights *w) { return w->heads ? w->heads : 8; }

// ---- small batches (B*N <= 16384, N <= 64): in_proj + attention of one graph in ONE launch ----
// At ~10 k rows the QKV GEMM and the attention kernel are two launch-latency-bound launches
// with an HBM round trip of the (R,384) projections between them.  Here one workgroup owns
// one graph: the (N,128) input tile goes to LDS, the four waves project it onto the 384
// in_proj columns in eight 48-column blocks on v_mfma_f32_16x16x4_f32 (A = input rows from
// LDS, B = 48 weight rows straight from L2 into registers, K = 128 split over the four
// 16-lane groups), the projections stay in LDS, and each wave then runs two heads of the
// attention exactly like encoder_attention_kernel.  (At large B*N the weight re-read per
// graph -- 196 KB -- would dominate; the GEMM path stays in charge there.)
typedef float f32x4q __attribute__((ext_vector_type(4)));
#define QA_XLD 132
#define QA_QLD 388

// in_proj of one graph: X_s (NT*16 x 128, LDS) -> Q_s (q | k | v, LDS)
template <int NTMAX>
__device__ __forceinline__ void qa_stage_project(const float *X_s, float *Q_s,
                                                 const float *__restrict__ Win,
                                                 const float *__restrict__ bin, int NT, int lane,
                                                 int wave) {
  const int i16 = lane & 15, q = lane >> 4;
  // ---- in_proj: wave w owns column blocks w and w + 4 (48 columns each); the weight rows of
  //      the second block are loaded while the first one is on the matrix cores ----------------
  auto load_w = [&](float (&w)[3][32], int blk) {
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
      const float *wr = Win + (size_t)(blk * 48 + ct * 16 + i16) * VRP_EMB + 32 * q;
#pragma unroll
      for (int s = 0; s < 32; s += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(wr + s);
        w[ct][s] = t.x; w[ct][s + 1] = t.y; w[ct][s + 2] = t.z; w[ct][s + 3] = t.w;
      }
    }
  };
  auto project = [&](const float (&w)[3][32], int blk) {
    f32x4q acc[NTMAX][3];
#pragma unroll
    for (int rt = 0; rt < NTMAX; ++rt)
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) acc[rt][ct] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
      float4 a[NTMAX];
#pragma unroll
      for (int rt = 0; rt < NTMAX; ++rt)
        a[rt] = (rt < NT)
                    ? *reinterpret_cast<const float4 *>(X_s + (rt * 16 + i16) * QA_XLD + 32 * q + s)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int rt = 0; rt < NTMAX; ++rt)
        if (rt < NT) {
#pragma unroll
          for (int ct = 0; ct < 3; ++ct) {
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, w[ct][s], acc[rt][ct], 0, 0, 0);
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].y, w[ct][s + 1], acc[rt][ct], 0, 0, 0);
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].z, w[ct][s + 2], acc[rt][ct], 0, 0, 0);
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].w, w[ct][s + 3], acc[rt][ct], 0, 0, 0);
          }
        }
    }
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
      const int col = blk * 48 + ct * 16 + i16;
      const float bb = bin[col];
#pragma unroll
      for (int rt = 0; rt < NTMAX; ++rt)
        if (rt < NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r)  // D: row = 4*(lane>>4) + r, col = lane & 15
            Q_s[(rt * 16 + 4 * q + r) * QA_QLD + col] = acc[rt][ct][r] + bb;
        }
    }
  };
  float w0[3][32], w1[3][32];
  load_w(w0, wave);       // (issued before the tile barrier would be even better; the tile
  load_w(w1, wave + 4);   //  loads above are short)
  project(w0, wave);
  project(w1, wave + 4);
}

// attention of one graph from Q_s; row i of the result goes to out + i*out_ld (global or LDS)
__device__ __forceinline__ void qa_stage_attention(const float *Q_s, int N, float *out, int out_ld,
                                                   int lane, int wave) {
  // ---- attention (graph_encoder.py:170-172,196): wave w runs heads 2w and 2w+1, lane = query;
  //      for N <= 32 both heads at once, one per 32-lane half -------------------------------------
  const int halves = (N <= 32) ? 1 : 2;
  for (int hh = 0; hh < halves; ++hh) {
    const int h = wave * 2 + ((N <= 32) ? (lane >> 5) : hh);
    const int i = (N <= 32) ? (lane & 31) : lane;
    if (i < N) {
      float qv[16];
#pragma unroll
      for (int d = 0; d < 16; d += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(Q_s + i * QA_QLD + h * 16 + d);
        qv[d] = t.x * 0.25f; qv[d + 1] = t.y * 0.25f; qv[d + 2] = t.z * 0.25f; qv[d + 3] = t.w * 0.25f;
      }
      float m = -INFINITY, l = 0.f, o[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) o[d] = 0.f;
      for (int j = 0; j < N; ++j) {
        float kk[16], vv[16];
#pragma unroll
        for (int d = 0; d < 16; d += 4) {
          const float4 tk = *reinterpret_cast<const float4 *>(Q_s + j * QA_QLD + 128 + h * 16 + d);
          const float4 tv = *reinterpret_cast<const float4 *>(Q_s + j * QA_QLD + 256 + h * 16 + d);
          kk[d] = tk.x; kk[d + 1] = tk.y; kk[d + 2] = tk.z; kk[d + 3] = tk.w;
          vv[d] = tv.x; vv[d + 1] = tv.y; vv[d + 2] = tv.z; vv[d + 3] = tv.w;
        }
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) sc = fmaf(qv[d], kk[d], sc);
        if (sc > m) {
          const float corr = expf(m - sc);
          l *= corr;
#pragma unroll
          for (int d = 0; d < 16; ++d) o[d] *= corr;
          m = sc;
        }
        const float pw = expf(sc - m);
        l += pw;
#pragma unroll
        for (int d = 0; d < 16; ++d) o[d] = fmaf(pw, vv[d], o[d]);
      }
      const float inv = 1.f / l;
      float *dst = out + (size_t)i * out_ld + h * 16;
#pragma unroll
      for (int d = 0; d < 16; d += 4)
        *reinterpret_cast<float4 *>(dst + d) =
            make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
    }
  }
}

template <int NTMAX>
__global__ __launch_bounds__(256, 1) void encoder_qkv_attention_kernel(
    const float *__restrict__ x, const float *__restrict__ Win, const float *__restrict__ bin,
    float *__restrict__ att, int N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *X_s = smem;                          // [NTMAX*16][QA_XLD]
  float *Q_s = smem + NTMAX * 16 * QA_XLD;    // [NTMAX*16][QA_QLD]  q | k | v of every node
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int NT = (N + 15) >> 4;
  for (int idx = tid; idx < NT * 16 * 32; idx += 256) {
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < N) v = *reinterpret_cast<const float4 *>(x + ((size_t)b * N + r) * VRP_EMB + c4);
    *reinterpret_cast<float4 *>(X_s + r * QA_XLD + c4) = v;
  }
  __syncthreads();
  qa_stage_project<NTMAX>(X_s, Q_s, Win, bin, NT, lane, wave);
  __syncthreads();
  qa_stage_attention(Q_s, N, att + (size_t)b * N * VRP_EMB, VRP_EMB, lane, wave);
}

template <int NTMAX>
static int launch_qkv_attention(const float *x, const float *Win, const float *bin, float *att,
                                int B, int N, hipStream_t st) {
  const size_t lds = sizeof(float) * (size_t)NTMAX * 16 * (QA_XLD + QA_QLD);
  static VrpAttrOnce attr_set;
  if (!attr_set.done() && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_qkv_attention_kernel<NTMAX>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("qkv_attention: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  hipLaunchKernelGGL(encoder_qkv_attention_kernel<NTMAX>, dim3(B), dim3(256), lds, st, x, Win, bin,
                     att, N);
  VRP_CHECK_LAUNCH("encoder_qkv_attention");
  return 0;
}

// ---- BatchNorm1d(128) over the flattened (B*N,128) view (graph_encoder.py:141-154) ---
// Batch statistics in two fixed-order stages (no atomics: bitwise reproducible): block k
// writes the fp64 column sums of x and x^2 over its rows (rows k, k+G, k+2G, ... in that
// order) to partial[k][256]; bn_stats_reduce_kernel adds the partials in block order.
#define BN_MAX_BLOCKS 512
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ x, int rows,
                                                       double *__restrict__ partial) {
  __shared__ double sh[2][256];
  const int c = threadIdx.x & 127, par = threadIdx.x >> 7;
  double s = 0.0, ss = 0.0;
  for (int r = blockIdx.x * 2 + par; r < rows; r += gridDim.x * 2) {
    const double v = (double)x[(size_t)r * VRP_EMB + c];
    s += v;
    ss += v * v;
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = ss;
  __syncthreads();
  if (par == 0) {
    partial[(size_t)blockIdx.x * 256 + c] = sh[0][c] + sh[0][c + 128];
    partial[(size_t)blockIdx.x * 256 + 128 + c] = sh[1][c] + sh[1][c + 128];
  }
}
// stats[j] = sum_k partial[k][j]: 1024 threads, quarter q = tid >> 8 of the block partials with
// four independent chains per thread (the loads stay in flight), chains and quarters combined in
// a fixed order -- the result does not depend on scheduling (one thread per column: 34 us)
__global__ __launch_bounds__(1024) void bn_stats_reduce_kernel(const double *__restrict__ partial,
                                                               int blocks,
                                                               double *__restrict__ stats) {
  __shared__ double sh[4][256];
  const int j = threadIdx.x & 255, q = threadIdx.x >> 8;
  const int per = (blocks + 3) / 4;
  const int k1 = min(blocks, (q + 1) * per);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int k = q * per;
  for (; k + 3 < k1; k += 4) {
    a0 += partial[(size_t)(k + 0) * 256 + j];
    a1 += partial[(size_t)(k + 1) * 256 + j];
    a2 += partial[(size_t)(k + 2) * 256 + j];
    a3 += partial[(size_t)(k + 3) * 256 + j];
  }
  for (; k < k1; ++k) a0 += partial[(size_t)k * 256 + j];
  sh[q][j] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q == 0) stats[j] = ((sh[0][j] + sh[1][j]) + sh[2][j]) + sh[3][j];
}

// sums: [256 final | BN_MAX_BLOCKS x 256 partials] doubles
static size_t bn_sums_bytes() { return vrp_align_up((size_t)(1 + BN_MAX_BLOCKS) * 256 * 8); }
static int launch_bn_stats(const float *x, int rows, double *sums, hipStream_t st) {
  int blocks = (rows + 31) / 32;
  if (blocks > BN_MAX_BLOCKS) blocks = BN_MAX_BLOCKS;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks), dim3(256), 0, st, x, rows, sums + 256);
  VRP_CHECK_LAUNCH("bn_stats");
  hipLaunchKernelGGL(bn_stats_reduce_kernel, dim3(1), dim3(1024), 0, st, sums + 256, blocks, sums);
  VRP_CHECK_LAUNCH("bn_stats_reduce");
  return 0;
}

// Eval mode: BatchNorm is a per-channel affine known before the layer runs; all 2L
// triples [mean | weight/sqrt(var+eps) | bias] are produced by ONE launch and applied in
// the epilogue of the GEMM that produces the normalised tensor.
__global__ void bn_eval_norms_kernel(vrp_encoder_weights w, float *__restrict__ norms) {
  const int c = threadIdx.x, l = blockIdx.x >> 1, second = blockIdx.x & 1;
  const vrp_encoder_layer &L = w.layer[l];
  const float *rm = second ? L.bn2_running_mean : L.bn1_running_mean;
  const float *rv = second ? L.bn2_running_var : L.bn1_running_var;
  const float *wt = second ? L.bn2_weight : L.bn1_weight;
  const float *bs = second ? L.bn2_bias : L.bn1_bias;
  float *o = norms + (size_t)blockIdx.x * 384;
  o[c] = rm[c];
  o[128 + c] = wt[c] / sqrtf(rv[c] + 1e-5f);
  o[256 + c] = bs[c];
}

// Train mode: normalise with the batch statistics (biased variance, eps 1e-5) gathered by
// bn_stats_kernel; block 0 also updates the running statistics (momentum 0.1, unbiased
// variance) and num_batches_tracked, like nn.BatchNorm1d.
__global__ __launch_bounds__(256) void bn_train_apply_kernel(
    float *__restrict__ x, size_t n4, const double *__restrict__ stats, int rows,
    const float *__restrict__ weight, const float *__restrict__ bias, float *running_mean,
    float *running_var, int64_t *num_batches_tracked) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 128) {
    const int c = threadIdx.x;
    const double m = stats[c] / rows;
    double v = stats[128 + c] / rows - m * m;
    if (v < 0.0) v = 0.0;
    const float unbiased = (float)(v * ((double)rows / (double)(rows > 1 ? rows - 1 : 1)));
    running_mean[c] = 0.9f * running_mean[c] + 0.1f * (float)m;
    running_var[c] = 0.9f * running_var[c] + 0.1f * unbiased;
    if (c == 0) *num_batches_tracked += 1;
  }
  if (i >= n4) return;
  const int c0 = (int)(i & 31) * 4;
  float4 v = reinterpret_cast<float4 *>(x)[i];
  float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j;
    const double m = stats[c] / rows;
    double var = stats[128 + c] / rows - m * m;
    if (var < 0.0) var = 0.0;
    o[j] = (o[j] - (float)m) * (weight[c] / sqrtf((float)var + 1e-5f)) + bias[c];
  }
  reinterpret_cast<float4 *>(x)[i] = make_float4(o[0], o[1], o[2], o[3]);
}

// ---- fused post-attention block (eval mode) --------------------------------------------
//   y1 = BN1(x + att Wo^T + bo);   y = BN2(y1 + relu(y1 W1^T + b1) W2^T + b2)
// (MultiHeadAttentionLayer.forward agents/graph_encoder.py:196-197 with BatchNorm as the
// per-channel affine it is in eval mode).  One workgroup owns RTW rows; the activations
// (att, x/y1, the 128-wide slices of the hidden layer) never leave LDS, the weights stream
// from L2 straight into registers.  Every stage is the same micro-kernel
//   C (RTW x 128) = A (RTW x 128, LDS) * Wslice^T (128 x 128)
// on v_mfma_f32_32x32x2_f32 (exact fp32): wave w owns output columns 32w..32w+31 for all
// rows, so each weight element is read once per workgroup.  The K = 128 inner dimension is
// split between the two 32-lane halves (half g walks k = 64g + s), a fixed permutation
// applied to both operands.  LDS rows are padded to 132 floats: the per-lane float4
// fragment reads (lane = row) and the column-contiguous result writes are conflict-free.
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define EB_LD 132

// The K = 128 inner dimension is split between the two 32-lane halves g as k = 8 S + 4 g + e
// (S = k-step 0..15, e = element of the lane's float4): the two lanes that read one weight row
// in one instruction take 32 consecutive bytes, so a fragment load touches 32 cache lines,
// not 64 (the vector memory pipe looks lines up one at a time; the weights are re-read by
// every workgroup at every stage).  The same permutation applies to the LDS operand.
#ifndef VRP_EB_KLAYOUT
#define VRP_EB_KLAYOUT 1
#endif
__device__ __forceinline__ constexpr int eb_kg(int g) { return VRP_EB_KLAYOUT ? 4 * g : 64 * g; }
__device__ __forceinline__ constexpr int eb_ks(int s) { return VRP_EB_KLAYOUT ? 2 * s : s; }  // s = 4 S
__device__ __forceinline__ void eb_load_w4(float (&w)[64], const float *wrow, int s) {
  const float4 t = *reinterpret_cast<const float4 *>(wrow + eb_ks(s));
  w[s] = t.x; w[s + 1] = t.y; w[s + 2] = t.z; w[s + 3] = t.w;
}
template <int MI>
__device__ __forceinline__ void eb_load_w(float (&w)[64], const float *wrow) {
#pragma unroll
  for (int s = 0; s < 64; s += 4) eb_load_w4(w, wrow, s);
}
struct EbNoPre { __device__ __forceinline__ void operator()(int) const {} };
// A fragments of k-step S + 1 are read from LDS before the MFMAs of k-step S are issued;
// `pre(s)` runs once per k-step ahead of its MFMAs (the callers request the next stage's
// weight fragment there, one 16-byte load per step, instead of sixteen loads per wave queueing
// up behind the vector memory pipe at a stage boundary).
template <int MI, typename Pre = EbNoPre>
__device__ __forceinline__ void eb_mma(f32x16 (&acc)[MI], const float *abuf, const float (&w)[64],
                                       int lane, Pre pre = Pre()) {
  const int i = lane & 31, g = lane >> 5;
  const float *ap = abuf + i * EB_LD + eb_kg(g);
  float4 a[2][MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const float4 *>(ap + mi * 32 * EB_LD);
#pragma unroll
  for (int s = 0; s < 64; s += 4) {
    const int cur = (s >> 2) & 1;
    if (s + 4 < 64) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        a[cur ^ 1][mi] = *reinterpret_cast<const float4 *>(ap + mi * 32 * EB_LD + eb_ks(s + 4));
    }
    pre(s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi].x, w[s], acc[mi], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi].y, w[s + 1], acc[mi], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi].z, w[s + 2], acc[mi], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi].w, w[s + 3], acc[mi], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int RTW>
__global__ __launch_bounds__(256, 1) void encoder_block_kernel(
    const float *__restrict__ att, const float *__restrict__ x, const float *__restrict__ Wo,
    const float *__restrict__ bo, const float *__restrict__ norm1, const float *__restrict__ W1,
    const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2,
    const float *__restrict__ norm2, float *__restrict__ y, int rows, int hidden) {
  constexpr int MI = RTW / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *bufA = smem;                 // att tile, then the hidden-layer slices
  float *bufB = smem + RTW * EB_LD;   // x tile, then y1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * RTW;
  const int j = lane & 31, g = lane >> 5;
  const int ncol = wave * 32 + j;     // this lane's output column / weight row

  // Two weight-fragment sets alternate (wa: Wo, then the W2 slices; wb: the W1 slices): the
  // loads of stage k+1 are issued before the MFMAs of stage k, so every weight fetch has a
  // whole MFMA stage to land.
  float wa[64], wb[64];
  eb_load_w<MI>(wa, Wo + (size_t)ncol * 128 + eb_kg(g));   // in flight while the tiles land
  for (int idx = tid; idx < RTW * 32; idx += 256) {
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vx = va;
    if (row0 + r < rows) {
      va = *reinterpret_cast<const float4 *>(att + (size_t)(row0 + r) * 128 + c4);
      vx = *reinterpret_cast<const float4 *>(x + (size_t)(row0 + r) * 128 + c4);
    }
    *reinterpret_cast<float4 *>(bufA + r * EB_LD + c4) = va;
    *reinterpret_cast<float4 *>(bufB + r * EB_LD + c4) = vx;
  }
  __syncthreads();

  f32x16 acc[MI], gacc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[mi][r] = 0.f; gacc[mi][r] = 0.f; }

  // ---- y1 = BN1(x + att Wo^T + bo) ------------------------------------------------------
  {
    const float bb = bo[ncol], mean = norm1[ncol], mult = norm1[128 + ncol], beta = norm1[256 + ncol];
    const float *w1c = W1 + (size_t)ncol * 128 + eb_kg(g);   // W1 slice 0, lands during this stage
    eb_mma<MI>(acc, bufA, wa, lane, [&](int s) { eb_load_w4(wb, w1c, s); });
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        float *p = bufB + row * EB_LD + ncol;
        *p = (acc[mi][r] + bb + *p - mean) * mult + beta;   // x -> y1 in place
      }
  }
  __syncthreads();  // y1 complete; nobody reads att (bufA) any more

  // ---- g = sum over 128-wide hidden slices of relu(y1 W1c^T + b1c) W2c^T --------------
  const int nchunk = hidden / 128;
  const float bb2 = b2[ncol], mean2 = norm2[ncol], mult2 = norm2[128 + ncol], beta2 = norm2[256 + ncol];
  for (int c = 0; c < nchunk; ++c) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
    {
      const float bb = b1[c * 128 + ncol];
      const float *w2c = W2 + (size_t)ncol * hidden + c * 128 + eb_kg(g);   // W2[:, slice c]
      eb_mma<MI>(acc, bufB, wb, lane, [&](int s) { eb_load_w4(wa, w2c, s); });   // wb = W1 slice c
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          bufA[row * EB_LD + ncol] = fmaxf(acc[mi][r] + bb, 0.f);
        }
    }
    __syncthreads();  // slice c of the hidden layer is in bufA
    if (c + 1 < nchunk) {
      const float *w1c = W1 + (size_t)((c + 1) * 128 + ncol) * 128 + eb_kg(g);
      eb_mma<MI>(gacc, bufA, wa, lane, [&](int s) { eb_load_w4(wb, w1c, s); });   // wa = W2[:, slice c]
    } else {
      eb_mma<MI>(gacc, bufA, wa, lane);
    }
    __syncthreads();  // everybody done with bufA before the next slice overwrites it
  }

  // ---- y = BN2(y1 + g + b2) ----------------------------------------------------------------
  {
    const float bb = bb2, mean = mean2, mult = mult2, beta = beta2;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (row0 + row < rows) {
          const float v = gacc[mi][r] + bb + bufB[row * EB_LD + ncol];
          y[(size_t)(row0 + row) * 128 + ncol] = (v - mean) * mult + beta;
        }
      }
  }
}

template <int RTW>
static int launch_encoder_block(const float *att, const float *x, const vrp_encoder_layer &L,
                                const float *norm1, const float *norm2, float *y, int rows,
                                int hidden, hipStream_t st) {
  const size_t lds = (size_t)2 * RTW * EB_LD * sizeof(float);
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_block_kernel<RTW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("encoder_block: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  hipLaunchKernelGGL(encoder_block_kernel<RTW>, dim3((rows + RTW - 1) / RTW), dim3(256), lds, st,
                     att, x, L.out_proj_weight, L.out_proj_bias, norm1, L.ff0_weight, L.ff0_bias,
                     L.ff2_weight, L.ff2_bias, norm2, y, rows, hidden);
  VRP_CHECK_LAUNCH("encoder_block");
  return 0;
}

// ---- the same fused block for SMALL row counts (R <= 16384) ------------------------------
// There the 32-row tiling above leaves the chip unbalanced (R = 10240: 320 workgroups on 256
// CUs, the doubly-loaded CUs set the duration).  This variant tiles rows in units of 16 on
// v_mfma_f32_16x16x4_f32 so that the row count per workgroup (16 * RT16) can be chosen to
// give every CU at most one workgroup.  Wave w owns output columns 32w..32w+31 (two 16-wide
// column tiles) for all rows; the K = 128 inner dimension is split over the four 16-lane
// groups (group q walks k = 32q + s), a fixed permutation applied to both operands.
typedef float f32x4v __attribute__((ext_vector_type(4)));
#include "encoder_x3.h"

__device__ __forceinline__ void eb16_load_w(float (&w)[2][32], const float *w0, const float *w1) {
#pragma unroll
  for (int s = 0; s < 32; s += 4) {
    const float4 t0 = *reinterpret_cast<const float4 *>(w0 + s);
    const float4 t1 = *reinterpret_cast<const float4 *>(w1 + s);
    w[0][s] = t0.x; w[0][s + 1] = t0.y; w[0][s + 2] = t0.z; w[0][s + 3] = t0.w;
    w[1][s] = t1.x; w[1][s + 1] = t1.y; w[1][s + 2] = t1.z; w[1][s + 3] = t1.w;
  }
}
template <int RT16>
__device__ __forceinline__ void eb16_mma(f32x4v (&acc)[RT16][2], const float *abuf,
                                         const float (&w)[2][32], int lane) {
  const int i16 = lane & 15, q = lane >> 4;
#pragma unroll
  for (int s = 0; s < 32; s += 4) {
    float4 a[RT16];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
      a[rt] = *reinterpret_cast<const float4 *>(abuf + (rt * 16 + i16) * EB_LD + 32 * q + s);
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, w[ct][s], acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].y, w[ct][s + 1], acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].z, w[ct][s + 2], acc[rt][ct], 0, 0, 0);
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].w, w[ct][s + 3], acc[rt][ct], 0, 0, 0);
      }
  }
}

// The stages of the fused block on a tile already in LDS: bufA = attention output (later the
// hidden-layer slices), bufB = layer input (y1 in place); wa holds the Wo fragment.  Rows
// [0, valid_rows) of the result go to y_tile (global, row stride y_ld = 128; or bufB itself
// with y_ld = EB_LD: every element is read and rewritten by the same thread).
template <int RT16>
__device__ __forceinline__ void eb16_block_stages(
    float *bufA, float *bufB, float (&wa)[2][32], const float *__restrict__ bo,
    const float *__restrict__ norm1, const float *__restrict__ W1, const float *__restrict__ b1,
    const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ norm2,
    float *y_tile, int valid_rows, int hidden, int lane, int wave, int y_ld = 128) {
  const int i16 = lane & 15, q = lane >> 4;
  const int col[2] = {wave * 32 + i16, wave * 32 + 16 + i16};
  float wb[2][32];
  f32x4v acc[RT16][2], gacc[RT16][2];
#pragma unroll
  for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[rt][ct][r] = 0.f; gacc[rt][ct][r] = 0.f; }

  // ---- y1 = BN1(x + att Wo^T + bo) ------------------------------------------------------
  eb16_load_w(wb, W1 + (size_t)col[0] * 128 + 32 * q, W1 + (size_t)col[1] * 128 + 32 * q);
  eb16_mma<RT16>(acc, bufA, wa, lane);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = col[ct];
    const float bb = bo[c], mean = norm1[c], mult = norm1[128 + c], beta = norm1[256 + c];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *p = bufB + (rt * 16 + 4 * q + r) * EB_LD + c;   // D: row = 4*(lane>>4)+r, col = lane&15
        *p = (acc[rt][ct][r] + bb + *p - mean) * mult + beta;  // x -> y1 in place
      }
  }
  __syncthreads();

  // ---- g = sum over 128-wide hidden slices of relu(y1 W1c^T + b1c) W2c^T --------------
  const int nchunk = hidden / 128;
  for (int ch = 0; ch < nchunk; ++ch) {
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rt][ct][r] = 0.f;
    eb16_load_w(wa, W2 + (size_t)col[0] * hidden + ch * 128 + 32 * q,
                W2 + (size_t)col[1] * hidden + ch * 128 + 32 * q);
    eb16_mma<RT16>(acc, bufB, wb, lane);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const float bb = b1[ch * 128 + col[ct]];
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bufA[(rt * 16 + 4 * q + r) * EB_LD + col[ct]] = fmaxf(acc[rt][ct][r] + bb, 0.f);
    }
    __syncthreads();
    if (ch + 1 < nchunk)
      eb16_load_w(wb, W1 + (size_t)((ch + 1) * 128 + col[0]) * 128 + 32 * q,
                  W1 + (size_t)((ch + 1) * 128 + col[1]) * 128 + 32 * q);
    eb16_mma<RT16>(gacc, bufA, wa, lane);
    __syncthreads();
  }

  // ---- y = BN2(y1 + g + b2) ----------------------------------------------------------------
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int c = col[ct];
    const float bb = b2[c], mean = norm2[c], mult = norm2[128 + c], beta = norm2[256 + c];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + 4 * q + r;
        if (row < valid_rows) {
          const float v = gacc[rt][ct][r] + bb + bufB[row * EB_LD + c];
          y_tile[(size_t)row * y_ld + c] = (v - mean) * mult + beta;
        }
      }
  }
}

template <int RT16>
__global__ __launch_bounds__(256, 1) void encoder_block16_kernel(
    const float *__restrict__ att, const float *__restrict__ x, const float *__restrict__ Wo,
    const float *__restrict__ bo, const float *__restrict__ norm1, const float *__restrict__ W1,
    const float *__restrict__ b1, const float *__restrict__ W2, const float *__restrict__ b2,
    const float *__restrict__ norm2, float *__restrict__ y, int rows, int hidden) {
  constexpr int RTW = 16 * RT16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *bufA = smem;                 // att tile, then the hidden-layer slices
  float *bufB = smem + RTW * EB_LD;   // x tile, then y1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * RTW;
  const int i16 = lane & 15, q = lane >> 4;
  const int col[2] = {wave * 32 + i16, wave * 32 + 16 + i16};   // this lane's weight rows / D columns

  float wa[2][32];
  eb16_load_w(wa, Wo + (size_t)col[0] * 128 + 32 * q, Wo + (size_t)col[1] * 128 + 32 * q);
  for (int idx = tid; idx < RTW * 32; idx += 256) {
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vx = va;
    if (row0 + r < rows) {
      va = *reinterpret_cast<const float4 *>(att + (size_t)(row0 + r) * 128 + c4);
      vx = *reinterpret_cast<const float4 *>(x + (size_t)(row0 + r) * 128 + c4);
    }
    *reinterpret_cast<float4 *>(bufA + r * EB_LD + c4) = va;
    *reinterpret_cast<float4 *>(bufB + r * EB_LD + c4) = vx;
  }
  __syncthreads();
  eb16_block_stages<RT16>(bufA, bufB, wa, bo, norm1, W1, b1, W2, b2, norm2,
                          y + (size_t)row0 * 128, rows - row0, hidden, lane, wave);
}

template <int RT16>
static int launch_encoder_block16(const float *att, const float *x, const vrp_encoder_layer &L,
                                  const float *norm1, const float *norm2, float *y, int rows,
                                  int hidden, hipStream_t st) {
  constexpr int RTW = 16 * RT16;
  const size_t lds = (size_t)2 * RTW * EB_LD * sizeof(float);
  static VrpAttrOnce attr_set;
  if (!attr_set.done() && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_block16_kernel<RT16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("encoder_block16: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  hipLaunchKernelGGL(encoder_block16_kernel<RT16>, dim3((rows + RTW - 1) / RTW), dim3(256), lds, st,
                     att, x, L.out_proj_weight, L.out_proj_bias, norm1, L.ff0_weight, L.ff0_bias,
                     L.ff2_weight, L.ff2_bias, norm2, y, rows, hidden);
  VRP_CHECK_LAUNCH("encoder_block16");
  return 0;
}

// One graph of the rollout set-up, by one wave: generate_mask on the fresh episode
// (tsp.py:106-148 via get_state) into mask buffer 0, the network inputs of E3 in registers, the
// node/depot embedding (graph_encoder.py:54,110-132) to `orow` (row stride `ld`: global
// memory or an LDS tile), zeroed episode accumulators.  lane = node for the env part, lane =
// embedding column (x2) after.
// `part` of `parts` waves share the embedding rows of one graph (rows part, part + parts, ...);
// part 0 also commits the mask and the accumulators.
struct SetupLoads {      // everything setup_graph_wave reads from global memory, in registers
  float fx[2][3], wn[2][3], bnv[2], wd[2][2], bdv[2];
  int v0, v1, dep, cur0;
  double load;
};
// Phase 1: every global load of graph b, none of its stores (the compiler keeps loads behind
// stores that may alias them, and a cold round trip is ~2 us) -- callable at the very top of a
// kernel, ahead of LDS set-up work that needs nothing from memory.
__device__ __forceinline__ SetupLoads setup_graph_load(const vrp_env &e, const vrp_encoder_weights &w,
                                                       int b, int lane) {
  const int N = e.N;
  SetupLoads s;
  // ---- features (E3) in registers: lane n holds node n (and n + 64) ------------------------
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = lane + 64 * i;
    const size_t r = (size_t)b * N + (n < N ? n : 0);
    s.fx[i][0] = (float)e.pos[2 * r];
    s.fx[i][1] = (float)e.pos[2 * r + 1];
    s.fx[i][2] = (e.kind == VRP_KIND_IRP) ? (float)e.demand[r] : 0.f;
  }
  // ---- embedding: lane owns columns lane and lane + 64 -------------------------------------
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = lane + 64 * j;
#pragma unroll
    for (int d = 0; d < 3; ++d) s.wn[j][d] = d < w.node_dim ? w.node_embed_weight[c * w.node_dim + d] : 0.f;
    s.bnv[j] = w.node_embed_bias[c];
#pragma unroll
    for (int d = 0; d < 2; ++d)
      s.wd[j][d] = (w.depot_embed_weight && d < w.depot_dim) ? w.depot_embed_weight[c * w.depot_dim + d] : 0.f;
    s.bdv[j] = w.depot_embed_weight ? w.depot_embed_bias[c] : 0.f;
  }
  // VRP_ENV_RESET_ON_ROLLOUT: the episode starts here (tsp.py:150-160,172-174; irp.py:47,184):
  // nothing visited, the vehicle on the depot with a full load
  const bool fresh = (e.flags & VRP_ENV_RESET_ON_ROLLOUT) != 0;
  const uint8_t *vis = e.visited + (size_t)b * N;
  s.v0 = (lane < N) ? (fresh ? 0 : vis[lane]) : 1;
  s.v1 = (lane + 64 < N) ? (fresh ? 0 : vis[lane + 64]) : 1;
  s.load = (e.kind == VRP_KIND_IRP && !fresh) ? e.load[b] : 1.0;
  s.dep = e.depot[b];
  s.cur0 = fresh ? s.dep : e.cur[b];
  return s;
}
// Phase 2: the stores -- generate_mask into mask buffer 0 (same code path as vrp_env_mask),
// zeroed accumulators, the embedding rows.
__device__ __forceinline__ void setup_graph_finish(const vrp_env &e, const vrp_encoder_weights &w,
                                                   int b, int lane, float *orow0, int ld,
                                                   float *__restrict__ acc_loss,
                                                   float *__restrict__ acc_logp, SetupLoads &s,
                                                   int part, int parts) {
  const int N = e.N;
  if (lane == 0 && part == 0) { acc_loss[b] = 0.f; acc_logp[b] = 0.f; }
  const bool fresh = (e.flags & VRP_ENV_RESET_ON_ROLLOUT) != 0;
  int v0 = s.v0, v1 = s.v1;
  const int dep = s.dep, cur0 = s.cur0;
  if (fresh && part == 0 && lane == 0) {
    e.cur[b] = dep;
    if (e.kind == VRP_KIND_IRP) e.load[b] = 1.0;
  }
  if (part == 0) {
    env_fixups_and_mask(e, b, lane, cur0 == dep, v0, v1, s.load, e.mask, dep);
  } else {  // the same flag fix-ups in registers only (tsp.py:141-146, vrp.py:28-31)
    const int n0 = lane, n1 = lane + 64;
    if (cur0 == dep) { if (n0 == dep) v0 = 1; if (n1 == dep) v1 = 1; }
    else if (e.kind != VRP_KIND_TSP) { if (n0 == dep) v0 = 0; if (n1 == dep) v1 = 0; }
    if (__all((n0 >= N || v0) && (n1 >= N || v1))) { if (n0 == dep) v0 = 0; if (n1 == dep) v1 = 0; }
  }
  // depot flag per node: VRP = the mask column just written (QUIRK graph_vrp_agent.py:67),
  // IRP = the is_depot column (graph_irp_agent.py:77-79), TSP = none.  For VRP the mask
  // equals the visited flags (no capacity overlay).
  const unsigned long long d0 = __ballot(e.kind == VRP_KIND_VRP ? (lane < N && v0) : lane == dep);
  const unsigned long long d1 =
      __ballot(e.kind == VRP_KIND_VRP ? (lane + 64 < N && v1) : lane + 64 == dep);
  const bool has_depot = w.depot_embed_weight != nullptr && e.kind != VRP_KIND_TSP;
  for (int n = part; n < N; n += parts) {
    const int src = n & 63;
    float x[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s.fx[0][d]), src));
      const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s.fx[1][d]), src));
      x[d] = n < 64 ? lo : hi;
    }
    const bool isdep = has_depot && (((n < 64 ? d0 : d1) >> src) & 1ull);
    float *orow = orow0 + (size_t)n * ld;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // (static indices: the weights beyond node_dim / depot_dim were loaded as zeros, and
      // fmaf(x, 0, v) = v exactly for the finite x of a feature row -- a loop bounded by the
      // runtime dimension indexes the register arrays dynamically and sends them to scratch)
      float vd = fmaf(x[1], s.wd[j][1], fmaf(x[0], s.wd[j][0], s.bdv[j]));
      float vn = fmaf(x[2], s.wn[j][2], fmaf(x[1], s.wn[j][1], fmaf(x[0], s.wn[j][0], s.bnv[j])));
      orow[lane + 64 * j] = isdep ? vd : vn;
    }
  }
}
__device__ __forceinline__ void setup_graph_wave(const vrp_env &e, const vrp_encoder_weights &w,
                                                 int b, int lane, float *orow0, int ld,
                                                 float *__restrict__ acc_loss,
                                                 float *__restrict__ acc_logp, int part = 0,
                                                 int parts = 1) {
  SetupLoads s = setup_graph_load(e, w, b, lane);
  setup_graph_finish(e, w, b, lane, orow0, ld, acc_loss, acc_logp, s, part, parts);
}

// ---- whole encoder in ONE launch (eval mode, small batches) -------------------------------
// Without batch statistics a graph never meets another graph in the encoder
// (graph_encoder.py:41-58,95-138,183-198), so a workgroup can take G whole graphs (G*N <= 48
// rows = three 16-row MFMA tiles; N = 20: two graphs = 40 rows) through ALL layers with the
// activations in LDS: in_proj -> per-graph attention -> out-proj + BN1 + FF + BN2, three
// times, one read of the embedded inputs and one write of the result.  Replaces 2 launches
// per layer (each ~2-3 us of dependent-launch latency at this size) and the global round trips
// between them.  EIGHT waves per workgroup, two per SIMD: a wave owns 16 output columns of the
// block stages, one 48-column block of in_proj and one attention head, so that while one wave
// of a SIMD writes its results to LDS or waits at a barrier the other one keeps the matrix
// pipe busy (same arithmetic and summation order as the 4-wave kernels above).
// The A fragments of k-step s + 1 are read from LDS before the MFMAs of k-step s are issued
// (the scheduling fences keep the compiler from sinking the reads next to their use, where
// every group of MFMAs would start with an LDS round trip).  Per accumulator the k order is
// unchanged.
// `pre(s)` runs once per k-step ahead of its MFMAs: the callers use it to request the NEXT
// stage's weight fragments one 16-byte load at a time.  (Eight waves issuing all their loads
// at a stage boundary queue up behind the vector memory pipe -- a fragment load touches 64
// cache lines -- and no wave issues an MFMA until its own loads are accepted.)
// How the 128-long inner dimension of the 8-wave kernels is spread over the four 16-lane
// groups q and the eight k-steps S (element e of the lane's float4):
//   KL 0: k = 32 q + 4 S + e          a weight-fragment load touches 64 cache lines
//   KL 2: k = 64 (S / 4) + 32 (q / 2) + 8 (S % 4) + 4 (q % 2) + e
//         32 lines per load (two lanes of a row share a 32-byte sector pair), and the lane
//         pairs (l, l + 32) that one ds_read_b128 pass serves are 32 banks apart
// The vector memory pipe looks lines up one by one, so for kernels that re-read their weights
// every stage (the stack kernel) the line count of a fragment load is what the loads cost:
// 172 -> 156 us at 512 x 20.  The large-batch kernel below loads its fragments once per
// workgroup and keeps KL 0 (408 vs 415 us per layer at 8192 x 40).
#ifndef VRP_KLAYOUT
#define VRP_KLAYOUT 2
#endif
template <int KL = VRP_KLAYOUT>
__device__ __forceinline__ constexpr int kq8(int q) {
  return KL == 0 ? 32 * q : KL == 1 ? 4 * q : 32 * (q >> 1) + 4 * (q & 1);
}
template <int KL = VRP_KLAYOUT>
__device__ __forceinline__ constexpr int ks8(int s) {   // s = 4 S: register index of the k-step
  return KL == 0 ? s : KL == 1 ? 4 * s : 64 * (s >> 4) + 2 * (s & 15);
}
struct NoPre { __device__ __forceinline__ void operator()(int) const {} };
__device__ __forceinline__ void load_w4(float *w, const float *w0, int s) {
  const float4 t = *reinterpret_cast<const float4 *>(w0 + ks8(s));
  w[s] = t.x; w[s + 1] = t.y; w[s + 2] = t.z; w[s + 3] = t.w;
}
template <int RT16, typename Pre = NoPre>
__device__ __forceinline__ void eb8_mma(f32x4v (&acc)[RT16], const float *abuf, const float (&w)[32],
                                        int lane, Pre pre = Pre()) {
  const int i16 = lane & 15, q = lane >> 4;
  const float *ap = abuf + i16 * EB_LD + kq8(q);
  float4 a[2][RT16];
#pragma unroll
  for (int rt = 0; rt < RT16; ++rt) a[0][rt] = *reinterpret_cast<const float4 *>(ap + rt * 16 * EB_LD);
#pragma unroll
  for (int s = 0; s < 32; s += 4) {
    const int cur = (s >> 2) & 1;
    if (s + 4 < 32) {
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt)
        a[cur ^ 1][rt] = *reinterpret_cast<const float4 *>(ap + rt * 16 * EB_LD + ks8(s + 4));
    }
    pre(s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][rt].x, w[s], acc[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][rt].y, w[s + 1], acc[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][rt].z, w[s + 2], acc[rt], 0, 0, 0);
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][rt].w, w[s + 3], acc[rt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Developer aid (make EXTRA=-DVRP_STACK_TRACE): shader-clock timestamps of every wave at the
// phase boundaries of the stack kernel, printed for two workgroups after each launch.
#ifdef VRP_STACK_TRACE
#define ST_SLOTS 96
__device__ unsigned long long g_stack_trace[512 * 8 * ST_SLOTS];
#define ST_MARK(i)                                                                      \
  if (lane == 0) g_stack_trace[((size_t)blockIdx.x * 8 + wave) * ST_SLOTS + (i)] =      \
      ((i) == 0 || (i) == ST_SLOTS - 1) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime()
#else
#define ST_MARK(i)
#endif

// one half of the fragments (k-steps [16 half, 16 half + 16) of every column tile)
__device__ __forceinline__ void qa8_load_w_half(float (&w)[3][32], const float *__restrict__ Win,
                                                int lane, int wave, int half) {
  const int i16 = lane & 15, q = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    const float *wr = Win + (size_t)(wave * 48 + ct * 16 + i16) * VRP_EMB + kq8(q);
#pragma unroll
    for (int s = 16 * half; s < 16 * half + 16; s += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(wr + ks8(s));
      w[ct][s] = t.x; w[ct][s + 1] = t.y; w[ct][s + 2] = t.z; w[ct][s + 3] = t.w;
    }
  }
}
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `last_hook` runs ahead of the MFMAs of the last stage (the large-batch kernel requests its
// next tile there).
template <int RT16, typename Last = NoHook, bool NEXT = true>
__device__ __forceinline__ void eb8_block_stages(
    float *bufA, float *bufA2, float *bufB, float (&wa)[32], const float *__restrict__ bo,
    const float *__restrict__ norm1, const float *__restrict__ W1, const float *__restrict__ b1,
    const float *__restrict__ W2, const float *__restrict__ b2, const float *__restrict__ norm2,
    float *y_tile, int valid_rows, int hidden, int lane, int wave, int y_ld,
    float (&win)[3][32], const float *Win_next, int st_base = 0, Last last_hook = Last()) {
  const int i16 = lane & 15, q = lane >> 4;
  const int c = wave * 16 + i16;  // this lane's weight row / D column
  float wb[32];
  f32x4v acc[RT16], gacc[RT16];
#pragma unroll
  for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[rt][r] = 0.f; gacc[rt][r] = 0.f; }
  // ---- y1 = BN1(x + att Wo^T + bo) ------------------------------------------------------
  const float bb_o = bo[c], mean1 = norm1[c], mult1 = norm1[128 + c], beta1 = norm1[256 + c];
  {
    const float *w1c = W1 + (size_t)c * 128 + kq8(q);   // W1 slice 0 for the first way up
    eb8_mma<RT16>(acc, bufA, wa, lane, [&](int s) { load_w4(wb, w1c, s); });
  }
  {
    const float bb = bb_o, mean = mean1, mult = mult1, beta = beta1;
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *p = bufB + (rt * 16 + 4 * q + r) * EB_LD + c;   // D: row = 4*(lane>>4)+r, col = lane&15
        *p = (acc[rt][r] + bb + *p - mean) * mult + beta;      // x -> y1 in place
      }
  }
  ST_MARK(st_base + 5);
  __syncthreads();
  ST_MARK(st_base + 6);
  // ---- g = sum over 128-wide hidden slices of relu(y1 W1c^T + b1c) W2c^T --------------
  // Slice ch goes up (y1 -> hidden slice in LDS) and comes down (slice -> g, in registers).
  // The slices alternate between two LDS buffers, so the way down of slice ch and the way up
  // of slice ch + 1 run back to back without a barrier between them: ONE barrier per slice,
  // and a wave that finishes a stage early finds its next 96 MFMAs ready instead of a barrier
  // (the SIMD's matrix pipe serves its two waves oldest first: whatever one wave does not
  // issue, the other one fills in).
  const int nchunk = hidden / 128;
  auto slice_up = [&](int ch, float *hb) {
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[rt][r] = 0.f;
    const float bb = b1[ch * 128 + c];
    const float *w2c = W2 + (size_t)c * hidden + ch * 128 + kq8(q);   // needed on the way down
    eb8_mma<RT16>(acc, bufB, wb, lane, [&](int s) { load_w4(wa, w2c, s); });
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        hb[(rt * 16 + 4 * q + r) * EB_LD + c] = fmaxf(acc[rt][r] + bb, 0.f);
  };
  slice_up(0, bufA);
  ST_MARK(st_base + 7);
  __syncthreads();
  ST_MARK(st_base + 8);
  for (int ch = 0; ch + 1 < nchunk; ++ch) {
    float *hcur = (ch & 1) ? bufA2 : bufA, *hnext = (ch & 1) ? bufA : bufA2;
    const float *w1c = W1 + (size_t)((ch + 1) * 128 + c) * 128 + kq8(q);
    eb8_mma<RT16>(gacc, hcur, wa, lane, [&](int s) { load_w4(wb, w1c, s); });
    ST_MARK(st_base + 9 + 3 * ch);
    slice_up(ch + 1, hnext);
    ST_MARK(st_base + 10 + 3 * ch);
    __syncthreads();
    ST_MARK(st_base + 11 + 3 * ch);
  }
  // way down of the last slice: the first half (k-steps 0..15) of the next layer's in_proj
  // fragments is fetched behind it; the other half follows at the start of that layer, behind
  // the first half's MFMAs (all 96 registers at once do not fit beside this stage)
  const float bb_2 = b2[c], mean2 = norm2[c], mult2 = norm2[128 + c], beta2 = norm2[256 + c];
  last_hook();
  {
    const float *wn = Win_next ? Win_next + (size_t)(wave * 48 + i16) * VRP_EMB + kq8(q) : nullptr;
    eb8_mma<RT16>(gacc, ((nchunk - 1) & 1) ? bufA2 : bufA, wa, lane, [&](int s) {
      // twelve loads over the first six k-steps: column tile ct = s / 8, k-steps (s % 8) * 2 ..
      if (NEXT && wn && s < 24) {
        const int ct = s / 8, k0 = (s % 8) * 2;
        load_w4(win[ct], wn + (size_t)ct * 16 * VRP_EMB, k0);
        load_w4(win[ct], wn + (size_t)ct * 16 * VRP_EMB, k0 + 4);
      }
    });
  }
  ST_MARK(st_base + 21);
  // (no barrier: the rest touches only this thread's own elements of bufB)
  // ---- y = BN2(y1 + g + b2) ----------------------------------------------------------------
  {
    const float bb = bb_2, mean = mean2, mult = mult2, beta = beta2;
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + 4 * q + r;
        if (row < valid_rows) {
          const float v = gacc[rt][r] + bb + bufB[row * EB_LD + c];
          y_tile[(size_t)row * y_ld + c] = (v - mean) * mult + beta;
        }
      }
  }
}

// in_proj of the tile: wave w owns column block w (48 columns = q, k or v of two heads)
// ALIAS: Q_s overlays X_s (large tiles): every wave finishes reading X_s (workgroup barrier)
// before any projection result is written.
template <int KL = VRP_KLAYOUT>
__device__ __forceinline__ void qa8_load_w(float (&w)[3][32], const float *__restrict__ Win,
                                           int lane, int wave) {
  const int i16 = lane & 15, q = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    const float *wr = Win + (size_t)(wave * 48 + ct * 16 + i16) * VRP_EMB + kq8<KL>(q);
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(wr + ks8<KL>(s));
      w[ct][s] = t.x; w[ct][s + 1] = t.y; w[ct][s + 2] = t.z; w[ct][s + 3] = t.w;
    }
  }
}
template <int NTMAX, bool ALIAS = false, typename Pre = NoPre, int KL = VRP_KLAYOUT>
__device__ __forceinline__ void qa8_project(const float *X_s, float *Q_s, const float (&w)[3][32],
                                            const float *__restrict__ bin, int lane, int wave,
                                            Pre pre = Pre(), int vrows = 1 << 30) {
  const int i16 = lane & 15, q = lane >> 4;
  f32x4q acc[NTMAX][3];
#pragma unroll
  for (int rt = 0; rt < NTMAX; ++rt)
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) acc[rt][ct] = {0.f, 0.f, 0.f, 0.f};
  float bb[3];  // fetched ahead of the MFMAs, not behind them
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) bb[ct] = bin[wave * 48 + ct * 16 + i16];
  const float *xp = X_s + i16 * QA_XLD + kq8<KL>(q);
  float4 a[2][NTMAX];
#pragma unroll
  for (int rt = 0; rt < NTMAX; ++rt) a[0][rt] = *reinterpret_cast<const float4 *>(xp + rt * 16 * QA_XLD);
#pragma unroll
  for (int s = 0; s < 32; s += 4) {
    const int cur = (s >> 2) & 1;
    if (s + 4 < 32) {
#pragma unroll
      for (int rt = 0; rt < NTMAX; ++rt)
        a[cur ^ 1][rt] = *reinterpret_cast<const float4 *>(xp + rt * 16 * QA_XLD + ks8<KL>(s + 4));
    }
    pre(s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
      for (int rt = 0; rt < NTMAX; ++rt) {
        const float av = k4 == 0 ? a[cur][rt].x : k4 == 1 ? a[cur][rt].y : k4 == 2 ? a[cur][rt].z : a[cur][rt].w;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w[ct][s + k4], acc[rt][ct], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (ALIAS) __syncthreads();
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    const int col = wave * 48 + ct * 16 + i16;
#pragma unroll
    for (int rt = 0; rt < NTMAX; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r)  // D: row = 4*(lane>>4) + r, col = lane & 15
        if (rt * 16 + 4 * q + r < vrows)   // (rows beyond the buffer: the one-graph large-N kernel)
          Q_s[(rt * 16 + 4 * q + r) * QA_QLD + col] = acc[rt][ct][r] + bb[ct];
  }
}

// Attention of the tile's graphs on the matrix cores (graph_encoder.py:170-172,196): wave = head,
// one graph at a time.  With q|k|v in LDS (Q_s) both products run as v_mfma_f32_16x16x4_f32:
//   S^T tile (tn, tm) = K(tn) Q(tm)^T: operands are one ds_read_b128 per 16-row tile (k = 4q + s,
//                the head's 16 columns split over the four lane groups); D[n][m] puts query row
//                m on the lane and its key columns n = 16tn + 4q + reg in registers -- so the
//                softmax over n is in-lane plus two cross-group steps, and
//   O^T (16 x m) = V^T P^T: P^T is ALREADY the B operand (k = key n = 16tn + 4q + reg, the
//                accumulator layout of the first product), V comes from LDS one float per MFMA.
// The result D[d][m] gives a lane four consecutive head columns of its query row: one 16-byte
// store.  NT = row tiles per graph (N <= 16 NT).
struct AttSinkF32 {   // attention output rows as fp32: one 16-byte store per lane
  float *out; int ld;
  __device__ __forceinline__ void operator()(int row, int col, float a, float b, float c, float d) const {
    *reinterpret_cast<float4 *>(out + (size_t)row * ld + col) = make_float4(a, b, c, d);
  }
};
struct AttSinkX3 {    // ... as three bf16 planes of an LDS tile (encoder_x3.h)
  __bf16 *tile; int plane_elems;
  __device__ __forceinline__ void operator()(int row, int col, float a, float b, float c, float d) const {
    x3_store4(tile, plane_elems, row, col, a, b, c, d);
  }
};
// GP graphs at a time (GP = 2 for the stack kernels' small tiles: a wave's life in this stage is a
// chain LDS read -> score MFMAs -> softmax -> value MFMAs -> store per graph; two graphs side by
// side give every link of the chain an independent twin to overlap with).  Per graph the
// arithmetic and its order are unchanged.
template <int NT, int GP, typename Sink>
__device__ __forceinline__ void qa8_attention_group(const float *Q_s, int N, int g0, int max_row,
                                                    Sink sink, int lane, int wave) {
  const int h = wave, i16 = lane & 15, q = lane >> 4;
  float4 qf[GP][NT], kf[GP][NT];
  float vv[GP][NT][4];  // V[16tn + 4q + r4][d = i16]: the A operand of the second product
#pragma unroll
  for (int u = 0; u < GP; ++u) {
    const int r0 = (g0 + u) * N;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int row = min(r0 + 16 * t + i16, max_row);
      const float4 a = *reinterpret_cast<const float4 *>(Q_s + row * QA_QLD + h * 16 + 4 * q);
      qf[u][t] = make_float4(a.x * 0.25f, a.y * 0.25f, a.z * 0.25f, a.w * 0.25f);  // 1/sqrt(16)
      kf[u][t] = *reinterpret_cast<const float4 *>(Q_s + row * QA_QLD + 128 + h * 16 + 4 * q);
    }
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        vv[u][tn][r4] = Q_s[min(r0 + 16 * tn + 4 * q + r4, max_row) * QA_QLD + 256 + h * 16 + i16];
  }
  f32x4v st[GP][NT][NT];  // st[tn][tm][r4] = S[m = 16tm + i16][n = 16tn + 4q + r4]
#pragma unroll
  for (int tn = 0; tn < NT; ++tn)
#pragma unroll
    for (int tm = 0; tm < NT; ++tm)
#pragma unroll
      for (int u = 0; u < GP; ++u) {
        f32x4v d = {0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][tn].x, qf[u][tm].x, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][tn].y, qf[u][tm].y, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][tn].z, qf[u][tm].z, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u][tn].w, qf[u][tm].w, d, 0, 0, 0);
        st[u][tn][tm] = d;
      }
#pragma unroll
  for (int tm = 0; tm < NT; ++tm) {
    float sum[GP];
    f32x4v o[GP];
#pragma unroll
    for (int u = 0; u < GP; ++u) {
      // softmax over the keys of query row m = 16tm + i16 (graph_encoder.py:172: no mask)
      float mx = -INFINITY;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          if (16 * tn + 4 * q + r4 < N) mx = fmaxf(mx, st[u][tn][tm][r4]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sm = 0.f;
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float pw = (16 * tn + 4 * q + r4 < N) ? exp_nonpos(st[u][tn][tm][r4] - mx) : 0.f;
          st[u][tn][tm][r4] = pw;
          sm += pw;
        }
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      sum[u] = sm;
    }
#pragma unroll
    for (int u = 0; u < GP; ++u) {
      f32x4v oo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tn = 0; tn < NT; ++tn)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          oo = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[u][tn][r4], st[u][tn][tm][r4], oo, 0, 0, 0);
      o[u] = oo;
    }
    // D[d = 4q + r4][m = i16]
    const int m = 16 * tm + i16;
    if (m < N) {
#pragma unroll
      for (int u = 0; u < GP; ++u) {
        const float inv = 1.f / sum[u];
        sink((g0 + u) * N + m, h * 16 + 4 * q, o[u][0] * inv, o[u][1] * inv, o[u][2] * inv, o[u][3] * inv);
      }
    }
  }
}
template <int NT, typename Sink>
__device__ __forceinline__ void qa8_stage_attention_mfma_to(const float *Q_s, int N, int graphs,
                                                            int max_row, Sink sink, int lane, int wave) {
  for (int g = 0; g < graphs; ++g) qa8_attention_group<NT, 1>(Q_s, N, g, max_row, sink, lane, wave);
}
// the same, two graphs side by side while there are two left
template <int NT, typename Sink>
__device__ __forceinline__ void qa8_stage_attention_mfma_pairs(const float *Q_s, int N, int graphs,
                                                               int max_row, Sink sink, int lane, int wave) {
  int g = 0;
  for (; g + 1 < graphs; g += 2) qa8_attention_group<NT, 2>(Q_s, N, g, max_row, sink, lane, wave);
  if (g < graphs) qa8_attention_group<NT, 1>(Q_s, N, g, max_row, sink, lane, wave);
}
// ---- the same attention, one QUERY ROW PER LANE, no matrix cores (round 6) ---------------------
// For the stack kernels' tiles (rows = graphs * N <= 48 <= 64 lanes; wave = head): a (graph, head)
// pair is 2 * 16 * N^2 multiply-adds per product -- at N = 20 the two products of the CU's
// sixteen pairs are 400 packed-FMA wave instructions per SIMD, while the MFMA form above pads 20
// keys and 20 queries to 32 x 32, runs on the fp32 MFMA (1/16 of the bf16 rate), and strings LDS
// reads -> score MFMAs -> two cross-lane maxima -> exponentials -> two cross-lane sums -> value
// MFMAs into one dependency chain per graph: 10-12 k cycles per layer in the round-5 trace, a
// fifth of the kernel.  Here a lane keeps its query, its running maximum / sum and its 16 output
// columns in registers and walks its graph's keys four at a time (online softmax: the rescale of
// a block of four keys is one exponential and eight packed multiplies); the K / V rows are
// 16-byte LDS reads that every lane of a graph makes at the same address (broadcast).  No
// cross-lane step, no padding work, nothing on the matrix pipe.  The exponent is exp2((s - m) *
// log2 e): the rounding of that product is a relative error of 6e-8 * (1 + |s - m|) on a weight
// e^-(m - s), i.e. at most 3e-8 of the largest weight.
template <typename Sink>
__device__ __forceinline__ void qa8_stage_attention_valu(const float *Q_s, int N, int rows,
                                                         Sink sink, int lane, int wave) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int h = wave;
  const int r = min(lane, rows - 1);          // idle lanes shadow the last row (nothing stored)
  const int kb = (r / N) * N;                 // first row of this lane's graph
  const float *qrow = Q_s + r * QA_QLD + h * 16;
  f32x2 q2[8], o2[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 t = *reinterpret_cast<const float4 *>(qrow + 4 * j);
    q2[2 * j] = f32x2{t.x * 0.25f, t.y * 0.25f};           // 1/sqrt(16), exact
    q2[2 * j + 1] = f32x2{t.z * 0.25f, t.w * 0.25f};
    o2[2 * j] = f32x2{0.f, 0.f};
    o2[2 * j + 1] = f32x2{0.f, 0.f};
  }
  const float l2e = 1.44269504088896341f;
  float m = -INFINITY, l = 0.f;
  const float *kbase = Q_s + kb * QA_QLD + 128 + h * 16;
  for (int n0 = 0; n0 < N; n0 += 4) {
    float sc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float *kr = kbase + min(n0 + i, N - 1) * QA_QLD;
      f32x2 a = {0.f, 0.f}, b = {0.f, 0.f};                // two chains of four
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 t = *reinterpret_cast<const float4 *>(kr + 4 * j);
        a = __builtin_elementwise_fma(q2[2 * j], f32x2{t.x, t.y}, a);
        b = __builtin_elementwise_fma(q2[2 * j + 1], f32x2{t.z, t.w}, b);
      }
      a += b;
      sc[i] = (n0 + i < N) ? a.x + a.y : -INFINITY;
    }
    const float mn = fmaxf(fmaxf(m, fmaxf(sc[0], sc[1])), fmaxf(sc[2], sc[3]));
    const float corr = __builtin_amdgcn_exp2f((m - mn) * l2e);   // first block: exp2(-inf) = 0
    float pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pw[i] = __builtin_amdgcn_exp2f((sc[i] - mn) * l2e);
    l = fmaf(l, corr, (pw[0] + pw[1]) + (pw[2] + pw[3]));
    // (a lane whose maximum did not move has corr = exp2(0) = 1 exactly: when that holds for the
    // whole wave -- usual after the first block or two -- the eight packed multiplies are skipped,
    // bit for bit the same result)
    if (__builtin_amdgcn_ballot_w64(mn != m) != 0ull) {
      const f32x2 c2 = {corr, corr};
#pragma unroll
      for (int j = 0; j < 8; ++j) o2[j] *= c2;
    }
    m = mn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float *vr = kbase + 128 + min(n0 + i, N - 1) * QA_QLD;
      const f32x2 p2 = {pw[i], pw[i]};                     // 0 for a key past the graph's last
#ifndef VRP_ATT_NOFENCE
      // (two keys' value rows in flight, not four: the kernel around this stage keeps two weight
      // fragments -- 96 registers -- alive across it)
      if (i == 2) __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 t = *reinterpret_cast<const float4 *>(vr + 4 * j);
        o2[2 * j] = __builtin_elementwise_fma(p2, f32x2{t.x, t.y}, o2[2 * j]);
        o2[2 * j + 1] = __builtin_elementwise_fma(p2, f32x2{t.z, t.w}, o2[2 * j + 1]);
      }
    }
  }
  if (lane < rows) {
    const float inv = 1.f / l;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      sink(lane, h * 16 + 4 * j, o2[2 * j].x * inv, o2[2 * j].y * inv, o2[2 * j + 1].x * inv,
           o2[2 * j + 1].y * inv);
  }
}
template <int NT>
__device__ __forceinline__ void qa8_stage_attention_mfma(const float *Q_s, int N, int graphs,
                                                         int max_row, float *out, int lane,
                                                         int wave, int out_ld) {
  qa8_stage_attention_mfma_to<NT>(Q_s, N, graphs, max_row, AttSinkF32{out, out_ld}, lane, wave);
}

// ---- out-proj + BN1 + FF + BN2 for LARGE row counts, eight waves on 16x16x4 MFMAs ---------
// The stack kernel's block stages (a wave owns 16 output columns; hidden slices alternate
// between two LDS buffers, one barrier per slice) on 80-row tiles, persistent workgroups, one
// per CU (three 80-row buffers = 127 KB of LDS).  Nothing of a tile's traffic is exposed:
//   * the attention rows of the NEXT tile are requested ahead of the last stage's MFMAs and
//     written behind them into the hidden buffer that stage does not read;
//   * the residual x is not staged at all: a lane fetches the 4 x RT16 elements of its
//     accumulator layout straight from global memory ahead of the out-proj MFMAs;
//   * y leaves in the BN2 epilogue.
template <int RT16>
__global__ __launch_bounds__(512) void encoder_block8_kernel(
    const float *__restrict__ att, const float *__restrict__ x, const float *__restrict__ Wo_,
    const float *__restrict__ bo_, const float *__restrict__ norm1_, const float *__restrict__ W1_,
    const float *__restrict__ b1_, const float *__restrict__ W2_, const float *__restrict__ b2_,
    const float *__restrict__ norm2_, float *__restrict__ y, int rows, int hidden, int ntiles) {
  constexpr int RTW = 16 * RT16, PF = RTW * 32 / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *hb0 = smem;                   // attention tile / hidden slices (roles alternate)
  float *hb1 = hb0 + RTW * EB_LD;
  float *bufB = hb1 + RTW * EB_LD;     // y1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int c = wave * 16 + i16;
  const int nchunk = hidden / 128;
  float4 pa[PF];
  auto fetch_att = [&](int tile) {
    const int row0 = tile * RTW;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      pa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < rows) pa[u] = *reinterpret_cast<const float4 *>(att + (size_t)(row0 + r) * 128 + c4);
    }
  };
  auto store_att = [&](float *dst) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      *reinterpret_cast<float4 *>(dst + r * EB_LD + c4) = pa[u];
    }
  };
  int tile = blockIdx.x;
  float *abuf = hb0, *other = hb1;     // abuf: this tile's attention rows
  if (tile < ntiles) { fetch_att(tile); store_att(abuf); }
  float wa[32], wb[32];   // wb enters a tile holding the Wo fragment
  {
    const float *wo = Wo_ + (size_t)c * 128 + kq8(q);
#pragma unroll
    for (int s = 0; s < 32; s += 4) load_w4(wb, wo, s);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    // (the weights do not change from tile to tile; an opaque zero keeps the compiler from
    // hoisting their loads out of this loop into registers it does not have)
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    const float *Wo = Wo_ + zero, *bo = bo_ + zero, *norm1 = norm1_ + zero, *W1 = W1_ + zero,
                *b1 = b1_ + zero, *W2 = W2_ + zero, *b2 = b2_ + zero, *norm2 = norm2_ + zero;
    const int row0 = tile * RTW, valid = rows - row0;
#pragma unroll
    for (int s = 0; s < 32; ++s) wa[s] = wb[s];
    // residual rows of this lane's accumulator elements (row = 16 rt + 4 q + r, column c)
    float xr[RT16][4];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + 4 * q + r;
        xr[rt][r] = row < valid ? x[(size_t)(row0 + row) * 128 + c] : 0.f;
      }
    f32x4v acc[RT16], gacc[RT16];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[rt][r] = 0.f; gacc[rt][r] = 0.f; }
    // ---- y1 = BN1(x + att Wo^T + bo) -> bufB ----------------------------------------------
    {
      const float bb = bo[c], mean = norm1[c], mult = norm1[128 + c], beta = norm1[256 + c];
      const float *w1c = W1 + (size_t)c * 128 + kq8(q);
      eb8_mma<RT16>(acc, abuf, wa, lane, [&](int s) { load_w4(wb, w1c, s); });
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bufB[(rt * 16 + 4 * q + r) * EB_LD + c] = (acc[rt][r] + bb + xr[rt][r] - mean) * mult + beta;
    }
    __syncthreads();
    // ---- hidden slices: up (y1 -> slice in LDS), down (slice -> g in registers) ------------
    auto slice_up = [&](int ch, float *hb) {
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rt][r] = 0.f;
      const float bb = b1[ch * 128 + c];
      const float *w2c = W2 + (size_t)c * hidden + ch * 128 + kq8(q);
      eb8_mma<RT16>(acc, bufB, wb, lane, [&](int s) { load_w4(wa, w2c, s); });
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hb[(rt * 16 + 4 * q + r) * EB_LD + c] = fmaxf(acc[rt][r] + bb, 0.f);
    };
    // slice ch lives in abuf (even ch) / other (odd ch): the attention rows are dead by now
    slice_up(0, abuf);
    __syncthreads();
    for (int ch = 0; ch + 1 < nchunk; ++ch) {
      float *hcur = (ch & 1) ? other : abuf, *hnext = (ch & 1) ? abuf : other;
      const float *w1c = W1 + (size_t)((ch + 1) * 128 + c) * 128 + kq8(q);
      eb8_mma<RT16>(gacc, hcur, wa, lane, [&](int s) { load_w4(wb, w1c, s); });
      slice_up(ch + 1, hnext);
      __syncthreads();
    }
    // last way down; the next tile's attention rows travel behind it into the free buffer
    float *hlast = ((nchunk - 1) & 1) ? other : abuf, *hfree = ((nchunk - 1) & 1) ? abuf : other;
    const float bb2 = b2[c], mean2 = norm2[c], mult2 = norm2[128 + c], beta2 = norm2[256 + c];
    const int next = tile + gridDim.x;
    if (next < ntiles) fetch_att(next);
    {
      const float *wo = Wo + (size_t)c * 128 + kq8(q);   // Wo fragment for the next tile
      eb8_mma<RT16>(gacc, hlast, wa, lane, [&](int s) { load_w4(wb, wo, s); });
    }
    if (next < ntiles) store_att(hfree);
    // ---- y = BN2(y1 + g + b2) ----------------------------------------------------------------
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rt * 16 + 4 * q + r;
        if (row < valid) {
          const float v = gacc[rt][r] + bb2 + bufB[row * EB_LD + c];
          y[(size_t)(row0 + row) * 128 + c] = (v - mean2) * mult2 + beta2;
        }
      }
    __syncthreads();   // next tile's attention rows complete, bufB and hlast free
    abuf = hfree; other = hlast;
  }
}

template <int RT16>
static int launch_encoder_block8(const float *att, const float *x, const vrp_encoder_layer &L,
                                 const float *norm1, const float *norm2, float *y, int rows,
                                 int hidden, hipStream_t st) {
  constexpr int RTW = 16 * RT16;
  const size_t lds = (size_t)3 * RTW * EB_LD * sizeof(float);
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_block8_kernel<RT16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("encoder_block8: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  const int ntiles = (rows + RTW - 1) / RTW;
  hipLaunchKernelGGL(encoder_block8_kernel<RT16>, dim3(min(ntiles, 256)), dim3(512), lds, st, att, x,
                     L.out_proj_weight, L.out_proj_bias, norm1, L.ff0_weight, L.ff0_bias,
                     L.ff2_weight, L.ff2_bias, norm2, y, rows, hidden, ntiles);
  VRP_CHECK_LAUNCH("encoder_block8");
  return 0;
}

// ---- C = epilogue(A W^T) for tall problems (M >= 20480 rows; N, K multiples of 128) ---------
// The train-mode encoder, the decoder backward and the N > 80 prologue run plain GEMMs; the
// LDS-tiled kernel in gemm.hip reaches 40-75 TFLOP/s on them at M = 81920 (its residual reads
// and result writes are 4-byte accesses in the accumulator layout).  This one reuses the
// block kernels' scheme: persistent workgroups on 80-row tiles, a wave owns 16 of the 128
// columns of a pass, the stages of a tile are (column pass, 128-wide K chunk); the A chunk of
// a stage sits in LDS (two buffers: the next chunk travels behind the current stage's MFMAs,
// one barrier per stage, none while K = 128 keeps the chunk), the W fragment of the next stage
// is requested one 16-byte load per k-step.  Same epilogue as gemm_nt_kernel (bias, residual,
// BatchNorm affine, ReLU, gate), applied in the accumulator layout.
template <int RT16>
__global__ __launch_bounds__(512) void gemm_rows_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ W_, int ldw,
    const float *__restrict__ bias_, const float *__restrict__ R, int ldr,
    const float *__restrict__ norm_, const float *__restrict__ gate, float *__restrict__ C, int ldc,
    int M, int N, int K, int relu, int ntiles) {
  constexpr int RTW = 16 * RT16, PF = RTW * 32 / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const Abuf0 = smem, *const Abuf1 = smem + RTW * EB_LD, *const Cs = smem + 2 * RTW * EB_LD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int ncb = N >> 7, nkc = K >> 7, nst = ncb * nkc;
  float4 pa[PF];
  auto fetchA = [&](int tile, int kc) {
    const int row0 = tile * RTW;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      pa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < M) pa[u] = *reinterpret_cast<const float4 *>(A + (size_t)(row0 + r) * lda + kc * 128 + c4);
    }
  };
  auto storeA = [&](float *dst) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      *reinterpret_cast<float4 *>(dst + r * EB_LD + c4) = pa[u];
    }
  };
  int tile = blockIdx.x;
  int cur = 0;
  float w[32], wn[32];
  if (tile < ntiles) { fetchA(tile, 0); storeA(Abuf0); }
  {
    const float *w0 = W_ + (size_t)(wave * 16 + i16) * ldw + kq8(q);
#pragma unroll
    for (int s = 0; s < 32; s += 4) load_w4(w, w0, s);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    int zero;   // (keeps the tile-invariant weight loads inside the loop, see encoder_block8_kernel)
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    const float *W = W_ + zero, *bias = bias_ ? bias_ + zero : nullptr, *norm = norm_ ? norm_ + zero : nullptr;
    const int row0 = tile * RTW, valid = M - row0;
    const int next_tile = tile + gridDim.x;
    f32x4v acc[RT16];
    for (int st = 0; st < nst; ++st) {
      const int cb = st / nkc, kc = st - cb * nkc;
      if (kc == 0) {
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) acc[rt] = f32x4v{0.f, 0.f, 0.f, 0.f};
      }
      const bool last_stage = st + 1 == nst;
      const bool tile_left = next_tile < ntiles;
      // the A chunk of the next stage (same chunk while K = 128 and the tile lasts)
      const bool fetch = (nkc > 1) ? (!last_stage || tile_left) : (last_stage && tile_left);
      if (fetch) fetchA(last_stage ? next_tile : tile, last_stage ? 0 : (kc + 1 == nkc ? 0 : kc + 1));
      const bool epi = kc + 1 == nkc;
      // residual / gate pieces of this pass (whole rows, 16 bytes per thread), requested ahead
      // of the last chunk's MFMAs
      float4 rv[PF], gv[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
        rv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        gv[u] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (epi && r < valid) {
          if (R) rv[u] = *reinterpret_cast<const float4 *>(R + (size_t)(row0 + r) * ldr + cb * 128 + c4);
          if (gate) gv[u] = *reinterpret_cast<const float4 *>(gate + (size_t)(row0 + r) * ldc + cb * 128 + c4);
        }
      }
      // bias / BatchNorm affine of this thread's four columns (the same for its PF rows)
      const int cq = cb * 128 + (tid & 31) * 4;
      float4 bz = make_float4(0.f, 0.f, 0.f, 0.f), nmean = bz, nbeta = bz;
      float4 nmult = make_float4(1.f, 1.f, 1.f, 1.f);
      if (epi) {
        if (bias) bz = *reinterpret_cast<const float4 *>(bias + cq);
        if (norm) {
          nmean = *reinterpret_cast<const float4 *>(norm + cq);
          nmult = *reinterpret_cast<const float4 *>(norm + 128 + cq);
          nbeta = *reinterpret_cast<const float4 *>(norm + 256 + cq);
        }
      }
      {
        // next stage's W fragment: stage st + 1 of this tile, or stage 0 of the next tile
        const int sn = last_stage ? 0 : st + 1;
        const int cbn = sn / nkc, kcn = sn - cbn * nkc;
        const float *wp = W + (size_t)(cbn * 128 + wave * 16 + i16) * ldw + kcn * 128 + kq8(q);
        eb8_mma<RT16>(acc, cur ? Abuf1 : Abuf0, w, lane, [&](int s) { load_w4(wn, wp, s); });
      }
      if (fetch) storeA(cur ? Abuf0 : Abuf1);
      if (epi) {
        // the 80 x 128 block of this pass goes through LDS so that residual, gate and result
        // move as 16-byte pieces of whole rows (in the accumulator layout a store instruction
        // writes four 64-byte fragments)
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) Cs[(rt * 16 + 4 * q + r) * EB_LD + wave * 16 + i16] = acc[rt][r];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
          if (r < valid) {
            const float4 a4 = *reinterpret_cast<const float4 *>(Cs + r * EB_LD + c4);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            const float rr[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
            const float gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
            const float bb[4] = {bz.x, bz.y, bz.z, bz.w};
            const float nm[4] = {nmean.x, nmean.y, nmean.z, nmean.w};
            const float nu[4] = {nmult.x, nmult.y, nmult.z, nmult.w};
            const float nb[4] = {nbeta.x, nbeta.y, nbeta.z, nbeta.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] += bb[e];
              if (R) v[e] += rr[e];
              if (norm) v[e] = (v[e] - nm[e]) * nu[e] + nb[e];
              if (relu) v[e] = fmaxf(v[e], 0.f);
              if (gate && !(gg[e] > 0.f)) v[e] = 0.f;
            }
            *reinterpret_cast<float4 *>(C + (size_t)(row0 + r) * ldc + cb * 128 + c4) =
                make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 32; ++s) w[s] = wn[s];
      if (fetch || epi) __syncthreads();   // next A chunk in place / everybody done with Cs
      if (fetch) cur ^= 1;
    }
  }
}

// ---- the same GEMM on the bf16 matrix cores (round 5; encoder_x3.h) ---------------------------
// gemm_rows_kernel's stages -- (column pass, 128-wide K chunk), the A chunk of a stage in LDS, the
// next one behind the current stage's MFMAs -- with three-plane operands: the threads that store
// an A chunk into LDS split it (64-row tiles: 2 x 48 KB of planes + the 33 KB fp32 image the
// epilogue goes through); the W fragment of the next stage arrives in fp32 behind the MFMAs, as
// before, and is split in registers once per stage (32 values per lane: W is a raw parameter or
// an activation here, nobody pre-split it).  Weights as the first MFMA operand: a lane owns four
// consecutive columns of a row of the result image (16-byte LDS stores).  Same epilogue.
template <int RT16>
__global__ __launch_bounds__(512) void gemm_rows_x3_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ W_, int ldw,
    const float *__restrict__ bias_, const float *__restrict__ R, int ldr,
    const float *__restrict__ norm_, const float *__restrict__ gate, float *__restrict__ C, int ldc,
    int M, int N, int K, int relu, int ntiles) {
  constexpr int RTW = 16 * RT16, PE = RTW * X3_PITCH, PF = RTW * 32 / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16 *const Abuf0 = reinterpret_cast<__bf16 *>(smem), *const Abuf1 = Abuf0 + 3 * PE;
  float *const Cs = reinterpret_cast<float *>(Abuf1 + 3 * PE);     // [RTW][EB_LD]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int ncb = N >> 7, nkc = K >> 7, nst = ncb * nkc;
  float4 pa[PF];
  auto fetchA = [&](int tile, int kc) {
    const int row0 = tile * RTW;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      pa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < M) pa[u] = *reinterpret_cast<const float4 *>(A + (size_t)(row0 + r) * lda + kc * 128 + c4);
    }
  };
  auto storeA = [&](__bf16 *dst) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      x3_store4v(dst, PE, r, c4, pa[u]);
    }
  };
  // a W fragment in fp32: lane (i16 = weight row, q), chunk j: k = 32 j + 8 q .. + 7 (two float4)
  float4 wn[8];
  auto load_wn = [&](const float *wrow, int piece) { wn[piece] = *reinterpret_cast<const float4 *>(wrow + 32 * (piece >> 1) + 4 * (piece & 1)); };
  Frag3 wf;
  auto split_wn = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x8[8] = {wn[2 * j].x, wn[2 * j].y, wn[2 * j].z, wn[2 * j].w,
                           wn[2 * j + 1].x, wn[2 * j + 1].y, wn[2 * j + 1].z, wn[2 * j + 1].w};
      x3_split8(x8, wf.p[0][j], wf.p[1][j], wf.p[2][j]);
    }
  };
  int tile = blockIdx.x;
  int cur = 0;
  if (tile < ntiles) { fetchA(tile, 0); storeA(Abuf0); }
  {
    const float *w0 = W_ + (size_t)(wave * 16 + i16) * ldw + 8 * q;
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) load_wn(w0, piece);
    split_wn();
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    int zero;   // (keeps the tile-invariant weight loads inside the loop, see encoder_block8_kernel)
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    const float *W = W_ + zero, *bias = bias_ ? bias_ + zero : nullptr, *norm = norm_ ? norm_ + zero : nullptr;
    const int row0 = tile * RTW, valid = M - row0;
    const int next_tile = tile + gridDim.x;
    f32x4v acc[RT16];
    for (int st = 0; st < nst; ++st) {
      const int cb = st / nkc, kc = st - cb * nkc;
      if (kc == 0) {
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt) acc[rt] = f32x4v{0.f, 0.f, 0.f, 0.f};
      }
      const bool last_stage = st + 1 == nst;
      const bool tile_left = next_tile < ntiles;
      const bool fetch = (nkc > 1) ? (!last_stage || tile_left) : (last_stage && tile_left);
      if (fetch) fetchA(last_stage ? next_tile : tile, last_stage ? 0 : (kc + 1 == nkc ? 0 : kc + 1));
      const bool epi = kc + 1 == nkc;
      float4 rv[PF], gv[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
        rv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        gv[u] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (epi && r < valid) {
          if (R) rv[u] = *reinterpret_cast<const float4 *>(R + (size_t)(row0 + r) * ldr + cb * 128 + c4);
          if (gate) gv[u] = *reinterpret_cast<const float4 *>(gate + (size_t)(row0 + r) * ldc + cb * 128 + c4);
        }
      }
      const int cq = cb * 128 + (tid & 31) * 4;
      float4 bz = make_float4(0.f, 0.f, 0.f, 0.f), nmean = bz, nbeta = bz;
      float4 nmult = make_float4(1.f, 1.f, 1.f, 1.f);
      if (epi) {
        if (bias) bz = *reinterpret_cast<const float4 *>(bias + cq);
        if (norm) {
          nmean = *reinterpret_cast<const float4 *>(norm + cq);
          nmult = *reinterpret_cast<const float4 *>(norm + 128 + cq);
          nbeta = *reinterpret_cast<const float4 *>(norm + 256 + cq);
        }
      }
      {
        // next stage's W fragment: stage st + 1 of this tile, or stage 0 of the next tile
        const int sn = last_stage ? 0 : st + 1;
        const int cbn = sn / nkc, kcn = sn - cbn * nkc;
        const float *wp = W + (size_t)(cbn * 128 + wave * 16 + i16) * ldw + kcn * 128 + 8 * q;
        x3_mma<RT16>(acc, cur ? Abuf1 : Abuf0, PE, wf, lane, [&](int it) { if (it < 8) load_wn(wp, it); });
      }
      if (fetch) storeA(cur ? Abuf0 : Abuf1);
      if (epi) {
        // the 64 x 128 block of this pass goes through LDS so that residual, gate and result
        // move as 16-byte pieces of whole rows
#pragma unroll
        for (int rt = 0; rt < RT16; ++rt)
          *reinterpret_cast<float4 *>(Cs + (rt * 16 + i16) * EB_LD + wave * 16 + 4 * q) =
              make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
          if (r < valid) {
            const float4 a4 = *reinterpret_cast<const float4 *>(Cs + r * EB_LD + c4);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            const float rr[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
            const float gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
            const float bb[4] = {bz.x, bz.y, bz.z, bz.w};
            const float nm[4] = {nmean.x, nmean.y, nmean.z, nmean.w};
            const float nu[4] = {nmult.x, nmult.y, nmult.z, nmult.w};
            const float nb[4] = {nbeta.x, nbeta.y, nbeta.z, nbeta.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] += bb[e];
              if (R) v[e] += rr[e];
              if (norm) v[e] = (v[e] - nm[e]) * nu[e] + nb[e];
              if (relu) v[e] = fmaxf(v[e], 0.f);
              if (gate && !(gg[e] > 0.f)) v[e] = 0.f;
            }
            *reinterpret_cast<float4 *>(C + (size_t)(row0 + r) * ldc + cb * 128 + c4) =
                make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
      split_wn();
      if (fetch || epi) __syncthreads();   // next A chunk in place / everybody done with Cs
      if (fetch) cur ^= 1;
    }
  }
}

// (Round 4, measured and NOT kept -- docs/rounds/DESIGN_rounds_1-5.md 3.5: the pass epilogue spread over the next pass's
// k-step groups, 127 vs 117 us at 81920 x 384 x 128; two four-wave workgroups of 48-row tiles per
// CU instead of one eight-wave workgroup of 80 rows, 187 vs 120 us.  PMC on this kernel: matrix
// pipe busy 52 % of the cycles, waves parked 27 % -- its eight waves reach the two barriers of a
// pass in lockstep.)
// ---- N = 128 NCT, K = 128 (the train-mode in_proj, the decoder backward's K / V projections and
// dO2: 81920..102400 x 384 x 128): ALL of W stays in registers -- a wave owns columns
// ct 128 + 16 wave + i16 of every 128-column block ct, NCT x 32 fragment registers loaded once per
// workgroup -- and a tile is ONE MFMA phase: an A fragment read from LDS feeds NCT column tiles (a
// third of gemm_rows_kernel's LDS reads), no weight traffic, no barrier inside the phase.  The
// column blocks then leave through two alternating 80 x 128 LDS images (whole rows, 16-byte
// pieces): ONE barrier per block instead of two, and the waves drift apart between them.  The next
// tile's rows travel in registers under the MFMA phase and are stored behind the first barrier of
// the epilogue (every wave has left the phase by then).
template <int NCT, bool GATE>
__global__ __launch_bounds__(512) void gemm_rows_wide_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ W, int ldw,
    const float *__restrict__ bias, const float *__restrict__ gate, float *__restrict__ C, int ldc,
    int M, int relu, int ntiles) {
  constexpr int RT16 = 5, RTW = 80, PF = RTW * 32 / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const Abuf = smem, *const Cs0 = smem + RTW * EB_LD, *const Cs1 = smem + 2 * RTW * EB_LD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  float4 pa[PF];
  auto fetchA = [&](int tile) {
    const int row0 = tile * RTW;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      pa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < M) pa[u] = *reinterpret_cast<const float4 *>(A + (size_t)(row0 + r) * lda + c4);
    }
  };
  auto storeA = [&]() {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      *reinterpret_cast<float4 *>(Abuf + r * EB_LD + c4) = pa[u];
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) { fetchA(tile); storeA(); }
  float w[NCT][32];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const float *w0 = W + (size_t)(ct * 128 + wave * 16 + i16) * ldw + kq8(q);
#pragma unroll
    for (int s = 0; s < 32; s += 4) load_w4(w[ct], w0, s);
  }
  __syncthreads();
  int par = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * RTW, valid = M - row0;
    const int next_tile = tile + gridDim.x;
    const bool more = next_tile < ntiles;
    if (more) fetchA(next_tile);
    f32x4v acc[NCT][RT16];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt) acc[ct][rt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    {
      const float *ap = Abuf + i16 * EB_LD + kq8(q);
      float4 a[2][RT16];
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt) a[0][rt] = *reinterpret_cast<const float4 *>(ap + rt * 16 * EB_LD);
#pragma unroll
      for (int s = 0; s < 32; s += 4) {
        const int cur = (s >> 2) & 1;
        if (s + 4 < 32) {
#pragma unroll
          for (int rt = 0; rt < RT16; ++rt)
            a[cur ^ 1][rt] = *reinterpret_cast<const float4 *>(ap + rt * 16 * EB_LD + ks8(s + 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
          for (int rt = 0; rt < RT16; ++rt) {
            const float av = k4 == 0 ? a[cur][rt].x : k4 == 1 ? a[cur][rt].y : k4 == 2 ? a[cur][rt].z : a[cur][rt].w;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
              acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w[ct][s + k4], acc[ct][rt], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      float *Cs = par ? Cs1 : Cs0;
      par ^= 1;
      float4 gv[GATE ? PF : 1];
      if (GATE) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
          gv[GATE ? u : 0] = make_float4(1.f, 1.f, 1.f, 1.f);
          if (r < valid)
            gv[GATE ? u : 0] = *reinterpret_cast<const float4 *>(gate + (size_t)(row0 + r) * ldc + ct * 128 + c4);
        }
      }
      const int cq = ct * 128 + (tid & 31) * 4;
      const float4 bz = bias ? *reinterpret_cast<const float4 *>(bias + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(rt * 16 + 4 * q + r) * EB_LD + wave * 16 + i16] = acc[ct][rt][r];
      __syncthreads();
      if (ct == 0 && more) storeA();   // every wave has left the MFMA phase
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
        if (r < valid) {
          const float4 a4 = *reinterpret_cast<const float4 *>(Cs + r * EB_LD + c4);
          float v[4] = {a4.x + bz.x, a4.y + bz.y, a4.z + bz.z, a4.w + bz.w};
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (GATE) {
            const float gg[4] = {gv[GATE ? u : 0].x, gv[GATE ? u : 0].y, gv[GATE ? u : 0].z, gv[GATE ? u : 0].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (!(gg[e] > 0.f)) v[e] = 0.f;
          }
          *reinterpret_cast<float4 *>(C + (size_t)(row0 + r) * ldc + ct * 128 + c4) =
              make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    if (NCT == 1) __syncthreads();
  }
}

template <int NCT, bool GATE>
static int launch_gemm_rows_wide(const float *A, int lda, const float *W, int ldw, const float *bias,
                                 const float *gate, float *C, int ldc, int M, int relu, hipStream_t st) {
  constexpr int RTW = 80;
  const size_t lds = (size_t)3 * RTW * EB_LD * sizeof(float);
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_rows_wide_kernel<NCT, GATE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("gemm_rows_wide: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  const int ntiles = (M + RTW - 1) / RTW;
  hipLaunchKernelGGL((gemm_rows_wide_kernel<NCT, GATE>), dim3(min(ntiles, 256)), dim3(512), lds, st, A,
                     lda, W, ldw, bias, gate, C, ldc, M, relu, ntiles);
  VRP_CHECK_LAUNCH("gemm_rows_wide");
  return 0;
}

int vrp_launch_gemm_rows(const float *A, int lda, const float *W, int ldw, const float *bias,
                         const float *R, int ldr, const float *norm, const float *gate, float *C,
                         int ldc, int M, int N, int K, int relu, hipStream_t st) {
  static const bool narrow = getenv("VRP_GEMM_ROWS_NARROW") != nullptr;   // A/B aid
  if (!narrow && K == 128 && !R && !norm) {
    if (N == 384)
      return gate ? launch_gemm_rows_wide<3, true>(A, lda, W, ldw, bias, gate, C, ldc, M, relu, st)
                  : launch_gemm_rows_wide<3, false>(A, lda, W, ldw, bias, gate, C, ldc, M, relu, st);
    if (N == 256)
      return gate ? launch_gemm_rows_wide<2, true>(A, lda, W, ldw, bias, gate, C, ldc, M, relu, st)
                  : launch_gemm_rows_wide<2, false>(A, lda, W, ldw, bias, gate, C, ldc, M, relu, st);
  }
  static const bool fp32 = getenv("VRP_GEMM_FP32") != nullptr;   // A/B aid: the fp32-MFMA kernel
  if (!fp32) {
    constexpr int RT16 = 4, RTW = 64;
    const size_t lds = (size_t)2 * 3 * RTW * X3_PITCH * 2 + (size_t)RTW * EB_LD * sizeof(float);
    static VrpAttrOnce attr_set;
    if (!attr_set.done()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_rows_x3_kernel<RT16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        vrp_set_error("gemm_rows_x3: cannot raise dynamic LDS to %zu bytes", lds);
        return 1;
      }
      attr_set.mark();
    }
    const int ntiles = (M + RTW - 1) / RTW;
    hipLaunchKernelGGL(gemm_rows_x3_kernel<RT16>, dim3(min(ntiles, 256)), dim3(512), lds, st, A, lda, W,
                       ldw, bias, R, ldr, norm, gate, C, ldc, M, N, K, relu, ntiles);
    VRP_CHECK_LAUNCH("gemm_rows_x3");
    return 0;
  }
  constexpr int RT16 = 5, RTW = 80;
  const size_t lds = (size_t)3 * RTW * EB_LD * sizeof(float);
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_rows_kernel<RT16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("gemm_rows: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  const int ntiles = (M + RTW - 1) / RTW;
  hipLaunchKernelGGL(gemm_rows_kernel<RT16>, dim3(min(ntiles, 256)), dim3(512), lds, st, A, lda, W,
                     ldw, bias, R, ldr, norm, gate, C, ldc, M, N, K, relu, ntiles);
  VRP_CHECK_LAUNCH("gemm_rows");
  return 0;
}

// ---- attention straight from the q|k|v rows in global memory (N > 64, and train mode) -----
// Same products and softmax as qa8_stage_attention_mfma, one workgroup per graph, wave = head,
// query tiles outermost so that only one row of score tiles is live (N <= 128: eight tiles).
// K and V fragments of the head are loaded once per graph (16-byte / 4-byte pieces of the
// 1536-byte q|k|v rows), Q per query tile.
// exp_nonpos (common.h) without the clamp: the arguments here are differences of finite scores
// (this kernel is VALU-bound on its softmax: 28 exponentials per lane and query tile at N = 100)
__device__ __forceinline__ float att_exp(float x) {
  const float l2e_hi = 1.44269502162933349609375f, l2e_lo = 1.9259629911e-8f;
  const float t = x * l2e_hi;
  float r = fmaf(x, l2e_hi, -t);
  r = fmaf(x, l2e_lo, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.693147180559945f, e);
}
// one graph, head h: `base` = its q|k|v rows (global memory or LDS), LD floats apart.
// HOLD: the head's K and V fragments sit in registers for the whole graph (global memory: loaded
// once); !HOLD (LDS image): they are read again for every query tile -- 56 registers less.
template <int NT, int LD, bool HOLD>
__device__ __forceinline__ void attention_rows_mfma(const float *base, float *__restrict__ out,
                                                    int N, int lane, int h) {
  const int i16 = lane & 15, q = lane >> 4;
  float4 kf[NT];
  float vv[NT][4];
  auto load_kv = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      kf[t] = *reinterpret_cast<const float4 *>(base + (size_t)min(16 * t + i16, N - 1) * LD + 128 + h * 16 + 4 * q);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        vv[t][r4] = base[(size_t)min(16 * t + 4 * q + r4, N - 1) * LD + 256 + h * 16 + i16];
    }
  };
  if (HOLD) load_kv();
  // the query rows of tile tm + 1 are requested before tile tm is worked on (a round trip per
  // tile in front of its first MFMA otherwise)
  float4 a_next = *reinterpret_cast<const float4 *>(base + (size_t)min(i16, N - 1) * LD + h * 16 + 4 * q);
#pragma unroll
  for (int tm = 0; tm < NT; ++tm) {
    if (16 * tm >= N) break;
    if (!HOLD) {
      asm volatile("" ::: "memory");   // (no hoisting of the seven tiles' reads out of the loop)
      load_kv();
    }
    const float4 a = a_next;
    if (tm + 1 < NT)
      a_next = *reinterpret_cast<const float4 *>(base + (size_t)min(16 * (tm + 1) + i16, N - 1) * LD + h * 16 + 4 * q);
    const float4 qf = make_float4(a.x * 0.25f, a.y * 0.25f, a.z * 0.25f, a.w * 0.25f);  // 1/sqrt(16)
    f32x4v st[NT];  // st[tn][r4] = S[m = 16tm + i16][n = 16tn + 4q + r4]
#pragma unroll
    for (int tn = 0; tn < NT; ++tn) {
      f32x4v d = {0.f, 0.f, 0.f, 0.f};
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[tn].x, qf.x, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[tn].y, qf.y, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[tn].z, qf.z, d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[tn].w, qf.w, d, 0, 0, 0);
      st[tn] = d;
    }
    // (only the last key tile can reach beyond N: 16 (NT - 1) < N by the choice of NT)
    float mx = -INFINITY;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        if (tn + 1 < NT || 16 * tn + 4 * q + r4 < N) mx = fmaxf(mx, st[tn][r4]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float pw = (tn + 1 < NT || 16 * tn + 4 * q + r4 < N) ? att_exp(st[tn][r4] - mx) : 0.f;
        st[tn][r4] = pw;
        sum += pw;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    // two chains (even / odd key tiles): a wave is alone with its dependent MFMAs here
    f32x4v o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tn = 0; tn < NT; ++tn)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        f32x4v &o = (tn & 1) ? o1 : o0;
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[tn][r4], st[tn][r4], o, 0, 0, 0);
      }
    const int m = 16 * tm + i16;   // D[d = 4q + r4][m = i16]
    if (m < N) {
      const float inv = 1.f / sum;
      *reinterpret_cast<float4 *>(out + (size_t)m * VRP_EMB + h * 16 + 4 * q) =
          make_float4((o0[0] + o1[0]) * inv, (o0[1] + o1[1]) * inv, (o0[2] + o1[2]) * inv,
                      (o0[3] + o1[3]) * inv);
    }
  }
}

template <int NT>
__global__ __launch_bounds__(512) void encoder_attention_mfma_kernel(const float *__restrict__ qkv,
                                                                      float *__restrict__ att, int N) {
  const int lane = threadIdx.x & 63;
  const int h = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  attention_rows_mfma<NT, 384, true>(qkv + (size_t)blockIdx.x * N * 384,
                               att + (size_t)blockIdx.x * N * VRP_EMB, N, lane, h);
}

static int launch_attention_mfma(const float *qkv, float *att, int B, int N, hipStream_t st) {
  switch ((N + 15) / 16) {
#define VRP_ATT_CASE(NT_)                                                                       \
    case NT_:                                                                                   \
      hipLaunchKernelGGL(encoder_attention_mfma_kernel<NT_>, dim3(B), dim3(512), 0, st, qkv, att, N); \
      break;
    VRP_ATT_CASE(1) VRP_ATT_CASE(2) VRP_ATT_CASE(3) VRP_ATT_CASE(4)
    VRP_ATT_CASE(5) VRP_ATT_CASE(6) VRP_ATT_CASE(7) VRP_ATT_CASE(8)
#undef VRP_ATT_CASE
    default:
      vrp_set_error("encoder attention: N=%d unsupported", N);
      return 2;
  }
  VRP_CHECK_LAUNCH("encoder_attention_mfma");
  return 0;
}

// ---- in_proj + attention of whole graphs in one launch, LARGE batches (eval mode) ----------
// The projection GEMM wrote q|k|v (B*N x 384 fp32: 503 MB at 8192 x 40) only for the attention
// kernel to read it back; here a workgroup takes G = 80 / N whole graphs (N = 40: two graphs =
// five 16-row MFMA tiles, no padded rows) from the layer input to the attention output with
// q|k|v in LDS: 8 waves, wave w projects column block w (48 columns) and then runs head w.
// X_s and Q_s share the LDS (124 KB), so one workgroup per CU: the workgroups are persistent
// (grid = CUs), keep their weight fragments in registers across tiles and fetch the next
// tile's rows into registers while the current tile is in the matrix cores.
template <int RT16>
__global__ __launch_bounds__(512) void encoder_qkv_attn8_kernel(const float *__restrict__ x,
                                                                 const float *__restrict__ Win,
                                                                 const float *__restrict__ bin,
                                                                 float *__restrict__ att, int B,
                                                                 int N, int G, int ntiles) {
  constexpr int RTW = 16 * RT16, PF = RTW * 32 / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Q_s = smem;   // [RTW][QA_QLD]; the first RTW * QA_XLD floats hold the input tile first
  float *X_s = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float w[3][32];
  qa8_load_w<0>(w, Win, lane, wave);
  float4 pf[PF];
  auto fetch = [&](int tile) {
    const int g0 = tile * G;
    const int rows = min(G, B - g0) * N;
    const size_t row0 = (size_t)g0 * N;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows) pf[u] = *reinterpret_cast<const float4 *>(x + (row0 + r) * VRP_EMB + c4);
    }
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const int g0 = tile * G;
    const int graphs = min(G, B - g0);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      *reinterpret_cast<float4 *>(X_s + r * QA_XLD + c4) = pf[u];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    qa8_project<RT16, true, NoPre, 0>(X_s, Q_s, w, bin, lane, wave);
    __syncthreads();
    float *o = att + (size_t)g0 * N * VRP_EMB;
    if (N <= 32) qa8_stage_attention_mfma<2>(Q_s, N, graphs, RTW - 1, o, lane, wave, VRP_EMB);
    else if (N <= 48) qa8_stage_attention_mfma<3>(Q_s, N, graphs, RTW - 1, o, lane, wave, VRP_EMB);
    else qa8_stage_attention_mfma<4>(Q_s, N, graphs, RTW - 1, o, lane, wave, VRP_EMB);
    __syncthreads();   // everybody done with Q_s before the next tile lands in X_s
  }
}

static int launch_qkv_attn8(const float *x, const float *Win, const float *bin, float *att, int B,
                            int N, hipStream_t st) {
  constexpr int RT16 = 5, RTW = 80;
  const size_t lds = (size_t)RTW * QA_QLD * sizeof(float);
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_qkv_attn8_kernel<RT16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("qkv_attn8: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  const int G = RTW / N, ntiles = (B + G - 1) / G;
  hipLaunchKernelGGL(encoder_qkv_attn8_kernel<RT16>, dim3(min(ntiles, 256)), dim3(512),
                     lds, st, x, Win, bin, att, B, N, G, ntiles);
  VRP_CHECK_LAUNCH("encoder_qkv_attn8");
  return 0;
}

// ---- the same for ONE graph of 64 < N <= 102 nodes per workgroup pass (eval mode) -----------
// Round 4.  At N = 100 (configs[4]) the 80-row kernel above has no whole graph to put into its
// five row tiles, so the layer ran the projection as a GEMM (q|k|v written: 315 MB at 2048 x 100)
// and encoder_attention_mfma_kernel read it back: 229 + 181 us per layer.  A graph's q|k|v is
// N x 384 floats: with rows of 388 that is 155 KB at N = 100 -- it fits the 160 KB of LDS if the
// input tile does not need a place of its own at the same time.  Layout: Q_s = rows 0 .. N-1 of
// q|k|v from offset 0; the (<= 112 x 132) input tile X_s from the offset of Q_s row 64.  The
// projection runs in two row halves: rows 0..63 (four tiles) read X_s rows 0..63 and write Q_s
// rows 0..63 -- below X_s, no overlap --, then rows 64.. read the rest of X_s, every wave passes
// a barrier, and their q|k|v rows go where X_s was.  Attention = the large-N routine
// (attention_rows_mfma: query tiles outermost, one row of score tiles live) on the LDS image.
// Persistent workgroups (grid = CUs): weight fragments stay in registers, the next graph's rows
// are requested while the current one is in the matrix cores.
template <int NT>
__global__ __launch_bounds__(512) void encoder_qkv_attn_graph_kernel(const float *__restrict__ x,
                                                                      const float *__restrict__ Win,
                                                                      const float *__restrict__ bin,
                                                                      float *__restrict__ att, int B,
                                                                      int N) {
  constexpr int RTW = 16 * NT, PF = RTW * 32 / 512;
  static_assert(NT >= 5 && NT <= 7, "64 < N <= 102");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Q_s = smem;
  float *X_s = smem + 64 * QA_QLD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float w[3][32];
  qa8_load_w<0>(w, Win, lane, wave);
  float4 pf[PF];
  auto fetch = [&](int g) {
    const size_t row0 = (size_t)g * N;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < N) pf[u] = *reinterpret_cast<const float4 *>(x + (row0 + r) * VRP_EMB + c4);
    }
  };
  int g = blockIdx.x;
  if (g < B) fetch(g);
  for (; g < B; g += gridDim.x) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      *reinterpret_cast<float4 *>(X_s + r * QA_XLD + c4) = pf[u];   // rows >= N: zeros
    }
    __syncthreads();
    qa8_project<4, false, NoPre, 0>(X_s, Q_s, w, bin, lane, wave);
    qa8_project<NT - 4, true, NoPre, 0>(X_s + 64 * QA_XLD, Q_s + 64 * QA_QLD, w, bin, lane, wave,
                                        NoPre(), N - 64);
    __syncthreads();
    // (requested here, not ahead of the projection: its accumulators and these 4 PF registers do
    // not fit side by side; the attention phase is long enough to cover the round trip)
    if (g + (int)gridDim.x < B) fetch(g + gridDim.x);
    attention_rows_mfma<NT, QA_QLD, false>(Q_s, att + (size_t)g * N * VRP_EMB, N, lane, wave);
    __syncthreads();   // everybody done with Q_s before the next graph lands in X_s
  }
}

static int launch_qkv_attn_graph(const float *x, const float *Win, const float *bin, float *att,
                                 int B, int N, hipStream_t st) {
  const int NT = (N + 15) / 16;
  // Q_s rows 0..N-1, or X_s (16 NT rows of 132 floats) behind Q_s row 64 -- whichever ends later
  const size_t fl = (size_t)N * QA_QLD > (size_t)64 * QA_QLD + (size_t)16 * NT * QA_XLD
                        ? (size_t)N * QA_QLD : (size_t)64 * QA_QLD + (size_t)16 * NT * QA_XLD;
  const size_t lds = fl * sizeof(float);
  const void *fn = NT == 5 ? reinterpret_cast<const void *>(&encoder_qkv_attn_graph_kernel<5>)
                 : NT == 6 ? reinterpret_cast<const void *>(&encoder_qkv_attn_graph_kernel<6>)
                           : reinterpret_cast<const void *>(&encoder_qkv_attn_graph_kernel<7>);
  static VrpAttrOnce attr_set[3];
  if (!attr_set[NT - 5].done()) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      vrp_set_error("qkv_attn_graph: cannot raise dynamic LDS to 160 KB");
      return 1;
    }
    attr_set[NT - 5].mark();
  }
  const dim3 grid(min(B, 256)), block(512);
  if (NT == 5) hipLaunchKernelGGL(encoder_qkv_attn_graph_kernel<5>, grid, block, lds, st, x, Win, bin, att, B, N);
  else if (NT == 6) hipLaunchKernelGGL(encoder_qkv_attn_graph_kernel<6>, grid, block, lds, st, x, Win, bin, att, B, N);
  else hipLaunchKernelGGL(encoder_qkv_attn_graph_kernel<7>, grid, block, lds, st, x, Win, bin, att, B, N);
  VRP_CHECK_LAUNCH("encoder_qkv_attn_graph");
  return 0;
}
// (N such that q|k|v rows and the input tile fit 160 KB of LDS, and enough graphs to fill the chip)
static bool qkv_attn_graph_applies(int train, int B, int N) {
  static const bool off = getenv("VRP_UNFUSED_QKV") != nullptr;
  if (off || train || N <= 64 || B < 256) return false;
  const int NT = (N + 15) / 16;
  const size_t a = (size_t)N * QA_QLD, b = (size_t)64 * QA_QLD + (size_t)16 * NT * QA_XLD;
  return (a > b ? a : b) * sizeof(float) <= 160 * 1024;
}

// What the decoder prologue needs from the finished embeddings (graph_decoder.py:75-77 and the
// constant part of the logits): written by the stack kernel's epilogue when it runs inside
// vrp_rollout, so that no separate pass over the embeddings is needed.
struct StackEpilogue {
  const float *mb;             // (128)  decoder's folded bias vector
  float *g, *cvec;             // (B,128) graph mean, (B,N) e_m . mb
  unsigned long long *hist;    // (2N,B) persistent step kernel's hand-off words (cleared here)
  int32_t *err;
  const float *warm;           // the decoder prologue's projection weights (786 KB): touched here,
  int warm_floats;             // one share per workgroup, so that they wait in every XCD's L2
                               // when the prologue starts (58 -> 55 us at 512 x 20)
  const float *wqgT, *bq;      // (128,384), (384): graph-embedding block of the glimpse query, and
  float *QG;                   // (B,384) = Wq_g g + bq of this workgroup's graphs (round 6: was a
                               // GEMM launch of its own between the encoder and the prologue)
};
// QG rows of the workgroup's graphs from their means in LDS (`gs`: [graphs][128]; `part`: scratch
// of 4 x 4 x 384 floats).  384 threads: thread = (four consecutive output columns c4, one quarter
// kg of the 128-long inner dimension): 32 coalesced 16-byte loads of WqgT, all in flight at once --
// the first version walked k = 0..127 per column with eight loads in flight and spent 13.5 k cycles
// (16 dependent L2 round trips) in a kernel whose whole epilogue was 4 k.  The four partial sums
// of a column are added in kg order, then the bias.  Four graphs per pass.
__device__ __forceinline__ void stack_epilogue_qg(const StackEpilogue &ep, const float *gs,
                                                  float *part, int g0, int graphs, int tid) {
  const int c4 = (tid % 96) * 4, kg = tid / 96;      // tid < 384: 96 column quads x 4 k-quarters
  for (int gb = 0; gb < graphs; gb += 4) {
    if (tid < 384) {
      float4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const float *wp = ep.wqgT + (size_t)(32 * kg) * 384 + c4;
#pragma unroll 16
      for (int k = 0; k < 32; ++k) {
        const float4 wv = *reinterpret_cast<const float4 *>(wp + (size_t)k * 384);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float gv = gs[min(gb + u, graphs - 1) * 128 + 32 * kg + k];
          acc[u].x = fmaf(gv, wv.x, acc[u].x); acc[u].y = fmaf(gv, wv.y, acc[u].y);
          acc[u].z = fmaf(gv, wv.z, acc[u].z); acc[u].w = fmaf(gv, wv.w, acc[u].w);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) *reinterpret_cast<float4 *>(part + (kg * 4 + u) * 384 + c4) = acc[u];
    }
    __syncthreads();
    if (tid < 384) {
      const float bias = ep.bq[tid];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (gb + u < graphs)
          ep.QG[(size_t)(g0 + gb + u) * 384 + tid] =
              bias + (((part[(0 * 4 + u) * 384 + tid] + part[(1 * 4 + u) * 384 + tid]) +
                       part[(2 * 4 + u) * 384 + tid]) + part[(3 * 4 + u) * 384 + tid]);
    }
    __syncthreads();
  }
}
struct StackSetup {            // rollout set-up fused in front (vrp_rollout): env may be null
  vrp_env env;
  float *acc_loss, *acc_logp;
  int32_t *notdone;
  int nflags, from_env;
};

template <int RT16>
__global__ __launch_bounds__(512) void encoder_stack_kernel(vrp_encoder_weights w,
                                                             const float *__restrict__ x,
                                                             const float *__restrict__ norms_in,
                                                             float *__restrict__ y, int B, int N,
                                                             int G, StackSetup su, StackEpilogue ep) {
  constexpr int RTW = 16 * RT16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *bufB = smem;                          // [RTW][EB_LD]   layer input, y1, layer output
  float *bufA = bufB + RTW * EB_LD;            // [RTW][EB_LD]   attention output, hidden slices
  float *Q_s = bufA + RTW * EB_LD;             // [RTW][QA_QLD]  q | k | v of every node
  float *norm_s = Q_s + RTW * QA_QLD;          // [16][384]      eval-mode BN affines (from_env)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int c = wave * 16 + i16;
  const int g0 = blockIdx.x * G;
  const int graphs = min(G, B - g0);
  const int rows = graphs * N;                 // valid rows of this workgroup's tile
  const size_t row0 = (size_t)g0 * N;
  const float *norms = norms_in;
  ST_MARK(0);
  ST_MARK(1);
  for (int idx = tid; idx < RTW * 32; idx += 512) {
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!su.from_env && r < rows) v = *reinterpret_cast<const float4 *>(x + (row0 + r) * VRP_EMB + c4);
    *reinterpret_cast<float4 *>(bufB + r * EB_LD + c4) = v;
    *reinterpret_cast<float4 *>(bufA + r * EB_LD + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float win[3][32];  // in_proj fragments of the coming layer (the first: behind the set-up)
  qa8_load_w_half(win, w.layer[0].in_proj_weight, lane, wave, 0);
  __builtin_amdgcn_sched_barrier(0);
  if (su.from_env) {
    // rollout set-up of this workgroup's graphs (one wave each), BN affines into LDS
    __syncthreads();
    if (blockIdx.x == 0 && tid < su.nflags) su.notdone[tid] = 0;
    {
      const int parts = graphs >= 8 ? 1 : 8 / graphs;  // waves per graph
      for (int gw = wave; gw < graphs * parts; gw += 8) {
        const int g = gw / parts, part = gw - g * parts;
        setup_graph_wave(su.env, w, g0 + g, lane, bufB + g * N * EB_LD, EB_LD, su.acc_loss,
                         su.acc_logp, part, parts);
      }
    }
    for (int i = tid; i < 2 * w.num_layers * 128; i += 512) {
      const int blk = i >> 7, cc = i & 127, l = blk >> 1, second = blk & 1;
      const vrp_encoder_layer &L = w.layer[l];
      const float *rm = second ? L.bn2_running_mean : L.bn1_running_mean;
      const float *rv = second ? L.bn2_running_var : L.bn1_running_var;
      const float *wt = second ? L.bn2_weight : L.bn1_weight;
      const float *bs = second ? L.bn2_bias : L.bn1_bias;
      float *o = norm_s + blk * 384;
      o[cc] = rm[cc];
      o[128 + cc] = wt[cc] / sqrtf(rv[cc] + 1e-5f);
      o[256 + cc] = bs[cc];
    }
    norms = norm_s;
  }
  ST_MARK(2);
  __syncthreads();
  for (int l = 0; l < w.num_layers; ++l) {
    const vrp_encoder_layer &L = w.layer[l];
    ST_MARK(4 + 24 * l);
    float wa[32];  // Wo fragment: in flight while in_proj and the attention run
    {
      // behind the MFMAs of k-steps 0..15 (fragments already here): the other half of the
      // in_proj fragments (k-steps 16..31, three column tiles), then the Wo fragment
      const float *wi = L.in_proj_weight + (size_t)(wave * 48 + i16) * VRP_EMB + kq8(q);
      const float *wo = L.out_proj_weight + (size_t)c * 128 + kq8(q);
      qa8_project<RT16, false>(bufB, Q_s, win, L.in_proj_bias, lane, wave, [&](int s) {
        if (s < 12) {
          const int ct = s / 4;
#pragma unroll
          for (int k = 16; k < 32; k += 4) load_w4(win[ct], wi + (size_t)ct * 16 * VRP_EMB, k);
        } else if (s < 28) {
          load_w4(wa, wo, 2 * (s - 12));
          load_w4(wa, wo, 2 * (s - 12) + 4);
        }
      });
    }
    ST_MARK(4 + 24 * l + 1);
    __syncthreads();
    ST_MARK(4 + 24 * l + 2);
    if (N <= 16) qa8_stage_attention_mfma<1>(Q_s, N, graphs, RTW - 1, bufA, lane, wave, EB_LD);
    else if (N <= 32) qa8_stage_attention_mfma<2>(Q_s, N, graphs, RTW - 1, bufA, lane, wave, EB_LD);
    else if (N <= 48) qa8_stage_attention_mfma<3>(Q_s, N, graphs, RTW - 1, bufA, lane, wave, EB_LD);
    else qa8_stage_attention_mfma<4>(Q_s, N, graphs, RTW - 1, bufA, lane, wave, EB_LD);
    ST_MARK(4 + 24 * l + 3);
    __syncthreads();
    ST_MARK(4 + 24 * l + 4);
    eb8_block_stages<RT16>(bufA, Q_s, bufB, wa, L.out_proj_bias, norms + (2 * l) * 384, L.ff0_weight,
                           L.ff0_bias, L.ff2_weight, L.ff2_bias, norms + (2 * l + 1) * 384,
                           bufB, RTW, w.hidden, lane, wave, EB_LD, win,
                           l + 1 < w.num_layers ? w.layer[l + 1].in_proj_weight : nullptr,
                           4 + 24 * l);
    ST_MARK(4 + 24 * l + 23);
    __syncthreads();
  }
  ST_MARK(ST_SLOTS - 3);
  // ---- result: coalesced 16-byte stores; the decoder's per-graph constants on the way -------
  for (int idx = tid; idx < rows * 32; idx += 512) {
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    *reinterpret_cast<float4 *>(y + (row0 + r) * VRP_EMB + c4) =
        *reinterpret_cast<const float4 *>(bufB + r * EB_LD + c4);
  }
  if (ep.g) {
    // graph embedding = mean over nodes (sum in node order, then divide; graph_decoder.py:75-77)
    for (int i = tid; i < graphs * 128; i += 512) {
      const int g = i >> 7, cc = i & 127;
      float s = 0.f;
      for (int n = 0; n < N; ++n) s += bufB[(g * N + n) * EB_LD + cc];
      ep.g[(size_t)(g0 + g) * VRP_EMB + cc] = s / (float)N;
    }
    // cvec[b][m] = e_m . mb: a wave per row, lane = two columns
    const float2 m = reinterpret_cast<const float2 *>(ep.mb)[lane];
    for (int r = wave; r < rows; r += 8) {
      const float2 ev = *reinterpret_cast<const float2 *>(bufB + r * EB_LD + 2 * lane);
      const float s = wave_sum(fmaf(ev.x, m.x, ev.y * m.y));
      if (lane == 0) ep.cvec[row0 + r] = s;
    }
    for (int i = tid; i < graphs * 2 * N; i += 512) {
      const int g = i / (2 * N), t = i - g * 2 * N;
      ep.hist[(size_t)t * B + g0 + g] = 0ull;
    }
    if (blockIdx.x == 0 && tid == 0) *ep.err = 0;
    // (round 4, measured and removed: Wq_g g + bq computed here -- one thread per output, the
    // weight row from L2 -- instead of the 6-us GEMM launch in front of the prologue: 0.284 vs
    // 0.280 ms per rollout; the kernel's tail is on the critical path of all 256 workgroups)
    if (ep.warm) {
      // workgroups b, b + 8, ... share an XCD: together they touch every 128-byte line once
      const int per_xcd = (gridDim.x + 7) >> 3, slot = blockIdx.x >> 3;
      const int lines = ep.warm_floats >> 5;
      float sink = 0.f;
      for (int ln = slot * 512 + tid; ln < lines; ln += per_xcd * 512) sink += ep.warm[(size_t)ln * 32];
      if (sink == 1.2345678e-30f) y[0] = sink;   // never true: the loads must not be elided
    }
  }
  ST_MARK(ST_SLOTS - 2);
  ST_MARK(ST_SLOTS - 1);
}

#ifdef VRP_STACK_TRACE
static void stack_trace_dump() {
  {
    static int calls = 0;
    if (++calls == 8) {   // a warm launch
      hipDeviceSynchronize();
      static unsigned long long h[512 * 8 * ST_SLOTS];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stack_trace), sizeof(h));
      for (int blk : {100}) {
        for (int wv = 0; wv < 8; ++wv) {
          const unsigned long long *t = h + ((size_t)blk * 8 + wv) * ST_SLOTS;
          const double mhz = 100.0 * (double)(t[ST_SLOTS - 2] - t[1]) / (double)(t[ST_SLOTS - 1] - t[0]);
          fprintf(stderr, "[stack trace] block %d wave %d: shader clock %.0f MHz, total %llu cyc\n", blk,
                  wv, mhz, t[ST_SLOTS - 2] - t[1]);
          fprintf(stderr, "  setup %llu = stage+zero %llu bar %llu graph %llu affines %llu bar %llu split %llu\n",
                  t[2] - t[1], t[76] - t[1], t[77] - t[76], t[78] - t[77], t[79] - t[78], t[80] - t[79], t[2] - t[80]);
          for (int l = 0; l < 3; ++l) {
            const unsigned long long *u = t + 4 + 24 * l;
            fprintf(stderr, "  L%d proj %llu bar %llu att %llu bar %llu out %llu bar %llu | up0 %llu bar %llu |", l,
                    u[1] - u[0], u[2] - u[1], u[3] - u[2], u[4] - u[3], u[5] - u[4], u[6] - u[5],
                    u[7] - u[6], u[8] - u[7]);
            for (int ch = 0; ch < 3; ++ch)
              fprintf(stderr, " dn %llu up %llu bar %llu |", u[9 + 3 * ch] - u[8 + 3 * ch],
                      u[10 + 3 * ch] - u[9 + 3 * ch], u[11 + 3 * ch] - u[10 + 3 * ch]);
            fprintf(stderr, " dn %llu bn2 %llu\n", u[21] - u[17], u[23] - u[21]);
          }
          fprintf(stderr, "  epilogue %llu\n", t[ST_SLOTS - 2] - t[ST_SLOTS - 3]);
        }
      }
    }
  }
}
#endif

template <int RT16>
static int launch_encoder_stack(const vrp_encoder_weights *w, const float *x, const float *norms,
                                float *y, int B, int N, const StackSetup &su, const StackEpilogue &ep,
                                hipStream_t st) {
  constexpr int RTW = 16 * RT16;
  const size_t lds = ((size_t)RTW * (2 * EB_LD + QA_QLD) + 16 * 384) * sizeof(float);
  static VrpAttrOnce attr_set;
  if (!attr_set.done() && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_stack_kernel<RT16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("encoder_stack: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  const int G = RTW / N;
  hipLaunchKernelGGL(encoder_stack_kernel<RT16>, dim3((B + G - 1) / G), dim3(512), lds, st, *w, x,
                     norms, y, B, N, G, su, ep);
  VRP_CHECK_LAUNCH("encoder_stack");
#ifdef VRP_STACK_TRACE
  stack_trace_dump();
#endif
  return 0;
}

// ---- the stack kernel on the bf16 matrix cores (round 5; encoder_x3.h) -----------------------
// Same decomposition -- G whole graphs per workgroup through all layers, eight waves, a wave owns
// 16 output columns of the block stages, one 48-column block of in_proj and one attention head --
// with every dense product as six bf16 MFMAs on three-plane operands:
//   * LDS holds the A operands as bf16 planes (layer input / y1; attention output and the even
//     hidden slices; the odd hidden slices share the q|k|v buffer): written once by the lane that
//     owns the accumulator element, read ready-made by all eight waves;
//   * the fp32 residual (layer input -> y1 -> layer output) never goes through LDS: a lane keeps
//     the twelve elements of its accumulator layout in registers across the layer;
//   * weights arrive as pre-split fragments (vrp_encoder_prepare), two fragment buffers alternate:
//     the fragment of stage i + 1 is requested when stage i starts;
//   * the attention itself (K = 16 per head) stays on the fp32 MFMA, q|k|v in fp32.
// Stage order per layer: P0 P1 P2 | attention | O | U0 | (D0 U1) (D1 U2) (D2 U3) | D3.
template <int RT16>
__global__ __launch_bounds__(512) void encoder_stack_x3_kernel(vrp_encoder_weights w,
                                                                const float *__restrict__ x,
                                                                const float *__restrict__ norms_in,
                                                                float *__restrict__ y, int B, int N,
                                                                int G, StackSetup su, StackEpilogue ep) {
  constexpr int RTW = 16 * RT16, PE = RTW * X3_PITCH;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16 *XB3 = reinterpret_cast<__bf16 *>(smem);    // [3][RTW][X3_PITCH]  layer input, then y1
  __bf16 *AT3 = XB3 + 3 * PE;                        // attention output, even hidden slices
  float *Q_s = reinterpret_cast<float *>(AT3 + 3 * PE);   // [RTW][QA_QLD] q | k | v (fp32)
  __bf16 *H1 = reinterpret_cast<__bf16 *>(Q_s);      // odd hidden slices (q|k|v are dead by then)
  float *stage = Q_s;                                // [RTW][EB_LD] fp32 rows: input, final output
  float *norm_s = Q_s + RTW * QA_QLD;                // [2 L][384] eval-mode BN affines (from_env)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int g0 = blockIdx.x * G;
  const int graphs = min(G, B - g0);
  const int rows = graphs * N;
  const size_t row0 = (size_t)g0 * N;
  const float *norms = norm_s;
  ST_MARK(0);
  ST_MARK(1);
  const int hidden = w.hidden, nchunk = hidden / 128, per_layer = x3_layer_frags(hidden);
  const __bf16 *split = reinterpret_cast<const __bf16 *>(w.split);
  Frag3 fa, fb;
  x3_load_frag(fa, split + (size_t)x3_frag_win(wave * 3) * X3_FRAG, lane);
  // rollout set-up fused in front: EVERY global load of it first -- the wave's graph (or its
  // part of one) and the thread's share of the BN statistics -- so that one round trip, not
  // three in a row, separates the launch from the first MFMA (round 6 trace: 11 k of the set-up's
  // 17 k cycles sat between the first barrier and the affines)
  const int su_parts = graphs >= 8 ? 1 : 8 / max(graphs, 1);   // waves per graph
  SetupLoads sl;
  float bnraw[2][4];
  if (su.from_env) {
    // the BN statistics FIRST: which layer a thread's element belongs to is wave-uniform (64
    // consecutive threads share i >> 7), so the four array pointers are scalar loads from the
    // kernel arguments -- indexed per thread, the compiler fetched the pointers themselves with
    // vector loads and the values behind them: two dependent round trips behind the graph's
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + 512 * u;
      const int blk = __builtin_amdgcn_readfirstlane(i >> 7);
      if (blk < 2 * w.num_layers) {
        const int cc = i & 127, l = blk >> 1, second = blk & 1;
        const vrp_encoder_layer &L = w.layer[l];
        bnraw[u][0] = (second ? L.bn2_running_mean : L.bn1_running_mean)[cc];
        bnraw[u][1] = (second ? L.bn2_running_var : L.bn1_running_var)[cc];
        bnraw[u][2] = (second ? L.bn2_weight : L.bn1_weight)[cc];
        bnraw[u][3] = (second ? L.bn2_bias : L.bn1_bias)[cc];
      }
    }
    if (wave < graphs * su_parts) sl = setup_graph_load(su.env, w, g0 + wave / su_parts, lane);
  }
  for (int idx = tid; idx < RTW * 32; idx += 512) {
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!su.from_env && r < rows) v = *reinterpret_cast<const float4 *>(x + (row0 + r) * VRP_EMB + c4);
    *reinterpret_cast<float4 *>(stage + r * EB_LD + c4) = v;
  }
  // the attention writes only the rows of real nodes: the others must be finite from the start
  // (they are multiplied by zero weights downstream, and 0 x NaN is NaN)
  for (int i = tid; i < 3 * PE / 8; i += 512)
    reinterpret_cast<float4 *>(AT3)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  ST_MARK(76);
  if (su.from_env) {
    __syncthreads();
    ST_MARK(77);
    if (blockIdx.x == 0 && tid < su.nflags) su.notdone[tid] = 0;
    {
      const int parts = su_parts;
      for (int gw = wave; gw < graphs * parts; gw += 8) {
        const int g = gw / parts, part = gw - g * parts;
        if (gw != wave) sl = setup_graph_load(su.env, w, g0 + g, lane);   // (more than eight graphs)
        setup_graph_finish(su.env, w, g0 + g, lane, stage + g * N * EB_LD, EB_LD, su.acc_loss,
                           su.acc_logp, sl, part, parts);
      }
    }
    ST_MARK(78);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + 512 * u;
      if (i < 2 * w.num_layers * 128) {
        float *o = norm_s + (i >> 7) * 384;
        const int cc = i & 127;
        o[cc] = bnraw[u][0];
        o[128 + cc] = bnraw[u][2] / sqrtf(bnraw[u][1] + 1e-5f);
        o[256 + cc] = bnraw[u][3];
      }
    }
    // (more than four layers: the x3 stack kernel takes at most five -- 1280 statistics, the third
    // pass over them the old way)
    for (int i = tid + 1024; i < 2 * w.num_layers * 128; i += 512) {
      const int blk = i >> 7, cc = i & 127, l = blk >> 1, second = blk & 1;
      const vrp_encoder_layer &L = w.layer[l];
      float *o = norm_s + blk * 384;
      o[cc] = (second ? L.bn2_running_mean : L.bn1_running_mean)[cc];
      o[128 + cc] = (second ? L.bn2_weight : L.bn1_weight)[cc] /
                    sqrtf((second ? L.bn2_running_var : L.bn1_running_var)[cc] + 1e-5f);
      o[256 + cc] = (second ? L.bn2_bias : L.bn1_bias)[cc];
    }
  } else {
    // (the affines always sit in LDS: a pointer that is LDS on one path and global memory on the
    // other makes every access a FLAT load, and a pending FLAT load turns every counted LDS wait
    // of the stage it is issued in into lgkmcnt(0))
    for (int i = tid; i < 2 * w.num_layers * 384; i += 512) norm_s[i] = norms_in[i];
  }
  ST_MARK(79);
  __syncthreads();
  ST_MARK(80);
  // the layer input in this lane's accumulator layout (row = 16 rt + i16, columns cq .. cq + 3):
  // kept in registers as the residual, and split into the A-operand planes
  const int cq = wave * 16 + 4 * q;
  float4 xres[RT16];
#pragma unroll
  for (int rt = 0; rt < RT16; ++rt) {
    const int row = rt * 16 + i16;
    xres[rt] = x3_ld4(stage + row * EB_LD + cq);
    x3_store4v(XB3, PE, row, cq, xres[rt]);
  }
  ST_MARK(2);
  __syncthreads();
  f32x4v acc[RT16], gacc[RT16];
  auto zero = [&](f32x4v (&a)[RT16]) {
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt) a[rt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  };
  // One layer.  Two fragment buffers with FIXED roles (round 6): on entry A holds the layer's first
  // in_proj fragment.  Stage -> buffer it reads / fragment requested piecewise under it:
  //   in_proj 0: A / B <- in_proj 1      in_proj 1: B / A <- in_proj 2     in_proj 2: A / B <- Wo
  //   attention: (A <- ff.0 slice 0, twelve loads at once: the vector memory pipe is idle here)
  //   out-proj: B / -      up 0: A / B <- ff.0 slice 1
  //   up ch+1: B / A <- ff.2 slice ch    down ch: A / B <- ff.0 slice ch+2 (last pass: the LAST ff.2 slice)
  //   last down: B / A <- the next layer's first in_proj fragment
  // Every fragment travels under the stage right before the one that uses it, none at a boundary.
  auto layer = [&](int l, Frag3 &A, Frag3 &B) {
    const vrp_encoder_layer &L = w.layer[l];
    const __bf16 *lf = split + (size_t)l * per_layer * X3_FRAG;
    ST_MARK(4 + 24 * l);
    // ---- in_proj: three 16-column tiles of this wave's 48-column block -> q | k | v (fp32) ----
    auto proj_tile = [&](int ct, const Frag3 &f, Frag3 &nf, const __bf16 *nsrc) {
      const int col0 = wave * 48 + ct * 16 + 4 * q;
      const float4 bb = x3_ld4(L.in_proj_bias + col0);
      zero(acc);
      x3_mma<RT16>(acc, XB3, PE, f, lane, X3FragStream(nf, nsrc, lane));
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt)
        *reinterpret_cast<float4 *>(Q_s + (rt * 16 + i16) * QA_QLD + col0) =
            make_float4(acc[rt][0] + bb.x, acc[rt][1] + bb.y, acc[rt][2] + bb.z, acc[rt][3] + bb.w);
    };
    // (every fragment but two is requested piecewise under the MFMAs of the stage before the one
    // that uses it, into the buffer that stage does not read: X3FragStream)
    proj_tile(0, A, B, lf + (size_t)x3_frag_win(wave * 3 + 1) * X3_FRAG);
    proj_tile(1, B, A, lf + (size_t)x3_frag_win(wave * 3 + 2) * X3_FRAG);
    proj_tile(2, A, B, lf + (size_t)x3_frag_wo(wave) * X3_FRAG);
    x3_load_frag(A, lf + (size_t)x3_frag_w1(wave) * X3_FRAG, lane);   // arrives during the attention
    ST_MARK(4 + 24 * l + 1);
    __syncthreads();
    ST_MARK(4 + 24 * l + 2);
    {
      const AttSinkX3 sink{AT3, PE};
#ifdef VRP_STACK_ATT_MFMA   // round 5's matrix-core form (A/B aid)
      if (N <= 16) qa8_stage_attention_mfma_pairs<1>(Q_s, N, graphs, RTW - 1, sink, lane, wave);
      else if (N <= 32) qa8_stage_attention_mfma_pairs<2>(Q_s, N, graphs, RTW - 1, sink, lane, wave);
      else qa8_stage_attention_mfma_to<3>(Q_s, N, graphs, RTW - 1, sink, lane, wave);
#else
      qa8_stage_attention_valu(Q_s, N, rows, sink, lane, wave);
#endif
    }
    ST_MARK(4 + 24 * l + 3);
    __syncthreads();
    ST_MARK(4 + 24 * l + 4);
    // ---- y1 = BN1(x + att Wo^T + bo) -----------------------------------------------------------
    {
      const float *n1 = norms + (2 * l) * 384;
      const float4 bb = x3_ld4(L.out_proj_bias + cq), mean = x3_ld4(n1 + cq),
                   mult = x3_ld4(n1 + 128 + cq), beta = x3_ld4(n1 + 256 + cq);
      zero(acc);
      x3_mma<RT16>(acc, AT3, PE, B, lane);
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt) {
        xres[rt].x = (acc[rt][0] + bb.x + xres[rt].x - mean.x) * mult.x + beta.x;
        xres[rt].y = (acc[rt][1] + bb.y + xres[rt].y - mean.y) * mult.y + beta.y;
        xres[rt].z = (acc[rt][2] + bb.z + xres[rt].z - mean.z) * mult.z + beta.z;
        xres[rt].w = (acc[rt][3] + bb.w + xres[rt].w - mean.w) * mult.w + beta.w;
        x3_store4v(XB3, PE, rt * 16 + i16, cq, xres[rt]);
      }
    }
    ST_MARK(4 + 24 * l + 5);
    __syncthreads();
    ST_MARK(4 + 24 * l + 6);
    // ---- g = sum over 128-wide hidden slices of relu(y1 W1c^T + b1c) W2c^T ----------------------
    // Slice ch + 1 goes up (MFMAs) BEFORE slice ch comes down: its epilogue -- ReLU, split, plane
    // stores -- is then issued between the MFMAs of the way down.
    auto up_store = [&](int rt, const float4 &b1v, __bf16 *hb) {
      x3_store4(hb, PE, rt * 16 + i16, cq, fmaxf(acc[rt][0] + b1v.x, 0.f), fmaxf(acc[rt][1] + b1v.y, 0.f),
                fmaxf(acc[rt][2] + b1v.z, 0.f), fmaxf(acc[rt][3] + b1v.w, 0.f));
    };
    zero(gacc);
    {
      const float4 b1v = x3_ld4(L.ff0_bias + cq);
      zero(acc);
      // (slice 0 goes up on A; B -- free since the out-projection -- takes ff.0 slice 1 under it, so
      // that NO fragment of the feed-forward is requested at a stage boundary: round 6)
      x3_mma<RT16>(acc, XB3, PE, A, lane,
                   X3FragStream(B, lf + (size_t)x3_frag_w1((nchunk > 1 ? 8 : 0) + wave) * X3_FRAG, lane));
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt) up_store(rt, b1v, AT3);
    }
    ST_MARK(4 + 24 * l + 7);
    __syncthreads();
    ST_MARK(4 + 24 * l + 8);
    for (int ch = 0; ch + 1 < nchunk; ++ch) {
      __bf16 *hcur = (ch & 1) ? H1 : AT3, *hnext = (ch & 1) ? AT3 : H1;
      const float4 b1v = x3_ld4(L.ff0_bias + (ch + 1) * 128 + cq);
      zero(acc);
      // up: slice ch + 1 (reads B); A -- free since the way up before -- takes the ff.2 slice of the
      // way down that follows
      x3_mma<RT16>(acc, XB3, PE, B, lane,
                   X3FragStream(A, lf + (size_t)x3_frag_w2(hidden, wave, ch) * X3_FRAG, lane));
      // B is free: the next ff.0 slice, or the LAST ff.2 slice, travels under the way down
      const X3FragStream sa(B, (ch + 2 < nchunk) ? lf + (size_t)x3_frag_w1((ch + 2) * 8 + wave) * X3_FRAG
                                                 : lf + (size_t)x3_frag_w2(hidden, wave, ch + 1) * X3_FRAG, lane);
      ST_MARK(4 + 24 * l + 9 + 3 * ch);
      x3_mma<RT16>(gacc, hcur, PE, A, lane, [&](int it) {                    // down: slice ch
        sa(it);
        if (it < RT16) up_store(it, b1v, hnext);
      });
      ST_MARK(4 + 24 * l + 10 + 3 * ch);
      __syncthreads();
      ST_MARK(4 + 24 * l + 11 + 3 * ch);
    }
    ST_MARK(4 + 24 * l + 17);
    {
      const float *n2 = norms + (2 * l + 1) * 384;
      const float4 bb = x3_ld4(L.ff2_bias + cq), mean = x3_ld4(n2 + cq),
                   mult = x3_ld4(n2 + 128 + cq), beta = x3_ld4(n2 + 256 + cq);
      // (hidden >= 256: see the loop)  The next layer's first in_proj fragment travels under it
      // (the last layer requests its own first fragment again: twelve loads nobody uses, cheaper
      // than a branch inside the MFMA sequence)
      x3_mma<RT16>(gacc, ((nchunk - 1) & 1) ? H1 : AT3, PE, B, lane,
                   X3FragStream(A, lf + (size_t)(l + 1 < w.num_layers ? per_layer + x3_frag_win(wave * 3) : 0) * X3_FRAG, lane));
      ST_MARK(4 + 24 * l + 21);
      // ---- y = BN2(y1 + g + b2): the next layer's input ----------------------------------------
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt) {
        xres[rt].x = (gacc[rt][0] + bb.x + xres[rt].x - mean.x) * mult.x + beta.x;
        xres[rt].y = (gacc[rt][1] + bb.y + xres[rt].y - mean.y) * mult.y + beta.y;
        xres[rt].z = (gacc[rt][2] + bb.z + xres[rt].z - mean.z) * mult.z + beta.z;
        xres[rt].w = (gacc[rt][3] + bb.w + xres[rt].w - mean.w) * mult.w + beta.w;
        if (l + 1 < w.num_layers) x3_store4v(XB3, PE, rt * 16 + i16, cq, xres[rt]);
      }
    }
    ST_MARK(4 + 24 * l + 23);
    __syncthreads();
  };
  for (int l = 0; l < w.num_layers; ++l) layer(l, fa, fb);   // (one copy of the layer's code: 41 -> 25 KB)
  ST_MARK(ST_SLOTS - 3);
  // ---- result: through the fp32 staging rows, coalesced 16-byte stores; the decoder's per-graph
  // constants on the way
#pragma unroll
  for (int rt = 0; rt < RT16; ++rt)
    *reinterpret_cast<float4 *>(stage + (rt * 16 + i16) * EB_LD + cq) = xres[rt];
  __syncthreads();
  for (int idx = tid; idx < rows * 32; idx += 512) {
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    *reinterpret_cast<float4 *>(y + (row0 + r) * VRP_EMB + c4) =
        *reinterpret_cast<const float4 *>(stage + r * EB_LD + c4);
  }
  if (ep.g) {
    float *gs = reinterpret_cast<float *>(XB3);   // [graphs][128]: the means once more, for QG below
    for (int i = tid; i < graphs * 128; i += 512) {
      const int g = i >> 7, cc = i & 127;
      float s = 0.f;
      for (int n = 0; n < N; ++n) s += stage[(g * N + n) * EB_LD + cc];
      ep.g[(size_t)(g0 + g) * VRP_EMB + cc] = s / (float)N;
      gs[i] = s / (float)N;
    }
    const float2 m = reinterpret_cast<const float2 *>(ep.mb)[lane];
    for (int r = wave; r < rows; r += 8) {
      const float2 ev = *reinterpret_cast<const float2 *>(stage + r * EB_LD + 2 * lane);
      const float s = wave_sum(fmaf(ev.x, m.x, ev.y * m.y));
      if (lane == 0) ep.cvec[row0 + r] = s;
    }
    for (int i = tid; i < graphs * 2 * N; i += 512) {
      const int g = i / (2 * N), t = i - g * 2 * N;
      ep.hist[(size_t)t * B + g0 + g] = 0ull;
    }
    if (blockIdx.x == 0 && tid == 0) *ep.err = 0;
    if (ep.QG) {   // (workgroup-uniform)
      __syncthreads();
      stack_epilogue_qg(ep, gs, gs + 16 * 128, g0, graphs, tid);   // (XB3's 36 KB: means, then 24 KB of partials)
    }
    if (ep.warm) {
      const int per_xcd = (gridDim.x + 7) >> 3, slot = blockIdx.x >> 3;
      const int lines = ep.warm_floats >> 5;
      float sink = 0.f;
      for (int ln = slot * 512 + tid; ln < lines; ln += per_xcd * 512) sink += ep.warm[(size_t)ln * 32];
      if (sink == 1.2345678e-30f) y[0] = sink;   // never true: the loads must not be elided
    }
  }
  ST_MARK(ST_SLOTS - 2);
  ST_MARK(ST_SLOTS - 1);
}

// the x3 stack kernel: pre-split weights present, at most five layers (LDS: 2 x 36 KB of operand
// planes + 73 KB q|k|v + 3 KB of BN affines per layer <= 160 KB), VRP_ENCODER_FP32=1 = A/B aid
static bool encoder_x3_enabled(const vrp_encoder_weights *w) {
  static const bool off = getenv("VRP_ENCODER_FP32") != nullptr;
  return !off && w->split != nullptr;
}
// the large-batch x3 kernels index their rows with 32-bit element offsets
static bool encoder_x3_rows_ok(long rows) { return rows * 128 < (1l << 31); }
template <int RT16>
static int launch_encoder_stack_x3(const vrp_encoder_weights *w, const float *x, const float *norms,
                                   float *y, int B, int N, const StackSetup &su,
                                   const StackEpilogue &ep, hipStream_t st) {
  constexpr int RTW = 16 * RT16;
  const size_t lds = (size_t)2 * 3 * RTW * X3_PITCH * 2 + (size_t)RTW * QA_QLD * 4 +
                     (size_t)2 * w->num_layers * 384 * 4;
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_stack_x3_kernel<RT16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      vrp_set_error("encoder_stack_x3: cannot raise dynamic LDS to 160 KB");
      return 1;
    }
    attr_set.mark();
  }
  const int G = RTW / N;
  hipLaunchKernelGGL(encoder_stack_x3_kernel<RT16>, dim3((B + G - 1) / G), dim3(512), lds, st, *w, x,
                     norms, y, B, N, G, su, ep);
  VRP_CHECK_LAUNCH("encoder_stack_x3");
#ifdef VRP_STACK_TRACE
  stack_trace_dump();
#endif
  return 0;
}
static bool encoder_stack_x3_applies(const vrp_encoder_weights *w) {
  return encoder_x3_enabled(w) && w->num_layers <= 5 && w->hidden >= 256;
}

// ---- out-proj + BN1 + FF + BN2 for LARGE row counts on the bf16 matrix cores -----------------
// encoder_block8_kernel's scheme (persistent workgroups, a wave owns 16 output columns, hidden
// slices alternate between two buffers, the next tile's attention rows travel behind the last
// stage) on 64-row tiles of three-plane operands: 3 buffers x 3 planes x 64 rows x 272 B = 153 KB.
// The attention rows arrive in fp32 and are split by the thread that stores them into LDS; the
// residual and y1 stay in the owning lane's registers.  Nine stages per tile, two fragment buffers
// that swap roles from tile to tile (the body is instantiated for both assignments).
template <int RT16>
__global__ __launch_bounds__(512) void encoder_block8_x3_kernel(
    const float *__restrict__ att, const float *__restrict__ x, const __bf16 *__restrict__ lf_,
    const float *__restrict__ bo, const float *__restrict__ norm1, const float *__restrict__ b1,
    const float *__restrict__ b2, const float *__restrict__ norm2, float *__restrict__ y, int rows,
    int hidden, int ntiles) {
  constexpr int RTW = 16 * RT16, PE = RTW * X3_PITCH, PF = RTW * 32 / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16 *hb0 = reinterpret_cast<__bf16 *>(smem);   // attention tile / hidden slices (roles alternate)
  __bf16 *hb1 = hb0 + 3 * PE;
  __bf16 *XB3 = hb1 + 3 * PE;                        // y1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  const int cq = wave * 16 + 4 * q;                  // this lane's four output columns
  const int nchunk = hidden / 128;
  float4 pa[PF];
  // (UNCONDITIONAL loads, the row index clamped and rows beyond the matrix zeroed on arrival: behind
  // a branch the compiler cannot count the requests in flight across the join and drains them
  // all -- round 6 trace: 4 k cycles for this fetch, 3 k for the residual rows below, per tile)
  auto fetch_att = [&](int tile) {
    const int row0 = tile * RTW;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      pa[u] = *reinterpret_cast<const float4 *>(att + (min(row0 + r, rows - 1) * 128 + c4));   // (32-bit offsets: rows * 128 < 2^31)
    }
  };
  auto store_att = [&](__bf16 *dst, int tile) {
    const int row0 = tile * RTW;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      x3_store4v(dst, PE, r, c4, row0 + r < rows ? pa[u] : make_float4(0.f, 0.f, 0.f, 0.f));
    }
  };
  int tile = blockIdx.x;
  __bf16 *abuf = hb0, *other = hb1;
  if (tile < ntiles) { fetch_att(tile); store_att(abuf, tile); }
  // fragment buffers: fa = out_proj and the ff.2 slices, fb = the ff.0 slices
  Frag3 fa, fb;
  x3_load_frag(fa, lf_ + (size_t)x3_frag_wo(wave) * X3_FRAG, lane);
  x3_load_frag(fb, lf_ + (size_t)x3_frag_w1(wave) * X3_FRAG, lane);
  __syncthreads();
  while (tile < ntiles) {
    // (the weights do not change from tile to tile; an opaque zero keeps the compiler from
    // hoisting the nine fragment loads out of this loop into registers it does not have)
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    const __bf16 *lf = lf_ + zero;
    // (bias / BatchNorm constants of this lane's four columns: 32 registers if kept across the
    // tile loop -- the kernel then spills, and a spill reload waits for every load in flight, the
    // prefetched weight fragment among them; fetched where they are used instead, from L1)
    const float *boz = bo + zero, *n1z = norm1 + zero, *b2z = b2 + zero, *n2z = norm2 + zero;
    const int row0 = tile * RTW, valid = rows - row0;
#ifdef VRP_STACK_TRACE
    const bool tr_ = tile == (int)(blockIdx.x + gridDim.x);
#define B8_MARK(i) if (tr_) { ST_MARK(i); }
#else
#define B8_MARK(i)
#endif
    B8_MARK(0); B8_MARK(1);
    // the residual rows in this lane's accumulator layout: row 16 rt + i16, columns cq .. cq + 3
    float4 xr[RT16];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt) {
      const int row = rt * 16 + i16;
      xr[rt] = x3_ld4(x + (min(row0 + row, rows - 1) * 128 + cq));   // (rows beyond the matrix: never stored)
    }
    f32x4v acc[RT16], gacc[RT16];
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt) { acc[rt] = f32x4v{0.f, 0.f, 0.f, 0.f}; gacc[rt] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
    // ---- y1 = BN1(x + att Wo^T + bo) ----------------------------------------------------------
    const float4 bb_o = x3_ld4(boz + cq), mean1 = x3_ld4(n1z + cq), mult1 = x3_ld4(n1z + 128 + cq),
                 beta1 = x3_ld4(n1z + 256 + cq);   // (requested ahead of the MFMAs that hide them)
    // (fb -- ff.0 slice 0 -- travels under the out-projection; every other fragment but one
    // under the stage before the one that uses it: X3FragStream, encoder_x3.h)
    x3_mma<RT16>(acc, abuf, PE, fa, lane, X3FragStream(fb, lf + (size_t)x3_frag_w1(wave) * X3_FRAG, lane));
    B8_MARK(2);
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt) {
      xr[rt].x = (acc[rt][0] + bb_o.x + xr[rt].x - mean1.x) * mult1.x + beta1.x;
      xr[rt].y = (acc[rt][1] + bb_o.y + xr[rt].y - mean1.y) * mult1.y + beta1.y;
      xr[rt].z = (acc[rt][2] + bb_o.z + xr[rt].z - mean1.z) * mult1.z + beta1.z;
      xr[rt].w = (acc[rt][3] + bb_o.w + xr[rt].w - mean1.w) * mult1.w + beta1.w;
      x3_store4v(XB3, PE, rt * 16 + i16, cq, xr[rt]);
    }
    B8_MARK(3);
    __syncthreads();
    B8_MARK(4);
    // ---- hidden slices: slice ch + 1 goes up BEFORE slice ch comes down; its epilogue (ReLU,
    // split, plane stores) is issued between the MFMAs of the way down --------------------------
    auto up_store = [&](int rt, const float4 &b1v, __bf16 *hb) {
      x3_store4(hb, PE, rt * 16 + i16, cq, fmaxf(acc[rt][0] + b1v.x, 0.f), fmaxf(acc[rt][1] + b1v.y, 0.f),
                fmaxf(acc[rt][2] + b1v.z, 0.f), fmaxf(acc[rt][3] + b1v.w, 0.f));
    };
    {
      const float4 b1v = x3_ld4(b1 + cq);
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt) acc[rt] = f32x4v{0.f, 0.f, 0.f, 0.f};
      x3_mma<RT16>(acc, XB3, PE, fb, lane);
      if (nchunk > 1) x3_load_frag(fb, lf + (size_t)x3_frag_w1(8 + wave) * X3_FRAG, lane);
      B8_MARK(5);
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt) up_store(rt, b1v, abuf);   // the attention rows are dead by now
    }
    B8_MARK(6);
    __syncthreads();
    B8_MARK(7);
    for (int ch = 0; ch + 1 < nchunk; ++ch) {
      __bf16 *hcur = (ch & 1) ? other : abuf, *hnext = (ch & 1) ? abuf : other;
      const float4 b1v = x3_ld4(b1 + (ch + 1) * 128 + cq);
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt) acc[rt] = f32x4v{0.f, 0.f, 0.f, 0.f};
      // up: slice ch + 1 (reads fb); fa -- free since the out-projection / the last way down --
      // takes the ff.2 slice of the way down that follows
      x3_mma<RT16>(acc, XB3, PE, fb, lane,
                   X3FragStream(fa, lf + (size_t)x3_frag_w2(hidden, wave, ch) * X3_FRAG, lane));
      // fb is free now.  Not the last pair: the next ff.0 slice.  The last pair: the LAST ff.2
      // slice -- the way down that follows the coming barrier has no way up in front of it to
      // hide a fragment requested after this pair's way down (measured: 4-8 k cycles exposed)
      const X3FragStream sb(fb, (ch + 2 < nchunk) ? lf + (size_t)x3_frag_w1((ch + 2) * 8 + wave) * X3_FRAG
                                                  : lf + (size_t)x3_frag_w2(hidden, wave, ch + 1) * X3_FRAG, lane);
      B8_MARK(8 + 3 * ch);
      x3_mma<RT16>(gacc, hcur, PE, fa, lane, [&](int it) {                    // down: slice ch
        sb(it);
        if (it < RT16) up_store(it, b1v, hnext);
      });
      B8_MARK(9 + 3 * ch);
      __syncthreads();
      B8_MARK(10 + 3 * ch);
    }
    // last way down; the next tile's attention rows travel behind it into the free buffer
    __bf16 *hlast = ((nchunk - 1) & 1) ? other : abuf, *hfree = ((nchunk - 1) & 1) ? abuf : other;
    const int next = tile + gridDim.x;
    fetch_att(min(next, ntiles - 1));   // (no branch around the loads; the last tile fetches itself again)
    const float4 bb_2 = x3_ld4(b2z + cq), mean2 = x3_ld4(n2z + cq), mult2 = x3_ld4(n2z + 128 + cq),
                 beta2 = x3_ld4(n2z + 256 + cq);   // (requested ahead of the MFMAs that hide them)
    B8_MARK(20);
    // the last ff.2 slice sits in fb (see the loop; hidden >= 256); the next tile's Wo travels under it
    x3_mma<RT16>(gacc, hlast, PE, fb, lane, X3FragStream(fa, lf + (size_t)x3_frag_wo(wave) * X3_FRAG, lane));
    B8_MARK(21);
    if (next < ntiles) store_att(hfree, next);
    B8_MARK(22);
    // ---- y = BN2(y1 + g + b2) ------------------------------------------------------------------
#pragma unroll
    for (int rt = 0; rt < RT16; ++rt) {
      const int row = rt * 16 + i16;
      if (row < valid)
        *reinterpret_cast<float4 *>(y + ((row0 + row) * 128 + cq)) =
            make_float4((gacc[rt][0] + bb_2.x + xr[rt].x - mean2.x) * mult2.x + beta2.x,
                        (gacc[rt][1] + bb_2.y + xr[rt].y - mean2.y) * mult2.y + beta2.y,
                        (gacc[rt][2] + bb_2.z + xr[rt].z - mean2.z) * mult2.z + beta2.z,
                        (gacc[rt][3] + bb_2.w + xr[rt].w - mean2.w) * mult2.w + beta2.w);
    }
    B8_MARK(23);
    __syncthreads();   // next tile's attention rows complete, XB3 and hlast free
    B8_MARK(24);
    abuf = hfree; other = hlast;
    tile = next;
  }
}

template <int RT16>
static int launch_encoder_block8_x3(const float *att, const float *x, const vrp_encoder_weights *w,
                                    int l, const float *norm1, const float *norm2, float *y, int rows,
                                    hipStream_t st) {
  constexpr int RTW = 16 * RT16;
  const size_t lds = (size_t)3 * 3 * RTW * X3_PITCH * 2;
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_block8_x3_kernel<RT16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("encoder_block8_x3: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  const vrp_encoder_layer &L = w->layer[l];
  const __bf16 *lf = reinterpret_cast<const __bf16 *>(w->split) +
                     (size_t)l * x3_layer_frags(w->hidden) * X3_FRAG;
  const int ntiles = (rows + RTW - 1) / RTW;
  hipLaunchKernelGGL(encoder_block8_x3_kernel<RT16>, dim3(min(ntiles, 256)), dim3(512), lds, st, att, x,
                     lf, L.out_proj_bias, norm1, L.ff0_bias, L.ff2_bias, norm2, y, rows, w->hidden,
                     ntiles);
  VRP_CHECK_LAUNCH("encoder_block8_x3");
#ifdef VRP_STACK_TRACE
  {
    static int calls = 0;
    if (++calls == 5) {
      hipDeviceSynchronize();
      static unsigned long long h[512 * 8 * ST_SLOTS];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stack_trace), sizeof(h));
      for (int wv : {0, 5}) {
        const unsigned long long *t = h + ((size_t)100 * 8 + wv) * ST_SLOTS;
        fprintf(stderr, "[block8_x3 trace] block 100 wave %d (second tile), cycles: xr-load+O-mma %llu | O-epi %llu | bar %llu | U0-mma %llu | U0-epi %llu | bar %llu |",
                wv, t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6]);
        for (int ch = 0; ch < 3; ++ch)
          fprintf(stderr, " U%d-mma %llu D%d+fill %llu bar %llu |", ch + 1, t[8 + 3 * ch] - (ch ? t[10 + 3 * (ch - 1)] : t[7]),
                  ch, t[9 + 3 * ch] - t[8 + 3 * ch], t[10 + 3 * ch] - t[9 + 3 * ch]);
        fprintf(stderr, " fetch %llu | D3-mma %llu | store_att %llu | BN2-epi %llu | bar %llu | total %llu\n", t[20] - t[16],
                t[21] - t[20], t[22] - t[21], t[23] - t[22], t[24] - t[23], t[24] - t[1]);
      }
    }
  }
#endif
  return 0;
}

// ---- in_proj + attention of whole graphs, LARGE batches, in_proj on the bf16 matrix cores ------
// encoder_qkv_attn8_kernel with the projection as six bf16 MFMAs per product: the input tile is
// split by the threads that store it into LDS (three planes, overlaid by q|k|v once every wave has
// finished reading them), the three 16-column fragments of a wave's 48-column block alternate
// between two fragment buffers.  The attention (K = 16 per head) stays on the fp32 MFMA.
template <int RT16>
__global__ __launch_bounds__(512) void encoder_qkv_attn8_x3_kernel(const float *__restrict__ x,
                                                                    const __bf16 *__restrict__ lf_,
                                                                    const float *__restrict__ bin,
                                                                    float *__restrict__ att, int B,
                                                                    int N, int G, int ntiles) {
  constexpr int RTW = 16 * RT16, PE = RTW * X3_PITCH, PF = RTW * 32 / 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Q_s = smem;                                   // [RTW][QA_QLD] q | k | v (fp32)
  __bf16 *X3 = reinterpret_cast<__bf16 *>(smem);       // [3][RTW][X3_PITCH]: the input tile first
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, q = lane >> 4;
  float4 pf[PF];
  auto fetch = [&](int tile) {
    const int g0 = tile * G;
    const int rows = min(G, B - g0) * N;
    const int row0 = g0 * N;   // (32-bit offsets: B N 128 < 2^31, checked by the launcher)
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      // (unconditional, clamped: see encoder_block8_x3_kernel; rows past the tile's graphs are
      // projected but never read by the attention)
      pf[u] = *reinterpret_cast<const float4 *>(x + ((row0 + min(r, rows - 1)) * VRP_EMB + c4));
    }
  };
  // fragment buffers: fb holds column tile 1 for good, fa alternates between tiles 0 and 2
  Frag3 fa, fb;
  x3_load_frag(fa, lf_ + (size_t)x3_frag_win(wave * 3) * X3_FRAG, lane);
  x3_load_frag(fb, lf_ + (size_t)x3_frag_win(wave * 3 + 1) * X3_FRAG, lane);
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    int zero;   // (keeps the fragment loads inside the loop: see encoder_block8_x3_kernel)
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    const __bf16 *lf = lf_ + zero;
    const float *binz = bin + zero;   // (bias pieces fetched where they are used: 12 registers less)
    const int g0 = tile * G;
    const int graphs = min(G, B - g0);
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      x3_store4v(X3, PE, r, c4, pf[u]);
    }
    __syncthreads();
    fetch(min(tile + (int)gridDim.x, ntiles - 1));   // (no branch around the loads)
    f32x4v acc[3][RT16];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt) acc[ct][rt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    x3_mma<RT16>(acc[0], X3, PE, fa, lane);
    // (column tile 2 travels under tile 1's MFMAs, one 1 KB piece per item: X3FragStream)
    x3_mma<RT16>(acc[1], X3, PE, fb, lane, X3FragStream(fa, lf + (size_t)x3_frag_win(wave * 3 + 2) * X3_FRAG, lane));
    x3_mma<RT16>(acc[2], X3, PE, fa, lane);
    x3_load_frag(fa, lf + (size_t)x3_frag_win(wave * 3) * X3_FRAG, lane);   // the next tile's first
    __syncthreads();   // every wave is done with the input planes: q|k|v may overlay them
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
      const int col0 = wave * 48 + ct * 16 + 4 * q;
      const float4 bb = x3_ld4(binz + col0);
#pragma unroll
      for (int rt = 0; rt < RT16; ++rt)
        *reinterpret_cast<float4 *>(Q_s + (rt * 16 + i16) * QA_QLD + col0) =
            make_float4(acc[ct][rt][0] + bb.x, acc[ct][rt][1] + bb.y,
                        acc[ct][rt][2] + bb.z, acc[ct][rt][3] + bb.w);
    }
    __syncthreads();
    float *o = att + (size_t)g0 * N * VRP_EMB;
    if (N <= 32) qa8_stage_attention_mfma<2>(Q_s, N, graphs, RTW - 1, o, lane, wave, VRP_EMB);
    else if (N <= 48) qa8_stage_attention_mfma<3>(Q_s, N, graphs, RTW - 1, o, lane, wave, VRP_EMB);
    else qa8_stage_attention_mfma<4>(Q_s, N, graphs, RTW - 1, o, lane, wave, VRP_EMB);
    __syncthreads();   // everybody done with Q_s before the next tile's planes land on it
  }
}

static int launch_qkv_attn8_x3(const float *x, const vrp_encoder_weights *w, int l, float *att, int B,
                               int N, hipStream_t st) {
  constexpr int RT16 = 5, RTW = 80;
  const size_t lds = (size_t)RTW * QA_QLD * sizeof(float);
  static VrpAttrOnce attr_set;
  if (!attr_set.done()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&encoder_qkv_attn8_x3_kernel<RT16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vrp_set_error("qkv_attn8_x3: cannot raise dynamic LDS to %zu bytes", lds);
      return 1;
    }
    attr_set.mark();
  }
  const __bf16 *lf = reinterpret_cast<const __bf16 *>(w->split) +
                     (size_t)l * x3_layer_frags(w->hidden) * X3_FRAG;
  const int G = RTW / N, ntiles = (B + G - 1) / G;
  hipLaunchKernelGGL(encoder_qkv_attn8_x3_kernel<RT16>, dim3(min(ntiles, 256)), dim3(512), lds, st, x, lf,
                     w->layer[l].in_proj_bias, att, B, N, G, ntiles);
  VRP_CHECK_LAUNCH("encoder_qkv_attn8_x3");
  return 0;
}

// ---- 64 < N <= 102, eval mode: encoder_qkv_attn_graph_kernel with the projection on the bf16 planes
// (round 6).  The q|k|v rows of one graph fill the LDS (155 KB at N = 100), so there is no room for
// a plane image of the input tile: the fp32 tile sits where q|k|v rows 64.. go (as in the fp32
// kernel, projection in two row halves) and a lane splits ITS OWN operand -- the 8 values of row
// i16, k = 32 j + 8 q .. + 7 -- in registers, once per (row tile, chunk), for the wave's three
// column tiles.  The three weight fragments (144 registers) are reloaded per graph: they are dead
// during the attention, which needs the registers.  Same products, k order and accumulators as
// x3_mma; six MFMAs per 32 k instead of eight fp32 MFMAs at a sixteenth of the rate.
// HOLD: rows whose q|k|v would land on input rows other waves still read (rows >= 64) keep their
// results in registers until a barrier; the others are stored row tile by row tile.
template <int RTN, bool HOLD>
__device__ __forceinline__ void qag_project_x3(const float *X_s, float *Q_s, const Frag3 (&f)[3],
                                               const float4 (&bb)[3], int lane, int wave, int vrows) {
  const int i16 = lane & 15, q = lane >> 4;
  f32x4v acc[HOLD ? RTN : 1][3];
  auto store = [&](int rt, const f32x4v (&a)[3]) {
    if (rt * 16 + i16 < vrows) {
#pragma unroll
      for (int ct = 0; ct < 3; ++ct)   // D[column][row]: four consecutive columns of row i16
        *reinterpret_cast<float4 *>(Q_s + (rt * 16 + i16) * QA_QLD + wave * 48 + ct * 16 + 4 * q) =
            make_float4(a[ct][0] + bb[ct].x, a[ct][1] + bb[ct].y, a[ct][2] + bb[ct].z, a[ct][3] + bb[ct].w);
    }
  };
  // the lane's operand of item (rt, j), one item ahead of the MFMAs that consume it
  const float *xp = X_s + i16 * QA_XLD + 8 * q;
  float4 a0 = *reinterpret_cast<const float4 *>(xp), a1 = *reinterpret_cast<const float4 *>(xp + 4);
#pragma unroll
  for (int rt = 0; rt < RTN; ++rt) {
    f32x4v (&d)[3] = acc[HOLD ? rt : 0];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) d[ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x8[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const int nit = rt * 4 + j + 1;
      if (nit < 4 * RTN) {
        const float *np = xp + (nit >> 2) * 16 * QA_XLD + 32 * (nit & 3);
        a0 = *reinterpret_cast<const float4 *>(np);
        a1 = *reinterpret_cast<const float4 *>(np + 4);
      }
      bf16x8 h, m, l;
      x3_split8(x8, h, m, l);
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) d[ct] = X3_MFMA(f[ct].p[1][j], m, d[ct]);
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) d[ct] = X3_MFMA(f[ct].p[2][j], h, d[ct]);
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) d[ct] = X3_MFMA(f[ct].p[0][j], l, d[ct]);
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) d[ct] = X3_MFMA(f[ct].p[1][j], h, d[ct]);
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) d[ct] = X3_MFMA(f[ct].p[0][j], m, d[ct]);
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) d[ct] = X3_MFMA(f[ct].p[0][j], h, d[ct]);
    }
    if (!HOLD) store(rt, d);
  }
  if (HOLD) {
    __syncthreads();   // every wave has read its rows: q|k|v may land on the input tile
#pragma unroll
    for (int rt = 0; rt < RTN; ++rt) store(rt, acc[HOLD ? rt : 0]);
  }
}
template <int NT>
__global__ __launch_bounds__(512) void encoder_qkv_attn_graph_x3_kernel(const float *__restrict__ x,
                                                                         const __bf16 *__restrict__ lf_,
                                                                         const float *__restrict__ bin,
                                                                         float *__restrict__ att, int B,
                                                                         int N) {
  constexpr int RTW = 16 * NT, PF = RTW * 32 / 512;
  static_assert(NT >= 5 && NT <= 7, "64 < N <= 102");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Q_s = smem;
  float *X_s = smem + 64 * QA_QLD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4;
  float4 pf[PF];
  auto fetch = [&](int g) {
    const int row0 = g * N;   // (32-bit offsets: B N 128 < 2^31, checked by the launcher)
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      pf[u] = *reinterpret_cast<const float4 *>(x + ((row0 + min(r, N - 1)) * VRP_EMB + c4));   // (clamped, no branch)
    }
  };
  int g = blockIdx.x;
  if (g < B) fetch(g);
  for (; g < B; g += gridDim.x) {
    int zero;   // (keeps the fragment loads inside the loop: see encoder_block8_x3_kernel)
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    const __bf16 *lf = lf_ + zero;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int idx = tid + 512 * u, r = idx >> 5, c4 = (idx & 31) * 4;
      *reinterpret_cast<float4 *>(X_s + r * QA_XLD + c4) = pf[u];   // rows >= N: copies of the last row, never read back
    }
    {
      Frag3 f[3];
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) x3_load_frag(f[ct], lf + (size_t)x3_frag_win(wave * 3 + ct) * X3_FRAG, lane);
      float4 bb[3];
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) bb[ct] = x3_ld4(bin + wave * 48 + ct * 16 + 4 * q);
      __syncthreads();
      qag_project_x3<4, false>(X_s, Q_s, f, bb, lane, wave, 64);
      qag_project_x3<NT - 4, true>(X_s + 64 * QA_XLD, Q_s + 64 * QA_QLD, f, bb, lane, wave, N - 64);
    }
    __syncthreads();
    fetch(min(g + (int)gridDim.x, B - 1));   // (after the projection: its registers are free now; no branch around the loads)
    attention_rows_mfma<NT, QA_QLD, false>(Q_s, att + (size_t)g * N * VRP_EMB, N, lane, wave);
    __syncthreads();   // everybody done with Q_s before the next graph lands in X_s
  }
}
static int launch_qkv_attn_graph_x3(const float *x, const vrp_encoder_weights *w, int l, float *att,
                                    int B, int N, hipStream_t st) {
  const int NT = (N + 15) / 16;
  const size_t fl = (size_t)N * QA_QLD > (size_t)64 * QA_QLD + (size_t)16 * NT * QA_XLD
                        ? (size_t)N * QA_QLD : (size_t)64 * QA_QLD + (size_t)16 * NT * QA_XLD;
  const size_t lds = fl * sizeof(float);
  const void *fn = NT == 5 ? reinterpret_cast<const void *>(&encoder_qkv_attn_graph_x3_kernel<5>)
                 : NT == 6 ? reinterpret_cast<const void *>(&encoder_qkv_attn_graph_x3_kernel<6>)
                           : reinterpret_cast<const void *>(&encoder_qkv_attn_graph_x3_kernel<7>);
  static VrpAttrOnce attr_set[3];
  if (!attr_set[NT - 5].done()) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      vrp_set_error("qkv_attn_graph_x3: cannot raise dynamic LDS to 160 KB");
      return 1;
    }
    attr_set[NT - 5].mark();
  }
  const __bf16 *lf = reinterpret_cast<const __bf16 *>(w->split) +
                     (size_t)l * x3_layer_frags(w->hidden) * X3_FRAG;
  const float *bin = w->layer[l].in_proj_bias;
  const dim3 grid(min(B, 256)), block(512);
  if (NT == 5) hipLaunchKernelGGL(encoder_qkv_attn_graph_x3_kernel<5>, grid, block, lds, st, x, lf, bin, att, B, N);
  else if (NT == 6) hipLaunchKernelGGL(encoder_qkv_attn_graph_x3_kernel<6>, grid, block, lds, st, x, lf, bin, att, B, N);
  else hipLaunchKernelGGL(encoder_qkv_attn_graph_x3_kernel<7>, grid, block, lds, st, x, lf, bin, att, B, N);
  VRP_CHECK_LAUNCH("encoder_qkv_attn_graph_x3");
  return 0;
}

// small batches, eval mode: all layers in one launch, G = 48 / N whole graphs per workgroup
static bool encoder_stack_applies(const vrp_encoder_weights *w, int train, int B, int N) {
  static const char *stack_off = getenv("VRP_ENCODER_NO_STACK");  // A/B aid
  return !train && !stack_off && enc_heads(w) == 8 && N <= 48 && w->hidden % 128 == 0 && w->num_layers <= 8 &&
         (B + 48 / N - 1) / (48 / N) <= 512;
}

struct EncWs {
  float *h0, *h1, *qkv, *att, *ff, *norm;  // norm: (2 VRP_MAX_LAYERS, 384) eval-mode BN affines
  double *stats;
};

static EncWs carve_encoder(void *ws, int B, int N, int hidden) {
  const size_t R = (size_t)B * N;
  char *p = (char *)ws;
  EncWs w;
  w.h0 = (float *)p;   p += vrp_align_up(R * 128 * 4);
  w.h1 = (float *)p;   p += vrp_align_up(R * 128 * 4);
  w.qkv = (float *)p;  p += vrp_align_up(R * 384 * 4);
  w.att = (float *)p;  p += vrp_align_up(R * 128 * 4);
  w.ff = (float *)p;   p += vrp_align_up(R * (size_t)hidden * 4);
  w.norm = (float *)p; p += vrp_align_up(2 * VRP_MAX_LAYERS * 384 * 4);
  w.stats = (double *)p;
  return w;
}

extern "C" int64_t vrp_encoder_workspace_bytes(int B, int N, int hidden) {
  const size_t R = (size_t)B * N;
  // + feature scratch used by vrp_rollout: x (R,3) fp32 and is_depot (R) u8
  return (int64_t)(3 * vrp_align_up(R * 128 * 4) + vrp_align_up(R * 384 * 4) +
                   vrp_align_up(R * (size_t)hidden * 4) + vrp_align_up(2 * VRP_MAX_LAYERS * 384 * 4) +
                   bn_sums_bytes() + vrp_align_up(R * 12) + vrp_align_up(R));
}

// ---- split weights for the bf16-matrix-core kernels (encoder_x3.h) -----------------------------
extern "C" int64_t vrp_encoder_split_bytes(int hidden, int num_layers) {
  return (int64_t)x3_layer_frags(hidden) * num_layers * X3_FRAG * 2;
}
extern "C" int vrp_encoder_prepare(const vrp_encoder_weights *w, void *split, void *stream) {
  VRP_REQUIRE(w && split, "encoder_prepare: NULL argument");
  VRP_REQUIRE(w->hidden >= 128 && w->hidden % 128 == 0 && w->num_layers >= 1 && w->num_layers <= VRP_MAX_LAYERS,
              "encoder_prepare: hidden=%d layers=%d", w->hidden, w->num_layers);
  const int threads = x3_layer_frags(w->hidden) * w->num_layers * 256;
  hipLaunchKernelGGL(x3_prepare_kernel, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     *w, reinterpret_cast<__bf16 *>(split));
  VRP_CHECK_LAUNCH("encoder_prepare");
  return 0;
}

static int batchnorm_train(float *x, int rows, const float *w, const float *b, float *rm,
                           float *rv, int64_t *nbt, const EncWs &ws, hipStream_t st) {
  if (int r = launch_bn_stats(x, rows, ws.stats, st)) return r;
  const size_t n4 = (size_t)rows * 32;
  hipLaunchKernelGGL(bn_train_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                     x, n4, ws.stats, rows, w, b, rm, rv, nbt);
  VRP_CHECK_LAUNCH("bn_train_apply");
  return 0;
}

// Rollout set-up in ONE launch (vrp_rollout only): generate_mask on the fresh episode
// (tsp.py:106-148 via get_state), the network inputs of E3, the node/depot embedding
// (graph_encoder.py:54,110-132), the eval-mode BatchNorm affines and the zeroed episode
// accumulators -- five tiny kernels otherwise, each ~5 us of dependent-launch latency.
// One wave per graph; lane = node for the env part, lane = embedding column (x2) after.
__global__ __launch_bounds__(256) void rollout_setup_kernel(
    vrp_env e, vrp_encoder_weights w, int want_norms, float *__restrict__ out,
    float *__restrict__ norms, float *__restrict__ acc_loss, float *__restrict__ acc_logp,
    int32_t *__restrict__ notdone, int nflags, int env_blocks) {
  const int lane = threadIdx.x & 63;
  if ((int)blockIdx.x >= env_blocks) {  // eval-mode BN affines, one block per (layer, bn)
    if (!want_norms || threadIdx.x >= 128) return;
    const int blk = blockIdx.x - env_blocks, c = threadIdx.x, l = blk >> 1, second = blk & 1;
    const vrp_encoder_layer &L = w.layer[l];
    const float *rm = second ? L.bn2_running_mean : L.bn1_running_mean;
    const float *rv = second ? L.bn2_running_var : L.bn1_running_var;
    const float *wt = second ? L.bn2_weight : L.bn1_weight;
    const float *bs = second ? L.bn2_bias : L.bn1_bias;
    float *o = norms + (size_t)blk * 384;
    o[c] = rm[c];
    o[128 + c] = wt[c] / sqrtf(rv[c] + 1e-5f);
    o[256 + c] = bs[c];
    return;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < nflags) notdone[threadIdx.x] = 0;
  // one workgroup per graph, its four waves share the embedding rows (a wave per graph left two
  // waves per SIMD to write 512 N bytes each: 67 us at 2048 x 100, 75 us at 8192 x 40)
  const int b = blockIdx.x;
  setup_graph_wave(e, w, b, lane, out + (size_t)b * e.N * VRP_EMB, VRP_EMB, acc_loss, acc_logp,
                   threadIdx.x >> 6, 4);
}

static int encoder_check(const vrp_encoder_weights *w, int B, int N) {
  VRP_REQUIRE(B > 0 && N > 0 && N <= VRP_MAX_NODES, "encoder: bad shape B=%d N=%d", B, N);
  VRP_REQUIRE(w->num_layers >= 1 && w->num_layers <= VRP_MAX_LAYERS, "encoder: num_layers=%d", w->num_layers);
  VRP_REQUIRE(w->hidden % 128 == 0, "encoder: hidden=%d must be a multiple of 128", w->hidden);
  VRP_REQUIRE(enc_heads(w) == 8 || enc_heads(w) == 4 || enc_heads(w) == 16,
              "encoder: heads=%d (4, 8 or 16)", w->heads);
  VRP_REQUIRE(w->node_dim >= 1 && w->node_dim <= 3, "encoder: node_dim=%d", w->node_dim);
  return 0;
}

static int encoder_layers(const vrp_encoder_weights *w, int train, int B, int N, float *cur,
                          float *emb, const EncWs &ws, hipStream_t st);

extern "C" int vrp_encoder_forward(const vrp_encoder_weights *w, int train, int B, int N,
                                   const float *x, const uint8_t *depot_mask, float *emb,
                                   void *workspace, void *stream) {
  VRP_REQUIRE(w && x && emb && workspace, "encoder: NULL argument");
  if (int r = encoder_check(w, B, N)) return r;
  VRP_REQUIRE(!depot_mask || w->depot_embed_weight, "encoder: depot mask without depot_embed");
  hipStream_t st = (hipStream_t)stream;
  const int R = B * N;
  EncWs ws = carve_encoder(workspace, B, N, w->hidden);
  float *cur = (w->num_layers % 2 == 0) ? emb : ws.h0;  // so that the last layer lands in emb
  hipLaunchKernelGGL(embed_kernel, dim3((R + 1) / 2), dim3(256), 0, st, x, depot_mask,
                     w->node_embed_weight, w->node_embed_bias, w->node_dim,
                     w->depot_embed_weight, w->depot_embed_bias, w->depot_dim, cur, R);
  VRP_CHECK_LAUNCH("embed");
  if (!train) {
    hipLaunchKernelGGL(bn_eval_norms_kernel, dim3(2 * w->num_layers), dim3(128), 0, st, *w,
                       ws.norm);
    VRP_CHECK_LAUNCH("bn_eval_norms");
  }
  return encoder_layers(w, train, B, N, cur, emb, ws, st);
}

// vrp_rollout's entry: mask init + features + embedding + BN affines + accumulator reset in
// one launch (rollout_setup_kernel), then the attention layers.
int vrp_encoder_forward_from_env(const vrp_encoder_weights *w, int train, const vrp_env *env,
                                 float *emb, void *workspace, float *acc_loss, float *acc_logp,
                                 int32_t *notdone, int nflags, const float *dec_mb, float *dec_g,
                                 float *dec_cvec, unsigned long long *dec_hist, int32_t *dec_err,
                                 const float *dec_warm, int dec_warm_floats,
                                 const float *dec_wqgT, const float *dec_bq, float *dec_QG,
                                 int *decoder_constants_done, hipStream_t st) {
  const int B = env->B, N = env->N;
  if (int r = encoder_check(w, B, N)) return r;
  VRP_REQUIRE(env->kind == VRP_KIND_TSP || w->depot_embed_weight,
              "encoder: VRP/IRP rollout needs depot_embed");
  VRP_REQUIRE(nflags <= 256, "rollout: more than 255 steps");
  EncWs ws = carve_encoder(workspace, B, N, w->hidden);
  *decoder_constants_done = 0;
  if (encoder_stack_applies(w, train, B, N)) {
    // one launch: set-up, all layers, and the decoder's per-graph constants
    StackSetup su;
    su.env = *env; su.acc_loss = acc_loss; su.acc_logp = acc_logp; su.notdone = notdone;
    su.nflags = nflags; su.from_env = 1;
    static const bool no_warm = getenv("VRP_NO_WARM") != nullptr;  // A/B aid
    StackEpilogue ep = {dec_mb, dec_g, dec_cvec, dec_hist, dec_err, no_warm ? nullptr : dec_warm,
                        dec_warm_floats, nullptr, nullptr, nullptr};
    *decoder_constants_done = dec_g != nullptr;   // bit 0: graph mean, cvec, cleared hand-off words
    if (encoder_stack_x3_applies(w)) {
      static const bool no_qg = getenv("VRP_NO_STACK_QG") != nullptr;   // A/B aid
      if (dec_g && dec_QG && !no_qg) {
        ep.wqgT = dec_wqgT; ep.bq = dec_bq; ep.QG = dec_QG;
        *decoder_constants_done |= 2;             // bit 1: QG = Wq_g g + bq
      }
      return launch_encoder_stack_x3<3>(w, nullptr, nullptr, emb, B, N, su, ep, st);
    }
    return launch_encoder_stack<3>(w, nullptr, nullptr, emb, B, N, su, ep, st);
  }
  float *cur = (w->num_layers % 2 == 0) ? emb : ws.h0;
  const int env_blocks = B;
  hipLaunchKernelGGL(rollout_setup_kernel, dim3(env_blocks + 2 * w->num_layers), dim3(256), 0, st,
                     *env, *w, train ? 0 : 1, cur, ws.norm, acc_loss, acc_logp, notdone, nflags,
                     env_blocks);
  VRP_CHECK_LAUNCH("rollout_setup");
  return encoder_layers(w, train, B, N, cur, emb, ws, st);
}

// What the encoder phase of vrp_rollout launches for this shape (profiles, bench line).
extern "C" const char *vrp_encoder_kernel_name(const vrp_encoder_weights *w, int train, int B, int N) {
  if (!w) return "?";
  if (encoder_stack_applies(w, train, B, N))
    return encoder_stack_x3_applies(w) ? "encoder_stack_x3_kernel<3>" : "encoder_stack_kernel<3>";
  const int R = B * N;
  if (train) return "gemm_nt / gemm_rows + encoder_attention_mfma + bn_* per layer (train mode)";
  if (N <= 64 && (80 / N) * N * 4 >= 3 * 80 && R >= 256 * 80)
    return encoder_x3_enabled(w) ? "encoder_qkv_attn8_x3_kernel<5> + encoder_block8_x3_kernel<4> per layer"
                                 : "encoder_qkv_attn8_kernel<5> + encoder_block8_kernel<5> per layer";
  return "qkv/attention + encoder_block* per layer";
}

static int encoder_layers(const vrp_encoder_weights *w, int train, int B, int N, float *cur,
                          float *emb, const EncWs &ws, hipStream_t st) {
  const int R = B * N;
  float *nxt = nullptr;
  if (encoder_stack_applies(w, train, B, N)) {
    StackSetup su = {};
    StackEpilogue ep = {};
    if (encoder_stack_x3_applies(w)) return launch_encoder_stack_x3<3>(w, cur, ws.norm, emb, B, N, su, ep, st);
    return launch_encoder_stack<3>(w, cur, ws.norm, emb, B, N, su, ep, st);
  }
  for (int l = 0; l < w->num_layers; ++l) {
    const vrp_encoder_layer &L = w->layer[l];
    // out = bn1(x + MHA(x))
    static const char *qa_env = getenv("VRP_UNFUSED_QKV");  // A/B aid
    const int heads = enc_heads(w);
    // (four or sixteen heads: every fused kernel is built for eight -- GEMM + VALU attention)
    const bool qa_off = qa_env != nullptr || heads != 8;
    if (N <= 64 && R <= 16 * 1024 && !qa_off) {
      // small batches: in_proj + attention of a graph in one launch
      int r;
      if (N <= 32) r = launch_qkv_attention<2>(cur, L.in_proj_weight, L.in_proj_bias, ws.att, B, N, st);
      else if (N <= 48) r = launch_qkv_attention<3>(cur, L.in_proj_weight, L.in_proj_bias, ws.att, B, N, st);
      else r = launch_qkv_attention<4>(cur, L.in_proj_weight, L.in_proj_bias, ws.att, B, N, st);
      if (r) return r;
    } else if (!train && N <= 64 && (80 / N) * N * 4 >= 3 * 80 && !qa_off) {
      // large batches, eval mode: in_proj + attention of 80 / N whole graphs per workgroup
      // (only when the graphs fill at least three quarters of the five row tiles)
      if (int r = encoder_x3_enabled(w) && encoder_x3_rows_ok((long)B * N) ? launch_qkv_attn8_x3(cur, w, l, ws.att, B, N, st)
                                        : launch_qkv_attn8(cur, L.in_proj_weight, L.in_proj_bias, ws.att, B, N, st))
        return r;
    } else if (heads == 8 && qkv_attn_graph_applies(train, B, N)) {
      // 64 < N <= 102, eval mode: in_proj + attention of one graph per workgroup pass, q|k|v in LDS
      static const bool qag_fp32 = getenv("VRP_QAG_FP32") != nullptr;   // A/B aid
      if (int r = encoder_x3_enabled(w) && encoder_x3_rows_ok((long)B * N) && !qag_fp32
                      ? launch_qkv_attn_graph_x3(cur, w, l, ws.att, B, N, st)
                      : launch_qkv_attn_graph(cur, L.in_proj_weight, L.in_proj_bias, ws.att, B, N, st)) return r;
    } else {
      if (int r = vrp_launch_gemm_nt(cur, 128, L.in_proj_weight, 128, L.in_proj_bias, nullptr, 0,
                                     ws.qkv, 384, R, 384, 128, 0, st)) return r;
      static const char *valu_att = getenv("VRP_ATTENTION_VALU");  // A/B aid
      if (!valu_att && heads == 8) {
        if (int r = launch_attention_mfma(ws.qkv, ws.att, B, N, st)) return r;
      } else {
        if (int r = launch_attention_valu(ws.qkv, ws.att, B, N, heads, st)) return r;
      }
    }
    static const char *unfused = getenv("VRP_ENCODER_UNFUSED");  // A/B aid
    if (!train && !unfused) {
      // eval: out-proj + BN1 + FF + BN2 in one kernel, activations stay in LDS
      nxt = (cur == emb) ? ws.h0 : emb;
      const float *n1 = ws.norm + (2 * l) * 384, *n2 = ws.norm + (2 * l + 1) * 384;
      // row tile: enough workgroups to occupy all 256 CUs at every batch size
      int r;
      static const char *rtw_env = getenv("VRP_BLOCK_RTW");  // A/B aid: "64", "128" or "8"
      if (rtw_env && rtw_env[0] == '6')
        r = launch_encoder_block<64>(ws.att, cur, L, n1, n2, nxt, R, w->hidden, st);
      else if (rtw_env && rtw_env[0] == '1')
        r = launch_encoder_block<128>(ws.att, cur, L, n1, n2, nxt, R, w->hidden, st);
      else if ((rtw_env && rtw_env[0] == '8') || (!rtw_env && R >= 256 * 80))
        // >= one 80-row tile per CU: persistent 8-wave kernel (827 vs 914 us per layer for
        // the 64-row kernel at 8192 x 40)
        r = encoder_x3_enabled(w) && encoder_x3_rows_ok(R) && w->hidden >= 256
                ? launch_encoder_block8_x3<4>(ws.att, cur, w, l, n1, n2, nxt, R, st)
                                  : launch_encoder_block8<5>(ws.att, cur, L, n1, n2, nxt, R, w->hidden, st);
      else if (R > 16 * 1024)  // two 64-row workgroups per CU beat one of 128 rows
        r = launch_encoder_block<64>(ws.att, cur, L, n1, n2, nxt, R, w->hidden, st);
      else if (R > 12 * 1024)   // small batches: at most one workgroup per CU (256 CUs)
        r = launch_encoder_block16<4>(ws.att, cur, L, n1, n2, nxt, R, w->hidden, st);
      else if (R > 8 * 1024)
        r = launch_encoder_block16<3>(ws.att, cur, L, n1, n2, nxt, R, w->hidden, st);
      else
        r = launch_encoder_block16<2>(ws.att, cur, L, n1, n2, nxt, R, w->hidden, st);
      if (r) return r;
      cur = nxt;
      continue;
    }
    if (int r = vrp_launch_gemm_nt_ex(ws.att, 128, L.out_proj_weight, 128, L.out_proj_bias, cur,
                                      128, train ? nullptr : ws.norm + (2 * l) * 384, ws.h1, 128,
                                      R, 128, 128, 0, st)) return r;
    if (train)
      if (int r = batchnorm_train(ws.h1, R, L.bn1_weight, L.bn1_bias, L.bn1_running_mean,
                                  L.bn1_running_var, L.bn1_num_batches_tracked, ws, st)) return r;
    // out = bn2(out + FF(out))
    if (int r = vrp_launch_gemm_nt(ws.h1, 128, L.ff0_weight, 128, L.ff0_bias, nullptr, 0, ws.ff,
                                   w->hidden, R, w->hidden, 128, 1, st)) return r;
    nxt = (cur == emb) ? ws.h0 : emb;
    if (int r = vrp_launch_gemm_nt_ex(ws.ff, w->hidden, L.ff2_weight, w->hidden, L.ff2_bias,
                                      ws.h1, 128, train ? nullptr : ws.norm + (2 * l + 1) * 384,
                                      nxt, 128, R, 128, w->hidden, 0, st)) return r;
    if (train)
      if (int r = batchnorm_train(nxt, R, L.bn2_weight, L.bn2_bias, L.bn2_running_mean,
                                  L.bn2_running_var, L.bn2_num_batches_tracked, ws, st)) return r;
    cur = nxt;
  }
  if (cur != emb) {  // defensive: cannot happen with the parity choice above
    vrp_set_error("encoder: internal buffer parity error");
    return 3;
  }
  return 0;
}


// =========================================================================================
// Train-mode forward with a tape, and the backward pass of the encoder (K4).
// =========================================================================================

// out-of-place train-mode BatchNorm: y = (z - mean) * invstd * gamma + beta with the batch
// statistics in `sums`; writes stats = [mean | invstd] for the backward pass
__global__ __launch_bounds__(256) void bn_train_apply_oop_kernel(
    const float *__restrict__ z, float *__restrict__ y, size_t n4,
    const double *__restrict__ sums, int rows, const float *__restrict__ weight,
    const float *__restrict__ bias, float *__restrict__ stats, float *running_mean,
    float *running_var, int64_t *num_batches_tracked, int update_running) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 128) {
    const int c = threadIdx.x;
    const double m = sums[c] / rows;
    double v = sums[128 + c] / rows - m * m;
    if (v < 0.0) v = 0.0;
    stats[c] = (float)m;
    stats[128 + c] = 1.f / sqrtf((float)v + 1e-5f);
    if (update_running) {
      const float unbiased = (float)(v * ((double)rows / (double)(rows > 1 ? rows - 1 : 1)));
      running_mean[c] = 0.9f * running_mean[c] + 0.1f * (float)m;
      running_var[c] = 0.9f * running_var[c] + 0.1f * unbiased;
      if (c == 0) *num_batches_tracked += 1;
    }
  }
  if (i >= n4) return;
  const int c0 = (int)(i & 31) * 4;
  const float4 v = reinterpret_cast<const float4 *>(z)[i];
  float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j;
    const double m = sums[c] / rows;
    double var = sums[128 + c] / rows - m * m;
    if (var < 0.0) var = 0.0;
    o[j] = (o[j] - (float)m) * (weight[c] / sqrtf((float)var + 1e-5f)) + bias[c];
  }
  reinterpret_cast<float4 *>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
}

struct LayerTape {
  float *X, *QKV, *ATT, *Z1, *Y1, *H, *Z2, *stats1, *stats2;
};
struct EncTape {
  LayerTape layer[VRP_MAX_LAYERS];
  float *OUT;      // output of the last layer = emb copy is NOT kept; X of layer l+1 = out of l
  double *sums;    // 256 doubles: batch-statistic scratch
};

static size_t layer_tape_floats(size_t R, int hidden) {
  return R * (128 + 384 + 128 + 128 + 128 + (size_t)hidden + 128) + 512;
}

extern "C" int64_t vrp_encoder_tape_bytes(int B, int N, int hidden, int num_layers) {
  const size_t R = (size_t)B * N;
  return (int64_t)(num_layers * vrp_align_up(layer_tape_floats(R, hidden) * 4) + bn_sums_bytes());
}

static EncTape carve_tape(void *tape, int B, int N, int hidden, int num_layers) {
  const size_t R = (size_t)B * N;
  EncTape t;
  char *p = (char *)tape;
  for (int l = 0; l < num_layers; ++l) {
    float *f = (float *)p;
    LayerTape &L = t.layer[l];
    L.X = f;    f += R * 128;
    L.QKV = f;  f += R * 384;
    L.ATT = f;  f += R * 128;
    L.Z1 = f;   f += R * 128;
    L.Y1 = f;   f += R * 128;
    L.H = f;    f += R * (size_t)hidden;
    L.Z2 = f;   f += R * 128;
    L.stats1 = f; f += 256;
    L.stats2 = f; f += 256;
    p += vrp_align_up(layer_tape_floats(R, hidden) * 4);
  }
  t.sums = (double *)p;
  t.OUT = nullptr;
  return t;
}

static int bn_train_taped(const float *z, float *y, int rows, const float *w, const float *b,
                          float *stats, float *rm, float *rv, int64_t *nbt, int update,
                          double *sums, hipStream_t st) {
  if (int r = launch_bn_stats(z, rows, sums, st)) return r;
  const size_t n4 = (size_t)rows * 32;
  hipLaunchKernelGGL(bn_train_apply_oop_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                     z, y, n4, sums, rows, w, b, stats, rm, rv, nbt, update);
  VRP_CHECK_LAUNCH("bn_train_apply_oop");
  return 0;
}

extern "C" int vrp_encoder_forward_tape(const vrp_encoder_weights *w, int B, int N, const float *x,
                                        const uint8_t *depot_mask, float *emb, void *tape,
                                        int update_running, void *stream) {
  VRP_REQUIRE(w && x && emb && tape, "encoder_tape: NULL argument");
  VRP_REQUIRE(B > 0 && N > 0 && N <= VRP_MAX_NODES, "encoder_tape: bad shape B=%d N=%d", B, N);
  VRP_REQUIRE(enc_heads(w) == 8 || enc_heads(w) == 4 || enc_heads(w) == 16, "encoder: heads=%d", w->heads);
  VRP_REQUIRE(w->num_layers >= 1 && w->num_layers <= VRP_MAX_LAYERS && w->hidden % 128 == 0,
              "encoder_tape: unsupported architecture");
  hipStream_t st = (hipStream_t)stream;
  const int R = B * N, L_ = w->num_layers;
  EncTape t = carve_tape(tape, B, N, w->hidden, L_);
  hipLaunchKernelGGL(embed_kernel, dim3((R + 1) / 2), dim3(256), 0, st, x, depot_mask,
                     w->node_embed_weight, w->node_embed_bias, w->node_dim,
                     w->depot_embed_weight, w->depot_embed_bias, w->depot_dim, t.layer[0].X, R);
  VRP_CHECK_LAUNCH("embed");
  for (int l = 0; l < L_; ++l) {
    const vrp_encoder_layer &P = w->layer[l];
    const LayerTape &T = t.layer[l];
    float *out = (l + 1 < L_) ? t.layer[l + 1].X : emb;
    if (int r = vrp_launch_gemm_nt(T.X, 128, P.in_proj_weight, 128, P.in_proj_bias, nullptr, 0,
                                   T.QKV, 384, R, 384, 128, 0, st)) return r;
    static const char *valu_att = getenv("VRP_ATTENTION_VALU");  // A/B aid
    if (!valu_att && enc_heads(w) == 8) {
      if (int r = launch_attention_mfma(T.QKV, T.ATT, B, N, st)) return r;
    } else {
      if (int r = launch_attention_valu(T.QKV, T.ATT, B, N, enc_heads(w), st)) return r;
    }
    if (int r = vrp_launch_gemm_nt(T.ATT, 128, P.out_proj_weight, 128, P.out_proj_bias, T.X, 128,
                                   T.Z1, 128, R, 128, 128, 0, st)) return r;
    if (int r = bn_train_taped(T.Z1, T.Y1, R, P.bn1_weight, P.bn1_bias, T.stats1,
                               P.bn1_running_mean, P.bn1_running_var, P.bn1_num_batches_tracked,
                               update_running, t.sums, st)) return r;
    if (int r = vrp_launch_gemm_nt(T.Y1, 128, P.ff0_weight, 128, P.ff0_bias, nullptr, 0, T.H,
                                   w->hidden, R, w->hidden, 128, 1, st)) return r;
    if (int r = vrp_launch_gemm_nt(T.H, w->hidden, P.ff2_weight, w->hidden, P.ff2_bias, T.Y1, 128,
                                   T.Z2, 128, R, 128, w->hidden, 0, st)) return r;
    if (int r = bn_train_taped(T.Z2, out, R, P.bn2_weight, P.bn2_bias, T.stats2,
                               P.bn2_running_mean, P.bn2_running_var, P.bn2_num_batches_tracked,
                               update_running, t.sums, st)) return r;
  }
  return 0;
}

// node/depot embedding gradients: column c of dX0 against [x0, x1, x2, 1], split by the depot
// flag.  out[c][0..3] node (d0,d1,d2,bias), out[c][4..6] depot (d0,d1,bias)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float *__restrict__ dX0,
                                                        const float *__restrict__ x,
                                                        const uint8_t *__restrict__ depot_mask,
                                                        int rows, int rows_per_block,
                                                        double *__restrict__ partial) {
  __shared__ double sh[2][128][7];
  const int c = threadIdx.x & 127, par = threadIdx.x >> 7;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  double a[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int r = r0 + par; r < r1; r += 2) {
    const double g = (double)dX0[(size_t)r * 128 + c];
    const float *xr = x + (size_t)r * 3;
    if (depot_mask && depot_mask[r]) {
      a[4] += g * xr[0]; a[5] += g * xr[1]; a[6] += g;
    } else {
      a[0] += g * xr[0]; a[1] += g * xr[1]; a[2] += g * xr[2]; a[3] += g;
    }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) sh[par][c][k] = a[k];
  __syncthreads();
  if (par == 0)
#pragma unroll
    for (int k = 0; k < 7; ++k)
      partial[((size_t)blockIdx.x * 128 + c) * 7 + k] = sh[0][c][k] + sh[1][c][k];
}

// ordered sum of the per-block partials, then the (128, 7) table is split into the parameters
__global__ void embed_bwd_scatter_kernel(const double *__restrict__ partial, int nblocks,
                                         int node_dim, int depot_dim, float *dWn, float *dbn,
                                         float *dWd, float *dbd) {
  const int c = threadIdx.x;
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < nblocks; ++b)
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k] += partial[((size_t)b * 128 + c) * 7 + k];
  for (int d = 0; d < node_dim; ++d) dWn[c * node_dim + d] = (float)acc[d];
  dbn[c] = (float)acc[3];
  if (dWd) {
    for (int d = 0; d < depot_dim; ++d) dWd[c * depot_dim + d] = (float)acc[4 + d];
    dbd[c] = (float)acc[6];
  }
}

#define EMB_BWD_BLOCKS 128
struct EncBwdWs {
  float *gA, *gB, *gC, *gQKV, *gH, *WT;
  double *embacc;
  void *slab, *bnws, *csws;
};

extern "C" int64_t vrp_encoder_backward_workspace_bytes(int B, int N, int hidden) {
  const size_t R = (size_t)B * N;
  const int big = hidden > 384 ? hidden : 384;
  return (int64_t)(3 * vrp_align_up(R * 128 * 4) + vrp_align_up(R * 384 * 4) +
                   vrp_align_up(R * (size_t)hidden * 4) + vrp_align_up((size_t)big * 128 * 4) +
                   vrp_align_up((size_t)EMB_BWD_BLOCKS * 128 * 7 * 8) +
                   vrp_align_up((size_t)vrp_gemm_tn_workspace_bytes((int)R, big, big)) +
                   vrp_align_up((size_t)vrp_bn_bwd_workspace_bytes()) +
                   vrp_align_up((size_t)vrp_colsum_workspace_bytes((int)R, big)));
}

static EncBwdWs carve_enc_bwd(void *ws, int B, int N, int hidden) {
  const size_t R = (size_t)B * N;
  const int big = hidden > 384 ? hidden : 384;
  char *p = (char *)ws;
  EncBwdWs w;
  w.gA = (float *)p;   p += vrp_align_up(R * 128 * 4);
  w.gB = (float *)p;   p += vrp_align_up(R * 128 * 4);
  w.gC = (float *)p;   p += vrp_align_up(R * 128 * 4);
  w.gQKV = (float *)p; p += vrp_align_up(R * 384 * 4);
  w.gH = (float *)p;   p += vrp_align_up(R * (size_t)hidden * 4);
  w.WT = (float *)p;   p += vrp_align_up((size_t)big * 128 * 4);
  w.embacc = (double *)p; p += vrp_align_up((size_t)EMB_BWD_BLOCKS * 128 * 7 * 8);
  w.slab = p;          p += vrp_align_up((size_t)vrp_gemm_tn_workspace_bytes((int)R, big, big));
  w.bnws = p;          p += vrp_align_up((size_t)vrp_bn_bwd_workspace_bytes());
  w.csws = p;
  return w;
}

extern "C" int vrp_encoder_backward(const vrp_encoder_weights *w, const vrp_encoder_grads *g, int B,
                                    int N, const float *x, const uint8_t *depot_mask,
                                    const void *tape, const float *d_emb, void *workspace,
                                    void *stream) {
  VRP_REQUIRE(w && g && x && tape && d_emb && workspace, "encoder_backward: NULL argument");
  hipStream_t st = (hipStream_t)stream;
  const int R = B * N, L_ = w->num_layers, Hd = w->hidden;
  EncTape t = carve_tape(const_cast<void *>(tape), B, N, Hd, L_);
  EncBwdWs s = carve_enc_bwd(workspace, B, N, Hd);
  const float *dout = d_emb;  // gradient w.r.t. the layer's output
  for (int l = L_ - 1; l >= 0; --l) {
    const vrp_encoder_layer &P = w->layer[l];
    const vrp_encoder_layer_grads &G = g->layer[l];
    const LayerTape &T = t.layer[l];
    // out = BN2(Z2),  Z2 = Y1 + relu(Y1 W1^T + b1) W2^T + b2
    if (int r = vrp_launch_bn_bwd(dout, T.Z2, T.stats2, P.bn2_weight, R, s.gB, G.bn2_weight,
                                  G.bn2_bias, 0, s.bnws, st)) return r;          // gB = dZ2
    if (int r = vrp_launch_colsum(s.gB, 128, R, 128, G.ff2_bias, 0, s.csws, st)) return r;
    if (int r = vrp_launch_gemm_tn(s.gB, 128, T.H, Hd, G.ff2_weight, R, 128, Hd, 0, s.slab, st))
      return r;                                                                  // dW2 (128,Hd)
    if (int r = vrp_launch_transpose(P.ff2_weight, 128, Hd, Hd, s.WT, st)) return r;   // (Hd,128)
    if (int r = vrp_launch_gemm_nt_full(s.gB, 128, s.WT, 128, nullptr, nullptr, 0, nullptr, T.H,
                                        s.gH, Hd, R, Hd, 128, 0, st)) return r;  // gH = dH (gated)
    if (int r = vrp_launch_colsum(s.gH, Hd, R, Hd, G.ff0_bias, 0, s.csws, st)) return r;
    if (int r = vrp_launch_gemm_tn(s.gH, Hd, T.Y1, 128, G.ff0_weight, R, Hd, 128, 0, s.slab, st))
      return r;                                                                  // dW1 (Hd,128)
    if (int r = vrp_launch_transpose(P.ff0_weight, Hd, 128, 128, s.WT, st)) return r;  // (128,Hd)
    if (int r = vrp_launch_gemm_nt(s.gH, Hd, s.WT, Hd, nullptr, s.gB, 128, s.gA, 128, R, 128, Hd,
                                   0, st)) return r;                             // gA = dY1
    // Y1 = BN1(Z1),  Z1 = X + ATT Wo^T + bo
    if (int r = vrp_launch_bn_bwd(s.gA, T.Z1, T.stats1, P.bn1_weight, R, s.gB, G.bn1_weight,
                                  G.bn1_bias, 0, s.bnws, st)) return r;          // gB = dZ1
    if (int r = vrp_launch_colsum(s.gB, 128, R, 128, G.out_proj_bias, 0, s.csws, st)) return r;
    if (int r = vrp_launch_gemm_tn(s.gB, 128, T.ATT, 128, G.out_proj_weight, R, 128, 128, 0,
                                   s.slab, st)) return r;
    if (int r = vrp_launch_transpose(P.out_proj_weight, 128, 128, 128, s.WT, st)) return r;
    if (int r = vrp_launch_gemm_nt(s.gB, 128, s.WT, 128, nullptr, nullptr, 0, s.gC, 128, R, 128,
                                   128, 0, st)) return r;                        // gC = dATT
    if (int r = vrp_launch_attention_bwd(T.QKV, s.gC, s.gQKV, B, N, st, enc_heads(w))) return r;
    if (int r = vrp_launch_colsum(s.gQKV, 384, R, 384, G.in_proj_bias, 0, s.csws, st)) return r;
    if (int r = vrp_launch_gemm_tn(s.gQKV, 384, T.X, 128, G.in_proj_weight, R, 384, 128, 0, s.slab,
                                   st)) return r;                                // dWin (384,128)
    if (int r = vrp_launch_transpose(P.in_proj_weight, 384, 128, 128, s.WT, st)) return r;  // (128,384)
    if (int r = vrp_launch_gemm_nt(s.gQKV, 384, s.WT, 384, nullptr, s.gB, 128, s.gA, 128, R, 128,
                                   384, 0, st)) return r;                        // gA = dX
    dout = s.gA;
  }
  int eb = (R + 31) / 32;
  if (eb > EMB_BWD_BLOCKS) eb = EMB_BWD_BLOCKS;
  const int erpb = (R + eb - 1) / eb;
  eb = (R + erpb - 1) / erpb;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(eb), dim3(256), 0, st, dout, x, depot_mask, R, erpb,
                     s.embacc);
  VRP_CHECK_LAUNCH("embed_bwd");
  hipLaunchKernelGGL(embed_bwd_scatter_kernel, dim3(1), dim3(128), 0, st, s.embacc, eb, w->node_dim,
                     w->depot_dim, g->node_embed_weight, g->node_embed_bias,
                     w->depot_embed_weight ? g->depot_embed_weight : nullptr,
                     g->depot_embed_bias);
  VRP_CHECK_LAUNCH("embed_bwd_scatter");
  return 0;
}

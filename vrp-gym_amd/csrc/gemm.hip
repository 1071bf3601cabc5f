// fp32 GEMM on the CDNA4 matrix cores: C = epilogue(A * W^T).
//   A (M,K) row-major activations, W (N,K) row-major weights (torch Linear layout).
// v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain in k order), so results stay
// within fp32 rounding of the reference's sgemm.  Used by the encoder projections
// (agents/graph_encoder.py:170-181) and the decoder's per-episode projections
// (agents/graph_decoder.py:83,94).
//
// Tile: BM x 128 per 256-thread workgroup, (BM,BK) = (128,32) or (64,64); the 4 waves sit
// 2x2 and own (BM/2) x 64 each as MFMA 32x32 accumulators.  LDS rows are padded to 33
// floats... (see slot() below).  Two LDS buffers: the next K-tile is fetched into
// registers while the current one feeds the MFMAs and staged into the other buffer,
// one barrier per K-tile.
// Epilogue (fused, in this order): + bias, + residual, BatchNorm affine
// ((v - mean) * mult + beta, eval mode), ReLU, gate (v := 0 where gate <= 0: ReLU backward).
#include <stdlib.h>
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BN 128

// LDS tile: float4 slots indexed (kq, row) -> kq*ROWS + (row ^ kq).  The XOR spreads the
// eight kq values of one row over distinct 16-byte bank groups, so both the staging
// writes (lanes along k) and the fragment reads (lanes along rows) are conflict-free
// 128-bit accesses.
template <int KQ_>
__device__ __forceinline__ int slot(int kq, int row, int rows) {
  return kq * rows + (row ^ (KQ_ == 4 ? (kq << 1) : kq));  // KQ = 4: keep 8 staging lanes apart
}

// RS = rows covered by one pass of the 256 threads (256 / (BK/4) lanes along k)
template <int AS, int WS, int RS>
__device__ __forceinline__ void gemm_fetch(float4 (&av)[AS], float4 (&wv)[WS], const float *A,
                                           int lda, const float *W, int ldw, int m0, int n0,
                                           int lr, int lq, int k0, int M) {
#pragma unroll
  for (int s = 0; s < AS; ++s) {
    const int r = m0 + lr + RS * s;
    av[s] = (r < M) ? *reinterpret_cast<const float4 *>(A + (size_t)r * lda + k0 + 4 * lq)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int s = 0; s < WS; ++s)
    wv[s] = *reinterpret_cast<const float4 *>(W + (size_t)(n0 + lr + RS * s) * ldw + k0 + 4 * lq);
}
template <int AS, int WS, int RS, int BM>
__device__ __forceinline__ void gemm_stage(const float4 (&av)[AS], const float4 (&wv)[WS],
                                           float4 *As, float4 *Ws, int lr, int lq) {
#pragma unroll
  for (int s = 0; s < AS; ++s) As[slot<256 / RS>(lq, lr + RS * s, BM)] = av[s];
#pragma unroll
  for (int s = 0; s < WS; ++s) Ws[slot<256 / RS>(lq, lr + RS * s, BN)] = wv[s];
}

template <int BM, int BK>
__global__ __launch_bounds__(256, (BK == 16 ? 4 : (BK == 64 ? 1 : 2))) void gemm_nt_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ W, int ldw,
    const float *__restrict__ bias, const float *__restrict__ R, int ldr,
    const float *__restrict__ norm, const float *__restrict__ gate, float *__restrict__ C,
    int ldc, int M, int N, int K, int relu) {
  constexpr int MI = BM / 64;   // 32-row MFMA tiles per wave along M
  constexpr int KQ = BK / 4;    // float4 along k per row
  constexpr int RS = 256 / KQ;  // rows staged per pass of the workgroup
  constexpr int AS = BM / RS;   // float4 loads per thread for the A tile
  constexpr int WS = BN / RS;   // ... and for the W tile
  __shared__ float4 As[2][KQ * BM];
  __shared__ float4 Ws[2][KQ * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = tid / KQ;   // row within an RS-row slab
  const int lq = tid % KQ;   // kq of this thread's float4 (lanes run along k)
  float4 av[AS], wv[WS];

  gemm_fetch<AS, WS, RS>(av, wv, A, lda, W, ldw, m0, n0, lr, lq, 0, M);
  gemm_stage<AS, WS, RS, BM>(av, wv, As[0], Ws[0], lr, lq);
  __syncthreads();
  const int fr = lane & 31, fk = lane >> 5;
  const int nkt = K / BK;
#define GEMM_COMPUTE(buf)                                                                     \
  _Pragma("unroll") for (int kq = 0; kq < KQ; ++kq) {                                         \
    float4 a4[MI], b4[2];                                                                     \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                            \
        a4[i] = As[buf][slot<KQ>(kq, wm * (BM / 2) + i * 32 + fr, BM)];                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
        b4[j] = Ws[buf][slot<KQ>(kq, wn * 64 + j * 32 + fr, BN)];                                 \
    /* MFMA 32x32x2 wants A[row][k0+fk]: k-pair 0 = (x,y), k-pair 1 = (z,w) */                \
    _Pragma("unroll") for (int pr = 0; pr < 2; ++pr) {                                        \
      float a[MI], b[2];                                                                      \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                          \
          a[i] = pr ? (fk ? a4[i].w : a4[i].z) : (fk ? a4[i].y : a4[i].x);                    \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
          b[j] = pr ? (fk ? b4[j].w : b4[j].z) : (fk ? b4[j].y : b4[j].x);                    \
      _Pragma("unroll") for (int i = 0; i < MI; ++i)                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0); \
    }                                                                                         \
  }
  for (int kt = 0; kt + 1 < nkt; ++kt) {
    const int buf = kt & 1;
    // next tile: in flight during the MFMA block, staged into the buffer nobody reads
    gemm_fetch<AS, WS, RS>(av, wv, A, lda, W, ldw, m0, n0, lr, lq, (kt + 1) * BK, M);
    GEMM_COMPUTE(buf)
    gemm_stage<AS, WS, RS, BM>(av, wv, As[buf ^ 1], Ws[buf ^ 1], lr, lq);
    __syncthreads();
  }
  {
    const int buf = (nkt - 1) & 1;
    GEMM_COMPUTE(buf)
  }

  // C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = lane & 31, rq = lane >> 5;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + col;
      const float bv = bias ? bias[n] : 0.f;
      float mean = 0.f, mult = 1.f, beta = 0.f;
      if (norm) { mean = norm[n]; mult = norm[128 + n]; beta = norm[256 + n]; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
        if (m < M) {
          float v = acc[i][j][r] + bv;
          if (R) v += R[(size_t)m * ldr + n];
          if (norm) v = (v - mean) * mult + beta;
          if (relu) v = fmaxf(v, 0.f);
          if (gate && !(gate[(size_t)m * ldc + n] > 0.f)) v = 0.f;  // ReLU backward
          C[(size_t)m * ldc + n] = v;
        }
      }
    }
}

// ---- small-problem variant (M*N <= 3M outputs, K = 128): the per-graph query projections of the decoder
// (B rows: graph embedding, first chosen node).  A 64-row tiling would give 3 x 8 workgroups
// at B = 512; this one tiles rows by 16 on v_mfma_f32_16x16x4_f32 (3 x 32 workgroups), keeps
// the whole K in one shot (A tile in LDS, weight fragments straight from L2 to registers)
// and has no K loop or barrier chain.  Wave w owns columns 32w..32w+31 of its 128-column
// block; the 128-long inner dimension is split over the four 16-lane groups (group q walks
// k = 32q + s), the same permutation on both operands.
typedef float f32x4m __attribute__((ext_vector_type(4)));
#define SG_LD 132

// Optional row gather: with gidx, output row m reads A row m * gstride + gidx[m] (the
// embedding of graph m's first chosen node, graph_decoder.py:111-113).
__global__ __launch_bounds__(256) void gemm_nt_m16_k128_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ W, int ldw,
    const float *__restrict__ bias, float *__restrict__ C, int ldc, int M,
    const int32_t *__restrict__ gidx, int gstride) {
  __shared__ __attribute__((aligned(16))) float As[16 * SG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 128 + wave * 32;
  float w[2][32];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const float *wr = W + (size_t)(n0 + 16 * ct + i16) * ldw + 32 * q;
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
      const float4 t = *reinterpret_cast<const float4 *>(wr + s);
      w[ct][s] = t.x; w[ct][s + 1] = t.y; w[ct][s + 2] = t.z; w[ct][s + 3] = t.w;
    }
  }
  for (int idx = tid; idx < 16 * 32; idx += 256) {
    const int r = idx >> 5, c4 = (idx & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + r < M) {
      const size_t src = gidx ? (size_t)(m0 + r) * gstride + gidx[m0 + r] : (size_t)(m0 + r);
      v = *reinterpret_cast<const float4 *>(A + src * lda + c4);
    }
    *reinterpret_cast<float4 *>(As + r * SG_LD + c4) = v;
  }
  __syncthreads();
  f32x4m acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int s = 0; s < 32; s += 4) {
    const float4 a = *reinterpret_cast<const float4 *>(As + i16 * SG_LD + 32 * q + s);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[ct][s], acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[ct][s + 1], acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[ct][s + 2], acc[ct], 0, 0, 0);
      acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[ct][s + 3], acc[ct], 0, 0, 0);
    }
  }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int n = n0 + 16 * ct + i16;  // D: col = lane & 15, row = 4 * (lane >> 4) + reg
    const float bb = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * q + r;
      if (m < M) C[(size_t)m * ldc + n] = acc[ct][r] + bb;
    }
  }
}

int vrp_launch_gemm_rows(const float *A, int lda, const float *W, int ldw, const float *bias,
                         const float *R, int ldr, const float *norm, const float *gate, float *C,
                         int ldc, int M, int N, int K, int relu, hipStream_t st);  // encoder.hip

// norm: optional BatchNorm affine (only for N == 128): [mean | mult | beta]
int vrp_launch_gemm_nt_full(const float *A, int lda, const float *W, int ldw, const float *bias,
                            const float *R, int ldr, const float *norm, const float *gate, float *C,
                            int ldc, int M, int N, int K, int relu, hipStream_t stream) {
  VRP_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  VRP_REQUIRE(N % BN == 0 && K % 64 == 0, "gemm: N=%d must be a multiple of %d and K=%d of 64",
              N, BN, K);
  VRP_REQUIRE((lda % 4) == 0 && (ldw % 4) == 0, "gemm: lda/ldw must be multiples of 4");
  VRP_REQUIRE(!norm || N == 128, "gemm: fused BatchNorm needs N == 128");
  const long tiles128 = (long)(N / BN) * ((M + 127) / 128);
  static const char *force = getenv("VRP_GEMM_VARIANT");  // tuning aid: "64x32", "64x64", "128x32", "rows", "default" (LDS-tiled kernels only)
  if (force && force[0] == '6') {
    dim3 grid(N / BN, (M + 63) / 64);
    if (force[3] == '3')
      hipLaunchKernelGGL((gemm_nt_kernel<64, 32>), grid, dim3(256), 0, stream, A, lda, W, ldw,
                         bias, R, ldr, norm, gate, C, ldc, M, N, K, relu);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<64, 64>), grid, dim3(256), 0, stream, A, lda, W, ldw,
                         bias, R, ldr, norm, gate, C, ldc, M, N, K, relu);
  } else if (force && force[0] == 's') {
    dim3 grid(N / BN, (M + 63) / 64);
    hipLaunchKernelGGL((gemm_nt_kernel<64, 16>), grid, dim3(256), 0, stream, A, lda, W, ldw, bias, R,
                       ldr, norm, gate, C, ldc, M, N, K, relu);
  } else if (force && force[0] == '1') {
    dim3 grid(N / BN, (M + 127) / 128);
    hipLaunchKernelGGL((gemm_nt_kernel<128, 32>), grid, dim3(256), 0, stream, A, lda, W, ldw, bias, R,
                       ldr, norm, gate, C, ldc, M, N, K, relu);
  } else if ((force && force[0] == 'r') || (!force && M >= 256 * 80 && K % 128 == 0)) {
    // at least one 80-row tile per CU: persistent A-stationary kernel (encoder.hip), whose
    // epilogue moves whole rows.  Measured (tools/gemm_rows_probe.py) against the LDS-tiled
    // kernels below, bias + residual + ReLU: 116 vs 209 us at 81920 x 384 x 128, 131 vs 268 at
    // 81920 x 512 x 128, 119 vs 156 at 81920 x 128 x 512, 253 vs 318 at 81920 x 384 x 384; bias
    // only: 104 vs 120, 126 vs 137, 828 vs 911 at 204800 x 1536 x 128, 399 vs 396 at
    // 327680 x 128 x 512.
    return vrp_launch_gemm_rows(A, lda, W, ldw, bias, R, ldr, norm, gate, C, ldc, M, N, K, relu, stream);
  } else if ((long)M * N <= (3L << 20) && K == 128 && !R && !norm && !gate && !relu) {
    // measured (tools/gemm_small_probe.py): 5.7 vs 15.5 us at 512x384, 20 vs 21 at 2048x1536,
    // 35 vs 28 at 4096x1536
    hipLaunchKernelGGL(gemm_nt_m16_k128_kernel, dim3(N / BN, (M + 15) / 16), dim3(256), 0, stream,
                       A, lda, W, ldw, bias, C, ldc, M, nullptr, 0);
    VRP_CHECK_LAUNCH("gemm_nt_m16");
    return 0;
  } else if (tiles128 >= 512) {
    // at least two workgroups per CU: 128x128x16 tiles, 32 KB of LDS
    // and 114 VGPRs -> 4 workgroups per CU overlap each other's barriers and load latency
    // (measured 89-108 TFLOP/s at M = 327680 vs 76-102 for the 128x128x32 tile at 2 per CU)
    dim3 grid(N / BN, (M + 127) / 128);
    hipLaunchKernelGGL((gemm_nt_kernel<128, 16>), grid, dim3(256), 0, stream, A, lda, W, ldw, bias, R,
                       ldr, norm, gate, C, ldc, M, N, K, relu);
  } else {
    // few workgroups: a workgroup's own latency is the kernel's duration.  Measured on
    // MI355X at M = 10240 (tools/gemm_probe.py): K = 128 -> 64x128x32 tiles (3 workgroups
    // per CU overlap each other's load latency), K >= 512 -> 64x128x64 (half the barriers).
    dim3 grid(N / BN, (M + 63) / 64);
    if (K >= 512)
      hipLaunchKernelGGL((gemm_nt_kernel<64, 64>), grid, dim3(256), 0, stream, A, lda, W, ldw,
                         bias, R, ldr, norm, gate, C, ldc, M, N, K, relu);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<64, 32>), grid, dim3(256), 0, stream, A, lda, W, ldw,
                         bias, R, ldr, norm, gate, C, ldc, M, N, K, relu);
  }
  VRP_CHECK_LAUNCH("gemm_nt");
  return 0;
}

// C (M,N) = A[m*gstride + gidx[m]] W^T for small problems (K = 128); returns -1 when the
// shape is outside the small-problem kernel's range (the caller then gathers and calls the
// general GEMM).
int vrp_launch_gemm_gather_k128(const float *A, int lda, const int32_t *gidx, int gstride,
                                const float *W, int ldw, float *C, int ldc, int M, int N,
                                hipStream_t stream) {
  if ((long)M * N > (3L << 20) || N % BN != 0) return -1;
  hipLaunchKernelGGL(gemm_nt_m16_k128_kernel, dim3(N / BN, (M + 15) / 16), dim3(256), 0, stream, A,
                     lda, W, ldw, nullptr, C, ldc, M, gidx, gstride);
  VRP_CHECK_LAUNCH("gemm_gather_m16");
  return 0;
}

int vrp_launch_gemm_nt_ex(const float *A, int lda, const float *W, int ldw, const float *bias,
                          const float *R, int ldr, const float *norm, float *C, int ldc, int M,
                          int N, int K, int relu, hipStream_t stream) {
  return vrp_launch_gemm_nt_full(A, lda, W, ldw, bias, R, ldr, norm, nullptr, C, ldc, M, N, K, relu,
                                 stream);
}

int vrp_launch_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                       const float *R, int ldr, float *C, int ldc, int M, int N, int K,
                       int relu, hipStream_t stream) {
  return vrp_launch_gemm_nt_ex(A, lda, W, ldw, bias, R, ldr, nullptr, C, ldc, M, N, K, relu,
                               stream);
}

extern "C" int vrp_gemm_nt_gated(const float *A, int lda, const float *W, int ldw,
                                 const float *residual, int ldr, const float *gate, float *C,
                                 int ldc, int M, int N, int K, void *stream) {
  return vrp_launch_gemm_nt_full(A, lda, W, ldw, nullptr, residual, ldr, nullptr, gate, C, ldc, M, N,
                                 K, 0, (hipStream_t)stream);
}

extern "C" int vrp_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                           const float *residual, int ldr, float *C, int ldc, int M, int N, int K,
                           int relu, void *stream) {
  return vrp_launch_gemm_nt(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, relu,
                            (hipStream_t)stream);
}

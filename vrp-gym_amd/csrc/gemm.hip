// fp32 GEMM on the CDNA4 matrix cores: C = epilogue(A * W^T).
//   A (M,K) row-major activations, W (N,K) row-major weights (torch Linear layout).
// v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain in k order), so results stay
// within fp32 rounding of the reference's sgemm.  Used by the encoder projections
// (agents/graph_encoder.py:170-181) and the decoder's per-episode projections
// (agents/graph_decoder.py:83,94).
//
// Tile: BM x 128 per 256-thread workgroup (BM = 128 or 64), BK = 32; the 4 waves sit
// 2x2 and own (BM/2) x 64 each as MFMA 32x32 accumulators.  LDS rows are padded to 33
// floats so the per-lane fragment reads (lane -> row, fixed k) are bank-conflict free.
// The next K-tile is fetched into registers while the current one feeds the MFMAs.
// Epilogue (fused, in this order): + bias, + residual, BatchNorm affine
// ((v - mean) * mult + beta, eval mode), ReLU.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BN 128
#define BK 32
#define LDT (BK + 1)

template <int BM>
__global__ __launch_bounds__(256) void gemm_nt_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ W, int ldw,
    const float *__restrict__ bias, const float *__restrict__ R, int ldr,
    const float *__restrict__ norm, float *__restrict__ C, int ldc, int M, int N, int K,
    int relu) {
  constexpr int MI = BM / 64;   // 32-row MFMA tiles per wave along M
  constexpr int AS = BM / 32;   // float4 loads per thread for the A tile
  __shared__ float As[BM * LDT];
  __shared__ float Ws[BN * LDT];
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

  const int lr = tid >> 3;        // 0..31  row within a 32-row slab
  const int lk = (tid & 7) * 4;   // k offset of this thread's float4
  float4 av[AS], wv[4];

  auto fetch = [&](int k0) {
#pragma unroll
    for (int s = 0; s < AS; ++s) {
      const int r = m0 + lr + 32 * s;
      av[s] = (r < M) ? *reinterpret_cast<const float4 *>(A + (size_t)r * lda + k0 + lk)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
      wv[s] = *reinterpret_cast<const float4 *>(W + (size_t)(n0 + lr + 32 * s) * ldw + k0 + lk);
  };
  auto stage = [&]() {
#pragma unroll
    for (int s = 0; s < AS; ++s) {
      float *d = As + (lr + 32 * s) * LDT + lk;
      d[0] = av[s].x; d[1] = av[s].y; d[2] = av[s].z; d[3] = av[s].w;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float *e = Ws + (lr + 32 * s) * LDT + lk;
      e[0] = wv[s].x; e[1] = wv[s].y; e[2] = wv[s].z; e[3] = wv[s].w;
    }
  };

  fetch(0);
  const int fr = lane & 31, fk = lane >> 5;
  for (int k0 = 0; k0 < K; k0 += BK) {
    stage();
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);  // in flight during the MFMA block below
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[MI], b[2];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = As[(wm * (BM / 2) + i * 32 + fr) * LDT + kk + fk];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Ws[(wn * 64 + j * 32 + fr) * LDT + kk + fk];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
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
          C[(size_t)m * ldc + n] = v;
        }
      }
    }
}

// norm: optional BatchNorm affine (only for N == 128): [mean | mult | beta]
int vrp_launch_gemm_nt_ex(const float *A, int lda, const float *W, int ldw, const float *bias,
                          const float *R, int ldr, const float *norm, float *C, int ldc, int M,
                          int N, int K, int relu, hipStream_t stream) {
  VRP_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  VRP_REQUIRE(N % BN == 0 && K % BK == 0, "gemm: N=%d must be a multiple of %d and K=%d of %d",
              N, BN, K, BK);
  VRP_REQUIRE((lda % 4) == 0 && (ldw % 4) == 0, "gemm: lda/ldw must be multiples of 4");
  VRP_REQUIRE(!norm || N == 128, "gemm: fused BatchNorm needs N == 128");
  const long tiles128 = (long)(N / BN) * ((M + 127) / 128);
  if (tiles128 >= 1024) {  // enough workgroups to fill 256 CUs several times over
    dim3 grid(N / BN, (M + 127) / 128);
    hipLaunchKernelGGL(gemm_nt_kernel<128>, grid, dim3(256), 0, stream, A, lda, W, ldw, bias, R,
                       ldr, norm, C, ldc, M, N, K, relu);
  } else {
    dim3 grid(N / BN, (M + 63) / 64);
    hipLaunchKernelGGL(gemm_nt_kernel<64>, grid, dim3(256), 0, stream, A, lda, W, ldw, bias, R,
                       ldr, norm, C, ldc, M, N, K, relu);
  }
  VRP_CHECK_LAUNCH("gemm_nt");
  return 0;
}

int vrp_launch_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                       const float *R, int ldr, float *C, int ldc, int M, int N, int K,
                       int relu, hipStream_t stream) {
  return vrp_launch_gemm_nt_ex(A, lda, W, ldw, bias, R, ldr, nullptr, C, ldc, M, N, K, relu,
                               stream);
}

extern "C" int vrp_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                           const float *residual, int ldr, float *C, int ldc, int M, int N, int K,
                           int relu, void *stream) {
  return vrp_launch_gemm_nt(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, relu,
                            (hipStream_t)stream);
}

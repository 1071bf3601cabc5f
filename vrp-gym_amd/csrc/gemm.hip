// fp32 GEMM on the CDNA4 matrix cores: C = A * W^T (+ bias) (+ residual) (ReLU).
//   A (M,K) row-major activations, W (N,K) row-major weights (torch Linear layout).
// v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain in k order), so results stay
// within fp32 rounding of the reference's sgemm.  Used by the encoder projections
// (agents/graph_encoder.py:170-181) and the decoder's per-episode projections
// (agents/graph_decoder.py:83,94).
//
// Tile: 128x128 per 256-thread workgroup, BK = 32; wave (wm,wn) owns a 64x64
// sub-tile as 2x2 MFMA 32x32 accumulators.  LDS rows are padded to 33 floats so
// that the per-lane fragment reads (lane -> row, fixed k) are bank-conflict free.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define BK 32
#define LDT (BK + 1)

__global__ __launch_bounds__(256) void gemm_nt_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ W, int ldw,
    const float *__restrict__ bias, const float *__restrict__ R, int ldr,
    float *__restrict__ C, int ldc, int M, int N, int K, int relu) {
  __shared__ float As[BM * LDT];
  __shared__ float Ws[BN * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: consecutive row-tiles (which share no data) are spread
  // over XCDs by the hardware's round-robin; column tiles of one row-tile reuse A
  // through L2, so they are made adjacent in dispatch order by the grid layout.
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = tid >> 3;        // 0..31  row within a 32-row slab
  const int lk = (tid & 7) * 4;   // k offset of this thread's float4

  for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r = lr + 32 * s;
      float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + r < M) av = *reinterpret_cast<const float4 *>(A + (size_t)(m0 + r) * lda + k0 + lk);
      float *d = As + r * LDT + lk;
      d[0] = av.x; d[1] = av.y; d[2] = av.z; d[3] = av.w;
      float4 wv = *reinterpret_cast<const float4 *>(W + (size_t)(n0 + r) * ldw + k0 + lk);
      float *e = Ws + r * LDT + lk;
      e[0] = wv.x; e[1] = wv.y; e[2] = wv.z; e[3] = wv.w;
    }
    __syncthreads();
    const int fr = lane & 31, fk = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a0 = As[(wm * 64 + fr) * LDT + kk + fk];
      float a1 = As[(wm * 64 + 32 + fr) * LDT + kk + fk];
      float b0 = Ws[(wn * 64 + fr) * LDT + kk + fk];
      float b1 = Ws[(wn * 64 + 32 + fr) * LDT + kk + fk];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int col = lane & 31, rq = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + col;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
        if (m < M) {
          float v = acc[i][j][r] + bv;
          if (R) v += R[(size_t)m * ldr + n];
          if (relu) v = fmaxf(v, 0.f);
          C[(size_t)m * ldc + n] = v;
        }
      }
    }
}

int vrp_launch_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                       const float *R, int ldr, float *C, int ldc, int M, int N, int K,
                       int relu, hipStream_t stream) {
  VRP_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  VRP_REQUIRE(N % BN == 0 && K % BK == 0, "gemm: N=%d must be a multiple of %d and K=%d of %d",
              N, BN, K, BK);
  VRP_REQUIRE((lda % 4) == 0 && (ldw % 4) == 0, "gemm: lda/ldw must be multiples of 4");
  dim3 grid(N / BN, (M + BM - 1) / BM);
  hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, stream, A, lda, W, ldw, bias, R, ldr, C,
                     ldc, M, N, K, relu);
  VRP_CHECK_LAUNCH("gemm_nt");
  return 0;
}

extern "C" int vrp_gemm_nt(const float *A, int lda, const float *W, int ldw, const float *bias,
                           const float *residual, int ldr, float *C, int ldc, int M, int N, int K,
                           int relu, void *stream) {
  return vrp_launch_gemm_nt(A, lda, W, ldw, bias, residual, ldr, C, ldc, M, N, K, relu,
                            (hipStream_t)stream);
}

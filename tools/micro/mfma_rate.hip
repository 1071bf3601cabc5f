// Microbenchmark: issue rate of the fp32 MFMAs on gfx950 (cycles per instruction per SIMD),
// for 1..4 waves per SIMD and 1..4 independent accumulators per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int KIND>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  unsigned long long t0 = 0, t1 = 0;
  if (KIND == 0) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else if (KIND == 2) {   // bf16 16x16x32 (round 5)
    bf16x8 pa, pb;
    for (int e = 0; e < 8; ++e) { pa[e] = (__bf16)(a + e); pb[e] = (__bf16)(b - e); }
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else if (KIND == 3) {   // bf16 32x32x16 (round 5)
    bf16x8 pa, pb;
    for (int e = 0; e < 8; ++e) { pa[e] = (__bf16)(a + e); pb[e] = (__bf16)(b - e); }
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int KIND>
void run(int waves_per_simd, int blocks) {
  float *out; unsigned long long *cyc;
  const int threads = 256 * waves_per_simd, iters = 2000;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, 8 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024]; hipMemcpy(h, cyc, 8 * blocks, hipMemcpyDeviceToHost);
  const double per_wave = (double)iters * 8 * NACC;                   // MFMAs issued by one wave
  const double flop = (KIND == 0 ? 2048.0 : KIND == 1 ? 4096.0 : KIND == 2 ? 16384.0 : 32768.0) * per_wave * waves_per_simd * 4 * blocks;
  // wave 0 is the oldest wave of its SIMD: the arbiter serves it first, so its own rate stays
  // one MFMA per 32 (64) cycles however many waves share the pipe; the chip-wide rate is the
  // event-timed figure
  printf("%s nacc %d waves/SIMD %d blocks %d: %.1f cycles per MFMA for the oldest wave (s_memtime), "
         "%.1f TFLOP/s chip-wide (events)\n",
         KIND == 0 ? "16x16x4" : KIND == 1 ? "32x32x2" : KIND == 2 ? "bf16 16x16x32" : "bf16 32x32x16", NACC, waves_per_simd, blocks, (double)h[0] / per_wave,
         flop / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1, 256}) {
    run<1, 0>(1, blocks); run<2, 0>(1, blocks); run<4, 0>(1, blocks);
    run<1, 0>(2, blocks); run<3, 0>(2, blocks); run<4, 0>(2, blocks); run<4, 0>(4, blocks);
    run<1, 1>(1, blocks); run<2, 1>(1, blocks); run<2, 1>(2, blocks); run<4, 1>(2, blocks);
    run<1, 2>(1, blocks); run<4, 2>(1, blocks); run<8, 2>(1, blocks); run<4, 2>(2, blocks); run<8, 2>(2, blocks); run<4, 2>(4, blocks);
    run<1, 3>(1, blocks); run<2, 3>(1, blocks); run<4, 3>(1, blocks); run<2, 3>(2, blocks); run<4, 3>(2, blocks);
  }
  return 0;
}

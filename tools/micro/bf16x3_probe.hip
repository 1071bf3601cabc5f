// Probe: fp32 matrix products on the bf16 matrix cores by operand splitting (x = h + m + l, three
// bf16 planes, 24 mantissa bits; six of the nine cross products kept: hh, hm, mh, hl, lh, mm)
// against the fp32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate on gfx950).
//   (a) accuracy: 16x16 tiles of C = A B^T, K = 128 / 512, random normal data, against fp64;
//   (b) rate: tiles per second of a wave that keeps A pre-split (static weights) and splits B on
//       the fly (activations), against the fp32 MFMA loop.
// build: hipcc --offload-arch=gfx950 -O3 -o bf16x3_probe bf16x3_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

struct Split { bf8 h, m, l; };
__device__ __forceinline__ Split split8(const float (&x)[8]) {
  Split s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    const float r1 = x[i] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    s.h[i] = h; s.m[i] = m; s.l[i] = (__bf16)r2;
  }
  return s;
}
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// small terms first, the dominant product last
__device__ __forceinline__ f4 mma6(const Split &a, const Split &b, f4 c) {
  c = MFMA_BF16(a.m, b.m, c);
  c = MFMA_BF16(a.h, b.l, c);
  c = MFMA_BF16(a.l, b.h, c);
  c = MFMA_BF16(a.h, b.m, c);
  c = MFMA_BF16(a.m, b.h, c);
  c = MFMA_BF16(a.h, b.h, c);
  return c;
}

// one wave per tile: C[t] (16x16) = A[t] (16xK) B[t]^T (16xK)
template <int MODE>   // 0: fp32 MFMA   1: bf16 x 6   2: bf16 x 3 (hh, hm, mh: two planes)
__global__ void tile_kernel(const float *A, const float *B, float *C, int K) {
  const int t = blockIdx.x, lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  const float *a = A + (size_t)t * 16 * K + (size_t)r * K, *b = B + (size_t)t * 16 * K + (size_t)r * K;
  f4 acc = {0, 0, 0, 0};
  if (MODE == 0) {
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k + g], b[k + g], acc, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 32) {
      float xa[8], xb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { xa[i] = a[k + 8 * g + i]; xb[i] = b[k + 8 * g + i]; }
      const Split sa = split8(xa), sb = split8(xb);
      if (MODE == 1) {
        acc = mma6(sa, sb, acc);
      } else {
        acc = MFMA_BF16(sa.h, sb.m, acc);
        acc = MFMA_BF16(sa.m, sb.h, acc);
        acc = MFMA_BF16(sa.h, sb.h, acc);
      }
    }
  }
  // D: lane (col j = r, row group g) holds rows 4g..4g+3 -- rows index A, columns index B
#pragma unroll
  for (int i = 0; i < 4; ++i) C[(size_t)t * 256 + (4 * g + i) * 16 + r] = acc[i];
}

// rate: every wave computes `tiles` output tiles of K = 128; A planes static in registers (one
// 16x128 block = 4 k-steps), B rows re-split for every tile from fp32 registers
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(const float *A, const float *B, float *C, int tiles) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float *a = A + (size_t)r * 128, *b = B + (size_t)(w & 15) * 16 * 128 + (size_t)r * 128;
  f4 total = {0, 0, 0, 0};
  if (MODE == 0) {
    float ar[32], br[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) { ar[k] = a[4 * k + g]; br[k] = b[4 * k + g]; }
    for (int t = 0; t < tiles; ++t) {
      f4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 32; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[k], br[k] + (float)t, acc, 0, 0, 0);
      total += acc;
    }
  } else {
    Split sa[4];
    float xb[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float xa[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { xa[i] = a[32 * k + 8 * g + i]; xb[k][i] = b[32 * k + 8 * g + i]; }
      sa[k] = split8(xa);
    }
    for (int t = 0; t < tiles; ++t) {
      f4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = xb[k][i] + (float)t;   // fresh activations every tile
        const Split sb = split8(x);
        acc = mma6(sa[k], sb, acc);
      }
      total += acc;
    }
  }
  if (total[0] == 123.456f) C[w] = total[0] + total[1] + total[2] + total[3];
}

static double gauss() {
  const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
  return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

int main() {
  srand(7);
  for (int K : {128, 512}) {
    const int T = 64;
    std::vector<float> A((size_t)T * 16 * K), B((size_t)T * 16 * K), C((size_t)T * 256);
    for (auto &x : A) x = (float)gauss();
    for (auto &x : B) x = (float)(gauss() * 0.1);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 0) hipLaunchKernelGGL(tile_kernel<0>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K);
      if (mode == 1) hipLaunchKernelGGL(tile_kernel<1>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K);
      if (mode == 2) hipLaunchKernelGGL(tile_kernel<2>, dim3(T), dim3(64), 0, 0, dA, dB, dC, K);
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
      double maxe = 0, sse = 0, maxe_f = 0, sse_f = 0, scale = 0;
      for (int t = 0; t < T; ++t)
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j < 16; ++j) {
            double ref = 0;
            float chain = 0.f;
            for (int k = 0; k < K; ++k) {
              ref += (double)A[((size_t)t * 16 + i) * K + k] * B[((size_t)t * 16 + j) * K + k];
              chain = fmaf(A[((size_t)t * 16 + i) * K + k], B[((size_t)t * 16 + j) * K + k], chain);
            }
            const double e = fabs(C[(size_t)t * 256 + i * 16 + j] - ref), ef = fabs(chain - ref);
            maxe = fmax(maxe, e); sse += e * e; maxe_f = fmax(maxe_f, ef); sse_f += ef * ef;
            scale += ref * ref;
          }
      const double n = T * 256.0;
      printf("K %4d %-22s max|err| %.3e rms %.3e   (host fmaf chain: max %.3e rms %.3e; rms |C| %.3f)\n", K,
             mode == 0 ? "fp32 MFMA 16x16x4" : mode == 1 ? "bf16 x 6 (3 planes)" : "bf16 x 3 (2 planes)",
             maxe, sqrt(sse / n), maxe_f, sqrt(sse_f / n), sqrt(scale / n));
    }
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  // ---- rate
  {
    std::vector<float> A(16 * 128, 0.5f), B(16 * 16 * 128, 0.25f);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1 << 20);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = 4000;
    for (int blocks : {256, 512, 1024}) {
      for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, dA, dB, dC, tiles);
          else hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, dA, dB, dC, tiles);
          hipEventRecord(e1);
          hipDeviceSynchronize();
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 16 * 16 * 128 * tiles * 4.0 * blocks;
        printf("rate: %-10s %4d workgroups x 4 waves: %.2f ms, %.1f fp32-equivalent TFLOP/s\n",
               mode == 0 ? "fp32 MFMA" : "bf16 x 6", blocks, ms, flop / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}

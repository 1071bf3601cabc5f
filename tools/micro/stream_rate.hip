// Microbenchmark: how fast can ONE workgroup per CU pull a private, once-read tile into registers?
// (the load phase of the raw-tile step kernel: 8 graphs x 51 KB per CU at VRP-100 x 2048.)
// Every wave streams `kb_per_wave` KB of its own contiguous region with 16-byte loads, `U`
// instructions in flight, and folds them into a checksum.  Variants: plain / nontemporal loads,
// LDS-DMA (global_load_lds_dwordx4 into a per-wave ring, read back with ds_read_b128), waves per
// workgroup, workgroups per CU, and the footprint (L2 / Infinity Cache / HBM).
// build: hipcc --offload-arch=gfx950 -O3 -o stream_rate stream_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int U, int MODE>   // MODE 0 plain, 1 nontemporal
__global__ __launch_bounds__(1024) void stream_k(const float4 *__restrict__ src, float *out,
                                                 int kb_per_wave) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const float4 *p = src + wave * (size_t)kb_per_wave * 64 + lane;   // 1 KB = 64 float4
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < kb_per_wave; i += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      if (MODE == 1) {
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p + (size_t)(i + u) * 64));
        v[u] = make_float4(t.x, t.y, t.z, t.w);
      } else {
        v[u] = p[(size_t)(i + u) * 64];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

// LDS-DMA: each wave owns a ring of R 1-KB slots; slot s of the ring is refilled as soon as it has
// been read back into registers.
template <int R>
__global__ __launch_bounds__(1024) void stream_dma_k(const float4 *__restrict__ src, float *out,
                                                     int kb_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float4 ring[];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + w;
  const float4 *p = src + wave * (size_t)kb_per_wave * 64 + lane;
  float4 *mine = ring + (size_t)w * R * 64;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < R; ++s)
    __builtin_amdgcn_global_load_lds(p + (size_t)s * 64, (__attribute__((address_space(3))) void *)(mine + s * 64), 16, 0, 0);
  for (int i = 0; i < kb_per_wave; i += R) {
#pragma unroll
    for (int s = 0; s < R; ++s) {
      // wait until at most R-1 DMAs are outstanding: the oldest one has landed
      if (R == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (R == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      const float4 v = mine[s * 64 + lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      const int nxt = i + R + s;
      // (past the end: reload the last KB, keeps the count of outstanding DMAs constant)
      __builtin_amdgcn_global_load_lds(p + (size_t)(nxt < kb_per_wave ? nxt : kb_per_wave - 1) * 64,
                                       (__attribute__((address_space(3))) void *)(mine + s * 64), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}

static float4 *buf;
static float *out;

template <typename F>
static float time_it(F launch, int reps = 8) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch();
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

template <int U, int MODE>
static void run(int wgs, int waves, int kb_per_wave) {
  const double bytes = (double)wgs * waves * kb_per_wave * 1024.0;
  const float us = time_it([&] { hipLaunchKernelGGL((stream_k<U, MODE>), dim3(wgs), dim3(64 * waves), 0, 0, buf, out, kb_per_wave); });
  printf("%s U=%2d wgs=%4d waves/wg=%2d KB/wave=%3d  %6.1f MB  %7.2f us  %5.2f TB/s  %5.1f GB/s per CU\n",
         MODE ? "nt   " : "plain", U, wgs, waves, kb_per_wave, bytes / 1e6, us, bytes / us / 1e6,
         bytes / us / 1e3 / 256.0);
}
template <int R>
static void run_dma(int wgs, int waves, int kb_per_wave) {
  const double bytes = (double)wgs * waves * kb_per_wave * 1024.0;
  const size_t lds = (size_t)waves * R * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_dma_k<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const float us = time_it([&] { hipLaunchKernelGGL((stream_dma_k<R>), dim3(wgs), dim3(64 * waves), lds, 0, buf, out, kb_per_wave); });
  printf("ldsdma R=%2d wgs=%4d waves/wg=%2d KB/wave=%3d  %6.1f MB  %7.2f us  %5.2f TB/s  %5.1f GB/s per CU\n",
         R, wgs, waves, kb_per_wave, bytes / 1e6, us, bytes / us / 1e6, bytes / us / 1e3 / 256.0);
}

int main() {
  const size_t cap = (size_t)1 << 30;
  hipMalloc(&buf, cap);
  hipMalloc(&out, 64);
  hipMemset(buf, 0, cap);
  // the tile kernel's geometry: 256 workgroups (one per CU) x 8 waves x 50 KB
  puts("== 256 x 8 waves x 50 KB (the raw-tile kernel at VRP-100 x 2048: 105 MB)");
  run<10, 0>(256, 8, 50); run<25, 0>(256, 8, 50); run<50, 0>(256, 8, 50);
  run<25, 1>(256, 8, 50); run<50, 1>(256, 8, 50);
  run_dma<4>(256, 8, 48); run_dma<8>(256, 8, 48); run_dma<16>(256, 8, 48);
  puts("== the same bytes over 16 waves per CU");
  run<25, 0>(256, 16, 25); run<25, 1>(256, 16, 25); run<25, 0>(512, 8, 25); run_dma<8>(256, 16, 24);
  puts("== 512 x 8 waves x 20 KB (TSP-40 x 8192, first pass over 16 graphs: 84 MB per 256 workgroups)");
  run<20, 0>(512, 8, 40); run<20, 1>(512, 8, 40); run<20, 0>(1024, 8, 20);
  puts("== footprint: 26 MB (L2 + Infinity Cache), 420 MB (HBM)");
  run<25, 0>(256, 8, 12); run<25, 0>(1024, 8, 50); run<25, 1>(1024, 8, 50); run<25, 0>(2048, 4, 50);
  puts("== high occupancy reference: 4096 workgroups x 4 waves x 16 KB");
  run<8, 0>(4096, 4, 16); run<8, 1>(4096, 4, 16);
  return 0;
}

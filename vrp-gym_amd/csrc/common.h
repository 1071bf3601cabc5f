// Shared device/host helpers for libvrpgym_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vrpgym_hip.h"

#define VRP_WAVE 64
#define VRP_D 384          // 3 * emb, decoder attention width
#define VRP_HD 48          // decoder head dim

void vrp_set_error(const char *fmt, ...);

#define VRP_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      vrp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

#define VRP_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      vrp_set_error(__VA_ARGS__);     \
      return 2;                       \
    }                                 \
  } while (0)

static inline size_t vrp_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- wave-level reductions (all 64 lanes participate) -----------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// argmax with lowest index among equal values (torch CPU argmax semantics).
__device__ __forceinline__ void wave_argmax(float &v, int &i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(v, o, 64);
    int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

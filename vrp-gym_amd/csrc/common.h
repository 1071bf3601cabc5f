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

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: a launcher
// remembers per device (not per process) that it has raised its kernel's limit.
#include <atomic>
struct VrpAttrOnce {
  std::atomic<unsigned long long> devices{0};
  static unsigned long long bit() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return 1ull << (dev & 63);
  }
  bool done() const { return (devices.load(std::memory_order_acquire) & bit()) != 0; }
  void mark() { devices.fetch_or(bit(), std::memory_order_release); }
};

static inline size_t vrp_align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// exp(x) for FINITE x <= 0 (softmax numerators: scores minus their finite maximum): exp2 of
// x*log2(e) with the product's rounding error carried into a first-order correction; ~1 ulp.
// The argument is clamped at -200 (exp = 0 in fp32 well before): -inf or an overflowing
// product would otherwise turn the correction term into 0 * NaN.
__device__ __forceinline__ float exp_nonpos(float x) {
  const float l2e_hi = 1.44269502162933349609375f, l2e_lo = 1.9259629911e-8f;
  x = fmaxf(x, -200.f);
  const float t = x * l2e_hi;
  float r = fmaf(x, l2e_hi, -t);
  r = fmaf(x, l2e_lo, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.693147180559945f, e);
}

// ---- counter-based random numbers (Philox4x32-10): instance generator, device RandomAgent and
// the in-kernel sampling noise of the throughput mode
__device__ __forceinline__ void vrp_philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// Exp(1) noise q of Categorical.sample's argmax(p / q) (agents/graph_decoder.py:104-107) for
// step t, graph b, node n, drawn in the kernel: counter = (b, n, t, tag), key = seed.  NOT the
// reference's CPU generator stream (parity runs ship host-drawn noise through io.noise).
// 32 random bits -> q in (0, inf): u = (23 bits + 1/2) * 2^-23 lies in [2^-24, 1 - 2^-24] and
// every such value is exact in fp32 (a 24-bit significand), so u never rounds to 1 and -log(u)
// never to -0 (with 24 bits + 1/2 the largest draw rounded to 1.0: q = -0, ratio p/q = -inf on
// the ONE selectable node of a forced move, and a masked node won the argmax); the clamp keeps q
// positive whatever the fast logarithm returns next to 1.
__device__ __forceinline__ float vrp_exp1_from_bits(uint32_t bits) {
  const float u = ((float)(bits >> 9) + 0.5f) * (1.0f / 8388608.0f);
  return fmaxf(-__logf(u), 1e-30f);
}
__device__ __forceinline__ float vrp_exp1_noise(uint64_t seed, int t, int b, int n) {
  uint32_t c[4] = {(uint32_t)b, (uint32_t)n, (uint32_t)t, 0x45585031u};
  vrp_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return vrp_exp1_from_bits(c[0]);
}

// ---- wave-level reductions (all 64 lanes participate) -----------------------
// DPP row shifts + row broadcasts (no LDS crossbar): after the six steps lane 63 holds
// the reduction of the whole wave; readlane(63) hands it back as a wave-uniform value.
#define VRP_DPP(old, src, ctrl, rowmask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), \
      __builtin_bit_cast(int, (float)(src)), (ctrl), (rowmask), 0xF, false))
__device__ __forceinline__ float wave_sum(float v) {
  v += VRP_DPP(0.f, v, 0x111, 0xF);  // row_shr:1
  v += VRP_DPP(0.f, v, 0x112, 0xF);  // row_shr:2
  v += VRP_DPP(0.f, v, 0x114, 0xF);  // row_shr:4
  v += VRP_DPP(0.f, v, 0x118, 0xF);  // row_shr:8   -> lane 15 of each row = row total
  v += VRP_DPP(0.f, v, 0x142, 0xA);  // row_bcast:15 into rows 1,3
  v += VRP_DPP(0.f, v, 0x143, 0xC);  // row_bcast:31 into rows 2,3 -> lane 63 = total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, VRP_DPP(v, v, 0x111, 0xF));
  v = fmaxf(v, VRP_DPP(v, v, 0x112, 0xF));
  v = fmaxf(v, VRP_DPP(v, v, 0x114, 0xF));
  v = fmaxf(v, VRP_DPP(v, v, 0x118, 0xF));
  v = fmaxf(v, VRP_DPP(v, v, 0x142, 0xA));
  v = fmaxf(v, VRP_DPP(v, v, 0x143, 0xC));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Eight wave-wide sums at once: the DPP steps of the eight values are interleaved, so that
// every DPP read is separated from the VALU write of its source by the other seven chains
// (no s_nop wait states); each value goes through exactly the tree of wave_sum.
__device__ __forceinline__ void wave_sum8(float (&v)[8]) {
#pragma unroll
  for (int h = 0; h < 8; ++h) v[h] += VRP_DPP(0.f, v[h], 0x111, 0xF);
#pragma unroll
  for (int h = 0; h < 8; ++h) v[h] += VRP_DPP(0.f, v[h], 0x112, 0xF);
#pragma unroll
  for (int h = 0; h < 8; ++h) v[h] += VRP_DPP(0.f, v[h], 0x114, 0xF);
#pragma unroll
  for (int h = 0; h < 8; ++h) v[h] += VRP_DPP(0.f, v[h], 0x118, 0xF);
#pragma unroll
  for (int h = 0; h < 8; ++h) v[h] += VRP_DPP(0.f, v[h], 0x142, 0xA);
#pragma unroll
  for (int h = 0; h < 8; ++h) v[h] += VRP_DPP(0.f, v[h], 0x143, 0xC);
#pragma unroll
  for (int h = 0; h < 8; ++h)
    v[h] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[h]), 63));
}
// sum over the 8 lanes that share lane>>3 (quad swaps + half-row mirror); every lane of
// the group ends with the group total
__device__ __forceinline__ float group8_sum(float v) {
  v += VRP_DPP(0.f, v, 0xB1, 0xF);   // quad_perm [1,0,3,2]
  v += VRP_DPP(0.f, v, 0x4E, 0xF);   // quad_perm [2,3,0,1]
  v += VRP_DPP(0.f, v, 0x141, 0xF);  // row_half_mirror: lane i <-> 7-i within 8 lanes
  return v;
}
// index of the maximum with the lowest index among equals (torch CPU argmax), one value
// per lane, lane id = candidate index
__device__ __forceinline__ int wave_argmax_lane(float v) {
  const float m = wave_max(v);
  const unsigned long long hit = __ballot(v == m);
  return __ffsll((long long)hit) - 1;
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

// fp32 operands as three bf16 planes (h, m, l: x = h + m + l up to 2^-24 |x|) and the six-product
// MFMA sequence that multiplies two of them at fp32 accuracy on the bf16 matrix cores: shared by
// the eval-mode encoder kernels (encoder_x3.h) and the decoder prologue's projections
// (decoder_prologue.hip).  Rationale and measurements: encoder_x3.h.
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define X3_FRAG 6144   // bf16 per weight fragment: 3 planes x 4 k-chunks x 64 lanes x 8

struct Bf3 { __bf16 h, m, l; };
__device__ __forceinline__ Bf3 x3_split(float x) {
  Bf3 s;
  s.h = (__bf16)x;
  const float r1 = x - (float)s.h;
  s.m = (__bf16)r1;
  s.l = (__bf16)(r1 - (float)s.m);
  return s;
}
// eight values at once, two per conversion instruction (v_cvt_pk_bf16_f32): the planes as the
// 16-byte MFMA operands
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x3_split8(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
  bf16x2 hp[4], mp[4], lp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const x3_f32x2 v = {x[2 * i], x[2 * i + 1]};
    hp[i] = __builtin_convertvector(v, bf16x2);
    const x3_f32x2 r1 = v - __builtin_convertvector(hp[i], x3_f32x2);
    mp[i] = __builtin_convertvector(r1, bf16x2);
    const x3_f32x2 r2 = r1 - __builtin_convertvector(mp[i], x3_f32x2);
    lp[i] = __builtin_convertvector(r2, bf16x2);
  }
  auto cat = [](const bf16x2 (&p)[4]) {
    const auto a = __builtin_shufflevector(p[0], p[1], 0, 1, 2, 3);
    const auto b = __builtin_shufflevector(p[2], p[3], 0, 1, 2, 3);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  h = cat(hp); m = cat(mp); l = cat(lp);
}
#define X3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

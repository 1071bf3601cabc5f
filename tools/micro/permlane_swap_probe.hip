// One level of the raw-tile kernel's reduce-scatter, select/shuffle version against v_permlane32_swap
// (gfx950).  Finding (ROCm 7.2 hipcc -O3): through __builtin_amdgcn_permlane32_swap the second result
// came back equal to the first in this context (correct in a smaller probe); the instruction itself,
// issued from inline asm, does what the ISA says: r0 = [a.lo | b.lo], r1 = [a.hi | b.hi].
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/permlane_swap_probe.hip -o /tmp/permlane_swap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float *o) {
  const int lane = threadIdx.x;
  float v[64];
  for (int i = 0; i < 64; ++i) v[i] = (float)(i * 100 + lane);
  const bool up = (lane & 32) != 0;
  // select version, element i = 3
  {
    const int i = 3, H = 32;
    const float keep = up ? v[i + H] : v[i];
    const float send = up ? v[i] : v[i + H];
    o[lane] = keep + __shfl_xor(send, H, 64);
    o[64 + lane] = keep;
    o[128 + lane] = __shfl_xor(send, H, 64);
  }
  {
    const int i = 3, H = 32;
    const unsigned x = __builtin_bit_cast(unsigned, v[i]), y = __builtin_bit_cast(unsigned, v[i + H]);
    unsigned xa = x, ya = y;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(xa), "+v"(ya));
    o[192 + lane] = __builtin_bit_cast(float, xa);
    o[256 + lane] = __builtin_bit_cast(float, ya);
  }
}
int main() {
  float *d, h[320];
  (void)hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *n[5] = {"select sum", "keep", "recv", "swap r0", "swap r1"};
  for (int j = 0; j < 5; ++j) {
    printf("%-10s:", n[j]);
    for (int l = 0; l < 64; l += 8) printf(" [%d]=%.0f", l, h[64 * j + l]);
    printf("\n");
  }
}

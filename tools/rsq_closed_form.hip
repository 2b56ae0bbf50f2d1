// rsq_closed_form.hip -- experiment: the captured x86 RSQRTPS table equals round((2 / sqrt(x_mid) - 1) * 4096) for every one of its 2048 entries (x_mid = midpoint of
// the 10-bit mantissa interval; checked in double precision by tools/isa... see profiles/archive/r02_rsqrt_closed_form.md).  Can the GPU recompute the entry instead of
// gathering it?  This program evaluates candidate instruction sequences for all 2048 indices on the device and counts mismatches against the table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "../limg_amd/csrc/limg_rsqrt_x86_table.h"

__global__ void k(uint32_t *out)
{
  const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x; // table index: bit 10 = exponent even, low 10 bits = mantissa top bits
  const uint32_t even = i >> 10, m10 = i & 1023u;
  // canonical input: odd exponent -> [1,2), even -> [0.5,1) (= [2,4) / 4); midpoint of the interval
  const uint32_t xb = ((even ? 126u : 127u) << 23) | (m10 << 13) | 0x1000u;
  const float x = __uint_as_float(xb);
  const float y = __builtin_amdgcn_rsqf(x);                  // odd: (0.707,1]; even: (1,1.414]
  const float s = even ? 4096.0f : 8192.0f;
  // variant 0: plain v_rsq_f32
  out[i] = (uint32_t)(int)__builtin_rintf(y * s) & 0xFFFu;
  // variant 1: one Newton step y1 = y * (1.5 - 0.5 x y^2) with fma
  const float h = 0.5f * x;
  const float e = __builtin_fmaf(-h * y, y, 0.5f);           // 0.5 - 0.5 x y^2
  const float y1 = __builtin_fmaf(y, e, y);
  out[2048 + i] = (uint32_t)(int)__builtin_rintf(y1 * s) & 0xFFFu;
  // variant 2: double precision
  const double yd = 1.0 / sqrt((double)x);
  out[4096 + i] = (uint32_t)(int)rint(yd * (double)s) & 0xFFFu;
}

int main()
{
  uint32_t *d, h[3 * 2048];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(8), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[3] = { "v_rsq_f32", "v_rsq_f32 + 1 Newton step (fma)", "double 1/sqrt" };
  for (int v = 0; v < 3; v++)
  {
    int bad = 0;
    for (int i = 0; i < 2048; i++)
      if (h[v * 2048 + i] != limg_rsqrt_x86_tab[i]) { if (bad < 8) printf("  %s: index %d got %03x want %03x\n", names[v], i, h[v * 2048 + i], limg_rsqrt_x86_tab[i]); bad++; }
    printf("{\"variant\": \"%s\", \"mismatches_of_2048\": %d}\n", names[v], bad);
  }
  return 0;
}

// cvt_pk_u8_check.hip -- does v_cvt_pk_u8_f32 equal clamp(round-to-nearest-even(x), 0, 255) (what a8 needs: src/limg_factorization.h:98-197 converts 255 * f with
// CVTPS2DQ, then packs with saturation)?  Checks every float in [-4, 260] whose fraction is within a few ulps of .0 / .5, a dense sweep, and the specials.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <cstring>

__global__ void k(const float *in, uint32_t *out, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t r;
  asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(in[i]));
  out[i] = r;
}

int main()
{
  std::vector<float> v;
  for (int kk = -8; kk <= 520; kk++)
  {
    const float c = kk * 0.5f;
    float lo = c, hi = c;
    for (int s = 0; s < 6; s++) { v.push_back(lo); v.push_back(hi); lo = nextafterf(lo, -1e9f); hi = nextafterf(hi, 1e9f); }
  }
  for (int i = 0; i < 2000000; i++) v.push_back(-4.0f + 264.0f * (float)i / 2000000.0f);
  const uint32_t sp[] = { 0x7FC00000u, 0xFFC00000u, 0x7F800000u, 0xFF800000u, 0x7F7FFFFFu, 0xFF7FFFFFu, 0x00000001u, 0x80000001u, 0x80000000u, 0x4F000000u, 0xCF000000u };
  for (uint32_t u : sp) { float f; memcpy(&f, &u, 4); v.push_back(f); }
  float *din; uint32_t *dout;
  hipMalloc(&din, v.size() * 4); hipMalloc(&dout, v.size() * 4);
  hipMemcpy(din, v.data(), v.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((unsigned)((v.size() + 255) / 256)), dim3(256), 0, 0, din, dout, (int)v.size());
  std::vector<uint32_t> out(v.size());
  hipMemcpy(out.data(), dout, v.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (size_t i = 0; i < v.size(); i++)
  {
    const float x = v[i];
    uint32_t want;
    if (x != x) want = 0; // CVTPS2DQ(NaN) = 0x80000000 -> saturating pack to u8 -> 0
    else { const float r = nearbyintf(x); want = r < 0.0f ? 0u : (r > 255.0f ? 255u : (uint32_t)r); }
    if (out[i] != want) { if (bad < 10) printf("  x = %.9g (0x%08x): got %u want %u\n", x, *(const uint32_t *)&v[i], out[i], want); bad++; }
  }
  printf("{\"check\": \"v_cvt_pk_u8_f32 == clamp(rne(x), 0, 255)\", \"inputs\": %zu, \"mismatches\": %ld}\n", v.size(), bad);
  return 0;
}

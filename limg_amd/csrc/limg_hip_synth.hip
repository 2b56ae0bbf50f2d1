// limg_hip_synth.hip -- bench/test support kernels that are not part of the encode path proper:
//   * the integer-defined synthetic inputs of SURVEY.md 8(d) (random-gradient, photo-noise), generated straight into HBM;
//   * the perceptual error sum behind `limg_compare` (reference: src/limg.cpp:2455-2491, src/limg_internal.h:376-410).
#include "limg_hip_internal.h"

namespace limg_hip
{
  namespace
  {
    __device__ __forceinline__ uint64_t sm64(uint64_t x)
    {
      x += 0x9E3779B97F4A7C15ULL;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
      return x ^ (x >> 31);
    }

    __global__ __launch_bounds__(256) void k_synth_random_gradient(uint32_t *out, uint32_t w, uint32_t h, uint64_t seed, int opaque, uint32_t y0)
    {
      const uint64_t total = (uint64_t)w * h;
      for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
      {
        const uint64_t x = i % w, y = i / w + y0;
        const uint64_t hh = sm64(seed ^ ((y >> 6) * 0x9E3779B97F4A7C15ULL + (x >> 6)));
        const uint64_t h2 = sm64(hh);
        const int64_t gx = (int64_t)(h2 & 7), gy = (int64_t)((h2 >> 3) & 7);
        int64_t s = (int64_t)(x & 63) * gx + (int64_t)(y & 63) * gy, m = 63 * (gx + gy);
        if (m == 0) { m = 1; s = 0; }
        uint32_t p = 0;
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          const int64_t c0 = (int64_t)((hh >> (8 * c)) & 255), c1 = (int64_t)((hh >> (32 + 8 * c)) & 255);
          int64_t v = (c0 * (m - s) + c1 * s + m / 2) / m;
          if (c == 3 && opaque) v = 255;
          p |= (uint32_t)(v & 255) << (8 * c);
        }
        out[i] = p;
      }
    }

    __global__ __launch_bounds__(256) void k_synth_photo_noise(uint32_t *out, uint32_t w, uint32_t h, uint64_t seed, uint32_t y0)
    {
      const uint64_t total = (uint64_t)w * h;
      for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
      {
        const uint64_t x = i % w, y = i / w + y0;
        const uint64_t ly = y >> 5, lx = x >> 5;
        const int64_t fy = (int64_t)(y & 31), fx = (int64_t)(x & 31);
        const uint64_t ha = sm64(seed ^ (ly * 0x9E3779B97F4A7C15ULL + lx)), hb = sm64(seed ^ (ly * 0x9E3779B97F4A7C15ULL + lx + 1));
        const uint64_t hc = sm64(seed ^ ((ly + 1) * 0x9E3779B97F4A7C15ULL + lx)), hd = sm64(seed ^ ((ly + 1) * 0x9E3779B97F4A7C15ULL + lx + 1));
        const uint64_t hn = sm64(seed * 31 + y * (uint64_t)w + x);
        uint32_t p = 0xFF000000u;
#pragma unroll
        for (int c = 0; c < 3; c++)
        {
          const int64_t a = (int64_t)((ha >> (8 * c)) & 255), b = (int64_t)((hb >> (8 * c)) & 255), cc = (int64_t)((hc >> (8 * c)) & 255), d = (int64_t)((hd >> (8 * c)) & 255);
          int64_t v = ((a * (32 - fx) + b * fx) * (32 - fy) + (cc * (32 - fx) + d * fx) * fy + 512) >> 10;
          v += (int64_t)((hn >> (8 * c)) & 15) - 8;
          v = v < 0 ? 0 : (v > 255 ? 255 : v);
          p |= (uint32_t)v << (8 * c);
        }
        out[i] = p;
      }
    }

    template <int CH>
    __global__ __launch_bounds__(256) void k_compare(const uint32_t *a, const uint32_t *b, uint64_t count, unsigned long long *dSum)
    {
      unsigned long long acc = 0;
      for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
      {
        const uint32_t pa = a[i], pb = b[i];
        const int e0 = (int)(pa & 0xFF) - (int)(pb & 0xFF);
        const uint32_t red = (uint32_t)(e0 * e0);
        const bool low = red < 0x4000;
        const int e1 = (int)((pa >> 8) & 0xFF) - (int)((pb >> 8) & 0xFF), e2 = (int)((pa >> 16) & 0xFF) - (int)((pb >> 16) & 0xFF);
        uint32_t err = red * (low ? 2u : 3u) + (uint32_t)(e1 * e1) * 4u + (uint32_t)(e2 * e2) * (low ? 3u : 2u);
        if (CH == 4)
        {
          const int e3 = (int)(pa >> 24) - (int)(pb >> 24);
          err += (uint32_t)(e3 * e3) * 3u;
        }
        acc += err;
      }
      // wave reduction, then one atomic per wave
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
      if ((threadIdx.x & 63) == 0) atomicAdd(dSum, acc);
    }
  }

  void launch_synth_random_gradient(uint32_t *out, uint32_t w, uint32_t h, uint64_t seed, int opaque, uint32_t y0, hipStream_t s)
  {
    hipLaunchKernelGGL(k_synth_random_gradient, dim3(2048), dim3(256), 0, s, out, w, h, seed, opaque, y0);
  }

  void launch_synth_photo_noise(uint32_t *out, uint32_t w, uint32_t h, uint64_t seed, uint32_t y0, hipStream_t s)
  {
    hipLaunchKernelGGL(k_synth_photo_noise, dim3(2048), dim3(256), 0, s, out, w, h, seed, y0);
  }

  void launch_compare(const uint32_t *a, const uint32_t *b, uint64_t count, int channels, unsigned long long *dErrorSum, hipStream_t s)
  {
    if (channels == 4) hipLaunchKernelGGL(k_compare<4>, dim3(1024), dim3(256), 0, s, a, b, count, dErrorSum);
    else hipLaunchKernelGGL(k_compare<3>, dim3(1024), dim3(256), 0, s, a, b, count, dErrorSum);
  }
}

// limg_hip_noise_gpu.hip -- the context's dither noise table, filled ON the GPU.
//
// reference: limg_encode_dither_aes_sse41 src/limg.cpp:824-879 -- a dither call over a full 8x8 block starts from the chain value h, runs eight AESDEC rounds on the
// state {h, ~h} with the fixed round key of :837, takes from every round the low byte of each of the eight 16-bit lanes as the eight pixels' noise, and hands the low
// half of the final state to the next call.  The walk does not depend on the image, so the noise of a chain of full blocks is one constant stream,
// 64 bytes per call, which the F step indexes with the block's position in the chain.
//
// The chain is serial, and the context used to walk it on one host thread (~50 ms for the 3.1 M calls an 8192^2 image can make) and upload the 200 MB: a 100 ms
// cliff on the first encode of every new size class.  limg_noise_checkpoints.h holds the chain value at every 1024th call (tools/make_noise_checkpoints.py; re-walked
// by tests/test_host.py), so every stretch of 1024 calls is an independent job: one lane each, a software AESDEC round on T-tables in LDS (the same tables as the host's
// fallback in limg_hip_noise.cpp, built per workgroup from the inverse S-box), 64-byte stores.  ~1 ms for any table size up to the checkpoints' reach.
#include "limg_hip_internal.h"
#include "limg_noise_checkpoints.h"

namespace limg_hip
{
  namespace
  {
    __constant__ uint8_t d_inv_sbox[256] = {
      0x52, 0x09, 0x6a, 0xd5, 0x30, 0x36, 0xa5, 0x38, 0xbf, 0x40, 0xa3, 0x9e, 0x81, 0xf3, 0xd7, 0xfb, 0x7c, 0xe3, 0x39, 0x82, 0x9b, 0x2f, 0xff, 0x87, 0x34, 0x8e, 0x43, 0x44, 0xc4, 0xde, 0xe9, 0xcb,
      0x54, 0x7b, 0x94, 0x32, 0xa6, 0xc2, 0x23, 0x3d, 0xee, 0x4c, 0x95, 0x0b, 0x42, 0xfa, 0xc3, 0x4e, 0x08, 0x2e, 0xa1, 0x66, 0x28, 0xd9, 0x24, 0xb2, 0x76, 0x5b, 0xa2, 0x49, 0x6d, 0x8b, 0xd1, 0x25,
      0x72, 0xf8, 0xf6, 0x64, 0x86, 0x68, 0x98, 0x16, 0xd4, 0xa4, 0x5c, 0xcc, 0x5d, 0x65, 0xb6, 0x92, 0x6c, 0x70, 0x48, 0x50, 0xfd, 0xed, 0xb9, 0xda, 0x5e, 0x15, 0x46, 0x57, 0xa7, 0x8d, 0x9d, 0x84,
      0x90, 0xd8, 0xab, 0x00, 0x8c, 0xbc, 0xd3, 0x0a, 0xf7, 0xe4, 0x58, 0x05, 0xb8, 0xb3, 0x45, 0x06, 0xd0, 0x2c, 0x1e, 0x8f, 0xca, 0x3f, 0x0f, 0x02, 0xc1, 0xaf, 0xbd, 0x03, 0x01, 0x13, 0x8a, 0x6b,
      0x3a, 0x91, 0x11, 0x41, 0x4f, 0x67, 0xdc, 0xea, 0x97, 0xf2, 0xcf, 0xce, 0xf0, 0xb4, 0xe6, 0x73, 0x96, 0xac, 0x74, 0x22, 0xe7, 0xad, 0x35, 0x85, 0xe2, 0xf9, 0x37, 0xe8, 0x1c, 0x75, 0xdf, 0x6e,
      0x47, 0xf1, 0x1a, 0x71, 0x1d, 0x29, 0xc5, 0x89, 0x6f, 0xb7, 0x62, 0x0e, 0xaa, 0x18, 0xbe, 0x1b, 0xfc, 0x56, 0x3e, 0x4b, 0xc6, 0xd2, 0x79, 0x20, 0x9a, 0xdb, 0xc0, 0xfe, 0x78, 0xcd, 0x5a, 0xf4,
      0x1f, 0xdd, 0xa8, 0x33, 0x88, 0x07, 0xc7, 0x31, 0xb1, 0x12, 0x10, 0x59, 0x27, 0x80, 0xec, 0x5f, 0x60, 0x51, 0x7f, 0xa9, 0x19, 0xb5, 0x4a, 0x0d, 0x2d, 0xe5, 0x7a, 0x9f, 0x93, 0xc9, 0x9c, 0xef,
      0xa0, 0xe0, 0x3b, 0x4d, 0xae, 0x2a, 0xf5, 0xb0, 0xc8, 0xeb, 0xbb, 0x3c, 0x83, 0x53, 0x99, 0x61, 0x17, 0x2b, 0x04, 0x7e, 0xba, 0x77, 0xd6, 0x26, 0xe1, 0x69, 0x14, 0x63, 0x55, 0x21, 0x0c, 0x7d
    };

    __device__ __forceinline__ uint32_t xtime(uint32_t x) { return ((x << 1) ^ ((x >> 7) * 0x1Bu)) & 0xFFu; }

    // one lane = one stretch of LIMG_NOISE_CHECKPOINT_EVERY calls; a workgroup is one wave (few, long jobs: 3072 of them for an 8192^2 image's table)
    __global__ __launch_bounds__(64) void k_noise_fill(uint8_t *noise, const uint64_t *checkpoints, uint32_t segments, uint64_t count)
    {
      // Td_r[x]: the InvMixColumns contribution of InvSubBytes(x) sitting in row r, as a little-endian column word (FIPS-197 5.3; same construction as the host's)
      __shared__ uint32_t td[4][256];
      for (uint32_t x = threadIdx.x; x < 256; x += 64)
      {
        const uint32_t s = d_inv_sbox[x], s2 = xtime(s), s4 = xtime(s2), s8 = xtime(s4);
        const uint32_t m9 = s8 ^ s, m11 = s8 ^ s2 ^ s, m13 = s8 ^ s4 ^ s, m14 = s8 ^ s4 ^ s2;
        td[0][x] = m14 | (m9 << 8) | (m13 << 16) | (m11 << 24);
        td[1][x] = m11 | (m14 << 8) | (m9 << 16) | (m13 << 24);
        td[2][x] = m13 | (m11 << 8) | (m14 << 16) | (m9 << 24);
        td[3][x] = m9 | (m13 << 8) | (m11 << 16) | (m14 << 24);
      }
      __syncthreads();
      const uint32_t seg = blockIdx.x * 64u + threadIdx.x;
      if (seg >= segments) return;
      const uint32_t key[4] = { 0xAB705E1Du, 0x824A73EAu, 0x06CB4CADu, 0x2A76E980u }; // src/limg.cpp:837 as four little-endian column words
      unsigned long long h = checkpoints[seg];
      const uint64_t first = (uint64_t)seg * LIMG_NOISE_CHECKPOINT_EVERY;
      const uint64_t n = min((uint64_t)LIMG_NOISE_CHECKPOINT_EVERY, count - first);
      uint4 *out = reinterpret_cast<uint4 *>(noise + first * 64);
      for (uint64_t k = 0; k < n; k++)
      {
        uint32_t st[4] = { (uint32_t)h, (uint32_t)(h >> 32), ~(uint32_t)h, ~(uint32_t)(h >> 32) }; // {h, ~h}
        uint32_t bytes[16];
#pragma unroll
        for (int j = 0; j < 8; j++)
        { // AESDEC: InvShiftRows, InvSubBytes, InvMixColumns, xor round key; output column c takes row r from input column (c - r) & 3
          uint32_t o[4];
#pragma unroll
          for (int c = 0; c < 4; c++)
            o[c] = td[0][st[c] & 0xFFu] ^ td[1][(st[(c + 3) & 3] >> 8) & 0xFFu] ^ td[2][(st[(c + 2) & 3] >> 16) & 0xFFu] ^ td[3][st[(c + 1) & 3] >> 24] ^ key[c];
#pragma unroll
          for (int c = 0; c < 4; c++) st[c] = o[c];
          // the eight pixels of this round: the low byte of each 16-bit lane of the state
          bytes[2 * j] = __builtin_amdgcn_perm(st[1], st[0], 0x06040200u);
          bytes[2 * j + 1] = __builtin_amdgcn_perm(st[3], st[2], 0x06040200u);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) out[k * 4 + q] = make_uint4(bytes[4 * q], bytes[4 * q + 1], bytes[4 * q + 2], bytes[4 * q + 3]);
        h = (unsigned long long)st[0] | ((unsigned long long)st[1] << 32);
      }
    }
  }

  namespace
  {
    // Images with partial edge blocks: the chain is data dependent (a call over N pixels is N / 8 AES rounds + N % 8 PCG steps, src/limg.cpp:824-879), so the host walks
    // it -- but only the chain VALUES: what it uploads per dither call is the value the call starts from (8 bytes) and its pixel count (1 byte), and this kernel
    // expands them into the 64 noise bytes per call the F step indexes.  One lane per call.  pcg: the reference's PCG dither (every pixel a PCG step, :799-822).
    __global__ __launch_bounds__(256) void k_noise_expand(uint8_t *noise, const unsigned long long *states, const uint8_t *pixels, uint32_t calls, int pcg)
    {
      __shared__ uint32_t td[4][256];
      for (uint32_t x = threadIdx.x; x < 256; x += 256)
      {
        const uint32_t s = d_inv_sbox[x], s2 = xtime(s), s4 = xtime(s2), s8 = xtime(s4);
        const uint32_t m9 = s8 ^ s, m11 = s8 ^ s2 ^ s, m13 = s8 ^ s4 ^ s, m14 = s8 ^ s4 ^ s2;
        td[0][x] = m14 | (m9 << 8) | (m13 << 16) | (m11 << 24);
        td[1][x] = m11 | (m14 << 8) | (m9 << 16) | (m13 << 24);
        td[2][x] = m13 | (m11 << 8) | (m14 << 16) | (m9 << 24);
        td[3][x] = m9 | (m13 << 8) | (m11 << 16) | (m14 << 24);
      }
      __syncthreads();
      const uint32_t k = blockIdx.x * 256u + threadIdx.x;
      if (k >= calls) return;
      const uint32_t key[4] = { 0xAB705E1Du, 0x824A73EAu, 0x06CB4CADu, 0x2A76E980u };
      unsigned long long h = states[k];
      const uint32_t n = pixels[k], rounds = (!pcg && n >= 8u) ? n / 8u : 0u;
      uint32_t bytes[16];
#pragma unroll
      for (int i = 0; i < 16; i++) bytes[i] = 0u;
      uint32_t st[4] = { (uint32_t)h, (uint32_t)(h >> 32), ~(uint32_t)h, ~(uint32_t)(h >> 32) };
#pragma unroll
      for (int j = 0; j < 8; j++)
      {
        if ((uint32_t)j < rounds)
        {
          uint32_t o[4];
#pragma unroll
          for (int c = 0; c < 4; c++)
            o[c] = td[0][st[c] & 0xFFu] ^ td[1][(st[(c + 3) & 3] >> 8) & 0xFFu] ^ td[2][(st[(c + 2) & 3] >> 16) & 0xFFu] ^ td[3][st[(c + 1) & 3] >> 24] ^ key[c];
#pragma unroll
          for (int c = 0; c < 4; c++) st[c] = o[c];
          bytes[2 * j] = __builtin_amdgcn_perm(st[1], st[0], 0x06040200u);
          bytes[2 * j + 1] = __builtin_amdgcn_perm(st[3], st[2], 0x06040200u);
        }
      }
      if (rounds) h = (unsigned long long)st[0] | ((unsigned long long)st[1] << 32);
      uint4 *out = reinterpret_cast<uint4 *>(noise + (size_t)k * 64);
#pragma unroll
      for (int q = 0; q < 4; q++) out[q] = make_uint4(bytes[4 * q], bytes[4 * q + 1], bytes[4 * q + 2], bytes[4 * q + 3]);
      // the pixels the AES rounds do not cover: PCG steps on the chain value (src/limg.cpp:866-876); rare (blocks whose pixel count is not a multiple of 8), so byte stores
      for (uint32_t i = rounds * 8u; i < n; i++)
      {
        h = h * 6364136223846793005ULL + 1ULL;
        const uint32_t xs = (uint32_t)(((h >> 18) ^ h) >> 27), rot = (uint32_t)(h >> 59);
        noise[(size_t)k * 64 + i] = (uint8_t)((xs >> rot) | (xs << ((0u - rot) & 31u)));
      }
    }
  }

  namespace
  {
    // The merged-block encoder's dither calls run over whole rectangles (N = 64 rx ry pixels minus what the image's edges cut off): N / 8 AES rounds in a row, then N % 8
    // PCG steps (src/limg.cpp:824-879 with rangeSize = N; :1541-1551).  One lane per call again -- the rounds of a call are a serial chain -- writing the call's N noise
    // bytes (one per pixel) at its offset in the noise buffer.  The host walks the same chain, but only for the values the calls start from.
    __global__ __launch_bounds__(256) void k_noise_expand_calls(uint8_t *noise, const unsigned long long *states, const unsigned long long *offsets, const uint32_t *pixels,
                                                               uint32_t calls, int pcg)
    {
      __shared__ uint32_t td[4][256];
      {
        const uint32_t x = threadIdx.x;
        const uint32_t s = d_inv_sbox[x], s2 = xtime(s), s4 = xtime(s2), s8 = xtime(s4);
        const uint32_t m9 = s8 ^ s, m11 = s8 ^ s2 ^ s, m13 = s8 ^ s4 ^ s, m14 = s8 ^ s4 ^ s2;
        td[0][x] = m14 | (m9 << 8) | (m13 << 16) | (m11 << 24);
        td[1][x] = m11 | (m14 << 8) | (m9 << 16) | (m13 << 24);
        td[2][x] = m13 | (m11 << 8) | (m14 << 16) | (m9 << 24);
        td[3][x] = m9 | (m13 << 8) | (m11 << 16) | (m14 << 24);
      }
      __syncthreads();
      const uint32_t k = blockIdx.x * 256u + threadIdx.x;
      if (k >= calls) return;
      const uint32_t key[4] = { 0xAB705E1Du, 0x824A73EAu, 0x06CB4CADu, 0x2A76E980u };
      unsigned long long h = states[k];
      const uint32_t n = pixels[k], rounds = (!pcg && n >= 8u) ? n / 8u : 0u;
      uint8_t *out = noise + offsets[k];
      uint32_t st[4] = { (uint32_t)h, (uint32_t)(h >> 32), ~(uint32_t)h, ~(uint32_t)(h >> 32) };
      const bool aligned = (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
      for (uint32_t j = 0; j < rounds; j++)
      {
        uint32_t o[4];
#pragma unroll
        for (int c = 0; c < 4; c++)
          o[c] = td[0][st[c] & 0xFFu] ^ td[1][(st[(c + 3) & 3] >> 8) & 0xFFu] ^ td[2][(st[(c + 2) & 3] >> 16) & 0xFFu] ^ td[3][st[(c + 1) & 3] >> 24] ^ key[c];
#pragma unroll
        for (int c = 0; c < 4; c++) st[c] = o[c];
        const uint32_t lo = __builtin_amdgcn_perm(st[1], st[0], 0x06040200u), hi = __builtin_amdgcn_perm(st[3], st[2], 0x06040200u);
        if (aligned) *reinterpret_cast<uint2 *>(out + 8 * (size_t)j) = make_uint2(lo, hi);
        else
        {
#pragma unroll
          for (int b = 0; b < 4; b++) { out[8 * (size_t)j + b] = (uint8_t)(lo >> (8 * b)); out[8 * (size_t)j + 4 + b] = (uint8_t)(hi >> (8 * b)); }
        }
      }
      if (rounds) h = (unsigned long long)st[0] | ((unsigned long long)st[1] << 32);
      for (uint32_t i = rounds * 8u; i < n; i++)
      {
        h = h * 6364136223846793005ULL + 1ULL;
        const uint32_t xs = (uint32_t)(((h >> 18) ^ h) >> 27), rot = (uint32_t)(h >> 59);
        out[i] = (uint8_t)((xs >> rot) | (xs << ((0u - rot) & 31u)));
      }
    }
  }

  void launch_noise_expand_calls(uint8_t *noise, const unsigned long long *dStates, const unsigned long long *dOffsets, const uint32_t *dPixels, size_t calls, bool pcg, hipStream_t s)
  {
    if (calls == 0) return;
    hipLaunchKernelGGL(k_noise_expand_calls, dim3((uint32_t)((calls + 255u) / 256u)), dim3(256), 0, s, noise, dStates, dOffsets, dPixels, (uint32_t)calls, pcg ? 1 : 0);
  }

  void launch_noise_expand(uint8_t *noise, const unsigned long long *dStates, const uint8_t *dPixels, size_t calls, bool pcg, hipStream_t s)
  {
    if (calls == 0) return;
    hipLaunchKernelGGL(k_noise_expand, dim3((uint32_t)((calls + 255u) / 256u)), dim3(256), 0, s, noise, dStates, dPixels, (uint32_t)calls, pcg ? 1 : 0);
  }

  const uint64_t *noise_checkpoints_host(size_t *pCount, size_t *pEvery)
  {
    static const uint64_t table[LIMG_NOISE_CHECKPOINT_COUNT] = LIMG_NOISE_CHECKPOINTS_INIT;
    if (pCount) *pCount = LIMG_NOISE_CHECKPOINT_COUNT;
    if (pEvery) *pEvery = LIMG_NOISE_CHECKPOINT_EVERY;
    return table;
  }

  // the chain value every LIMG_NOISE_FAR_EVERY calls, through 2^27 calls: what reaches beyond the dense table (limg_hip_api.hip ensure_checkpoints turns the far values an
  // image needs into dense ones: LIMG_NOISE_FAR_EVERY calls on foot per far value, on host threads)
  const uint64_t *noise_checkpoints_far_host(size_t *pCount, size_t *pEvery)
  {
    static const uint64_t table[LIMG_NOISE_FAR_COUNT] = LIMG_NOISE_FAR_INIT;
    if (pCount) *pCount = LIMG_NOISE_FAR_COUNT;
    if (pEvery) *pEvery = LIMG_NOISE_FAR_EVERY;
    return table;
  }

  // entries [0, count) of the AES noise stream into `noise` (device); dCheckpoints = dense checkpoints (every LIMG_NOISE_CHECKPOINT_EVERY calls) covering them, on the device.
  void launch_noise_fill(uint8_t *noise, const uint64_t *dCheckpoints, size_t count, hipStream_t s)
  {
    const uint32_t segments = (uint32_t)((count + LIMG_NOISE_CHECKPOINT_EVERY - 1) / LIMG_NOISE_CHECKPOINT_EVERY);
    if (segments == 0) return;
    hipLaunchKernelGGL(k_noise_fill, dim3((segments + 63u) / 64u), dim3(64), 0, s, noise, dCheckpoints, segments, (uint64_t)count);
  }
}

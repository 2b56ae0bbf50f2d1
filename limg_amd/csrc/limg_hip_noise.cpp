// limg_hip_noise.cpp -- host side of the dither chain (reference: src/limg.cpp:824-879 AES path, :799-822 PCG tail).
//
// The reference's dither PRNG state walk is independent of the image data and of the shift: the n-th dither call of a
// chain starts from G^n(seed0) where G = "eight AESDEC rounds on {h, ~h}, keep the low 64 bits" for a full 8x8 block
// (verified against the reference by tests/golden/chain.json).  So the noise every full block adds
// is a constant stream: call k, pixel p -> the low byte of 16-bit lane (p & 7) of the state after round (p >> 3) + 1.
// This file produces that stream (64 bytes per call) once per context; the kernels index it with the exclusive scan of
// the per-block dither-call counts.  For images with partial edge blocks the chain depends on the data (G_N differs per
// block size): `limg_hip_walk_chain` then evaluates it sequentially from the per-block call counts.
//
// Host-only translation unit (no HIP): uses AES-NI when the CPU has it, a T-table software round otherwise.
#include <stdint.h>
#include <string.h>
#include <stddef.h>

#include <mutex>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace limg_hip
{
  namespace
  {
    const uint8_t kInvSbox[256] = {
      0x52, 0x09, 0x6a, 0xd5, 0x30, 0x36, 0xa5, 0x38, 0xbf, 0x40, 0xa3, 0x9e, 0x81, 0xf3, 0xd7, 0xfb, 0x7c, 0xe3, 0x39, 0x82, 0x9b, 0x2f, 0xff, 0x87, 0x34, 0x8e, 0x43, 0x44, 0xc4, 0xde, 0xe9, 0xcb,
      0x54, 0x7b, 0x94, 0x32, 0xa6, 0xc2, 0x23, 0x3d, 0xee, 0x4c, 0x95, 0x0b, 0x42, 0xfa, 0xc3, 0x4e, 0x08, 0x2e, 0xa1, 0x66, 0x28, 0xd9, 0x24, 0xb2, 0x76, 0x5b, 0xa2, 0x49, 0x6d, 0x8b, 0xd1, 0x25,
      0x72, 0xf8, 0xf6, 0x64, 0x86, 0x68, 0x98, 0x16, 0xd4, 0xa4, 0x5c, 0xcc, 0x5d, 0x65, 0xb6, 0x92, 0x6c, 0x70, 0x48, 0x50, 0xfd, 0xed, 0xb9, 0xda, 0x5e, 0x15, 0x46, 0x57, 0xa7, 0x8d, 0x9d, 0x84,
      0x90, 0xd8, 0xab, 0x00, 0x8c, 0xbc, 0xd3, 0x0a, 0xf7, 0xe4, 0x58, 0x05, 0xb8, 0xb3, 0x45, 0x06, 0xd0, 0x2c, 0x1e, 0x8f, 0xca, 0x3f, 0x0f, 0x02, 0xc1, 0xaf, 0xbd, 0x03, 0x01, 0x13, 0x8a, 0x6b,
      0x3a, 0x91, 0x11, 0x41, 0x4f, 0x67, 0xdc, 0xea, 0x97, 0xf2, 0xcf, 0xce, 0xf0, 0xb4, 0xe6, 0x73, 0x96, 0xac, 0x74, 0x22, 0xe7, 0xad, 0x35, 0x85, 0xe2, 0xf9, 0x37, 0xe8, 0x1c, 0x75, 0xdf, 0x6e,
      0x47, 0xf1, 0x1a, 0x71, 0x1d, 0x29, 0xc5, 0x89, 0x6f, 0xb7, 0x62, 0x0e, 0xaa, 0x18, 0xbe, 0x1b, 0xfc, 0x56, 0x3e, 0x4b, 0xc6, 0xd2, 0x79, 0x20, 0x9a, 0xdb, 0xc0, 0xfe, 0x78, 0xcd, 0x5a, 0xf4,
      0x1f, 0xdd, 0xa8, 0x33, 0x88, 0x07, 0xc7, 0x31, 0xb1, 0x12, 0x10, 0x59, 0x27, 0x80, 0xec, 0x5f, 0x60, 0x51, 0x7f, 0xa9, 0x19, 0xb5, 0x4a, 0x0d, 0x2d, 0xe5, 0x7a, 0x9f, 0x93, 0xc9, 0x9c, 0xef,
      0xa0, 0xe0, 0x3b, 0x4d, 0xae, 0x2a, 0xf5, 0xb0, 0xc8, 0xeb, 0xbb, 0x3c, 0x83, 0x53, 0x99, 0x61, 0x17, 0x2b, 0x04, 0x7e, 0xba, 0x77, 0xd6, 0x26, 0xe1, 0x69, 0x14, 0x63, 0x55, 0x21, 0x0c, 0x7d
    };

    // round key of src/limg.cpp:837: _mm_set_epi64x(0x2A76E98006CB4CAD, 0x824A73EAAB705E1D), as four little-endian column words
    const uint32_t kKey[4] = { 0xAB705E1Du, 0x824A73EAu, 0x06CB4CADu, 0x2A76E980u };

    uint32_t g_td[4][256]; // Td_r[x]: InvMixColumns contribution of InvSubBytes(x) sitting in row r, as a little-endian column word
    bool g_td_ready = false;

    inline uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1B)); }

    void build_tables_once()
    {
      for (int x = 0; x < 256; x++)
      {
        const uint8_t s = kInvSbox[x];
        const uint8_t s2 = xtime(s), s4 = xtime(s2), s8 = xtime(s4);
        const uint8_t m9 = (uint8_t)(s8 ^ s), m11 = (uint8_t)(s8 ^ s2 ^ s), m13 = (uint8_t)(s8 ^ s4 ^ s), m14 = (uint8_t)(s8 ^ s4 ^ s2);
        // InvMixColumns matrix rows: [14 11 13 9; 9 14 11 13; 13 9 14 11; 11 13 9 14]; input row r feeds matrix column r
        g_td[0][x] = (uint32_t)m14 | ((uint32_t)m9 << 8) | ((uint32_t)m13 << 16) | ((uint32_t)m11 << 24);
        g_td[1][x] = (uint32_t)m11 | ((uint32_t)m14 << 8) | ((uint32_t)m9 << 16) | ((uint32_t)m13 << 24);
        g_td[2][x] = (uint32_t)m13 | ((uint32_t)m11 << 8) | ((uint32_t)m14 << 16) | ((uint32_t)m9 << 24);
        g_td[3][x] = (uint32_t)m9 | ((uint32_t)m13 << 8) | ((uint32_t)m11 << 16) | ((uint32_t)m14 << 24);
      }
      g_td_ready = true;
    }

    void build_tables()
    { // contexts on different threads may get here together
      static std::once_flag once;
      std::call_once(once, build_tables_once);
    }

    // AESDEC: InvShiftRows, InvSubBytes, InvMixColumns, xor round key.  State = 4 little-endian column words.
    inline void aesdec_soft(uint32_t st[4])
    {
      uint32_t o[4];
      for (int c = 0; c < 4; c++)
      {
        // output column c takes row r from input column (c - r) & 3
        o[c] = g_td[0][st[c] & 0xFF] ^ g_td[1][(st[(c + 3) & 3] >> 8) & 0xFF] ^ g_td[2][(st[(c + 2) & 3] >> 16) & 0xFF] ^ g_td[3][(st[(c + 1) & 3] >> 24) & 0xFF] ^ kKey[c];
      }
      memcpy(st, o, 16);
    }

#if defined(__x86_64__)
    __attribute__((target("aes,sse4.1"))) void walk_aesni(uint64_t &h, unsigned rounds, uint8_t *noise)
    {
      const __m128i key = _mm_set_epi64x(0x2A76E98006CB4CADLL, (long long)0x824A73EAAB705E1DULL);
      const __m128i pick = _mm_set_epi8(-1, -1, -1, -1, -1, -1, -1, -1, 14, 12, 10, 8, 6, 4, 2, 0);
      __m128i st = _mm_set_epi64x((long long)~h, (long long)h);
      unsigned j = 0;
      if (noise && rounds >= 64 && (reinterpret_cast<uintptr_t>(noise) & 7u) == 0)
      { // long calls (rectangles of the merged-block encoder): the bytes are written once and read by the GPU only -- stream them past the
        // caches the concurrently running merge lives in, as whole 64-byte lines
        for (; j < rounds && (reinterpret_cast<uintptr_t>(noise + 8 * j) & 63u); j++)
        {
          st = _mm_aesdec_si128(st, key);
          _mm_storel_epi64(reinterpret_cast<__m128i *>(noise + 8 * j), _mm_shuffle_epi8(st, pick));
        }
        for (; j + 8 <= rounds; j += 8)
        {
          __m128i *line = reinterpret_cast<__m128i *>(noise + 8 * j);
          for (int k = 0; k < 4; k++)
          {
            st = _mm_aesdec_si128(st, key);
            const __m128i lo = _mm_shuffle_epi8(st, pick);
            st = _mm_aesdec_si128(st, key);
            _mm_stream_si128(line + k, _mm_unpacklo_epi64(lo, _mm_shuffle_epi8(st, pick)));
          }
        }
        _mm_sfence();
      }
      for (; j < rounds; j++)
      {
        st = _mm_aesdec_si128(st, key);
        if (noise) _mm_storel_epi64(reinterpret_cast<__m128i *>(noise + 8 * j), _mm_shuffle_epi8(st, pick));
      }
      h = (uint64_t)_mm_cvtsi128_si64(st);
    }
#endif

    void walk_soft(uint64_t &h, unsigned rounds, uint8_t *noise)
    {
      build_tables();
      uint32_t st[4];
      const uint64_t inv = ~h;
      memcpy(st, &h, 8);
      memcpy(st + 2, &inv, 8);
      for (unsigned j = 0; j < rounds; j++)
      {
        aesdec_soft(st);
        if (noise)
          for (int i = 0; i < 8; i++) noise[8 * j + i] = (uint8_t)(st[i >> 1] >> (16 * (i & 1)));
      }
      memcpy(&h, st, 8);
    }

    bool have_aesni()
    {
#if defined(__x86_64__)
      static const bool v = __builtin_cpu_supports("aes") && __builtin_cpu_supports("sse4.1");
      return v;
#else
      return false;
#endif
    }

    inline uint32_t pcg_step(uint64_t &h)
    { // src/limg.cpp:866-871
      h = h * 6364136223846793005ULL + 1;
      const uint32_t xorshifted_hi = (uint32_t)(((h >> 18) ^ h) >> 27);
      const uint32_t rot_hi = (uint32_t)(h >> 59);
      return (xorshifted_hi >> rot_hi) | (xorshifted_hi << ((uint32_t)(-(int32_t)rot_hi) & 31));
    }
  }

  // One dither call over `n` pixels starting from chain value `h`: writes n noise bytes (padded to 64), returns G_n(h).
  // pcg == true: the reference's fallback for hosts without AES-NI (src/limg.cpp:799-822): every pixel is a PCG step.
  uint64_t chain_call(uint64_t h, unsigned n, uint8_t *noise64, bool forceSoft, bool pcg)
  {
    const unsigned rounds = (!pcg && n >= 8) ? n / 8 : 0;
    if (rounds)
    {
#if defined(__x86_64__)
      if (have_aesni() && !forceSoft) walk_aesni(h, rounds, noise64);
      else
#endif
        walk_soft(h, rounds, noise64);
    }
    for (unsigned i = rounds * 8; i < n; i++)
    {
      const uint32_t r = pcg_step(h);
      if (noise64) noise64[i] = (uint8_t)r;
    }
    return h;
  }

  // The merged-block encoder's chain walk for one batch of rectangles, in creation order (src/limg.cpp:1541-1551 calls limg_encode_dither once per dithered
  // factor of a rectangle, :824-879): per dither call the value it starts from, where its noise bytes go and its pixel count are recorded (the device re-runs the
  // rounds: k_noise_expand_calls) and the chain moves on by G_N.  One function so that the AESDEC rounds sit inline in the loop: the walk is a single dependent chain
  // (~4 cycles per round), and a call per dither call -- set-up of {h, ~h}, dispatch on the CPU's features, the PCG tail's loop -- cost a third of it on rectangles
  // of a few blocks.  shiftWords: the rectangles' result words, `stride` bytes apart, dither calls in bits 24..31.
#if defined(__x86_64__)
  __attribute__((target("aes,sse4.1"))) static uint64_t walk_batch_aesni(uint64_t h, size_t count, const uint8_t *shiftWords, size_t stride, const uint32_t *npx,
                                                                        unsigned long long *noiseBase, unsigned long long *callState, unsigned long long *callOff,
                                                                        uint32_t *callPx, uint64_t &noiseOff, size_t &callCount, size_t maxCalls)
  {
    const __m128i key = _mm_set_epi64x(0x2A76E98006CB4CADLL, (long long)0x824A73EAAB705E1DULL);
    for (size_t i = 0; i < count; i++)
    {
      noiseBase[i] = noiseOff;
      uint32_t word;
      memcpy(&word, shiftWords + i * stride, 4);
      const uint32_t calls = word >> 24, n = npx[i], rounds = n >= 8 ? n / 8 : 0, tail = n - rounds * 8;
      for (uint32_t k = 0; k < calls && callCount < maxCalls; k++, noiseOff += n, callCount++)
      {
        callState[callCount] = h; callOff[callCount] = noiseOff; callPx[callCount] = n;
        if (rounds)
        {
          __m128i st = _mm_set_epi64x((long long)~h, (long long)h);
          for (uint32_t j = 0; j < rounds; j++) st = _mm_aesdec_si128(st, key);
          h = (uint64_t)_mm_cvtsi128_si64(st);
        }
        for (uint32_t j = 0; j < tail; j++) (void)pcg_step(h);
      }
    }
    return h;
  }
#endif

  // The 8x8 path's chain walk over an image with partial edge blocks (limg_hip_api.hip: the whole image, or the last block row of an image whose width is whole
  // blocks), raster order, chains restarting at the strip partition's boundaries (src/limg.cpp:2114-2134): same recording as above, per dither call the value it starts
  // from and its pixel count; per work strip (32 blocks) the index of its first call.  Template on the AES-NI use so that the rounds are inline in the hot variant.
  namespace
  {
    template <bool NI>
#if defined(__x86_64__)
    __attribute__((target("aes,sse4.1")))
#endif
    void walk_blocks_impl(uint64_t &hIo, size_t &callIo, uint32_t by0, uint32_t by1, uint32_t blocksX, uint32_t stripsX, size_t sizeX, size_t sizeY, uint32_t chainCount, uint32_t chainRows,
                          const uint32_t *shifts, uint32_t *stripBase, unsigned long long *states, uint8_t *pixels, size_t maxCalls, bool pcg)
    {
#if defined(__x86_64__)
      const __m128i key = _mm_set_epi64x(0x2A76E98006CB4CADLL, (long long)0x824A73EAAB705E1DULL);
#endif
      auto chain_of = [&](uint32_t row) -> uint32_t { if (chainCount <= 1 || chainRows == 0) return 0u; const uint32_t c = row / chainRows; return c < chainCount - 1 ? c : chainCount - 1; };
      uint64_t h = hIo;
      size_t call = callIo;
      for (uint32_t by = by0; by < by1; by++)
      {
        if (by != 0 && chain_of(by) != chain_of(by - 1)) h = 0xCA7F00D15BADF00DULL; // src/limg.cpp:1893
        const unsigned ry = (unsigned)((sizeY - (size_t)by * 8) < 8 ? (sizeY - (size_t)by * 8) : 8);
        for (uint32_t bx = 0; bx < blocksX; bx++)
        {
          if (bx % 32u == 0) stripBase[(size_t)by * stripsX + bx / 32u] = (uint32_t)call;
          const unsigned rx = (unsigned)((sizeX - (size_t)bx * 8) < 8 ? (sizeX - (size_t)bx * 8) : 8);
          const uint32_t n = rx * ry, calls = shifts[(size_t)by * blocksX + bx] >> 24, rounds = (!pcg && n >= 8) ? n / 8 : 0, tail = n - rounds * 8;
          for (uint32_t k = 0; k < calls && call < maxCalls; k++, call++)
          {
            states[call] = h; pixels[call] = (uint8_t)n;
            if (NI)
            {
#if defined(__x86_64__)
              if (rounds)
              {
                __m128i st = _mm_set_epi64x((long long)~h, (long long)h);
                for (uint32_t j = 0; j < rounds; j++) st = _mm_aesdec_si128(st, key);
                h = (uint64_t)_mm_cvtsi128_si64(st);
              }
              for (uint32_t j = 0; j < tail; j++) (void)pcg_step(h);
#endif
            }
            else h = chain_call(h, n, nullptr, false, pcg);
          }
        }
      }
      hIo = h; callIo = call;
    }
  }

  // Block rows [by0, by1) of the walk: continues from chain value `h` at call index `call` (both updated), restarts the chain where the partition says so.  The rows of
  // an image may be walked in pieces, in order (the band pipeline of limg_hip_api.hip), and chains that start at the seed may be walked by different threads at once.
  void chain_walk_rows(uint64_t &h, size_t &call, uint32_t by0, uint32_t by1, uint32_t blocksX, uint32_t stripsX, size_t sizeX, size_t sizeY, uint32_t chainCount, uint32_t chainRows,
                       const uint32_t *shifts, uint32_t *stripBase, unsigned long long *states, uint8_t *pixels, size_t maxCalls, bool pcg)
  {
#if defined(__x86_64__)
    if (!pcg && have_aesni()) { walk_blocks_impl<true>(h, call, by0, by1, blocksX, stripsX, sizeX, sizeY, chainCount, chainRows, shifts, stripBase, states, pixels, maxCalls, pcg); return; }
#endif
    walk_blocks_impl<false>(h, call, by0, by1, blocksX, stripsX, sizeX, sizeY, chainCount, chainRows, shifts, stripBase, states, pixels, maxCalls, pcg);
  }

  size_t chain_walk_blocks(uint64_t h0, uint32_t blocksX, uint32_t blocksY, uint32_t stripsX, size_t sizeX, size_t sizeY, uint32_t chainCount, uint32_t chainRows, const uint32_t *shifts,
                           uint32_t *stripBase, unsigned long long *states, uint8_t *pixels, size_t maxCalls, bool pcg)
  {
    uint64_t h = h0;
    size_t call = 0;
    chain_walk_rows(h, call, 0, blocksY, blocksX, stripsX, sizeX, sizeY, chainCount, chainRows, shifts, stripBase, states, pixels, maxCalls, pcg);
    return call;
  }

  uint64_t chain_walk_batch(uint64_t h, size_t count, const uint8_t *shiftWords, size_t stride, const uint32_t *npx, unsigned long long *noiseBase, unsigned long long *callState,
                            unsigned long long *callOff, uint32_t *callPx, uint64_t &noiseOff, size_t &callCount, size_t maxCalls, bool pcg)
  {
#if defined(__x86_64__)
    if (!pcg && have_aesni()) return walk_batch_aesni(h, count, shiftWords, stride, npx, noiseBase, callState, callOff, callPx, noiseOff, callCount, maxCalls);
#endif
    for (size_t i = 0; i < count; i++)
    {
      noiseBase[i] = noiseOff;
      uint32_t word;
      memcpy(&word, shiftWords + i * stride, 4);
      const uint32_t calls = word >> 24;
      for (uint32_t k = 0; k < calls && callCount < maxCalls; k++, noiseOff += npx[i], callCount++)
      {
        callState[callCount] = h; callOff[callCount] = noiseOff; callPx[callCount] = npx[i];
        h = chain_call(h, npx[i], nullptr, false, pcg);
      }
    }
    return h;
  }

  // Chain values of a chain of full-block calls: pOut[i] = the value call number i * every starts from (pOut[0] = h), for i * every < calls.  Returns the value
  // after `calls` calls.  (Generator and checker of limg_noise_checkpoints.h.)
  uint64_t chain_checkpoints(uint64_t h, size_t calls, size_t every, uint64_t *pOut, bool pcg)
  {
    for (size_t k = 0; k < calls; k++)
    {
      if (pOut && every && k % every == 0) pOut[k / every] = h;
      h = chain_call(h, 64, nullptr, false, pcg);
    }
    return h;
  }

  // Static table for chains made of full 8x8 blocks only: entries [first, first + count) given the chain value at `first`.
  // Returns the chain value after the last generated entry (so the table can be grown later).
  uint64_t fill_noise_table(uint64_t h, uint8_t *noise, size_t count, bool pcg)
  {
    for (size_t k = 0; k < count; k++) h = chain_call(h, 64, noise + k * 64, false, pcg);
    return h;
  }
}

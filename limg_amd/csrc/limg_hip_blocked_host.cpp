// limg_hip_blocked_host.cpp -- host stages of the merged-block encoder (reference: limg_blocked_encode3d_test, src/limg.cpp:1774-1885).
//
// The reference merges 8x8 blocks greedily in raster order: a seed block grows a rectangle right / down (and, for a second attempt
// from the centre third, in all four directions) while every block of the next row / column strip is unused and "matches" the seed
// (src/limg.cpp:1288-1496).  Which rectangle a seed gets depends on everything claimed before it, so the scan itself is serial; the
// expensive part -- the similarity predicate, ~2500 flops a pair -- is not: the GPU evaluates it for every block against its
// 18 x 18 neighbourhood (k_blocked_match; offsets -5 .. +12) and this scan only looks bits up.  Pairs outside that window are evaluated
// here, with the same float operations in the same order (built with -ffp-contract=off, like the kernels).
//
// Host-only translation unit (no HIP).
#include <math.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <stdint.h>
#include <string.h>

#include <functional>
#include <vector>

#include "../../include/limg_hip.h"

namespace limg_hip
{
  constexpr int kMatchLo = 5, kMatchHi = 12; // keep in sync with limg_hip_internal.h
  constexpr int kMatchSide = kMatchLo + kMatchHi + 1;
  constexpr int kMatchWords = (kMatchSide * kMatchSide + 63) / 64;

  struct HostRegion { uint32_t ox, oy, rx, ry, keep; };

  uint64_t chain_call(uint64_t h, unsigned n, uint8_t *noise, bool forceSoft, bool pcg); // limg_hip_noise.cpp (any n; `noise` holds n bytes)

  namespace
  {
    // limg_color_error_state_3d (src/limg_internal.h:426-452) with limg_dot's serial order (:357-366)
    struct State { float nA[4], nB[4], nC[4], invA, invB, invC; };

    inline float dot(const float *a, const float *b, int ch)
    {
      float sum = 0.0f;
      for (int i = 0; i < ch; i++) sum += a[i] * b[i];
      return sum;
    }

    void init_state(const limg_hip_block_record &r, int ch, State &s)
    {
      memset(&s, 0, sizeof(s));
      bool nz[3] = { false, false, false };
      for (int i = 0; i < ch; i++)
      {
        s.nA[i] = (float)((int)r.dirA_max[i] - (int)r.dirA_min[i]);
        s.nB[i] = (float)((int)r.dirB_mag[i] - (int)r.dirB_offset[i]);
        s.nC[i] = (float)((int)r.dirC_mag[i] - (int)r.dirC_offset[i]);
        nz[0] |= s.nA[i] != 0; nz[1] |= s.nB[i] != 0; nz[2] |= s.nC[i] != 0;
      }
      if (nz[0]) s.invA = 1.0f / dot(s.nA, s.nA, ch);
      if (nz[1]) s.invB = 1.0f / dot(s.nB, s.nB, ch);
      if (nz[2]) s.invC = 1.0f / dot(s.nC, s.nC, ch);
    }

    // src/limg_factorization.h:9-42
    void colour_factors(const float *color, const limg_hip_block_record &in, const State &s, int ch, float f[3])
    {
      float t[4], est[4];
      for (int i = 0; i < ch; i++) t[i] = color[i] - (float)in.dirA_min[i];
      f[0] = dot(t, s.nA, ch) * s.invA;
      for (int i = 0; i < ch; i++) { est[i] = (float)in.dirA_min[i] + f[0] * s.nA[i]; t[i] = (color[i] - est[i]) - (float)in.dirB_offset[i]; }
      f[1] = dot(t, s.nB, ch) * s.invB;
      for (int i = 0; i < ch; i++) { est[i] = est[i] + f[1] * s.nB[i]; t[i] = (color[i] - est[i]) - (float)in.dirC_offset[i]; }
      f[2] = dot(t, s.nC, ch) * s.invC;
    }
  }

  namespace
  {
    // The 27 colours of src/limg.cpp:1219-1243 (x fastest, then y, then z; factors 0, 0.5, 1) projected into the seed's basis: the per-iteration
    // term  |fa| / lenA0 + |0.5 - fb| * 2 / lenA1 + |0.5 - fc| * 2 / lenA2.  The iterations are independent (only their sum is ordered), so they can be
    // evaluated side by side -- every lane performs exactly the scalar operation sequence.
    void terms27(const limg_hip_block_record &a, const State &sa, const State &sb, int ch, const float invA[3], float out[32])
    {
      for (int i = 0; i < 27; i++)
      {
        const float xf = (i % 3) * 0.5f, yf = ((i / 3) % 3) * 0.5f, zf = (i / 9) * 0.5f;
        float color[4], fa[3];
        for (int c = 0; c < ch; c++) color[c] = sb.nA[c] * xf + sb.nB[c] * yf + sb.nC[c] * zf;
        colour_factors(color, a, sa, ch, fa);
        out[i] = fabsf(fa[0]) * invA[0] + fabsf(0.5f - fa[1]) * invA[1] + fabsf(0.5f - fa[2]) * invA[2];
      }
    }

#if defined(__x86_64__)
    __attribute__((target("avx2"))) void terms27_avx2(const limg_hip_block_record &a, const State &sa, const State &sb, int ch, const float invA[3], float out[32])
    {
      alignas(32) static const float XF[32] = { 0, .5f, 1, 0, .5f, 1, 0, .5f, 1, 0, .5f, 1, 0, .5f, 1, 0, .5f, 1, 0, .5f, 1, 0, .5f, 1, 0, .5f, 1, 0, 0, 0, 0, 0 };
      alignas(32) static const float YF[32] = { 0, 0, 0, .5f, .5f, .5f, 1, 1, 1, 0, 0, 0, .5f, .5f, .5f, 1, 1, 1, 0, 0, 0, .5f, .5f, .5f, 1, 1, 1, 0, 0, 0, 0, 0 };
      alignas(32) static const float ZF[32] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, .5f, .5f, .5f, .5f, .5f, .5f, .5f, .5f, .5f, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
      const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7FFFFFFF)), half = _mm256_set1_ps(0.5f), zero = _mm256_setzero_ps();
      for (int g = 0; g < 4; g++)
      {
        const __m256 xf = _mm256_load_ps(XF + 8 * g), yf = _mm256_load_ps(YF + 8 * g), zf = _mm256_load_ps(ZF + 8 * g);
        __m256 color[4], t[4], est[4];
        for (int c = 0; c < ch; c++)
          color[c] = _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(_mm256_set1_ps(sb.nA[c]), xf), _mm256_mul_ps(_mm256_set1_ps(sb.nB[c]), yf)), _mm256_mul_ps(_mm256_set1_ps(sb.nC[c]), zf));
        // colour_factors(color, a, sa): same operations, same order (limg_dot starts from 0 and adds the products in index order)
        __m256 dot = zero;
        for (int c = 0; c < ch; c++) { t[c] = _mm256_sub_ps(color[c], _mm256_set1_ps((float)a.dirA_min[c])); dot = _mm256_add_ps(dot, _mm256_mul_ps(t[c], _mm256_set1_ps(sa.nA[c]))); }
        const __m256 fa = _mm256_mul_ps(dot, _mm256_set1_ps(sa.invA));
        dot = zero;
        for (int c = 0; c < ch; c++)
        {
          est[c] = _mm256_add_ps(_mm256_set1_ps((float)a.dirA_min[c]), _mm256_mul_ps(fa, _mm256_set1_ps(sa.nA[c])));
          t[c] = _mm256_sub_ps(_mm256_sub_ps(color[c], est[c]), _mm256_set1_ps((float)a.dirB_offset[c]));
        }
        for (int c = 0; c < ch; c++) dot = _mm256_add_ps(dot, _mm256_mul_ps(t[c], _mm256_set1_ps(sa.nB[c])));
        const __m256 fb = _mm256_mul_ps(dot, _mm256_set1_ps(sa.invB));
        dot = zero;
        for (int c = 0; c < ch; c++)
        {
          est[c] = _mm256_add_ps(est[c], _mm256_mul_ps(fb, _mm256_set1_ps(sa.nB[c])));
          t[c] = _mm256_sub_ps(_mm256_sub_ps(color[c], est[c]), _mm256_set1_ps((float)a.dirC_offset[c]));
        }
        for (int c = 0; c < ch; c++) dot = _mm256_add_ps(dot, _mm256_mul_ps(t[c], _mm256_set1_ps(sa.nC[c])));
        const __m256 fc = _mm256_mul_ps(dot, _mm256_set1_ps(sa.invC));
        const __m256 term = _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(_mm256_and_ps(fa, absmask), _mm256_set1_ps(invA[0])),
                                                        _mm256_mul_ps(_mm256_and_ps(_mm256_sub_ps(half, fb), absmask), _mm256_set1_ps(invA[1]))),
                                          _mm256_mul_ps(_mm256_and_ps(_mm256_sub_ps(half, fc), absmask), _mm256_set1_ps(invA[2])));
        _mm256_storeu_ps(out + 8 * g, term);
      }
    }
#endif
  }

  // limg_encode_3d_matches (src/limg.cpp:1137-1268)
  bool blocked_matches_host(int ch, const limg_hip_block_record &a, const limg_hip_block_record &b)
  {
    State sa, sb;
    init_state(a, ch, sa);
    init_state(b, ch, sb);
    const float w[4] = { 2, 4, 3, 3 };
    float avgDiffSq = 0, lenA[3] = { 3, 3, 3 }, lenB[3] = { 3, 3, 3 };
    for (int i = 0; i < ch; i++)
    {
      const float d = a.avg[i] - b.avg[i];
      avgDiffSq += d * d * w[i];
      lenA[0] += (sa.nA[i] * sa.nA[i]) * w[i]; lenB[0] += (sb.nA[i] * sb.nA[i]) * w[i];
      lenA[1] += (sa.nB[i] * sa.nB[i]) * w[i]; lenB[1] += (sb.nB[i] * sb.nB[i]) * w[i];
      lenA[2] += (sa.nC[i] * sa.nC[i]) * w[i]; lenB[2] += (sb.nC[i] * sb.nC[i]) * w[i];
    }
    const float sumA = lenA[0] + lenA[1] + lenA[2], sumB = lenB[0] + lenB[1] + lenB[2];
    const float ratio = (sumA + 1) / (sumB + 1);
    const float maxAvg = (float)(16 * 3 * ch), maxRange = (float)(200 * 3 * ch);
    if (avgDiffSq < maxAvg && sumA < maxRange && sumB < maxRange) return true;
    if (ratio > 1.375f || ratio < (1.f / 1.375f)) return false;
    float invA[3], invB[3];
    for (int i = 0; i < 3; i++) { invA[i] = 1.0f / lenA[i]; invB[i] = 1.0f / lenB[i]; }
    for (int i = 1; i < 3; i++) { invA[i] *= 2.f; invB[i] *= 2.f; }
    float fb[3];
    colour_factors(a.avg, b, sb, ch, fb); // loop-invariant upstream (:1236-1239 builds a colour it then does not pass)
    const float termB = fabsf(fb[0]) * invB[0] + fabsf(0.5f - fb[1]) * invB[1] + fabsf(0.5f - fb[2]) * invB[2];
    float termA[32];
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) terms27_avx2(a, sa, sb, ch, invA, termA);
    else
#endif
      terms27(a, sa, sb, ch, invA, termA);
    float sum = 0;
    for (int i = 0; i < 27; i++) { sum += termA[i]; sum += termB; } // upstream's accumulation order: iteration by iteration, A term then B term
    return sum * (1.f / 27) < 3.0f;
  }

  namespace
  {
#ifndef LIMG_MERGE_PF_CENTRE
#define LIMG_MERGE_PF_CENTRE 4
#endif
    constexpr uint32_t kCentreAhead = LIMG_MERGE_PF_CENTRE; // (A/B hook: 0 = the look-ahead requests the seeds' own rows only)

    struct Merge
    {
      const limg_hip_block_record *rec;
      const unsigned long long *bits;
      uint32_t bx, by;
      int ch;
      std::vector<uint8_t> used;
      const std::function<void(uint32_t)> *needRow = nullptr; // the similarity bits arrive band by band: called before a seed row's bits are first read
      mutable uint32_t rowsSeen = 0;
      const std::function<void()> *needRecords = nullptr;     // the records themselves are only read for pairs outside the window (rare): called before the first such read
      mutable bool recordsSeen = false;
      mutable uint32_t pfRow = 0xFFFFFFFFu, pfCol = 0, pfAhead = 0; // look-ahead of `find` that requests the similarity rows of the seeds to come
      mutable bool pfTiny = false;
      const uint8_t *flags = nullptr; // per seed, from the GPU: bit 0 = a rectangle of >= 3 x 3 is possible at all, bit 1 = any rectangle is (necessary conditions)

      // One expansion (src/limg.cpp:1288-1384) from the seed at (ox, oy).  A strip joins when every block of it is unused
      // (src/limg.cpp:1121-1135) and matches the seed (:1271-1286); neither test has side effects, so they are fused per block.
      // minSide > 0 (first attempt of the large-rectangle pass only): give up as soon as the rectangle can no longer reach minSide x minSide --
      // it grows monotonically and a direction that failed stays off, so the caller would discard it anyway (src/limg.cpp:1425-1427).
      void expand(uint32_t &ox, uint32_t &oy, uint32_t &rx, uint32_t &ry, bool upLeft, uint32_t minSide = 0) const
      {
        const uint32_t sx = ox, sy = oy;
        if (needRow && sy >= rowsSeen) { (*needRow)(sy); rowsSeen = sy + 1; }
        const size_t seed = (size_t)sy * bx + sx;
        const unsigned long long *row = bits ? bits + seed * kMatchWords : nullptr;
        auto ok = [&](uint32_t cx, uint32_t cy) -> bool {
          const size_t ci = (size_t)cy * bx + cx;
          if (used[ci]) return false;
          const unsigned ux = cx - sx + kMatchLo, uy = cy - sy + kMatchLo; // wraps for offsets below -kMatchLo
          if (row && ux < (unsigned)kMatchSide && uy < (unsigned)kMatchSide)
          {
            const unsigned cell = uy * kMatchSide + ux;
            return (row[cell >> 6] >> (cell & 63)) & 1ull;
          }
          if (needRecords && !recordsSeen) { (*needRecords)(); recordsSeen = true; }
          return blocked_matches_host(ch, rec[seed], rec[ci]);
        };
        auto column = [&](uint32_t x, uint32_t y0, uint32_t n) { for (uint32_t i = 0; i < n; i++) if (!ok(x, y0 + i)) return false; return true; };
        auto line = [&](uint32_t y, uint32_t x0, uint32_t n) { for (uint32_t i = 0; i < n; i++) if (!ok(x0 + i, y)) return false; return true; };
        bool up = upLeft, down = true, left = upLeft, right = true;
        while (up || down || left || right)
        {
          if (right) { if (ox + rx + 1 < bx && column(ox + rx, oy, ry)) rx++; else { right = false; if (rx < minSide) return; } }
          if (down) { if (oy + ry + 1 < by && line(oy + ry, ox, rx)) ry++; else { down = false; if (ry < minSide) return; } }
          if (upLeft)
          {
            if (up) { if (oy > 0 && line(oy - 1, ox, rx)) { oy--; ry++; } else up = false; }
            if (left) { if (ox > 0 && column(ox - 1, oy, ry)) { ox--; rx++; } else left = false; }
          }
        }
      }
      // The same expansion for the common case -- right / down only, everything inside the precomputed window -- on ROW MASKS: bit dx of avail(dy) says
      // "block (sx + dx, sy + dy) is unused and matches the seed" (13 columns: offsets 0 .. kMatchHi).  A column joins when its bit is set in the AND of the rows
      // taken so far, a row when its mask covers the columns taken so far: a few operations per step instead of a loop over the strip's blocks, and a row is only
      // looked at when the rectangle gets there (most expansions end after one or two steps).  Returns false when the rectangle reaches the window's edge or the
      // generic form is needed: the caller then runs `expand` from scratch (same result by construction: both evaluate the same predicate on the same blocks).
      bool expand_fast(const uint32_t sx, const uint32_t sy, uint32_t &rx, uint32_t &ry, const uint32_t minSide) const
      {
        if (!bits) return false;
        if (needRow && sy >= rowsSeen) { (*needRow)(sy); rowsSeen = sy + 1; }
        const unsigned long long *row = bits + ((size_t)sy * bx + sx) * kMatchWords;
        const uint32_t wmax = bx - sx < (uint32_t)kMatchHi + 1u ? bx - sx : (uint32_t)kMatchHi + 1u; // columns of the window that exist
        const uint32_t hmax = by - sy < (uint32_t)kMatchHi + 1u ? by - sy : (uint32_t)kMatchHi + 1u;
        auto avail = [&](const uint32_t dy) -> uint32_t
        {
          const unsigned cell = (dy + kMatchLo) * kMatchSide + kMatchLo; // first of the row's 13 cells; 323 at most, so word + 1 exists whenever the run crosses a word
          const unsigned w = cell >> 6, sh = cell & 63;
          unsigned long long m = row[w] >> sh;
          if (sh > 64 - (kMatchHi + 1)) m |= row[w + 1] << (64 - sh);
          const uint8_t *u = &used[(size_t)(sy + dy) * bx + sx];
          uint32_t free_ = 0;
#if defined(__x86_64__)
          // 16 flag bytes at once (`used` carries 16 bytes of padding behind its last row); columns past the image's right edge belong to the next row: masked off
          free_ = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(u)), _mm_setzero_si128())) & ((1u << wmax) - 1u);
#else
          for (uint32_t dx = 0; dx < wmax; dx++) free_ |= (u[dx] ? 0u : 1u) << dx;
#endif
          return (uint32_t)m & free_ & ((1u << (kMatchHi + 1)) - 1u);
        };
        uint32_t colAnd = avail(0) | 1u; // (the seed itself: unused by construction, its own cell is never set)
        rx = 1; ry = 1;
        bool down = true, right = true;
        while (down || right)
        {
          if (right)
          {
            if (sx + rx + 1 < bx)
            {
              if (rx >= wmax || rx > (uint32_t)kMatchHi) return false; // the next column lies outside the window
              if ((colAnd >> rx) & 1u) rx++; else { right = false; if (rx < minSide) return true; }
            }
            else { right = false; if (rx < minSide) return true; }
          }
          if (down)
          {
            if (sy + ry + 1 < by)
            {
              if (ry >= hmax || ry > (uint32_t)kMatchHi) return false;
              const uint32_t a = avail(ry), need = (1u << rx) - 1u;
              if ((a & need) == need) { colAnd &= a; ry++; } else { down = false; if (ry < minSide) return true; }
            }
            else { down = false; if (ry < minSide) return true; }
          }
        }
        return true;
      }
      // src/limg.cpp:1386-1496
      bool find(bool acceptTiny, uint32_t &staticX, uint32_t &staticY, HostRegion &out) const
      {
        uint32_t ox = staticX, oy = staticY;
        for (; oy < by; oy++)
        {
          const uint8_t *urow = &used[(size_t)oy * bx];
          if (flags && needRow && oy >= rowsSeen) { (*needRow)(oy); rowsSeen = oy + 1; }
          const uint8_t *frow = flags ? flags + (size_t)oy * bx : nullptr;
          const uint8_t need = acceptTiny ? 2 : 1;
          // The similarity bits of an 8192^2 image are 50 MB, and a seed's 48-byte row is first touched when its expansion starts: a DRAM round trip per seed.  The
          // seeds to come are known (unused, flagged), so their rows are requested a few candidates ahead (`pf`: the column the look-ahead has reached in this row).
          // (the look-ahead's position survives from one call to the next -- `find` returns at every rectangle it finds)
          if (pfRow != oy || pfTiny != acceptTiny) { pfRow = oy; pfTiny = acceptTiny; pfCol = ox; pfAhead = 0; }
          auto prefetch_ahead = [&]()
          {
            if (!bits || !frow) return;
            if (needRow && oy >= rowsSeen) return; // (this band's bits are not on the host yet: nothing to request)
            for (; pfCol < bx && pfAhead < 6; pfCol++)
              if (!urow[pfCol] && (frow[pfCol] & need))
              {
                const char *q = reinterpret_cast<const char *>(bits + ((size_t)oy * bx + pfCol) * kMatchWords);
                __builtin_prefetch(q); __builtin_prefetch(q + 47);
                // large-rectangle pass: a seed whose first attempt succeeds is followed by an expansion from the centre third, (rx / 3, ry / 3) blocks further in -- another
                // similarity row, first touched when it is needed: a DRAM round trip per large rectangle (32 K of them in an 8192^2 photo-noise image: 6-7 ms of
                // the merge's 25).  Which row is not known before the first attempt is over, but the candidates are few: offsets 1 .. 4 in both directions.
                // Same box, tools/r04/run36.sh: merge 25.1-25.7 -> 18.1 ms (offsets 1 .. 2: 19.0), one image 32.9 -> 28.2 ms.
                if (!acceptTiny)
                  for (uint32_t dy = 1; dy <= kCentreAhead && oy + dy < by; dy++)
                    for (uint32_t dx = 1; dx <= kCentreAhead && pfCol + dx < bx; dx++)
                    {
                      const char *qc = reinterpret_cast<const char *>(bits + ((size_t)(oy + dy) * bx + pfCol + dx) * kMatchWords);
                      __builtin_prefetch(qc); __builtin_prefetch(qc + 47);
                    }
                pfAhead++;
              }
          };
          for (; ox < bx; ox++)
          {
#if defined(__x86_64__) && !defined(LIMG_MERGE_NO_SIMD_SCAN)
            // the next seed that is unused and flagged, sixteen at a time (both arrays carry 16 bytes of padding behind their last row; what a load takes from the
            // next row is masked off): most seeds of a pass are skipped -- in use, or flagged hopeless -- and one branch per seed was a third of the merge
            if (frow)
            {
              const __m128i needv = _mm_set1_epi8((char)need), zero = _mm_setzero_si128();
              while (ox < bx)
              {
                const __m128i u = _mm_loadu_si128(reinterpret_cast<const __m128i *>(urow + ox)), f = _mm_loadu_si128(reinterpret_cast<const __m128i *>(frow + ox));
                uint32_t cand = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(u, zero)) & ~(uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_and_si128(f, needv), zero)) & 0xFFFFu;
                if (bx - ox < 16u) cand &= (1u << (bx - ox)) - 1u;
                if (cand) { ox += (uint32_t)__builtin_ctz(cand); break; }
                ox += 16u;
              }
              if (ox >= bx) break;
            }
#endif
            if (urow[ox]) continue;
            if (frow && !(frow[ox] & need)) continue; // cannot become a rectangle of the wanted kind whatever is in use: same outcome as growing and discarding
            if (pfCol <= ox) { pfCol = ox + 1; pfAhead = 0; } else if (pfAhead > 0) pfAhead--;
            prefetch_ahead();
            uint32_t x = ox, y = oy, rx = 1, ry = 1;
            if (!expand_fast(ox, oy, rx, ry, acceptTiny ? 0u : 3u)) { rx = 1; ry = 1; expand(x, y, rx, ry, false, acceptTiny ? 0u : 3u); }
            if (rx == 1 && ry == 1) continue;
            if (!acceptTiny)
            {
              if (!(rx >= 3 && ry >= 3)) continue;
              uint32_t cx = ox + rx / 3, cy = oy + ry / 3, crx = rx / 3, cry = ry / 3; // second attempt from the centre third, all four directions
              expand(cx, cy, crx, cry, true);
              if ((uint64_t)crx * cry > (uint64_t)rx * ry)
              {
                out = { cx, cy, crx, cry, 0u };
                staticX = ox; staticY = oy;
                return true;
              }
            }
            out = { ox, oy, rx, ry, 0u };
            staticX = ox + rx; staticY = oy;
            return true;
          }
          ox = 0;
        }
        staticX = ox; staticY = oy;
        return false;
      }
      void claim(const HostRegion &r)
      {
        for (size_t y = r.oy; y < (size_t)r.oy + r.ry; y++) memset(&used[y * bx + r.ox], 1, r.rx);
      }
    };
  }

  // src/limg.cpp:1813-1881: large rectangles, then small ones, then the remaining single blocks.  `progress` (optional) is told how many rectangles
  // of `out` are final every few thousand, so that a consumer can work on them while the scan goes on; `out` never reallocates (reserved up front).
  void blocked_merge(const limg_hip_block_record *pass1, const unsigned long long *matchBits, uint32_t blocksX, uint32_t blocksY, int channels, std::vector<HostRegion> &out,
                     const std::function<void(size_t)> *progress, const std::function<void(uint32_t)> *needSeedRow, const uint8_t *seedFlags,
                     const std::function<void()> *needRecords)
  {
    Merge m;
    m.needRecords = matchBits ? needRecords : nullptr;
    if (!matchBits && needRecords) (*needRecords)(); // (no window: every pair is evaluated from the records)
    m.rec = pass1; m.bits = matchBits; m.bx = blocksX; m.by = blocksY; m.ch = channels; m.needRow = needSeedRow; m.flags = matchBits ? seedFlags : nullptr;
    m.used.assign((size_t)blocksX * blocksY + 16, 0); // (+ 16: expand_fast reads 16 flag bytes at a time)
    out.clear();
    out.reserve((size_t)blocksX * blocksY);
    size_t told = 0;
    // (the first report comes early: the consumer's pipeline -- a GPU round trip, then the serial chain walk that ends the call -- should start as soon as there is anything)
    auto tell = [&](bool force) { if (progress && (force || out.size() - told >= (told == 0 ? 1024u : 4096u))) { told = out.size(); (*progress)(told); } };
    for (int tiny = 0; tiny < 2; tiny++)
    {
      uint32_t sx = 0, sy = 0;
      HostRegion r;
      while (m.find(tiny != 0, sx, sy, r)) { m.claim(r); out.push_back(r); tell(false); }
    }
    for (uint32_t y = 0; y < blocksY; y++)
      for (uint32_t x = 0; x < blocksX; x++)
        if (!m.used[(size_t)y * blocksX + x]) { out.push_back({ x, y, 1u, 1u, 1u }); tell(false); }
    tell(true);
  }

  // One dither call over n pixels (src/limg.cpp:824-879 / :799-822), any n: floor(n / 8) AES rounds on {h, ~h}, then n % 8 PCG steps on the
  // low 64 bits; writes n noise bytes (the byte pixel i ANDs with its dither mask) and returns the next chain value.
  uint64_t chain_call_n(uint64_t h, size_t n, uint8_t *noise, bool pcg) { return chain_call(h, (unsigned)n, noise, false, pcg); }
}

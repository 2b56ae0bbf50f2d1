/* TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the limg encode hot path.  See limg_oracle.h.
 *
 * Written from the behaviour of the reference's SSE4.1/AES-NI functions (the normative path), in plain
 * scalar C with the operation order of the x86 instructions spelled out.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference/).  Compile WITHOUT fast-math and without FP contraction
 * (oracle/Makefile: -O2 -ffp-contract=off): the float stage relies on IEEE single-precision op-by-op rounding.
 *
 * Not the product: nothing under limg_amd/ may include, link or call this file.
 */
#include "limg_oracle.h"
#include "limg_rsqrt_x86_table.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define LIMG_BLOCK 8 /* limg_MinBlockSize, src/limg_internal.h:158 */

/* ------------------------------------------------------------------------------------------------------------------ */
/* x86 instruction semantics used by the float stage                                               */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* RSQRTPS as implemented by the golden container's Intel CPU (table captured by tools/make_rsqrt_table.py). */
float limg_oracle_rsqrt_x86(float x)
{
  if (x != x) return x;
  const uint32_t b = f2u(x);
  const int e = (int)((b >> 23) & 0xFF);
  const uint32_t idx = ((e & 1) ? 0u : 1024u) + ((b >> 13) & 0x3FFu);
  const int k = (e - 127) >> 1; /* arithmetic shift == floor */
  return u2f(((uint32_t)(126 - k) << 23) | ((uint32_t)limg_rsqrt_x86_tab[idx] << 11));
}

/* MINPS / MAXPS: second operand when unordered */
static inline float minps(float a, float b) { return a < b ? a : b; }
static inline float maxps(float a, float b) { return a > b ? a : b; }

/* CVTPS2DQ (round-to-nearest-even, "integer indefinite" on NaN / overflow) */
static inline int32_t cvtps(float x)
{
  if (!(x >= -2147483648.0f && x < 2147483648.0f)) return INT32_MIN;
  return (int32_t)lrintf(x);
}

/* DPPS with mask 0xFF (4 ch) / 0x7F (3 ch): (x0y0 + x1y1) + (x2y2 + x3y3), products rounded individually */
static inline float dpps(const float a[4], const float b[4], int channels)
{
  const float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2];
  const float p3 = channels == 4 ? a[3] * b[3] : 0.0f;
  return (p0 + p1) + (p2 + p3);
}

static inline void px_to_float(uint32_t px, float out[4])
{
  out[0] = (float)(px & 0xFF); out[1] = (float)((px >> 8) & 0xFF); out[2] = (float)((px >> 16) & 0xFF); out[3] = (float)(px >> 24);
}

/* The "sign-normalised unit vector" step shared by all direction passes
 * (src/limg_factorization.h:411-428 / 605-623 and the two analogous blocks per function).
 * Returns 0 when every lane of d is zero (pixel skipped). */
static int unit_contribution(const float d[4], int channels, float out[4])
{
  if (d[0] == 0.0f && d[1] == 0.0f && d[2] == 0.0f && d[3] == 0.0f)
    return 0;
  /* preferenceBias = _mm_set_ps(0, 1e, 2e, 3e): lane0 = 3*eps ... lane3 = 0 (:393 / :589) */
  static const float bias[4] = { FLT_EPSILON * 3, FLT_EPSILON * 2, FLT_EPSILON * 1, 0.0f };
  float mb[4], xb[4];
  for (int i = 0; i < 4; i++) { mb[i] = d[i] - bias[i]; xb[i] = d[i] + bias[i]; }
  const float half_min0 = minps(mb[0], mb[2]), half_min1 = minps(mb[1], mb[3]);
  const float half_max0 = maxps(xb[0], xb[2]), half_max1 = maxps(xb[1], xb[3]);
  const float abs_min = fabsf(minps(half_min0, half_min1));
  const float mx = maxps(half_max0, half_max1);
  float inv = limg_oracle_rsqrt_x86(dpps(d, d, channels));
  if (abs_min > mx) inv = u2f(f2u(inv) ^ 0x80000000u);
  for (int i = 0; i < 4; i++) out[i] = d[i] * inv;
  return 1;
}

/* Sum of the per-pixel contributions of one direction pass.
 * X86: pixel order, skipped pixels really skipped (src/limg_factorization.h:402-431).
 * TREE: balanced binary tree over the pixel index, index bit 0 combined first, skipped / missing pixels are +0.0f. */
static void sum_contributions(float (*v)[4], const int *valid, size_t n, int float_mode, float out[4])
{
  if (float_mode == LIMG_ORACLE_FLOAT_X86 || n > 64) /* the tree what-if is defined for 8x8 blocks only */
  {
    for (int c = 0; c < 4; c++)
    {
      float s = 0.0f;
      for (size_t i = 0; i < n; i++)
        if (valid[i]) s = s + v[i][c];
      out[c] = s;
    }
  }
  else
  {
    float t[64];
    for (int c = 0; c < 4; c++)
    {
      for (size_t i = 0; i < 64; i++) t[i] = (i < n && valid[i]) ? v[i][c] : 0.0f;
      for (size_t stride = 1; stride < 64; stride <<= 1)
        for (size_t i = 0; i < 64; i += 2 * stride) t[i] = t[i] + t[i + stride];
      out[c] = t[0];
    }
  }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a4-a6: channel sums (src/limg.cpp:466-497) + direction fit and extrema
 * (src/limg_factorization.h:578-794 for 4 channels, :382-576 for 3 channels) */
void limg_oracle_block_fit(const uint32_t *px, size_t n, int channels, int float_mode, limg_oracle_record *out)
{
  limg_oracle_block_fit_gathered(px, n, n, channels, float_mode, out);
}

/* `sum_n` pixels enter the channel sums, `n` everything else.  Upstream's SIMD sum loop (src/limg.cpp:478-487) always consumes at least 4 pixels:
 * for a block of fewer than 4 it also adds px[n..3], i.e. whatever the previous block left in the gather buffer (src/limg.cpp:1890,1899-1905: the buffer
 * lives outside the block loops).  The drivers below keep such a persistent buffer and pass sum_n = max(n, 4), which reproduces upstream bit for bit
 * wherever a previous block exists; the very first block of a strip would read uninitialised stack upstream (zeroes here). */
void limg_oracle_block_fit_gathered(const uint32_t *px, size_t n, size_t sum_n, int channels, int float_mode, limg_oracle_record *out)
{
  uint32_t sum[4] = { 0, 0, 0, 0 };
  for (size_t i = 0; i < sum_n; i++)
    for (int c = 0; c < 4; c++) sum[c] += (px[i] >> (8 * c)) & 0xFF;

  const float inv_count = 1.0f / (float)n;
  float avg[4];
  for (int c = 0; c < 4; c++) avg[c] = (float)(int32_t)sum[c] * inv_count;
  if (channels == 3) avg[3] = 0.0f; /* lane 3 is stack garbage upstream and never observable (:396-402) */

  /* any pixel count: the merged-block encoder fits whole regions with the same function (src/limg.cpp:1752-1760) */
  const size_t cap = n > 64 ? n : 64;
  float (*contrib)[4] = (float (*)[4])malloc(cap * sizeof(float[4]));
  float (*pxf)[4] = (float (*)[4])malloc(cap * sizeof(float[4]));
  float (*est)[4] = (float (*)[4])malloc(cap * sizeof(float[4]));
  int *valid = (int *)malloc(cap * sizeof(int));
  float dirA[4] = { 0, 0, 0, 0 }, dirB[4] = { 0, 0, 0, 0 }, dirC[4] = { 0, 0, 0, 0 };
  float minA = 0, maxA = 0, minB = 0, maxB = 0, minC = 0, maxC = 0;

  for (size_t i = 0; i < n; i++) px_to_float(px[i], pxf[i]);

  /* pass 1 (:602-628 / :402-431) */
  for (size_t i = 0; i < n; i++)
  {
    float d[4];
    for (int c = 0; c < 4; c++) d[c] = pxf[i][c] - avg[c];
    if (channels == 3) d[3] = 0.0f; /* zero_alpha */
    valid[i] = unit_contribution(d, channels, contrib[i]);
  }
  sum_contributions(contrib, valid, n, float_mode, dirA);
  for (int c = 0; c < 4; c++) dirA[c] = dirA[c] * inv_count;

  if (!(dirA[0] == 0.0f && dirA[1] == 0.0f && dirA[2] == 0.0f && dirA[3] == 0.0f))
  {
    const float invA = 1.0f / dpps(dirA, dirA, channels);

    /* pass 2 (:652-688 / :451-491) */
    for (size_t i = 0; i < n; i++)
    {
      float l[4], e[4];
      for (int c = 0; c < 4; c++) l[c] = pxf[i][c] - avg[c];
      const float fA = dpps(l, dirA, channels) * invA;
      minA = minps(minA, fA);
      maxA = maxps(maxA, fA);
      for (int c = 0; c < 4; c++) { est[i][c] = avg[c] + fA * dirA[c]; e[c] = pxf[i][c] - est[i][c]; }
      if (channels == 3) e[3] = 0.0f;
      valid[i] = unit_contribution(e, channels, contrib[i]);
    }
    sum_contributions(contrib, valid, n, float_mode, dirB);
    for (int c = 0; c < 4; c++) dirB[c] = dirB[c] * inv_count;

    minB = minC = FLT_MAX;
    maxB = maxC = -FLT_MAX;

    if (channels == 4)
    {
      const float invB = 1.0f / dpps(dirB, dirB, 4);

      /* pass 3 (:701-738) */
      for (size_t i = 0; i < n; i++)
      {
        float l[4], e[4];
        for (int c = 0; c < 4; c++) l[c] = pxf[i][c] - est[i][c];
        const float fB = dpps(l, dirB, 4) * invB;
        minB = minps(minB, fB);
        maxB = maxps(maxB, fB);
        for (int c = 0; c < 4; c++) { est[i][c] = est[i][c] + fB * dirB[c]; e[c] = pxf[i][c] - est[i][c]; }
        valid[i] = unit_contribution(e, 4, contrib[i]);
      }
      sum_contributions(contrib, valid, n, float_mode, dirC);
      for (int c = 0; c < 4; c++) dirC[c] = dirC[c] * inv_count;

      const float invC = 1.0f / dpps(dirC, dirC, 4);

      /* pass 4 (:748-758).  NOTE: upstream never advances `pEstimate` in this loop (:752 has no `pEstimate++`), so
       * every pixel is measured against the A+B estimate of pixel 0.  Restated literally (bug-compatible). */
      for (size_t i = 0; i < n; i++)
      {
        float l[4];
        for (int c = 0; c < 4; c++) l[c] = pxf[i][c] - est[0][c];
        const float fC = dpps(l, dirC, 4) * invC;
        minC = minps(minC, fC);
        maxC = maxps(maxC, fC);
      }
    }
    else
    {
      /* dirC = dirA x dirB (:498-507); lane 3 = A3*B3 - A3*B3 */
      dirC[0] = dirA[1] * dirB[2] - dirA[2] * dirB[1];
      dirC[1] = dirA[2] * dirB[0] - dirA[0] * dirB[2];
      dirC[2] = dirA[0] * dirB[1] - dirA[1] * dirB[0];
      dirC[3] = dirA[3] * dirB[3] - dirA[3] * dirB[3];

      const float invB = 1.0f / dpps(dirB, dirB, 3);
      const float invC = 1.0f / dpps(dirC, dirC, 3);

      /* pass 3 (:517-541): B and C extrema together */
      for (size_t i = 0; i < n; i++)
      {
        float l[4], e[4];
        for (int c = 0; c < 4; c++) l[c] = pxf[i][c] - est[i][c];
        const float fB = dpps(l, dirB, 3) * invB;
        minB = minps(minB, fB);
        maxB = maxps(maxB, fB);
        for (int c = 0; c < 4; c++) e[c] = pxf[i][c] - (est[i][c] + fB * dirB[c]);
        const float fC = dpps(e, dirC, 3) * invC;
        minC = minps(minC, fC);
        maxC = maxps(maxC, fC);
      }
    }
  }

  /* :764-790 / :545-575 -- cvtps2dq then truncation to int16 */
  memset(out, 0, sizeof(*out));
  for (int c = 0; c < channels; c++)
  {
    out->avg[c] = avg[c];
    out->dirA_min[c] = (int16_t)cvtps(avg[c] + minA * dirA[c]);
    out->dirA_max[c] = (int16_t)cvtps(avg[c] + maxA * dirA[c]);
    out->dirB_offset[c] = (int16_t)cvtps(minB * dirB[c]);
    out->dirB_mag[c] = (int16_t)cvtps(maxB * dirB[c]);
    out->dirC_offset[c] = (int16_t)cvtps(minC * dirC[c]);
    out->dirC_mag[c] = (int16_t)cvtps(maxC * dirC[c]);
  }
  free(contrib); free(pxf); free(est); free(valid);
}

/* a7: src/limg_internal.h:426-452;  a8: src/limg_factorization.h:149-197 (4 ch) / :98-147 (3 ch) */
void limg_oracle_block_factors(const uint32_t *px, size_t n, int channels, const limg_oracle_record *rec, uint8_t *A, uint8_t *B, uint8_t *C)
{
  float nA[4] = { 0, 0, 0, 0 }, nB[4] = { 0, 0, 0, 0 }, nC[4] = { 0, 0, 0, 0 };
  float mnA[4] = { 0, 0, 0, 0 }, ofB[4] = { 0, 0, 0, 0 }, ofC[4] = { 0, 0, 0, 0 };
  int nzA = 0, nzB = 0, nzC = 0;
  for (int c = 0; c < channels; c++)
  {
    nA[c] = (float)((int)rec->dirA_max[c] - (int)rec->dirA_min[c]);
    nB[c] = (float)((int)rec->dirB_mag[c] - (int)rec->dirB_offset[c]);
    nC[c] = (float)((int)rec->dirC_mag[c] - (int)rec->dirC_offset[c]);
    nzA |= nA[c] != 0; nzB |= nB[c] != 0; nzC |= nC[c] != 0;
    mnA[c] = (float)rec->dirA_min[c]; ofB[c] = (float)rec->dirB_offset[c]; ofC[c] = (float)rec->dirC_offset[c];
  }
  /* limg_dot: serial ((0 + n0*n0) + n1*n1) + ... (src/limg_internal.h:357-366) */
  float invA = 0, invB = 0, invC = 0;
  if (nzA) { float s = 0; for (int c = 0; c < channels; c++) s += nA[c] * nA[c]; invA = 1.0f / s; }
  if (nzB) { float s = 0; for (int c = 0; c < channels; c++) s += nB[c] * nB[c]; invB = 1.0f / s; }
  if (nzC) { float s = 0; for (int c = 0; c < channels; c++) s += nC[c] * nC[c]; invC = 1.0f / s; }

  for (size_t i = 0; i < n; i++)
  {
    float col[4], t[4], est[4];
    px_to_float(px[i], col);
    for (int c = 0; c < 4; c++) t[c] = col[c] - mnA[c];
    const float fa = dpps(t, nA, channels) * invA;
    int32_t q = cvtps(255.0f * fa); q = q < 0xFF ? q : 0xFF; q = q > 0 ? q : 0;
    A[i] = (uint8_t)q;
    for (int c = 0; c < 4; c++) { est[c] = mnA[c] + nA[c] * fa; t[c] = (col[c] - est[c]) - ofB[c]; }
    const float fb = dpps(t, nB, channels) * invB;
    q = cvtps(255.0f * fb); q = q < 0xFF ? q : 0xFF; q = q > 0 ? q : 0;
    B[i] = (uint8_t)q;
    for (int c = 0; c < 4; c++) { est[c] = est[c] + nB[c] * fb; t[c] = (col[c] - est[c]) - ofC[c]; }
    const float fc = dpps(t, nC, channels) * invC;
    q = cvtps(255.0f * fc); q = q < 0xFF ? q : 0xFF; q = q > 0 ? q : 0;
    C[i] = (uint8_t)q;
  }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* integer stage                                                                                                      */

/* (1 << s) + decode_bias(s), src/limg_bit_crush_simd.h:611-619 (s = 0 and s = 8 are UB upstream; x86 result: bias 0) */
static const uint32_t k_shift_mul[9] = { 1, 2, 4, 8, 17, 36, 85, 255, 256 };

typedef struct
{
  uint32_t nA[4], nB[4], nC[4]; /* as uint32 for PMULLD wrap-around */
  uint32_t mA[4], mB[4], mC[4]; /* (min << 8) + 128 */
  uint32_t mul[3];
} recon_consts;

/* src/limg_bit_crush_simd.h:568-625 (trial) and src/limg_decode.h:139-196 / :40-101 (decode) share this setup;
 * `decode3` selects the 3-channel decoder's alpha lane (minX lane 3 = 0xFFFF, src/limg_decode.h:95-97). */
static void recon_setup(const limg_oracle_record *rec, int channels, const uint8_t shift[3], int decode3, recon_consts *k)
{
  int32_t nA[4] = { 0, 0, 0, 0 }, nB[4] = { 0, 0, 0, 0 }, nC[4] = { 0, 0, 0, 0 }, mA[4] = { 0, 0, 0, 0 }, mB[4] = { 0, 0, 0, 0 }, mC[4] = { 0, 0, 0, 0 };
  for (int c = 0; c < channels; c++)
  {
    nA[c] = rec->dirA_max[c] - rec->dirA_min[c];
    nB[c] = rec->dirB_mag[c] - rec->dirB_offset[c];
    nC[c] = rec->dirC_mag[c] - rec->dirC_offset[c];
    mA[c] = rec->dirA_min[c]; mB[c] = rec->dirB_offset[c]; mC[c] = rec->dirC_offset[c];
  }
  if (shift[0] > 7) for (int c = 0; c < 3; c++) nA[c] = 0;
  if (shift[1] > 7) for (int c = 0; c < 3; c++) { nB[c] = 0; mB[c] = 0; }
  if (shift[2] > 7) for (int c = 0; c < 3; c++) { nC[c] = 0; mC[c] = 0; }
  if (decode3) { mA[3] = mB[3] = mC[3] = 0xFFFF; }
  for (int c = 0; c < 4; c++)
  {
    k->nA[c] = (uint32_t)nA[c]; k->nB[c] = (uint32_t)nB[c]; k->nC[c] = (uint32_t)nC[c];
    k->mA[c] = ((uint32_t)mA[c] << 8) + 128u; k->mB[c] = ((uint32_t)mB[c] << 8) + 128u; k->mC[c] = ((uint32_t)mC[c] << 8) + 128u;
  }
  for (int i = 0; i < 3; i++) k->mul[i] = k_shift_mul[shift[i]];
}

static inline int32_t sra8(uint32_t v) { return (int32_t)v >> 8; } /* PSRAD: arithmetic (gcc/clang: >> on negative int is arithmetic) */

static inline int32_t recon_channel(const recon_consts *k, int c, uint32_t dA, uint32_t dB, uint32_t dC)
{
  int32_t e = sra8(dA * k->nA[c] + k->mA[c]);
  e = (int32_t)((uint32_t)e + (uint32_t)sra8(dB * k->nB[c] + k->mB[c]));
  e = (int32_t)((uint32_t)e + (uint32_t)sra8(dC * k->nC[c] + k->mC[c]));
  e = e > 0 ? e : 0;
  return e < 0xFF ? e : 0xFF;
}

/* a9: src/limg_bit_crush_simd.h:562-810 (4 ch) / :311-560 (3 ch).  NOTE: lane 0 of `error_1234_` is e0 + e2 + e1
 * (:753-754 adds `error_ >> 1 lane`, not `error_13_24_ >> 1 lane`), i.e. the alpha term never reaches the
 * per-pixel or per-block error in either variant; restated literally. */
static int trial_core(const uint32_t *px, size_t n, int channels, const limg_oracle_record *rec, const uint8_t *A, const uint8_t *B, const uint8_t *C,
                      const uint8_t shift[3], size_t maxPixel, size_t maxBlock, size_t *pBlockError)
{
  recon_consts k;
  recon_setup(rec, channels, shift, 0, &k);
  uint32_t block_error = 0;
  for (size_t i = 0; i < n; i++)
  {
    const uint32_t dA = (uint32_t)(A[i] >> shift[0]) * k.mul[0];
    const uint32_t dB = (uint32_t)(B[i] >> shift[1]) * k.mul[1];
    const uint32_t dC = (uint32_t)(C[i] >> shift[2]) * k.mul[2];
    uint32_t dsq[3];
    for (int c = 0; c < 3; c++)
    {
      const int32_t d = (int32_t)((px[i] >> (8 * c)) & 0xFF) - recon_channel(&k, c, dA, dB, dC);
      dsq[c] = (uint32_t)(d * d);
    }
    const int low_red = (int32_t)dsq[0] < 0x4000;
    const uint32_t err = dsq[0] * (low_red ? 2u : 3u) + dsq[2] * (low_red ? 3u : 2u) + dsq[1] * 4u;
    block_error += err;
    if ((size_t)(int32_t)err > maxPixel) return 0;
  }
  const size_t be = (size_t)(int32_t)block_error;
  *pBlockError = be;
  return (be * 0x10) < maxBlock * n;
}

static void thresholds(uint32_t ef, size_t *maxPixel, size_t *maxBlock)
{
  /* src/limg.cpp:2190-2191, :2207-2212 */
  *maxPixel = (size_t)0x6 * (ef / 2) * 7;
  *maxBlock = (size_t)0x4 * (ef / 2) * 7;
}

/* statistics helper (tools/trial_early_exit.py): the per-pixel error of one trial for EVERY pixel, i.e. trial_core without its early return */
void limg_oracle_block_trial_pixel_errors(const uint32_t *px, size_t n, int channels, const limg_oracle_record *rec, const uint8_t *A, const uint8_t *B, const uint8_t *C,
                                          const uint8_t shift[3], uint32_t *pErr)
{
  recon_consts k;
  recon_setup(rec, channels, shift, 0, &k);
  for (size_t i = 0; i < n; i++)
  {
    const uint32_t dA = (uint32_t)(A[i] >> shift[0]) * k.mul[0];
    const uint32_t dB = (uint32_t)(B[i] >> shift[1]) * k.mul[1];
    const uint32_t dC = (uint32_t)(C[i] >> shift[2]) * k.mul[2];
    uint32_t dsq[3];
    for (int c = 0; c < 3; c++)
    {
      const int32_t d = (int32_t)((px[i] >> (8 * c)) & 0xFF) - recon_channel(&k, c, dA, dB, dC);
      dsq[c] = (uint32_t)(d * d);
    }
    const int low_red = (int32_t)dsq[0] < 0x4000;
    pErr[i] = dsq[0] * (low_red ? 2u : 3u) + dsq[2] * (low_red ? 3u : 2u) + dsq[1] * 4u;
  }
}

int limg_oracle_block_trial(const uint32_t *px, size_t n, int channels, const limg_oracle_record *rec, const uint8_t *A, const uint8_t *B, const uint8_t *C,
                            const uint8_t shift[3], uint32_t error_factor, uint64_t *pBlockError)
{
  size_t mp, mb, be = 0;
  thresholds(error_factor, &mp, &mb);
  const int ok = trial_core(px, n, channels, rec, A, B, C, shift, mp, mb, &be);
  if (pBlockError) *pBlockError = be;
  return ok;
}

typedef struct
{
  const uint32_t *px; size_t n; int channels; const limg_oracle_record *rec; const uint8_t *A, *B, *C; size_t maxPixel, maxBlock; uint32_t trials;
} search_ctx;

static int try_shift(search_ctx *s, uint8_t a, uint8_t b, uint8_t c, size_t *pBlockError)
{
  const uint8_t sh[3] = { a, b, c };
  s->trials++;
  return trial_core(s->px, s->n, s->channels, s->rec, s->A, s->B, s->C, sh, s->maxPixel, s->maxBlock, pBlockError);
}

/* a10: src/limg_bit_crush.h:331-392 */
static void guess_shift(search_ctx *s, uint8_t shift[3], size_t *pMinBlockError)
{
  size_t be, min_be = (size_t)-1;
  if (try_shift(s, 4, 5, 6, &be))
  {
    shift[0] = 4; shift[1] = 5; shift[2] = 6; min_be = be;
    if (try_shift(s, 5, 8, 8, &be)) { shift[0] = 5; shift[1] = 8; shift[2] = 8; min_be = be; }
    else if (try_shift(s, 4, 6, 8, &be)) { shift[0] = 4; shift[1] = 6; shift[2] = 8; min_be = be; }
  }
  else if (try_shift(s, 2, 4, 5, &be)) { shift[0] = 2; shift[1] = 4; shift[2] = 5; min_be = be; }
  *pMinBlockError = min_be;
}

/* the "(Potentially) check other max shifts" tail shared by :617-665 and :780-829 */
static void equal_sum_pass(search_ctx *s, uint8_t shift[3], size_t max_shift, size_t min_block_error)
{
  size_t be;
  uint8_t a = shift[0], b = shift[1], c = (uint8_t)(shift[2] + 1);
  for (; a <= 8; a++)
  {
    for (; b <= 8; b++)
    {
      for (; c <= 8; c++)
      {
        if ((size_t)(a + b + c) == max_shift)
        {
          if (try_shift(s, a, b, c, &be))
          {
            if (min_block_error > be) { shift[0] = a; shift[1] = b; shift[2] = c; min_block_error = be; }
          }
          else
            break;
        }
      }
      if (c == 0) break;
      c = 0;
    }
    if (b == 0) break;
    b = 0;
  }
}

/* a11: src/limg_bit_crush.h:502-666 (loop-carried resets restated literally, all counters uint8_t) */
static void stepwise_shift(search_ctx *s, uint8_t shift[3], size_t minBlockError, int fastBitCrush)
{
  uint8_t max_shift = (uint8_t)(shift[0] + shift[1] + shift[2]);
  size_t min_block_error = minBlockError, be;

  { /* coarse :510-556 */
    uint8_t a = shift[0] & 15, b = shift[1] & 15, c = (uint8_t)((shift[2] & 15) + 2);
    for (; a <= 8; a += 2)
    {
      for (; b <= 8; b += 2)
      {
        for (; c <= 8; c += 2)
        {
          if (a + b + c > max_shift)
          {
            if (try_shift(s, a, b, c, &be)) { shift[0] = a; shift[1] = b; shift[2] = c; max_shift = (uint8_t)(a + b + c); min_block_error = be; }
            else
              break;
          }
        }
        if (c == b) break;
        c = b;
      }
      if (b == a) break;
      b = a;
    }
  }

  { /* fine :558-614 */
    const uint8_t pre_a = shift[0], pre_b = shift[1], pre_c = shift[2];
    const size_t max_a = !(pre_a & 1) && pre_a != 8, max_b = !(pre_b & 1) && pre_b != 8, max_c = !(pre_c & 1) && pre_c != 8;
    uint8_t fine_shift = 0, a = 0, b = 0, c = 1;
    for (; a <= max_a; a++)
    {
      for (; b <= max_b; b++)
      {
        for (; c <= max_c; c++)
        {
          if (a + b + c > fine_shift)
          {
            if (try_shift(s, (uint8_t)(pre_a + a), (uint8_t)(pre_b + b), (uint8_t)(pre_c + c), &be))
            {
              shift[0] = (uint8_t)(pre_a + a); shift[1] = (uint8_t)(pre_b + b); shift[2] = (uint8_t)(pre_c + c);
              max_shift = (uint8_t)(shift[0] + shift[1] + shift[2]);
              fine_shift = (uint8_t)(a + b + c);
              min_block_error = be;
            }
            else
              break;
          }
        }
        if (c == 0) break;
        c = 0;
      }
      if (b == 0) break;
      b = 0;
    }
  }

  if (max_shift > 0 && !fastBitCrush) /* :617 -- unreachable with the flag wiring of src/limg.cpp:2196-2197, kept for completeness */
    equal_sum_pass(s, shift, max_shift, min_block_error);
}

/* a12: src/limg_bit_crush.h:668-830 (`--accurate-bit-crushing`); extractPixel only rotates the scan start (no effect on the result) */
static void pixel_preference_shift(search_ctx *s, uint8_t shift[3], int fastBitCrush)
{
  size_t max_shift = 0, min_block_error = (size_t)-1, be;
  if (try_shift(s, 4, 5, 6, &be))
  {
    shift[0] = 4; shift[1] = 5; shift[2] = 6; max_shift = 15; min_block_error = be;
    if (try_shift(s, 5, 8, 8, &be)) { shift[0] = 5; shift[1] = 8; shift[2] = 8; max_shift = 21; min_block_error = be; }
    else if (try_shift(s, 4, 6, 8, &be)) { shift[0] = 4; shift[1] = 6; shift[2] = 8; max_shift = 18; min_block_error = be; }
  }
  else if (try_shift(s, 2, 4, 5, &be)) { shift[0] = 2; shift[1] = 4; shift[2] = 5; max_shift = 11; min_block_error = be; }

  {
    uint8_t a = 0, b = 0, c = 1;
    for (; a <= 8; a++)
    {
      for (; b <= 8; b++)
      {
        for (; c <= 8; c++)
        {
          if ((size_t)(a + b + c) > max_shift && (a != shift[0] || b != shift[1] || c != shift[2]))
          {
            if (try_shift(s, a, b, c, &be)) { shift[0] = a; shift[1] = b; shift[2] = c; max_shift = (size_t)(a + b + c); min_block_error = be; }
            else
              break;
          }
        }
        if (c == 0) break;
        c = 0;
      }
      if (b == 0) break;
      b = 0;
    }
  }

  if (max_shift > 0 && !fastBitCrush)
    equal_sum_pass(s, shift, max_shift, min_block_error);
}

/* dispatch of src/limg.cpp:1922-1945 with the flag wiring of :2192-2197 */
uint32_t limg_oracle_block_search(const uint32_t *px, size_t n, int channels, const limg_oracle_record *rec, const uint8_t *A, const uint8_t *B, const uint8_t *C,
                                  uint32_t error_factor, int fast, uint8_t shift[3])
{
  search_ctx s = { px, n, channels, rec, A, B, C, 0, 0, 0 };
  thresholds(error_factor, &s.maxPixel, &s.maxBlock);
  shift[0] = shift[1] = shift[2] = 0;
  if (error_factor == 0) return 0; /* crushBits = errorFactor != 0 */
  if (!fast)
    pixel_preference_shift(&s, shift, 0);
  else
  {
    size_t min_be = (size_t)-1;
    guess_shift(&s, shift, &min_be);
    stepwise_shift(&s, shift, min_be, 1);
  }
  return s.trials;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* dither                                                                                                             */

static const uint8_t k_inv_sbox[256] = {
  0x52, 0x09, 0x6a, 0xd5, 0x30, 0x36, 0xa5, 0x38, 0xbf, 0x40, 0xa3, 0x9e, 0x81, 0xf3, 0xd7, 0xfb, 0x7c, 0xe3, 0x39, 0x82, 0x9b, 0x2f, 0xff, 0x87, 0x34, 0x8e, 0x43, 0x44, 0xc4, 0xde, 0xe9, 0xcb,
  0x54, 0x7b, 0x94, 0x32, 0xa6, 0xc2, 0x23, 0x3d, 0xee, 0x4c, 0x95, 0x0b, 0x42, 0xfa, 0xc3, 0x4e, 0x08, 0x2e, 0xa1, 0x66, 0x28, 0xd9, 0x24, 0xb2, 0x76, 0x5b, 0xa2, 0x49, 0x6d, 0x8b, 0xd1, 0x25,
  0x72, 0xf8, 0xf6, 0x64, 0x86, 0x68, 0x98, 0x16, 0xd4, 0xa4, 0x5c, 0xcc, 0x5d, 0x65, 0xb6, 0x92, 0x6c, 0x70, 0x48, 0x50, 0xfd, 0xed, 0xb9, 0xda, 0x5e, 0x15, 0x46, 0x57, 0xa7, 0x8d, 0x9d, 0x84,
  0x90, 0xd8, 0xab, 0x00, 0x8c, 0xbc, 0xd3, 0x0a, 0xf7, 0xe4, 0x58, 0x05, 0xb8, 0xb3, 0x45, 0x06, 0xd0, 0x2c, 0x1e, 0x8f, 0xca, 0x3f, 0x0f, 0x02, 0xc1, 0xaf, 0xbd, 0x03, 0x01, 0x13, 0x8a, 0x6b,
  0x3a, 0x91, 0x11, 0x41, 0x4f, 0x67, 0xdc, 0xea, 0x97, 0xf2, 0xcf, 0xce, 0xf0, 0xb4, 0xe6, 0x73, 0x96, 0xac, 0x74, 0x22, 0xe7, 0xad, 0x35, 0x85, 0xe2, 0xf9, 0x37, 0xe8, 0x1c, 0x75, 0xdf, 0x6e,
  0x47, 0xf1, 0x1a, 0x71, 0x1d, 0x29, 0xc5, 0x89, 0x6f, 0xb7, 0x62, 0x0e, 0xaa, 0x18, 0xbe, 0x1b, 0xfc, 0x56, 0x3e, 0x4b, 0xc6, 0xd2, 0x79, 0x20, 0x9a, 0xdb, 0xc0, 0xfe, 0x78, 0xcd, 0x5a, 0xf4,
  0x1f, 0xdd, 0xa8, 0x33, 0x88, 0x07, 0xc7, 0x31, 0xb1, 0x12, 0x10, 0x59, 0x27, 0x80, 0xec, 0x5f, 0x60, 0x51, 0x7f, 0xa9, 0x19, 0xb5, 0x4a, 0x0d, 0x2d, 0xe5, 0x7a, 0x9f, 0x93, 0xc9, 0x9c, 0xef,
  0xa0, 0xe0, 0x3b, 0x4d, 0xae, 0x2a, 0xf5, 0xb0, 0xc8, 0xeb, 0xbb, 0x3c, 0x83, 0x53, 0x99, 0x61, 0x17, 0x2b, 0x04, 0x7e, 0xba, 0x77, 0xd6, 0x26, 0xe1, 0x69, 0x14, 0x63, 0x55, 0x21, 0x0c, 0x7d
};

static inline uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1B)); }
static inline uint8_t gmul(uint8_t x, int m)
{
  const uint8_t x2 = xtime(x), x4 = xtime(x2), x8 = xtime(x4);
  switch (m)
  {
  case 9: return (uint8_t)(x8 ^ x);
  case 11: return (uint8_t)(x8 ^ x2 ^ x);
  case 13: return (uint8_t)(x8 ^ x4 ^ x);
  default: return (uint8_t)(x8 ^ x4 ^ x2); /* 14 */
  }
}

/* AESDEC xmm, key (FIPS-197 equivalent-inverse-cipher round): InvShiftRows, InvSubBytes, InvMixColumns, xor key.
 * Round key of src/limg.cpp:837: high qword 0x2A76E98006CB4CAD, low qword 0x824A73EAAB705E1D. */
static void aesdec_round(uint8_t st[16])
{
  static const uint8_t key[16] = { 0x1D, 0x5E, 0x70, 0xAB, 0xEA, 0x73, 0x4A, 0x82, 0xAD, 0x4C, 0xCB, 0x06, 0x80, 0xE9, 0x76, 0x2A };
  uint8_t t[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++)
      t[r + 4 * c] = k_inv_sbox[st[r + 4 * ((c - r) & 3)]];
  for (int c = 0; c < 4; c++)
  {
    const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
    st[4 * c + 0] = (uint8_t)(gmul(a0, 14) ^ gmul(a1, 11) ^ gmul(a2, 13) ^ gmul(a3, 9) ^ key[4 * c + 0]);
    st[4 * c + 1] = (uint8_t)(gmul(a0, 9) ^ gmul(a1, 14) ^ gmul(a2, 11) ^ gmul(a3, 13) ^ key[4 * c + 1]);
    st[4 * c + 2] = (uint8_t)(gmul(a0, 13) ^ gmul(a1, 9) ^ gmul(a2, 14) ^ gmul(a3, 11) ^ key[4 * c + 2]);
    st[4 * c + 3] = (uint8_t)(gmul(a0, 11) ^ gmul(a1, 13) ^ gmul(a2, 9) ^ gmul(a3, 14) ^ key[4 * c + 3]);
  }
}

static inline uint32_t pcg_step(uint64_t *h)
{
  /* src/limg.cpp:809-814 */
  *h = *h * 6364136223846793005ULL + 1;
  const uint32_t xorshifted_hi = (uint32_t)(((*h >> 18) ^ *h) >> 27);
  const uint32_t rot_hi = (uint32_t)(*h >> 59);
  return (xorshifted_hi >> rot_hi) | (xorshifted_hi << ((uint32_t)(-(int32_t)rot_hi) & 31));
}

static inline uint8_t dither_px(uint8_t f, uint32_t rnd, int shift)
{
  const int32_t size = (1 << shift) - 1, offset = 1 << (shift - 1);
  int32_t v = (int32_t)f + ((int32_t)(rnd & (uint32_t)size) - offset);
  v = v < 0 ? 0 : (v > 0xFF ? 0xFF : v);
  return (uint8_t)((uint8_t)v >> shift);
}

/* a13: src/limg.cpp:824-879;  a14: src/limg.cpp:799-822.  `f == NULL` walks the state only (limg_oracle_chain_step). */
static uint64_t dither_impl(int shift, size_t n, uint64_t hash, uint8_t *f, int dither_mode)
{
  if (shift > 7) return hash;
  size_t i = 0;
  if (dither_mode == LIMG_ORACLE_DITHER_AES)
  {
    if (n >= 8)
    {
      uint8_t st[16];
      const uint64_t inv = ~hash;
      memcpy(st, &hash, 8);
      memcpy(st + 8, &inv, 8);
      for (; i + 8 <= n; i += 8)
      {
        aesdec_round(st);
        if (f)
          for (int j = 0; j < 8; j++)
          {
            /* 16-bit lanes: (lane & ditherSize) - ditherOffset, added to the zero-extended byte, clamped, shifted */
            const uint32_t lane = (uint32_t)st[2 * j] | ((uint32_t)st[2 * j + 1] << 8);
            f[i + j] = dither_px(f[i + j], lane, shift);
          }
      }
      memcpy(&hash, st, 8);
    }
  }
  for (; i < n; i++)
  {
    const uint32_t r = pcg_step(&hash);
    if (f) f[i] = dither_px(f[i], r, shift);
  }
  return hash;
}

uint64_t limg_oracle_dither(int shift, size_t n, uint64_t hash, uint8_t *f, int dither_mode) { return dither_impl(shift, n, hash, f, dither_mode); }
uint64_t limg_oracle_chain_step(size_t n, uint64_t hash, int dither_mode) { return dither_impl(1, n, hash, NULL, dither_mode); }

/* ------------------------------------------------------------------------------------------------------------------ */
/* a16: src/limg_decode.h:137-236 (4 ch) / :36-135 (3 ch) */
void limg_oracle_block_decode(uint32_t *out, size_t stride, size_t rx, size_t ry, int channels, const limg_oracle_record *rec, const uint8_t *A, const uint8_t *B,
                              const uint8_t *C, const uint8_t shift[3])
{
  recon_consts k;
  recon_setup(rec, channels, shift, channels == 3, &k);
  size_t i = 0;
  for (size_t y = 0; y < ry; y++)
    for (size_t x = 0; x < rx; x++, i++)
    {
      const uint32_t dA = (uint32_t)A[i] * k.mul[0], dB = (uint32_t)B[i] * k.mul[1], dC = (uint32_t)C[i] * k.mul[2];
      uint32_t p = 0;
      for (int c = 0; c < 4; c++) p |= (uint32_t)recon_channel(&k, c, dA, dB, dC) << (8 * c);
      out[y * stride + x] = p;
    }
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* driver: src/limg.cpp:1887-2103 (per strip) and :2105-2138 (strip partition)                                        */

void limg_oracle_default_config(limg_oracle_config *cfg)
{
  memset(cfg, 0, sizeof(*cfg));
  cfg->error_factor = 100;
  cfg->fast_bit_crush = 1;
  cfg->float_mode = LIMG_ORACLE_FLOAT_X86;
  cfg->dither_mode = LIMG_ORACLE_DITHER_AES;
  cfg->forced_shift[0] = cfg->forced_shift[1] = cfg->forced_shift[2] = -1;
}

typedef struct
{
  const uint32_t *pIn; size_t sizeX, sizeY; int channels; const limg_oracle_info *info; const limg_oracle_config *cfg;
  limg_oracle_record *pRecords; uint8_t *pShifts, *pPreA, *pPreB, *pPreC;
  size_t y_start, y_end; uint64_t trials;
} strip_job;

static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void encode_strip(strip_job *j)
{
  const size_t sizeX = j->sizeX, sizeY = j->sizeY;
  const int channels = j->channels;
  const limg_oracle_config *cfg = j->cfg;
  const size_t blocksX = (sizeX + LIMG_BLOCK - 1) / LIMG_BLOCK;
  uint64_t ditherLast = 0xCA7F00D15BADF00DULL; /* per strip, src/limg.cpp:1893 */
  uint32_t pixels[64] = { 0 }; /* persists across the blocks of the strip, like upstream's (src/limg.cpp:1890) */
  uint8_t A[64], B[64], C[64];
  static const uint8_t bit_to_pattern[9] = { 0, 0x22, 0x44, 0x66, 0x88, 0xAA, 0xCC, 0xEE, 0xFF };

  for (size_t y = j->y_start; y < j->y_end; y += LIMG_BLOCK)
    for (size_t x = 0; x < sizeX; x += LIMG_BLOCK)
    {
      const size_t rx = sizeX - x < LIMG_BLOCK ? sizeX - x : LIMG_BLOCK;
      const size_t ry = sizeY - y < LIMG_BLOCK ? sizeY - y : LIMG_BLOCK;
      const size_t n = rx * ry;
      for (size_t yy = 0; yy < ry; yy++) memcpy(pixels + yy * rx, j->pIn + (y + yy) * sizeX + x, rx * sizeof(uint32_t));

      limg_oracle_record rec;
      limg_oracle_block_fit_gathered(pixels, n, n < 4 ? 4 : n, channels, cfg->float_mode, &rec);
      limg_oracle_block_factors(pixels, n, channels, &rec, A, B, C);

      if (j->pPreA)
        for (size_t yy = 0, i = 0; yy < ry; yy++)
          for (size_t xx = 0; xx < rx; xx++, i++)
          {
            const size_t o = (y + yy) * sizeX + x + xx;
            j->pPreA[o] = A[i]; j->pPreB[o] = B[i]; j->pPreC[o] = C[i];
          }

      uint8_t shift[3] = { 0, 0, 0 };
      if (cfg->forced_shift[0] >= 0)
      {
        for (int i = 0; i < 3; i++) shift[i] = (uint8_t)cfg->forced_shift[i];
      }
      else
        j->trials += limg_oracle_block_search(pixels, n, channels, &rec, A, B, C, cfg->error_factor, cfg->fast_bit_crush, shift);

      if (shift[0] || shift[1] || shift[2])
      { /* src/limg.cpp:1947-1958: dither only factors with shift not in {0, 8}; those keep their raw byte */
        if (shift[0] && shift[0] != 8) ditherLast = dither_impl(shift[0], n, ditherLast, A, cfg->dither_mode);
        if (shift[1] && shift[1] != 8) ditherLast = dither_impl(shift[1], n, ditherLast, B, cfg->dither_mode);
        if (shift[2] && shift[2] != 8) ditherLast = dither_impl(shift[2], n, ditherLast, C, cfg->dither_mode);
      }

      const size_t bi = (y / LIMG_BLOCK) * blocksX + x / LIMG_BLOCK;
      if (j->pRecords) j->pRecords[bi] = rec;
      if (j->pShifts) memcpy(j->pShifts + 3 * bi, shift, 3);

      if (j->info)
      { /* a15: src/limg.cpp:2004-2093 */
        const limg_oracle_info *info = j->info;
        const uint32_t shift_val = 0xFF000000u | ((uint32_t)bit_to_pattern[shift[0]] << 16) | ((uint32_t)bit_to_pattern[shift[1]] << 8) | bit_to_pattern[shift[2]];
        uint32_t col[6] = { 0, 0, 0, 0, 0, 0 };
        for (int c = 0; c < channels; c++)
        {
          col[0] |= (uint32_t)clampi(rec.dirA_min[c], 0, 0xFF) << (8 * c);
          col[1] |= (uint32_t)clampi(rec.dirA_max[c], 0, 0xFF) << (8 * c);
          col[2] |= (uint32_t)clampi(rec.dirB_offset[c] + 0x80, 0, 0xFF) << (8 * c);
          col[3] |= (uint32_t)clampi(rec.dirB_mag[c] + 0x80, 0, 0xFF) << (8 * c);
          col[4] |= (uint32_t)clampi(rec.dirC_offset[c] + 0x80, 0, 0xFF) << (8 * c);
          col[5] |= (uint32_t)clampi(rec.dirC_mag[c] + 0x80, 0, 0xFF) << (8 * c);
        }
        if (channels == 3)
          for (int k = 0; k < 6; k++) col[k] |= 0xFF000000u;
        for (size_t yy = 0, i = 0; yy < ry; yy++)
          for (size_t xx = 0; xx < rx; xx++, i++)
          {
            const size_t o = (y + yy) * sizeX + x + xx;
            info->pFactorsA[o] = (uint8_t)(A[i] << shift[0]); /* shift 8 => 0 */
            info->pFactorsB[o] = (uint8_t)(B[i] << shift[1]);
            info->pFactorsC[o] = (uint8_t)(C[i] << shift[2]);
            info->pShiftABCX[o] = shift_val;
            info->pColAMin[o] = col[0]; info->pColAMax[o] = col[1]; info->pColBMin[o] = col[2];
            info->pColBMax[o] = col[3]; info->pColCMin[o] = col[4]; info->pColCMax[o] = col[5];
          }
        limg_oracle_block_decode(info->pDecoded + y * sizeX + x, sizeX, rx, ry, channels, &rec, A, B, C, shift);
      }
    }
}

typedef struct { strip_job *jobs; size_t count; size_t next; pthread_mutex_t mtx; } strip_queue;

static void *strip_worker(void *p)
{
  strip_queue *q = (strip_queue *)p;
  for (;;)
  {
    pthread_mutex_lock(&q->mtx);
    const size_t i = q->next++;
    pthread_mutex_unlock(&q->mtx);
    if (i >= q->count) break;
    encode_strip(&q->jobs[i]);
  }
  return NULL;
}

int limg_oracle_encode3d(const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, const limg_oracle_info *pInfo, const limg_oracle_config *cfg,
                         limg_oracle_record *pRecords, uint8_t *pShifts, uint8_t *pPreA, uint8_t *pPreB, uint8_t *pPreC, uint64_t *pTrialCount)
{
  if (pIn == NULL || cfg == NULL) return 102; /* limg_error_ArgumentNull */

  /* strip partition, src/limg.cpp:2114-2134 */
  size_t thread_count = 1, y_range = sizeY;
  if (cfg->pool_threads > 0)
  {
    thread_count = (size_t)cfg->pool_threads * 4;
    y_range = ((sizeY / LIMG_BLOCK) / thread_count) * LIMG_BLOCK;
    if (y_range == 0)
    {
      thread_count = (size_t)cfg->pool_threads;
      y_range = ((sizeY / LIMG_BLOCK) / thread_count) * LIMG_BLOCK;
    }
  }

  strip_job *jobs = (strip_job *)calloc(thread_count, sizeof(strip_job));
  if (!jobs) return 104;
  size_t y_start = 0;
  for (size_t i = 0; i < thread_count; i++)
  {
    strip_job *j = &jobs[i];
    j->pIn = pIn; j->sizeX = sizeX; j->sizeY = sizeY; j->channels = hasAlpha ? 4 : 3; j->info = pInfo; j->cfg = cfg;
    j->pRecords = pRecords; j->pShifts = pShifts; j->pPreA = pPreA; j->pPreB = pPreB; j->pPreC = pPreC;
    j->y_start = y_start;
    j->y_end = (i + 1 == thread_count) ? sizeY : y_start + y_range;
    y_start += y_range;
  }

  size_t workers = cfg->worker_threads > 1 ? (size_t)cfg->worker_threads : 1;
  if (workers > thread_count) workers = thread_count;
  strip_queue q = { jobs, thread_count, 0, PTHREAD_MUTEX_INITIALIZER };
  if (workers <= 1)
    strip_worker(&q);
  else
  {
    pthread_t *th = (pthread_t *)calloc(workers, sizeof(pthread_t));
    for (size_t i = 0; i < workers; i++) pthread_create(&th[i], NULL, strip_worker, &q);
    for (size_t i = 0; i < workers; i++) pthread_join(th[i], NULL);
    free(th);
  }
  uint64_t trials = 0;
  for (size_t i = 0; i < thread_count; i++) trials += jobs[i].trials;
  if (pTrialCount) *pTrialCount = trials;
  free(jobs);
  return 0;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* a18: src/limg_internal.h:376-410 and src/limg.cpp:2455-2491 */
static inline size_t color_error(uint32_t a, uint32_t b, int channels)
{
  int e0 = (int)(a & 0xFF) - (int)(b & 0xFF);
  const size_t red = (size_t)(e0 * e0);
  static const uint8_t f_low[4] = { 2, 4, 3, 3 }, f_high[4] = { 3, 4, 2, 3 };
  const uint8_t *f = red < 0x4000 ? f_low : f_high;
  size_t err = red * f[0];
  for (int c = 1; c < channels; c++)
  {
    const int e = (int)((a >> (8 * c)) & 0xFF) - (int)((b >> (8 * c)) & 0xFF);
    err += (size_t)(e * e) * f[c];
  }
  return err;
}

double limg_oracle_compare(const uint32_t *pA, const uint32_t *pB, size_t sizeX, size_t sizeY, int hasAlpha, double *pMse, double *pMax)
{
  const int channels = hasAlpha ? 4 : 3;
  size_t error = 0;
  const size_t maxError = color_error(0u, 0xFFFFFFFFu, channels);
  for (size_t i = 0; i < sizeX * sizeY; i++) error += color_error(pA[i], pB[i], channels);
  const double mse = (double)error / (double)(sizeX * sizeY);
  const double psnr = 10.0 * log10((double)maxError / mse);
  if (pMse) *pMse = mse;
  if (pMax) *pMax = (double)maxError;
  return psnr;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* helpers                                                                                                            */

uint64_t limg_oracle_fnv1a64(const void *p, size_t n)
{
  const uint8_t *b = (const uint8_t *)p;
  uint64_t h = 0xCBF29CE484222325ULL;
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001B3ULL;
  return h;
}

void limg_oracle_pack_record3(const limg_oracle_record *in, void *out48)
{
  uint8_t *o = (uint8_t *)out48;
  memcpy(o, in->avg, 12);
  const int16_t *src[6] = { in->dirA_min, in->dirA_max, in->dirB_offset, in->dirB_mag, in->dirC_offset, in->dirC_mag };
  for (int k = 0; k < 6; k++) memcpy(o + 12 + 6 * k, src[k], 6);
}

void limg_oracle_unpack_record3(const void *in48, limg_oracle_record *out)
{
  const uint8_t *i = (const uint8_t *)in48;
  memset(out, 0, sizeof(*out));
  memcpy(out->avg, i, 12);
  int16_t *dst[6] = { out->dirA_min, out->dirA_max, out->dirB_offset, out->dirB_mag, out->dirC_offset, out->dirC_mag };
  for (int k = 0; k < 6; k++) memcpy(dst[k], i + 12 + 6 * k, 6);
}

static inline uint64_t sm64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

void limg_oracle_synth_random_gradient(uint32_t *out, size_t w, size_t h, uint64_t seed, int opaque)
{
  for (size_t y = 0; y < h; y++)
    for (size_t x = 0; x < w; x++)
    {
      const uint64_t hh = sm64(seed ^ ((uint64_t)(y >> 6) * 0x9E3779B97F4A7C15ULL + (uint64_t)(x >> 6)));
      const uint64_t h2 = sm64(hh);
      const int64_t gx = (int64_t)(h2 & 7), gy = (int64_t)((h2 >> 3) & 7);
      int64_t s = (int64_t)(x & 63) * gx + (int64_t)(y & 63) * gy, m = 63 * (gx + gy);
      if (m == 0) { m = 1; s = 0; }
      uint32_t p = 0;
      for (int c = 0; c < 4; c++)
      {
        const int64_t c0 = (int64_t)((hh >> (8 * c)) & 255), c1 = (int64_t)((hh >> (32 + 8 * c)) & 255);
        int64_t v = (c0 * (m - s) + c1 * s + m / 2) / m;
        if (c == 3 && opaque) v = 255;
        p |= (uint32_t)(v & 255) << (8 * c);
      }
      out[y * w + x] = p;
    }
}

void limg_oracle_synth_photo_noise(uint32_t *out, size_t w, size_t h, uint64_t seed)
{
  for (size_t y = 0; y < h; y++)
    for (size_t x = 0; x < w; x++)
    {
      const uint64_t ly = y >> 5, lx = x >> 5;
      const int64_t fy = (int64_t)(y & 31), fx = (int64_t)(x & 31);
      const uint64_t ha = sm64(seed ^ (ly * 0x9E3779B97F4A7C15ULL + lx)), hb = sm64(seed ^ (ly * 0x9E3779B97F4A7C15ULL + lx + 1));
      const uint64_t hc = sm64(seed ^ ((ly + 1) * 0x9E3779B97F4A7C15ULL + lx)), hd = sm64(seed ^ ((ly + 1) * 0x9E3779B97F4A7C15ULL + lx + 1));
      const uint64_t hn = sm64(seed * 31 + (uint64_t)y * (uint64_t)w + (uint64_t)x);
      uint32_t p = 0xFF000000u;
      for (int c = 0; c < 3; c++)
      {
        const int64_t a = (int64_t)((ha >> (8 * c)) & 255), b = (int64_t)((hb >> (8 * c)) & 255), cc = (int64_t)((hc >> (8 * c)) & 255), d = (int64_t)((hd >> (8 * c)) & 255);
        int64_t v = ((a * (32 - fx) + b * fx) * (32 - fy) + (cc * (32 - fx) + d * fx) * fy + 512) >> 10;
        v += (int64_t)((hn >> (8 * c)) & 15) - 8;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        p |= (uint32_t)v << (8 * c);
      }
      out[y * w + x] = p;
    }
}

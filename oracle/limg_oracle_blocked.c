/* TEST INFRASTRUCTURE -- CPU restatement of the reference's merged-block encoder `limg_blocked_encode3d_test`
 * (src/limg.cpp:2329-2453), built on the per-block pieces of limg_oracle.c.
 *
 *   pass 1   src/limg.cpp:1088-1119   every 8x8 block: channel sums + direction fit (a4-a6)
 *   match    src/limg.cpp:1137-1269   "may block b join the region seeded by block a" (float heuristics, restated op for op)
 *   search   src/limg.cpp:1271-1496   greedy raster search for rectangles of matching, unused blocks
 *   region   src/limg.cpp:1498-1772   re-fit over the region's pixels (row-major), factors, shift search, dither, planes, decode
 *   driver   src/limg.cpp:1774-1885   large regions, then small regions, then the remaining single blocks (which keep their pass-1 fit)
 *
 * One dither chain runs through all regions in creation order (src/limg_internal.h:706-713).  Pinned against the real reference
 * by tests/test_oracle_blocked.py (all 13 planes written by upstream; pBlockError is never written upstream).
 */
#include "limg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BLK 8

/* ---- src/limg_internal.h:426-452 + src/limg_factorization.h:9-42: scalar error state and single-colour factors ------------------- */
typedef struct { float nA[4], nB[4], nC[4], invA, invB, invC; } err_state;

static float dot_seq(const float *a, const float *b, int channels)
{ /* limg_dot, src/limg_internal.h:357-366: sum = 0; sum += a[i] * b[i] in index order */
  float sum = 0.0f;
  for (int i = 0; i < channels; i++) sum += a[i] * b[i];
  return sum;
}

static void init_state(const limg_oracle_record *in, int channels, err_state *s)
{
  int nz[3] = { 0, 0, 0 };
  memset(s, 0, sizeof(*s));
  for (int i = 0; i < channels; i++)
  {
    s->nA[i] = (float)((int)in->dirA_max[i] - (int)in->dirA_min[i]);
    s->nB[i] = (float)((int)in->dirB_mag[i] - (int)in->dirB_offset[i]);
    s->nC[i] = (float)((int)in->dirC_mag[i] - (int)in->dirC_offset[i]);
    nz[0] |= s->nA[i] != 0; nz[1] |= s->nB[i] != 0; nz[2] |= s->nC[i] != 0;
  }
  if (nz[0]) s->invA = 1.0f / dot_seq(s->nA, s->nA, channels);
  if (nz[1]) s->invB = 1.0f / dot_seq(s->nB, s->nB, channels);
  if (nz[2]) s->invC = 1.0f / dot_seq(s->nC, s->nC, channels);
}

static void colour_factors(const float *color, const limg_oracle_record *in, const err_state *s, int channels, float *fa, float *fb, float *fc)
{
  float t[4], est[4];
  for (int i = 0; i < channels; i++) t[i] = (float)(color[i] - (float)in->dirA_min[i]);
  const float facA = dot_seq(t, s->nA, channels) * s->invA;
  for (int i = 0; i < channels; i++)
  {
    est[i] = (float)in->dirA_min[i] + facA * s->nA[i];
    t[i] = (color[i] - est[i]) - (float)in->dirB_offset[i];
  }
  const float facB = dot_seq(t, s->nB, channels) * s->invB;
  for (int i = 0; i < channels; i++)
  {
    est[i] = est[i] + facB * s->nB[i];
    t[i] = (color[i] - est[i]) - (float)in->dirC_offset[i];
  }
  *fa = facA; *fb = facB; *fc = dot_seq(t, s->nC, channels) * s->invC;
}

/* src/limg.cpp:1137-1262.  a = the region's seed block, b = the candidate. */
int limg_oracle_blocked_matches(int channels, const limg_oracle_record *a, const limg_oracle_record *b)
{
  err_state sa, sb;
  init_state(a, channels, &sa);
  init_state(b, channels, &sb);
  static const float w[4] = { 2, 4, 3, 3 };
  float avgDiffSq = 0;
  float lenA[3] = { 3, 3, 3 }, lenB[3] = { 3, 3, 3 };
  for (int i = 0; i < channels; i++)
  {
    const float d = a->avg[i] - b->avg[i];
    avgDiffSq += d * d * w[i];
    lenA[0] += (sa.nA[i] * sa.nA[i]) * w[i]; lenB[0] += (sb.nA[i] * sb.nA[i]) * w[i];
    lenA[1] += (sa.nB[i] * sa.nB[i]) * w[i]; lenB[1] += (sb.nB[i] * sb.nB[i]) * w[i];
    lenA[2] += (sa.nC[i] * sa.nC[i]) * w[i]; lenB[2] += (sb.nC[i] * sb.nC[i]) * w[i];
  }
  const float sumA = lenA[0] + lenA[1] + lenA[2], sumB = lenB[0] + lenB[1] + lenB[2];
  const float ratio = (sumA + 1) / (sumB + 1);
  const float maxAcceptAvgDiff = (float)(16 * 3 * channels), maxAcceptRange = (float)(200 * 3 * channels);
  if (avgDiffSq < maxAcceptAvgDiff && sumA < maxAcceptRange && sumB < maxAcceptRange) return 1;
  const float maxRatio = 1.375f;
  if (ratio > maxRatio || ratio < (1.f / maxRatio)) return 0;

  float invA[3], invB[3];
  for (int i = 0; i < 3; i++) { invA[i] = 1.0f / lenA[i]; invB[i] = 1.0f / lenB[i]; }
  for (int i = 1; i < 3; i++) { invA[i] *= 2.f; invB[i] *= 2.f; }

  float color[4], fa, fb, fc, sumFactors = 0;
  for (int z = 0; z < 3; z++)
  {
    const float zf = z * 0.5f;
    for (int y = 0; y < 3; y++)
    {
      const float yf = y * 0.5f;
      for (int x = 0; x < 3; x++)
      {
        const float xf = x * 0.5f;
        for (int i = 0; i < channels; i++) color[i] = sb.nA[i] * xf + sb.nB[i] * yf + sb.nC[i] * zf;
        colour_factors(color, a, &sa, channels, &fa, &fb, &fc);
        sumFactors += fabsf(fa) * invA[0] + fabsf(0.5f - fb) * invA[1] + fabsf(0.5f - fc) * invA[2];
        /* upstream builds a second colour from stateA here and then does not use it: the call below takes a.avg (:1236-1239) */
        colour_factors(a->avg, b, &sb, channels, &fa, &fb, &fc);
        sumFactors += fabsf(fa) * invB[0] + fabsf(0.5f - fb) * invB[1] + fabsf(0.5f - fc) * invB[2];
      }
    }
  }
  const float avgFactors = sumFactors * (1.f / (3 * 3 * 3));
  return avgFactors < 3.0f;
}

/* ---- src/limg.cpp:1121-1135, :1271-1286, :1288-1384: area checks and rectangle growth ------------------------------------------------ */
typedef struct
{
  const uint32_t *pIn; size_t sizeX, sizeY, blockX, blockY; int channels;
  const limg_oracle_config *cfg;
  limg_oracle_record *decomp;
  uint8_t *inUse;
  const limg_oracle_blocked_info *info;
  uint64_t ditherLast; uint32_t blockIndex;
  limg_oracle_region *regions; size_t regionCap, regionCount;
} bctx;

static int area_unused(const bctx *c, size_t ox, size_t oy, size_t rx, size_t ry)
{
  for (size_t y = 0; y < ry; y++)
    for (size_t x = 0; x < rx; x++)
      if (c->inUse[(oy + y) * c->blockX + ox + x]) return 0;
  return 1;
}

static int area_matches(const bctx *c, size_t ox, size_t oy, size_t rx, size_t ry, const limg_oracle_record *seed)
{
  for (size_t y = 0; y < ry; y++)
    for (size_t x = 0; x < rx; x++)
      if (!limg_oracle_blocked_matches(c->channels, seed, &c->decomp[(oy + y) * c->blockX + ox + x])) return 0;
  return 1;
}

/* grows right, down, (up, left) one block row / column at a time while the new strip is unused and matches the seed block */
static void expand(const bctx *c, size_t *pox, size_t *poy, size_t *prx, size_t *pry, int upLeft, limg_oracle_record *outSeed)
{
  size_t ox = *pox, oy = *poy, rx = *prx, ry = *pry;
  int up = upLeft, down = 1, left = upLeft, right = 1;
  const limg_oracle_record seed = c->decomp[ox + oy * c->blockX];
  while (up || down || left || right)
  {
    if (right)
    {
      if (ox + rx + 1 < c->blockX && area_unused(c, ox + rx, oy, 1, ry) && area_matches(c, ox + rx, oy, 1, ry, &seed)) rx++;
      else right = 0;
    }
    if (down)
    {
      if (oy + ry + 1 < c->blockY && area_unused(c, ox, oy + ry, rx, 1) && area_matches(c, ox, oy + ry, rx, 1, &seed)) ry++;
      else down = 0;
    }
    if (upLeft)
    {
      if (up)
      {
        if (oy > 0 && area_unused(c, ox, oy - 1, rx, 1) && area_matches(c, ox, oy - 1, rx, 1, &seed)) { oy--; ry++; }
        else up = 0;
      }
      if (left)
      {
        if (ox > 0 && area_unused(c, ox - 1, oy, 1, ry) && area_matches(c, ox - 1, oy, 1, ry, &seed)) { ox--; rx++; }
        else left = 0;
      }
    }
  }
  *pox = ox; *poy = oy; *prx = rx; *pry = ry; *outSeed = seed;
}

/* src/limg.cpp:1386-1496.  Returns 1 with a rectangle, 0 when the raster scan is exhausted. */
static int find_block(const bctx *c, int acceptTiny, size_t *staticX, size_t *staticY, size_t *pox, size_t *poy, size_t *prx, size_t *pry, limg_oracle_record *seed)
{
  size_t ox = *staticX, oy = *staticY;
  for (; oy < c->blockY; oy++)
  {
    for (; ox < c->blockX; ox++)
    {
      if (c->inUse[oy * c->blockX + ox]) continue;
      size_t x = ox, y = oy, rx = 1, ry = 1;
      expand(c, &x, &y, &rx, &ry, 0, seed);
      if (rx == 1 && ry == 1) continue;
      const limg_oracle_record first = *seed;
      if (!acceptTiny)
      {
        if (rx >= 3 && ry >= 3)
        { /* retry from the centre third, growing in all four directions; keep it if it covers more blocks */
          size_t cx = ox + rx / 3, cy = oy + ry / 3, crx = rx / 3, cry = ry / 3;
          expand(c, &cx, &cy, &crx, &cry, 1, seed);
          if (crx * cry > rx * ry)
          {
            *pox = cx; *poy = cy; *prx = crx; *pry = cry;
            *staticX = ox; *staticY = oy;
            return 1;
          }
          *pox = ox; *poy = oy; *prx = rx; *pry = ry;
          *staticX = ox + rx; *staticY = oy;
          *seed = first;
          return 1;
        }
      }
      else
      { /* rx > 1 || ry > 1 holds here */
        *pox = ox; *poy = oy; *prx = rx; *pry = ry;
        *staticX = ox + rx; *staticY = oy;
        *seed = first;
        return 1;
      }
    }
    ox = 0;
  }
  *staticX = ox; *staticY = oy;
  return 0;
}

/* ---- src/limg.cpp:1498-1772: one region ------------------------------------------------------------------------------------------------ */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static int encode_region(bctx *c, size_t ox, size_t oy, size_t rx, size_t ry, const limg_oracle_record *keep)
{
  size_t x_px = rx * BLK, y_px = ry * BLK;
  if (ox + rx == c->blockX && (c->sizeX % BLK)) x_px = x_px - BLK + c->sizeX % BLK;
  if (oy + ry == c->blockY && (c->sizeY % BLK)) y_px = y_px - BLK + c->sizeY % BLK;
  const size_t n = x_px * y_px, sizeX = c->sizeX, px0 = ox * BLK, py0 = oy * BLK;
  const int channels = c->channels;
  const limg_oracle_config *cfg = c->cfg;

  uint32_t *px = (uint32_t *)malloc(n * sizeof(uint32_t));
  uint8_t *A = (uint8_t *)malloc(3 * n), *B = A + n, *C = B + n;
  if (!px || !A) { free(px); free(A); return 104; }
  for (size_t yy = 0; yy < y_px; yy++) memcpy(px + yy * x_px, c->pIn + (py0 + yy) * sizeX + px0, x_px * sizeof(uint32_t));

  limg_oracle_record rec;
  if (keep) rec = *keep;
  else limg_oracle_block_fit(px, n, channels, cfg->float_mode, &rec);
  limg_oracle_block_factors(px, n, channels, &rec, A, B, C);

  uint8_t shift[3] = { 0, 0, 0 };
  if (cfg->forced_shift[0] >= 0)
    for (int i = 0; i < 3; i++) shift[i] = (uint8_t)cfg->forced_shift[i];
  else
    limg_oracle_block_search(px, n, channels, &rec, A, B, C, cfg->error_factor, cfg->fast_bit_crush, shift);

  uint8_t calls = 0;
  if (shift[0] || shift[1] || shift[2])
  {
    if (shift[0] && shift[0] != 8) { c->ditherLast = limg_oracle_dither(shift[0], n, c->ditherLast, A, cfg->dither_mode); calls++; }
    if (shift[1] && shift[1] != 8) { c->ditherLast = limg_oracle_dither(shift[1], n, c->ditherLast, B, cfg->dither_mode); calls++; }
    if (shift[2] && shift[2] != 8) { c->ditherLast = limg_oracle_dither(shift[2], n, c->ditherLast, C, cfg->dither_mode); calls++; }
  }

  if (c->regions && c->regionCount < c->regionCap)
  {
    limg_oracle_region *r = &c->regions[c->regionCount];
    r->ox = (uint32_t)ox; r->oy = (uint32_t)oy; r->rx = (uint32_t)rx; r->ry = (uint32_t)ry;
    r->keep = keep != NULL; r->calls = calls;
    memcpy(r->shift, shift, 3);
    r->rec = rec;
  }
  c->regionCount++;

  const limg_oracle_blocked_info *info = c->info;
  if (info)
  {
    static const uint8_t pat[9] = { 0, 0x22, 0x44, 0x66, 0x88, 0xAA, 0xCC, 0xEE, 0xFF };
    const uint32_t shift_val = 0xFF000000u | ((uint32_t)pat[shift[0]] << 16) | ((uint32_t)pat[shift[1]] << 8) | pat[shift[2]];
    uint32_t col[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < channels; i++)
    {
      col[0] |= (uint32_t)clampi(rec.dirA_min[i], 0, 0xFF) << (8 * i);
      col[1] |= (uint32_t)clampi(rec.dirA_max[i], 0, 0xFF) << (8 * i);
      col[2] |= (uint32_t)clampi(rec.dirB_offset[i] + 0x80, 0, 0xFF) << (8 * i);
      col[3] |= (uint32_t)clampi(rec.dirB_mag[i] + 0x80, 0, 0xFF) << (8 * i);
      col[4] |= (uint32_t)clampi(rec.dirC_offset[i] + 0x80, 0, 0xFF) << (8 * i);
      col[5] |= (uint32_t)clampi(rec.dirC_mag[i] + 0x80, 0, 0xFF) << (8 * i);
    }
    if (channels == 3)
      for (int k = 0; k < 6; k++) col[k] |= 0xFF000000u;
    /* src/limg.cpp:1629-1636: 110 (3 ch) / 136 (4 ch) header bits + (8 - shift) bits per factor per pixel, rounded per pixel */
    const size_t staticBits = (size_t)channels * 9 * 2 + (size_t)channels * 8 + 2 * 16;
    const size_t bits = staticBits + n * (size_t)((8 - shift[0]) + (8 - shift[1]) + (8 - shift[2]));
    const uint8_t bpp = (uint8_t)((bits + n / 2) / n);
    size_t i = 0;
    for (size_t yy = 0; yy < y_px; yy++)
      for (size_t xx = 0; xx < x_px; xx++, i++)
      {
        const size_t o = (py0 + yy) * sizeX + px0 + xx;
        info->pFactorsA[o] = (uint8_t)(A[i] << shift[0]);
        info->pFactorsB[o] = (uint8_t)(B[i] << shift[1]);
        info->pFactorsC[o] = (uint8_t)(C[i] << shift[2]);
        info->pBitsPerPixel[o] = bpp;
        info->pShiftABCX[o] = shift_val;
        info->pColAMin[o] = col[0]; info->pColAMax[o] = col[1]; info->pColBMin[o] = col[2];
        info->pColBMax[o] = col[3]; info->pColCMin[o] = col[4]; info->pColCMax[o] = col[5];
        info->pBlockIndex[o] = 0xFF000000u | c->blockIndex;
      }
    limg_oracle_block_decode(info->pDecoded + py0 * sizeX + px0, sizeX, x_px, y_px, channels, &rec, A, B, C, shift);
  }
  free(px); free(A);
  return 0;
}

static void claim(bctx *c, size_t ox, size_t oy, size_t rx, size_t ry)
{
  c->blockIndex++;
  for (size_t y = oy; y < oy + ry; y++)
    for (size_t x = ox; x < ox + rx; x++) c->inUse[x + y * c->blockX] = 1;
}

int limg_oracle_blocked_encode3d(const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, const limg_oracle_blocked_info *pInfo, const limg_oracle_config *cfg,
                                 limg_oracle_record *pPass1, limg_oracle_region *pRegions, size_t regionCap, size_t *pRegionCount)
{
  if (!pIn || !cfg) return 102;
  bctx c;
  memset(&c, 0, sizeof(c));
  c.pIn = pIn; c.sizeX = sizeX; c.sizeY = sizeY; c.channels = hasAlpha ? 4 : 3; c.cfg = cfg; c.info = pInfo;
  c.blockX = (sizeX + BLK - 1) / BLK; c.blockY = (sizeY + BLK - 1) / BLK;
  c.ditherLast = 0xCA7F00D15BADF00DULL;
  c.regions = pRegions; c.regionCap = regionCap;
  const size_t blocks = c.blockX * c.blockY;
  c.decomp = (limg_oracle_record *)calloc(blocks, sizeof(limg_oracle_record));
  c.inUse = (uint8_t *)calloc(blocks, 1);
  if (!c.decomp || !c.inUse) { free(c.decomp); free(c.inUse); return 104; }

  /* pass 1: src/limg.cpp:1088-1119 (the thread pool only splits this loop; it carries no state) */
  uint32_t pixels[64] = { 0 }; /* persistent gather buffer: blocks of fewer than 4 pixels also sum what the previous block left in it (see limg_oracle_block_fit_gathered) */
  for (size_t by = 0; by < c.blockY; by++)
    for (size_t bx = 0; bx < c.blockX; bx++)
    {
      const size_t rx = sizeX - bx * BLK < BLK ? sizeX - bx * BLK : BLK, ry = sizeY - by * BLK < BLK ? sizeY - by * BLK : BLK;
      for (size_t yy = 0; yy < ry; yy++) memcpy(pixels + yy * rx, pIn + (by * BLK + yy) * sizeX + bx * BLK, rx * sizeof(uint32_t));
      limg_oracle_block_fit_gathered(pixels, rx * ry, rx * ry < 4 ? 4 : rx * ry, c.channels, cfg->float_mode, &c.decomp[by * c.blockX + bx]);
    }
  if (pPass1) memcpy(pPass1, c.decomp, blocks * sizeof(limg_oracle_record));

  int result = 0;
  for (int acceptTiny = 0; acceptTiny < 2 && !result; acceptTiny++)
  { /* src/limg.cpp:1817-1861: large rectangles first, then anything larger than one block */
    size_t sx = 0, sy = 0, ox, oy, rx, ry;
    limg_oracle_record seed;
    while (!result && find_block(&c, acceptTiny, &sx, &sy, &ox, &oy, &rx, &ry, &seed))
    {
      claim(&c, ox, oy, rx, ry);
      result = encode_region(&c, ox, oy, rx, ry, NULL);
    }
  }
  /* src/limg.cpp:1863-1881: what is left keeps its pass-1 decomposition */
  for (size_t y = 0; y < c.blockY && !result; y++)
    for (size_t x = 0; x < c.blockX && !result; x++)
    {
      if (c.inUse[x + y * c.blockX]) continue;
      const limg_oracle_record d = c.decomp[x + y * c.blockX];
      claim(&c, x, y, 1, 1);
      result = encode_region(&c, x, y, 1, 1, &d);
    }
  if (pRegionCount) *pRegionCount = c.regionCount;
  free(c.decomp); free(c.inUse);
  return result;
}

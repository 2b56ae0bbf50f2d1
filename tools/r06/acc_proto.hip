// tools/r06/acc_proto.hip -- MEASUREMENT PROTOTYPE, not part of the product ("the trial + automaton alone ... same-box A/B ... kill criterion 1.2 x").
//
// The accurate shift search (src/limg_bit_crush.h:668-830; trial src/limg_bit_crush_simd.h:562-810) on pre-computed inputs -- per block its 64 pixels, its 64 x 3
// pre-dither factor bytes and its record -- in two mappings, both driven by the same 19 K-state automaton the library uses (limg_search_table_accurate.h), both writing
// the shift triple the reference's search ends with (checked against the CPU oracle by tools/r06/acc_proto.py):
//   mode 0  lane == pixel, wave == block: the library's mapping (terms cached between trials, shifts in scalar registers, one scalar load per trial);
//   mode 1  four blocks per wave, 16 lanes per block, one QUARTER of the block (16 pixels) per step: a trial stops at the first quarter that holds a pixel over the
//           limit or takes the partial block sum over its limit; every quarter wave walks its own automaton state; a finished quarter draws the wave's next block.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/r06/libacc_proto.so tools/r06/acc_proto.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "../../limg_amd/csrc/limg_search_table_accurate.h"

namespace
{
  typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
  struct __attribute__((aligned(32))) Entry { uint32_t w[8]; };

  __device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
  template <int CTRL, int ROWMASK> __device__ __forceinline__ int dpp(int oldv, int v) { return __builtin_amdgcn_update_dpp(oldv, v, CTRL, ROWMASK, 0xF, false); }
  __device__ __forceinline__ uint32_t wave_sum(uint32_t x)
  {
    int v = (int)x;
    v += dpp<0xB1, 0xF>(0, v); v += dpp<0x4E, 0xF>(0, v); v += dpp<0x141, 0xF>(0, v); v += dpp<0x140, 0xF>(0, v);
    v += dpp<0x142, 0xA>(0, v); v += dpp<0x143, 0xC>(0, v);
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
  }
  // sum over the 16 lanes of a DPP row, left in every lane of the row
  __device__ __forceinline__ uint32_t row_sum(uint32_t x)
  {
    int v = (int)x;
    v += dpp<0xB1, 0xF>(0, v); v += dpp<0x4E, 0xF>(0, v); v += dpp<0x141, 0xF>(0, v); v += dpp<0x140, 0xF>(0, v);
    return (uint32_t)v;
  }
  __device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
  __device__ __forceinline__ int mad_i24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
  __device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
  __device__ __forceinline__ uint32_t mul_u24_uniform(uint32_t a, uint32_t uniformB) { uint32_t r; asm("v_mul_u32_u24 %0, %2, %1" : "=v"(r) : "v"(a), "s"(uniformB)); return r; }
  __device__ __forceinline__ int mul_i24(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
  __device__ __forceinline__ int med3_i32(int a, int b, int c) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

  // the packed trial of the library (limg_hip_kernels.hip "a9, packed form"): negated, biased terms; see there for why this is exact
  __device__ __forceinline__ constexpr int term_bias(int factor) { return factor == 2 ? 0x2000 : 0x3000; }
  __device__ __forceinline__ int term_const(int f, int c, int lo) { return 255 - ((lo << 8) + 128) + (c < 2 ? (term_bias(f) << 8) : 0); }
  template <bool UNIFORM = false>
  __device__ __forceinline__ void make_terms(uint32_t f, uint32_t s, uint32_t mul, const int n[3], const int m[3], uint32_t &tRG, int &tB)
  {
    const int d = UNIFORM ? (int)mul_u24_uniform(f >> (s & 31u), mul) : (int)mul_u24(f >> (s & 31u), mul); // (mode 0: shift and multiplier are scalar registers, as in the library)
    const int t0 = mad_i24(d, n[0], m[0]), t1 = mad_i24(d, n[1], m[1]), t2 = mad_i24(d, n[2], m[2]);
    tRG = __builtin_amdgcn_perm((uint32_t)t1, (uint32_t)t0, 0x06050201u);
    tB = t2 >> 8;
  }
  __device__ __forceinline__ uint32_t pixel_error(uint32_t dRG, int dBraw, uint32_t loRG, uint32_t hiRG, int pxBlo, int pxB)
  {
    ushort2_t eu = __builtin_bit_cast(ushort2_t, dRG);
    eu = __builtin_elementwise_max(eu, __builtin_bit_cast(ushort2_t, loRG));
    eu = __builtin_elementwise_min(eu, __builtin_bit_cast(ushort2_t, hiRG));
    const int dB = med3_i32(dBraw, pxBlo, pxB);
    const ushort2_t sq = eu * eu;
    const uint32_t sqB = (uint32_t)mul_i24(dB, dB);
    const bool low_red = sq.x < 0x4000;
    const uint32_t half = __builtin_amdgcn_udot2(sq, __builtin_bit_cast(ushort2_t, 0x00020001u), sqB, false);
    const uint32_t extra = low_red ? sqB : (__builtin_bit_cast(uint32_t, sq) & 0xFFFFu);
    return (half << 1) + extra;
  }

  typedef unsigned int uint8s_t __attribute__((ext_vector_type(8)));
  __device__ __forceinline__ uint8s_t sload8(const Entry *base, uint32_t byteOffset)
  {
    uint8s_t v;
    asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(base), "s"(byteOffset) : "memory");
    return v;
  }

  struct Params
  {
    const uint32_t *px;    // [blocks][64]
    const uint32_t *fac;   // [blocks][64]: fA | fB << 8 | fC << 16 (pre-dither)
    const int16_t *rec;    // [blocks][24]: dirA_min[4] dirA_max[4] dirB_offset[4] dirB_mag[4] dirC_offset[4] dirC_mag[4]
    const Entry *table;
    uint32_t *shifts;      // [blocks]: a | b << 8 | c << 16
    uint32_t blocks, maxPixel32, blockLimit;
  };

  // record -> the trial's integer operands: n = -(max - min), m = term_const (factor A's without the pixel)
  __device__ __forceinline__ void record_consts(const int16_t *r, int nA[3], int nB[3], int nC[3], int cA[3], int cB[3], int cC[3])
  {
#pragma unroll
    for (int c = 0; c < 3; c++)
    {
      const int loA = r[c], hiA = r[4 + c], loB = r[8 + c], hiB = r[12 + c], loC = r[16 + c], hiC = r[20 + c];
      nA[c] = -(hiA - loA); nB[c] = -(hiB - loB); nC[c] = -(hiC - loC);
      cA[c] = term_const(0, c, loA); cB[c] = term_const(1, c, loB); cC[c] = term_const(2, c, loC);
    }
  }

  // ---- mode 0: the library's mapping ------------------------------------------------------------------------------------------------
  __global__ __launch_bounds__(256, 6) void k_acc_lane_pixel(const Params p)
  {
    const int lane = lane_id();
    const uint32_t nWaves = gridDim.x * 4u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    for (uint32_t blk = wave; blk < p.blocks; blk += nWaves)
    {
      const uint32_t px = p.px[(size_t)blk * 64 + lane], fac = p.fac[(size_t)blk * 64 + lane];
      const uint32_t fA = fac & 0xFF, fB = (fac >> 8) & 0xFF, fC = fac >> 16;
      int nA[3], nB[3], nC[3], mA[3], mB[3], mC[3];
      {
        int16_t r[24];
#pragma unroll
        for (int i = 0; i < 24; i++) r[i] = (int16_t)sgpr((int)p.rec[(size_t)blk * 24 + i]);
        record_consts(r, nA, nB, nC, mA, mB, mC);
#pragma unroll
        for (int c = 0; c < 3; c++) mA[c] += (int)(((px >> (8 * c)) & 0xFF) << 8);
        // uniform values kept in vector registers, as the library does: they are operands of v_mad_i32_i24, which takes one scalar operand at most
#pragma unroll
        for (int c = 0; c < 3; c++) asm volatile("" : "+v"(nA[c]), "+v"(nB[c]), "+v"(nC[c]), "+v"(mB[c]), "+v"(mC[c]));
      }
      const uint32_t R = px & 0xFF, G = (px >> 8) & 0xFF;
      const uint32_t hiRG = (R + 0x8000u) | ((G + 0x8000u) << 16), loRG = (R + 0x8000u - 255u) | ((G + 0x8000u - 255u) << 16);
      const int pxB = (int)((px >> 16) & 0xFF), pxBlo = pxB - 255;
      uint32_t tA_RG = 0, tB_RG = 0, tC_RG = 0;
      int tA_B = 0, tB_B = 0, tC_B = 0;
      uint32_t bestA = 0, bestB = 0, bestC = 0, minBe = 0xFFFFFFFFu;
      uint8s_t e = sload8(p.table, 0u);
      uint32_t mask = 7u;
      while (!(e[0] >> 31))
      {
        const uint32_t a = e[0] & 31u;
        if (mask & 1u) make_terms<true>(fA, a, e[5], nA, mA, tA_RG, tA_B);
        if (mask & 2u)
        {
          if (e[3] > 7) { tB_RG = (uint32_t)term_bias(1) * 0x10001u; tB_B = 0; }
          else make_terms<true>(fB, e[3], e[6], nB, mB, tB_RG, tB_B);
        }
        if (mask & 4u)
        {
          if (e[4] > 7) { tC_RG = (uint32_t)term_bias(2) * 0x10001u; tC_B = 0; }
          else make_terms<true>(fC, e[4], e[7], nC, mC, tC_RG, tC_B);
        }
        const uint32_t err = pixel_error(tA_RG + tB_RG + tC_RG, tA_B + tB_B + tC_B, loRG, hiRG, pxBlo, pxB);
        uint32_t off = e[2];
        if (__builtin_amdgcn_ballot_w64(err > p.maxPixel32) == 0ull)
        {
          const uint32_t be = wave_sum(err);
          if (be < p.blockLimit)
          {
            off = e[1];
            if (!(e[0] & 0x20u) || be < minBe) { bestA = a; bestB = e[3]; bestC = e[4]; minBe = be; }
          }
        }
        mask = off >> 24;
        e = sload8(p.table, off & 0xFFFFFFu);
      }
      if (lane == 0) p.shifts[blk] = bestA | (bestB << 8) | (bestC << 16);
    }
  }

  // ---- mode 1: four blocks per wave, a quarter of a block per step, early exit ---------------------------------------------------------
  constexpr uint32_t kBig = 1u << 27; // a pixel over its limit counts as this much: the quarter's sum then fails the block test by itself (16 x 2^27 < 2^32)

  __global__ __launch_bounds__(256, 6) void k_acc_quarter(const Params p, const uint32_t blocksPerWave)
  {
    __shared__ uint2 sPix[4][4][64]; // [wave][quarter][pixel]: (px, fac)
    const int lane = lane_id(), l = lane & 15, q = lane >> 4, wv = threadIdx.x >> 6;
    const uint32_t wave = blockIdx.x * 4u + (uint32_t)wv;
    const uint32_t first = wave * blocksPerWave, last = min(first + blocksPerWave, p.blocks);
    uint32_t next = first; // the wave's next unassigned block (uniform)
    // per quarter (replicated over its 16 lanes)
    uint32_t blk = 0;
    bool valid = false, need = true; // need: this quarter has to draw a block
    int nA[3] = { 0, 0, 0 }, nB[3] = { 0, 0, 0 }, nC[3] = { 0, 0, 0 }, cA[3] = { 0, 0, 0 }, cB[3] = { 0, 0, 0 }, cC[3] = { 0, 0, 0 };
    uint32_t e0 = 0, ePass = 0, eFail = 0, sB = 0, sC = 0, mulA = 1, mulB = 1, mulC = 1;
    uint32_t g = 0, acc = 0, best = 0, minBe = 0xFFFFFFFFu;
    for (;;)
    {
      // ---- quarters that need a block draw one (in quarter order), load its pixels and constants, start at state 0 ----
      const uint64_t needM = __builtin_amdgcn_ballot_w64(need && l == 0);
      if (needM)
      {
        const uint32_t rank = (uint32_t)__builtin_popcountll(needM & ((1ull << (q * 16)) - 1ull)); // quarters in front of this one that also draw
        if (need)
        {
          blk = next + rank;
          valid = blk < last;
          need = false;
          g = 0; acc = 0; best = 0; minBe = 0xFFFFFFFFu;
          if (valid)
          {
#pragma unroll
            for (int k = 0; k < 4; k++) sPix[wv][q][l + 16 * k] = make_uint2(p.px[(size_t)blk * 64 + l + 16 * k], p.fac[(size_t)blk * 64 + l + 16 * k]);
            int16_t r[24];
#pragma unroll
            for (int i = 0; i < 24; i++) r[i] = p.rec[(size_t)blk * 24 + i];
            record_consts(r, nA, nB, nC, cA, cB, cC);
            const uint4 x0 = *reinterpret_cast<const uint4 *>(p.table), x1 = *(reinterpret_cast<const uint4 *>(p.table) + 1);
            e0 = x0.x; ePass = x0.y; eFail = x0.z; sB = x0.w; sC = x1.x; mulA = x1.y; mulB = x1.z; mulC = x1.w;
          }
        }
        next += (uint32_t)__builtin_popcountll(needM);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (__builtin_amdgcn_ballot_w64(valid) == 0ull) break;

      // ---- one step: the quarter's next 16 pixels under its current triple ----
      const uint2 pf = sPix[wv][q][g * 16u + (uint32_t)l];
      const uint32_t px = pf.x, fac = pf.y;
      const uint32_t fA = fac & 0xFF, fB = (fac >> 8) & 0xFF, fC = fac >> 16;
      int mA[3];
#pragma unroll
      for (int c = 0; c < 3; c++) mA[c] = cA[c] + (int)(((px >> (8 * c)) & 0xFF) << 8);
      const uint32_t R = px & 0xFF, G = (px >> 8) & 0xFF;
      const uint32_t hiRG = (R + 0x8000u) | ((G + 0x8000u) << 16), loRG = (R + 0x8000u - 255u) | ((G + 0x8000u - 255u) << 16);
      const int pxB = (int)((px >> 16) & 0xFF), pxBlo = pxB - 255;
      uint32_t tA_RG, tB_RG, tC_RG;
      int tA_B, tB_B, tC_B;
      make_terms(fA, e0, mulA, nA, mA, tA_RG, tA_B);
      make_terms(fB, sB, mulB, nB, cB, tB_RG, tB_B);
      make_terms(fC, sC, mulC, nC, cC, tC_RG, tC_B);
      if (sB > 7u) { tB_RG = (uint32_t)term_bias(1) * 0x10001u; tB_B = 0; }
      if (sC > 7u) { tC_RG = (uint32_t)term_bias(2) * 0x10001u; tC_B = 0; }
      uint32_t err = pixel_error(tA_RG + tB_RG + tC_RG, tA_B + tB_B + tC_B, loRG, hiRG, pxBlo, pxB);
      err = err > p.maxPixel32 ? kBig : err;
      acc += row_sum(err);
      const bool fail = acc >= p.blockLimit;
      const bool done = valid && (fail || g == 3u);
      const bool pass = done && !fail;
      if (pass && (!(e0 & 0x20u) || acc < minBe)) { best = (e0 & 31u) | (sB << 8) | (sC << 16); minBe = acc; }
      g = (done || !valid) ? 0u : g + 1u;
      const uint32_t off = (pass ? ePass : eFail) & 0xFFFFFFu;
      acc = done ? 0u : acc;
      // ---- quarters whose trial ended fetch their next state; a final state ends the block ----
      if (__builtin_amdgcn_ballot_w64(done) != 0ull)
      {
        if (done)
        {
          const uint4 *ep = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(p.table) + off);
          const uint4 x0 = ep[0], x1 = ep[1];
          e0 = x0.x; ePass = x0.y; eFail = x0.z; sB = x0.w; sC = x1.x; mulA = x1.y; mulB = x1.z; mulC = x1.w;
          if (e0 >> 31)
          {
            if (l == 0) p.shifts[blk] = best;
            need = true;
            valid = false;
          }
        }
      }
    }
  }
}

extern "C"
{
  // expands the compact automaton the way the library does (limg_hip_api.hip ensure_accurate_table) into `out` (host, LIMG_SEARCH_ACC_STATES * 8 dwords)
  int acc_proto_states(void) { return LIMG_SEARCH_ACC_STATES; }
  void acc_proto_table(uint32_t *wide)
  {
    static const uint32_t compact[LIMG_SEARCH_ACC_STATES][2] = LIMG_SEARCH_ACC_TABLE_INIT;
    static const uint32_t mul[9] = { 1, 2, 4, 8, 17, 36, 85, 255, 256 };
    for (size_t i = 0; i < (size_t)LIMG_SEARCH_ACC_STATES; i++)
    {
      const uint32_t w0 = compact[i][0], w1 = compact[i][1];
      uint32_t *e = &wide[i * 8];
      for (int k = 0; k < 8; k++) e[k] = 0;
      if (w0 >> 31) { e[0] = 1u << 31; continue; }
      const uint32_t a = w0 & 15u, b = (w0 >> 4) & 15u, cc = (w0 >> 8) & 15u;
      e[0] = a | ((w0 & 0x1000u) ? 0x20u : 0u);
      e[1] = (w1 & 0xFFFFu) * 32u; e[2] = (w1 >> 16) * 32u;
      e[3] = b; e[4] = cc; e[5] = mul[a]; e[6] = mul[b]; e[7] = mul[cc];
    }
    for (size_t i = 0; i < (size_t)LIMG_SEARCH_ACC_STATES; i++)
    {
      uint32_t *e = &wide[i * 8];
      if (e[0] >> 31) continue;
      for (int k = 1; k <= 2; k++)
      {
        const uint32_t *n = &wide[(e[k] / 32u) * 8];
        uint32_t mask = 0;
        if (!(n[0] >> 31)) mask = ((n[0] & 31u) != (e[0] & 31u) ? 1u : 0u) | (n[3] != e[3] ? 2u : 0u) | (n[4] != e[4] ? 4u : 0u);
        e[k] |= mask << 24;
      }
    }
  }

  // mode 0 / 1 on DEVICE pointers; returns 0 or a hipError_t
  int acc_proto_run(int mode, const uint32_t *px, const uint32_t *fac, const int16_t *rec, const uint32_t *table, uint32_t *shifts, uint32_t blocks, uint32_t maxPixel32,
                    uint32_t blockLimit, int workgroups, void *stream)
  {
    Params p = { px, fac, rec, reinterpret_cast<const Entry *>(table), shifts, blocks, maxPixel32, blockLimit };
    if (mode == 0) hipLaunchKernelGGL(k_acc_lane_pixel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, p);
    else
    {
      const uint32_t waves = (uint32_t)workgroups * 4u, per = (blocks + waves - 1u) / waves;
      hipLaunchKernelGGL(k_acc_quarter, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, p, per);
    }
    return (int)hipGetLastError();
  }
}

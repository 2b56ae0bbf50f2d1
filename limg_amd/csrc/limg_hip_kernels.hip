// limg_hip_kernels.hip -- gfx950 kernels of the limg encode hot path.
//
// Path (reference file:line, all relative to the upstream repository):
//   E step (fit_search_strip)   : block gather src/limg.cpp:1899-1905, channel sums :466-497, direction fit + extrema
//                                 src/limg_factorization.h:578-794 (4 ch) / :382-576 (3 ch), colour-error state
//                                 src/limg_internal.h:426-452, per-pixel factors src/limg_factorization.h:98-197, shift search
//                                 src/limg_bit_crush.h:331-392 + :502-666 (accurate mode :668-830) on top of the trial
//                                 src/limg_bit_crush_simd.h:311-810
//   F step (dither_store_strip) : dither src/limg.cpp:824-879 / :799-822 (noise bytes from the context's table), plane stores
//                                 :2004-2093, integer decode src/limg_decode.h:36-236
//   chain position              : the dither chain order of src/limg.cpp:1893,1951-1958 = exclusive prefix of the per-block
//                                 dither-call counts (decoupled look-back in the persistent kernel, k_strip_scan in the split path)
//
// Kernels: k_encode_persistent (one launch per image -- or per list of images of one shape, limg_hip_encode3d_batch_device: the tickets
// then run through the strips of image 0, image 1, ... --: every workgroup loops over work strips, E step of a new strip then the
// F step of the strip it fitted one iteration earlier) and the three-launch split path k_fit_search / k_strip_scan /
// k_dither_store (images with partial edge blocks, whose chain has to be walked on the host; `_perf` mode; A/B testing).
// For images made of whole blocks the float stage (channel sums, direction fit, extrema, record) runs before them as its own
// kernel with one lane per block (k_fit_tpb, limg_hip_fit_tpb.hip; template parameter PREFIT here): the E step then starts from
// the records.  The float-stage code in this file is the lane == pixel form that images with partial edge blocks keep.
//
// Work decomposition: one 256-thread workgroup owns a "work strip" of 32 adjacent 8x8 image blocks (256 x 8 pixels): its
// eight 1 KiB pixel rows are read with 16-byte-per-lane loads into LDS, each of the 4 waves then owns 8 blocks and works
// with lane == pixel (wave64 == 64 pixels == one block) in the E step; per-block control flow (the shift search) is wave-uniform.
// The F step's per-pixel part works with lane == (block of the wave's 8, block row), 8 pixels per lane (phase_f_rows).
//
// Float-stage numerics are those of the reference's SSE4.1 path executed strictly (see DESIGN.md "numerics"):
//   * DPPS summation order (x0y0 + x1y1) + (x2y2 + x3y3), no FMA contraction (built with -ffp-contract=off);
//   * RSQRTPS through the captured 2048-entry table (limg_rsqrt_x86_table.h), read through the vector L1;
//   * the three direction accumulations run in *pixel order*: the per-pixel unit vectors of a batch of 4 blocks are parked in
//     LDS and 16 lanes (4 blocks x 4 channels) each walk one serial 64-term chain -- ~2 instructions per term for 16
//     chains at once instead of a 64-step dependent chain per block;
//   * correctly rounded division (hipcc default), once per batch and lane-parallel; round-to-nearest-even conversions.
#include "limg_hip_device.h"
#include "limg_search_table.h"

#include <type_traits>

namespace limg_hip
{
  namespace
  {
#ifndef LIMG_PRIO_E
#define LIMG_PRIO_E 2
#endif
    constexpr int kThreads = 256;
    constexpr int kWaves = 4;
    constexpr int kBlocksPerWave = 8;
    constexpr int kRowDw = 264; // LDS pixel-row stride in dwords: 256 px + 8 pad => bank = (8*row + x) mod 32, conflict-free per 32-lane half
    constexpr int kVDw = 260;   // per-block stride of the parked contributions: 64 px * 4 ch + 4 pad => the (block, channel) walkers hit 32 distinct banks

    // decision automaton of the default shift search (tools/make_search_table.py); read with scalar loads
    struct __attribute__((aligned(32))) SearchEntry { uint32_t w[8]; };
    __constant__ SearchEntry d_search_tab[LIMG_SEARCH_STATES] = LIMG_SEARCH_TABLE_INIT;

    // ---- a9, packed form ------------------------------------------------------------------------------------------------
    // Same integers as `trial` above, arranged for gfx950's packed 16-bit VALU:
    //  * per factor X the three RGB terms  tXc = (decX * nX[c] + (minX[c] << 8) + 128) >> 8  are kept between trials (R,G packed in one VGPR, B in another) and
    //    only recomputed when that factor's shift changes;
    //  * they are kept NEGATED: -floor(x / 256) == floor((255 - x) / 256), so (d * -n + (255 - m)) >> 8 is minus the term at the same cost, and factor A's
    //    additive constant also carries the pixel (<< 8, per lane).  The three cached values of a channel then sum to  px - estimate  directly: no subtraction in
    //    the trial;
    //  * px - clamp(S, 0, 255) == clamp(px - S, px - 255, px), so the clamp and the difference are one max and one min against per-pixel bounds prepared once per
    //    block;
    //  * the R and G halves of a packed term carry a bias (A 0x3000, B 0x3000, C 0x2000, folded into the additive constants) that keeps every half a positive
    //    16-bit number -- one plain 32-bit add3 then adds the halves independently -- and the biases sum to 0x8000: the sum is the difference in OFFSET BINARY, which
    //    unsigned v_pk_max / v_pk_min clamp correctly against bounds biased the same way, and whose square modulo 2^16 is the square of the difference itself
    //    ((e + 0x8000)^2 = e^2 + 0x10000 e + 2^30, |e| <= 255).  So the bias is never removed;
    //  * the weighted squared error is one v_dot2_u32_u16, one select and one shift-add.
    // Valid while every term stays inside (-0x2000, 0x2000): a term is (d * n + (min << 8) + 128) >> 8 with d <= 255 and n = max - min, so
    // |term| <= |min| + |n| + 1 <= 3 L + 1 when every record value is at most L in magnitude: L = p.recordLimit = 2700 (3 * 2700 + 1 = 8101 < 8192).  Then every
    // biased half lies in (0, 0x5100) and three of them sum to less than 65536 (no carry between the halves or out of the register).  A fit of byte pixels cannot
    // get near it (|A| <= 765, |B| <= 1020, |C| <= 2040); phase E falls back to the generic 32-bit form otherwise.
    typedef short short2_t __attribute__((ext_vector_type(2)));
    typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ constexpr int term_bias(int factor) { return factor == 2 ? 0x2000 : 0x3000; } // sum over the factors == 0x8000
    // additive constant of factor f, channel c, for a record minimum `lo`: negated, rounding constant reflected, RG halves biased
    __device__ __forceinline__ int term_const(int f, int c, int lo) { return 255 - ((lo << 8) + 128) + (c < 2 ? (term_bias(f) << 8) : 0); }

    struct TrialState
    {
      // per pixel, fixed for the block
      uint32_t fA, fB, fC;
      uint32_t loRG, hiRG; // (R - 255 + 0x8000) | (G - 255 + 0x8000) << 16 and (R + 0x8000) | (G + 0x8000) << 16
      int pxB, pxBlo;
      // record view: n* = -(max - min) (wave-uniform), m* = term_const(...) (wave-uniform for B and C; factor A's also carry the pixel's channel << 8, per lane)
      int nA[3], nB[3], nC[3];
      int mA[3], mB[3], mC[3];
      // cached terms and the shifts they were built for
      uint32_t tA_RG, tB_RG, tC_RG;
      int tA_B, tB_B, tC_B;
      uint32_t cA, cB, cC;
    };

    __device__ __forceinline__ void make_terms(const uint32_t f, const uint32_t s, const uint32_t mul, const int n[3], const int m[3], uint32_t &tRG, int &tB)
    {
      const int d = (int)mul_u24_uniform(f >> (s & 31u), mul); // mul == shift_mul(s); <= 255 * 256; shift and multiplier are wave-uniform in the packed trial
      const int t0 = mad_i24(d, n[0], m[0]), t1 = mad_i24(d, n[1], m[1]), t2 = mad_i24(d, n[2], m[2]);
      tRG = __builtin_amdgcn_perm((uint32_t)t1, (uint32_t)t0, 0x06050201u); // ((t1 >> 8) & 0xFFFF) << 16 | ((t0 >> 8) & 0xFFFF)
      tB = t2 >> 8;
    }

    // the three factors' cached terms, each rebuilt on demand (shift 8: f >> 8 == 0 => term == minA, as upstream; for B and C upstream zeroes min too,
    // src/limg_bit_crush_simd.h:593-609)
    __device__ __forceinline__ void rebuild_A(TrialState &t, const uint32_t sA, const uint32_t mul) { make_terms(t.fA, sA, mul, t.nA, t.mA, t.tA_RG, t.tA_B); t.cA = sA; }
    __device__ __forceinline__ void rebuild_B(TrialState &t, const uint32_t sB, const uint32_t mul)
    {
      if (sB > 7) { t.tB_RG = (uint32_t)term_bias(1) * 0x10001u; t.tB_B = 0; }
      else make_terms(t.fB, sB, mul, t.nB, t.mB, t.tB_RG, t.tB_B);
      t.cB = sB;
    }
    __device__ __forceinline__ void rebuild_C(TrialState &t, const uint32_t sC, const uint32_t mul)
    {
      if (sC > 7) { t.tC_RG = (uint32_t)term_bias(2) * 0x10001u; t.tC_B = 0; }
      else make_terms(t.fC, sC, mul, t.nC, t.mC, t.tC_RG, t.tC_B);
      t.cC = sC;
    }

    // the trial proper on the cached terms: clamp, differences, weighted squared error per pixel
    template <bool FULL>
    __device__ __forceinline__ uint32_t trial_pixel_error(const TrialState &t, const bool active)
    {
      const uint32_t dRG = t.tA_RG + t.tB_RG + t.tC_RG; // (R - estimate + 0x8000) | (G - estimate + 0x8000) << 16: no carry crosses the halves
      const int dBraw = t.tA_B + t.tB_B + t.tC_B;        // B - estimate
      ushort2_t eu = __builtin_bit_cast(ushort2_t, dRG);
      eu = __builtin_elementwise_max(eu, __builtin_bit_cast(ushort2_t, t.loRG));
      eu = __builtin_elementwise_min(eu, __builtin_bit_cast(ushort2_t, t.hiRG));
      int dB = med3_i32(dBraw, t.pxBlo, t.pxB); // clamp(px - S, px - 255, px)
      const ushort2_t sq = eu * eu; // (d + 0x8000)^2 mod 2^16 == d^2 <= 65025
      const uint32_t sqB = (uint32_t)mul_i24(dB, dB);
      // weights (R, G, B) = (2, 4, 3) while dR^2 < 0x4000, else (3, 4, 2)  ==  2 * (dR^2 + 2 dG^2 + dB^2) + (dB^2 or dR^2): one dot product with constant weights,
      // one select (the red square is picked out of the packed pair by the select's operand modifier), one shift-add
      const bool low_red = sq.x < 0x4000;
      const uint32_t half = __builtin_amdgcn_udot2(sq, __builtin_bit_cast(ushort2_t, 0x00020001u), sqB, false);
      const uint32_t extra = low_red ? sqB : (__builtin_bit_cast(uint32_t, sq) & 0xFFFFu);
      uint32_t err = (half << 1) + extra;
      if (!FULL) err = active ? err : 0u;
      return err;
    }

    // a10 + a11 as a table-driven automaton: one trial loop; the outcome of a trial picks the byte offset of the next state's 32-byte entry, which one scalar load
    // fetches.  The scalar side of the loop is kept minimal -- the scalar unit (one per CU) is a co-bottleneck of this kernel: 8 extra scalar instructions per
    // trial cost 10 % (measured) -- so an entry says WHICH factors its triple changes against its predecessor's (the automaton is a tree: no compares against
    // cached shifts), holds byte offsets (no shifts) and the re-expansion multipliers, and the table's base address stays in SGPRs.  The load is NOT issued
    // ahead for both outcomes: the other waves of the SIMD cover its latency, and the two address computations, the second load and the selects between two
    // prefetched entries were scalar instructions too (measured equal, with less code).
    typedef unsigned int uint8s_t __attribute__((ext_vector_type(8)));
    __device__ __forceinline__ uint8s_t sload8(const SearchEntry *base, uint32_t byteOffset)
    {
      uint8s_t v;
      asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(base), "s"(byteOffset) : "memory");
      return v;
    }

    template <bool FULL>
    __device__ __forceinline__ void search_fast_automaton(TrialState &t, const bool active, const uint32_t maxPixel32, const uint32_t blockLimit, uint32_t shift[3])
    {
      const SearchEntry *tab = d_search_tab;
      asm volatile("" : "+s"(tab)); // opaque: otherwise the address is rematerialised (s_getpc + 2 adds) in every iteration
      // entry 0 as immediates (the opaque base above would make reading it a memory round trip per block)
      // its three factors are built here, unconditionally and with immediate operands (the loop then starts with nothing to rebuild): the cached terms need no
      // initial value at all
      constexpr uint32_t root[8] = LIMG_SEARCH_ROOT;
      rebuild_A(t, root[0] & 31u, root[5]);
      rebuild_B(t, root[3], root[6]);
      rebuild_C(t, root[4], root[7]);
      uint8s_t e = { root[0] & ~0xE0u, root[1], root[2], root[3], root[4], root[5], root[6], root[7] };
      while (!(e[0] >> 31))
      { // every field sits in an SGPR of its own: no extraction.  (e[0] & 31 is the shift amount as v_lshrrev_b32 reads it -- the mask costs nothing)
        if (e[0] & 0x20u) rebuild_A(t, e[0] & 31u, e[5]);
        if (e[0] & 0x40u) rebuild_B(t, e[3], e[6]);
        if (e[0] & 0x80u) rebuild_C(t, e[4], e[7]);
        const uint32_t err = trial_pixel_error<FULL>(t, active);
        // two tails on purpose: a pixel failure (the common way to fail) needs no outcome flag, no select and no block sum
        uint32_t off;
        if (__builtin_amdgcn_ballot_w64(err > maxPixel32) != 0ull) off = e[2];
        else off = (wave_sum(err) < blockLimit) ? e[1] : e[2]; // be * 16 < maxBlock * n, see phase E
        e = sload8(tab, off);
      }
      shift[0] = e[0] & 31u; shift[1] = e[3]; shift[2] = e[4];
    }

    // a12 as an automaton (limg_search_table_accurate.h, a DAG of ~19 k states in global memory, expanded by the context): which trials the accurate search runs
    // depends on pass / fail outcomes only, so its three nested scalar loops -- which, not the trials, were the cost of this mode -- become one table walk.  What
    // the block errors decide stays here: a passing phase-1 trial becomes the result; a passing phase-2 trial only if its error is below the best so far
    // (src/limg_bit_crush.h:774-826; `have` is always set by then).  A state has several predecessors, so the factors to rebuild come from comparing with the cached
    // shifts (t.cA..cC).
    template <bool FULL>
    __device__ __forceinline__ void search_accurate_automaton(TrialState &t, const bool active, const uint32_t maxPixel32, const uint32_t blockLimit, const uint32_t *table,
                                                              uint32_t shift[3])
    {
      const SearchEntry *tab = reinterpret_cast<const SearchEntry *>(table);
      uint32_t bestA = 0, bestB = 0, bestC = 0, minBe = 0xFFFFFFFFu;
      // Measured and NOT adopted (LIMG_ACC_CACHE=1 builds it): the accurate search walks the shift cube row by row -- c innermost
      // (src/limg_bit_crush.h:700-760) -- so factor C's shift changes with nearly every one of its ~70 trials per block while it only takes nine values; its terms for
      // the shifts 0..7 can be built once per block and picked per trial out of a 16-register vector with the wave-uniform shift as the index (VGPR index mode:
      // s_set_gpr_idx_on, two v_mov, s_set_gpr_idx_off; one 16-wide vector because LLVM expands a dynamic extract of up to 8 elements into compares and selects).
      // At equal occupancy that is 2 % faster (4.39 vs 4.49 ms at 5 workgroups per CU), but its 16 registers cost the sixth workgroup per CU, which is worth 7.5 %
      // (4.16 ms without the cache at 6).
#ifndef LIMG_ACC_CACHE
#define LIMG_ACC_CACHE 0
#endif
      typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
      u32x16_t cT;
      if (LIMG_ACC_CACHE)
      {
        constexpr uint32_t mulOf[8] = { 1, 2, 4, 8, 17, 36, 85, 255 }; // (1 << s) + decode_bias(s)
#pragma unroll
        for (int sft = 0; sft < 8; sft++)
        {
          uint32_t rg; int bl;
          make_terms(t.fC, (uint32_t)sft, mulOf[sft], t.nC, t.mC, rg, bl);
          cT[sft] = rg; cT[8 + sft] = (uint32_t)bl;
        }
      }
      { // the first triple is the fast search's: built unconditionally, so that the cached terms need no initial value
        constexpr uint32_t root[8] = LIMG_SEARCH_ROOT;
        rebuild_A(t, root[0] & 31u, root[5]);
        rebuild_B(t, root[3], root[6]);
        if (LIMG_ACC_CACHE) { t.tC_RG = cT[root[4]]; t.tC_B = (int)cT[8 + root[4]]; t.cC = root[4]; }
        else rebuild_C(t, root[4], root[7]);
      }
      uint8s_t e = sload8(tab, 0u);
      // which factors state 0's triple changes against the root triple built above (every later edge carries its mask in bits 24..26 of the successor offset)
      constexpr uint32_t rootT[8] = LIMG_SEARCH_ROOT;
      uint32_t mask = ((e[0] & 31u) != (rootT[0] & 31u) ? 1u : 0u) | (e[3] != rootT[3] ? 2u : 0u) | (e[4] != rootT[4] ? 4u : 0u);
      while (!(e[0] >> 31))
      {
        const uint32_t a = e[0] & 31u;
        if (mask & 1u) rebuild_A(t, a, e[5]);
        if (mask & 2u) rebuild_B(t, e[3], e[6]);
        if (LIMG_ACC_CACHE)
        {
          if (mask & 4u)
          {
            const uint32_t c = e[4];
            if (c > 7) { t.tC_RG = (uint32_t)term_bias(2) * 0x10001u; t.tC_B = 0; }
            else { t.tC_RG = cT[c]; t.tC_B = (int)cT[8 + c]; }
          }
        }
        else if (mask & 4u) rebuild_C(t, e[4], e[7]);
        const uint32_t err = trial_pixel_error<FULL>(t, active);
        uint32_t off = e[2];
        if (__builtin_amdgcn_ballot_w64(err > maxPixel32) == 0ull)
        {
          const uint32_t be = wave_sum(err);
          if (be < blockLimit) // be * 16 < maxBlock * n, see phase E
          {
            off = e[1];
            if (!(e[0] & 0x20u) || be < minBe) { bestA = a; bestB = e[3]; bestC = e[4]; minBe = be; }
          }
        }
        mask = off >> 24;
        e = sload8(tab, off & 0xFFFFFFu);
      }
      shift[0] = bestA; shift[1] = bestB; shift[2] = bestC;
    }

    // generic-path search (see phase E): real function, rarely if ever executed
    __device__ __attribute__((noinline)) uint32_t search_generic(uint32_t px, uint32_t fA, uint32_t fB, uint32_t fC, const int16_t *rec /* LDS */, bool active,
                                                                 uint32_t maxPixel32, uint64_t maxBlockN, bool fast)
    {
      RecU r;
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
        const int loA = rec[c], hiA = rec[4 + c], loB = rec[8 + c], hiB = rec[12 + c], loC = rec[16 + c], hiC = rec[20 + c];
        r.nA[c] = sgpr(hiA - loA); r.nB[c] = sgpr(hiB - loB); r.nC[c] = sgpr(hiC - loC);
        r.mA[c] = sgpr((int)(((uint32_t)loA << 8) + 128u)); r.mB[c] = sgpr((int)(((uint32_t)loB << 8) + 128u)); r.mC[c] = sgpr((int)(((uint32_t)loC << 8) + 128u));
      }
      uint32_t shift[3] = { 0, 0, 0 };
      auto T = [&](uint32_t a, uint32_t bb, uint32_t c, uint32_t &be2) -> bool { return trial(px, fA, fB, fC, r, a, bb, c, active, maxPixel32, maxBlockN, be2); };
      if (fast) search_fast(T, shift);
      else search_accurate(T, shift);
      return shift[0] | (shift[1] << 8) | (shift[2] << 16);
    }

    // =====================================================================================================================
    // phase F / kernel 3: dither (a13), plane stores (a15), decode (a16) for one work strip
    // =====================================================================================================================

    // The strip's factor bytes in LDS (written by the E step lane == pixel, read by the F step lane == (block, row) 8 bytes at a time): [3 planes][8 rows] of
    // 256 bytes at a row stride of 320 -- 80 dwords = 16 mod 64 banks, so the 32 lanes (4 rows x 8 blocks) that a ds_read_b64 serves at a time hit 64 distinct banks
    // (at a stride of 256 all rows of a block share two banks).
    constexpr int kFacRow = 320, kFacPlane = 8 * kFacRow, kFacBytes = 3 * kFacPlane;

    // LDS areas of phase F.  In the fused kernel they overlay the (then dead) parked-contribution area of k_fit_search.
    struct StripLds
    {
      uint8_t *fac;    // [3][8][kFacRow]  pre-dither factor bytes of the strip (plane-row layout)
      uint32_t *dec;   // [4 waves][8 rows][64]  decoded pixels
      uint8_t *out;    // == fac: a lane's output byte replaces the pre-dither byte it has just read (same index)
      uint32_t *cst;   // [7][32][4]  per-block constants of the 7 block-uniform planes, each four times over: a 16-byte store takes it from one ds_read_b128
      int32_t *nm;     // [32 blocks][2][3][4]  effective integer normals / additive constants of the decode
      uint32_t *shift; // [32]  shift words
      uint32_t *first; // [32]  first dither-call index of each block
      uint32_t *flags; // [32]  bit 0: some record value beyond p.recordLimit (generic 32-bit decode), bit 1: the alpha lane varies inside the block; bits 8..15: its value when it does not
      const int16_t *rec; // record of block sb at rec + sb * recStride
      int recStride;
    };
    constexpr int kPhaseFBytes = kFacBytes + 8192 + 3584 + 3072 + 128 + 128 + 128; // the output factor bytes replace the pre-dither ones in place

    __device__ __forceinline__ StripLds carve_phase_f(uint8_t *base, const int16_t *rec, int recStride)
    {
      StripLds L;
      L.fac = base;
      L.dec = reinterpret_cast<uint32_t *>(base + kFacBytes);
      L.out = base;
      L.cst = reinterpret_cast<uint32_t *>(base + kFacBytes + 8192);
      L.nm = reinterpret_cast<int32_t *>(base + kFacBytes + 8192 + 3584);
      L.shift = reinterpret_cast<uint32_t *>(base + kFacBytes + 8192 + 3584 + 3072);
      L.first = L.shift + 32;
      L.flags = L.shift + 64;
      L.rec = rec; L.recStride = recStride;
      return L;
    }

    // Per-wave preparation from records + shifts (lane-parallel over the wave's 8 blocks): the 7 block-uniform plane values
    // (src/limg.cpp:2006-2036) and the effective decode constants (src/limg_decode.h:139-196 / :40-101).
    // The decode's additive constants of the R and G lanes carry the packed form's biases (term_bias: 0x3000, 0x3000, 0x2000 -- they sum to 0x8000): see phase_f_rows.
    __device__ __forceinline__ constexpr int decode_bias(int factor, int c) { return c < 2 ? (term_bias(factor) << 8) : 0; }

    template <int CH>
    __device__ __forceinline__ void phase_f_prepare(const StripLds &L, int lane, int wave, int recordLimit)
    {
      if (lane < 56)
      {
        const int b = lane / 7, k = lane - b * 7, sb = wave * kBlocksPerWave + b;
        const int16_t *rec = L.rec + sb * L.recStride;
        uint32_t v;
        if (k == 0)
        {
          const uint32_t w = L.shift[sb];
          const uint32_t pat[3] = { (w & 0xFF), ((w >> 8) & 0xFF), ((w >> 16) & 0xFF) };
          // bit_to_pattern {0,0x22,...,0xEE,0xFF}: 0x22 * s, except s == 8 -> 0xFF
          const uint32_t pa = pat[0] == 8 ? 0xFFu : pat[0] * 0x22u, pb = pat[1] == 8 ? 0xFFu : pat[1] * 0x22u, pc = pat[2] == 8 ? 0xFFu : pat[2] * 0x22u;
          v = 0xFF000000u | (pa << 16) | (pb << 8) | pc;
        }
        else
        {
          v = 0;
#pragma unroll
          for (int c = 0; c < CH; c++)
          {
            int q = rec[(k - 1) * 4 + c] + (k >= 3 ? 0x80 : 0);
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            v |= (uint32_t)q << (8 * c);
          }
          if (CH == 3) v |= 0xFF000000u;
        }
        reinterpret_cast<uint4 *>(L.cst)[k * kStripBlocks + sb] = make_uint4(v, v, v, v);
      }
      if (lane < kBlocksPerWave)
      { // per-block flags of the decode: whether the alpha lane is one value for the block (and which); bit 0 -- a record value beyond the packed form's range, never
        // from a fit of byte pixels -- is OR-ed in below by whichever lane meets such a value
        const int sb = wave * kBlocksPerWave + lane;
        const int16_t *rec = L.rec + sb * L.recStride;
        uint32_t fl;
        if (CH == 3) fl = 255u << 8; // src/limg_decode.h:95-97: the three 0xFFFF minima clamp to 255
        else
        {
          const bool varies = rec[7] != rec[3] || rec[15] != rec[11] || rec[23] != rec[19]; // an alpha normal (max - min) is never zeroed, not even at shift 8
          int a = rec[3] + rec[11] + rec[19]; // ((m << 8) + 128) >> 8 == m for each of the three terms
          a = a < 0 ? 0 : (a > 255 ? 255 : a);
          fl = varies ? 2u : ((uint32_t)a << 8);
        }
        L.flags[sb] = fl;
      }
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 2; r++)
      {
        const int idx = r * 64 + lane;
        if (idx < 96)
        {
          const int b = idx / 12, fc = idx - b * 12, f = fc >> 2, c = fc & 3, sb = wave * kBlocksPerWave + b;
          const int16_t *rec = L.rec + sb * L.recStride;
          const uint32_t sh = (L.shift[sb] >> (8 * f)) & 0xFF;
          const int lo = rec[f * 8 + c], hi = rec[f * 8 + 4 + c];
          if ((uint32_t)(lo + recordLimit) > 2u * (uint32_t)recordLimit || (uint32_t)(hi + recordLimit) > 2u * (uint32_t)recordLimit) atomicOr(&L.flags[sb], 1u); // |value| > limit
          int n = hi - lo, m = lo;
          if (c < 3)
          {
            if (sh > 7) { n = 0; if (f > 0) m = 0; }
          }
          else if (CH == 3) { n = 0; m = 0xFFFF; }
          int *dst = L.nm + sb * 24;
          dst[f * 4 + c] = n;
          dst[12 + f * 4 + c] = (int)(((uint32_t)m << 8) + 128u + (uint32_t)(c < 2 ? (f == 2 ? 0x200000 : 0x300000) : 0)); // + decode_bias(f, c)
        }
      }
    }

    // the 7 block-uniform planes, straight from registers: 16 bytes per lane = four rows of 256 contiguous bytes (8 blocks x 8 px) per store instruction where
    // the rows allow it (p.vecPlanes: width a multiple of 4, 16-byte aligned planes), 4 bytes per lane = one row per instruction otherwise
    // The 35 bytes per pixel of output planes are written once and never read by this library: stored NON-TEMPORALLY (global_store ... nt) they do not push the
    // data the kernels DO come back to out of the L2 -- a strip's parked results (8 KiB written by its E step, read by its F step), the records and k_fit_tpb's
    // rows.  -DLIMG_PLANE_STORES_TEMPORAL builds the plain stores (A/B, tools/r05/ab_nt_stores.sh, same box: 4096^2 gradient 0.331 -> 0.285 ms, config 4 72.6 ->
    // 74.9 Gpx/s, 8192^2 photo-noise 1.374 -> 1.360 ms; the HBM byte counters do not move -- the parked data still goes out and comes back -- the time does).
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ void plane_store16(void *dst, const uint4 &v)
    {
#ifdef LIMG_PLANE_STORES_TEMPORAL
      *reinterpret_cast<uint4 *>(dst) = v;
#else
      __builtin_nontemporal_store(u32x4_t{ v.x, v.y, v.z, v.w }, reinterpret_cast<u32x4_t *>(dst));
#endif
    }
    __device__ __forceinline__ void plane_store8(void *dst, const uint2 &v)
    {
#ifdef LIMG_PLANE_STORES_TEMPORAL
      *reinterpret_cast<uint2 *>(dst) = v;
#else
      __builtin_nontemporal_store(u32x2_t{ v.x, v.y }, reinterpret_cast<u32x2_t *>(dst));
#endif
    }

    template <class P, class IO>
    // halves: bit 0 = the strip's rows 0..3, bit 1 = rows 4..7 (the F step issues them at two different points, each beside a memory round trip of its own; the
    // one-row-per-instruction form stores everything with bit 0)
    __device__ __forceinline__ void phase_f_store_const(const P &p, const IO &io, const StripLds &L, uint32_t x0, uint32_t y0, uint32_t ry, int lane, int wave, const uint32_t halves = 3u)
    {
      const uint32_t wx0 = x0 + wave * 64;
      if (wx0 >= p.sizeX) return;
      const uint32_t ww = min(p.sizeX - wx0, 64u);
      uint32_t *planes[7] = { io.info.pShiftABCX, io.info.pColAMin, io.info.pColAMax, io.info.pColBMin, io.info.pColBMax, io.info.pColCMin, io.info.pColCMax };
      if (p.vecPlanes)
      {
        const uint32_t col = ((uint32_t)lane & 15u) * 4u, rsub = (uint32_t)lane >> 4; // 16 lanes per row, 4 rows per instruction
        const uint4 *cst = reinterpret_cast<const uint4 *>(L.cst) + wave * kBlocksPerWave + (col >> 3); // read per store: the LDS pipe has the room, registers do not
        if (col < ww)
#pragma unroll
          for (uint32_t half = 0; half < 2; half++)
          {
            const uint32_t row = half * 4 + rsub;
            if (row < ry && ((halves >> half) & 1u))
            {
              size_t g = (size_t)(y0 + row) * p.sizeX + wx0 + col;
              asm volatile("" : "+v"(g)); // one offset for the seven planes (left to itself the compiler adds its three loop-invariant parts to every plane's base separately)
#pragma unroll
              for (int k = 0; k < 7; k++) plane_store16(planes[k] + g, cst[k * kStripBlocks]);
            }
          }
        return;
      }
      if (!(halves & 1u)) return;
      uint32_t cst[7];
#pragma unroll
      for (int k = 0; k < 7; k++) cst[k] = L.cst[(k * kStripBlocks + wave * kBlocksPerWave + (lane >> 3)) * 4];
      if ((uint32_t)lane < ww)
        for (uint32_t row = 0; row < ry; row++)
        {
          const size_t g = (size_t)(y0 + row) * p.sizeX + wx0 + lane;
#pragma unroll
          for (int k = 0; k < 7; k++) planes[k][g] = cst[k];
        }
    }

    // ---- phase F for strips of whole 8x8 blocks: lane == (block j of the wave's 8, row r), 8 pixels per lane ------------------------------------------------
    // With lane == pixel (phase_f_pixels below, which strips with partial blocks keep) everything per block is scalar work -- shift fields, dither on / off
    // branches, multipliers: ~50 scalar instructions a block on a scalar unit the search already loads -- every plane goes through LDS staging to reach
    // 16-byte stores, and the decode runs unpacked.  Here a lane owns one row of one block:
    //  * its per-block values (shifts, multipliers, dither masks, decode constants) are ordinary per-lane registers: no scalar code at all;
    //  * its 8 pixels leave as two 16-byte stores (pDecoded) and three 8-byte stores (factor planes) straight from registers; the inputs are three 8-byte
    //    LDS reads (pre-dither factor bytes) and up to three 8-byte loads of the noise stream;
    //  * byte lanes are addressed by SDWA operand selects: one v_add_u32_sdwa adds byte i of the factor dword and byte i of the (pre-masked) noise dword, one
    //    v_and_b32_sdwa with dst_sel:BYTE_i inserts the crushed byte (clamped value with the dropped bits cleared == (v >> s) << s) into the output dword;
    //  * the decode (a16, src/limg_decode.h:137-236) takes the trial's packed form: per factor three 24-bit multiply-adds, the R and G terms packed into one
    //    register by one v_perm_b32 (both >> 8 included), biased so that a plain v_add3_u32 sums the halves independently (0x3000 + 0x3000 + 0x2000 = 0x8000: the sum is
    //    the estimate in offset binary, clamped by unsigned packed max / min against 0x8000 / 0x80FF, and its low byte IS the clamped estimate); the alpha lane is one
    //    value per block unless its normals are non-zero (wave-uniform test).  Valid for record values up to p.recordLimit like the trial; beyond (never from a fit
    //    of byte pixels) the wave takes the plain 32-bit form.
    template <int B> __device__ __forceinline__ uint32_t add_byte_sdwa(uint32_t a, uint32_t b)
    {
      uint32_t r;
      if (B == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(b));
      else if (B == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(b));
      else if (B == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(b));
      else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(b));
      return r;
    }
    // acc.byte[B] = (a & b) & 0xFF, the other bytes of acc kept
    template <int B> __device__ __forceinline__ void and_into_byte_sdwa(uint32_t &acc, uint32_t a, uint32_t b)
    {
      if (B == 0) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "v"(a), "v"(b));
      else if (B == 1) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "v"(a), "v"(b));
      else if (B == 2) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "v"(a), "v"(b));
      else asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(acc) : "v"(a), "v"(b));
    }

    // One factor of the lane's 8 pixels: dither, crushed output bytes (stored by the caller), and the factor's terms of the decode added to the accumulators
    // (first factor: assigned).  accRG: R and G terms packed and biased, accB: B terms, accA: alpha terms (only when the wave has a block whose alpha varies).
    template <int K, bool ALPHA>
    __device__ __forceinline__ void rows_factor(const uint2 fac, const uint2 nzm, const uint32_t negHalf, const uint32_t shr, const uint32_t mul, const uint32_t keep, const int4 n,
                                                const int4 m, uint32_t accRG[8], int accB[8], int accA[8], uint32_t &outLo, uint32_t &outHi)
    {
      outLo = 0; outHi = 0;
      auto pixel = [&](auto BI, auto HI)
      {
        constexpr int B = decltype(BI)::value, H = decltype(HI)::value, I = H * 4 + B;
        // src/limg.cpp:824-879 per byte: v = clamp(f + (noise & ditherSize) - ditherOffset, 0, 255) >> shift; a factor that does not dither has mask 0, offset 0, shift 0
        const uint32_t t = (uint32_t)med3_i32((int)(add_byte_sdwa<B>(H ? fac.y : fac.x, H ? nzm.y : nzm.x) + negHalf), 0, 255);
        and_into_byte_sdwa<B>(H ? outHi : outLo, t, keep); // (v >> s) << s: what the factor plane holds (src/limg.cpp:2054-2062; shift 8 => 0)
        const int d = (int)mul_u24(t >> shr, mul);          // a16: dec = v * mul (the raw byte at shift 8)
        const int t0 = mad_i24(d, n.x, m.x), t1 = mad_i24(d, n.y, m.y), t2 = mad_i24(d, n.z, m.z);
        const uint32_t rg = __builtin_amdgcn_perm((uint32_t)t1, (uint32_t)t0, 0x06050201u);
        if (K == 0) { accRG[I] = rg; accB[I] = t2 >> 8; }
        else { accRG[I] += rg; accB[I] += t2 >> 8; }
        if (ALPHA)
        {
          const int ta = mad_i24(d, n.w, m.w) >> 8;
          if (K == 0) accA[I] = ta; else accA[I] += ta;
          asm volatile("" : "+v"(accA[I]));
        }
        // the accumulators are materialised here: otherwise the packing of this factor's terms sinks to their next use (the next factor's adds) and every pixel's three
        // products stay live until then
        asm volatile("" : "+v"(accRG[I]), "+v"(accB[I]));
      };
      // two pixels at a time (the scheduler would otherwise run all eight pixels' multiply-adds ahead of their packing: 24 temporaries)
      pixel(std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
      pixel(std::integral_constant<int, 1>(), std::integral_constant<int, 0>());
      __builtin_amdgcn_sched_barrier(0);
      pixel(std::integral_constant<int, 2>(), std::integral_constant<int, 0>());
      pixel(std::integral_constant<int, 3>(), std::integral_constant<int, 0>());
      __builtin_amdgcn_sched_barrier(0);
      pixel(std::integral_constant<int, 0>(), std::integral_constant<int, 1>());
      pixel(std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
      __builtin_amdgcn_sched_barrier(0);
      pixel(std::integral_constant<int, 2>(), std::integral_constant<int, 1>());
      pixel(std::integral_constant<int, 3>(), std::integral_constant<int, 1>());
      __builtin_amdgcn_sched_barrier(0);
    }

    // `between`: work that does not depend on the noise bytes (the seven uniform planes' stores: 28 of the 35 output bytes per pixel), run right after the noise
    // loads are issued -- their round trip to HBM (the table is 200 MB: no cache holds it) then runs beside those stores instead of in front of the decode
    template <int CH, class P, class IO, class BETWEEN>
    __device__ __forceinline__ void phase_f_rows(const P &p, const IO &io, const StripLds &L, const uint32_t strip, const uint32_t x0, const uint32_t y0, const int lane, const int wave,
                                                 BETWEEN &&between)
    {
      const uint32_t j = (uint32_t)lane & 7u, r = (uint32_t)lane >> 3;
      const uint32_t sb = (uint32_t)wave * kBlocksPerWave + j, bx = strip * kStripBlocks + sb;
      const bool valid = bx < p.blocksX;
      const uint32_t w = L.shift[sb], fl = L.flags[sb]; // (shift word 0 for blocks past the right edge)
      uint32_t call = L.first[sb];
      const size_t g = (size_t)(y0 + r) * p.sizeX + x0 + sb * kBlock; // the lane's 8 pixels in every plane
      // the noise bytes of the lane's row for every factor that dithers: requested first, used factor by factor
      uint2 nz[3];
#pragma unroll
      for (int k = 0; k < 3; k++)
      {
        const uint32_t s = (w >> (8 * k)) & 0xFFu;
        nz[k] = make_uint2(0u, 0u);
        if (((s - 1u) < 7u) && valid) // shifts 1..7 dither (src/limg.cpp:1951-1958)
        {
          nz[k] = *reinterpret_cast<const uint2 *>(p.noise + (size_t)min(call, p.noiseLast) * 64 + r * 8);
          call++;
        }
      }
      between();
      const uint8_t *facRow = L.fac + r * kFacRow + sb * kBlock;
      const int *nm = L.nm + sb * 24;
#ifdef LIMG_X_NOGEN
      const bool generic = false;
#else
      const bool generic = __builtin_amdgcn_ballot_w64((fl & 1u) != 0u) != 0ull;   // wave-uniform
#endif
#ifdef LIMG_X_NOALPHA
      const bool anyAlpha = false;
#else
      const bool anyAlpha = CH == 4 && __builtin_amdgcn_ballot_w64((fl & 2u) != 0u) != 0ull;
#endif
      const bool rawEscape = !p.fullPlanes && p.streamRaw; // compact stream: a factor at shift 8 keeps its raw byte (raw-escape of the container)
      // per-lane constants of factor k from its shift s: the shift the dither applies (0 unless 1..7), minus half the dither range, the re-expansion multiplier
      // (1 << s) + decode_bias(s) with decode_bias = {0,0,0,0,1,4,21,127,0} = byte s of a constant pair (selector 8: a zero sign fill), the bits the crushed byte keeps
      auto consts = [&](int k, uint32_t &negHalf, uint32_t &shr, uint32_t &mul, uint32_t &keep, uint2 &fq, uint2 &nzm)
      {
        const uint32_t s = (w >> (8 * k)) & 0xFFu;
        shr = (((s - 1u) < 7u) && valid) ? s : 0u;
        negHalf = 0u - ((1u << shr) >> 1);
        mul = (1u << s) + __builtin_amdgcn_perm(0x7F150401u, 0u, s);
        keep = (0xFFu << s) & 0xFFu;
        if (rawEscape && s == 8) keep = 0xFFu;
        fq = *reinterpret_cast<const uint2 *>(facRow + k * kFacPlane);
        const uint32_t m4 = __builtin_amdgcn_perm(0u, (1u << shr) - 1u, 0u); // noise & ditherSize for four pixels at a time: the mask's byte in all four lanes
        nzm = make_uint2(nz[k].x & m4, nz[k].y & m4);
      };
      uint8_t *planes8[3] = { io.info.pFactorsA, io.info.pFactorsB, io.info.pFactorsC };
      auto store_factor = [&](int k, uint32_t lo, uint32_t hi)
      {
        if (!valid) return;
        if (p.vecFactors8) plane_store8(planes8[k] + g, make_uint2(lo, hi));
        else
        {
#pragma unroll
          for (int i = 0; i < 4; i++) { planes8[k][g + i] = (uint8_t)(lo >> (8 * i)); planes8[k][g + 4 + i] = (uint8_t)(hi >> (8 * i)); }
        }
      };
      uint32_t px[8];
      if (!generic)
      {
        uint32_t accRG[8], lo, hi, negHalf, shr, mul, keep;
        int accB[8], accA[8];
        uint2 fq, nzm;
        // (scheduling barriers: left alone the compiler interleaves the three factors and keeps everything live at once -- 113 VGPRs, where 80 are allowed)
#define LIMG_ROWS_FACTOR(K, ALPHA)                                                                                                                         \
        consts(K, negHalf, shr, mul, keep, fq, nzm);                                                                                                      \
        rows_factor<K, ALPHA>(fq, nzm, negHalf, shr, mul, keep, *reinterpret_cast<const int4 *>(nm + 4 * K), *reinterpret_cast<const int4 *>(nm + 12 + 4 * K), accRG, accB, accA, lo, hi); \
        store_factor(K, lo, hi);                                                                                                                          \
        __builtin_amdgcn_sched_barrier(0)
        if (anyAlpha)
        { // some block of this wave has a varying alpha lane (its normals are live even at shift 8)
          LIMG_ROWS_FACTOR(0, true); LIMG_ROWS_FACTOR(1, true); LIMG_ROWS_FACTOR(2, true);
        }
        else
        {
          LIMG_ROWS_FACTOR(0, false); LIMG_ROWS_FACTOR(1, false); LIMG_ROWS_FACTOR(2, false);
        }
#undef LIMG_ROWS_FACTOR
        if (!p.fullPlanes) return;
        const uint32_t alphaConst = fl & 0xFF00u;
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
          ushort2_t e = __builtin_bit_cast(ushort2_t, accRG[i]); // estimate + 0x8000 in both halves
          e = __builtin_elementwise_max(e, __builtin_bit_cast(ushort2_t, 0x80008000u));
          e = __builtin_elementwise_min(e, __builtin_bit_cast(ushort2_t, 0x80FF80FFu));
          uint32_t ba = (uint32_t)med3_i32(accB[i], 0, 255);
          if (anyAlpha) ba |= (uint32_t)med3_i32(accA[i], 0, 255) << 8;
          else ba |= alphaConst;
          px[i] = __builtin_amdgcn_perm(ba, __builtin_bit_cast(uint32_t, e), 0x05040200u); // R = low byte of the low half, G = low byte of the high half, B, A
        }
      }
      else
      { // a record value beyond the packed form's range somewhere in this wave (never from a fit of byte pixels): any int16 record, 32-bit terms, the low 32 bits of
        // the products like PMULLD (the form of phase_f_pixels).  A rolled loop, one pixel at a time, constants re-read from LDS: this path must not set the
        // kernel's register count.
        uint32_t outLo[3] = { 0, 0, 0 }, outHi[3] = { 0, 0, 0 };
#pragma unroll 1
        for (int i = 0; i < 8; i++)
        {
          const uint32_t bsh = 8u * ((uint32_t)i & 3u);
          int d[3];
#pragma unroll
          for (int k = 0; k < 3; k++)
          {
            uint32_t negHalf, shr, mul, keep;
            uint2 fq, nzm;
            consts(k, negHalf, shr, mul, keep, fq, nzm);
            const uint32_t fb = ((i < 4 ? fq.x : fq.y) >> bsh) & 0xFFu, nb = ((i < 4 ? nzm.x : nzm.y) >> bsh) & 0xFFu;
            const uint32_t t = (uint32_t)med3_i32((int)(fb + nb + negHalf), 0, 255);
            if (i < 4) outLo[k] |= (t & keep) << bsh; else outHi[k] |= (t & keep) << bsh;
            d[k] = (int)mul_u24(t >> shr, mul);
          }
          uint32_t out = 0;
#pragma unroll
          for (int c = 0; c < 4; c++)
          {
            int est = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) est += mad_i24(d[k], nm[k * 4 + c], nm[12 + k * 4 + c] - decode_bias(k, c)) >> 8;
            out |= (uint32_t)med3_i32(est, 0, 255) << (8 * c);
          }
          if (valid && p.fullPlanes) io.info.pDecoded[g + i] = out;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) store_factor(k, outLo[k], outHi[k]);
        return;
      }
      if (!valid) return; // (strips of whole blocks: every row r < 8 exists)
      uint32_t *dst = io.info.pDecoded + g;
      if (p.vecDecoded)
      {
        plane_store16(dst, make_uint4(px[0], px[1], px[2], px[3]));
        plane_store16(dst + 4, make_uint4(px[4], px[5], px[6], px[7]));
      }
      else
      {
#pragma unroll
        for (int i = 0; i < 8; i++) dst[i] = px[i];
      }
    }

    // dither + decode of the wave's 8 blocks into the per-wave staging areas, then the per-pixel planes' stores
    template <int CH, class P, class IO>
    __device__ __forceinline__ void phase_f_pixels(const P &p, const IO &io, const StripLds &L, uint32_t strip, uint32_t x0, uint32_t y0, uint32_t ry, int lane, int wave, int tid)
    {
      uint32_t *dec = L.dec + wave * 512;
      uint8_t *out = L.out; // [3 planes][8 rows][256 px]: strip-wide rows, so that the stores below write whole 128-byte lines
      // The noise bytes are requested for a group of kNoiseGroup blocks at a time (up to 3 independent 64-byte loads per block in flight): fetched block by
      // block, each block would expose a full memory round trip; all 8 at once (24 registers) pushes the kernel over the 80 VGPRs that 6 workgroups per CU allow.
      constexpr int kNoiseGroup = 4;
#pragma unroll
      for (int g0 = 0; g0 < kBlocksPerWave; g0 += kNoiseGroup)
      {
      uint32_t nz8[kNoiseGroup][3];
#pragma unroll
      for (int bb = 0; bb < kNoiseGroup; bb++)
      {
        const uint32_t sb = wave * kBlocksPerWave + g0 + bb;
        const uint32_t w = (uint32_t)sgpr((int)L.shift[sb]);
        uint32_t call = (uint32_t)sgpr((int)L.first[sb]);
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
          const uint32_t s = (w >> (8 * k)) & 0xFF;
          nz8[bb][k] = 0;
          if (s != 0 && s != 8)
          {
            nz8[bb][k] = p.noise[(size_t)min(call, p.noiseLast) * 64 + lane];
            call++;
          }
        }
      }
#pragma unroll
      for (int bb = 0; bb < kNoiseGroup; bb++)
      {
        const int b = g0 + bb;
        const uint32_t sb = wave * kBlocksPerWave + b;
        const uint32_t bx = strip * kStripBlocks + sb;
        if (bx >= p.blocksX) continue;
        const uint32_t rx = min(p.sizeX - bx * kBlock, (uint32_t)kBlock), n = rx * ry;
        const bool active = (uint32_t)lane < n;
        uint32_t lx, ly;
        if (rx == 8) { lx = lane & 7; ly = lane >> 3; }
        else { const uint32_t l = active ? (uint32_t)lane : 0u; ly = l / rx; lx = l - ly * rx; }
        const uint32_t o = ly * kFacRow + sb * kBlock + lx;
        const uint32_t w = (uint32_t)sgpr((int)L.shift[sb]);
        const uint32_t shift[3] = { w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFF };

        uint32_t f[3];
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
          uint32_t v = L.fac[k * kFacPlane + o];
          const uint32_t s = shift[k];
          if (s != 0 && s != 8)
          { // src/limg.cpp:824-879: (lane16 & ditherSize) - ditherOffset, add, clamp, shift
            int t = (int)v + ((int)(nz8[bb][k] & ((1u << s) - 1u)) - (int)(1u << (s - 1)));
            t = t < 0 ? 0 : (t > 255 ? 255 : t);
            v = (uint32_t)t >> s;
          }
          f[k] = v;
        }

        if (!p.fullPlanes)
        { // compact mode: only the crushed factor bytes are wanted
          if (active)
          {
#pragma unroll
            for (int k = 0; k < 3; k++) out[k * kFacPlane + o] = (uint8_t)(f[k] << ((p.streamRaw && shift[k] == 8) ? 0u : shift[k]));
          }
          continue;
        }
        // decode: dec_k = byte * mul_k, est_c = sum_k (dec_k * n_k[c] + m_k[c]) >> 8, clamp.  24-bit multiplies are exact here:
        // dec <= 255 * 256 and |n| <= 65535 (difference of two int16), and v_mad_i32_i24 keeps the low 32 bits like PMULLD.
        const int *nm = L.nm + sb * 24;
        uint32_t decoded = 0;
        const int dA = (int)(f[0] * shift_mul(shift[0])), dB = (int)(f[1] * shift_mul(shift[1])), dC = (int)(f[2] * shift_mul(shift[2]));
        const int4 nA = *reinterpret_cast<const int4 *>(nm), nB = *reinterpret_cast<const int4 *>(nm + 4), nC = *reinterpret_cast<const int4 *>(nm + 8);
        const int4 mA = *reinterpret_cast<const int4 *>(nm + 12), mB = *reinterpret_cast<const int4 *>(nm + 16), mC = *reinterpret_cast<const int4 *>(nm + 20);
        const int nAa[4] = { nA.x, nA.y, nA.z, nA.w }, nBa[4] = { nB.x, nB.y, nB.z, nB.w }, nCa[4] = { nC.x, nC.y, nC.z, nC.w };
        const int mAa[4] = { mA.x - decode_bias(0, 0), mA.y - decode_bias(0, 1), mA.z, mA.w }, mBa[4] = { mB.x - decode_bias(1, 0), mB.y - decode_bias(1, 1), mB.z, mB.w },
                  mCa[4] = { mC.x - decode_bias(2, 0), mC.y - decode_bias(2, 1), mC.z, mC.w }; // (this form is exact for any record: no bias)
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          int est = (mad_i24(dA, nAa[c], mAa[c]) >> 8) + (mad_i24(dB, nBa[c], mBa[c]) >> 8) + (mad_i24(dC, nCa[c], mCa[c]) >> 8);
          est = est < 0 ? 0 : (est > 255 ? 255 : est);
          decoded |= (uint32_t)est << (8 * c);
        }
        if (active)
        {
          const uint32_t wo = ly * 64 + b * kBlock + lx;
          dec[wo] = decoded;
#pragma unroll
          for (int k = 0; k < 3; k++) out[k * kFacPlane + o] = (uint8_t)(f[k] << shift[k]); // shift 8 => 0 (src/limg.cpp:2054-2062)
        }
      }
      } // noise groups
      wave_lds_fence();

      const uint32_t wx0 = x0 + wave * 64;
      if (wx0 < p.sizeX && p.fullPlanes)
      {
        const uint32_t ww = min(p.sizeX - wx0, 64u);
        if ((uint32_t)lane < ww)
          for (uint32_t row = 0; row < ry; row++) io.info.pDecoded[(size_t)(y0 + row) * p.sizeX + wx0 + lane] = dec[row * 64 + lane];
      }
      __syncthreads(); // the three factor planes are stored strip-wide: 16 bytes per lane, whole rows of 256 bytes
      {
        const uint32_t stripW = min(p.sizeX - x0, (uint32_t)(kStripBlocks * kBlock));
        uint8_t *planes8[3] = { io.info.pFactorsA, io.info.pFactorsB, io.info.pFactorsC };
        if (p.vecFactors)
        {
          for (int i = tid; i < 384; i += kThreads)
          {
            const int pl = i >> 7, row = (i & 127) >> 4, col = (i & 15) * 16;
            if ((uint32_t)row < ry && (uint32_t)col < stripW)
              *reinterpret_cast<uint4 *>(planes8[pl] + (size_t)(y0 + row) * p.sizeX + x0 + col) = *reinterpret_cast<const uint4 *>(out + pl * kFacPlane + row * kFacRow + col);
          }
        }
        else
        {
          for (int i = tid; i < 3 * 2048; i += kThreads)
          {
            const int pl = i >> 11, row = (i & 2047) >> 8, col = i & 255;
            if ((uint32_t)row < ry && (uint32_t)col < stripW) planes8[pl][(size_t)(y0 + row) * p.sizeX + x0 + col] = out[pl * kFacPlane + row * kFacRow + col];
          }
        }
      }
    }

    // exclusive prefix of the dither-call counts of the strip's 32 blocks (wave 0), on top of the strip's base
    __device__ __forceinline__ void phase_f_first_calls(const StripLds &L, uint32_t base, int lane)
    {
      const uint32_t w = lane < kStripBlocks ? L.shift[lane] : 0u;
      const uint32_t calls = w >> 24;
      uint32_t incl = calls;
#pragma unroll
      for (int off = 1; off < 32; off <<= 1)
      {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += up;
      }
      if (lane < kStripBlocks) L.first[lane] = base + incl - calls;
    }

    // =====================================================================================================================
    // kernel 1: fit + factors + shift search
    // =====================================================================================================================

    // Per-block state in LDS.  The first 120 bytes are the float-stage state; once the record has been produced they are
    // dead and the same bytes carry what phase E needs (`BlkE` view).
    struct BlkF
    {
      float avg[4], dirA[4], dirB[4], dirC[4], est0[4]; // 80
      float mm[6];                                      // 104: minA maxA minB maxB minC maxC
      float inv_count, invA, invB, invC;                // 120
      uint32_t flags, n;                                // 128
      int16_t rec[24];                                  // 176
      float pad[4];                                     // 192
    };
    struct BlkE
    {
      float nrm[3][4]; // 48: float normals (max - min) of A, B, C            (slot order x0 x2 x1 x3)
      float off[3][4]; // 96: float dirA_min, dirB_offset, dirC_offset         (slot order)
      float invN[3];   // 108
    };
    static_assert(sizeof(BlkE) <= 120, "BlkE must fit the dead float-stage fields");
    static_assert(sizeof(BlkF) == 192, "BlkF layout");

    enum : int { kDirA = 0, kDirB = 1, kDirC = 2 };
    constexpr int kBatch = 4; // blocks per wave whose pass contributions are parked at a time
    constexpr uint32_t kBig = 16u; // some |record value| > kRecordLimit => generic 32-bit trial

    // Pixel-order accumulation (as `serial_sums`) followed, lane-parallel over the wave's 8 blocks, by everything the next
    // phase needs of the new direction: 1 / (dir . dir) with the DPPS order (correctly rounded division, once per 8 blocks)
    // and the all-zero flag.
    template <int CH, int WHICH, bool FAST>
    __device__ __forceinline__ void serial_sums2(const float *V, BlkF *blk, int lane)
    {
      wave_lds_fence();
      if (lane < 4 * kBatch)
      {
        const int b = lane >> 2, c = lane & 3;
        const float *src = V + b * kVDw + c;
        float s = 0.0f;
#pragma unroll 16
        for (int i = 0; i < 64; i++) s = s + src[i * 4];
        const float dir = s * blk[b].inv_count;
        float *dst = WHICH == kDirA ? blk[b].dirA : (WHICH == kDirB ? blk[b].dirB : blk[b].dirC);
        dst[c] = dir;
        // (p0 + p1) + (p2 + p3) inside each quad of lanes (slot order x0 x2 x1 x3: channels 0,1 sit in slots 0,2); float add is
        // commutative, so the two xor butterflies give exactly that
        float p = (CH == 3 && c == 3) ? 0.0f : dir * dir;
        p = p + __int_as_float(dpp<0x4E, 0xF>(0, __float_as_int(p))); // slot ^ 2
        p = p + __int_as_float(dpp<0xB1, 0xF>(0, __float_as_int(p))); // slot ^ 1
        uint32_t z = (dir == 0.0f) ? 1u : 0u;
        z &= (uint32_t)dpp<0xB1, 0xF>(0, (int)z);
        z &= (uint32_t)dpp<0x4E, 0xF>(0, (int)z);
        const float inv = FAST ? __builtin_amdgcn_rcpf(p) : 1.0f / p;
        if (c == 0)
        {
          if (WHICH == kDirA) { blk[b].invA = inv; if (z) blk[b].flags |= kZeroA | kZeroB | kZeroC; }
          else if (WHICH == kDirB) { blk[b].invB = inv; if (z) blk[b].flags |= kZeroB | kZeroC; }
          else { blk[b].invC = inv; if (z) blk[b].flags |= kZeroC; }
        }
      }
      wave_lds_fence();
    }

    // ---- decoupled look-back over the per-strip dither-call counts (fused path) ------------------------------------------
    // One 8-byte descriptor per work strip: value in the low word, status in the high word (0 = nothing yet, 1 = this strip's
    // own count, 2 = inclusive count of the chain up to and including this strip; the inclusive VALUE kBasePoison = a look-back
    // gave up here or earlier in the chain).  Written and read with relaxed agent-scope 8-byte atomics only: value and status travel in one granule, so no other ordering is needed.
    // Progress: EVERY strip id is drawn from the atomic ticket by a workgroup that is already running (k_encode_persistent), so
    // the holders of all smaller ids are resident whatever else shares the GPU -- other contexts' persistent kernels included --
    // and each of them publishes its count at the end of an E step, which never waits.  A look-back therefore terminates
    // without any assumption about how many workgroups of the grid are resident.  (Reference: strips on a thread pool always
    // complete and the entry points are re-entrant, src/limg.cpp:1890-1893, :2131-2136.)
    // The spin is bounded all the same (a protocol bug must not hang the GPU).  A timeout is LOUD: the strip raises the
    // context's sticky status word, publishes the poison value as its inclusive count and stores none of its chain-dependent planes;
    // every later strip of the chain finds the poison at once (no second spin), hands it on and stores nothing either.  (A poison STATUS of
    // its own, tested with one more ballot per poll, cost the 4-channel kernel two spilled VGPRs at its 80-register limit; the value does not.)  The host-pointer entries and
    // limg_hip_check_device_status then return limg_hip_error_Generic.
    constexpr uint32_t kDescAggregate = 1u, kDescInclusive = 2u;
    constexpr uint32_t kBasePoison = 0xFFFFFFFFu; // (a chain has < 2^26 dither calls: 3 per block)

    __device__ __forceinline__ void desc_store(unsigned long long *d, uint32_t status, uint32_t value)
    {
      __hip_atomic_store(d, ((unsigned long long)status << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ unsigned long long desc_load(unsigned long long *d)
    {
      return __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // bound of one look-back wait, in polls (~seconds); fault injection (a shorter bound, a strip that never publishes) exists in the test build only
    template <class P>
    __device__ __forceinline__ uint32_t lookback_spin_bound(const P &p)
    {
#ifdef LIMG_HIP_TEST_HOOKS
      return p.lookbackSpins;
#else
      (void)p;
      return 1u << 22;
#endif
    }
    template <class P>
    __device__ __forceinline__ bool publishes(const P &p, uint32_t id)
    {
#ifdef LIMG_HIP_TEST_HOOKS
      return id != p.testSkipStrip;
#else
      (void)p; (void)id;
      return true;
#endif
    }

    // called by all 64 lanes of one wave; returns the number of dither calls of the chain before strip `id`, or kBasePoison
    template <class P>
    __device__ __forceinline__ uint32_t lookback_base(const P &p, uint32_t id, uint32_t headId, uint32_t agg, int lane)
    {
      if (id == headId) return 0u;
      uint32_t base = 0;
      int hi = (int)id - 1; // nearest predecessor not yet accounted for
      for (;;)
      {
        const int j = hi - lane; // lane 0 looks at the nearest one
        const bool inrange = j >= (int)headId;
        unsigned long long d = ((unsigned long long)kDescInclusive << 32); // before the chain head: inclusive 0
        uint32_t spins = 0;
        for (;;)
        {
          if (inrange) d = desc_load(p.desc + j);
          const uint64_t incl = __builtin_amdgcn_ballot_w64((uint32_t)(d >> 32) == kDescInclusive);
          const uint64_t none = __builtin_amdgcn_ballot_w64((uint32_t)(d >> 32) == 0u);
          // every strip nearer than the nearest inclusive one must have published at least its own count
          const uint64_t nearer = incl ? ((incl & (0ull - incl)) - 1ull) : ~0ull;
          if ((none & nearer) == 0ull)
          {
            const uint32_t v = (uint32_t)d;
            if (incl)
            {
              const int fl = __builtin_ctzll(incl);
              const uint32_t vi = (uint32_t)__builtin_amdgcn_readlane((int)v, fl);
              base += wave_sum(lane <= fl ? v : 0u);
              return vi == kBasePoison ? kBasePoison : base; // (a poisoned predecessor publishes "inclusive, kBasePoison")
            }
            base += wave_sum(v);
            break;
          }
          if (++spins > lookback_spin_bound(p))
          {
            if (lane == 0) atomicExch(p.timeout, 1u);
            return kBasePoison;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        hi -= 64;
      }
    }

    // LDS of an E task (fit + search of one work strip); the F task's areas overlay `V`.
    constexpr int kLdsStrip = 0, kLdsV = kLdsStrip + 8 * kRowDw * 4, kLdsVBytes = kWaves * kBatch * kVDw * 4;
    constexpr int kLdsBlk = kLdsV + kLdsVBytes, kLdsCalls = kLdsBlk + kStripBlocks * 192, kLdsTotal = kLdsCalls + 32; // 4 per-wave call counts + the phase-E block queue
    static_assert(kLdsTotal <= 32768 - 16, "5 workgroups per CU");
    // PREFIT (float stage done by k_fit_tpb): the parked-contribution area shrinks to the 7.5 KiB factor-byte staging area; E layout 24.3 KiB, F overlay 23.9 KiB
    // => 6 workgroups per CU
    struct LdsLayout { int v, blk, calls, trialc, total; };
    constexpr int kTrialConstDw = 20; // per block: nA[3] nB[3] nC[3] mA[3] mB[3] mC[3] (+2 pad): the integer view of the record the packed trial multiplies with
    template <bool PREFIT> __device__ __host__ constexpr LdsLayout lds_layout()
    {
      if (!PREFIT) return LdsLayout{ kLdsV, kLdsBlk, kLdsCalls, kLdsV, kLdsTotal }; // (no trial-constant table in this layout)
      const int v = kLdsStrip + 8 * kRowDw * 4, blk = v + kFacBytes, calls = blk + kStripBlocks * 192, tc = calls + 32, e = tc + kStripBlocks * kTrialConstDw * 4;
      const int f = kLdsStrip + kPhaseFBytes + kStripBlocks * 48 + 16; // the F step's overlay incl. its record copy
      return LdsLayout{ v, blk, calls, tc, e > f ? e : f };
    }
    static_assert(lds_layout<true>().total <= 163840 / 6, "6 workgroups per CU with the float stage in its own kernel");


    // PERSIST == false: split path, the strip's call count goes to p.stripCalls for k_strip_scan.
    // PERSIST == true : persistent kernel, the count is published as an "aggregate" look-back descriptor.
    // parked results of one strip (persistent kernel): pre-dither factor bytes, records (int16 part), shift words
    constexpr int kParkFac = 0, kParkRec = 6144, kParkShift = 6144 + 1536, kParkBytes = 8192;

    // PREFIT: the float stage already ran in k_fit_tpb (limg_hip_fit_tpb.hip, one lane per block); this step loads the records and goes on with phase E.
    // id: the strip's number in ticket / look-back order (all images of a batch); local: its number inside its image (geometry); rowBase: the image's first
    // block row in the per-block scratch arrays; head0: the id of the image's first strip
    // ACC: the accurate search (fastBitCrushing == false) instead of the default one -- a kernel variant of its own, so that neither search's registers and code
    // weigh on the other
    template <int CH, bool PERSIST, bool FAST, bool PREFIT, bool ACC, class P, class IO>
    __device__ __forceinline__ void fit_search_strip(const P &p, const IO &io, const uint32_t id, const uint32_t local, const uint32_t rowBase, const uint32_t head0, uint8_t *lds,
                                                     uint8_t *park, const int tid)
    {
      // the 4 KiB RSQRTPS table is read straight from global memory (it lives in the CU's vector L1): keeping a copy in LDS would
      // cost the fifth workgroup per CU
      const unsigned short *s_rsq = d_rsqrt_x86_tab;
      uint32_t *s_strip = reinterpret_cast<uint32_t *>(lds + kLdsStrip);
      constexpr LdsLayout LL = lds_layout<PREFIT>();
      float *s_V = reinterpret_cast<float *>(lds + LL.v);
      BlkF *s_blk = reinterpret_cast<BlkF *>(lds + LL.blk);
      uint32_t *s_calls = reinterpret_cast<uint32_t *>(lds + LL.calls);
      int *s_trialc = reinterpret_cast<int *>(lds + LL.trialc);

      const int lane = tid & 63, wave = tid >> 6;
      const uint32_t strip = local % p.stripsX, by = local / p.stripsX;
      const uint32_t byS = rowBase + by; // block row in the per-block scratch arrays (records, shift words)
      const uint32_t x0 = strip * (kStripBlocks * kBlock), y0 = by * kBlock;
      const uint32_t stripW = min(p.sizeX - x0, (uint32_t)(kStripBlocks * kBlock)); // pixels
      const uint32_t ry = min(p.sizeY - y0, (uint32_t)kBlock);

      // PREFIT: the records of the wave's 8 blocks as k_fit_tpb left them (16 dwords per block: avg, then the 24 int16; 16 lanes per block) are REQUESTED here, next to
      // the strip's pixels, and used behind the barrier below: one memory round trip per strip instead of two.  Round 4, same-box A/B (tools/r04/run19.sh): persistent
      // kernel 1.079-1.085 -> 1.054-1.059 ms on 8192^2 photo-noise, config 4 15.01-15.13 -> 14.46-14.65 ms.  (The same idea for the F step -- the strip's own look-back
      // descriptor and the first window requested before the parked data -- costs 4 spilled registers and was measured 0.5-1 % slower: not kept.)
      uint32_t recVal[2] = { 0u, 0u };
      if (PREFIT)
      {
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
          const int b = r * 4 + (lane >> 4), w = lane & 15;
          const uint32_t bx = strip * kStripBlocks + wave * kBlocksPerWave + b;
          if (bx < p.blocksX) recVal[r] = w >= 4 ? reinterpret_cast<const uint32_t *>(p.records + (size_t)byS * p.blocksX + bx)[w] // (the averages in words 0..3 are not needed here)
                                                 : reinterpret_cast<const uint32_t *>(p.invN)[((size_t)byS * p.blocksX + bx) * 4 + w];      // 1 / |n|^2 of A, B, C from k_fit_tpb
        }
      }
      // ---- stage: the strip's pixel rows into LDS (the rsqrt table is loaded by the caller) ----------------------------
      if (p.vecIn)
      {
#pragma unroll
        for (int pass = 0; pass < 2; pass++)
        {
          const uint32_t row = pass * 4 + (tid >> 6), col = (tid & 63) * 4; // 4 px per lane
          if (row < ry && col < stripW)
          {
            const uint4 v = *reinterpret_cast<const uint4 *>(io.in + (size_t)(y0 + row) * p.sizeX + x0 + col); // (as a non-temporal load: measured, no difference -- tools/r05/ab_nt_loads.sh)
            *reinterpret_cast<uint4 *>(&s_strip[row * kRowDw + col]) = v;
          }
        }
      }
      else
      {
        for (uint32_t i = tid; i < 8 * 256; i += kThreads)
        {
          const uint32_t row = i >> 8, col = i & 255;
          if (row < ry && col < stripW) s_strip[row * kRowDw + col] = io.in[(size_t)(y0 + row) * p.sizeX + x0 + col];
        }
      }
      __syncthreads();

      float *V = s_V + wave * kBatch * kVDw;
      BlkF *blk = s_blk + wave * kBlocksPerWave;

      // per-block geometry (wave-uniform)
      auto geom = [&](int b, uint32_t &rx, uint32_t &n) -> bool
      {
        const uint32_t bx = strip * kStripBlocks + wave * kBlocksPerWave + b;
        if (bx >= p.blocksX) { rx = 0; n = 0; return false; }
        rx = min(p.sizeX - bx * kBlock, (uint32_t)kBlock);
        n = rx * ry;
        return true;
      };

      if (PREFIT)
      {
        // the records requested above: into the phase-E view, the park slot, and the range flags
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
          const int b = r * 4 + (lane >> 4), w = lane & 15;
          const uint32_t sb = wave * kBlocksPerWave + b, bx = strip * kStripBlocks + sb;
          const uint32_t val = recVal[r];
          if (w < 3) reinterpret_cast<BlkE *>(&blk[b])->invN[w] = __uint_as_float(val); // its place in the phase-E view (nothing else of the float-stage fields is live with PREFIT)
          if (w >= 4)
          {
            reinterpret_cast<uint32_t *>(blk[b].rec)[w - 4] = val;
            if (PERSIST) reinterpret_cast<uint32_t *>(park + kParkRec)[sb * 12 + (w - 4)] = val;
          }
          const int lo = (int)(int16_t)(val & 0xFFFFu), hi = (int)(int16_t)(val >> 16);
          uint32_t big = (w >= 4 && (lo > p.recordLimit || lo < -p.recordLimit || hi > p.recordLimit || hi < -p.recordLimit)) ? 1u : 0u;
          big |= (uint32_t)dpp<0xB1, 0xF>(0, (int)big);  // OR over the block's 16 lanes (one DPP row)
          big |= (uint32_t)dpp<0x4E, 0xF>(0, (int)big);
          big |= (uint32_t)dpp<0x141, 0xF>(0, (int)big);
          big |= (uint32_t)dpp<0x140, 0xF>(0, (int)big);
          if (w == 0) { blk[b].flags = (bx < p.blocksX ? kValid : 0u) | (big ? kBig : 0u); blk[b].n = bx < p.blocksX ? 64u : 0u; }
        }
        wave_lds_fence();
      }
      else
      {
      // The float stage runs in batches of kBatch blocks per wave: the parked contributions of one batch are what limits the
      // workgroups per CU (LDS), and 4 blocks x 4 waves keep it at 5 workgroups per CU.
      // Per-block values of the batch stay in registers across the phases (the loops over i are fully unrolled).
#pragma unroll 1
      for (int h = 0; h < kBlocksPerWave / kBatch; h++)
      {
      uint32_t px8[kBatch];
      V4 est8[kBatch];
      BlkF *const blkh = blk + h * kBatch;

      // ---- phase A: sums, average, first direction pass (a4, a5/a6 pass 1) ------------------------------------------
#pragma unroll
      for (int i = 0; i < kBatch; i++)
      {
        const int b = h * kBatch + i;
        uint32_t rx, n;
        px8[i] = 0;
        est8[i].a = float2_t{ 0.0f, 0.0f }; est8[i].b = float2_t{ 0.0f, 0.0f };
        if (!geom(b, rx, n))
        {
          if (lane == 0) { blk[b].flags = 0; blk[b].n = 0; blk[b].inv_count = 0.0f; }
          continue;
        }
        const uint32_t sb = wave * kBlocksPerWave + b;
        uint32_t lx, ly;
        if (rx == 8) { lx = lane & 7; ly = lane >> 3; }
        else { const uint32_t l = (uint32_t)lane < n ? (uint32_t)lane : 0u; ly = l / rx; lx = l - ly * rx; }
        uint32_t px = s_strip[ly * kRowDw + sb * kBlock + lx];
        px = (uint32_t)lane < n ? px : 0u;
        px8[i] = px;
        const V4 pf = px_to_v4(px);
        uint32_t pxs = px;
        if (n < 4)
        { // Upstream's SIMD sum loop consumes at least 4 pixels (src/limg.cpp:478-487): a block of fewer than 4 also sums what the previous block of its
          // strip left at positions n..3 of the gather buffer (:1890, :1899-1905).  Only the bottom-right corner block can be that small; the previous
          // block in raster order is its left neighbour, or the last block of the row above for a one-block-wide image.  The first block of a strip
          // has no predecessor (uninitialised stack upstream): nothing is added then.
          const uint32_t bxx = strip * kStripBlocks + sb;
          uint32_t chainStart = 0;
          if (p.chainCount > 1 && p.chainRows != 0) chainStart = min(by / p.chainRows, p.chainCount - 1) * p.chainRows;
          const bool hasPrev = bxx > 0 || by > chainStart;
          if (hasPrev && (uint32_t)lane >= n && lane < 4)
          {
            const uint32_t pbx = bxx > 0 ? bxx - 1 : p.blocksX - 1, pby = bxx > 0 ? by : by - 1;
            const uint32_t prx = min(p.sizeX - pbx * kBlock, (uint32_t)kBlock);
            const uint32_t row = (uint32_t)lane / prx, col = (uint32_t)lane - row * prx; // gather index -> position inside the previous block
            pxs = io.in[(size_t)(pby * kBlock + row) * p.sizeX + pbx * kBlock + col];
          }
        }
        const uint32_t s02 = wave_sum(pxs & 0x00FF00FFu), s13 = wave_sum((pxs >> 8) & 0x00FF00FFu);
        float inv_count = 0.015625f;
        if (n != 64) inv_count = 1.0f / (float)n;
        V4 avg;
        avg.a = float2_t{ (float)(int)(s02 & 0xFFFF), (float)(int)(s02 >> 16) } * inv_count;
        avg.b = float2_t{ (float)(int)(s13 & 0xFFFF), CH == 4 ? (float)(int)(s13 >> 16) : 0.0f } * inv_count;
        V4 d = pf - avg;
        mask_alpha<CH>(d);
        st4(V + i * kVDw + lane * 4, unit4<CH, FAST>(s_rsq, d, (uint32_t)lane < n));
        if (lane == 0)
        {
          st4(blk[b].avg, avg);
          *reinterpret_cast<float4 *>(blk[b].dirB) = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4 *>(blk[b].dirC) = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4 *>(blk[b].est0) = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4 *>(blk[b].mm) = make_float4(0.f, 0.f, 0.f, 0.f);
          blk[b].mm[4] = 0.0f; blk[b].mm[5] = 0.0f;
          blk[b].inv_count = inv_count; blk[b].n = n; blk[b].flags = kValid;
        }
      }
      serial_sums2<CH, kDirA, FAST>(V, blkh, lane);

      // ---- phase B: factor A extrema, residual -> second direction (pass 2) --------------------------------------------
      {
        static_assert(kBatch == 4, "wave_reduce4_min_max");
        float mnv[kBatch], mxv[kBatch];
        bool did[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; i++)
        {
          const int b = h * kBatch + i;
          uint32_t rx, n;
          did[i] = false; mnv[i] = 0.0f; mxv[i] = 0.0f;
          if (!geom(b, rx, n)) continue;
          if ((uint32_t)sgpr((int)blk[b].flags) & kZeroA) continue;
          did[i] = true;
          const V4 dirA = ld4(blk[b].dirA), avg = ld4(blk[b].avg);
          const float invA = blk[b].invA;
          const V4 pf = px_to_v4(px8[i]);
          const bool active = (uint32_t)lane < n;
          const float fA = dp4<CH, FAST>(pf - avg, dirA) * invA;
          mnv[i] = active ? fA : 0.0f; mxv[i] = mnv[i]; // min / max start at 0 upstream (src/limg_factorization.h:633-634)
          est8[i] = avg + dirA * fA;
          V4 e = pf - est8[i];
          mask_alpha<CH>(e);
          st4(V + i * kVDw + lane * 4, unit4<CH, FAST>(s_rsq, e, active));
        }
        wave_reduce4_min_max(mnv, mxv);
#pragma unroll
        for (int i = 0; i < kBatch; i++)
          if (did[i] && lane == 0) { blk[h * kBatch + i].mm[0] = vmin(mnv[i], 0.0f); blk[h * kBatch + i].mm[1] = vmax(mxv[i], 0.0f); }
      }
      serial_sums2<CH, kDirB, FAST>(V, blkh, lane);

      // ---- phase C: factor B (and, 3 ch, C) extrema; 4 ch: residual -> third direction (pass 3) ---------------------
      float mnCv[kBatch], mxCv[kBatch];
      bool didC[kBatch];
#pragma unroll
      for (int i = 0; i < kBatch; i++) { didC[i] = false; mnCv[i] = FLT_MAX; mxCv[i] = -FLT_MAX; }
#pragma unroll
      for (int i = 0; i < kBatch; i++)
      {
        const int b = h * kBatch + i;
        uint32_t rx, n;
        if (!geom(b, rx, n)) continue;
        if ((uint32_t)sgpr((int)blk[b].flags) & kZeroB) continue; // 1/0 = inf => every fB is NaN upstream => B and C collapse to 0
        const V4 dirB = ld4(blk[b].dirB);
        const float invB = blk[b].invB;
        const V4 pf = px_to_v4(px8[i]);
        const bool active = (uint32_t)lane < n;
        const float fB = dp4<CH, FAST>(pf - est8[i], dirB) * invB;
        float mnB = active ? fB : FLT_MAX, mxB = active ? fB : -FLT_MAX;
        if (CH == 4)
        {
          didC[i] = true; mnCv[i] = mnB; mxCv[i] = mxB; // reduced for the whole batch after the loop
          est8[i] = est8[i] + dirB * fB;
          st4(V + i * kVDw + lane * 4, unit4<CH, FAST>(s_rsq, pf - est8[i], active));
          if (lane == 0) st4(blk[b].est0, est8[i]);
        }
        else
        {
          // dirC = dirA x dirB (src/limg_factorization.h:498-507); slots: a = (x0, x2), b = (x1, x3)
          const V4 dirA = ld4(blk[b].dirA);
          V4 dirC;
          dirC.a.x = dirA.b.x * dirB.a.y - dirA.a.y * dirB.b.x; // A1 B2 - A2 B1
          dirC.b.x = dirA.a.y * dirB.a.x - dirA.a.x * dirB.a.y; // A2 B0 - A0 B2
          dirC.a.y = dirA.a.x * dirB.b.x - dirA.b.x * dirB.a.x; // A0 B1 - A1 B0
          dirC.b.y = 0.0f;
          const bool zeroC = sgpr((dirC.a.x == 0.0f && dirC.b.x == 0.0f && dirC.a.y == 0.0f) ? 1 : 0) != 0;
          float mnC = 0.0f, mxC = 0.0f;
          if (!zeroC)
          {
            const float invC = FAST ? __builtin_amdgcn_rcpf(dp4<CH, FAST>(dirC, dirC)) : 1.0f / dp4<CH, FAST>(dirC, dirC);
            const V4 e = pf - (est8[i] + dirB * fB);
            const float fC = dp4<CH, FAST>(e, dirC) * invC;
            mnC = active ? fC : FLT_MAX; mxC = active ? fC : -FLT_MAX;
            wave_min_max(mnB, mxB);
            wave_min_max(mnC, mxC);
          }
          else
            wave_min_max(mnB, mxB);
          if (lane == 0)
          {
            blk[b].mm[2] = mnB; blk[b].mm[3] = mxB; blk[b].mm[4] = mnC; blk[b].mm[5] = mxC;
            st4(blk[b].dirC, dirC);
            if (zeroC) blk[b].flags |= kZeroC;
          }
        }
      }
      if (CH == 4)
      {
        wave_reduce4_min_max(mnCv, mxCv);
#pragma unroll
        for (int i = 0; i < kBatch; i++)
          if (didC[i] && lane == 0) { blk[h * kBatch + i].mm[2] = mnCv[i]; blk[h * kBatch + i].mm[3] = mxCv[i]; }
        // blocks that skipped phase C left stale pass-2 contributions in V; their dirC is never used (flags)
        serial_sums2<CH, kDirC, FAST>(V, blkh, lane);
        // ---- phase D: factor C extrema (pass 4).  Upstream never advances its estimate pointer in this loop
        //      (src/limg_factorization.h:748-758), so every pixel is measured against pixel 0's A+B estimate.
        float mnDv[kBatch], mxDv[kBatch];
        bool didD[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; i++)
        {
          const int b = h * kBatch + i;
          uint32_t rx, n;
          didD[i] = false; mnDv[i] = FLT_MAX; mxDv[i] = -FLT_MAX;
          if (!geom(b, rx, n)) continue;
          if ((uint32_t)sgpr((int)blk[b].flags) & kZeroC) continue;
          didD[i] = true;
          const V4 dirC = ld4(blk[b].dirC), est0 = ld4(blk[b].est0);
          const float invC = blk[b].invC;
          const V4 pf = px_to_v4(px8[i]);
          const bool active = (uint32_t)lane < n;
          const float fC = dp4<CH, FAST>(pf - est0, dirC) * invC;
          mnDv[i] = active ? fC : FLT_MAX; mxDv[i] = active ? fC : -FLT_MAX;
        }
        wave_reduce4_min_max(mnDv, mxDv);
#pragma unroll
        for (int i = 0; i < kBatch; i++)
          if (didD[i] && lane == 0) { blk[h * kBatch + i].mm[4] = mnDv[i]; blk[h * kBatch + i].mm[5] = mxDv[i]; }
      }
      } // batches
      wave_lds_fence();

      // ---- records (src/limg_factorization.h:764-790): 8 lanes per block, 3 values per lane --------------------------
      {
        const int b = lane >> 3, j = lane & 7;
        const uint32_t flags = blk[b].flags;
        uint32_t big = 0;
#pragma unroll
        for (int r = 0; r < 3; r++)
        {
          const int kc = j + 8 * r, k = kc >> 2, c = kc & 3; // k = 2 r + (j >> 2): A for r == 0, B for r == 1, C for r == 2
          const float *dir = r == 0 ? blk[b].dirA : (r == 1 ? blk[b].dirB : blk[b].dirC);
          const int sl = slot_of(c);
          float m = blk[b].mm[k], dv = dir[sl];
          const bool dead = (r == 2 && (flags & kZeroC)) || (r >= 1 && (flags & kZeroB)) || (flags & kZeroA);
          if (dead) { m = 0.0f; dv = 0.0f; }
          float val = m * dv;
          if (r == 0) val = blk[b].avg[sl] + val;
          int q = cvt_rne(val);
          if ((CH == 3 && c == 3) || !(flags & kValid)) q = 0;
          big |= (q > p.recordLimit || q < -p.recordLimit) ? 1u : 0u;
          blk[b].rec[kc] = (int16_t)q;
        }
        big |= (uint32_t)dpp<0xB1, 0xF>(0, (int)big);
        big |= (uint32_t)dpp<0x4E, 0xF>(0, (int)big);
        big |= (uint32_t)dpp<0x141, 0xF>(0, (int)big);
        if (j == 0 && big) blk[b].flags = flags | kBig;
      }
      wave_lds_fence();
      // record -> global (16 dwords per block: avg, then the 24 int16); persistent kernel: the int16 part is parked
      {
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
          const int b = r * 4 + (lane >> 4), w = lane & 15;
          const uint32_t sb = wave * kBlocksPerWave + b, bx = strip * kStripBlocks + sb;
          const uint32_t val = w < 4 ? __float_as_uint(blk[b].avg[slot_of(w)]) : reinterpret_cast<const uint32_t *>(blk[b].rec)[w - 4];
          if (bx < p.blocksX && (!PERSIST || p.compactOut)) reinterpret_cast<uint32_t *>(p.records + (size_t)byS * p.blocksX + bx)[w] = val;
          if (PERSIST && w >= 4) reinterpret_cast<uint32_t *>(park + kParkRec)[sb * 12 + (w - 4)] = val;
        }
      }
      } // !PREFIT
      if (!PERSIST && p.fitOnly) return; // pass 1 of the merged-block encoder: the records are all it needs (uniform for the workgroup)
      // phase-E view (overlays the dead float-stage fields): float normals / offsets and 1 / |n|^2 in the serial limg_dot
      // order (src/limg_internal.h:426-452).  One lane per (block, factor, channel): 96 of 128 lane slots.
      {
        float nrm[2], off[2], invn[2];
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
          const int idx = min(r * 64 + lane, 95);
          const int b = idx / 12, fc = idx - b * 12, f = fc >> 2, c = fc & 3;
          const int lo = blk[b].rec[f * 8 + c], hi = blk[b].rec[f * 8 + 4 + c];
          nrm[r] = (float)(hi - lo); off[r] = (float)lo;
          if (PREFIT) { invn[r] = 0.0f; continue; } // k_fit_tpb left 1 / |n|^2 with the record
          const float sq = nrm[r] * nrm[r];
          const float s0 = __int_as_float(dpp<0x00, 0xF>(0, __float_as_int(sq))), s1 = __int_as_float(dpp<0x55, 0xF>(0, __float_as_int(sq)));
          const float s2 = __int_as_float(dpp<0xAA, 0xF>(0, __float_as_int(sq))), s3 = __int_as_float(dpp<0xFF, 0xF>(0, __float_as_int(sq)));
          float s = ((0.0f + s0) + s1) + s2;
          if (CH == 4) s = s + s3;
          const bool nz = (s0 != 0.0f) || (s1 != 0.0f) || (s2 != 0.0f) || (CH == 4 && s3 != 0.0f);
          invn[r] = nz ? (FAST ? __builtin_amdgcn_rcpf(s) : 1.0f / s) : 0.0f;
        }
        wave_lds_fence(); // every lane has read what it needs of the float-stage fields before the overlay is written
#pragma unroll
        for (int r = 0; r < 2; r++)
        {
          const int idx = r * 64 + lane;
          if (idx < 96)
          {
            const int b = idx / 12, fc = idx - b * 12, f = fc >> 2, c = fc & 3;
            BlkE *e = reinterpret_cast<BlkE *>(&blk[b]);
            e->nrm[f][slot_of(c)] = nrm[r]; e->off[f][slot_of(c)] = off[r]; // slot order x0 x2 x1 x3, see V4
            if (!PREFIT && c == 0) e->invN[f] = invn[r];
            if (PREFIT && c < 3)
            { // the packed trial's integer operands (negated, see "a9, packed form"), once per block here instead of per lane in phase E
              int *tc = s_trialc + (wave * kBlocksPerWave + b) * kTrialConstDw;
              tc[f * 3 + c] = -(int)nrm[r];
              tc[9 + f * 3 + c] = term_const(f, c, (int)off[r]);
            }
          }
        }
      }
      uint32_t *s_queue = s_calls + 4;
      if (tid == 0) *s_queue = 0;
      __syncthreads(); // all waves are done with V: wave 0's V region becomes the factor-byte staging area

      uint8_t *stage = reinterpret_cast<uint8_t *>(s_V); // [3 planes][8 rows][256 px]
      uint32_t waveCalls = 0;
      // (No zeroing of the parked shift words of blocks past the right edge here: with the dynamic queue below another wave may already have parked a real
      //  word for a block of this wave's range by the time this wave gets to issue such a store -- a rare lost update, found in round 2.  The F step masks
      //  the words of blocks that do not exist instead.)

      // ---- phase E: per-pixel factors (a8) + shift search (a10-a12) ----------------------------------------------------
      // The strip's 32 blocks are handed out dynamically: the number of trials differs from block to block (2 .. 20), and with a fixed 8 blocks per wave
      // the fastest wave would idle at the next barrier.  Everything phase E touches of a block lives in LDS and is indexed by the block, not the wave.
      auto grab = [&]() -> uint32_t { uint32_t v = 0; if (lane == 0) v = atomicAdd(s_queue, 1u); return (uint32_t)sgpr((int)v); };
      uint32_t sbNext = grab();
#pragma unroll 1
      for (;;)
      {
        const uint32_t sb = sbNext;
        if (sb >= (uint32_t)kStripBlocks) break;
        uint32_t qv = 0;
        if (lane == 0) qv = atomicAdd(s_queue, 1u); // the next block's index arrives while this one is searched
        const uint32_t bx = strip * kStripBlocks + sb;
        if (bx >= p.blocksX) { sbNext = (uint32_t)sgpr((int)qv); continue; }
        const uint32_t rx = min(p.sizeX - bx * kBlock, (uint32_t)kBlock), n = rx * ry;
        BlkF *const blkE = s_blk + sb;
        uint32_t lx, ly;
        if (rx == 8) { lx = lane & 7; ly = lane >> 3; }
        else { const uint32_t l = (uint32_t)lane < n ? (uint32_t)lane : 0u; ly = l / rx; lx = l - ly * rx; }
        uint32_t px = s_strip[ly * kRowDw + sb * kBlock + lx];
        const bool active = (uint32_t)lane < n;
        px = active ? px : 0u;
        const BlkE *be = reinterpret_cast<const BlkE *>(blkE);

        uint32_t fA, fB, fC;
        { // a8 (src/limg_factorization.h:149-197): fa = ((px - Amin) . nA) * invA, est = Amin + nA * fa, fb from px - est - Boff, ...
          const V4 pv = px_to_v4(px);
          const V4 nA = ld4(be->nrm[0]), mnA = ld4(be->off[0]);
          const float fa = dp4<CH, FAST>(pv - mnA, nA) * be->invN[0];
          fA = cvt_u8_rne_sat(255.0f * fa);
          const V4 nB = ld4(be->nrm[1]), ofB = ld4(be->off[1]);
          V4 est = mnA + nA * fa;
          const float fb = dp4<CH, FAST>((pv - est) - ofB, nB) * be->invN[1];
          fB = cvt_u8_rne_sat(255.0f * fb);
          const V4 nC = ld4(be->nrm[2]), ofC = ld4(be->off[2]);
          est = est + nB * fb;
          const float fc = dp4<CH, FAST>((pv - est) - ofC, nC) * be->invN[2];
          fC = cvt_u8_rne_sat(255.0f * fc);
        }

        uint32_t shift[3] = { 0, 0, 0 };
        if (p.forced[0] >= 0)
        {
          shift[0] = (uint32_t)p.forced[0]; shift[1] = (uint32_t)p.forced[1]; shift[2] = (uint32_t)p.forced[2];
        }
        else if (p.crushBits)
        {
          const uint64_t maxBlockN = p.maxBlock * (uint64_t)n;
          // be * 16 < maxBlock * n  <=>  be < ceil(maxBlock * n / 16); clamped to 32 bits (be itself never gets near 2^32).  Full blocks: from the host.
          uint32_t blockLimit = p.blockLimitFull;
          if (n != 64)
          {
            const uint64_t lim64 = (maxBlockN + 15ull) >> 4;
            blockLimit = (uint32_t)sgpr((int)(lim64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)lim64));
          }
          const bool big = ((uint32_t)sgpr((int)blkE->flags) & kBig) != 0;
          if (!big)
          {
            TrialState t;
            t.fA = fA; t.fB = fB; t.fC = fC;
            const uint32_t R = px & 0xFF, G = (px >> 8) & 0xFF;
            t.hiRG = (R + 0x8000u) | ((G + 0x8000u) << 16);
            t.loRG = (R + 0x8000u - 255u) | ((G + 0x8000u - 255u) << 16);
            t.pxB = (int)((px >> 16) & 0xFF);
            t.pxBlo = t.pxB - 255;
            if (PREFIT)
            { // uniform values, kept in VGPRs (they are operands of v_mad_i32_i24); prepared lane-parallel with the phase-E view above
              const int *tc = s_trialc + sb * kTrialConstDw;
#pragma unroll
              for (int c = 0; c < 3; c++)
              {
                t.nA[c] = tc[c]; t.nB[c] = tc[3 + c]; t.nC[c] = tc[6 + c];
                t.mA[c] = tc[9 + c]; t.mB[c] = tc[12 + c]; t.mC[c] = tc[15 + c];
              }
            }
            else
            { // lane == pixel float stage: no LDS left over for the table (other waves' parked contributions are still live when this wave gets here)
#pragma unroll
              for (int c = 0; c < 3; c++)
              {
                const int loA = blkE->rec[c], hiA = blkE->rec[4 + c], loB = blkE->rec[8 + c], hiB = blkE->rec[12 + c], loC = blkE->rec[16 + c], hiC = blkE->rec[20 + c];
                t.nA[c] = loA - hiA; t.nB[c] = loB - hiB; t.nC[c] = loC - hiC;
                t.mA[c] = term_const(0, c, loA); t.mB[c] = term_const(1, c, loB); t.mC[c] = term_const(2, c, loC);
              }
            }
            // per pixel: factor A's (negated) terms become  channel - term, so the three terms of a channel sum to  channel - estimate
            t.mA[0] += (int)(R << 8); t.mA[1] += (int)(G << 8); t.mA[2] += t.pxB << 8;
            // (the flag goes through an opaque scalar: hoisted out of the block loop as a boolean it comes back as a lane mask that is negated with two vector
            //  instructions per block)
            if (!ACC)
            {
              if (n == 64) search_fast_automaton<true>(t, true, p.maxPixel32, blockLimit, shift);
              else search_fast_automaton<false>(t, active, p.maxPixel32, blockLimit, shift);
            }
            else
            {
              if (n == 64) search_accurate_automaton<true>(t, true, p.maxPixel32, blockLimit, p.accTable, shift);
              else search_accurate_automaton<false>(t, active, p.maxPixel32, blockLimit, p.accTable, shift);
            }
          }
          else
          {
            // out-of-range record (never produced by a fit of byte pixels; kept so that no input can break exactness):
            // generic 32-bit trial, deliberately a real call so that none of it is speculated into the common path
            const uint32_t packed = (uint32_t)sgpr((int)search_generic(px, fA, fB, fC, blkE->rec, active, p.maxPixel32, maxBlockN, !ACC)); // uniform: keeps the shift triple (and the bookkeeping below) on the scalar unit for the common path too
            shift[0] = packed & 0xFF; shift[1] = (packed >> 8) & 0xFF; shift[2] = (packed >> 16) & 0xFF;
          }
        }

        // dither calls this block will make (src/limg.cpp:1951-1958)
        // a shift of 1..7 dithers (one call), 0 and 8 do not; as scalar arithmetic -- ((s & 7) + 7) >> 3 -- because a boolean would go through a lane mask and a vector select
        const uint32_t calls = (((shift[0] & 7u) + 7u) >> 3) + (((shift[1] & 7u) + 7u) >> 3) + (((shift[2] & 7u) + 7u) >> 3);
        waveCalls += calls;
        if (p.stripWords)
        { // stream mode: the block's payload size in 8-byte words rides in bits 8.. of the same per-wave counter (calls of a strip stay below 256): a field of 8 - shift
          // bits per pixel is that many words; a factor at shift 8 has none unless its alpha normal is non-zero (raw-byte escape, limg_hip_stream.hip)
          uint32_t words = 0;
#pragma unroll
          for (int k = 0; k < 3; k++)
          {
            const uint32_t sk = shift[k];
            if (sk < 8u) words += 8u - sk;
            else if (CH == 4 && sgpr((int)blkE->rec[8 * k + 3]) != sgpr((int)blkE->rec[8 * k + 7])) words += 8u;
          }
          waveCalls += words << 8;
        }

        const size_t bi = (size_t)byS * p.blocksX + bx;
        const uint32_t word = shift[0] | (shift[1] << 8) | (shift[2] << 16) | (calls << 24);
        if (lane == 0)
        {
          if (!PERSIST || p.compactOut) p.shifts[bi] = word;
          if (PERSIST) reinterpret_cast<uint32_t *>(park + kParkShift)[sb] = word;
        }
        if (p.storePlanes && active)
        {
          const uint32_t o = ly * kFacRow + sb * kBlock + lx;
          stage[o] = (uint8_t)fA; stage[kFacPlane + o] = (uint8_t)fB; stage[2 * kFacPlane + o] = (uint8_t)fC;
        }
        sbNext = (uint32_t)sgpr((int)qv);
      }
      if (lane == 0) s_calls[wave] = waveCalls;
      __syncthreads();
      if (tid == 0)
      {
        const uint32_t aggAll = s_calls[0] + s_calls[1] + s_calls[2] + s_calls[3];
        const uint32_t agg = aggAll & 0xFFu; // (bits 8..: the strip's payload words, stream mode)
        if (p.stripWords) p.stripWords[id] = aggAll >> 8;
        if (PERSIST)
        { // publish the count; if the predecessor's inclusive count is already there, publish ours as inclusive right away
          uint32_t headId = head0;
          if (p.chainCount > 1 && p.chainRows != 0)
          {
            uint32_t c = by / p.chainRows;
            c = c < p.chainCount - 1 ? c : p.chainCount - 1;
            headId = head0 + c * p.chainRows * p.stripsX;
          }
#ifdef LIMG_HIP_TEST_HOOKS
          if (id == p.testSkipStrip) {} // (test build: a strip that never publishes)
          else
#endif
          if (id == headId) desc_store(p.desc + id, kDescInclusive, agg);
          else
          {
            const unsigned long long d = desc_load(p.desc + id - 1);
            if ((uint32_t)(d >> 32) == kDescInclusive) desc_store(p.desc + id, kDescInclusive, (uint32_t)d == kBasePoison ? kBasePoison : (uint32_t)d + agg);
            else desc_store(p.desc + id, kDescAggregate, agg);
          }
        }
        else p.stripCalls[id] = agg;
      }
      if (PERSIST)
      { // park the pre-dither factor bytes (private, L2-resident scratch; same workgroup reads them back)
        for (int i = tid; i < 384; i += kThreads) reinterpret_cast<uint4 *>(park + kParkFac)[i] = *reinterpret_cast<const uint4 *>(stage + (i >> 4) * kFacRow + (i & 15) * 16); // 24 rows of 256 bytes
        return;
      }

      // ---- pre-dither factor bytes -> the caller's factor planes (rewritten in place by k_dither_store) -------------
      if (p.storePlanes)
      {
        uint8_t *planes[3] = { io.info.pFactorsA, io.info.pFactorsB, io.info.pFactorsC };
        if (p.vecFactors)
        {
          for (int i = tid; i < 384; i += kThreads)
          {
            const int pl = i >> 7, row = (i & 127) >> 4, col = (i & 15) * 16;
            if ((uint32_t)row < ry && (uint32_t)col < stripW)
              *reinterpret_cast<uint4 *>(planes[pl] + (size_t)(y0 + row) * p.sizeX + x0 + col) = *reinterpret_cast<const uint4 *>(stage + pl * kFacPlane + row * kFacRow + col);
          }
        }
        else
        {
          for (int i = tid; i < 3 * 2048; i += kThreads)
          {
            const int pl = i >> 11, row = (i & 2047) >> 8, col = i & 255;
            if ((uint32_t)row < ry && (uint32_t)col < stripW) planes[pl][(size_t)(y0 + row) * p.sizeX + x0 + col] = stage[pl * kFacPlane + row * kFacRow + col];
          }
        }
      }
    }

    // =====================================================================================================================
    // kernel 2: exclusive scan of the per-strip dither-call counts in raster order, restarting at every chain boundary
    // =====================================================================================================================
    __global__ __launch_bounds__(1024) void k_strip_scan(const EncodeParams p)
    {
      __shared__ uint32_t s_sum[1024];
      __shared__ uint32_t s_flag[1024];
      const uint32_t total = p.blocksY * p.stripsX;
      const uint32_t per = (total + 1023u) / 1024u;
      const uint32_t t = threadIdx.x;
      const uint32_t begin = min(t * per, total), end = min(begin + per, total);
      auto chain_of = [&](uint32_t e) -> uint32_t
      {
        const uint32_t row = e / p.stripsX;
        if (p.chainCount <= 1 || p.chainRows == 0) return 0u;
        const uint32_t c = row / p.chainRows;
        return c < p.chainCount - 1 ? c : p.chainCount - 1;
      };
      auto is_head = [&](uint32_t e) -> bool { return e == 0 || chain_of(e) != chain_of(e - 1); };

      uint32_t sum = 0, flag = 0;
      for (uint32_t e = begin; e < end; e++)
      {
        if (is_head(e)) { sum = 0; flag = 1; }
        sum += p.stripCalls[e];
      }
      s_sum[t] = sum; s_flag[t] = flag;
      __syncthreads();
      // segmented inclusive scan (Hillis-Steele) over the 1024 partials
      for (uint32_t off = 1; off < 1024; off <<= 1)
      {
        uint32_t vs = s_sum[t], vf = s_flag[t];
        if (t >= off && !vf) { vs += s_sum[t - off]; vf = s_flag[t - off]; }
        __syncthreads();
        s_sum[t] = vs; s_flag[t] = vf;
        __syncthreads();
      }
      uint32_t run = (t == 0) ? 0u : s_sum[t - 1]; // calls in the current chain before this thread's chunk
      for (uint32_t e = begin; e < end; e++)
      {
        if (is_head(e)) run = 0;
        p.stripBase[e] = run;
        run += p.stripCalls[e];
      }
      // cross-GPU single chain (limg_hip_encode3d_chain_device): the dither calls of this whole image strip, for the exchange between the E and the F step
      if (p.chainCallsOut && end == total && begin < end) *p.chainCallsOut = run;
    }

    // =====================================================================================================================
    // kernel 3 (split path): phase F as its own launch
    // =====================================================================================================================
    // F task: dither + stores + decode of one work strip from the parked per-block results.
    // PERSIST == false: the strip's chain position comes from k_strip_scan (p.stripBase);
    // PERSIST == true : from the look-back over the descriptors, and the strip's inclusive count is published first thing.
    static_assert(kPhaseFBytes + kStripBlocks * 48 <= kLdsTotal - kLdsStrip - 16, "the F step's LDS overlays the E step's");

    template <int CH, bool PERSIST, class P, class IO>
    __device__ __forceinline__ void dither_store_strip(const P &p, const IO &io, const uint32_t id, const uint32_t local, const uint32_t rowBase, const uint32_t head0, uint8_t *fbase,
                                                       const uint8_t *park, const int tid)
    {
      int16_t *s_rec = reinterpret_cast<int16_t *>(fbase + kPhaseFBytes); // [32][24]
      const int lane = tid & 63, wave = tid >> 6;
      const uint32_t strip = local % p.stripsX, by = local / p.stripsX;
      const uint32_t byS = rowBase + by;
      const uint32_t x0 = strip * (kStripBlocks * kBlock), y0 = by * kBlock;
      const uint32_t stripW = min(p.sizeX - x0, (uint32_t)(kStripBlocks * kBlock));
      const uint32_t ry = min(p.sizeY - y0, (uint32_t)kBlock);
      const uint32_t nBlocks = min(p.blocksX - strip * kStripBlocks, (uint32_t)kStripBlocks);
      const uint8_t *planesIn[3] = { io.info.pFactorsA, io.info.pFactorsB, io.info.pFactorsC };
      const StripLds L = carve_phase_f(fbase, s_rec, 24);

      if (PERSIST)
      {
        // (rolled on purpose: unrolled, the second loop became 19 dword loads and ~100 vector instructions of address arithmetic per wave; its 384 dwords go as 96 16-byte pieces)
        static_assert(kStripBlocks * 12 * 4 == 96 * 16, "record copy");
#pragma unroll 1
        for (int i = tid; i < 384; i += kThreads) *reinterpret_cast<uint4 *>(L.fac + (i >> 4) * kFacRow + (i & 15) * 16) = reinterpret_cast<const uint4 *>(park + kParkFac)[i];
        if (tid < 96) reinterpret_cast<uint4 *>(s_rec)[tid] = reinterpret_cast<const uint4 *>(park + kParkRec)[tid];
        if (tid < kStripBlocks) L.shift[tid] = (uint32_t)tid < nBlocks ? reinterpret_cast<const uint32_t *>(park + kParkShift)[tid] : 0u; // nothing is parked for blocks past the right edge
      }
      else
      {
        if (p.vecFactors)
        {
          for (int i = tid; i < 384; i += kThreads)
          {
            const int pl = i >> 7, row = (i & 127) >> 4, col = (i & 15) * 16;
            if ((uint32_t)row < ry && (uint32_t)col < stripW)
              *reinterpret_cast<uint4 *>(L.fac + pl * kFacPlane + row * kFacRow + col) = *reinterpret_cast<const uint4 *>(planesIn[pl] + (size_t)(y0 + row) * p.sizeX + x0 + col);
          }
        }
        else
        {
          for (int i = tid; i < 3 * 2048; i += kThreads)
          {
            const int pl = i >> 11, row = (i & 2047) >> 8, col = i & 255;
            if ((uint32_t)row < ry && (uint32_t)col < stripW) L.fac[pl * kFacPlane + row * kFacRow + col] = planesIn[pl][(size_t)(y0 + row) * p.sizeX + x0 + col];
          }
        }
        // records (12 dwords of int16 per block) and shift words
        for (int i = tid; i < kStripBlocks * 12; i += kThreads)
        {
          const int sb = i / 12, w = i - sb * 12;
          uint32_t v = 0;
          if ((uint32_t)sb < nBlocks) v = reinterpret_cast<const uint32_t *>(p.records + (size_t)byS * p.blocksX + strip * kStripBlocks + sb)[4 + w];
          reinterpret_cast<uint32_t *>(s_rec + sb * 24)[w] = v;
        }
        if (tid < kStripBlocks) L.shift[tid] = (uint32_t)tid < nBlocks ? p.shifts[(size_t)byS * p.blocksX + strip * kStripBlocks + tid] : 0u;
      }
      __syncthreads();
      phase_f_prepare<CH>(L, lane, wave, p.recordLimit);
      wave_lds_fence();
      // strips of whole blocks (always in the persistent kernel; in the split path unless the image has partial edge blocks): the uniform planes' stores run inside
      // phase_f_rows, behind its noise loads; otherwise here
      const bool rowsPath = PERSIST || (p.sizeX % kBlock == 0 && ry == (uint32_t)kBlock);
      // the uniform planes (base-independent: 28 of the 35 output bytes per pixel): rows 0..3 here, beside wave 0's look-back; rows 4..7 behind the noise loads
      if (p.fullPlanes) phase_f_store_const(p, io, L, x0, y0, ry, lane, wave, rowsPath ? 1u : 3u);
      if (wave == 0)
      {
        uint32_t base;
        if (PERSIST)
        {
          uint32_t headId = head0;
          if (p.chainCount > 1 && p.chainRows != 0)
          {
            uint32_t c = by / p.chainRows;
            c = c < p.chainCount - 1 ? c : p.chainCount - 1;
            headId = head0 + c * p.chainRows * p.stripsX;
          }
          const uint32_t w = lane < kStripBlocks ? L.shift[lane] : 0u;
          const uint32_t agg = wave_sum(w >> 24);
          const unsigned long long own = desc_load(p.desc + id);
          if ((uint32_t)(own >> 32) == kDescInclusive) base = (uint32_t)sgpr((int)((uint32_t)own == kBasePoison ? kBasePoison : (uint32_t)own - agg));
          else
          {
            base = lookback_base(p, id, headId, agg, lane);
            if (lane == 0 && publishes(p, id)) desc_store(p.desc + id, kDescInclusive, base == kBasePoison ? kBasePoison : base + agg);
          }
#ifdef LIMG_HIP_TEST_HOOKS
          if (id == p.testBaseErrStrip) base += 1u; // (test build: this strip alone dithers from the wrong place)
#endif
        }
        else
        {
          base = p.stripBase[id];
          if (p.chainBase) base += (uint32_t)*p.chainBase; // first dither call of this image strip inside a chain that started on another GPU
        }
        phase_f_first_calls(L, base, lane);
      }
      __syncthreads();
      if (PERSIST && L.first[0] == kBasePoison) return; // the look-back gave up (status word raised): nothing that depends on the chain position is stored (completing the
                                                        // block-uniform planes' rows 4..7 here costs the 4-channel kernel two spilled VGPRs: they stay half written, the call has failed anyway)
      if (rowsPath) phase_f_rows<CH>(p, io, L, strip, x0, y0, lane, wave, [&]() { if (p.fullPlanes) phase_f_store_const(p, io, L, x0, y0, ry, lane, wave, 2u); });
      else phase_f_pixels<CH>(p, io, L, strip, x0, y0, ry, lane, wave, tid);
    }

    // ---- kernels ---------------------------------------------------------------------------------------------------------
    template <int CH, bool FAST, bool PREFIT, bool ACC>
    __global__ __launch_bounds__(kThreads) void k_fit_search(const EncodeParams p)
    {
      __shared__ __attribute__((aligned(16))) uint8_t s_lds[lds_layout<PREFIT>().total];
      fit_search_strip<CH, false, FAST, PREFIT, ACC>(p, p.io, blockIdx.x, blockIdx.x, 0u, 0u, s_lds, nullptr, (int)threadIdx.x);
    }

    template <int CH>
    __global__ __launch_bounds__(kThreads) void k_dither_store(const EncodeParams p)
    {
      __shared__ __attribute__((aligned(16))) uint8_t s_lds[kPhaseFBytes + kStripBlocks * 48];
      if (p.chainBase && *p.chainBase == ~0ull) return; // cross-GPU chain aborted by a rank (k_chain_base): nothing is stored, the status word is already raised
      dither_store_strip<CH, false>(p, p.io, blockIdx.x, blockIdx.x, 0u, 0u, s_lds, nullptr, (int)threadIdx.x);
    }

    // Both steps are inlined into the loop.  Left alone, LLVM hoists every lane-dependent address computation of both steps
    // out of the loop (they only depend on threadIdx) and keeps them all live: 194 VGPRs.  Passing the thread id through an
    // empty asm at the top of each iteration makes it opaque per iteration, which keeps the two steps' live ranges apart.
    // Persistent single-launch encode: 6 workgroups per CU (5 with the float stage inside) loop over the work strips (ticket order).  Each iteration runs the
    // VALU-bound E step (fit + search) of a new strip and then the HBM-bound F step (dither, decode, all plane stores) of the
    // strip the SAME workgroup fitted one iteration earlier, whose parked results sit in a private, L2-resident 8 KiB slot.
    // The one-iteration lag means that by the time an F step asks for its strip's position in the dither chain, every
    // earlier strip has long published its call count, so the look-back does not wait; and since the workgroups of a CU
    // drift apart, E and F steps of different workgroups overlap on every CU.
    // Progress: a look-back only waits for strips with smaller tickets; those were drawn earlier by workgroups that are running, and an E step never waits
    // (see "decoupled look-back" above: no strip id is ever derived from blockIdx).
    template <int CH, bool FAST, bool PREFIT, bool ACC>
#ifndef LIMG_ACC_WG
#define LIMG_ACC_WG 6
#endif
    __global__ __launch_bounds__(kThreads, PREFIT ? (ACC ? LIMG_ACC_WG : 6) : 5) void k_encode_persistent(const EncodeParams p)
    {
      __shared__ __attribute__((aligned(16))) uint8_t s_lds[lds_layout<PREFIT>().total];
      __shared__ uint32_t s_ticket;
      const int tid = (int)threadIdx.x;
      const uint32_t S = p.imageStrips * p.batchCount; // the strips of all images of a batch, image after image
      uint8_t *park = p.park + (size_t)blockIdx.x * 2 * kParkBytes;
      uint32_t prev = 0xFFFFFFFFu, prevImg = 0, slot = 0;
      // The two steps read the parameters straight from the kernel-argument segment, through a pointer the compiler cannot see through from one step to the
      // next: a field is then fetched (one scalar load, scalar-cache hit) where a step uses it.  Read from `p`, all ~70 dwords are loaded once before the
      // loop and kept alive across it -- more SGPRs than there are, so the compiler parks them in VGPR lanes and pays a quarter-rate v_readlane per use.
      typedef const __attribute__((address_space(4))) EncodeParams KernArgs;
      KernArgs *const kargs = (KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
      // the caller's pointers of the image a strip belongs to: the kernel arguments themselves for a single image, an entry of the device table for a batch (same
      // address space, same scalar loads).  The table was written by a copy that precedes this launch on the stream and is not modified while the kernel runs.
      typedef const __attribute__((address_space(4))) ImageIO KernIO;
      auto io_of = [](KernArgs *a, uint32_t img) -> KernIO * { return a->batchCount > 1 ? (KernIO *)a->batch + img : &a->io; };
      // All workgroups start their first E step together, so their first F steps (latency-bound, little vector work) coincide too, and it takes a few iterations
      // of data-dependent search lengths before the E and F steps of a CU's six workgroups interleave.  Workgroups are dealt out to the 256 CUs residency slot
      // by slot, so slot k (= blockIdx.x / 256) starts k * 3.4 us late: measured -0.5 % on the kernel, and harmless where the placement differs.
      for (uint32_t i = 0; i < (blockIdx.x >> 8); i++) __builtin_amdgcn_s_sleep(127);
      __builtin_amdgcn_s_setprio(LIMG_PRIO_E);
      for (;;)
      {
        __syncthreads(); // the previous step's LDS use is over (and the rsqrt table is in place)
        // EVERY strip id is drawn from the ticket, the first one included: a workgroup that holds id t is resident, and so is the holder of every id < t (it drew
        // earlier).  Round 4 let workgroup i take strip i without asking (one atomic round trip less per workgroup): with a second persistent kernel on the GPU
        // (another context's stream) a resident workgroup then waited for strips of workgroups that were never dispatched, and both kernels spun into the
        // look-back's bound.  Never derive a strip id from blockIdx.
        if (tid == 0) s_ticket = atomicAdd(p.ticket, 1u);
        __syncthreads();
        const uint32_t t = (uint32_t)sgpr((int)s_ticket);
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        KernArgs *pe = kargs;
        asm volatile("" : "+s"(pe));
        uint32_t img = 0;
        if (t < S)
        {
          if (pe->batchCount > 1) img = t / pe->imageStrips; // once per strip, on the scalar unit
          const uint32_t head0 = img * pe->imageStrips;
          fit_search_strip<CH, true, FAST, PREFIT, ACC>(*pe, *io_of(pe, img), t, t - head0, img * pe->blocksY, head0, s_lds, park + slot * kParkBytes, tid_e);
        }
        if (prev != 0xFFFFFFFFu)
        {
          __syncthreads();
          int tid_f = tid;
          asm volatile("" : "+v"(tid_f));
          KernArgs *pf = kargs;
          asm volatile("" : "+s"(pf));
          // The F step is latency-bound with little vector work, the E step is what keeps the vector unit busy: E-step waves get the issue priority (s_setprio),
          // F-step waves take the slots they leave.  Measured: -2.5 % on the kernel (the other way round: +0.8 %).
          __builtin_amdgcn_s_setprio(0);
          const uint32_t head0 = prevImg * pf->imageStrips;
          dither_store_strip<CH, true>(*pf, *io_of(pf, prevImg), prev, prev - head0, prevImg * pf->blocksY, head0, s_lds + kLdsStrip, park + (slot ^ 1u) * kParkBytes, tid_f);
          __builtin_amdgcn_s_setprio(LIMG_PRIO_E);
        }
        if (t >= S) break;
        prev = t; prevImg = img;
        slot ^= 1u;
      }
    }
  } // namespace

  // kernel variant by (channels, float mode, float stage already done by k_fit_tpb, accurate search)
#define LIMG_DISPATCH_ACC(KERNEL, GRID, BLOCK, S, P, ACC)                                               \
  do                                                                                                    \
  {                                                                                                     \
    const int v_ = (channels == 4 ? 4 : 0) | ((P).floatFast ? 2 : 0) | ((P).prefit ? 1 : 0);            \
    switch (v_)                                                                                         \
    {                                                                                                   \
    case 0: hipLaunchKernelGGL((KERNEL<3, false, false, ACC>), GRID, BLOCK, 0, S, P); break;            \
    case 1: hipLaunchKernelGGL((KERNEL<3, false, true, ACC>), GRID, BLOCK, 0, S, P); break;             \
    case 2: hipLaunchKernelGGL((KERNEL<3, true, false, ACC>), GRID, BLOCK, 0, S, P); break;             \
    case 3: hipLaunchKernelGGL((KERNEL<3, true, true, ACC>), GRID, BLOCK, 0, S, P); break;              \
    case 4: hipLaunchKernelGGL((KERNEL<4, false, false, ACC>), GRID, BLOCK, 0, S, P); break;            \
    case 5: hipLaunchKernelGGL((KERNEL<4, false, true, ACC>), GRID, BLOCK, 0, S, P); break;             \
    case 6: hipLaunchKernelGGL((KERNEL<4, true, false, ACC>), GRID, BLOCK, 0, S, P); break;             \
    default: hipLaunchKernelGGL((KERNEL<4, true, true, ACC>), GRID, BLOCK, 0, S, P); break;             \
    }                                                                                                   \
  } while (0)
#define LIMG_DISPATCH(KERNEL, GRID, BLOCK, S, P)                                                        \
  do                                                                                                    \
  {                                                                                                     \
    if ((P).fast || !(P).crushBits) LIMG_DISPATCH_ACC(KERNEL, GRID, BLOCK, S, P, false);                \
    else LIMG_DISPATCH_ACC(KERNEL, GRID, BLOCK, S, P, true);                                            \
  } while (0)

  void launch_fit_search(const EncodeParams &p, int channels, hipStream_t s)
  {
    const dim3 grid(p.stripsX * p.blocksY), block(kThreads);
    LIMG_DISPATCH(k_fit_search, grid, block, s, p);
  }

  void launch_encode_persistent(const EncodeParams &p, int channels, int workgroups, hipStream_t s)
  {
    const uint32_t strips = p.imageStrips * p.batchCount;
    const dim3 grid(strips < (uint32_t)workgroups ? strips : (uint32_t)workgroups), block(kThreads);
    LIMG_DISPATCH(k_encode_persistent, grid, block, s, p);
  }

  namespace
  {
    // exclusive prefix over the ranks before `rank` of the all-gathered per-rank dither-call totals.  A rank whose E step failed joins the all-gather with ~0: the
    // base is then poisoned too (k_dither_store stores nothing) and the context's sticky "aborted chain" word is raised for limg_hip_check_device_status.
    __global__ void k_chain_base(const unsigned long long *calls, int rank, int world, unsigned long long *base, uint32_t *aborted)
    {
      unsigned long long run = 0;
      bool poison = false;
      for (int r = 0; r < world; r++)
      {
        poison = poison || calls[r] == ~0ull;
        if (r < rank) run += calls[r];
      }
      *base = poison ? ~0ull : run;
      if (poison) *aborted = 1u;
    }
  }
  namespace
  {
    // The reference's bit statistics (src/limg.cpp:1971-1999, printed at :2232-2248): per factor the bits kept, summed over the pixels, and a histogram of the
    // pixels by shift -- a function of the per-block shift words and the blocks' pixel counts alone.  out[0..2] += (8 - shift) * n, out[3 + 9 f + shift] += n.
    __global__ __launch_bounds__(256) void k_shift_stats(const uint32_t *shifts, uint32_t blocksX, uint32_t blocksY, uint32_t rows /* blocksY x images */, uint32_t sizeX, uint32_t sizeY,
                                                         unsigned long long *out)
    {
      __shared__ uint32_t s_acc[30];
      if (threadIdx.x < 30) s_acc[threadIdx.x] = 0;
      __syncthreads();
      const uint64_t total = (uint64_t)blocksX * rows;
      // a workgroup covers at most 256 * 16 blocks of <= 64 pixels and <= 8 bits each: the 32-bit partial sums cannot overflow
      const uint64_t begin = (uint64_t)blockIdx.x * 4096u;
      for (uint64_t i = begin + threadIdx.x; i < min(begin + 4096u, total); i += 256u)
      {
        const uint32_t row = (uint32_t)(i / blocksX), bx = (uint32_t)(i - (uint64_t)row * blocksX), by = row % blocksY;
        const uint32_t rx = min(sizeX - bx * kBlock, (uint32_t)kBlock), ry = min(sizeY - by * kBlock, (uint32_t)kBlock), n = rx * ry;
        const uint32_t w = shifts[i];
#pragma unroll
        for (int f = 0; f < 3; f++)
        {
          const uint32_t sh = min((w >> (8 * f)) & 0xFFu, 8u);
          atomicAdd(&s_acc[f], (8u - sh) * n);
          atomicAdd(&s_acc[3 + 9 * f + sh], n);
        }
      }
      __syncthreads();
      if (threadIdx.x < 30 && s_acc[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)s_acc[threadIdx.x]);
    }
  }
  void launch_shift_stats(const uint32_t *dShifts, uint32_t blocksX, uint32_t blocksY, uint32_t rows, uint32_t sizeX, uint32_t sizeY, unsigned long long *dOut30, hipStream_t s)
  {
    const uint64_t total = (uint64_t)blocksX * rows;
    hipLaunchKernelGGL(k_shift_stats, dim3((uint32_t)((total + 4095u) / 4096u)), dim3(256), 0, s, dShifts, blocksX, blocksY, rows, sizeX, sizeY, dOut30);
  }

  void launch_chain_base(const unsigned long long *dCalls, int rank, int world, unsigned long long *dBase, uint32_t *dAborted, hipStream_t s)
  {
    hipLaunchKernelGGL(k_chain_base, dim3(1), dim3(1), 0, s, dCalls, rank, world, dBase, dAborted);
  }

  void launch_strip_scan(const EncodeParams &p, hipStream_t s) { hipLaunchKernelGGL(k_strip_scan, dim3(1), dim3(1024), 0, s, p); }

  void launch_dither_store(const EncodeParams &p, int channels, hipStream_t s)
  {
    const dim3 grid(p.stripsX * p.blocksY), block(kThreads);
    if (channels == 4) hipLaunchKernelGGL(k_dither_store<4>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_dither_store<3>, grid, block, 0, s, p);
  }
} // namespace limg_hip

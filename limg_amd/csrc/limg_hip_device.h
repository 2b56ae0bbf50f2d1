// limg_hip_device.h -- device helpers shared by the kernel files (limg_hip_kernels.hip: 8x8 blocks; limg_hip_blocked.hip: merged regions):
// wave64 reductions, the x86 float semantics of the reference's SSE path (DPPS order, RSQRTPS table, sign-normalised unit vectors), the generic
// 32-bit bit-crush trial (a9) and the literal shift searches (a10-a12).  Everything lives in an anonymous namespace: each including
// translation unit gets its own copy (including the 4 KiB RSQRTPS table).  Reference file:line citations are at each function.
#ifndef LIMG_HIP_DEVICE_H
#define LIMG_HIP_DEVICE_H

#include "limg_hip_internal.h"
#include "limg_rsqrt_x86_table.h"

#include <float.h>

namespace limg_hip
{
  namespace
  {
    __device__ __attribute__((aligned(16))) const unsigned short d_rsqrt_x86_tab[2048] = LIMG_RSQRT_X86_TAB_INIT;

    enum : uint32_t { kZeroA = 1u, kZeroB = 2u, kZeroC = 4u, kValid = 8u };

    // ---- wave64 helpers -------------------------------------------------------------------------------------------
    __device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

    __device__ __forceinline__ void wave_lds_fence()
    {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    template <int CTRL, int ROWMASK>
    __device__ __forceinline__ int dpp(int oldv, int v) { return __builtin_amdgcn_update_dpp(oldv, v, CTRL, ROWMASK, 0xF, false); }

    // integer sum over the wave; result is wave-uniform
    __device__ __forceinline__ uint32_t wave_sum(uint32_t x)
    {
      int v = (int)x;
      v += dpp<0xB1, 0xF>(0, v);  // quad_perm [1,0,3,2]
      v += dpp<0x4E, 0xF>(0, v);  // quad_perm [2,3,0,1]
      v += dpp<0x141, 0xF>(0, v); // row_half_mirror
      v += dpp<0x140, 0xF>(0, v); // row_mirror
      v += dpp<0x142, 0xA>(0, v); // row_bcast:15
      v += dpp<0x143, 0xC>(0, v); // row_bcast:31
      return (uint32_t)__builtin_amdgcn_readlane(v, 63);
    }

    __device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ float sgprf(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

    // ---- x86 float semantics --------------------------------------------------------------------------------------
    // DPPS 0xFF / 0x7F
    template <int CH>
    __device__ __forceinline__ float dpps(const float a[4], const float b[4])
    {
      const float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2];
      const float p3 = CH == 4 ? a[3] * b[3] : 0.0f;
      return (p0 + p1) + (p2 + p3);
    }

    __device__ __forceinline__ void px_to_float(uint32_t px, float f[4])
    {
      f[0] = (float)(px & 0xFF); f[1] = (float)((px >> 8) & 0xFF); f[2] = (float)((px >> 16) & 0xFF); f[3] = (float)(px >> 24);
    }

    __device__ __forceinline__ int cvt_rne(float x) { return (int)__builtin_rintf(x); }
    // clamp(round-to-nearest-even(x), 0, 255), NaN -> 0, in one instruction: what CVTPS2DQ + the saturating packs of a8 (src/limg_factorization.h:98-197) give.
    // Checked against nearbyintf + clamp on 2 M inputs incl. every tie and the specials (tools/cvt_pk_u8_check.hip).
    __device__ __forceinline__ uint32_t cvt_u8_rne_sat(float x) { uint32_t r; asm("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(x)); return r; }

    __device__ __forceinline__ void store_v(float *V, int lane, const float v[4])
    {
      *reinterpret_cast<float4 *>(V + lane * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }

    // ---- integer stage ------------------------------------------------------------------------------------------------
    // (1 << s) + decode_bias(s)  (src/limg_bit_crush_simd.h:611-619): 1,2,4,8,17,36,85,255,256
    __device__ __forceinline__ uint32_t shift_mul(uint32_t s)
    {
      const uint64_t biasPacked = (1ull << 28) | (4ull << 35) | (21ull << 42) | (127ull << 49); // 7 bits per shift value
      return (1u << s) + (uint32_t)((biasPacked >> (7 * s)) & 127u);
    }

    struct RecU // wave-uniform integer view of a record for the reconstruct (RGB lanes only; alpha never reaches the trial error)
    {
      int nA[3], nB[3], nC[3], mA[3], mB[3], mC[3];
    };

    // a9, per pixel: the weighted squared error of one pixel for a shift triple (src/limg_bit_crush_simd.h:627-770; alpha never reaches it)
    __device__ __forceinline__ uint32_t trial_error(const uint32_t px, const uint32_t fA, const uint32_t fB, const uint32_t fC, const RecU &r, const uint32_t sA,
                                                    const uint32_t sB, const uint32_t sC)
    {
      const uint32_t dA = (fA >> sA) * shift_mul(sA), dB = (fB >> sB) * shift_mul(sB), dC = (fC >> sC) * shift_mul(sC);
      uint32_t dsq[3];
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
        const int nA = sA > 7 ? 0 : r.nA[c], nB = sB > 7 ? 0 : r.nB[c], nC = sC > 7 ? 0 : r.nC[c];
        const int mA = r.mA[c], mB = sB > 7 ? 128 : r.mB[c], mC = sC > 7 ? 128 : r.mC[c];
        int est = ((int)(dA * (uint32_t)nA + (uint32_t)mA) >> 8) + ((int)(dB * (uint32_t)nB + (uint32_t)mB) >> 8) + ((int)(dC * (uint32_t)nC + (uint32_t)mC) >> 8);
        est = est < 0 ? 0 : (est > 255 ? 255 : est);
        const int d = (int)((px >> (8 * c)) & 0xFF) - est;
        dsq[c] = (uint32_t)(d * d);
      }
      const bool low_red = (int)dsq[0] < 0x4000;
      return dsq[0] * (low_red ? 2u : 3u) + dsq[2] * (low_red ? 3u : 2u) + dsq[1] * 4u;
    }

    // a9: one bit-crush trial of a block of <= 64 pixels (lane = pixel).  Returns pass / fail (wave-uniform); blockError valid on pass.
    __device__ __forceinline__ bool trial(const uint32_t px, const uint32_t fA, const uint32_t fB, const uint32_t fC, const RecU &r, const uint32_t sA,
                                          const uint32_t sB, const uint32_t sC, const bool active, const uint32_t maxPixel32, const uint64_t maxBlockN,
                                          uint32_t &blockError)
    {
      uint32_t err = trial_error(px, fA, fB, fC, r, sA, sB, sC);
      err = active ? err : 0u;
      const bool any_fail = __builtin_amdgcn_ballot_w64(err > maxPixel32) != 0ull;
      const uint32_t be = wave_sum(err);
      blockError = be;
      return !any_fail && ((uint64_t)be * 16ull < maxBlockN);
    }

    // 24-bit integer multiplies (full rate; v_mul_lo_u32 is quarter rate).  Operands here always fit: see kRecordLimit.
    __device__ __forceinline__ int mul_i24(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    __device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    // same with a wave-uniform factor straight from its SGPR (src0 of the VOP2 form): no v_mov to bring it into a VGPR first
    __device__ __forceinline__ uint32_t mul_u24_uniform(uint32_t a, uint32_t uniformB) { uint32_t r; asm("v_mul_u32_u24 %0, %2, %1" : "=v"(r) : "v"(a), "s"(uniformB)); return r; }
    __device__ __forceinline__ int med3_i32(int a, int b, int c) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
    __device__ __forceinline__ int mad_i24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

    // a10-a12 search driver; everything in here is wave-uniform
    // BE = type of the block error (32 bits suffice for an 8x8 block; merged regions of any size use 64)
    template <typename BE = uint32_t, typename TRY>
    __device__ __forceinline__ void search_fast(TRY &&T, uint32_t shift[3])
    {
      BE be;
      // guess, src/limg_bit_crush.h:331-392
      if (T(4, 5, 6, be))
      {
        shift[0] = 4; shift[1] = 5; shift[2] = 6;
        if (T(5, 8, 8, be)) { shift[0] = 5; shift[1] = 8; shift[2] = 8; }
        else if (T(4, 6, 8, be)) { shift[0] = 4; shift[1] = 6; shift[2] = 8; }
      }
      else if (T(2, 4, 5, be)) { shift[0] = 2; shift[1] = 4; shift[2] = 5; }

      // stepwise, src/limg_bit_crush.h:502-614 (uint8_t counters upstream; values stay < 16 here so plain ints behave identically)
      uint32_t max_shift = shift[0] + shift[1] + shift[2];
      {
        uint32_t a = shift[0] & 15, b = shift[1] & 15, c = (shift[2] & 15) + 2;
        for (; a <= 8; a += 2)
        {
          for (; b <= 8; b += 2)
          {
            for (; c <= 8; c += 2)
            {
              if (a + b + c > max_shift)
              {
                if (T(a, b, c, be)) { shift[0] = a; shift[1] = b; shift[2] = c; max_shift = a + b + c; }
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
      {
        const uint32_t pre_a = shift[0], pre_b = shift[1], pre_c = shift[2];
        const uint32_t max_a = (!(pre_a & 1) && pre_a != 8) ? 1 : 0, max_b = (!(pre_b & 1) && pre_b != 8) ? 1 : 0, max_c = (!(pre_c & 1) && pre_c != 8) ? 1 : 0;
        uint32_t fine = 0, a = 0, b = 0, c = 1;
        for (; a <= max_a; a++)
        {
          for (; b <= max_b; b++)
          {
            for (; c <= max_c; c++)
            {
              if (a + b + c > fine)
              {
                if (T(pre_a + a, pre_b + b, pre_c + c, be)) { shift[0] = pre_a + a; shift[1] = pre_b + b; shift[2] = pre_c + c; fine = a + b + c; }
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
    }

    template <typename BE = uint32_t, typename TRY>
    __device__ __forceinline__ void search_accurate(TRY &&T, uint32_t shift[3])
    {
      // src/limg_bit_crush.h:668-830
      BE be, min_be = ~(BE)0;
      uint32_t max_shift = 0;
      bool have = false; // min_block_error == (size_t)-1 upstream
      if (T(4, 5, 6, be))
      {
        shift[0] = 4; shift[1] = 5; shift[2] = 6; max_shift = 15; min_be = be; have = true;
        if (T(5, 8, 8, be)) { shift[0] = 5; shift[1] = 8; shift[2] = 8; max_shift = 21; min_be = be; }
        else if (T(4, 6, 8, be)) { shift[0] = 4; shift[1] = 6; shift[2] = 8; max_shift = 18; min_be = be; }
      }
      else if (T(2, 4, 5, be)) { shift[0] = 2; shift[1] = 4; shift[2] = 5; max_shift = 11; min_be = be; have = true; }
      {
        uint32_t a = 0, b = 0, c = 1;
        for (; a <= 8; a++)
        {
          for (; b <= 8; b++)
          {
            while (c <= 8)
            { // upstream's `for (; c <= 8; c++)`; the iterations that cannot try anything (a + b + c <= max_shift) are stepped over at once -- walking them
              // one by one made this scalar loop, not the trials, the cost of the accurate mode
              if (a + b + c <= max_shift)
              {
                const uint32_t first = max_shift - a - b + 1; // > c
                c = first > 8 ? 9 : first;
                continue;
              }
              if (a != shift[0] || b != shift[1] || c != shift[2])
              {
                if (T(a, b, c, be)) { shift[0] = a; shift[1] = b; shift[2] = c; max_shift = a + b + c; min_be = be; have = true; }
                else
                  break;
              }
              c++;
            }
            if (c == 0) break;
            c = 0;
          }
          if (b == 0) break;
          b = 0;
        }
      }
      if (max_shift > 0)
      {
        uint32_t a = shift[0], b = shift[1], c = shift[2] + 1;
        for (; a <= 8; a++)
        {
          for (; b <= 8; b++)
          {
            while (c <= 8)
            { // as above: only c == max_shift - a - b can try anything (max_shift is fixed in this phase)
              const uint32_t sum = a + b + c;
              if (sum != max_shift)
              {
                const uint32_t only = max_shift - a - b; // meaningful when sum < max_shift
                c = (sum < max_shift && only <= 8) ? only : 9;
                continue;
              }
              if (T(a, b, c, be))
              {
                if (!have || min_be > be) { shift[0] = a; shift[1] = b; shift[2] = c; min_be = be; have = true; }
              }
              else
                break;
              c++;
            }
            if (c == 0) break;
            c = 0;
          }
          if (b == 0) break;
          b = 0;
        }
      }
    }

    // exact min / max over the wave of two values at once (no NaN present); results wave-uniform.
    // Hand-written DPP: the two chains interleave so each needs only one wait state between dependent steps.
    __device__ __forceinline__ void wave_min_max(float &mn, float &mx)
    {
      asm volatile(
          "s_nop 1\n\t"
          "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 0\n\t"
          "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 0\n\t"
          "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 0\n\t"
          "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 0\n\t"
          "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
          "s_nop 0\n\t"
          "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
          "s_nop 1"
          : "+v"(mn), "+v"(mx));
      mn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mn), 63));
      mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 63));
    }

    __device__ __forceinline__ float vmin_(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    __device__ __forceinline__ float vmax_(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

    // Exact min and max of FOUR independent value pairs over the wave at once, results wave-uniform.  gfx950's lane-swap instructions fold the values into
    // one register on the way down: v_permlane32_swap puts the two halves of a pair of values side by side (one op then reduces both from 64 to 32 lanes),
    // v_permlane16_swap does the same for the two pairs (32 -> 16 lanes), and four DPP steps finish all four 16-lane rows together: 10 VALU instructions
    // per four values instead of 24.  The min and the max chain are interleaved so that each DPP step needs only one wait state.
    __device__ __forceinline__ void wave_reduce4_min_max(float mn[4], float mx[4])
    {
      auto swap32 = [](float a, float b, float &x, float &y) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false); x = __uint_as_float(r[0]); y = __uint_as_float(r[1]); };
      auto swap16 = [](float a, float b, float &x, float &y) { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false); x = __uint_as_float(r[0]); y = __uint_as_float(r[1]); };
      float x, y;
      swap32(mn[0], mn[1], x, y); const float n01 = vmin_(x, y); // lanes 0..31: value 0 (64 -> 32 lanes), lanes 32..63: value 1
      swap32(mx[0], mx[1], x, y); const float x01 = vmax_(x, y);
      swap32(mn[2], mn[3], x, y); const float n23 = vmin_(x, y);
      swap32(mx[2], mx[3], x, y); const float x23 = vmax_(x, y);
      swap16(n01, n23, x, y); float n = vmin_(x, y); // rows of 16 lanes: values 0, 2, 1, 3
      swap16(x01, x23, x, y); float m = vmax_(x, y);
      asm volatile(
          "s_nop 1\n\t"
          "v_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 0\n\t"
          "v_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 0\n\t"
          "v_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 0\n\t"
          "v_min_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
          "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1"
          : "+v"(n), "+v"(m));
      const int rows[4] = { 0, 32, 16, 48 }; // value i sits in the row starting at lane rows[i]
#pragma unroll
      for (int i = 0; i < 4; i++)
      {
        mn[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(n), rows[i]));
        mx[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), rows[i]));
      }
    }

    __device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
    __device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
    __device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    __device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

    // ---- 4-channel float vectors in "DPPS order" -----------------------------------------------------------------------------
    // A pixel-space vector lives in two register pairs a = (x0, x2), b = (x1, x3): then DPPS's (x0y0 + x1y1) + (x2y2 + x3y3)
    // is  m = a*a', n = b*b' (v_pk_mul_f32), s = m + n = (p0 + p1, p2 + p3) (v_pk_add_f32), s.x + s.y  -- 4 instructions, each
    // product and sum rounded exactly like the SSE code.  The LDS copies (avg, dirA..C, est0, parked contributions) use the same
    // slot order [x0, x2, x1, x3]; `slot_of` maps a channel to its slot.
    typedef float float2_t __attribute__((ext_vector_type(2)));
    struct V4 { float2_t a, b; };
    __device__ __forceinline__ constexpr int slot_of(int c) { return c == 1 ? 2 : (c == 2 ? 1 : c); }
    __device__ __forceinline__ V4 ld4(const float *p) { const float4 v = *reinterpret_cast<const float4 *>(p); V4 r; r.a = float2_t{ v.x, v.y }; r.b = float2_t{ v.z, v.w }; return r; }
    __device__ __forceinline__ void st4(float *p, const V4 &v) { *reinterpret_cast<float4 *>(p) = make_float4(v.a.x, v.a.y, v.b.x, v.b.y); }
    __device__ __forceinline__ V4 px_to_v4(uint32_t px)
    {
      V4 r;
      r.a = float2_t{ (float)(px & 0xFF), (float)((px >> 16) & 0xFF) };
      r.b = float2_t{ (float)((px >> 8) & 0xFF), (float)(px >> 24) };
      return r;
    }
    __device__ __forceinline__ V4 operator-(const V4 &x, const V4 &y) { V4 r; r.a = x.a - y.a; r.b = x.b - y.b; return r; }
    __device__ __forceinline__ V4 operator+(const V4 &x, const V4 &y) { V4 r; r.a = x.a + y.a; r.b = x.b + y.b; return r; }
    __device__ __forceinline__ V4 operator*(const V4 &x, float s) { V4 r; r.a = x.a * s; r.b = x.b * s; return r; }
    // FAST (limg_hip_options.float_mode = 1): the same dot product with the second pair of products fused into the first (v_pk_fma_f32): one
    // instruction and two roundings fewer; not the reference's bits, covered by the FAST-mode tolerance contract (DESIGN.md "numerics").
    // the last add of a dot product, kept out of the vectoriser's sight: paired with a neighbouring pixel's it becomes one v_pk_add_f32 plus three v_mov to
    // line the operands up -- dearer than the two full-rate v_add_f32 it replaces
    __device__ __forceinline__ float hadd(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    template <int CH, bool FAST = false>
    __device__ __forceinline__ float dp4(const V4 &x, const V4 &y)
    {
      if (FAST)
      {
        float2_t yb = y.b;
        if (CH == 3) yb.y = 0.0f;
        const float2_t s2 = __builtin_elementwise_fma(x.b, yb, x.a * y.a);
        return s2.x + s2.y;
      }
      const float2_t m = x.a * y.a;
      float2_t n = x.b * y.b;
      if (CH == 3) n.y = 0.0f; // DPPS mask 0x7F
      const float2_t s2 = m + n;
      return hadd(s2.x, s2.y);
    }
    template <int CH>
    __device__ __forceinline__ void mask_alpha(V4 &v) { if (CH == 3) v.b.y = 0.0f; }

    // Sign-normalised unit vector of one pixel's difference vector (src/limg_factorization.h:605-623 and its twins in every
    // pass): bias each lane by {3e,2e,e,0}, flip the sign when |min over lanes| > max over lanes, scale by RSQRTPS(d.d).
    // min/max by v_min3/v_max3: their only differences from MINPS/MAXPS are NaN handling and the sign of a zero result,
    // neither of which can reach the comparison's outcome; the NaN-producing degenerate cases never get here (kZero* flags).
    // RSQRTPS = the captured Intel table: index = [exponent lsb : top 10 mantissa bits], exponent = 126 - floor((e - 127) / 2).
    // FAST: the hardware's v_rsq_f32 (1 ulp) instead of the 12-bit x86 table -- unit vectors 2^-12 closer to unit length than the reference's.
    // ZERO_BY_LEN2: decide "all lanes zero" from d . d != 0 instead of OR-ing the four bit patterns (three instructions fewer).  Equivalent for everything the
    // fit can feed it: a non-zero difference of a byte pixel and a float average / estimate is far above 1e-19, so its square does not underflow.  Used by
    // k_fit_tpb (which has the registers for it); the lane == pixel path keeps the bit test (there it cost a spill).
    // TAB32 (k_fit_tpb's LDS copy): `tab` points at 2048 dwords T[j] = table[j ^ 0x400] << 11 -- entry already in mantissa position, index without the flip --
    // and the exponent comes from one subtraction: (0x5f3fffff - (bits >> 1)) & 0xff800000 == ((380 - e) >> 1) << 23 (the constant's clear bit 22 turns an odd
    // exponent into the borrow the floor needs; the mantissa bits below never borrow).  Five instructions instead of eight, same bits.
    template <int CH, bool FAST = false, bool ZERO_BY_LEN2 = false, bool TAB32 = false>
    __device__ __forceinline__ V4 unit4(const unsigned short *tab, const V4 &d, bool active)
    {
      const float2_t biasA = { FLT_EPSILON * 3, FLT_EPSILON * 1 }, biasB = { FLT_EPSILON * 2, 0.0f };
      const float2_t mbA = d.a - biasA, mbB = d.b - biasB, xbA = d.a + biasA, xbB = d.b + biasB;
      const float mn = vmin3(mbA.x, mbA.y, vmin(mbB.x, mbB.y));
      const float mx = vmax3(xbA.x, xbA.y, vmax(xbB.x, xbB.y));
      const float len2 = dp4<CH, FAST>(d, d);
      bool use;
      // (EXACT + ZERO_BY_LEN2: no zero test at all -- len2 == 0 means d == 0 here, the table's value for 0 is finite, and 0 * finite adds nothing to a sum
      //  that starts at +0, bit for bit what selecting 0 does.  FAST: v_rsq_f32(0) is infinite, so the test stays.)
      if (ZERO_BY_LEN2) use = FAST ? ((len2 != 0.0f) && active) : active;
      else
      {
        const uint32_t anybits = (__float_as_uint(d.a.x) | __float_as_uint(d.a.y) | __float_as_uint(d.b.x) | __float_as_uint(d.b.y)) << 1;
        use = (anybits != 0u) && active;
      }
      float inv;
      if (FAST) inv = __builtin_amdgcn_rsqf(len2);
      else
      {
        // RSQRTPS table lookup; for skipped lanes len2 == 0 => index 0x400, exponent garbage: result discarded below
        const uint32_t bits = __float_as_uint(len2);
        if (TAB32)
        {
          const uint32_t tvs = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(tab) + ((bits >> 11) & 0x1FFCu));
          inv = __uint_as_float(((0x5f3fffffu - (bits >> 1)) & 0xff800000u) | tvs);
        }
        else
        {
        const uint32_t idx2 = ((bits >> 12) & 0xFFEu) ^ 0x800u; // byte offset into the u16 table
        const uint32_t tv = *reinterpret_cast<const unsigned short *>(reinterpret_cast<const unsigned char *>(tab) + idx2);
        const uint32_t ex = (380u - (bits >> 23)) >> 1; // 126 - floor((e - 127) / 2)
        inv = __uint_as_float((ex << 23) | (tv << 11));
        }
      }
      inv = (-mn > mx) ? -inv : inv; // |min| > max  (min >= 0 can never satisfy either form)
      inv = use ? inv : 0.0f;
      return d * inv;
    }

  }
}

#endif

// limg_hip_fit_tpb.hip -- the float stage (a4-a6: channel sums, 3/4-pass direction fit, extrema, record) with ONE LANE PER 8x8 BLOCK.
//
// reference: limg_encode_sum_to_decomposition_state src/limg.cpp:466-497, limg_encode_get_block_factors_accurate_from_state_3d_{3,4}
// src/limg_factorization.h:382-576 / :578-794, record rounding :764-790.
//
// Why a second mapping.  With lane == pixel (limg_hip_kernels.hip) every per-block quantity of the fit is a wave reduction: two channel sums, six extrema and --
// the expensive one, because the reference accumulates in pixel order -- three direction sums that have to be parked in LDS and walked by 16 lanes.  The float
// stage is a chain of four passes whose per-pixel arithmetic is tiny next to that bookkeeping (~510 of the kernel's ~1060 VALU instructions per block).  With
// lane == block the same per-pixel arithmetic is issued once per pixel for 64 blocks (identical cost per block), and every reduction disappears: a direction sum
// is four plain `v_add_f32` per pixel in the natural loop order (which IS the reference's order), extrema are `v_min/v_max` in the loop, per-block "uniform" values
// are ordinary registers.  No barrier, no DPP, no readlane: a workgroup is one wave that owns 64 adjacent blocks (512 x 8 pixels).
// The per-pixel helpers (unit4, dp4: DPPS order, captured RSQRTPS table, no contraction) are the very functions of limg_hip_device.h, so the records are
// bit-identical to the lane == pixel path's -- the parity tests run both.  Only whole 8x8 blocks: images with partial edge blocks keep the other path.
//
// LDS (only when the input rows are not 16-byte aligned; see DIRECT below): the 64 blocks' pixels, block-major with a stride of 68 dwords (16-byte aligned rows; 68 = 4 mod 64 makes every 16-lane group of a ds_read_b128 hit 64
// distinct banks).  17 KiB per wave.  The passes re-read the pixels from LDS (2 x ds_read_b128 per row).
#include "limg_hip_device.h"

namespace limg_hip
{
  namespace
  {
    constexpr int kTpbStride = 68;

    // DIRECT (input rows 16-byte aligned, the normal case): every lane reads its block's rows straight from global memory (two 16-byte loads per row and pass;
    // the rows are re-read from L2 by passes 2-4) -- no LDS but the RSQRTPS table (8 KiB per workgroup of four waves), so the registers (166 => 3 waves per SIMD
    // = 12 per CU) set the occupancy.  Otherwise the 64 blocks' pixels are staged in LDS first (dword loads), 17 KiB per wave, two waves per workgroup
    // (2 x 17 + 8 KiB => 3 workgroups = 6 waves per CU).  Occupancy is not what the kernel lacks: forced down to 61 VGPRs (8 waves per SIMD) it runs at the same
    // speed -- it is issue-bound on its instruction count.
#ifndef LIMG_TPB_WG_WAVES
#define LIMG_TPB_WG_WAVES 4
#endif
    template <bool DIRECT> constexpr int tpb_waves() { return DIRECT ? LIMG_TPB_WG_WAVES : 2; } // waves per workgroup: they share one copy of the table

    // The eight rows of a lane's block, two per iteration, the NEXT two already requested: a pass is a chain of (load 32 bytes, 8 pixels of arithmetic) per row, and
    // with few waves on a SIMD -- a 4096^2 image is one round of 4 waves per SIMD, the sub-batch pipeline leaves this kernel one -- nothing else covers the load's
    // latency.  (The last iteration re-requests rows 6 and 7: harmless, and the loop stays free of a branch.)
#ifndef LIMG_TPB_WAVES_PER_SIMD
#define LIMG_TPB_WAVES_PER_SIMD 5
#endif
    template <class BODY>
    __device__ __forceinline__ void for_rows(const uint32_t *my, const uint32_t pitch, BODY &&body)
    {
#ifdef LIMG_TPB_NO_PREFETCH // A/B: the round-3 form, one row per iteration, loaded where it is used
#pragma unroll 1
      for (int r = 0; r < 8; r++)
      {
        const uint4 u = *reinterpret_cast<const uint4 *>(my + r * pitch), w = *reinterpret_cast<const uint4 *>(my + r * pitch + 4);
        const uint32_t q[8] = { u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w };
        body(r, q);
      }
      return;
#endif
      uint4 u0 = *reinterpret_cast<const uint4 *>(my), w0 = *reinterpret_cast<const uint4 *>(my + 4);
      uint4 u1 = *reinterpret_cast<const uint4 *>(my + pitch), w1 = *reinterpret_cast<const uint4 *>(my + pitch + 4);
#pragma unroll 1
      for (int r = 0; r < 8; r += 2)
      {
        const uint32_t *nx = my + (r + 2 < 8 ? r + 2 : 6) * pitch;
        const uint4 nu0 = *reinterpret_cast<const uint4 *>(nx), nw0 = *reinterpret_cast<const uint4 *>(nx + 4);
        const uint4 nu1 = *reinterpret_cast<const uint4 *>(nx + pitch), nw1 = *reinterpret_cast<const uint4 *>(nx + pitch + 4);
        {
          const uint32_t q[8] = { u0.x, u0.y, u0.z, u0.w, w0.x, w0.y, w0.z, w0.w };
          body(r, q);
        }
        {
          const uint32_t q[8] = { u1.x, u1.y, u1.z, u1.w, w1.x, w1.y, w1.z, w1.w };
          body(r + 1, q);
        }
        u0 = nu0; w0 = nw0; u1 = nu1; w1 = nw1;
      }
    }

    template <int CH, bool FAST, bool DIRECT>
    // (8 waves per SIMD asked for explicitly: left to itself the compiler settles on 61 registers or on 132 depending on details of the epilogue; measured equal in
    //  speed on large images -- the kernel is issue-bound -- but a 4096^2 image is a single round of 4096 waves, where residency is what there is)
    __global__ __launch_bounds__(64 * tpb_waves<DIRECT>(), DIRECT ? LIMG_TPB_WAVES_PER_SIMD : 1) void k_fit_tpb(const EncodeParams p)
    {
      constexpr int kTpbWaves = tpb_waves<DIRECT>();
      __shared__ __attribute__((aligned(16))) uint32_t s_pxAll[kTpbWaves][DIRECT ? 4 : 64 * kTpbStride];
      // the RSQRTPS table in LDS: a per-lane gather of 64 unrelated 2-byte entries costs the texture path ~64 address cycles per wave instruction from global
      // memory, but only a few LDS cycles (random banks)
      // 8 KiB: T[j] = table[j ^ 0x400] << 11, the form unit4<..., TAB32> reads (entry in mantissa position, no index flip)
      __shared__ __attribute__((aligned(16))) uint32_t s_tab[FAST ? 4 : 2048];
      const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
      // Next to a persistent kernel (sub-batch pipeline) this kernel has ONE wave per SIMD against the other's five: at equal priority it would get a sixth of the
      // issue slots and become the pipeline's critical path.  Raised above the E step's priority, its one wave takes what a single wave can issue.
      if (p.fitPrio == 3) __builtin_amdgcn_s_setprio(3);
      else if (p.fitPrio == 2) __builtin_amdgcn_s_setprio(2);
      else if (p.fitPrio == 1) __builtin_amdgcn_s_setprio(1);
      if (!FAST)
      {
        const uint4 *src = reinterpret_cast<const uint4 *>(d_rsqrt_x86_tab);
        for (int i = (int)threadIdx.x; i < 256; i += 64 * kTpbWaves)
        {
          const uint4 v = src[i]; // entries 8 i .. 8 i + 7
          const uint32_t q[4] = { v.x, v.y, v.z, v.w };
          uint32_t *dst = s_tab + ((8 * i) ^ 0x400);
#pragma unroll
          for (int k = 0; k < 4; k++) { dst[2 * k] = (q[k] & 0xFFFFu) << 11; dst[2 * k + 1] = (q[k] >> 16) << 11; }
        }
        __syncthreads();
      }
      const unsigned short *tab = reinterpret_cast<const unsigned short *>(s_tab);
      uint32_t *s_px = s_pxAll[wave];
      const uint32_t unitsX = (p.blocksX + 63u) / 64u;
      const uint32_t unitId = blockIdx.x * kTpbWaves + wave;
      if (unitId >= unitsX * p.blocksY * p.batchCount) return; // whole wave; after the only barrier
      const uint32_t unit = unitId % unitsX, byS = unitId / unitsX; // byS: block row counted through all images of a batch (= row in the per-block scratch)
      const uint32_t img = p.batchCount > 1 ? byS / p.blocksY : 0u, by = byS - img * p.blocksY;
      const uint32_t *const pin = p.batchCount > 1 ? p.batch[img].in : p.io.in;
      const uint32_t bx0 = unit * 64u, x0 = bx0 * kBlock, y0 = by * kBlock;
      const uint32_t nBlocks = min(p.blocksX - bx0, 64u), widthPx = nBlocks * kBlock;
      if (p.zeroLookback)
      { // this wave's 64 blocks are two work strips of the persistent kernel: clear their look-back descriptors (and, once, the ticket) instead of a memset launch
        if ((uint32_t)lane < 2u && unit * 2u + (uint32_t)lane < p.stripsX) p.desc[(size_t)byS * p.stripsX + unit * 2u + (uint32_t)lane] = 0ull;
        if (unitId == 0 && lane < 4) p.ticket[lane] = 0u;
      }

      // ---- stage the 8 pixel rows (coalesced 16 bytes per lane) into the block-major layout ----
      if (!DIRECT)
      {
      if (p.vecIn)
      {
#pragma unroll
        for (int row = 0; row < 8; row++)
          for (uint32_t col = (uint32_t)lane * 4u; col < widthPx; col += 256u)
          {
            const uint4 v = *reinterpret_cast<const uint4 *>(pin + (size_t)(y0 + row) * p.sizeX + x0 + col);
            *reinterpret_cast<uint4 *>(&s_px[(col >> 3) * kTpbStride + row * 8 + (col & 7u)]) = v;
          }
      }
      else
      {
        for (int row = 0; row < 8; row++)
          for (uint32_t col = (uint32_t)lane; col < widthPx; col += 64u)
            s_px[(col >> 3) * kTpbStride + row * 8 + (col & 7u)] = pin[(size_t)(y0 + row) * p.sizeX + x0 + col];
      }
      }
      wave_lds_fence();
      if ((uint32_t)lane >= nBlocks) return;
      const uint32_t *my = DIRECT ? pin + (size_t)y0 * p.sizeX + x0 + lane * 8 : s_px + lane * kTpbStride;
      const uint32_t pitch = DIRECT ? p.sizeX : 8u;

      // ---- a4: channel sums (src/limg.cpp:466-497); two channels per 32-bit accumulator, 64 * 255 < 2^16 ----
      uint32_t s02 = 0, s13 = 0;
#pragma unroll
      for (int r = 0; r < 8; r++)
      {
        const uint4 u = *reinterpret_cast<const uint4 *>(my + r * pitch), w = *reinterpret_cast<const uint4 *>(my + r * pitch + 4);
        const uint32_t q[8] = { u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w };
#pragma unroll
        for (int i = 0; i < 8; i++) { s02 += q[i] & 0x00FF00FFu; s13 += (q[i] >> 8) & 0x00FF00FFu; }
      }
      const float inv_count = 0.015625f;
      V4 avg;
      avg.a = float2_t{ (float)(int)(s02 & 0xFFFF), (float)(int)(s02 >> 16) } * inv_count;
      avg.b = float2_t{ (float)(int)(s13 & 0xFFFF), CH == 4 ? (float)(int)(s13 >> 16) : 0.0f } * inv_count;
      const V4 zero4 = { float2_t{ 0.0f, 0.0f }, float2_t{ 0.0f, 0.0f } };

      // the epilogue of a direction sum: dir = sum / N, 1 / (dir . dir) in DPPS order, all-zero test (== serial_sums2 of limg_hip_kernels.hip)
      auto finish_dir = [&](const V4 &acc, V4 &dir, float &inv, bool &zero)
      {
        dir = acc * inv_count;
        const float pp = dp4<CH>(dir, dir); // plain products even in FAST mode: the lane == pixel path does the same
        zero = dir.a.x == 0.0f && dir.a.y == 0.0f && dir.b.x == 0.0f && dir.b.y == 0.0f;
        inv = FAST ? __builtin_amdgcn_rcpf(pp) : 1.0f / pp;
      };

      // ---- pass 1 (src/limg_factorization.h:602-628): sign-normalised unit vectors of px - avg, summed in pixel order ----
      V4 dirA, dirB = zero4, dirC = zero4, est0 = zero4;
      float invA, invB = 0.0f, invC = 0.0f;
      bool zeroA, zeroB = true, zeroC = true;
      float mm[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
      {
        V4 acc = zero4;
        for_rows(my, pitch, [&](const int, const uint32_t (&q)[8])
        {
#pragma unroll
          for (int i = 0; i < 8; i++)
          {
            V4 d = px_to_v4(q[i]) - avg;
            mask_alpha<CH>(d);
            acc = acc + unit4<CH, FAST, true, true>(tab, d, true);
          }
        });
        finish_dir(acc, dirA, invA, zeroA);
      }

      // ---- pass 2 (:652-688): factor A extrema, residual -> second direction ----
      if (!zeroA)
      {
        V4 acc = zero4;
        float mn = 0.0f, mx = 0.0f; // upstream starts them at 0 (:633-634)
        for_rows(my, pitch, [&](const int, const uint32_t (&q)[8])
        {
#pragma unroll
          for (int i = 0; i < 8; i++)
          {
            const V4 pf = px_to_v4(q[i]);
            const float fA = dp4<CH, FAST>(pf - avg, dirA) * invA;
            mn = vmin(mn, fA); mx = vmax(mx, fA);
            V4 e = pf - (avg + dirA * fA);
            mask_alpha<CH>(e);
            acc = acc + unit4<CH, FAST, true, true>(tab, e, true);
          }
        });
        mm[0] = mn; mm[1] = mx;
        finish_dir(acc, dirB, invB, zeroB);
      }

      // ---- pass 3 ----
      if (!zeroA && !zeroB)
      {
        float mnB = FLT_MAX, mxB = -FLT_MAX;
        if (CH == 4)
        { // :701-738: factor B extrema, residual -> third direction; the A+B estimate of pixel 0 is what pass 4 measures every pixel against (:748-758)
          V4 acc = zero4;
          for_rows(my, pitch, [&](const int r, const uint32_t (&q)[8])
          {
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
              const V4 pf = px_to_v4(q[i]);
              const float fA = dp4<CH, FAST>(pf - avg, dirA) * invA; // recomputed, same operations => same bits as pass 2's
              const V4 estA = avg + dirA * fA;
              const float fB = dp4<CH, FAST>(pf - estA, dirB) * invB;
              mnB = vmin(mnB, fB); mxB = vmax(mxB, fB);
              const V4 estB = estA + dirB * fB;
              if (r == 0 && i == 0) est0 = estB;
              acc = acc + unit4<CH, FAST, true, true>(tab, pf - estB, true);
            }
          });
          mm[2] = mnB; mm[3] = mxB;
          finish_dir(acc, dirC, invC, zeroC);
          // ---- pass 4 (:748-758) ----
          if (!zeroC)
          {
            float mnC = FLT_MAX, mxC = -FLT_MAX;
            for_rows(my, pitch, [&](const int, const uint32_t (&q)[8])
            {
#pragma unroll
              for (int i = 0; i < 8; i++)
              {
                const float fC = dp4<CH, FAST>(px_to_v4(q[i]) - est0, dirC) * invC;
                mnC = vmin(mnC, fC); mxC = vmax(mxC, fC);
              }
            });
            mm[4] = mnC; mm[5] = mxC;
          }
        }
        else
        { // 3 channels (:498-541): dirC = dirA x dirB, B and C extrema in one pass; slots: a = (x0, x2), b = (x1, x3)
          dirC.a.x = dirA.b.x * dirB.a.y - dirA.a.y * dirB.b.x; // A1 B2 - A2 B1
          dirC.b.x = dirA.a.y * dirB.a.x - dirA.a.x * dirB.a.y; // A2 B0 - A0 B2
          dirC.a.y = dirA.a.x * dirB.b.x - dirA.b.x * dirB.a.x; // A0 B1 - A1 B0
          dirC.b.y = 0.0f;
          zeroC = dirC.a.x == 0.0f && dirC.b.x == 0.0f && dirC.a.y == 0.0f;
          if (!zeroC) invC = FAST ? __builtin_amdgcn_rcpf(dp4<CH, FAST>(dirC, dirC)) : 1.0f / dp4<CH, FAST>(dirC, dirC);
          float mnC = zeroC ? 0.0f : FLT_MAX, mxC = zeroC ? 0.0f : -FLT_MAX;
          for_rows(my, pitch, [&](const int, const uint32_t (&q)[8])
          {
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
              const V4 pf = px_to_v4(q[i]);
              const float fA = dp4<CH, FAST>(pf - avg, dirA) * invA;
              const V4 estA = avg + dirA * fA;
              const float fB = dp4<CH, FAST>(pf - estA, dirB) * invB;
              mnB = vmin(mnB, fB); mxB = vmax(mxB, fB);
              if (!zeroC)
              {
                const V4 e = pf - (estA + dirB * fB);
                const float fC = dp4<CH, FAST>(e, dirC) * invC;
                mnC = vmin(mnC, fC); mxC = vmax(mxC, fC);
              }
            }
          });
          mm[2] = mnB; mm[3] = mxB; mm[4] = mnC; mm[5] = mxC;
        }
      }

      // ---- record (src/limg_factorization.h:764-790): avg + extreme * dir for A, extreme * dir for B and C, round to nearest even, int16 ----
      const bool deadA = zeroA, deadB = zeroA || zeroB, deadC = zeroA || zeroB || zeroC;
      auto comp = [](const V4 &v, int c) -> float { return c == 0 ? v.a.x : (c == 1 ? v.b.x : (c == 2 ? v.a.y : v.b.y)); };
      uint32_t words[16];
#pragma unroll
      for (int c = 0; c < 4; c++) words[c] = __float_as_uint(comp(avg, c));
      // Factor by factor, so that nothing but the factor's eight values is live: the record's int16 pairs, and -- a7 (limg_init_color_error_state_3d,
      // src/limg_internal.h:426-452) -- 1 / (n . n) of the INTEGER normal max - min in the serial limg_dot order, 0 for an all-zero normal.  Here that is three
      // divisions per 64 blocks; in the E step (lane == pixel, one lane per (block, factor, channel)) it was four DPP broadcasts and a correctly rounded division per
      // lane and strip: ~7 vector instructions per block.
      float inv[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
      {
        const bool dead = r == 0 ? deadA : (r == 1 ? deadB : deadC);
        const V4 &dir = r == 0 ? dirA : (r == 1 ? dirB : dirC);
        int16_t lo[4], hi[4];
        float sum = 0.0f;
        bool nz = false;
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          float dv = comp(dir, c), mLo = mm[2 * r], mHi = mm[2 * r + 1];
          if (dead) { mLo = 0.0f; mHi = 0.0f; dv = 0.0f; }
          float vLo = mLo * dv, vHi = mHi * dv;
          if (r == 0) { vLo = comp(avg, c) + vLo; vHi = comp(avg, c) + vHi; }
          int qLo = cvt_rne(vLo), qHi = cvt_rne(vHi);
          if (CH == 3 && c == 3) { qLo = 0; qHi = 0; }
          lo[c] = (int16_t)qLo; hi[c] = (int16_t)qHi;
          if (CH == 4 || c < 3)
          {
            const float n = (float)((int)hi[c] - (int)lo[c]);
            const float sq = n * n;
            sum = sum + sq; // ((0 + s0) + s1) + s2 (+ s3)
            nz = nz || sq != 0.0f;
          }
        }
        inv[r] = nz ? (FAST ? __builtin_amdgcn_rcpf(sum) : 1.0f / sum) : 0.0f;
        words[4 + 4 * r] = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
        words[5 + 4 * r] = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
        words[6 + 4 * r] = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
        words[7 + 4 * r] = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
      }
      uint4 *dst = reinterpret_cast<uint4 *>(p.records + (size_t)byS * p.blocksX + bx0 + lane);
#pragma unroll
      for (int i = 0; i < 4; i++) dst[i] = make_uint4(words[4 * i], words[4 * i + 1], words[4 * i + 2], words[4 * i + 3]);
      reinterpret_cast<float4 *>(p.invN)[(size_t)byS * p.blocksX + bx0 + lane] = make_float4(inv[0], inv[1], inv[2], 0.0f);
    }
  }

  namespace
  {
    // the image table of a batched encode travels to the device inside kernel arguments (copied at launch: no host buffer has to outlive the call, no blocking copy)
    struct TableChunk { ImageIO e[32]; uint32_t n; };
    static_assert(sizeof(TableChunk) <= 3584, "kernel-argument segment");
    __global__ void k_set_batch_table(ImageIO *dst, const TableChunk chunk)
    {
      const uint32_t *src = reinterpret_cast<const uint32_t *>(chunk.e);
      uint32_t *out = reinterpret_cast<uint32_t *>(dst);
      for (uint32_t i = threadIdx.x; i < chunk.n * (uint32_t)(sizeof(ImageIO) / 4); i += blockDim.x) out[i] = src[i];
    }
  }

  void launch_set_batch_table(ImageIO *dTable, const ImageIO *hTable, size_t count, hipStream_t s)
  {
    for (size_t i0 = 0; i0 < count; i0 += 32)
    {
      TableChunk ch;
      ch.n = (uint32_t)(count - i0 < 32 ? count - i0 : 32);
      for (uint32_t i = 0; i < ch.n; i++) ch.e[i] = hTable[i0 + i];
      for (uint32_t i = ch.n; i < 32; i++) ch.e[i] = ImageIO{};
      hipLaunchKernelGGL(k_set_batch_table, dim3(1), dim3(256), 0, s, dTable + i0, ch);
    }
  }

  void launch_fit_tpb(const EncodeParams &p, int channels, hipStream_t s)
  {
    const uint32_t units = ((p.blocksX + 63u) / 64u) * p.blocksY * p.batchCount;
    const int v = (channels == 4 ? 4 : 0) | (p.floatFast ? 2 : 0) | (p.vecIn ? 1 : 0);
#define LIMG_TPB_LAUNCH(CH, FAST, DIRECT) hipLaunchKernelGGL((k_fit_tpb<CH, FAST, DIRECT>), dim3((units + tpb_waves<DIRECT>() - 1) / tpb_waves<DIRECT>()), dim3(64 * tpb_waves<DIRECT>()), 0, s, p)
    switch (v)
    {
    case 0: LIMG_TPB_LAUNCH(3, false, false); break;
    case 1: LIMG_TPB_LAUNCH(3, false, true); break;
    case 2: LIMG_TPB_LAUNCH(3, true, false); break;
    case 3: LIMG_TPB_LAUNCH(3, true, true); break;
    case 4: LIMG_TPB_LAUNCH(4, false, false); break;
    case 5: LIMG_TPB_LAUNCH(4, false, true); break;
    case 6: LIMG_TPB_LAUNCH(4, true, false); break;
    default: LIMG_TPB_LAUNCH(4, true, true); break;
    }
#undef LIMG_TPB_LAUNCH
  }
}

// limg_hip_blocked.hip -- GPU stages of the merged-block encoder (reference: limg_blocked_encode3d_test, src/limg.cpp:1774-1885, :2329-2453).
//
//   k_blocked_match        the block-similarity predicate `limg_encode_3d_matches` (src/limg.cpp:1137-1268) for every block as seed against
//                          the 18 x 18 blocks around it (offsets -5 .. +12): one wave per seed, lane = candidate, 64 pairs per step, results as ballot masks.
//                          The host's greedy raster merge (limg_hip_blocked_host.cpp) only looks these bits up.
//   k_blocked_fit_search   one wave per rectangle ("region") of the merge: gather (src/limg.cpp:1747-1748), channel sums, direction fit
//                          over all N pixels of the region in row-major order (same functions as for an 8x8 block, src/limg_factorization.h
//                          :578-794 / :382-576 with N = 64 rx ry), per-pixel factors, shift search (a7-a12 on N pixels).
//   k_blocked_store        one wave per region: dither (noise bytes of the host's chain walk), planes (src/limg.cpp:1604-1700), decode (a16).
//
// Lane = pixel of a 64-pixel chunk; a region of N pixels is a loop over ceil(N / 64) chunks.  The three direction sums are serial in pixel
// order upstream; here a chunk's per-pixel unit vectors are parked in LDS (slot-planar, so the walkers read float4s) and 4 lanes -- one per
// channel -- add them to running sums carried from chunk to chunk.  All arithmetic helpers are the ones of the 8x8 kernels (limg_hip_device.h).
// Pass-1 (every 8x8 block's own fit, src/limg.cpp:1088-1119) is the 8x8 path's E step in `fitOnly` mode (limg_hip_kernels.hip).
#include "limg_hip_device.h"

namespace limg_hip
{
  namespace
  {
    __device__ __forceinline__ void scratch_fence()
    { // A wave re-reads global scratch that it alone wrote (other lanes' stores): wait for the stores; the CU's vector L1 is write-through and
      // coherent for accesses from the same CU, which is what workgroup scope expresses.  (Agent scope would write the whole L2 back: the
      // per-XCD L2s are not coherent with each other, and that costs ~10x on this kernel.)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    // ---- similarity predicate ---------------------------------------------------------------------------------------------------------
    struct MState { float nA[4], nB[4], nC[4], invA, invB, invC; };

    template <int CH>
    __device__ __forceinline__ float dot_seq(const float *a, const float *b)
    { // limg_dot (src/limg_internal.h:357-366): sum = 0; sum += a[i] * b[i]
      float sum = 0.0f;
#pragma unroll
      for (int i = 0; i < CH; i++) sum += a[i] * b[i];
      return sum;
    }

    template <int CH>
    __device__ __forceinline__ void m_init(const limg_hip_block_record &r, MState &s)
    { // src/limg_internal.h:426-452
      bool nzA = false, nzB = false, nzC = false;
#pragma unroll
      for (int i = 0; i < 4; i++) { s.nA[i] = 0.0f; s.nB[i] = 0.0f; s.nC[i] = 0.0f; }
#pragma unroll
      for (int i = 0; i < CH; i++)
      {
        s.nA[i] = (float)((int)r.dirA_max[i] - (int)r.dirA_min[i]);
        s.nB[i] = (float)((int)r.dirB_mag[i] - (int)r.dirB_offset[i]);
        s.nC[i] = (float)((int)r.dirC_mag[i] - (int)r.dirC_offset[i]);
        nzA |= s.nA[i] != 0; nzB |= s.nB[i] != 0; nzC |= s.nC[i] != 0;
      }
      s.invA = nzA ? 1.0f / dot_seq<CH>(s.nA, s.nA) : 0.0f;
      s.invB = nzB ? 1.0f / dot_seq<CH>(s.nB, s.nB) : 0.0f;
      s.invC = nzC ? 1.0f / dot_seq<CH>(s.nC, s.nC) : 0.0f;
    }

    template <int CH>
    __device__ __forceinline__ void m_factors(const float *color, const limg_hip_block_record &in, const MState &s, float f[3])
    { // src/limg_factorization.h:9-42
      float t[4], est[4];
#pragma unroll
      for (int i = 0; i < CH; i++) t[i] = color[i] - (float)in.dirA_min[i];
      f[0] = dot_seq<CH>(t, s.nA) * s.invA;
#pragma unroll
      for (int i = 0; i < CH; i++) { est[i] = (float)in.dirA_min[i] + f[0] * s.nA[i]; t[i] = (color[i] - est[i]) - (float)in.dirB_offset[i]; }
      f[1] = dot_seq<CH>(t, s.nB) * s.invB;
#pragma unroll
      for (int i = 0; i < CH; i++) { est[i] = est[i] + f[1] * s.nB[i]; t[i] = (color[i] - est[i]) - (float)in.dirC_offset[i]; }
      f[2] = dot_seq<CH>(t, s.nC) * s.invC;
    }

    // src/limg.cpp:1137-1268; same operations in the same order as blocked_matches_host
    template <int CH>
    __device__ bool m_matches(const limg_hip_block_record &a, const limg_hip_block_record &b)
    {
      MState sa, sb;
      m_init<CH>(a, sa);
      m_init<CH>(b, sb);
      const float w[4] = { 2, 4, 3, 3 };
      float avgDiffSq = 0, lenA[3] = { 3, 3, 3 }, lenB[3] = { 3, 3, 3 };
#pragma unroll
      for (int i = 0; i < CH; i++)
      {
        const float d = a.avg[i] - b.avg[i];
        avgDiffSq += d * d * w[i];
        lenA[0] += (sa.nA[i] * sa.nA[i]) * w[i]; lenB[0] += (sb.nA[i] * sb.nA[i]) * w[i];
        lenA[1] += (sa.nB[i] * sa.nB[i]) * w[i]; lenB[1] += (sb.nB[i] * sb.nB[i]) * w[i];
        lenA[2] += (sa.nC[i] * sa.nC[i]) * w[i]; lenB[2] += (sb.nC[i] * sb.nC[i]) * w[i];
      }
      const float sumA = lenA[0] + lenA[1] + lenA[2], sumB = lenB[0] + lenB[1] + lenB[2];
      const float ratio = (sumA + 1) / (sumB + 1);
      const float maxAvg = (float)(16 * 3 * CH), maxRange = (float)(200 * 3 * CH);
      if (avgDiffSq < maxAvg && sumA < maxRange && sumB < maxRange) return true;
      if (ratio > 1.375f || ratio < (1.f / 1.375f)) return false;
      float invA[3], invB[3];
#pragma unroll
      for (int i = 0; i < 3; i++) { invA[i] = 1.0f / lenA[i]; invB[i] = 1.0f / lenB[i]; }
#pragma unroll
      for (int i = 1; i < 3; i++) { invA[i] *= 2.f; invB[i] *= 2.f; }
      float fb[3];
      m_factors<CH>(a.avg, b, sb, fb);
      const float termB = fabsf(fb[0]) * invB[0] + fabsf(0.5f - fb[1]) * invB[1] + fabsf(0.5f - fb[2]) * invB[2];
      float sum = 0;
#pragma unroll 1
      for (int z = 0; z < 3; z++)
#pragma unroll 1
        for (int y = 0; y < 3; y++)
#ifdef LIMG_MATCH_ROLL_X
#pragma unroll 1
#else
#pragma unroll
#endif
          for (int x = 0; x < 3; x++)
          {
            const float xf = x * 0.5f, yf = y * 0.5f, zf = z * 0.5f;
            float color[4], fa[3];
#pragma unroll
            for (int i = 0; i < CH; i++) color[i] = sb.nA[i] * xf + sb.nB[i] * yf + sb.nC[i] * zf;
            m_factors<CH>(color, a, sa, fa);
            sum += fabsf(fa[0]) * invA[0] + fabsf(0.5f - fa[1]) * invA[1] + fabsf(0.5f - fa[2]) * invA[2];
            sum += termB;
          }
      return sum * (1.f / 27) < 3.0f;
    }

    // The same predicate for two candidates per lane, on float2 operands: every operation is elementwise and rounds exactly like its scalar
    // twin (v_pk_mul_f32 / v_pk_add_f32, IEEE division per component), so both components equal m_matches bit for bit at half the VALU issue.
    template <typename T>
    __device__ __forceinline__ float2_t splat2(T v) { return float2_t{ (float)v, (float)v }; }
    __device__ __forceinline__ float2_t pair2(float x, float y) { return float2_t{ x, y }; }

    template <int CH, typename TA, typename TB>
    __device__ __forceinline__ float2_t dot_seq2(const TA *a, const TB *b)
    {
      float2_t sum = { 0.0f, 0.0f };
#pragma unroll
      for (int i = 0; i < CH; i++) sum += a[i] * b[i];
      return sum;
    }

    // src/limg_factorization.h:9-42 with any mix of uniform (float) and per-candidate (float2) operands
    template <int CH, typename TC, typename TR>
    __device__ __forceinline__ void m_factors2(const TC *color, const TR *mnA, const TR *ofB, const TR *ofC, const TR *nA, const TR *nB, const TR *nC, TR invA, TR invB, TR invC,
                                               float2_t f[3])
    {
      float2_t t[4], est[4];
#pragma unroll
      for (int i = 0; i < CH; i++) t[i] = color[i] - mnA[i];
      f[0] = dot_seq2<CH>(t, nA) * invA;
#pragma unroll
      for (int i = 0; i < CH; i++) { est[i] = mnA[i] + f[0] * nA[i]; t[i] = (color[i] - est[i]) - ofB[i]; }
      f[1] = dot_seq2<CH>(t, nB) * invB;
#pragma unroll
      for (int i = 0; i < CH; i++) { est[i] = est[i] + f[1] * nB[i]; t[i] = (color[i] - est[i]) - ofC[i]; }
      f[2] = dot_seq2<CH>(t, nC) * invC;
    }

    template <int CH>
    __device__ void m_matches_pair(const limg_hip_block_record &a, const limg_hip_block_record &b0, const limg_hip_block_record &b1, bool &r0, bool &r1)
    {
      MState sa;
      m_init<CH>(a, sa);
      float amnA[4], aofB[4], aofC[4];
      float2_t nA[4], nB[4], nC[4], mnA[4], ofB[4], ofC[4], avgB[4];
#pragma unroll
      for (int i = 0; i < CH; i++)
      {
        amnA[i] = (float)a.dirA_min[i]; aofB[i] = (float)a.dirB_offset[i]; aofC[i] = (float)a.dirC_offset[i];
        nA[i] = pair2((float)((int)b0.dirA_max[i] - (int)b0.dirA_min[i]), (float)((int)b1.dirA_max[i] - (int)b1.dirA_min[i]));
        nB[i] = pair2((float)((int)b0.dirB_mag[i] - (int)b0.dirB_offset[i]), (float)((int)b1.dirB_mag[i] - (int)b1.dirB_offset[i]));
        nC[i] = pair2((float)((int)b0.dirC_mag[i] - (int)b0.dirC_offset[i]), (float)((int)b1.dirC_mag[i] - (int)b1.dirC_offset[i]));
        mnA[i] = pair2((float)b0.dirA_min[i], (float)b1.dirA_min[i]);
        ofB[i] = pair2((float)b0.dirB_offset[i], (float)b1.dirB_offset[i]);
        ofC[i] = pair2((float)b0.dirC_offset[i], (float)b1.dirC_offset[i]);
        avgB[i] = pair2(b0.avg[i], b1.avg[i]);
      }
      // candidate states (src/limg_internal.h:426-452): 1 / |n|^2, or 0 for an all-zero normal
      float2_t invBA = 1.0f / dot_seq2<CH>(nA, nA), invBB = 1.0f / dot_seq2<CH>(nB, nB), invBC = 1.0f / dot_seq2<CH>(nC, nC);
      {
        bool zA0 = true, zA1 = true, zB0 = true, zB1 = true, zC0 = true, zC1 = true;
#pragma unroll
        for (int i = 0; i < CH; i++)
        {
          zA0 &= nA[i].x == 0.0f; zA1 &= nA[i].y == 0.0f; zB0 &= nB[i].x == 0.0f; zB1 &= nB[i].y == 0.0f; zC0 &= nC[i].x == 0.0f; zC1 &= nC[i].y == 0.0f;
        }
        invBA = pair2(zA0 ? 0.0f : invBA.x, zA1 ? 0.0f : invBA.y);
        invBB = pair2(zB0 ? 0.0f : invBB.x, zB1 ? 0.0f : invBB.y);
        invBC = pair2(zC0 ? 0.0f : invBC.x, zC1 ? 0.0f : invBC.y);
      }
      const float w[4] = { 2, 4, 3, 3 };
      float lenA[3] = { 3, 3, 3 };
      float2_t avgDiffSq = { 0, 0 }, lenB[3] = { { 3, 3 }, { 3, 3 }, { 3, 3 } };
#pragma unroll
      for (int i = 0; i < CH; i++)
      {
        const float2_t d = a.avg[i] - avgB[i];
        avgDiffSq += d * d * w[i];
        lenA[0] += (sa.nA[i] * sa.nA[i]) * w[i]; lenB[0] += (nA[i] * nA[i]) * w[i];
        lenA[1] += (sa.nB[i] * sa.nB[i]) * w[i]; lenB[1] += (nB[i] * nB[i]) * w[i];
        lenA[2] += (sa.nC[i] * sa.nC[i]) * w[i]; lenB[2] += (nC[i] * nC[i]) * w[i];
      }
      const float sumA = lenA[0] + lenA[1] + lenA[2];
      const float2_t sumB = lenB[0] + lenB[1] + lenB[2];
      const float2_t ratio = (sumA + 1) / (sumB + 1);
      const float maxAvg = (float)(16 * 3 * CH), maxRange = (float)(200 * 3 * CH);
      const bool q0 = avgDiffSq.x < maxAvg && sumA < maxRange && sumB.x < maxRange, q1 = avgDiffSq.y < maxAvg && sumA < maxRange && sumB.y < maxRange;
      const bool j0 = ratio.x > 1.375f || ratio.x < (1.f / 1.375f), j1 = ratio.y > 1.375f || ratio.y < (1.f / 1.375f);
      r0 = q0; r1 = q1;
      if ((q0 || j0) && (q1 || j1)) return;
      float invA[3];
      float2_t invB[3];
#pragma unroll
      for (int i = 0; i < 3; i++) { invA[i] = 1.0f / lenA[i]; invB[i] = 1.0f / lenB[i]; }
#pragma unroll
      for (int i = 1; i < 3; i++) { invA[i] *= 2.f; invB[i] *= 2.f; }
      float2_t fb[3];
      m_factors2<CH>(a.avg, mnA, ofB, ofC, nA, nB, nC, invBA, invBB, invBC, fb);
      const float2_t termB = __builtin_elementwise_abs(fb[0]) * invB[0] + __builtin_elementwise_abs(0.5f - fb[1]) * invB[1] + __builtin_elementwise_abs(0.5f - fb[2]) * invB[2];
      float2_t sum = { 0, 0 };
#pragma unroll 1
      for (int z = 0; z < 3; z++)
#pragma unroll 1
        for (int y = 0; y < 3; y++)
#ifdef LIMG_MATCH_ROLL_X
#pragma unroll 1
#else
#pragma unroll
#endif
          for (int x = 0; x < 3; x++)
          {
            const float xf = x * 0.5f, yf = y * 0.5f, zf = z * 0.5f;
            float2_t color[4], fa[3];
#pragma unroll
            for (int i = 0; i < CH; i++) color[i] = nA[i] * xf + nB[i] * yf + nC[i] * zf;
            m_factors2<CH>(color, amnA, aofB, aofC, sa.nA, sa.nB, sa.nC, sa.invA, sa.invB, sa.invC, fa);
            sum += __builtin_elementwise_abs(fa[0]) * invA[0] + __builtin_elementwise_abs(0.5f - fa[1]) * invA[1] + __builtin_elementwise_abs(0.5f - fa[2]) * invA[2];
            sum += termB;
          }
      const float2_t avgF = sum * (1.f / 27);
      if (!q0 && !j0) r0 = avgF.x < 3.0f;
      if (!q1 && !j1) r1 = avgF.y < 3.0f;
    }

    // the two early exits of the predicate (src/limg.cpp:1168-1194) for one candidate: 1 = match, 2 = no match, 0 = the 27-colour loop decides
    template <int CH>
    __device__ __forceinline__ int m_early(const limg_hip_block_record &a, const MState &sa, const limg_hip_block_record &b)
    {
      MState sb;
      m_init<CH>(b, sb);
      const float w[4] = { 2, 4, 3, 3 };
      float avgDiffSq = 0, lenA[3] = { 3, 3, 3 }, lenB[3] = { 3, 3, 3 };
#pragma unroll
      for (int i = 0; i < CH; i++)
      {
        const float d = a.avg[i] - b.avg[i];
        avgDiffSq += d * d * w[i];
        lenA[0] += (sa.nA[i] * sa.nA[i]) * w[i]; lenB[0] += (sb.nA[i] * sb.nA[i]) * w[i];
        lenA[1] += (sa.nB[i] * sa.nB[i]) * w[i]; lenB[1] += (sb.nB[i] * sb.nB[i]) * w[i];
        lenA[2] += (sa.nC[i] * sa.nC[i]) * w[i]; lenB[2] += (sb.nC[i] * sb.nC[i]) * w[i];
      }
      const float sumA = lenA[0] + lenA[1] + lenA[2], sumB = lenB[0] + lenB[1] + lenB[2];
      const float ratio = (sumA + 1) / (sumB + 1);
      const float maxAvg = (float)(16 * 3 * CH), maxRange = (float)(200 * 3 * CH);
      if (avgDiffSq < maxAvg && sumA < maxRange && sumB < maxRange) return 1;
      if (ratio > 1.375f || ratio < (1.f / 1.375f)) return 2;
      return 0;
    }

    __device__ __forceinline__ void cell_to_block(const BlockedParams &p, uint32_t sx, uint32_t sy, int cell, bool &ok, size_t &idx)
    {
      const int dy = cell / kMatchSide - kMatchLo, dx = cell - (cell / kMatchSide) * kMatchSide - kMatchLo;
      const uint32_t cx = sx + (uint32_t)dx, cy = sy + (uint32_t)dy; // wraps for negative offsets => fails the range test
      ok = cell < kMatchCells && (dx | dy) != 0 && cx < p.blocksX && cy < p.blocksY;
      idx = ok ? (size_t)cy * p.blocksX + cx : 0;
    }

    // A cheap CERTAIN outcome for most of the cells the two early exits leave open.  The predicate's last test is avgF = (sum over 27 colours of termA + termB) / 27 < 3,
    // every term of the form |f0| / len0 + |0.5 - f1| * 2 / len1 + |0.5 - f2| * 2 / len2 with f = the factors of a colour c in a block's state (m_factors).
    //
    // Certain match.  By the triangle and Cauchy-Schwarz inequalities alone (no orthogonality assumed, so they hold for the rounded float operations up to a factor
    // (1 + 2^-23)^k, k < 200):   D := |c| + |dirA_min|;  |f0| <= D / |nA|;  |f1| <= (2 D + |ofB|) / |nB|;  |f2| <= (4 D + |ofB| + |ofC|) / |nC|   (a factor whose normal is
    // all zero is exactly 0), i.e. term <= alpha + beta * D with per-block constants alpha, beta.  For termA the state is the seed's and |c| <= S_b := |nA| + |nB| + |nC| of the
    // candidate (the 27 colours are nA x + nB y + nC z, x, y, z in {0, 0.5, 1}); for termB the state is the candidate's and c the seed's average.  So
    //   avgF <= alpha_a + beta_a (S_b + M_a) + alpha_b + beta_b (|avg_a| + M_b) =: U,   M = |dirA_min|,
    // and U * 1.01 < 2.5 proves avgF < 3 whatever the 27-colour loop would have rounded to.  On noisy content (large normals, tiny terms) this decides EVERY open cell --
    // the 27-colour evaluation, 5/6 of this kernel's time there, is not run at all; on smooth gradients it decides 3 %.
    //
    // Certain failure, from the same constants.  The factors are affine in the colour, so a term moves by at most beta * |c - c'| between two colours (same inequalities
    // on differences), and the first of the 27 colours is the zero vector whatever the candidate.  With T0 := the seed's term at colour 0 (once per seed, by the loop's own
    // operations): term(c) >= T0 - beta_a |c|, and the 27 colours' norms add up to at most 27 * 0.5 * S_b (x, y and z each average 0.5), so
    //   (sum of the 27 termA) / 27 >= T0 - 0.5 beta_a S_b - (rounding: < 2e-5 U_a);   termB >= 0, and a float sum of non-negative terms is monotone:
    //   T0 - 1.01 * 0.5 beta_a S_b - 0.01 U_a > 3.01  proves avgF > 3.
    // On smooth gradients (small normals: a seed's zero-colour term is ~8) this decides 53 % of the open pairs, on noisy content none.
    // Cost: k_blocked_bounds (one lane per block: { alpha, beta, M, S }), then one 16-byte load and a dozen operations per open cell.
    template <int CH>
    __global__ __launch_bounds__(256) void k_blocked_bounds(const BlockedParams p)
    {
      const uint32_t i = blockIdx.x * 256 + threadIdx.x;
      if (i >= p.blocksX * p.blocksY) return;
      const limg_hip_block_record r = p.pass1[i];
      MState s;
      m_init<CH>(r, s);
      const float w[4] = { 2, 4, 3, 3 };
      float len[3] = { 3, 3, 3 }, qA = 0, qB = 0, qC = 0, qM = 0, qOB = 0, qOC = 0;
#pragma unroll
      for (int k = 0; k < CH; k++)
      {
        len[0] += (s.nA[k] * s.nA[k]) * w[k]; len[1] += (s.nB[k] * s.nB[k]) * w[k]; len[2] += (s.nC[k] * s.nC[k]) * w[k];
        qA += s.nA[k] * s.nA[k]; qB += s.nB[k] * s.nB[k]; qC += s.nC[k] * s.nC[k];
        const float m = (float)r.dirA_min[k], ob = (float)r.dirB_offset[k], oc = (float)r.dirC_offset[k];
        qM += m * m; qOB += ob * ob; qOC += oc * oc;
      }
      const float NA = sqrtf(qA), NB = sqrtf(qB), NC = sqrtf(qC), OB = sqrtf(qOB), OC = sqrtf(qOC);
      float alpha = 0.5f * 2.0f / len[1] + 0.5f * 2.0f / len[2], beta = 0;
      if (qA > 0) beta += 1.0f / (NA * len[0]);
      if (qB > 0) { alpha += (OB / NB) * 2.0f / len[1]; beta += 4.0f / (NB * len[1]); }
      if (qC > 0) { alpha += ((OB + OC) / NC) * 2.0f / len[2]; beta += 8.0f / (NC * len[2]); }
      reinterpret_cast<float4 *>(p.matchBound)[i] = make_float4(alpha, beta, sqrtf(qM), (NA + NB + NC) * 1.0001f);
    }

    // One wave per seed.  Step 1: every cell of the window through the early exits (cheap), lane = cell; the undecided cells are compacted into
    // a list in LDS.  Step 2: the expensive loop over the list only, two candidates per lane.
    template <int CH>
    // Three workgroups per CU (168 VGPRs, three dwords of scratch) instead of the two the compiler takes by itself (218): 14.7-15.0 -> 13.8-14.0 ms per 8192^2 image and
    // 3.23 -> 3.49 Gpixel/s over four contexts (tools/r04/run27.sh, same box); four (128 VGPRs, 344 B of scratch, or 192 B with the x loop rolled) is slower: 17.7 ms.
#ifndef LIMG_MATCH_WGS
#define LIMG_MATCH_WGS 3
#endif
    __global__ __launch_bounds__(256, LIMG_MATCH_WGS) void k_blocked_match(const BlockedParams p)
    {
      __shared__ unsigned short sList[4][kMatchWords * 64];
      __shared__ unsigned long long sWords[4][kMatchWords];
      const int wave = threadIdx.x >> 6;
      const uint32_t seed = p.seedBase + blockIdx.x * 4 + wave;
      const int lane = lane_id();
      if (seed >= p.seedBase + p.seedCount) return;
      const uint32_t sy = seed / p.blocksX, sx = seed - sy * p.blocksX;
      const limg_hip_block_record a = p.pass1[seed];
      MState sa;
      m_init<CH>(a, sa);
      float boundA = 0, boundB = 0, boundM = 0, boundAvg = 0, boundT0 = 0; // the seed's share of the certain-match / certain-failure bounds (k_blocked_bounds)
      if (p.matchBound)
      {
        const float4 ba = reinterpret_cast<const float4 *>(p.matchBound)[seed];
        float q = 0;
#pragma unroll
        for (int k = 0; k < CH; k++) q += a.avg[k] * a.avg[k];
        boundA = ba.x; boundB = ba.y; boundM = ba.z; boundAvg = sqrtf(q);
        // the seed's term at the zero colour (x = y = z = 0 of the 27-colour loop), with the loop's operations
        const float w[4] = { 2, 4, 3, 3 }, zero[4] = { 0, 0, 0, 0 };
        float len[3] = { 3, 3, 3 }, f0[3];
#pragma unroll
        for (int k = 0; k < CH; k++) { len[0] += (sa.nA[k] * sa.nA[k]) * w[k]; len[1] += (sa.nB[k] * sa.nB[k]) * w[k]; len[2] += (sa.nC[k] * sa.nC[k]) * w[k]; }
        m_factors<CH>(zero, a, sa, f0);
        boundT0 = fabsf(f0[0]) * (1.0f / len[0]) + fabsf(0.5f - f0[1]) * ((1.0f / len[1]) * 2.f) + fabsf(0.5f - f0[2]) * ((1.0f / len[2]) * 2.f);
      }
      uint32_t count = 0;
      for (int c = 0; c < kMatchWords; c++)
      {
        const int cell = c * 64 + lane;
        bool ok;
        size_t idx;
        cell_to_block(p, sx, sy, cell, ok, idx);
        int e = 2;
        if (ok) e = m_early<CH>(a, sa, p.pass1[idx]);
        if (e == 0 && p.matchBound)
        {
          const float4 bb = reinterpret_cast<const float4 *>(p.matchBound)[idx];
          const float UA = boundA + boundB * (bb.w + boundM), U = UA + (bb.x + bb.y * (boundAvg + bb.z));
          if (U * 1.01f < 2.5f) e = 1; // (a NaN or an infinity fails both comparisons: the 27-colour loop decides)
          else if (boundT0 - 1.01f * (0.5f * boundB * bb.w) - 0.01f * UA > 3.01f) e = 2;
        }
        const unsigned long long yes = __builtin_amdgcn_ballot_w64(e == 1), open = __builtin_amdgcn_ballot_w64(e == 0);
        if (lane == 0) sWords[wave][c] = yes;
        if (e == 0) sList[wave][count + __builtin_amdgcn_mbcnt_hi((uint32_t)(open >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)open, 0u))] = (unsigned short)cell;
        count += (uint32_t)__builtin_popcountll(open);
      }
      wave_lds_fence();
#ifdef LIMG_MATCH_SKIP2 // timing experiment only (wrong results; tools/r04/run9.sh): how much of the kernel is the expensive evaluation?
      count = 0;
#endif
      for (uint32_t k = 0; k < count; k += 128)
      {
        const uint32_t i0 = k + lane, i1 = k + 64 + lane;
        const bool v0 = i0 < count, v1 = i1 < count;
        const int cell0 = sList[wave][v0 ? i0 : 0], cell1 = sList[wave][v1 ? i1 : 0];
        bool ok0, ok1, r0 = false, r1 = false;
        size_t idx0, idx1;
        cell_to_block(p, sx, sy, cell0, ok0, idx0);
        cell_to_block(p, sx, sy, cell1, ok1, idx1);
        m_matches_pair<CH>(a, p.pass1[idx0], p.pass1[idx1], r0, r1);
        if (v0 && r0) atomicOr(&sWords[wave][cell0 >> 6], 1ull << (cell0 & 63));
        if (v1 && r1) atomicOr(&sWords[wave][cell1 >> 6], 1ull << (cell1 & 63));
      }
      wave_lds_fence();
      if (lane < kMatchWords) p.matchBits[(size_t)seed * kMatchWords + lane] = sWords[wave][lane];
      if (lane == 0)
      { // necessary conditions the host scan tests before it bothers to grow a rectangle from this seed: the growth order is right, down, right, down
        // (src/limg.cpp:1307-1335), so a rectangle of >= 3 x 3 blocks needs all eight neighbours of the seed's 3x3 window, and any rectangle at all needs the
        // right or the lower neighbour
        auto bit = [&](int dx, int dy) -> uint32_t { const int cell = (dy + kMatchLo) * kMatchSide + dx + kMatchLo; return (uint32_t)(sWords[wave][cell >> 6] >> (cell & 63)) & 1u; };
        const uint32_t all8 = bit(1, 0) & bit(2, 0) & bit(0, 1) & bit(1, 1) & bit(2, 1) & bit(0, 2) & bit(1, 2) & bit(2, 2);
        p.matchFlags[seed] = (uint8_t)(all8 | ((bit(1, 0) | bit(0, 1)) << 1));
      }
    }

    // ---- regions ----------------------------------------------------------------------------------------------------------------------
    struct Geo { uint32_t px0, py0, xpx, ypx, n; };

    __device__ __forceinline__ Geo region_geo(const BlockedParams &p, const RegionDesc &R)
    { // src/limg.cpp:1722-1740: the last block column / row may be partial
      Geo g;
      g.px0 = R.ox * kBlock; g.py0 = R.oy * kBlock;
      g.xpx = R.rx * kBlock; g.ypx = R.ry * kBlock;
      if (R.ox + R.rx == p.blocksX && (p.sizeX % kBlock)) g.xpx = g.xpx - kBlock + p.sizeX % kBlock;
      if (R.oy + R.ry == p.blocksY && (p.sizeY % kBlock)) g.ypx = g.ypx - kBlock + p.sizeY % kBlock;
      g.n = g.xpx * g.ypx;
      return g;
    }

    __device__ __forceinline__ float bcast(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

    // 64-bit sum over the wave of per-lane 64-bit values whose wave total stays far below 2^63
    __device__ __forceinline__ uint64_t wave_sum64(uint64_t v)
    {
      const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
      return (uint64_t)wave_sum(lo & 0xFFFFu) + ((uint64_t)wave_sum(lo >> 16) << 16) + ((uint64_t)wave_sum(hi) << 32);
    }

    template <int CH>
    __global__ __launch_bounds__(64) void k_blocked_fit_search(const BlockedParams p)
    {
      __shared__ __attribute__((aligned(16))) float s_park[4][64 + 4]; // one chunk's unit vectors, slot-planar; plane stride 68 floats: the four walkers' 16-byte reads hit different banks
      const uint32_t r = p.order ? p.order[blockIdx.x] : blockIdx.x;
      const int lane = lane_id();
      const RegionDesc R = p.regions[r];
      const Geo g = region_geo(p, R);
      const uint32_t n = g.n, cap = p.scratchCap;
      uint32_t *spx = p.scratchPx + R.scratch;
      uint8_t *sf = p.scratchFac + R.scratch;
      const unsigned short *tab = d_rsqrt_x86_tab;

      // ---- gather (row-major inside the region) + channel sums (32-bit lanes like the reference's, src/limg.cpp:466-497) ----------------
      uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
      for (uint32_t i = lane; i < n; i += 64)
      {
        const uint32_t row = i / g.xpx, col = i - row * g.xpx;
        const uint32_t px = p.in[(size_t)(g.py0 + row) * p.sizeX + g.px0 + col];
        spx[i] = px;
        s0 += px & 0xFF; s1 += (px >> 8) & 0xFF; s2 += (px >> 16) & 0xFF; s3 += px >> 24;
      }
      scratch_fence();

      limg_hip_block_record rec;
      memset(&rec, 0, sizeof(rec));
      if (R.keep)
      {
        rec = p.pass1[(size_t)R.oy * p.blocksX + R.ox];
      }
      else
      {
        s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
        const float inv_count = 1.0f / (float)n;
        V4 avg;
        avg.a = float2_t{ (float)(int)s0, (float)(int)s2 } * inv_count;
        avg.b = float2_t{ (float)(int)s1, CH == 4 ? (float)(int)s3 : 0.0f } * inv_count;
        const V4 zero4 = { float2_t{ 0.0f, 0.0f }, float2_t{ 0.0f, 0.0f } };
        V4 dirA = zero4, dirB = zero4, dirC = zero4;
        float mm[6] = { 0, 0, 0, 0, 0, 0 };
        bool zeroA = true, zeroB = true, zeroC = true;

        // Pixel-order sums of the parked unit vectors (the three direction sums are serial in pixel order upstream).  Round 5: a chunk's 64 vectors are parked in LDS
        // (slot-planar: 4 planes of 64 floats) and lane s (< 4) adds plane s to its running sum right there, before the next chunk overwrites them -- the sum is carried
        // from chunk to chunk in the walker lanes' registers, so the order of the additions is the reference's whatever the rectangle's size, and nothing is parked in
        // global memory.  (Rounds 1-4 parked ALL N vectors of a pass in global scratch and walked them afterwards: 96 of the kernel's 117 bytes of HBM traffic per
        // pixel, and every walk a chain of global-memory round trips -- ~160 cycles per four terms, later ~40 with sixteen terms requested ahead; a batch's kernel
        // ends with its largest rectangle, whose three walks were most of its life.  Counters: profiles/archive/r05_final_blocked_summary.txt.)
        float walk = 0.0f; // lanes 0..3: the running sum of their slot plane
        auto park_and_walk = [&](const V4 &u, uint32_t count /* pixels of this chunk, 1..64: wave-uniform */) {
          s_park[0][lane] = u.a.x; s_park[1][lane] = u.a.y; s_park[2][lane] = u.b.x; s_park[3][lane] = u.b.y;
          wave_lds_fence();
          if (lane < 4)
          {
            const float *src = s_park[lane];
            if (count == 64u)
            {
#pragma unroll
              for (int h = 0; h < 4; h++) // sixteen terms at a time: their four LDS reads are issued together, the adds are one dependent chain
              {
                float4 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = *reinterpret_cast<const float4 *>(src + 16 * h + 4 * q);
#pragma unroll
                for (int q = 0; q < 4; q++) { walk = walk + v[q].x; walk = walk + v[q].y; walk = walk + v[q].z; walk = walk + v[q].w; }
              }
            }
            else
              for (uint32_t i = 0; i < count; i++) walk = walk + src[i];
          }
          wave_lds_fence(); // (the next chunk's vectors overwrite the planes)
        };
        // the pass's sum / N in slot order x0 x2 x1 x3 (see V4), broadcast to every lane; the walkers start the next pass from zero
        auto serial_sum = [&]() -> V4 {
          const float s = walk * inv_count;
          walk = 0.0f;
          V4 d;
          d.a = float2_t{ bcast(s, 0), bcast(s, 1) };
          d.b = float2_t{ bcast(s, 2), bcast(s, 3) };
          return d;
        };
        auto is_zero = [](const V4 &d) { return d.a.x == 0.0f && d.a.y == 0.0f && d.b.x == 0.0f && d.b.y == 0.0f; };

        // pass 1 (src/limg_factorization.h:602-628 / :402-431)
        for (uint32_t base = 0; base < n; base += 64)
        {
          const uint32_t i = base + lane;
          const bool active = i < n;
          const V4 pf = px_to_v4(active ? spx[i] : 0u);
          V4 d = pf - avg;
          mask_alpha<CH>(d);
          const V4 u = unit4<CH>(tab, d, active);
          park_and_walk(u, min(64u, n - base));
        }
        dirA = serial_sum();
        zeroA = is_zero(dirA);
        if (!zeroA)
        {
          const float invA = 1.0f / dp4<CH>(dirA, dirA);
          // pass 2 (:652-688 / :451-491)
          float mn = 0.0f, mx = 0.0f;
          for (uint32_t base = 0; base < n; base += 64)
          {
            const uint32_t i = base + lane;
            const bool active = i < n;
            const V4 pf = px_to_v4(active ? spx[i] : 0u);
            const float fA = dp4<CH>(pf - avg, dirA) * invA;
            if (active) { mn = vmin(mn, fA); mx = vmax(mx, fA); }
            V4 e = pf - (avg + dirA * fA);
            mask_alpha<CH>(e);
            const V4 u = unit4<CH>(tab, e, active);
            park_and_walk(u, min(64u, n - base));
          }
          wave_min_max(mn, mx);
          mm[0] = mn; mm[1] = mx;
          dirB = serial_sum();
          zeroB = is_zero(dirB);
          if (!zeroB)
          {
            const float invB = 1.0f / dp4<CH>(dirB, dirB);
            float mnB = FLT_MAX, mxB = -FLT_MAX, mnC = FLT_MAX, mxC = -FLT_MAX;
            if (CH == 4)
            {
              // pass 3 (:701-738)
              V4 est0 = zero4;
              for (uint32_t base = 0; base < n; base += 64)
              {
                const uint32_t i = base + lane;
                const bool active = i < n;
                const V4 pf = px_to_v4(active ? spx[i] : 0u);
                const float fA = dp4<CH>(pf - avg, dirA) * invA;
                const V4 est = avg + dirA * fA;
                const float fB = dp4<CH>(pf - est, dirB) * invB;
                if (active) { mnB = vmin(mnB, fB); mxB = vmax(mxB, fB); }
                const V4 est2 = est + dirB * fB;
                const V4 u = unit4<CH>(tab, pf - est2, active);
                park_and_walk(u, min(64u, n - base));
                if (base == 0) est0 = est2;
              }
              wave_min_max(mnB, mxB);
              est0.a = float2_t{ bcast(est0.a.x, 0), bcast(est0.a.y, 0) };
              est0.b = float2_t{ bcast(est0.b.x, 0), bcast(est0.b.y, 0) };
              dirC = serial_sum();
              zeroC = is_zero(dirC);
              if (!zeroC)
              {
                // pass 4 (:748-758): upstream never advances its estimate pointer -- every pixel is measured against pixel 0's A+B estimate
                const float invC = 1.0f / dp4<CH>(dirC, dirC);
                for (uint32_t base = 0; base < n; base += 64)
                {
                  const uint32_t i = base + lane;
                  const bool active = i < n;
                  const V4 pf = px_to_v4(active ? spx[i] : 0u);
                  const float fC = dp4<CH>(pf - est0, dirC) * invC;
                  if (active) { mnC = vmin(mnC, fC); mxC = vmax(mxC, fC); }
                }
                wave_min_max(mnC, mxC);
              }
            }
            else
            {
              // dirC = dirA x dirB (:498-507); slots: a = (x0, x2), b = (x1, x3); pass 3 (:517-541): B and C extrema together
              dirC.a.x = dirA.b.x * dirB.a.y - dirA.a.y * dirB.b.x;
              dirC.b.x = dirA.a.y * dirB.a.x - dirA.a.x * dirB.a.y;
              dirC.a.y = dirA.a.x * dirB.b.x - dirA.b.x * dirB.a.x;
              dirC.b.y = 0.0f;
              zeroC = dirC.a.x == 0.0f && dirC.b.x == 0.0f && dirC.a.y == 0.0f;
              const float invC = zeroC ? 0.0f : 1.0f / dp4<CH>(dirC, dirC);
              for (uint32_t base = 0; base < n; base += 64)
              {
                const uint32_t i = base + lane;
                const bool active = i < n;
                const V4 pf = px_to_v4(active ? spx[i] : 0u);
                const float fA = dp4<CH>(pf - avg, dirA) * invA;
                const V4 est = avg + dirA * fA;
                const float fB = dp4<CH>(pf - est, dirB) * invB;
                if (active) { mnB = vmin(mnB, fB); mxB = vmax(mxB, fB); }
                if (!zeroC)
                {
                  const V4 e = pf - (est + dirB * fB);
                  const float fC = dp4<CH>(e, dirC) * invC;
                  if (active) { mnC = vmin(mnC, fC); mxC = vmax(mxC, fC); }
                }
              }
              wave_min_max(mnB, mxB);
              if (!zeroC) wave_min_max(mnC, mxC);
            }
            mm[2] = mnB; mm[3] = mxB;
            if (!zeroC) { mm[4] = mnC; mm[5] = mxC; }
          }
        }
        // records (:764-790 / :545-575).  A direction that is exactly zero makes upstream divide by zero: every factor of that pass and of the
        // later ones is NaN and the x86 conversion yields int16 0; same rule as the 8x8 kernel.
        const float av[4] = { avg.a.x, avg.b.x, avg.a.y, avg.b.y };
        const float dA[4] = { dirA.a.x, dirA.b.x, dirA.a.y, dirA.b.y }, dB[4] = { dirB.a.x, dirB.b.x, dirB.a.y, dirB.b.y }, dC[4] = { dirC.a.x, dirC.b.x, dirC.a.y, dirC.b.y };
        const bool deadA = zeroA, deadB = zeroA || zeroB, deadC = deadB || zeroC;
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          rec.avg[c] = (CH == 3 && c == 3) ? 0.0f : av[c];
          if (CH == 3 && c == 3) continue;
          rec.dirA_min[c] = (int16_t)cvt_rne(av[c] + (deadA ? 0.0f : mm[0] * dA[c]));
          rec.dirA_max[c] = (int16_t)cvt_rne(av[c] + (deadA ? 0.0f : mm[1] * dA[c]));
          rec.dirB_offset[c] = (int16_t)cvt_rne(deadB ? 0.0f : mm[2] * dB[c]);
          rec.dirB_mag[c] = (int16_t)cvt_rne(deadB ? 0.0f : mm[3] * dB[c]);
          rec.dirC_offset[c] = (int16_t)cvt_rne(deadC ? 0.0f : mm[4] * dC[c]);
          rec.dirC_mag[c] = (int16_t)cvt_rne(deadC ? 0.0f : mm[5] * dC[c]);
        }
      }

      // ---- a7 (src/limg_internal.h:426-452): float normals, 1 / |n|^2 in limg_dot's serial order ---------------------------------------
      const int16_t *lo3[3] = { rec.dirA_min, rec.dirB_offset, rec.dirC_offset }, *hi3[3] = { rec.dirA_max, rec.dirB_mag, rec.dirC_mag };
      V4 nrm[3], off[3];
      float invN[3];
      RecU ru;
#pragma unroll
      for (int f = 0; f < 3; f++)
      {
        float nv[4], ov[4];
#pragma unroll
        for (int c = 0; c < 4; c++) { nv[c] = (float)((int)hi3[f][c] - (int)lo3[f][c]); ov[c] = (float)lo3[f][c]; }
        float s = ((0.0f + nv[0] * nv[0]) + nv[1] * nv[1]) + nv[2] * nv[2];
        if (CH == 4) s = s + nv[3] * nv[3];
        const bool nz = nv[0] != 0.0f || nv[1] != 0.0f || nv[2] != 0.0f || (CH == 4 && nv[3] != 0.0f);
        invN[f] = nz ? 1.0f / s : 0.0f;
        nrm[f].a = float2_t{ nv[0], nv[2] }; nrm[f].b = float2_t{ nv[1], nv[3] };
        off[f].a = float2_t{ ov[0], ov[2] }; off[f].b = float2_t{ ov[1], ov[3] };
      }
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
        ru.nA[c] = (int)rec.dirA_max[c] - (int)rec.dirA_min[c]; ru.nB[c] = (int)rec.dirB_mag[c] - (int)rec.dirB_offset[c]; ru.nC[c] = (int)rec.dirC_mag[c] - (int)rec.dirC_offset[c];
        ru.mA[c] = (int)(((uint32_t)(int)rec.dirA_min[c] << 8) + 128u); ru.mB[c] = (int)(((uint32_t)(int)rec.dirB_offset[c] << 8) + 128u);
        ru.mC[c] = (int)(((uint32_t)(int)rec.dirC_offset[c] << 8) + 128u);
      }

      // ---- a8 (src/limg_factorization.h:98-197): per-pixel factor bytes -> scratch ------------------------------------------------------
      for (uint32_t base = 0; base < n; base += 64)
      {
        const uint32_t i = base + lane;
        if (i >= n) continue;
        const V4 pv = px_to_v4(spx[i]);
        const float fa = dp4<CH>(pv - off[0], nrm[0]) * invN[0];
        V4 est = off[0] + nrm[0] * fa;
        const float fb = dp4<CH>((pv - est) - off[1], nrm[1]) * invN[1];
        est = est + nrm[1] * fb;
        const float fc = dp4<CH>((pv - est) - off[2], nrm[2]) * invN[2];
        sf[i] = (uint8_t)cvt_u8_rne_sat(255.0f * fa);
        sf[(size_t)cap + i] = (uint8_t)cvt_u8_rne_sat(255.0f * fb);
        sf[2 * (size_t)cap + i] = (uint8_t)cvt_u8_rne_sat(255.0f * fc);
      }
      scratch_fence();

      // ---- a9-a12: shift search over the region's N pixels ---------------------------------------------------------------------------------
      uint32_t shift[3] = { 0, 0, 0 };
      if (p.forced[0] >= 0)
      {
        shift[0] = (uint32_t)p.forced[0]; shift[1] = (uint32_t)p.forced[1]; shift[2] = (uint32_t)p.forced[2];
      }
      else if (p.crushBits)
      {
        const uint64_t maxBlockN = p.maxBlock * (uint64_t)n;
        auto T = [&](uint32_t a, uint32_t b, uint32_t c, uint64_t &be) -> bool {
          uint64_t acc = 0;
          for (uint32_t base = 0; base < n; base += 64)
          {
            const uint32_t i = base + lane;
            uint32_t err = 0;
            if (i < n) err = trial_error(spx[i], sf[i], sf[(size_t)cap + i], sf[2 * (size_t)cap + i], ru, a, b, c);
            if (__builtin_amdgcn_ballot_w64(err > p.maxPixel32) != 0ull) return false; // src/limg_bit_crush_simd.h:772-781: first offending pixel ends the trial
            acc += err;
          }
          be = wave_sum64(acc);
          return be * 16ull < maxBlockN;
        };
        if (p.fast) search_fast<uint64_t>(T, shift);
        else search_accurate<uint64_t>(T, shift);
      }
      const uint32_t calls = (shift[0] && shift[0] != 8 ? 1u : 0u) + (shift[1] && shift[1] != 8 ? 1u : 0u) + (shift[2] && shift[2] != 8 ? 1u : 0u);
      if (lane == 0)
      {
        RegionOut o;
        o.rec = rec;
        o.shiftWord = shift[0] | (shift[1] << 8) | (shift[2] << 16) | (calls << 24);
        o.pad[0] = o.pad[1] = o.pad[2] = 0;
        p.out[r] = o;
      }
    }

    // Which rectangle each workgroup of a batch's launches takes: a launch of one wave per rectangle lives as long as its longest wave and workgroups are dispatched
    // in index order, so the large rectangles -- more than four blocks: their three pixel-order walks are most of such a launch's life -- go FIRST, largest first
    // (counting sort over the size in blocks), and everything else follows in creation order (raster locality).  Round 5 measured the ordering (k_blocked_fit_search
    // -11 % on photo-noise) but built it on the worker thread, which is the thread that walks the dither chain: the image took longer, and it was dropped.  Here it is
    // one small workgroup on the batch's own stream in front of the batch's kernel; the host does nothing.
    __global__ __launch_bounds__(1024) void k_blocked_order(const BlockedParams p)
    {
      __shared__ uint32_t sBins[1024], sWaveSmall[16];
      __shared__ uint32_t sLarge, sCarry;
      const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
      const uint32_t n = p.nRegions;
      sBins[tid] = 0;
      if (tid == 0) sCarry = 0;
      __syncthreads();
      for (uint32_t i = tid; i < n; i += 1024)
      {
        const uint32_t nb = p.regions[i].rx * p.regions[i].ry;
        if (nb > 4u) atomicAdd(&sBins[nb < 1023u ? nb : 1023u], 1u);
      }
      __syncthreads();
      // first slot of every size class, largest class first: exclusive suffix sum over the bins (Hillis-Steele in LDS, the thread of bin b reads bin b + off)
      uint32_t own = sBins[tid], suf = own;
      for (int off = 1; off < 1024; off <<= 1)
      {
        __syncthreads();
        sBins[tid] = suf;
        __syncthreads();
        if (tid + off < 1024) suf += sBins[tid + off];
      }
      __syncthreads();
      sBins[tid] = suf - own; // rectangles in larger classes
      if (tid == 0) sLarge = suf; // all large rectangles
      __syncthreads();
      const uint32_t nLarge = sLarge;
      for (uint32_t base = 0; base < n; base += 1024)
      {
        const uint32_t i = base + tid;
        uint32_t nb = 0;
        if (i < n) nb = p.regions[i].rx * p.regions[i].ry;
        const bool small = i < n && nb <= 4u;
        if (i < n && !small) p.order[atomicAdd(&sBins[nb < 1023u ? nb : 1023u], 1u)] = i; // (within a class the order does not matter)
        const uint64_t m = __builtin_amdgcn_ballot_w64(small);
        if (lane == 0) sWaveSmall[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t before = sCarry;
        for (int w = 0; w < wave; w++) before += sWaveSmall[w];
        if (small) p.order[nLarge + before + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0)
        {
          uint32_t t = sCarry;
          for (int w = 0; w < 16; w++) t += sWaveSmall[w];
          sCarry = t;
        }
        __syncthreads();
      }
    }

    template <int CH>
    __global__ __launch_bounds__(64) void k_blocked_store(const BlockedParams p)
    {
      const uint32_t r = p.order ? p.order[blockIdx.x] : blockIdx.x;
      const int lane = lane_id();
      const RegionDesc R = p.regions[r];
      const Geo g = region_geo(p, R);
      const uint32_t n = g.n, cap = p.scratchCap;
      const uint8_t *sf = p.scratchFac + R.scratch;
      const RegionOut O = p.out[r];
      const limg_hip_block_record &rec = O.rec;
      const uint32_t shift[3] = { O.shiftWord & 0xFF, (O.shiftWord >> 8) & 0xFF, (O.shiftWord >> 16) & 0xFF };

      // block-uniform plane values (src/limg.cpp:1604-1627) and the decoder's constants (src/limg_decode.h:139-196 / :40-101)
      const int16_t *vec[6] = { rec.dirA_min, rec.dirA_max, rec.dirB_offset, rec.dirB_mag, rec.dirC_offset, rec.dirC_mag };
      uint32_t col[6];
#pragma unroll
      for (int k = 0; k < 6; k++)
      {
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < CH; c++)
        {
          int q = (int)vec[k][c] + (k >= 2 ? 0x80 : 0);
          q = q < 0 ? 0 : (q > 255 ? 255 : q);
          v |= (uint32_t)q << (8 * c);
        }
        if (CH == 3) v |= 0xFF000000u;
        col[k] = v;
      }
      const uint32_t pa = shift[0] == 8 ? 0xFFu : shift[0] * 0x22u, pb = shift[1] == 8 ? 0xFFu : shift[1] * 0x22u, pc = shift[2] == 8 ? 0xFFu : shift[2] * 0x22u;
      const uint32_t shiftVal = 0xFF000000u | (pa << 16) | (pb << 8) | pc;
      // src/limg.cpp:1629-1636: 110 / 136 header bits + (8 - shift) bits per factor and pixel, per-pixel average rounded to nearest
      const uint64_t bits = (uint64_t)(CH * 18 + CH * 8 + 32) + (uint64_t)n * ((8 - shift[0]) + (8 - shift[1]) + (8 - shift[2]));
      const uint8_t bpp = (uint8_t)((bits + n / 2) / n);
      const uint32_t blockIndex = 0xFF000000u | (p.regionBase + r + 1u);
      int nn[3][4], mc[3][4];
#pragma unroll
      for (int f = 0; f < 3; f++)
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          int nv = (int)vec[2 * f + 1][c] - (int)vec[2 * f][c], m = (int)vec[2 * f][c];
          if (c < 3) { if (shift[f] > 7) { nv = 0; if (f > 0) m = 0; } }
          else if (CH == 3) { nv = 0; m = 0xFFFF; }
          nn[f][c] = nv;
          mc[f][c] = (int)(((uint32_t)m << 8) + 128u);
        }
      const int mulA = (int)shift_mul(shift[0]), mulB = (int)shift_mul(shift[1]), mulC = (int)shift_mul(shift[2]);
      // noise: one byte per pixel per dither call, the region's calls back to back in A, B, C order
      const uint8_t *nz[3];
      {
        const uint8_t *q = p.noise + p.noiseBase[r];
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
          nz[k] = q;
          if (shift[k] != 0 && shift[k] != 8) q += n;
        }
      }

      if (p.vecStore)
      { // images of whole blocks: a lane owns FOUR consecutive pixels of a rectangle row (widths are multiples of 8): one 16-byte store per 32-bit plane and one dword
        // per byte plane where the pixel-per-lane loop below issues four stores of 4 / 1 bytes -- the 13 planes of a 64-pixel rectangle leave in 13 store
        // instructions of 8 x 32-byte row pieces instead of 13 x 8 of them (the kernel is bound by its store instructions: 2.9 GB at 1.5 TB/s)
        for (uint32_t base = 0; base < n; base += 256)
        {
          const uint32_t i = base + (uint32_t)lane * 4u;
          if (i >= n) continue;
          const uint32_t row = i / g.xpx, colx = i - row * g.xpx;
          const size_t o = (size_t)(g.py0 + row) * p.sizeX + g.px0 + colx;
          uint32_t fq[3]; // the four pixels' crushed values of a factor, a byte each
#pragma unroll
          for (int k = 0; k < 3; k++)
          {
            uint32_t f4 = *reinterpret_cast<const uint32_t *>(sf + (size_t)k * cap + i);
            const uint32_t s = shift[k];
            if (s != 0 && s != 8)
            {
              const uint32_t z4 = *reinterpret_cast<const uint32_t *>(nz[k] + i);
              uint32_t out = 0;
#pragma unroll
              for (int q = 0; q < 4; q++)
              {
                int t = (int)((f4 >> (8 * q)) & 0xFFu) + ((int)((z4 >> (8 * q)) & ((1u << s) - 1u)) - (int)(1u << (s - 1)));
                t = t < 0 ? 0 : (t > 255 ? 255 : t);
                out |= ((uint32_t)t >> s) << (8 * q);
              }
              f4 = out;
            }
            fq[k] = f4;
          }
          // byte planes: (v << shift) per byte; shift 8 => 0 (the uint8 store upstream)
          uint32_t st[3];
#pragma unroll
          for (int k = 0; k < 3; k++) st[k] = shift[k] > 7 ? 0u : ((fq[k] << shift[k]) & ((0xFFu << shift[k]) & 0xFFu) * 0x01010101u);
          *reinterpret_cast<uint32_t *>(p.info.pFactorsA + o) = st[0];
          *reinterpret_cast<uint32_t *>(p.info.pFactorsB + o) = st[1];
          *reinterpret_cast<uint32_t *>(p.info.pFactorsC + o) = st[2];
          *reinterpret_cast<uint32_t *>(p.info.pBitsPerPixel + o) = (uint32_t)bpp * 0x01010101u;
          *reinterpret_cast<uint4 *>(p.info.pShiftABCX + o) = make_uint4(shiftVal, shiftVal, shiftVal, shiftVal);
          *reinterpret_cast<uint4 *>(p.info.pColAMin + o) = make_uint4(col[0], col[0], col[0], col[0]);
          *reinterpret_cast<uint4 *>(p.info.pColAMax + o) = make_uint4(col[1], col[1], col[1], col[1]);
          *reinterpret_cast<uint4 *>(p.info.pColBMin + o) = make_uint4(col[2], col[2], col[2], col[2]);
          *reinterpret_cast<uint4 *>(p.info.pColBMax + o) = make_uint4(col[3], col[3], col[3], col[3]);
          *reinterpret_cast<uint4 *>(p.info.pColCMin + o) = make_uint4(col[4], col[4], col[4], col[4]);
          *reinterpret_cast<uint4 *>(p.info.pColCMax + o) = make_uint4(col[5], col[5], col[5], col[5]);
          *reinterpret_cast<uint4 *>(p.info.pBlockIndex + o) = make_uint4(blockIndex, blockIndex, blockIndex, blockIndex);
          uint32_t dec[4];
#pragma unroll
          for (int q = 0; q < 4; q++)
          {
            const int dA = (int)((fq[0] >> (8 * q)) & 0xFFu) * mulA, dB = (int)((fq[1] >> (8 * q)) & 0xFFu) * mulB, dC = (int)((fq[2] >> (8 * q)) & 0xFFu) * mulC;
            uint32_t decoded = 0;
#pragma unroll
            for (int c = 0; c < 4; c++)
            {
              int est = (mad_i24(dA, nn[0][c], mc[0][c]) >> 8) + (mad_i24(dB, nn[1][c], mc[1][c]) >> 8) + (mad_i24(dC, nn[2][c], mc[2][c]) >> 8);
              est = est < 0 ? 0 : (est > 255 ? 255 : est);
              decoded |= (uint32_t)est << (8 * c);
            }
            dec[q] = decoded;
          }
          *reinterpret_cast<uint4 *>(p.info.pDecoded + o) = make_uint4(dec[0], dec[1], dec[2], dec[3]);
        }
        return;
      }
      for (uint32_t base = 0; base < n; base += 64)
      {
        const uint32_t i = base + lane;
        if (i >= n) continue;
        const uint32_t row = i / g.xpx, colx = i - row * g.xpx;
        const size_t o = (size_t)(g.py0 + row) * p.sizeX + g.px0 + colx;
        uint32_t v[3];
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
          uint32_t f = sf[(size_t)k * cap + i];
          const uint32_t s = shift[k];
          if (s != 0 && s != 8)
          { // src/limg.cpp:824-879: (lane16 & ditherSize) - ditherOffset, add, clamp, shift
            int t = (int)f + ((int)(nz[k][i] & ((1u << s) - 1u)) - (int)(1u << (s - 1)));
            t = t < 0 ? 0 : (t > 255 ? 255 : t);
            f = (uint32_t)t >> s;
          }
          v[k] = f;
        }
        // (plain stores: a lane writes ONE pixel of each plane, a wave a few short row pieces of the rectangle; stored non-temporally -- what pays for the 8x8 path's
        //  whole-line stores -- these partial lines bypass the L2's write combining: measured 4.3-4.7 -> 5.5 ms per image for the expansion + store kernels)
        p.info.pFactorsA[o] = (uint8_t)(v[0] << shift[0]); // shift 8 => 0, like the uint8 store upstream
        p.info.pFactorsB[o] = (uint8_t)(v[1] << shift[1]);
        p.info.pFactorsC[o] = (uint8_t)(v[2] << shift[2]);
        p.info.pBitsPerPixel[o] = bpp;
        p.info.pShiftABCX[o] = shiftVal;
        p.info.pColAMin[o] = col[0]; p.info.pColAMax[o] = col[1]; p.info.pColBMin[o] = col[2];
        p.info.pColBMax[o] = col[3]; p.info.pColCMin[o] = col[4]; p.info.pColCMax[o] = col[5];
        p.info.pBlockIndex[o] = blockIndex;
        // a16
        const int dA = (int)v[0] * mulA, dB = (int)v[1] * mulB, dC = (int)v[2] * mulC;
        uint32_t decoded = 0;
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
          int est = (mad_i24(dA, nn[0][c], mc[0][c]) >> 8) + (mad_i24(dB, nn[1][c], mc[1][c]) >> 8) + (mad_i24(dC, nn[2][c], mc[2][c]) >> 8);
          est = est < 0 ? 0 : (est > 255 ? 255 : est);
          decoded |= (uint32_t)est << (8 * c);
        }
        p.info.pDecoded[o] = decoded;
      }
    }
  }

  void launch_blocked_bounds(const BlockedParams &p, hipStream_t s)
  {
    const uint32_t blocks = p.blocksX * p.blocksY;
    if (blocks == 0 || !p.matchBound) return;
    if (p.channels == 4) hipLaunchKernelGGL(k_blocked_bounds<4>, dim3((blocks + 255) / 256), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k_blocked_bounds<3>, dim3((blocks + 255) / 256), dim3(256), 0, s, p);
  }

  void launch_blocked_match(const BlockedParams &p, hipStream_t s)
  {
    if (p.seedCount == 0) return;
    const dim3 grid((p.seedCount + 3) / 4), block(256);
    if (p.channels == 4) hipLaunchKernelGGL(k_blocked_match<4>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_blocked_match<3>, grid, block, 0, s, p);
  }

  void launch_blocked_fit_search(const BlockedParams &p, hipStream_t s)
  {
    if (p.nRegions == 0) return;
    if (p.channels == 4) hipLaunchKernelGGL(k_blocked_fit_search<4>, dim3(p.nRegions), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(k_blocked_fit_search<3>, dim3(p.nRegions), dim3(64), 0, s, p);
  }

  void launch_blocked_order(const BlockedParams &p, hipStream_t s)
  {
    if (p.nRegions == 0 || !p.order) return;
    hipLaunchKernelGGL(k_blocked_order, dim3(1), dim3(1024), 0, s, p);
  }

  void launch_blocked_store(const BlockedParams &p, hipStream_t s)
  {
    if (p.nRegions == 0) return;
    if (p.channels == 4) hipLaunchKernelGGL(k_blocked_store<4>, dim3(p.nRegions), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(k_blocked_store<3>, dim3(p.nRegions), dim3(64), 0, s, p);
  }
}

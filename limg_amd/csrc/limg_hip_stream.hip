// limg_hip_stream.hip -- the compact "LMG3" stream: pack (after a compact-mode encode) and decode kernels.
//
// Upstream has no serialised format and no limg_encode()/limg_decode(): `limg_encode3d_test` hands back
// debug planes.  This container holds exactly what the reference's decoder (a16, src/limg_decode.h:36-236) consumes -- the six
// int16 vectors of the block record, the shift triple and the crushed factor values at (8 - shift) bits each -- so that
//   decode_stream(encode_stream(image)) == the reference's pDecoded, bit for bit.
// One escape keeps that exact: with 4 channels the reference zeroes only the RGB normals of a factor whose shift is 8
// (src/limg_bit_crush_simd.h:589-609, src/limg_decode.h:150-170) and then multiplies the *raw* factor byte into the alpha lane
// (SURVEY.md 0.7); where that alpha normal is non-zero the raw byte is kept at 8 bits and flagged in the entry.
//
// Layout (little endian, every section 8-byte aligned):
//   limg_hip_stream_header (64 B) | limg_hip_stream_block[blocksX*blocksY] (56 B each, raster order) | payload words (8 B each)
// Payload of one block, at entry.payloadWord: factor A field, then B, then C; a field of b bits/pixel takes b words and stores
// pixel (row r, column x) of the 8x8 grid at bit (r*8 + x) * b -- i.e. every block row is exactly b bytes.  Pixels outside the
// image (partial edge blocks) are stored as 0.
//
// Work mapping.  Pack: tile = 256 consecutive blocks (raster order) per 256-thread workgroup; thread t prepares block t's entry; then
// each wave walks its 64 blocks in groups of 8 with lane = (block j = lane & 7, block row r = lane >> 3), so a lane owns the
// 8 pixels of one block row: 8 B of each factor plane, 32 B of the decoded image, b bytes of each payload field.
// Decode: the same lane mapping, but persistent -- a wave strides over units of 64 blocks on its own, nothing synchronises
// across waves (k_stream_decode).
#include "limg_hip_internal.h"

namespace limg_hip
{
  namespace
  {
    constexpr int kTile = 256;
    constexpr int kEntry = 56;
    constexpr int kGroupBytes = 8 * 192; // payload of 8 blocks, worst case
    constexpr int kPackedLimit = 2700;   // |record value| up to which the packed decode's 16-bit terms are exact (limg_hip_kernels.hip "a9, packed form": 3 * 2700 + 1 < 0x2000)
    typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

    __device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
    __device__ __forceinline__ int mad_i24(int a, int b, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
    __device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    __device__ __forceinline__ int med3_i32(int a, int b, int c) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
    __device__ __forceinline__ int add3(int a, int b, int c) { int r; asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
    __device__ __forceinline__ uint32_t bfe(uint32_t v, uint32_t off, uint32_t width) { uint32_t r; asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(off), "v"(width)); return r; }
    __device__ __forceinline__ uint32_t lshl_or(uint32_t a, uint32_t sh, uint32_t b) { uint32_t r; asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(sh), "v"(b)); return r; }
    __device__ __forceinline__ void wave_lds_fence()
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // (1 << s) + bias(s), src/limg_bit_crush_simd.h:611-619 / src/limg_decode.h:172-178
    __device__ __forceinline__ uint32_t shift_mul(uint32_t s)
    {
      return s < 4 ? (1u << s) : (s == 4 ? 17u : (s == 5 ? 36u : (s == 6 ? 85u : (s == 7 ? 255u : 256u))));
    }

    // bits per pixel of the three factor fields + raw-escape mask: bA | bB << 8 | bC << 16 | rawMask << 24.
    // mn3 / mx3: lane 3 (alpha) of dir{A,B,C}_{min|offset} / _{max|mag}.
    __device__ __forceinline__ uint32_t field_bits(uint32_t shiftWord, const int mn3[3], const int mx3[3], int channels)
    {
      uint32_t r = 0;
#pragma unroll
      for (int k = 0; k < 3; k++)
      {
        const uint32_t s = (shiftWord >> (8 * k)) & 0xFF;
        uint32_t b = s >= 8 ? 0u : 8u - s;
        if (s >= 8 && channels == 4 && mn3[k] != mx3[k]) { b = 8; r |= 1u << (24 + k); }
        r |= b << (8 * k);
      }
      return r;
    }

    __device__ __forceinline__ uint32_t words_of(uint32_t bits) { return (bits & 0xFF) + ((bits >> 8) & 0xFF) + ((bits >> 16) & 0xFF); }

    // ---- pack ------------------------------------------------------------------------------------------------------------

    // per tile: payload words
    __global__ __launch_bounds__(kTile) void k_stream_count(const StreamParams p)
    {
      __shared__ uint32_t sWave[4];
      const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
      const uint32_t g = blockIdx.x * kTile + tid;
      uint32_t words = 0;
      if (g < p.nBlocks)
      {
        const limg_hip_block_record &rec = p.records[g];
        const int mn3[3] = { rec.dirA_min[3], rec.dirB_offset[3], rec.dirC_offset[3] }, mx3[3] = { rec.dirA_max[3], rec.dirB_mag[3], rec.dirC_mag[3] };
        words = words_of(field_bits(p.shifts[g] & 0xFFFFFFu, mn3, mx3, (int)p.channels));
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) words += (uint32_t)__shfl_xor((int)words, off, 64);
      if (lane == 0) sWave[wave] = words;
      __syncthreads();
      if (tid == 0) p.tileBase[blockIdx.x] = sWave[0] + sWave[1] + sWave[2] + sWave[3];
    }

    // exclusive scan of the tile totals in place (one workgroup) + the header
    __global__ __launch_bounds__(1024) void k_stream_scan(const StreamParams p)
    {
      __shared__ unsigned long long sWave[16];
      __shared__ unsigned long long sCarry;
      const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
      if (tid == 0) sCarry = 0;
      __syncthreads();
      for (uint32_t base = 0; base < p.nTiles; base += 1024)
      {
        const uint32_t i = base + tid;
        const unsigned long long v = i < p.nTiles ? p.tileBase[i] : 0u;
        unsigned long long incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
        {
          const unsigned long long up = (unsigned long long)__shfl_up((long long)incl, off, 64);
          if (lane >= off) incl += up;
        }
        if (lane == 63) sWave[wave] = incl;
        __syncthreads();
        unsigned long long pre = sCarry;
        for (int w = 0; w < wave; w++) pre += sWave[w];
        // entry.payloadWord is 32 bits: the host refuses images whose worst-case payload would not fit (limg_hip_stream_bound)
        if (i < p.nTiles) p.tileBase[i] = (uint32_t)(pre + incl - v);
        __syncthreads();
        if (tid == 1023) sCarry = pre + incl;
        __syncthreads();
      }
      if (tid == 0)
      {
        limg_hip_stream_header h;
        h.magic = LIMG_HIP_STREAM_MAGIC; h.version = LIMG_HIP_STREAM_VERSION;
        h.sizeX = p.sizeX; h.sizeY = p.sizeY; h.channels = p.channels; h.errorFactor = p.errorFactor;
        h.blocksX = p.blocksX; h.blocksY = p.blocksY;
        h.payloadWords = sCarry;
        h.totalBytes = sizeof(limg_hip_stream_header) + (unsigned long long)p.nBlocks * kEntry + sCarry * 8ull;
        h.flags = p.flags; h.reserved[0] = h.reserved[1] = h.reserved[2] = 0;
        *reinterpret_cast<limg_hip_stream_header *>(p.stream) = h;
      }
    }

    __global__ __launch_bounds__(kTile) void k_stream_pack(const StreamParams p)
    {
      __shared__ uint32_t sBits[kTile], sOff[kTile], sWave[4];
      __shared__ __align__(16) uint32_t sEntry[kTile * kEntry / 4];
      __shared__ __align__(16) uint8_t sStage[4][kGroupBytes];
      const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
      const uint32_t tile = blockIdx.x, g = tile * kTile + tid;

      uint32_t bits = 0, words = 0;
      if (g < p.nBlocks)
      {
        const uint4 *rp = reinterpret_cast<const uint4 *>(p.records + g) + 1; // skip avg[4]
        const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2];                        // {dirA_min, dirA_max}, {dirB_offset, dirB_mag}, {dirC_offset, dirC_mag}
        const int mn3[3] = { (int)(int16_t)(r0.y >> 16), (int)(int16_t)(r1.y >> 16), (int)(int16_t)(r2.y >> 16) };
        const int mx3[3] = { (int)(int16_t)(r0.w >> 16), (int)(int16_t)(r1.w >> 16), (int)(int16_t)(r2.w >> 16) };
        const uint32_t sw = p.shifts[g] & 0xFFFFFFu;
        bits = field_bits(sw, mn3, mx3, (int)p.channels);
        words = words_of(bits);
        uint32_t *e = sEntry + tid * (kEntry / 4);
        e[0] = r0.x; e[1] = r0.y; e[2] = r0.z; e[3] = r0.w; e[4] = r1.x; e[5] = r1.y; e[6] = r1.z; e[7] = r1.w;
        e[8] = r2.x; e[9] = r2.y; e[10] = r2.z; e[11] = r2.w;
        e[12] = sw | (bits & 0xFF000000u);
      }
      uint32_t incl = words;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1)
      {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= off) incl += up;
      }
      if (lane == 63) sWave[wave] = incl;
      __syncthreads();
      uint32_t off = p.tileBase[tile] + incl - words;
      for (int w = 0; w < wave; w++) off += sWave[w];
      sBits[tid] = bits; sOff[tid] = off;
      if (g < p.nBlocks) sEntry[tid * (kEntry / 4) + 13] = off;
      __syncthreads();

      const uint32_t inTile = min((uint32_t)kTile, p.nBlocks - tile * kTile);
      {
        uint2 *dst = reinterpret_cast<uint2 *>(p.stream + sizeof(limg_hip_stream_header) + (size_t)tile * kTile * kEntry);
        const uint2 *src = reinterpret_cast<const uint2 *>(sEntry);
        for (uint32_t i = tid; i < inTile * (kEntry / 8); i += kTile) dst[i] = src[i];
      }

      uint2 *payload = reinterpret_cast<uint2 *>(p.stream + sizeof(limg_hip_stream_header) + (size_t)p.nBlocks * kEntry);
      const int j = lane & 7, r = lane >> 3;
      const bool aligned = (p.sizeX & 7u) == 0;
      uint8_t *stage = sStage[wave];
      for (int grp = 0; grp < 8; grp++)
      {
        const uint32_t jb = wave * 64 + grp * 8;
        if (jb >= inTile) break; // wave-uniform
        const uint32_t nValid = min(8u, inTile - jb);
        const uint32_t t = jb + j;
        const bool valid = (uint32_t)j < nValid;
        const uint32_t bw = valid ? sBits[t] : 0u, myOff = valid ? sOff[t] : 0u;
        const uint32_t off0 = sOff[jb];
        const uint32_t endWord = (uint32_t)__shfl((int)(myOff + words_of(bw)), (int)nValid - 1, 64);
        if (valid)
        {
          const uint32_t gg = tile * kTile + t, by = gg / p.blocksX, bx = gg - by * p.blocksX;
          const uint32_t y = by * 8 + r, x0 = bx * 8;
          uint32_t fieldByte = (myOff - off0) * 8;
#pragma unroll
          for (int k = 0; k < 3; k++)
          {
            const uint32_t b = (bw >> (8 * k)) & 0xFF;
            if (b == 0) continue;
            uint2 raw = make_uint2(0, 0);
            if (y < p.sizeY)
            {
              const uint8_t *src = p.fac[k] + (size_t)y * p.sizeX + x0;
              if (aligned) raw = *reinterpret_cast<const uint2 *>(src);
              else
              {
                const uint32_t nx = min(8u, p.sizeX - x0);
                unsigned long long acc = 0;
                for (uint32_t i = 0; i < nx; i++) acc |= (unsigned long long)src[i] << (8 * i);
                raw = make_uint2((uint32_t)acc, (uint32_t)(acc >> 32));
              }
            }
            // the planes hold (v << shift); raw-escaped fields hold the raw byte (stream mode of the encode kernel)
            const uint32_t sh = 8 - b;
            const unsigned long long bytes = ((unsigned long long)raw.y << 32) | raw.x;
            unsigned long long packed = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) packed |= (unsigned long long)(((uint32_t)(bytes >> (8 * i)) & 0xFFu) >> sh) << (i * b);
            uint8_t *dst = stage + fieldByte + r * b;
            for (uint32_t i = 0; i < b; i++) dst[i] = (uint8_t)(packed >> (8 * i));
            fieldByte += b * 8;
          }
        }
        wave_lds_fence();
        {
          const uint2 *src = reinterpret_cast<const uint2 *>(stage);
          const uint32_t n = endWord - off0;
          for (uint32_t i = lane; i < n; i += 64) payload[(size_t)off0 + i] = src[i];
        }
        wave_lds_fence();
      }
    }


    // ---- pack, strip form (round 6) -------------------------------------------------------------------------------------------
    // The three kernels above read the records twice and the shift words twice (64 MiB of records for a count), pack with lane = (block, row) on 64-byte row segments, do
    // the bit packing in 64-bit arithmetic pixel by pixel and store bytes into LDS one at a time.  For images of whole blocks the job is now:
    //  * nothing to count: the encode kernel's E step, which has every block's shifts in scalar registers, leaves the payload words of each of its work strips
    //    (<= 32 consecutive blocks of one block row) in EncodeParams::stripWords -- one more add into the counter it keeps for the dither calls;
    //  * k_stream_scan_strips: exclusive prefix over the strips in place (one workgroup, 8 strips per thread and round) + the header;
    //  * k_stream_pack_strips: persistent one-wave workgroups striding over the strips; no dependence between strips, so no ticket and no look-back.  lane = (block j
    //    = lane & 31, row half h = lane >> 5): a (plane, row) of the strip is 256 contiguous bytes, all 12 row loads of a lane are in flight before the first is
    //    used, and the NEXT strip's shift words and records are requested before this strip's rows are waited for.  A row's 8 values of b bits are squeezed with three
    //    mask-and-shift steps per dword (4 pixels at a time); the four rows of a lane are 4 b bytes = b dwords, appended to a 64-bit accumulator and flushed to the
    //    wave's LDS run a dword at a time (the two halves of a block own disjoint dwords); the run -- the strip's contiguous payload -- and the 32 entries leave with
    //    512-byte stores.
    // (A one-launch form with a decoupled look-back over per-unit totals was built first and measured: 0.27 ms -- at start-up every wave in flight walks back through
    // all its predecessors, 64 descriptors per round trip -- against 0.131 ms for the three kernels; `tools/r06/` keeps the numbers.)
    // Widths or heights that are not whole blocks keep the three-kernel form (byte gathers at the edges).
    // One workgroup.  A round covers 32 K strips as 8 slabs of 4096: thread t holds strips 4096 q + 4 t .. + 3 of every slab q -- each load and store is a fully
    // coalesced 16 bytes per lane (a thread that owned 32 CONSECUTIVE strips asked its CU's address unit for 64 lines per instruction: 11 us; the first form, one
    // slab per loop iteration with the loads inside the loop: 32 us) -- all 8 requested before anything is added; the 8 slabs' wave scans run side by side, the
    // 8 x 16 wave totals are turned into their exclusive prefix by one wave, two barriers per round.  8192^2 is one round.
    __global__ __launch_bounds__(1024) void k_stream_scan_strips(const StreamParams p)
    {
      __shared__ unsigned long long sPart[8 * 16]; // [slab][wave] totals, then their exclusive prefix
      __shared__ unsigned long long sCarry, sRound;
      const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
      if (tid == 0) sCarry = 0;
      __syncthreads();
      const bool vec = (p.nStrips & 3u) == 0; // 16-byte accesses need whole groups of 4 (the buffer itself is 256-byte aligned)
      for (uint32_t base = 0; base < p.nStrips; base += 32768)
      {
        uint32_t v[8][4];
#pragma unroll
        for (int q = 0; q < 8; q++)
        {
          const uint32_t i0 = base + (uint32_t)q * 4096u + (uint32_t)tid * 4u;
          if (vec)
          {
            uint4 t = make_uint4(0, 0, 0, 0);
            if (i0 < p.nStrips) t = *reinterpret_cast<const uint4 *>(p.stripWords + i0);
            v[q][0] = t.x; v[q][1] = t.y; v[q][2] = t.z; v[q][3] = t.w;
          }
          else
          {
#pragma unroll
            for (int k = 0; k < 4; k++) v[q][k] = i0 + k < p.nStrips ? p.stripWords[i0 + k] : 0u;
          }
        }
        // (a slab's 4096 strips hold at most 4096 x 32 x 24 words: 32 bits, and the wave scans are DPP adds -- eight 64-bit shuffle scans per wave went through the
        // one LDS of the CU this kernel runs on and cost as much as everything else in it)
        uint32_t mine[8], incl[8];
#pragma unroll
        for (int q = 0; q < 8; q++) incl[q] = mine[q] = v[q][0] + v[q][1] + v[q][2] + v[q][3];
#pragma unroll
        for (int q = 0; q < 8; q++)
        {
          int x = (int)incl[q];
          x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false); // row_shr:1
          x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false); // row_shr:2
          x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false); // row_shr:4
          x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false); // row_shr:8
          x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
          x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
          incl[q] = (uint32_t)x;
        }
        if (lane == 63)
#pragma unroll
          for (int q = 0; q < 8; q++) sPart[q * 16 + wave] = incl[q];
        __syncthreads();
        if (wave == 0)
        { // exclusive prefix over the 128 partial sums (slab-major = strip order): a lane owns two neighbours
          const unsigned long long a0 = sPart[2 * lane], a1 = sPart[2 * lane + 1];
          unsigned long long in2 = a0 + a1;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1)
          {
            const unsigned long long up = (unsigned long long)__shfl_up((long long)in2, off, 64);
            if (lane >= off) in2 += up;
          }
          const unsigned long long ex = in2 - a0 - a1;
          sPart[2 * lane] = ex; sPart[2 * lane + 1] = ex + a0;
          if (lane == 63) sRound = in2;
        }
        __syncthreads();
        const unsigned long long carry = sCarry;
#pragma unroll
        for (int q = 0; q < 8; q++)
        {
          const uint32_t i0 = base + (uint32_t)q * 4096u + (uint32_t)tid * 4u;
          // entry.payloadWord is 32 bits: the host refuses images whose worst-case payload would not fit (limg_hip_stream_bound)
          const uint32_t e0 = (uint32_t)(carry + sPart[q * 16 + wave] + incl[q] - mine[q]);
          const uint32_t e1 = e0 + v[q][0], e2 = e1 + v[q][1], e3 = e2 + v[q][2];
          if (vec)
          {
            if (i0 < p.nStrips) *reinterpret_cast<uint4 *>(p.stripWords + i0) = make_uint4(e0, e1, e2, e3);
          }
          else
          {
            const uint32_t e[4] = { e0, e1, e2, e3 };
#pragma unroll
            for (int k = 0; k < 4; k++)
              if (i0 + k < p.nStrips) p.stripWords[i0 + k] = e[k];
          }
        }
        __syncthreads();
        if (tid == 0) sCarry = carry + sRound;
        __syncthreads();
      }
      if (tid == 0)
      {
        limg_hip_stream_header h;
        h.magic = LIMG_HIP_STREAM_MAGIC; h.version = LIMG_HIP_STREAM_VERSION;
        h.sizeX = p.sizeX; h.sizeY = p.sizeY; h.channels = p.channels; h.errorFactor = p.errorFactor;
        h.blocksX = p.blocksX; h.blocksY = p.blocksY;
        h.payloadWords = sCarry;
        h.totalBytes = sizeof(limg_hip_stream_header) + (unsigned long long)p.nBlocks * kEntry + sCarry * 8ull;
        h.flags = p.flags; h.reserved[0] = h.reserved[1] = h.reserved[2] = 0;
        *reinterpret_cast<limg_hip_stream_header *>(p.stream) = h;
      }
    }

    // 8 bytes (lo = pixels 0..3, hi = 4..7), each holding its value in the TOP b bits (sh = 8 - b; raw-escaped fields: b = 8) -> the 8 values in 8 b consecutive bits
    __device__ __forceinline__ unsigned long long squeeze_row(uint32_t lo, uint32_t hi, uint32_t sh, uint32_t b, uint32_t m4, uint32_t mPair)
    {
      uint32_t z[2];
#pragma unroll
      for (int h = 0; h < 2; h++)
      {
        const uint32_t x = ((h ? hi : lo) >> sh) & m4;                       // v0 | v1 << 8 | v2 << 16 | v3 << 24
        const uint32_t y = ((x >> sh) & mPair) | (x & 0x00FF00FFu);           // v0 | v1 << b in the low half, v2 | v3 << b in the high half
        z[h] = (y & 0xFFFFu) | ((y >> 16) << (2u * b));                      // 4 b bits
      }
      return (unsigned long long)z[0] | ((unsigned long long)z[1] << (4u * b));
    }

    struct StripSmall { uint4 r0, r1, r2; uint32_t sw, base; };
    __device__ __forceinline__ void load_strip_small(const StreamParams &p, uint32_t strip, int j, StripSmall &o)
    {
      o.r0 = o.r1 = o.r2 = make_uint4(0, 0, 0, 0); o.sw = 0; o.base = 0;
      if (strip >= p.nStrips) return;
      const uint32_t by = strip / p.stripsX, sx = strip - by * p.stripsX, bx = sx * 32u + (uint32_t)j;
      o.base = p.stripWords[strip];
      if (bx < p.blocksX)
      {
        const size_t g = (size_t)by * p.blocksX + bx;
        const uint4 *rp = reinterpret_cast<const uint4 *>(p.records + g) + 1; // skip avg[4]
        o.r0 = rp[0]; o.r1 = rp[1]; o.r2 = rp[2];
        o.sw = p.shifts[g] & 0xFFFFFFu;
      }
    }

    struct StripRows { uint2 raw[3][4]; uint32_t bits; };
    // field sizes of the lane's block + the request for its four rows of every field the block has
    __device__ __forceinline__ void issue_strip_rows(const StreamParams &p, uint32_t strip, int j, int h, const StripSmall &sm, StripRows &o)
    {
      o.bits = 0;
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int r = 0; r < 4; r++) o.raw[k][r] = make_uint2(0u, 0u);
      if (strip >= p.nStrips) return;
      const uint32_t by = strip / p.stripsX, sx = strip - by * p.stripsX, bx = sx * 32u + (uint32_t)j;
      if (bx >= p.blocksX) return;
      const int mn3[3] = { (int)(int16_t)(sm.r0.y >> 16), (int)(int16_t)(sm.r1.y >> 16), (int)(int16_t)(sm.r2.y >> 16) };
      const int mx3[3] = { (int)(int16_t)(sm.r0.w >> 16), (int)(int16_t)(sm.r1.w >> 16), (int)(int16_t)(sm.r2.w >> 16) };
      o.bits = field_bits(sm.sw, mn3, mx3, (int)p.channels);
#pragma unroll
      for (int k = 0; k < 3; k++)
      {
        if (((o.bits >> (8 * k)) & 0xFFu) == 0u) continue;
        const uint8_t *src = p.fac[k] + (size_t)(by * 8u + (uint32_t)h * 4u) * p.sizeX + bx * 8u;
#pragma unroll
        for (int r = 0; r < 4; r++) o.raw[k][r] = *reinterpret_cast<const uint2 *>(src + (size_t)r * p.sizeX);
      }
    }

    // Software pipeline, two strips deep: while strip i is packed, the rows of strip i + 1 and the shift words / records of strip i + 2 are in flight.
    __global__ __launch_bounds__(64) void k_stream_pack_strips(const StreamParams p)
    {
      __shared__ __align__(16) uint32_t sRun[32 * 48];   // the strip's payload, worst case (24 words per block)
      __shared__ __align__(16) uint32_t sEnt[32 * 14];   // its entries
      const int lane = lane_id(), j = lane & 31, h = lane >> 5;
      uint2 *const payload = reinterpret_cast<uint2 *>(p.stream + sizeof(limg_hip_stream_header) + (size_t)p.nBlocks * kEntry);
      StripSmall cur, nxt;
      StripRows rows, rowsNext;
      load_strip_small(p, blockIdx.x, j, cur);
      load_strip_small(p, blockIdx.x + p.nWaves, j, nxt);
      issue_strip_rows(p, blockIdx.x, j, h, cur, rows);
      for (uint32_t strip = blockIdx.x; strip < p.nStrips; strip += p.nWaves)
      {
        issue_strip_rows(p, strip + p.nWaves, j, h, nxt, rowsNext); // (waits for `nxt`, requested one iteration ago)
        StripSmall nxt2;
        load_strip_small(p, strip + 2u * p.nWaves, j, nxt2);
        const uint32_t by = strip / p.stripsX, sx = strip - by * p.stripsX;
        const uint32_t inStrip = min(32u, p.blocksX - sx * 32u);
        const uint32_t bits = rows.bits, words = words_of(bits);
        // exclusive prefix of the words over the strip's blocks (both halves compute it)
        uint32_t incl = words;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1)
        {
          const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 32);
          if (j >= off) incl += up;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, 31, 32);
        const uint32_t excl = incl - words;
        if (h == 0)
        {
          uint32_t *e = sEnt + j * 14;
          *reinterpret_cast<uint2 *>(e + 0) = make_uint2(cur.r0.x, cur.r0.y); *reinterpret_cast<uint2 *>(e + 2) = make_uint2(cur.r0.z, cur.r0.w);
          *reinterpret_cast<uint2 *>(e + 4) = make_uint2(cur.r1.x, cur.r1.y); *reinterpret_cast<uint2 *>(e + 6) = make_uint2(cur.r1.z, cur.r1.w);
          *reinterpret_cast<uint2 *>(e + 8) = make_uint2(cur.r2.x, cur.r2.y); *reinterpret_cast<uint2 *>(e + 10) = make_uint2(cur.r2.z, cur.r2.w);
          *reinterpret_cast<uint2 *>(e + 12) = make_uint2(cur.sw | (bits & 0xFF000000u), cur.base + excl);
        }
        {
          uint32_t *field = sRun + 2u * excl; // dwords
#pragma unroll
          for (int k = 0; k < 3; k++)
          {
            const uint32_t b = (bits >> (8 * k)) & 0xFFu;
            if (b == 0u) continue;
            const uint32_t sh = 8u - b, m1 = (1u << b) - 1u, m4 = m1 * 0x01010101u, mPair = (m1 * 0x00010001u) << b;
            uint32_t *dst = field + (uint32_t)h * b; // rows 4 h .. 4 h + 3 are bytes [4 h b, 4 h b + 4 b) of the field: b dwords
            const uint32_t len0 = b < 4u ? b : 4u, len1 = b - len0; // a row goes in as its low (up to) 4 bytes, then the rest: never more than 3 + 4 bytes in acc
            unsigned long long acc = 0;
            uint32_t fill = 0; // bytes in acc (< 4 between appends)
#pragma unroll
            for (int r = 0; r < 4; r++)
            {
              const unsigned long long v = squeeze_row(rows.raw[k][r].x, rows.raw[k][r].y, sh, b, m4, mPair);
              acc |= (unsigned long long)(uint32_t)v << (8u * fill);
              fill += len0;
              if (fill >= 4u) { *dst++ = (uint32_t)acc; acc >>= 32; fill -= 4u; }
              acc |= (unsigned long long)(uint32_t)(v >> 32) << (8u * fill);
              fill += len1;
              if (fill >= 4u) { *dst++ = (uint32_t)acc; acc >>= 32; fill -= 4u; }
            }
            field += 2u * b;
          }
        }
        wave_lds_fence();
        {
          uint2 *edst = reinterpret_cast<uint2 *>(p.stream + sizeof(limg_hip_stream_header) + ((size_t)by * p.blocksX + sx * 32u) * kEntry);
          const uint2 *esrc = reinterpret_cast<const uint2 *>(sEnt);
          for (uint32_t i = lane; i < inStrip * (kEntry / 8); i += 64) edst[i] = esrc[i];
          // the run in 16-byte stores: the payload area is 8-byte aligned, so a run that starts on an odd word sends that word ahead (the LDS side is then read at
          // 8-byte alignment, which ds_read_b128 does not allow: two ds_read_b64)
          uint2 *pdst = payload + (size_t)cur.base;
          const uint2 *psrc = reinterpret_cast<const uint2 *>(sRun);
          const uint32_t odd = (uint32_t)((reinterpret_cast<uintptr_t>(pdst) >> 3) & 1u) & (total ? 1u : 0u);
          if (odd && lane == 0) pdst[0] = psrc[0];
          const uint32_t pairs = (total - odd) >> 1;
          uint4 *p4 = reinterpret_cast<uint4 *>(pdst + odd);
          for (uint32_t i = lane; i < pairs; i += 64)
          {
            const uint2 a = psrc[odd + 2u * i], b2 = psrc[odd + 2u * i + 1u];
            p4[i] = make_uint4(a.x, a.y, b2.x, b2.y);
          }
          if (((total - odd) & 1u) && lane == 0) pdst[total - 1u] = psrc[total - 1u];
        }
        wave_lds_fence(); // the run is read: the next strip may overwrite it
        cur = nxt; nxt = nxt2; rows = rowsNext;
      }
    }

    // ---- decode ----------------------------------------------------------------------------------------------------------
    // PERSISTENT since round 5.  The work unit is what a wave always owned: 64 consecutive blocks (raster order), walked in 8 groups of 8 with lane = (block j = lane & 7,
    // block row r = lane >> 3).  Until round 4 a workgroup was 4 such waves behind one barrier and lived for one tile of 256 blocks: header check, entry loads (56-byte
    // records, one per lane), their turn into decode constants, barrier, and only then the first payload request -- a serial prologue per workgroup, 4096 of them in four
    // rounds over the residency slots (VALU busy 0.71 at 16 waves per CU).  Now a wave loops over units with NO workgroup-level synchronisation at all (its LDS is its
    // own): the NEXT unit's entries are requested before the current unit's groups are decoded, and a unit's first payload run is requested as soon as its offsets are
    // known, ahead of the (long) constant preparation.  Units go to waves with a fixed stride: there is no dependence between units, so no ticket is needed.
    // The re-expansion multiplier is folded into the normals (value * (mul * n) == (value * mul) * n exactly: 8 + 21 bits for records the packed form accepts);
    // blocks with larger records (never from a fit of byte pixels) keep both apart and take the generic loop.
    // a16 for the 8 pixels of one block row in the packed form of the F step (limg_hip_kernels.hip phase_f_rows), factor by factor: per factor and pixel one v_bfe_u32 and
    // three 24-bit multiply-adds (the re-expansion multiplier sits in the normals).  The terms never live as 32-bit values: one v_perm_b32 packs a pixel's R and G terms
    // (>> 8 included) into the halves of a register, another the B terms of a PAIR of pixels; the additive constants carry biases (0x3000 + 0x3000 + 0x2000 = 0x8000
    // over the three factors) so that plain 32-bit adds sum the halves independently and the sums are the estimates in offset binary, which unsigned packed max / min
    // clamp.  ALPHA: some block of the group has a varying alpha lane: that lane as 32-bit terms; otherwise it is one value per block (alphaRep: the byte
    // at bits 8..15 and 24..31).  A template on ALPHA, not a run-time test inside the loops: the test per (factor, pixel) compiled to 24 scalar branches per row.
    template <bool ALPHA>
    __device__ __forceinline__ void decode_row_packed(const int *nm, const uint32_t lo[3], const uint32_t hi[3], const uint32_t bb[3], uint32_t alphaRep, uint32_t px[8])
    {
      uint32_t accRG[8], accBB[4];
      int accA[8];
#pragma unroll
      for (int k = 0; k < 3; k++)
      {
        const int4 n = *reinterpret_cast<const int4 *>(nm + 4 * k), m = *reinterpret_cast<const int4 *>(nm + 12 + 4 * k);
#pragma unroll
        for (int i = 0; i < 8; i += 2)
        {
          const int d0 = (int)bfe(i < 4 ? lo[k] : hi[k], (uint32_t)(i & 3) * bb[k], bb[k]), d1 = (int)bfe(i < 4 ? lo[k] : hi[k], (uint32_t)((i + 1) & 3) * bb[k], bb[k]);
          const int r0 = mad_i24(d0, n.x, m.x), g0 = mad_i24(d0, n.y, m.y), b0 = mad_i24(d0, n.z, m.z);
          const int r1 = mad_i24(d1, n.x, m.x), g1 = mad_i24(d1, n.y, m.y), b1 = mad_i24(d1, n.z, m.z);
          const uint32_t rg0 = __builtin_amdgcn_perm((uint32_t)g0, (uint32_t)r0, 0x06050201u), rg1 = __builtin_amdgcn_perm((uint32_t)g1, (uint32_t)r1, 0x06050201u);
          const uint32_t bbp = __builtin_amdgcn_perm((uint32_t)b1, (uint32_t)b0, 0x06050201u);
          if (k == 0) { accRG[i] = rg0; accRG[i + 1] = rg1; accBB[i >> 1] = bbp; }
          else { accRG[i] += rg0; accRG[i + 1] += rg1; accBB[i >> 1] += bbp; }
          if (ALPHA)
          {
            const int a0 = mad_i24(d0, n.w, m.w) >> 8, a1 = mad_i24(d1, n.w, m.w) >> 8;
            if (k == 0) { accA[i] = a0; accA[i + 1] = a1; } else { accA[i] += a0; accA[i + 1] += a1; }
            asm volatile("" : "+v"(accA[i]), "+v"(accA[i + 1]));
          }
          asm volatile("" : "+v"(accRG[i]), "+v"(accRG[i + 1]), "+v"(accBB[i >> 1])); // materialised here (otherwise the packing sinks to the next factor's adds and the products stay live)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 8; i += 2)
      {
        ushort2_t bv = __builtin_bit_cast(ushort2_t, accBB[i >> 1]); // B estimates of the pair + 0x8000
        bv = __builtin_elementwise_max(bv, __builtin_bit_cast(ushort2_t, 0x80008000u));
        bv = __builtin_elementwise_min(bv, __builtin_bit_cast(ushort2_t, 0x80FF80FFu));
        uint32_t ba = __builtin_bit_cast(uint32_t, bv); // bytes: B of pixel i, 0x80, B of pixel i + 1, 0x80
        if (!ALPHA) ba = (ba & 0x00FF00FFu) | alphaRep;  // ... the block's alpha value in the odd bytes
#pragma unroll
        for (int q = 0; q < 2; q++)
        {
          ushort2_t ev = __builtin_bit_cast(ushort2_t, accRG[i + q]); // R and G estimates + 0x8000
          ev = __builtin_elementwise_max(ev, __builtin_bit_cast(ushort2_t, 0x80008000u));
          ev = __builtin_elementwise_min(ev, __builtin_bit_cast(ushort2_t, 0x80FF80FFu));
          if (!ALPHA) px[i + q] = __builtin_amdgcn_perm(ba, __builtin_bit_cast(uint32_t, ev), q == 0 ? 0x05040200u : 0x07060200u); // R, G = low bytes of ev's halves; B, A from `ba`
          else px[i + q] = lshl_or((uint32_t)med3_i32(accA[i + q], 0, 255), 24u, __builtin_amdgcn_perm(ba, __builtin_bit_cast(uint32_t, ev), q == 0 ? 0x0C040200u : 0x0C060200u));
        }
      }
    }

    struct DecodeWaveLds
    {
      int nm[64][24];           // per block: 12 effective normals (multiplier folded in unless the block is `big`), 12 additive constants
      uint32_t bits[64], off[64], mul[64], bx[64], by[64], flags[64];
      uint8_t stage[kGroupBytes + 16];
    };
    static_assert(sizeof(DecodeWaveLds) * 4 <= 163840 / 4, "4 workgroups of 4 waves per CU");

    __device__ __forceinline__ void load_entry(const DecodeParams &p, uint32_t g, uint32_t e[14])
    {
      if (g < p.nBlocks)
      {
        const uint2 *ep = reinterpret_cast<const uint2 *>(p.stream + sizeof(limg_hip_stream_header) + (size_t)g * kEntry);
#pragma unroll
        for (int i = 0; i < 7; i++) { const uint2 v = ep[i]; e[2 * i] = v.x; e[2 * i + 1] = v.y; }
      }
      else
      {
#pragma unroll
        for (int i = 0; i < 14; i++) e[i] = 0u;
      }
    }

    __global__ __launch_bounds__(kTile, 4) void k_stream_decode(const DecodeParams p)
    {
      __shared__ __align__(16) DecodeWaveLds sW[4];
      const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
      DecodeWaveLds &S = sW[wave];

      // every wave validates the header it is about to trust (scalar loads; a mismatch raises the context's status word)
      const limg_hip_stream_header *h = reinterpret_cast<const limg_hip_stream_header *>(p.stream);
      const unsigned long long payloadWords = h->payloadWords;
      const bool ok = h->magic == LIMG_HIP_STREAM_MAGIC && h->version == LIMG_HIP_STREAM_VERSION && h->sizeX == p.sizeX && h->sizeY == p.sizeY &&
                      h->blocksX == p.blocksX && h->blocksY == p.blocksY && (h->channels == 3 || h->channels == 4) &&
                      payloadWords <= (unsigned long long)p.nBlocks * 24ull && // 3 fields x 8 words at most per block: bounds the product below
                      sizeof(limg_hip_stream_header) + (unsigned long long)p.nBlocks * kEntry + payloadWords * 8ull <= p.streamBytes;
      if (!ok)
      {
        if (tid == 0) atomicOr(p.status, 1u);
        return;
      }
      const int channels = (int)h->channels;
      const uint32_t nUnits = (p.nBlocks + 63u) / 64u, nWaves = gridDim.x * 4u;
      uint32_t unit = blockIdx.x * 4u + (uint32_t)wave;
      if (unit >= nUnits) return; // (wave-uniform; nothing below synchronises across waves)
      const uint2 *payload = reinterpret_cast<const uint2 *>(p.stream + sizeof(limg_hip_stream_header) + (size_t)p.nBlocks * kEntry);
      const int j = lane & 7, r = lane >> 3;
      uint8_t *stage = S.stage;
      const bool rowAligned = (p.sizeX & 3u) == 0;

      uint32_t e[14];
      load_entry(p, unit * 64u + (uint32_t)lane, e);
      for (;;)
      {
        const uint32_t g = unit * 64u + (uint32_t)lane;
        const uint32_t inUnit = min(64u, p.nBlocks - unit * 64u);
        const uint32_t sw = e[12];
        // ---- 1. what the first payload request needs: field widths and offsets ----
        {
          uint32_t bits = 0;
#pragma unroll
          for (int f = 0; f < 3; f++)
          {
            const uint32_t s = min((sw >> (8 * f)) & 0xFFu, 8u);
            const bool raw = (sw >> (24 + f)) & 1u;
            bits |= (s == 8 ? (raw ? 8u : 0u) : 8u - s) << (8 * f);
          }
          S.bits[lane] = bits; S.off[lane] = e[13];
        }
        wave_lds_fence();
        // Per group of 8 blocks: where its payload run lies.  The run of group g + 1 is requested (into registers) before group g
        // is decoded, so the HBM round trip hides behind ~400 VALU instructions.
        struct Group { uint32_t t, bw, myOff, off0, n; bool valid, any, ok; };
        auto group_info = [&](int grp) {
          Group G;
          const uint32_t jb = (uint32_t)grp * 8u;
          G.any = grp < 8 && jb < inUnit; // wave-uniform
          G.t = jb + j; G.bw = 0; G.myOff = 0; G.off0 = 0; G.n = 0; G.valid = false; G.ok = false;
          if (!G.any) return G;
          const uint32_t nValid = min(8u, inUnit - jb);
          G.valid = (uint32_t)j < nValid;
          G.bw = G.valid ? S.bits[G.t] : 0u;
          G.myOff = G.valid ? S.off[G.t] : 0u;
          G.off0 = S.off[jb];
          // the group's payload is one contiguous run in a stream this library wrote; anything else (corrupt offsets) is refused
          // All of this in 64 bits: offsets come from the (untrusted) stream, and 32-bit sums such as 0xFFFFFFF0 + 24 wrap to small values that pass.
          const unsigned long long myEnd = (unsigned long long)G.myOff + words_of(G.bw);
          const uint32_t lastOff = (uint32_t)__shfl((int)G.myOff, (int)nValid - 1, 64), lastWords = (uint32_t)__shfl((int)words_of(G.bw), (int)nValid - 1, 64);
          const unsigned long long endWord = (unsigned long long)lastOff + lastWords;
          const bool sane = G.myOff >= G.off0 && myEnd <= endWord && endWord >= G.off0 && endWord - G.off0 <= (unsigned long long)(kGroupBytes / 8) && endWord <= payloadWords;
          G.ok = __builtin_amdgcn_ballot_w64(G.valid && !sane) == 0;
          G.n = G.ok ? (uint32_t)(endWord - G.off0) : 0u;
          return G;
        };
        // The payload run of a group: up to three 8-byte loads per lane, issued as inline assembly and waited for with an explicit COUNTED s_waitcnt.  Why by hand: vmcnt counts loads and stores alike and retires them in order, so "wait until at most
        // two are outstanding" right behind a group's two stores means "my three loads, issued before them, are back" -- while the compiler's own bookkeeping for a load
        // that is consumed across the loop's back edge ends in vmcnt(0), which also drains those stores: every wave then sat out the HBM write latency once per group
        // (SQ_WAIT_ANY: 0.51 of the waves' cycles).  The compiler believes the three registers are defined where the asm statement stands; nothing may read them before
        // the hand-written s_waitcnt -- the only reader is stage_run() below, and tests/test_decode_isa.py proves on the compiled assembly that nothing else is.
        auto fetch = [&](const Group &G, unsigned long long buf[3]) {
#pragma unroll
          for (int i = 0; i < 3; i++)
          {
            if ((uint32_t)(lane + 64 * i) < G.n) // (G.n: validated against the payload's size in group_info; a typical run needs the first of the three only)
            {
              const uint2 *src = payload + ((size_t)G.off0 + (uint32_t)(lane + 64 * i));
              asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(buf[i]) : "v"(src) : "memory");
            }
          }
        };
        auto stage_run = [&](const Group &G, const unsigned long long buf[3]) {
          unsigned long long *dst = reinterpret_cast<unsigned long long *>(stage);
#pragma unroll
          for (int i = 0; i < 3; i++)
            if ((uint32_t)(lane + 64 * i) < G.n) dst[lane + 64 * i] = buf[i];
        };
        Group cur = group_info(0);
        unsigned long long buf[3];
        fetch(cur, buf);

        // ---- 2. the block's decode constants (lane == block of the unit), under the first payload request ----
        if (g < p.nBlocks)
        {
          uint32_t mul = 0, big = 0;
          // (pass 1: is any record value beyond the packed form's range?  decides whether the multiplier may be folded into the normals)
#pragma unroll
          for (int w = 0; w < 12; w++)
          {
            const int lo16 = (int)(int16_t)e[w], hi16 = (int)(int16_t)(e[w] >> 16);
            big |= (lo16 > kPackedLimit || lo16 < -kPackedLimit || hi16 > kPackedLimit || hi16 < -kPackedLimit) ? 1u : 0u;
          }
#pragma unroll
          for (int f = 0; f < 3; f++)
          {
            const uint32_t s = min((sw >> (8 * f)) & 0xFFu, 8u);
            const int fmul = (int)shift_mul(s);
            mul |= (big ? (uint32_t)fmul : 1u) << (10 * f);
#pragma unroll
            for (int c = 0; c < 4; c++)
            {
              // vector f: min/offset at int16 index f*8 + c, max/mag at f*8 + 4 + c
              const int mnv = (int)(int16_t)(e[f * 4 + (c >> 1)] >> (16 * (c & 1)));
              const int mxv = (int)(int16_t)(e[f * 4 + 2 + (c >> 1)] >> (16 * (c & 1)));
              int n = mxv - mnv, m = mnv;
              if (c < 3)
              {
                if (s > 7) { n = 0; if (f > 0) m = 0; } // src/limg_decode.h:150-170
              }
              else if (channels == 3) { n = 0; m = 0xFFFF; } // src/limg_decode.h:95-97
              S.nm[lane][f * 4 + c] = big ? n : n * fmul; // |n| <= 5400 and mul <= 256: 21 bits, a 24-bit operand
              S.nm[lane][12 + f * 4 + c] = (int)(((uint32_t)m << 8) + 128u + (uint32_t)(c < 3 ? (f == 2 ? 0x200000 : 0x300000) : 0)); // R, G and B carry the packed form's biases
            }
          }
          { // per-block flags of the packed decode (same rules as the F step's phase_f_prepare, limg_hip_kernels.hip)
            uint32_t fl = big;
            if (channels == 3) fl |= 255u << 8;
            else
            {
              const int a0 = (int)(int16_t)(e[1] >> 16), a1 = (int)(int16_t)(e[3] >> 16), b0 = (int)(int16_t)(e[5] >> 16), b1 = (int)(int16_t)(e[7] >> 16), c0 = (int)(int16_t)(e[9] >> 16),
                        c1 = (int)(int16_t)(e[11] >> 16); // lane 3 of dirA_min, dirA_max, dirB_offset, dirB_mag, dirC_offset, dirC_mag
              int a = a0 + b0 + c0;
              a = a < 0 ? 0 : (a > 255 ? 255 : a);
              fl |= (a0 != a1 || b0 != b1 || c0 != c1) ? 2u : ((uint32_t)a << 8);
            }
            S.flags[lane] = fl;
          }
          S.mul[lane] = mul;
          const uint32_t by = g / p.blocksX;
          S.by[lane] = by; S.bx[lane] = g - by * p.blocksX;
        }
        // ---- 3. the first run into the LDS stage (the one full wait per unit: nothing counted lies between its loads and here), then the NEXT unit's entries:
        //         in flight while this unit's groups are decoded ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage_run(cur, buf);
        const uint32_t next = unit + nWaves;
        uint32_t en[14];
        if (next < nUnits) load_entry(p, next * 64u + (uint32_t)lane, en);
        wave_lds_fence();

        for (int grp = 0; grp < 8; grp++)
        {
          if (!cur.any) break; // wave-uniform
          const Group G = cur; // its run is in the stage
          cur = group_info(grp + 1);
          fetch(cur, buf); // the next group's run: three loads, then (below) this group's two stores, then the counted wait
          // inconsistent offsets: the stream is refused (status word) and the group decodes as if it had no blocks -- every lane then unpacks nothing and stores to the
          // sink.  No branch around the decode: the two stores below must lie on EVERY path between the payload loads above and the counted wait behind them.
          if (!G.ok && lane == 0) atomicOr(p.status, 2u);
          const uint32_t t = G.t, bw = G.ok ? G.bw : 0u, off0 = G.off0, myOff = G.ok ? G.myOff : off0;
          const bool valid = G.valid && G.ok;
          const uint32_t fl = valid ? S.flags[t] : 0u;
          const bool generic = __builtin_amdgcn_ballot_w64((fl & 1u) != 0u) != 0ull, anyAlpha = __builtin_amdgcn_ballot_w64((fl & 2u) != 0u) != 0ull; // wave-uniform
          wave_lds_fence();
          { // (no `if (valid)` around this: lanes without a block decode zeros and store to the sink -- see the stores below)
            const uint32_t y = S.by[t] * 8 + r, x0 = S.bx[t] * 8;
            uint32_t fieldByte = (myOff - off0) * 8;
            unsigned long long packed[3];
            uint32_t bb[3];
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
              const uint32_t b = (bw >> (8 * k)) & 0xFF;
              bb[k] = b;
              const uint32_t o = fieldByte + r * b;
              const uint32_t *wp = reinterpret_cast<const uint32_t *>(stage + (o & ~3u));
              const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2];
              const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, o & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, o & 3u);
              packed[k] = ((unsigned long long)hi << 32) | lo;
              fieldByte += b * 8;
            }
            const int *nm = S.nm[t];
            // values 0..3 of a row sit in the low dword (4 b <= 32 bits), values 4..7 in the low dword of (packed >> 4 b)
            uint32_t lo[3], hi[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { lo[k] = (uint32_t)packed[k]; hi[k] = (uint32_t)(packed[k] >> (4 * bb[k])); }
            uint32_t px[8];
            if (!generic)
            {
              if (anyAlpha) decode_row_packed<true>(nm, lo, hi, bb, 0u, px);
              else decode_row_packed<false>(nm, lo, hi, bb, (fl & 0xFF00u) * 0x10001u, px);
            }
            else
            { // record values beyond the packed form's range somewhere in this group (never from a fit of byte pixels): 32-bit terms, the low 32 bits of the products
              // like PMULLD; a block of the group that is NOT beyond the range has its multiplier in its normals and 1 here
              const uint32_t mulw = S.mul[t];
              const uint32_t mulA = mulw & 0x3FF, mulB = (mulw >> 10) & 0x3FF, mulC = (mulw >> 20) & 0x3FF;
#pragma unroll
              for (int i = 0; i < 8; i++)
              {
                const uint32_t sel = (uint32_t)i & 3u;
                const int dA = (int)mul_u24(bfe(i < 4 ? lo[0] : hi[0], sel * bb[0], bb[0]), mulA);
                const int dB = (int)mul_u24(bfe(i < 4 ? lo[1] : hi[1], sel * bb[1], bb[1]), mulB);
                const int dC = (int)mul_u24(bfe(i < 4 ? lo[2] : hi[2], sel * bb[2], bb[2]), mulC);
                uint32_t out = 0;
#pragma unroll
                for (int c = 0; c < 4; c++)
                {
                  const int bA = c < 3 ? 0x300000 : 0, bC = c < 3 ? 0x200000 : 0;
                  const int est = add3(mad_i24(dA, nm[c], nm[12 + c] - bA) >> 8, mad_i24(dB, nm[4 + c], nm[16 + c] - bA) >> 8, mad_i24(dC, nm[8 + c], nm[20 + c] - bC) >> 8);
                  out |= (uint32_t)med3_i32(est, 0, 255) << (8 * c);
                }
                px[i] = out;
              }
            }
            // EVERY lane issues exactly two 16-byte stores per group, on every path: a block row wholly inside the image to its place, anything else to the sink.
            // With the three payload loads above equally unconditional, the wait for the next run is a counted one (vmcnt(2): "all but my last two stores")
            // and no longer drains the stores of the group just decoded -- which is what the waves spent half their life on (SQ_WAIT_ANY 0.51 of their cycles).
            const bool inImage = valid && y < p.sizeY;
            const bool whole = inImage && rowAligned && x0 + 8 <= p.sizeX;
            uint32_t *dst = whole ? p.out + (size_t)y * p.sizeX + x0 : p.sink + lane * 8;
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4_t{ px[0], px[1], px[2], px[3] }, reinterpret_cast<u32x4_t *>(dst));     // (written once, never read here: kept out of the L2's way,
            __builtin_nontemporal_store(u32x4_t{ px[4], px[5], px[6], px[7] }, reinterpret_cast<u32x4_t *>(dst) + 1); //  like the encoder's planes)
            if (inImage && !whole)
            { // partial edge blocks, rows that are not 16-byte aligned: pixel by pixel
              uint32_t *row = p.out + (size_t)y * p.sizeX + x0;
#pragma unroll
              for (int i = 0; i < 8; i++)
                if (x0 + i < p.sizeX) row[i] = px[i];
            }
          }
          wave_lds_fence(); // every lane is done reading this group's run
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); // all but the two stores above (more on the edge path: then this waits for a few stores too): the run of `cur` is in
          stage_run(cur, buf);
          wave_lds_fence();
        }
        if (next >= nUnits) break; // (wave-uniform)
        unit = next;
#pragma unroll
        for (int i = 0; i < 14; i++) e[i] = en[i];
      }
    }
  }

  void launch_stream_pack(const StreamParams &p, hipStream_t s)
  {
    if (p.nWaves)
    { // strip form (images of whole blocks): the encode kernel has left the strips' payload words
      hipLaunchKernelGGL(k_stream_scan_strips, dim3(1), dim3(1024), 0, s, p);
      hipLaunchKernelGGL(k_stream_pack_strips, dim3(p.nWaves), dim3(64), 0, s, p);
      return;
    }
    hipLaunchKernelGGL(k_stream_count, dim3(p.nTiles), dim3(kTile), 0, s, p);
    hipLaunchKernelGGL(k_stream_scan, dim3(1), dim3(1024), 0, s, p);
    hipLaunchKernelGGL(k_stream_pack, dim3(p.nTiles), dim3(kTile), 0, s, p);
  }

  void launch_stream_decode(const DecodeParams &p, hipStream_t s)
  {
    // persistent: one workgroup of four waves per residency slot (4 per CU), every wave strides over the units of 64 blocks
    static int slots = 0;
    if (!slots)
    {
      int dev = 0;
      hipDeviceProp_t prop;
      slots = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount * 4 : 1024;
    }
    const uint32_t units = (p.nBlocks + 63u) / 64u, need = (units + 3u) / 4u;
    hipLaunchKernelGGL(k_stream_decode, dim3(need < (uint32_t)slots ? need : (uint32_t)slots), dim3(kTile), 0, s, p);
  }
}

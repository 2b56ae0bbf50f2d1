// Internal declarations shared by the host API (limg_hip_api.hip) and the kernel files.  Not part of the C ABI.
#ifndef LIMG_HIP_INTERNAL_H
#define LIMG_HIP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "../../include/limg_hip.h"

namespace limg_hip
{
  constexpr int kBlock = 8;        // limg_MinBlockSize (reference: src/limg_internal.h:158)
  constexpr int kStripBlocks = 32; // image blocks per workgroup ("work strip": 256 x 8 px)
  constexpr uint64_t kDitherSeed = 0xCA7F00D15BADF00DULL; // reference: src/limg.cpp:1893

  // The caller's pointers of one image: input pixels and the 11 output planes.  A batched encode (limg_hip_encode3d_batch_device) holds one of these per image
  // in a device table; the kernels read them with scalar loads where they use them.
  struct ImageIO
  {
    const uint32_t *in;
    limg_hip_encode3d_info info;
  };

  // Everything one encode needs on the device.  Passed by value to the kernels.
  struct EncodeParams
  {
    ImageIO io;              // image 0 (the only one of a single-image encode)
    // batched encode: `batchCount` images of the same shape in ONE launch pair; image i = batch[i]; work strip ids, look-back descriptors and the per-block
    // scratch (records, shift words) run image after image (image i owns strips [i * imageStrips, (i + 1) * imageStrips) and block rows [i * blocksY, ...));
    // every image starts its own dither chain(s) (the reference's batch loop calls the encoder once per image: src/main.cpp:278-323)
    const ImageIO *batch;
    uint32_t batchCount, imageStrips;
    uint32_t sizeX, sizeY, blocksX, blocksY, stripsX;
    uint32_t maxPixel32;     // min(maxPixelBitCrushError, 2^32 - 1)
    uint64_t maxBlock;       // maxBlockBitCrushError
    uint32_t blockLimitFull; // min(ceil(maxBlock * 64 / 16), 2^32 - 1): what a full block's error sum is compared with (be * 16 < maxBlock * n)
    int32_t crushBits, fast; // reference: src/limg.cpp:2192-2197
    const uint32_t *accTable; // !fast: the accurate search's automaton, 8 dwords per state (limg_search_table_accurate.h expanded by the context)
    int32_t forced[3];       // -1 or forced shift
    int32_t recordLimit;     // |record value| above which a block takes the generic 32-bit trial (2700: the bound the packed trial's 16-bit terms are proven for; the test build lowers it to exercise the generic path)
    // chain partition (reference: src/limg.cpp:2114-2134), in block rows
    uint32_t chainCount, chainRows; // chain c (< chainCount-1) owns block rows [c*chainRows, (c+1)*chainRows); the last owns the rest
    // per-block scratch / compact outputs
    limg_hip_block_record *records;
    float *invN;           // per block 4 floats: 1 / |normal|^2 of factors A, B, C (src/limg_internal.h:426-452), 0; written by k_fit_tpb next to the record, read by the E step
    uint32_t *shifts;      // per block: sA | sB<<8 | sC<<16 | calls<<24
    uint32_t *stripCalls;  // per work strip: number of dither calls (blocksY * stripsX, raster order)
    uint32_t *stripBase;   // per work strip: index of its first dither call inside its chain (exclusive scan)
    uint32_t *stripWords;  // stream mode (or NULL): per work strip the payload words of its blocks in the compact stream -- what the packer's scan starts from
    // (split path: the pre-dither factor bytes live in the caller's factor planes between the two kernels)
    int32_t storePlanes;   // 0: _perf behaviour
    int32_t fullPlanes;    // 0: compact mode -- only the three factor planes (+ records / shift words) are written
    // dither noise: byte p of call k = low byte of the 16-bit lane the reference ANDs with ditherSize for pixel p
    const uint8_t *noise;
    uint32_t noiseLast;    // index of the table's last entry: call indices are clamped to it (a chain base handed in by the caller, limg_hip_encode3d_chain_device phase 2, cannot make the F step read past the table)
    // fused single-kernel path: per-strip look-back descriptors (status << 32 | value), work ticket, error word
    unsigned long long *desc;
    uint32_t *ticket;   // [0] = next strip id
    int32_t zeroLookback; // k_fit_tpb clears `ticket` (16 bytes) and the descriptors of the strips its waves cover (the persistent launch follows it on the stream)
    uint32_t *timeout;  // sticky: set when a look-back spin gave up; lives outside the per-launch words, cleared only by limg_hip_check_device_status
#ifdef LIMG_HIP_TEST_HOOKS // liblimg_hip_test.so only (include/limg_hip_test_hooks.h): the product's kernels carry none of this
    uint32_t lookbackSpins; // bound of one look-back wait, in polls (the product: kLookbackSpins)
    uint32_t testBaseErrStrip; // test hook: the strip with this id dithers from a chain position that is off by one call (~0: none); its published count stays right
    uint32_t testSkipStrip; // test hook: the strip with this id never publishes its dither-call count (~0: none) -- what a lost predecessor looks like to the look-back
#endif
    uint8_t *park;      // persistent kernel: per workgroup two 8 KiB slots holding a strip's parked results between its E and F steps
    int32_t compactOut; // persistent kernel: also write records / shift words to the raster-order arrays
    int32_t streamRaw;  // compact mode only: factors with shift 8 store their raw byte instead of 0 (input of the stream packer)
    int32_t fitOnly;    // split path: stop after the records (pass 1 of the merged-block encoder, src/limg.cpp:1088-1119)
    // cross-GPU single dither chain (split path only): the scan writes this image strip's dither-call total; the F step adds the strip's first call index
    unsigned long long *chainCallsOut;
    const unsigned long long *chainBase;
    int32_t fitPrio;    // k_fit_tpb: wave priority (s_setprio 0..3); raised when it runs next to a persistent kernel (batched encode as a pipeline)
    int32_t prefit;     // host dispatch only: p.records already hold the fit (k_fit_tpb ran first)
    int32_t floatFast;  // host dispatch only: FAST float stage (limg_hip_options.float_mode = 1)
    int32_t vecIn;      // rows of pIn may be read 16 bytes per lane (sizeX % 4 == 0 and pIn 16-byte aligned); otherwise dword loads
    int32_t vecPlanes;  // the seven block-uniform uint32 planes may be stored 16 bytes per lane (sizeX % 4 == 0 and all seven 16-byte aligned)
    int32_t vecFactors; // the three factor planes may be accessed 16 bytes per lane (sizeX % 16 == 0 and all three 16-byte aligned)
    int32_t vecFactors8; // ... 8 bytes per lane (sizeX % 8 == 0 and all three 8-byte aligned)
    int32_t vecDecoded; // pDecoded may be stored 16 bytes per lane (sizeX % 4 == 0 and 16-byte aligned)
  };

  // stream pack (limg_hip_stream.hip): from the compact outputs of an encode (factor planes, records, shift words)
  struct StreamParams
  {
    uint32_t sizeX, sizeY, blocksX, blocksY, nBlocks, nTiles, channels, errorFactor, flags;
    const uint8_t *fac[3];
    const limg_hip_block_record *records;
    const uint32_t *shifts;
    uint8_t *stream;
    uint32_t *tileBase; // three-kernel form (widths that are not whole blocks): per tile of 256 blocks payload words, then (after the scan) the tile's first payload word
    // strip form (widths in whole blocks): the encode kernel leaves every work strip's payload words (EncodeParams::stripWords); k_stream_scan_strips turns them into
    // the strips' first payload words in place and writes the header; k_stream_pack_strips packs one strip (<= 32 consecutive blocks of one block row) per wave step
    uint32_t *stripWords;
    uint32_t stripsX, nStrips, nWaves;
  };

  struct DecodeParams
  {
    uint32_t sizeX, sizeY, blocksX, blocksY, nBlocks;
    const uint8_t *stream;
    unsigned long long streamBytes;
    uint32_t *out;
    uint32_t *status; // bit 0: header mismatch, bit 1: inconsistent payload offsets
    uint32_t *sink;   // 2 KiB nobody reads: where lanes whose block row is not (wholly) inside the image send their two 16-byte stores, so that EVERY lane of EVERY group issues
                      // exactly two stores and the wait for the next payload run can be counted (s_waitcnt vmcnt(2)) instead of draining the stores (vmcnt(0)); see k_stream_decode
  };

  const uint64_t *noise_checkpoints_host(size_t *pCount, size_t *pEvery);
  const uint64_t *noise_checkpoints_far_host(size_t *pCount, size_t *pEvery);
  void launch_noise_fill(uint8_t *noise, const uint64_t *dCheckpoints, size_t count, hipStream_t s);
  void launch_noise_expand(uint8_t *noise, const unsigned long long *dStates, const uint8_t *dPixels, size_t calls, bool pcg, hipStream_t s);
  void launch_noise_expand_calls(uint8_t *noise, const unsigned long long *dStates, const unsigned long long *dOffsets, const uint32_t *dPixels, size_t calls, bool pcg, hipStream_t s);
  void launch_set_batch_table(ImageIO *dTable, const ImageIO *hTable, size_t count, hipStream_t s);
  void launch_fit_tpb(const EncodeParams &p, int channels, hipStream_t s);
  void launch_fit_search(const EncodeParams &p, int channels, hipStream_t s);
  void launch_encode_persistent(const EncodeParams &p, int channels, int workgroups, hipStream_t s);
  void launch_strip_scan(const EncodeParams &p, hipStream_t s);
  void launch_dither_store(const EncodeParams &p, int channels, hipStream_t s);
  void launch_shift_stats(const uint32_t *dShifts, uint32_t blocksX, uint32_t blocksY, uint32_t rows, uint32_t sizeX, uint32_t sizeY, unsigned long long *dOut30, hipStream_t s);
  void launch_chain_base(const unsigned long long *dCalls, int rank, int world, unsigned long long *dBase, uint32_t *dAborted, hipStream_t s);

  // ---- merged-block encoder (limg_hip_blocked.hip; reference: limg_blocked_encode3d_test, src/limg.cpp:1774-1885, :2329-2453) ----
  // similarity bits are precomputed for candidate offsets dx, dy in [-kMatchLo, +kMatchHi] blocks around every seed: rectangles grow right / down from
  // their seed (far), and up / left only in the second attempt from the centre third (near); measured on the synthetic workloads, this window answers
  // 99.7 % of the merge's queries (the rest is evaluated on the host)
  constexpr int kMatchLo = 5, kMatchHi = 12;
  constexpr int kMatchSide = kMatchLo + kMatchHi + 1;   // 18
  constexpr int kMatchCells = kMatchSide * kMatchSide;  // 324
  constexpr int kMatchWords = (kMatchCells + 63) / 64;  // 6 x 64 bits per seed

  struct RegionDesc // one rectangle of 8x8 blocks, in creation (= block index = dither chain) order
  {
    uint32_t ox, oy, rx, ry; // blocks
    uint32_t keep;           // 1: single block that keeps its pass-1 fit (src/limg.cpp:1863-1881)
    uint32_t scratch;        // first element of this region in the scratch arrays (multiple of 4)
    uint32_t pad[2];
  };
  struct RegionOut
  {
    limg_hip_block_record rec;
    uint32_t shiftWord; // sA | sB << 8 | sC << 16 | ditherCalls << 24
    uint32_t pad[3];
  };
  struct BlockedParams
  {
    const uint32_t *in;
    uint32_t sizeX, sizeY, blocksX, blocksY, channels;
    uint32_t maxPixel32;
    uint64_t maxBlock;
    int32_t crushBits, fast, forced[3];
    const limg_hip_block_record *pass1;
    unsigned long long *matchBits; // [blocks][kMatchWords]
    uint32_t seedBase, seedCount;  // k_blocked_match: the seeds of this launch (a band of block rows)
    uint8_t *matchFlags;           // per seed: bit 0 = its 3x3 neighbourhood (right / down) matches entirely, bit 1 = its right or its lower neighbour matches
    float *matchBound;             // per block, 4 floats (k_blocked_bounds): the coefficients of an upper bound of the predicate's 27-colour average; nullptr = not used
    const RegionDesc *regions; // of this launch (a batch of consecutive rectangles)
    uint32_t nRegions;
    uint32_t regionBase;       // index of regions[0] in creation order (block index = regionBase + r + 1)
    RegionOut *out;
    int32_t vecStore;          // k_blocked_store: 4 pixels per lane, 16-byte plane stores (image of whole blocks, sizeX % 4 == 0, every plane suitably aligned)
    uint32_t *order;           // per launch (or NULL): the rectangle each workgroup takes -- large ones first (k_blocked_order)
    uint32_t *scratchPx; // gathered pixels, region-major (src/limg.cpp:1747-1748)
    uint8_t *scratchFac; // 3 planes of scratchCap bytes: pre-dither factor bytes
    uint32_t scratchCap;
    const uint8_t *noise;              // one byte per pixel per dither call, region after region in chain order
    const unsigned long long *noiseBase; // per region: offset of its first call's bytes
    limg_hip_blocked_encode3d_info info;
  };

  void launch_blocked_bounds(const BlockedParams &p, hipStream_t s);
  void launch_blocked_match(const BlockedParams &p, hipStream_t s);
  void launch_blocked_fit_search(const BlockedParams &p, hipStream_t s);
  void launch_blocked_store(const BlockedParams &p, hipStream_t s);
  void launch_blocked_order(const BlockedParams &p, hipStream_t s);

  // host side of the merged-block encoder (limg_hip_blocked_host.cpp): the greedy raster merge and the dither chain walk
  struct HostRegion { uint32_t ox, oy, rx, ry, keep; };
  void blocked_merge(const limg_hip_block_record *pass1, const unsigned long long *matchBits, uint32_t blocksX, uint32_t blocksY, int channels, std::vector<HostRegion> &out,
                     const std::function<void(size_t)> *progress = nullptr, const std::function<void(uint32_t)> *needSeedRow = nullptr, const uint8_t *seedFlags = nullptr,
                     const std::function<void()> *needRecords = nullptr);
  bool blocked_matches_host(int channels, const limg_hip_block_record &seed, const limg_hip_block_record &cand);
  uint64_t chain_call_n(uint64_t h, size_t n, uint8_t *noise, bool pcg);
  uint64_t chain_walk_batch(uint64_t h, size_t count, const uint8_t *shiftWords, size_t stride, const uint32_t *npx, unsigned long long *noiseBase, unsigned long long *callState,
                            unsigned long long *callOff, uint32_t *callPx, uint64_t &noiseOff, size_t &callCount, size_t maxCalls, bool pcg); // limg_hip_noise.cpp

  void launch_stream_pack(const StreamParams &p, hipStream_t s);
  void launch_stream_decode(const DecodeParams &p, hipStream_t s);

  void launch_synth_random_gradient(uint32_t *out, uint32_t w, uint32_t h, uint64_t seed, int opaque, uint32_t y0, hipStream_t s);
  void launch_synth_photo_noise(uint32_t *out, uint32_t w, uint32_t h, uint64_t seed, uint32_t y0, hipStream_t s);
  void launch_compare(const uint32_t *a, const uint32_t *b, uint64_t count, int channels, unsigned long long *dErrorSum, hipStream_t s);
}

#endif

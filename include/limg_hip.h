/* limg_hip.h -- C ABI of liblimg_hip.so: the MI355X (gfx950) implementation of limg's encode hot path.
 *
 * Drop-in boundary for the reference's `limg_encode3d_test` / `limg_encode3d_test_perf` / `limg_compare` and (further down)
 * `limg_blocked_encode3d_test` (reference: src/limg.h:27-48; the reference has no FFI layer, its API is plain C++ functions), plus
 * the build-defined compact stream pair the task names `limg_encode` / `limg_decode`.  Every entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no C++ or torch types.  A header-only C++ shim with the *exact* reference
 * signatures is in include/limg_hip_shim.hpp.  There is no CPU fallback: without a HIP device limg_hip_init fails.
 *
 * Data layout (identical to the reference, src/limg.h:29-33 and SURVEY.md 8(b)):
 *   pIn            row-major uint32 RGBA8, byte 0 = R, sizeX*sizeY elements, row stride sizeX
 *   8 uint32 planes (pDecoded, pShiftABCX, pColAMin, pColAMax, pColBMin, pColBMax, pColCMin, pColCMax)
 *   3 uint8  planes (pFactorsA, pFactorsB, pFactorsC), each sizeX*sizeY elements, row stride sizeX
 * Ownership: the caller allocates and frees every image/plane buffer; the library never retains caller pointers.
 * The context owns device staging buffers, per-block scratch and the dither noise table.
 */
#ifndef LIMG_HIP_H
#define LIMG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numeric values as `enum limg_result` (src/limg.h:9-18) */
typedef enum limg_hip_result
{
  limg_hip_success = 0,
  limg_hip_error_Generic = 100, /* also: any HIP runtime failure */
  limg_hip_error_InvalidParameter,
  limg_hip_error_ArgumentNull,
  limg_hip_error_OutOfBounds,
  limg_hip_error_MemoryAllocationFailure
} limg_hip_result;

/* same member order and meaning as `struct limg_encode3d_info` (src/limg.h:29-33) */
typedef struct limg_hip_encode3d_info
{
  uint32_t *pDecoded, *pShiftABCX, *pColAMin, *pColAMax, *pColBMin, *pColBMax, *pColCMin, *pColCMax;
  uint8_t *pFactorsA, *pFactorsB, *pFactorsC;
} limg_hip_encode3d_info;

/* `limg_encode_3d_output<4>` (src/limg_internal.h:343-353): the per-block decomposition record, 64 bytes.
 * For 3-channel input lane 3 of every array is 0 (the reference's 48-byte <3> layout is this minus lane 3). */
typedef struct limg_hip_block_record
{
  float avg[4];
  int16_t dirA_min[4], dirA_max[4], dirB_offset[4], dirB_mag[4], dirC_offset[4], dirC_mag[4];
} limg_hip_block_record;

/* Optional per-block ("compact") outputs of the device entry point; any pointer may be NULL. Device pointers. */
typedef struct limg_hip_compact_out
{
  limg_hip_block_record *pRecords; /* blocksX*blocksY, raster order */
  uint32_t *pShifts;               /* blocksX*blocksY: shiftA | shiftB << 8 | shiftC << 16 | ditherCalls << 24 */
} limg_hip_compact_out;

/* Knobs that have no parameter in the reference signature (the reference has none of these: src/limg.h:27-48).  Fill with limg_hip_default_options, then set what
 * you need.  The struct is VERSIONED BY ITS SIZE: `struct_size` is sizeof(limg_hip_options) as the caller's compiler saw it (limg_hip_default_options below writes
 * it); the library reads and writes only that many bytes and gives every member beyond them its default, so members can be appended without breaking a caller
 * built against an earlier header.  Nothing here changes results except forced_shift, dither_pcg and float_mode; there are no fault-injection or test knobs in this
 * library (those live in liblimg_hip_test.so, include/limg_hip_test_hooks.h, which the test suite loads instead). */
typedef struct limg_hip_options
{
  uint32_t struct_size;    /* sizeof(limg_hip_options) in the caller's build; limg_hip_set_options rejects 0, anything not a multiple of 4 and anything too short for forced_shift */
  int32_t forced_shift[3]; /* all three in 0..8: bypass the shift search (a10-a12) with this triple; otherwise {-1,-1,-1} */
  int32_t force_split_kernels; /* non-0: use the three-launch path (fit+search, scan, dither+store) even where the persistent kernel applies */
  int32_t dither_pcg;          /* non-0: the reference's PCG dither (src/limg.cpp:799-822, what it runs on hosts without AES-NI) instead of the AES one */
  int32_t float_mode;          /* 0 (default) = EXACT: the float stage op for op as the reference's strict SSE build (DPPS order, x86 RSQRTPS table, correctly
                                  rounded divisions): every plane bit-identical to the reference.  1 = FAST: hardware v_rsq_f32 / v_rcp_f32 and fused multiply-adds;
                                  contract: the integer stage stays bit-exact given the same records, extrema within +-2 LSB on >= 99.9 % of blocks, perceptual PSNR
                                  within 0.10 dB of EXACT; the 8x8 path only (the merged-block encoder always runs EXACT) */
  int32_t legacy_float_stage;  /* non-0: run the float stage inside the E step with lane == pixel (round-1 mapping; what images with partial edge blocks always use)
                                  instead of the one-lane-per-block kernel k_fit_tpb.  Same bits either way; A/B switch for tests and the bench */
  int32_t collect_stats;       /* non-0: every encode (8x8 path and merged-block encoder) also leaves the reference's bit statistics for limg_hip_last_stats */
  int32_t host_noise_table;    /* non-0: build the context's dither noise table on the host (one serial AES walk + upload, ~100 ms per new size class: what rounds 1-2 did)
                                  instead of filling it on the GPU from the embedded chain checkpoints.  Same bytes; A/B switch for tests */
  int32_t batch_sub_images;    /* limg_hip_encode3d_batch_device runs a long list as a PIPELINE of sub-batches: the float-stage kernel of sub-batch k + 1 on a stream of
                                  the context's own next to the persistent kernel of sub-batch k (which leaves it a residency slot: 5 workgroups per CU instead of 6).
                                  0 = automatic (sub-batches of 8 for lists of 32 images and more, of 4 from 16 images, none below); > 0 = sub-batches of this many
                                  images; < 0 = never: one launch pair for the whole list.  Same planes in every case */
  int32_t ragged_bands;        /* images with a partial last block COLUMN and one dither chain (poolThreads == 0): the host's chain walk -- the floor of this class -- is
                                  pipelined with the GPU in this many bands of block rows (E step band by band, the walk under it, every band's F step under the next band's
                                  walk).  0 = automatic (16 bands from 64 block rows x 32 block columns on), N > 0 = N bands, < 0 = off (one E launch, walk, one F launch) */
  int32_t ragged_walk_threads; /* ... with several chains (poolThreads > 0) the chains are independent, as on the reference's thread pool (src/limg.cpp:2114-2134): they are
                                  walked on this many host threads (0 = one per chain, at most 16 and the host's hardware threads; 1 = serial) */
} limg_hip_options;

typedef struct limg_hip_context limg_hip_context;

/* Thread safety.  The reference is re-entrant (stack scratch only, src/limg.cpp:1890-1891).  Here a context owns device scratch, so:
 *   - the blocking HOST-pointer entries (limg_hip_encode3d, _encode3d_perf, _compare, _blocked_encode3d, _encode_stream, _decode_stream) take a mutex inside the
 *     context: any number of threads may call them on one shared context at once (this is what the C++ shim's limg_encode3d_test & co. rely on); the calls run
 *     one after the other;
 *   - the asynchronous *_device entries enqueue kernels that use the context's scratch after the call has returned: use one context per HIP stream / thread
 *     for those (contexts are independent and cheap next to an image: tests/test_gpu_parity.py::test_two_contexts_on_two_threads);
 *   - limg_hip_init / limg_hip_shutdown / limg_hip_set_options of one context must not race with calls on that context. */

/* Create / destroy a context bound to HIP device `device` (-1 = current device).  No reference analogue
 * (the reference keeps no state besides CPUID flags, src/limg_simd.cpp:57-60). */
limg_hip_result limg_hip_init(int device, limg_hip_context **ppCtx);
void limg_hip_shutdown(limg_hip_context **ppCtx);
/* Defaults into the first `structSize` bytes of *pOptions (struct_size = structSize).  Call it through limg_hip_default_options, which passes the size of the struct
 * as the CALLER was compiled. */
void limg_hip_default_options_sized(limg_hip_options *pOptions, size_t structSize);
static inline void limg_hip_default_options(limg_hip_options *pOptions) { limg_hip_default_options_sized(pOptions, sizeof(limg_hip_options)); }
/* Takes pOptions->struct_size bytes; members beyond them get their defaults. */
limg_hip_result limg_hip_set_options(limg_hip_context *pCtx, const limg_hip_options *pOptions);
/* pOptions->struct_size must be set on entry (= the room the caller has); that many bytes are written.  Read, change one member, set: nothing else is reset. */
limg_hip_result limg_hip_get_options(const limg_hip_context *pCtx, limg_hip_options *pOptions);

/* Replaces `limg_encode3d_test` (src/limg.h:35, src/limg.cpp:2175-2265).  HOST pointers; blocking.
 * poolThreads: 0 == `pThreadPool = nullptr` (one dither chain over the image); T > 0 == a pool of T threads, i.e. T*4
 * row strips each restarting the dither chain (src/limg.cpp:2114-2134).  bool parameters are ints (0 / non-0). */
limg_hip_result limg_hip_encode3d(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, limg_hip_encode3d_info *pInfo,
                                  uint32_t errorFactor, int poolThreads, int fastBitCrushing);

/* Replaces `limg_encode3d_test_perf` (src/limg.h:37, src/limg.cpp:2267-2327): same work, nothing stored. HOST pointer. */
limg_hip_result limg_hip_encode3d_perf(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, uint32_t errorFactor,
                                       int poolThreads, int fastBitCrushing);

/* Device-resident variant of `limg_encode3d_test`: every pointer (pIn, the 11 planes inside *pInfo, pCompact members)
 * is a DEVICE pointer; pInfo / pCompact themselves are host structs.  Asynchronous on `stream` (a hipStream_t passed
 * as void*; NULL = default stream).  pInfo == NULL gives the `_perf` behaviour (fit + search only).
 * Compact mode (SURVEY.md 8(d), 8.05 B/px): pInfo with the eight uint32 plane pointers all NULL and the three factor planes set
 * writes only the crushed factor bytes; together with pCompact (records + shift words) that is everything a decoder needs.
 * Alignment: any (4-byte aligned pIn / uint32 planes) is accepted.  The kernels use 16-byte accesses on pIn when it is 16-byte aligned and
 * sizeX % 4 == 0, and on the factor planes when all three are 16-byte aligned and sizeX % 16 == 0; other pointers (sliced, offset) take
 * dword / byte paths with identical results (hipMalloc and torch allocations are 256-byte aligned). */
limg_hip_result limg_hip_encode3d_device(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha,
                                         const limg_hip_encode3d_info *pInfo, const limg_hip_compact_out *pCompact, uint32_t errorFactor, int poolThreads,
                                         int fastBitCrushing, void *stream);

/* A list of images of ONE shape, each with its own 11 planes: replaces the reference's per-file loop `for (file) limg_encode3d_test(...)` (src/main.cpp:278-323, what
 * its `--count` benchmark and BASELINE configs 2 / 4 run).  ppIn[i] / pInfos[i] are the DEVICE pointers of image i (the arrays themselves are host memory and may be
 * freed when the call returns).  Every image gets the planes a single limg_hip_encode3d_device call would give it (own dither chain(s), same poolThreads rule), but
 * the whole list goes through one launch pair -- one float-stage grid over the blocks of all images, one persistent launch over the strips of all images -- so a
 * small image's ramp-up and drain are paid once per list, not once per image.  Images with partial edge blocks, compact mode (planes NULL) and the A/B options
 * fall back to one encode per image, same results.  Asynchronous on `stream`. */
limg_hip_result limg_hip_encode3d_batch_device(limg_hip_context *pCtx, size_t count, const uint32_t *const *ppIn, size_t sizeX, size_t sizeY, int hasAlpha,
                                               const limg_hip_encode3d_info *pInfos, uint32_t errorFactor, int poolThreads, int fastBitCrushing, void *stream);

/* Replaces `limg_compare` (src/limg.h:48, src/limg.cpp:2455-2491): perceptual PSNR; HOST pointers. */
double limg_hip_compare(limg_hip_context *pCtx, const uint32_t *pImageA, const uint32_t *pImageB, size_t sizeX, size_t sizeY, int hasAlpha,
                        double *pMeanSquaredError, double *pMaxPossibleSquaredError);
/* same on DEVICE pointers (blocking on `stream`) */
double limg_hip_compare_device(limg_hip_context *pCtx, const uint32_t *pImageA, const uint32_t *pImageB, size_t sizeX, size_t sizeY, int hasAlpha,
                               double *pMeanSquaredError, double *pMaxPossibleSquaredError, void *stream);

/* Synthetic inputs of SURVEY.md 8(d), generated directly in HBM (DEVICE pointer, async on stream).
 * (y0, fullWidth) let a rank generate only its row strip of a larger image. */
limg_hip_result limg_hip_synth_random_gradient_device(uint32_t *pOut, size_t width, size_t height, uint64_t seed, int opaque, size_t y0, void *stream);
limg_hip_result limg_hip_synth_photo_noise_device(uint32_t *pOut, size_t width, size_t height, uint64_t seed, size_t y0, void *stream);

/* Waits for the device and returns limg_hip_error_Generic if the persistent kernel's bounded look-back wait ever timed out (protocol safety net; the host-pointer
 * entry points call it themselves and return the error).  Such a timeout is never silent: the strip that gave up and every later strip of its dither chain store
 * NOTHING that depends on the chain position -- their pDecoded / pFactorsA/B/C pixels keep what the caller's buffers held, the block-uniform planes may be partly
 * written -- so a caller of the asynchronous *_device entries that skips this check cannot end up with wrongly dithered planes that look complete.
 * Progress does not depend on what else runs on the GPU: every work strip id is drawn from an atomic ticket by a workgroup that is already resident, so any number
 * of contexts may have persistent kernels in flight on their own streams at once (tests/test_gpu_concurrency.py). */
limg_hip_result limg_hip_check_device_status(limg_hip_context *pCtx);

/* The statistics the reference's limg_encode3d_test / limg_blocked_encode3d_test print themselves (src/limg.cpp:2232-2248; counters :1971-1999, :1561-1590):
 * pCounters30[f] (f = 0..2: factor A, B, C) = bits kept, summed over the pixels ((8 - shift) per pixel); pCounters30[3 + 9 f + s] = pixels whose block crushed factor
 * f by s bits.  "Average Block Bits" = (c[0] + c[1] + c[2]) / pixels; the histogram line of factor f = c[3 + 9 f + s] * 100 / pixels for s = 0..8 ("8 bit" .. "0 bit").
 * Of the context's last encode made with limg_hip_options.collect_stats set (a batched encode: all its images together); waits for that encode.
 * The library itself prints nothing (SURVEY 8(b) "Side effects"); include/limg_hip_shim.hpp's limg_print_stats and tools/limg_hip_cli.cpp print upstream's lines. */
limg_hip_result limg_hip_last_stats(limg_hip_context *pCtx, uint64_t *pCounters30, uint64_t *pPixels);
/* limg_hip_encode3d / limg_hip_blocked_encode3d that ALSO return the counters of that very encode (upstream keeps them on the stack of the call, src/limg.cpp:1975-1976,
 * so its printout is per call whatever other threads do): encode and fetch happen under the context's mutex, collect_stats is on for the call only.  What the C++
 * shim's limg_encode3d_test / limg_blocked_encode3d_test call when limg_hip_shim::print_stats(true) is set -- safe from any number of threads on one context. */
limg_hip_result limg_hip_encode3d_stats(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, limg_hip_encode3d_info *pInfo,
                                        uint32_t errorFactor, int poolThreads, int fastBitCrushing, uint64_t *pCounters30, uint64_t *pPixels);

/* Per-kernel timing for the bench (HIP events recorded on the stream each profiled encode is launched on).
 * limg_hip_profile_end writes 3 floats per profiled encode: k_fit_search, k_strip_scan (or the host chain walk of ragged
 * images), k_dither_store, in milliseconds; returns the number of encodes written or -1. */
limg_hip_result limg_hip_profile_begin(limg_hip_context *pCtx);
int limg_hip_profile_end(limg_hip_context *pCtx, float *pMs, int maxEncodes);

/* Host-only helpers (no GPU touched): the data-independent dither noise stream and the strip partition rule.
 *  limg_hip_host_noise_table : `calls` x 64 noise bytes of a chain of full 8x8 blocks starting at the reference's seed
 *                              (src/limg.cpp:1893); byte p of call k is what the AES dither ANDs with ditherSize for pixel p.
 *  limg_hip_host_chain_call  : one dither call's state walk over `pixelCount` pixels (64 for an 8x8 block, any count for a rectangle of the
 *                              merged-block encoder): returns the next chain value (src/limg.cpp:824-879), optionally writing the noise
 *                              bytes (buffer of max(64, pixelCount) bytes).
 *                              forceSoftwareAes bit 0: do not use AES-NI; bit 1: PCG dither instead of AES.
 *  limg_hip_host_partition   : src/limg.cpp:2114-2134 in block rows: chain c < count-1 owns rows [c*rows, (c+1)*rows), the last the rest. */
limg_hip_result limg_hip_host_noise_table(uint8_t *pOut, size_t calls);
/*  limg_hip_noise_table_device: the same stream written by the GPU (what a context does for itself on the first encode of a size class): `calls` x 64 bytes into a
 *                              DEVICE buffer (16-byte aligned), asynchronous on `stream`; at most 2^27 calls (the reach of the embedded checkpoints: dense ones -- every
 *                              1024 calls -- through 16 Mi calls, far ones -- every 65536 -- beyond, which the context makes dense on host threads when an image of
 *                              more than 5.59 M blocks needs them). */
limg_hip_result limg_hip_noise_table_device(limg_hip_context *pCtx, uint8_t *pOutDevice, size_t calls, void *stream);
uint64_t limg_hip_host_chain_call(uint64_t chainValue, size_t pixelCount, uint8_t *pNoise64, int forceSoftwareAes);
limg_hip_result limg_hip_host_partition(size_t sizeY, int poolThreads, uint32_t *pChainCount, uint32_t *pChainBlockRows);
/*  limg_hip_host_chain_checkpoints: walks `calls` full-block dither calls from the reference's seed (src/limg.cpp:1893) and writes the chain value that call
 *                              i * every starts from to pOut[i] (may be NULL); returns the value after the last call.  The library's embedded checkpoint
 *                              table (one value every 1024 calls, from which the GPU fills the noise table in parallel) is generated and checked with it. */
uint64_t limg_hip_host_chain_checkpoints(size_t calls, size_t every, uint64_t *pOut, int pcg);
/*  limg_hip_host_dense_checkpoints: the chain values that calls number (first + k) * 1024, k < count, start from -- what the GPU's noise-table fill is given -- from the
 *                              embedded tables alone: the dense one where it reaches, the far one + 65536 calls on foot per far value (host threads) beyond;
 *                              limg_hip_error_OutOfBounds beyond 2^27 calls.  Equal to limg_hip_host_chain_checkpoints' serial walk (tests/test_host.py). */
limg_hip_result limg_hip_host_dense_checkpoints(size_t first, size_t count, uint64_t *pOut);

/* ---- merged-block encoder ----------------------------------------------------------------------------------------------------------
 * Replaces `limg_blocked_encode3d_test` (src/limg.h:46, src/limg.cpp:2329-2453), what the reference's CLI runs on a single file
 * (src/main.cpp:255): per-8x8 fit, greedy merge of similar neighbouring blocks into rectangles, re-fit + bit crush + dither + decode
 * per rectangle.  Struct = `limg_blocked_encode3d_info` (src/limg.h:39-44), same member order.  pBlockError is never written (it is not
 * upstream either) and may be NULL.  The pool argument of the reference only splits its first pass and has no effect on the result, so it
 * has no counterpart here.  Division of labour: fits, the block-similarity predicate, the per-rectangle work and all plane
 * stores run on the GPU; the greedy raster scan over precomputed similarity bits and the (inherently serial) dither chain walk run on the host. */
typedef struct limg_hip_blocked_encode3d_info
{
  uint32_t *pDecoded;
  uint8_t *pFactorsA, *pFactorsB, *pFactorsC, *pBlockError, *pBitsPerPixel;
  uint32_t *pShiftABCX, *pColAMin, *pColAMax, *pColBMin, *pColBMax, *pColCMin, *pColCMax, *pBlockIndex;
} limg_hip_blocked_encode3d_info;

typedef struct limg_hip_region { uint32_t ox, oy, rx, ry; } limg_hip_region; /* in 8x8 blocks, creation (= block index) order */

/* HOST pointers, blocking. */
limg_hip_result limg_hip_blocked_encode3d(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, limg_hip_blocked_encode3d_info *pInfo,
                                          uint32_t errorFactor, int fastBitCrushing);
limg_hip_result limg_hip_blocked_encode3d_stats(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, limg_hip_blocked_encode3d_info *pInfo,
                                                uint32_t errorFactor, int fastBitCrushing, uint64_t *pCounters30, uint64_t *pPixels);
/* DEVICE pointers (pIn and the planes inside *pInfo); returns when everything has been enqueued on `stream` -- the call itself waits for
 * the intermediate device results its host stages need. */
limg_hip_result limg_hip_blocked_encode3d_device(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha,
                                                 const limg_hip_blocked_encode3d_info *pInfo, uint32_t errorFactor, int fastBitCrushing, void *stream);
/* The rectangles of the context's last merged-block encode (copies up to `capacity`, always reports the count). */
limg_hip_result limg_hip_blocked_regions(limg_hip_context *pCtx, limg_hip_region *pRegions, size_t capacity, size_t *pCount);
/* Milliseconds of the last merged-block encode: [0] pass 1 + similarity bits (GPU, incl. their copy to the host), [1] greedy merge (host, this thread), and --
 * overlapped with [1] on a worker thread, summed over its batches -- [2] per-rectangle fit + search (GPU, incl. copies), [3] dither chain walk (host),
 * [4] noise upload + dither/decode/store launch; [5] wall-clock total. */
limg_hip_result limg_hip_blocked_timing(limg_hip_context *pCtx, double *pMs6);
/* The similarity bits of the context's last merged-block encode (limg_hip_host_blocked_match_words() words per block, the layout of limg_hip_host_blocked_match_bits):
 * copies up to `capacityWords`, always reports the count.  For tests: the GPU kernel's bits against the host evaluation of the same records. */
limg_hip_result limg_hip_blocked_match_bits(limg_hip_context *pCtx, uint64_t *pBits, size_t capacityWords, size_t *pWords);
/* GPU time of the last merged-block encode's launches, from HIP events on the streams they run on: [0] pass 1 (the 8x8 path's float stage), [1] the similarity
 * kernels (16 bands, back to back), and summed over the worker's batches [2] the per-rectangle fit + search kernel, [3] chain-value upload + noise expansion +
 * dither/decode/store kernel.  (What the GPU side costs with no host in the way: bench.py's kernel-only rate of the merged-block encoder.) */
limg_hip_result limg_hip_blocked_kernel_timing(limg_hip_context *pCtx, double *pMs4);
/* Host-only (no GPU touched): the block-similarity predicate `limg_encode_3d_matches` (src/limg.cpp:1137-1268) as the host merge evaluates it
 * for candidates outside the precomputed window; records in `limg_hip_block_record` layout. */
int limg_hip_host_blocked_matches(int channels, const limg_hip_block_record *pSeed, const limg_hip_block_record *pCandidate);
/* Host-only: the greedy raster merge (src/limg.cpp:1386-1496, :1813-1881) over per-block fits; pMatchBits = the similarity bits in the layout the
 * GPU kernel produces (limg_hip_host_blocked_match_words() 64-bit words per block) or NULL (every pair is evaluated on the host).  Writes up to
 * `capacity` rectangles in creation order, always reports the count. */
limg_hip_result limg_hip_host_blocked_merge(const limg_hip_block_record *pFits, const uint64_t *pMatchBits, size_t blocksX, size_t blocksY, int channels,
                                            limg_hip_region *pRegions, size_t capacity, size_t *pCount);
/* Host-only: the similarity bits of every block against its neighbourhood, computed on the host (tests, benchmarks of the merge). */
size_t limg_hip_host_blocked_match_words(void);
limg_hip_result limg_hip_host_blocked_match_bits(const limg_hip_block_record *pFits, size_t blocksX, size_t blocksY, int channels, uint64_t *pMatchBits);

/* ---- compact stream ("LMG3") -----------------------------------------------------------------------------------------------
 * The north-star names `limg_encode()` / `limg_decode()` and a bitstream; upstream has neither (src/limg.h:27-48 is the whole API,
 * SURVEY.md 0.1 / 8(f) #2).  These entry points are the build-defined pair over a container that holds exactly what the reference's
 * decoder (`limg_decode_block_from_factors_3d`, src/limg_decode.h:36-236, called at src/limg.cpp:2093) consumes, so that
 * decode(encode(image)) equals the reference's pDecoded plane bit for bit.  Layout, little endian, sections 8-byte aligned:
 *   limg_hip_stream_header | limg_hip_stream_block[blocksX * blocksY] (raster order) | payload (8-byte words)
 * Block payload at `payloadWord`: factor A field, B field, C field; a field of b = 8 - shift bits per pixel is b words, pixel
 * (row r, column x) of the 8x8 grid at bit (8 r + x) b (pixels outside the image are 0).  shift 8 => no field, except the
 * raw-byte escape (bit 24 + k of `shift`): 4-channel blocks whose factor-k alpha normal is non-zero keep the raw 8-bit factor,
 * because the reference multiplies it into the alpha lane even at shift 8. */
#define LIMG_HIP_STREAM_MAGIC 0x33474D4Cu /* "LMG3" */
#define LIMG_HIP_STREAM_VERSION 1u

typedef struct limg_hip_stream_header
{
  uint32_t magic, version;
  uint32_t sizeX, sizeY;
  uint32_t channels;    /* 3 or 4 (hasAlpha) */
  uint32_t errorFactor; /* informational */
  uint32_t blocksX, blocksY;
  uint64_t payloadWords; /* 8-byte words after the block table */
  uint64_t totalBytes;   /* header + table + payload */
  uint32_t flags;        /* bit 0: fast bit crushing, bit 1: PCG dither (informational) */
  uint32_t reserved[3];
} limg_hip_stream_header; /* 64 bytes */

typedef struct limg_hip_stream_block
{
  int16_t dirA_min[4], dirA_max[4], dirB_offset[4], dirB_mag[4], dirC_offset[4], dirC_mag[4]; /* `limg_encode_3d_output` minus avg */
  uint32_t shift;       /* shiftA | shiftB << 8 | shiftC << 16 | rawEscapeMask << 24 */
  uint32_t payloadWord; /* first payload word of this block */
} limg_hip_stream_block; /* 56 bytes */

/* Worst-case stream size for an image (what `capacity` must be at least); 0 if the image is too large for 32-bit payload offsets. */
size_t limg_hip_stream_bound(size_t sizeX, size_t sizeY);

/* "limg_encode": the encode hot path (same parameters as limg_hip_encode3d_device) followed by the stream packer.  DEVICE pointers,
 * asynchronous on `stream`; the stream's size lands in its header (`totalBytes`).  pBytes (host, may be NULL) receives it too,
 * which makes the call wait for the stream. */
limg_hip_result limg_hip_encode_stream_device(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, uint8_t *pStream,
                                              size_t capacity, size_t *pBytes, uint32_t errorFactor, int poolThreads, int fastBitCrushing, void *stream);
/* "limg_decode" (a16 for every block): DEVICE pointers, asynchronous.  sizeX / sizeY must match the header (the kernel checks and
 * reports a mismatch or inconsistent offsets through limg_hip_check_device_status as limg_hip_error_InvalidParameter). */
limg_hip_result limg_hip_decode_stream_device(limg_hip_context *pCtx, const uint8_t *pStream, size_t streamBytes, uint32_t *pOut, size_t sizeX, size_t sizeY,
                                              void *stream);
/* HOST-pointer variants (blocking). */
limg_hip_result limg_hip_encode_stream(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, uint8_t *pStream, size_t capacity,
                                       size_t *pBytes, uint32_t errorFactor, int poolThreads, int fastBitCrushing);
limg_hip_result limg_hip_decode_stream(limg_hip_context *pCtx, const uint8_t *pStream, size_t streamBytes, uint32_t *pOut, size_t outPixels);
/* Host-only: validates a header (first 64 bytes suffice) and reports the image shape. */
limg_hip_result limg_hip_stream_info(const uint8_t *pStream, size_t streamBytes, size_t *pSizeX, size_t *pSizeY, int *pHasAlpha, size_t *pTotalBytes);

/* ---- multi-GPU (one process per GPU; RCCL over xGMI) ---------------------------------------------------------------------------------
 * The reference's only parallelism is row strips over a std::thread pool (src/limg.cpp:2105-2138); across GPUs the same strips
 * go one per rank.  Blocks are independent except for the dither chain, so the data path needs no collective in strip-restart mode (each strip
 * restarts its chain like the reference's pool strips: plain limg_hip_encode3d_device on the strip).  Two things do cross xGMI:
 *   * limg_hip_gather_stream              : reassembly of the compact streams on one rank (variable-size gather: an 8-byte all-gather of the sizes, then
 *                                           grouped ncclSend / ncclRecv of exactly the used bytes);
 *   * limg_hip_encode3d_single_chain_device: ONE dither chain through all strips in rank order, i.e. the result of the reference run with
 *                                           pThreadPool == nullptr (src/limg.cpp:1893, :2110): one 8-byte all-gather of the per-strip dither-call totals
 *                                           between the E step (fit + search) and the F step (dither + stores) of every rank.
 * RCCL is resolved at run time from the process (dlopen "librccl.so.1"): the library has no link-time dependency on it.  Failures map to
 * limg_hip_error_Generic.  The communicator belongs to the context; the caller only transports the 128-byte id from rank 0 to the others. */
#define LIMG_HIP_COMM_ID_BYTES 128
limg_hip_result limg_hip_comm_unique_id(uint8_t *pId128);                                                  /* rank 0: ncclGetUniqueId */
limg_hip_result limg_hip_comm_init(limg_hip_context *pCtx, const uint8_t *pId128, int rank, int worldSize); /* every rank: ncclCommInitRank on the context's device */
limg_hip_result limg_hip_comm_destroy(limg_hip_context *pCtx);
/* ncclCommUserRank / ncclCommCount of the context's communicator and ncclGetVersion (any pointer may be NULL): evidence for logs and bench lines. */
limg_hip_result limg_hip_comm_info(limg_hip_context *pCtx, int *pRank, int *pRanks, int *pRcclVersion);
/* pStream / pGathered are DEVICE pointers.  On `root`, piece r lands at pGathered + pOffsets[r] (16-byte aligned, ready for limg_hip_decode_stream_device);
 * pOffsets (host, worldSize + 1 entries) also receives the total.  Blocks until the sizes are known; the transfers are asynchronous on `stream`. */
limg_hip_result limg_hip_gather_stream(limg_hip_context *pCtx, const uint8_t *pStream, size_t streamBytes, int root, uint8_t *pGathered, size_t capacity,
                                       uint64_t *pOffsets, void *stream);
/* This rank's strip of `stripRows` rows (whole 8x8 blocks: sizeX, stripRows multiples of 8) of a taller image whose strips go to the ranks in order;
 * blocksBefore = number of 8x8 blocks in the strips of the ranks before this one (sizes the dither noise table).  DEVICE pointers, asynchronous. */
limg_hip_result limg_hip_encode3d_single_chain_device(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t stripRows, int hasAlpha,
                                                      const limg_hip_encode3d_info *pInfo, uint32_t errorFactor, int fastBitCrushing, size_t blocksBefore, void *stream);
/* Abort rule of the two collective entries (the reference's analogue, row strips on one thread pool, cannot half-fail: src/limg.cpp:2114-2136): a rank never leaves its
 * peers inside a collective.  limg_hip_gather_stream decides "fits / does not fit" from all-gathered numbers, so every rank returns limg_hip_error_OutOfBounds alike
 * before anything is posted.  limg_hip_encode3d_single_chain_device: a rank whose own E step failed (bad arguments, allocation, launch) still joins the 8-byte
 * all-gather, with the poison value ~0, and returns its error; on every other rank the call -- it is asynchronous -- returns limg_hip_success, the F step stores
 * NOTHING, and the next limg_hip_check_device_status of that context returns limg_hip_error_Generic ("a rank of the communicator aborted ..."), once. */
/* The two halves of the above without the exchange, for callers that move the counts themselves (and for single-GPU tests of the chain arithmetic):
 * phase 1 = E step + scan, writes this strip's dither-call total to *pCallsDevice; phase 2 = F step, its first dither call is *pChainBaseDevice.
 * Phase 2 must follow phase 1 of the same strip on the same context with nothing in between (the context holds the strip's intermediate results);
 * a phase 2 without that is refused with limg_hip_error_InvalidParameter. */
limg_hip_result limg_hip_encode3d_chain_device(limg_hip_context *pCtx, const uint32_t *pIn, size_t sizeX, size_t stripRows, int hasAlpha, const limg_hip_encode3d_info *pInfo,
                                               uint32_t errorFactor, int fastBitCrushing, int phase, uint64_t *pCallsDevice, const uint64_t *pChainBaseDevice,
                                               size_t blocksBefore, void *stream);
/* Host-only helpers (no GPU): the offset arithmetic of the gather (pOffsets: count + 1 entries) and the exclusive prefix of the call totals. */
limg_hip_result limg_hip_host_gather_offsets(const uint64_t *pSizes, int count, uint64_t *pOffsets);
limg_hip_result limg_hip_host_chain_bases(const uint64_t *pCalls, int count, uint64_t *pBases);

/* Introspection for the bench: names and launch count of the kernels one encode enqueues, bytes of context-owned HBM. */
size_t limg_hip_context_device_bytes(const limg_hip_context *pCtx);
const char *limg_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* LIMG_HIP_H */

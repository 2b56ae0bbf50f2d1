// limg_hip_api.hip -- host side of liblimg_hip.so: context, buffers, launch sequencing, the C ABI of include/limg_hip.h.
// Mirrors the reference's driver (src/limg.cpp:2175-2265 threshold/flag setup, :2105-2138 strip partition).
#include "limg_hip_internal.h"
#ifdef LIMG_HIP_TEST_HOOKS
#include "../../include/limg_hip_test_hooks.h"
#define TOPT(c, member) ((c)->topt.member)
#else
#define TOPT(c, member) 0 /* the product has no test hooks: every use folds to the default */
#endif
#include "limg_hip_rccl.h"
#include "limg_search_table_accurate.h"

#include <math.h>
#include <stdio.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

namespace limg_hip
{
  uint64_t chain_call(uint64_t h, unsigned n, uint8_t *noise64, bool forceSoft, bool pcg);
  uint64_t fill_noise_table(uint64_t h, uint8_t *noise, size_t count, bool pcg);
  uint64_t chain_checkpoints(uint64_t h, size_t calls, size_t every, uint64_t *pOut, bool pcg);
  void chain_walk_rows(uint64_t &h, size_t &call, uint32_t by0, uint32_t by1, uint32_t blocksX, uint32_t stripsX, size_t sizeX, size_t sizeY, uint32_t chainCount, uint32_t chainRows,
                       const uint32_t *shifts, uint32_t *stripBase, unsigned long long *states, uint8_t *pixels, size_t maxCalls, bool pcg);
  size_t chain_walk_blocks(uint64_t h0, uint32_t blocksX, uint32_t blocksY, uint32_t stripsX, size_t sizeX, size_t sizeY, uint32_t chainCount, uint32_t chainRows, const uint32_t *shifts,
                           uint32_t *stripBase, unsigned long long *states, uint8_t *pixels, size_t maxCalls, bool pcg);
}

using namespace limg_hip;

#define HIP_TRY(expr)                                                                                                     \
  do                                                                                                                      \
  {                                                                                                                       \
    const hipError_t e_ = (expr);                                                                                         \
    if (e_ != hipSuccess)                                                                                                 \
    {                                                                                                                     \
      fprintf(stderr, "limg_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__);            \
      return limg_hip_error_Generic;                                                                                      \
    }                                                                                                                     \
  } while (0)

struct DevBuf
{
  void *p = nullptr;
  size_t cap = 0;
  limg_hip_result ensure(size_t bytes)
  {
    if (bytes <= cap) return limg_hip_success;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return limg_hip_error_MemoryAllocationFailure; }
    cap = bytes;
    return limg_hip_success;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// pinned host memory owned by the context (staging of the merged-block encoder's host stages: no zero fill, full-rate PCIe copies)
struct HostBuf
{
  void *p = nullptr;
  size_t cap = 0;
  limg_hip_result ensure(size_t bytes)
  {
    if (bytes <= cap) return limg_hip_success;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4; // grow with slack: sizes depend on the image content
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return limg_hip_error_MemoryAllocationFailure; }
    cap = want;
    return limg_hip_success;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct limg_hip_context
{
  // The reference's entry points are re-entrant (scratch on the stack, src/limg.cpp:1890-1891; the only globals are CPUID flags, src/limg_simd.cpp:57-60), so a
  // caller may encode from several threads at once.  A context owns device scratch, so the blocking host-pointer entries (what the shim's limg_encode3d_test & co.
  // call) serialise on this mutex: any number of threads may share one context through them.  The asynchronous *_device entries enqueue work that uses that
  // scratch after they return: one context per stream there (documented in limg_hip.h).
  std::recursive_mutex hostEntry;
  int device = 0;
  limg_hip_options opt;
#ifdef LIMG_HIP_TEST_HOOKS
  limg_hip_test_options topt; // liblimg_hip_test.so only (include/limg_hip_test_hooks.h)
#endif
  DevBuf records, shifts, stripCalls, stripBase; // per-block / per-strip scratch
  DevBuf invN;                                   // per block 1 / |normal|^2 of the three factors (k_fit_tpb -> E step)
  DevBuf noise;                                  // static dither noise table (full-block chains)
  bool noisePcg = false;                         // which generator the table was built with
  size_t noiseCount = 0;                         // entries generated so far
  uint64_t noiseNext = kDitherSeed;              // chain value after the last generated entry
  DevBuf noiseDyn;                               // data-dependent chains (images with partial blocks)
  DevBuf noiseStates;                            // ... their per-call chain values + pixel counts as the host uploads them (k_noise_expand -> noiseDyn)
  DevBuf noiseCk;                                // the chain checkpoints (limg_noise_checkpoints.h) on the device: the GPU fills the noise table from them
  size_t noiseCkCount = 0;                       // ... how many dense values (every 1024th call) are there: the embedded ones, or more (ensure_checkpoints)
  std::vector<uint64_t> noiseCkHost;             // ... and, once an image has reached beyond the embedded dense values, the host copy they were uploaded from
  DevBuf park;                                   // persistent kernel: 2 x 8 KiB per workgroup
  DevBuf batchTable;                             // batched encode: one ImageIO per image
  hipStream_t fitStream = nullptr;               // batched encode in sub-batches: k_fit_tpb of sub-batch k + 1 runs here, next to the persistent kernel of sub-batch k
  std::vector<hipEvent_t> pipeEvents;            // ... and the events that fork it from / join it to the caller's stream
  HostBuf hStage;                                // pinned staging of the ragged paths' host step (shift words down; chain bases and noise up)
  hipEvent_t hStageEvent = nullptr;              // ... recorded behind the last asynchronous H2D copy that reads it: waited for before it is written, grown or freed again
  bool hStageBusy = false;
  std::vector<hipEvent_t> raggedEvents;          // banded ragged encode: "the shift words of band b are down"
  DevBuf stats;                                  // limg_hip_options.collect_stats: the reference's 3 + 27 bit counters of the last encode
  hipStream_t statsStream = nullptr;
  int statsState = 0;                            // 0 = none, 1 = on the device (8x8 path), 2 = in statsHost (merged-block encoder)
  bool statsAccumulate = false;                  // a batched encode in several launch pairs: the pairs after the first add to the counters instead of restarting them
  uint64_t statsHost[30] = { 0 };
  uint64_t statsPixels = 0;
  DevBuf lookback;                               // fused path: ticket (16 B) then one 8-byte descriptor per work strip
  DevBuf accTable;                               // accurate search: automaton expanded to 32-byte entries (built on the first accurate encode)
  DevBuf devStatus;                              // sticky look-back timeout word: never touched by the per-launch memset, cleared by limg_hip_check_device_status
  DevBuf in, planes;                             // staging for the host-pointer entry points
  hipStream_t hostCopyStream = nullptr;          // ... the downloads of the finished bands (second host thread)
  hipStream_t hostStream = nullptr;              // ... in row bands: the bands' kernels run here, their events tell the download thread when a band is done
  std::vector<hipEvent_t> hostEvents;
  DevBuf hostWords;                              // ... per band its dither-call total and its chain base (one chain through the bands)
  DevBuf cmp;                                    // 8-byte accumulator of limg_hip_compare
  DevBuf bFlags, bBound;
  DevBuf bOrder; // merged-block encoder: per batch the order its workgroups take the rectangles in
  DevBuf bMatch, bRegions, bOut, bPx, bFac, bNoise, bNoiseBase; // merged-block encoder: similarity bits, region table / results, scratch (gathered pixels, factor bytes), noise
  HostBuf hFlags;
  HostBuf hRec, hBits, hDesc, hOut, hNoise, hNoiseBase;
  hipStream_t workStream = nullptr; // the merged-block encoder's worker thread launches on its own stream
  std::vector<hipStream_t> workStreams; // ... its fit + search batches round-robin on these
  hipStream_t storeStream = nullptr; // ... and the noise expansion + store kernels of a batch on a second one
  DevBuf bCalls;                     // per dither call of the merged-block encoder: chain value, noise offset, pixel count (host walk -> k_noise_expand_calls)
  std::vector<hipEvent_t> workEvents;        // one per batch of the merged-block encoder's worker that is in flight on the GPU
  hipStream_t copyStream = nullptr;      // copies of the similarity-bit bands, behind the kernels that produce them
  std::vector<hipEvent_t> bandEvents;
  std::vector<HostRegion> lastRegions;
  size_t lastBlocks = 0;                     // blocks of the last merged-block encode (what hBits / lastRegions describe)
  double blockedMs[6] = { 0, 0, 0, 0, 0, 0 };
  double blockedKernelMs[4] = { 0, 0, 0, 0 }; // the last merged-block encode, HIP events: pass 1 (k_fit_tpb) / the k_blocked_match launches / the k_blocked_fit_search launches /
                                             // the noise-expansion + store launches (the last two summed over the worker's batches)
  std::vector<hipEvent_t> workTimers;        // [4 i .. 4 i + 3]: begin / end of batch slot i's fit + search kernel, begin / end of its expansion + store kernels;
                                             // [4 kInFlight ..]: begin of pass 1, end of pass 1 = begin of the similarity kernels, their end
  // multi-GPU (RCCL over xGMI): one communicator per context, created by limg_hip_comm_init
  ncclComm_t comm = nullptr;
  int commRank = 0, commWorld = 1;
  // limg_hip_encode3d_chain_device: phase 2 is only valid right after phase 1 of the same strip (the context holds the intermediate results)
  const void *chainIn = nullptr;
  size_t chainX = 0, chainY = 0, chainBefore = 0;
  const void *chainFac[3] = { nullptr, nullptr, nullptr }; // phase 1 left the pre-dither factor bytes in these planes
  int chainAlpha = 0, chainFast = 0;
  uint32_t chainEf = 0;
  DevBuf commWords; // [0] this rank's value, [1] its chain base, [8 ...] the all-gathered values
  DevBuf streamFac, streamTiles, streamUnits, streamStatus, streamBuf; // stream packer: 3 factor planes, per-tile payload words; decode status word; host-entry staging
  // optional per-kernel timing (bench): 4 events per encode, recorded on the caller's stream, read back in one go
  int persistentWorkgroups = 1280; // 5 x the device's CU count (set at init): the unit the launches scale (x 6 / 5 with the float stage in its own kernel)
  bool forceSplit = false; // options: run the three-kernel path even where the fused kernel applies (A/B, tests)
  bool profiling = false;
  std::vector<hipEvent_t> events;
  size_t eventsUsed = 0;
};

namespace
{
  // Developer print-outs of the merged-block encoder's pipeline (stderr): compile-time switches (-DLIMG_HIP_DEBUG_TIMELINE / -DLIMG_HIP_DEBUG_TIMING through
  // limg_amd.build.build(extra_flags=...)), never the environment -- the shipped library reads no environment variable and prints nothing on success.
#ifdef LIMG_HIP_DEBUG_TIMELINE
  constexpr bool kDebugTimeline = true;
#else
  constexpr bool kDebugTimeline = false;
#endif
#ifdef LIMG_HIP_DEBUG_TIMING
  constexpr bool kDebugTiming = true;
#else
  constexpr bool kDebugTiming = false;
#endif
  constexpr size_t kNoiseChunk = 1u << 16; // table growth granularity (entries)

  // The accurate search's automaton (tools/make_search_table.py, src/limg_bit_crush.h:668-830) in the form the kernel's scalar loads want: 8 dwords per state,
  // every field in a dword of its own = { a | phase2 << 5 | final << 31, byte offset on pass, byte offset on fail, b, c, mul(a), mul(b), mul(c) }.  Bits 24..26 of the two
  // offsets say which factors' shifts the SUCCESSOR's triple changes against this state's (A, B, C): a state of this DAG has several predecessors, so the change mask is a
  // property of the edge -- with it the kernel needs no record of the shifts its cached terms were built for and no three compares per trial.
  limg_hip_result ensure_accurate_table(limg_hip_context *c)
  {
    if (c->accTable.p) return limg_hip_success;
    static const uint32_t compact[LIMG_SEARCH_ACC_STATES][2] = LIMG_SEARCH_ACC_TABLE_INIT;
    static const uint32_t mul[9] = { 1, 2, 4, 8, 17, 36, 85, 255, 256 }; // (1 << s) + decode_bias(s), src/limg_bit_crush_simd.h:611-619
    std::vector<uint32_t> wide((size_t)LIMG_SEARCH_ACC_STATES * 8);
    for (size_t i = 0; i < (size_t)LIMG_SEARCH_ACC_STATES; i++)
    {
      const uint32_t w0 = compact[i][0], w1 = compact[i][1];
      uint32_t *e = &wide[i * 8];
      if (w0 >> 31) { e[0] = 1u << 31; continue; }
      const uint32_t a = w0 & 15u, b = (w0 >> 4) & 15u, cc = (w0 >> 8) & 15u;
      e[0] = a | ((w0 & 0x1000u) ? 0x20u : 0u);
      e[1] = (w1 & 0xFFFFu) * 32u; e[2] = (w1 >> 16) * 32u;
      e[3] = b; e[4] = cc;
      e[5] = mul[a]; e[6] = mul[b]; e[7] = mul[cc];
    }
    static_assert((size_t)LIMG_SEARCH_ACC_STATES * 32u < (1u << 24), "offsets leave bits 24..26 free");
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
    limg_hip_result r = c->accTable.ensure(wide.size() * 4);
    if (r != limg_hip_success) return r;
    if (hipMemcpy(c->accTable.p, wide.data(), wide.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { c->accTable.release(); return limg_hip_error_Generic; }
    return limg_hip_success;
  }

  // How far the GPU-filled noise table reaches: the far checkpoints' last value + one far stretch (2^27 calls).
  size_t checkpoint_reach()
  {
    size_t farCount = 0, farEvery = 0;
    (void)noise_checkpoints_far_host(&farCount, &farEvery);
    return farCount * farEvery;
  }

  // work(0) .. work(n - 1), each on a host thread of its own where the host lets us start one.  A thread that cannot be created (EAGAIN under a pid / thread limit) or a
  // pool that cannot be allocated must neither leave joinable threads behind (their destructor calls std::terminate) nor send an exception across the extern "C"
  // boundary: whatever did not get a thread runs on the calling thread.  `work` itself must not throw.
  template <class F>
  void run_on_threads(unsigned n, F &&work) noexcept
  {
    std::thread *pool = n > 1 ? new (std::nothrow) std::thread[n - 1] : nullptr; // default-constructed: not joinable
    unsigned started = 0;
    if (pool)
      for (; started < n - 1; started++)
      {
        try { pool[started] = std::thread(work, started + 1); }
        catch (...) { break; }
      }
    work(0u);
    for (unsigned t = started + 1; t < n; t++) work(t);
    for (unsigned t = 0; t < started; t++) pool[t].join();
    delete[] pool;
  }

  // Dense chain values (every LIMG_NOISE_CHECKPOINT_EVERY = 1024 calls) number first .. first + count - 1 into pOut: the embedded dense table where it reaches (16 Mi
  // calls), beyond it the embedded FAR values (every 65536 calls) walked on foot -- 65536 calls of 8 AES rounds per far value = 0.5 ms, far values independent of each
  // other: on up to 16 host threads.  false beyond the far table's reach.
  bool dense_checkpoints_host(size_t first, size_t count, uint64_t *pOut)
  {
    size_t ckCount = 0, ckEvery = 0, farCount = 0, farEvery = 0;
    const uint64_t *ck = noise_checkpoints_host(&ckCount, &ckEvery);
    const uint64_t *far = noise_checkpoints_far_host(&farCount, &farEvery);
    const size_t perFar = farEvery / ckEvery;
    if (count == 0) return true;
    if (first + count > farCount * perFar) return false;
    size_t k = 0;
    for (; k < count && first + k < ckCount; k++) pOut[k] = ck[first + k];
    if (k == count) return true;
    const size_t j0 = (first + k) / perFar, j1 = (first + count - 1) / perFar + 1; // far stretches touched
    unsigned threads = std::thread::hardware_concurrency();
    if (threads == 0 || threads > 16) threads = 16;
    if (threads > j1 - j0) threads = (unsigned)(j1 - j0);
    uint64_t scratch[16][64]; // one far stretch's dense values per thread (nothing may throw inside the threads)
    if (perFar > 64) return false;
    auto work = [&](unsigned t) {
      uint64_t *tmp = scratch[t];
      for (size_t j = j0 + t; j < j1; j += threads)
      {
        (void)chain_checkpoints(far[j], farEvery, ckEvery, tmp, false);
        for (size_t q = 0; q < perFar; q++)
        {
          const size_t idx = j * perFar + q;
          if (idx >= first + k && idx < first + count) pOut[idx - first] = tmp[q];
        }
      }
    };
    run_on_threads(threads, work);
    return true;
  }

  // Dense chain checkpoints covering dither calls [0, calls) on the device (c->noiseCk): the embedded table once per context, more when an image reaches beyond it
  // (more than 5.59 M blocks: the missing values come from the far table, dense_checkpoints_host).  Blocking copies: whichever stream fills a noise table later
  // finds them there (an asynchronous copy on the first caller's stream would order nothing for a second stream), and a failed copy leaves no buffer behind that
  // later encodes would trust.  (A buffer that grows is freed first: hipFree waits for the fill kernels that may still read it.)
  limg_hip_result ensure_checkpoints(limg_hip_context *c, size_t calls)
  {
    size_t ckCount = 0, ckEvery = 0;
    const uint64_t *ck = noise_checkpoints_host(&ckCount, &ckEvery);
    size_t need = (calls + ckEvery - 1) / ckEvery;
    if (need < ckCount) need = ckCount;
    if (c->noiseCk.p && need <= c->noiseCkCount) return limg_hip_success;
    if (calls > checkpoint_reach()) return limg_hip_error_InvalidParameter;
    const uint64_t *src = ck;
    if (need > ckCount)
    {
      try
      {
        std::vector<uint64_t> &v = c->noiseCkHost;
        if (v.empty()) v.assign(ck, ck + ckCount);
        const size_t have = v.size();
        if (need > have)
        {
          v.resize(need);
          if (!dense_checkpoints_host(have, need - have, v.data() + have)) { v.resize(have); return limg_hip_error_InvalidParameter; }
        }
        src = v.data();
      }
      catch (...) { c->noiseCkHost.clear(); return limg_hip_error_MemoryAllocationFailure; }
    }
    limg_hip_result r;
    c->noiseCkCount = 0;
    if ((r = c->noiseCk.ensure(need * 8)) != limg_hip_success) return r;
    if (hipMemcpy(c->noiseCk.p, src, need * 8, hipMemcpyHostToDevice) != hipSuccess)
    {
      c->noiseCk.release();
      fprintf(stderr, "limg_hip: upload of the dither chain checkpoints failed\n");
      return limg_hip_error_Generic;
    }
    c->noiseCkCount = need;
    return limg_hip_success;
  }

  limg_hip_result grow_noise_table(limg_hip_context *c, size_t entries, hipStream_t stream)
  {
    const bool pcg = c->opt.dither_pcg != 0;
    if (pcg != c->noisePcg) c->noiseCount = 0;
    if (entries <= c->noiseCount) return limg_hip_success;
    const size_t want = ((entries + kNoiseChunk - 1) / kNoiseChunk) * kNoiseChunk;
    if (!pcg && want <= checkpoint_reach() && c->opt.host_noise_table == 0)
    { // the AES stream, on the GPU from the embedded chain checkpoints (limg_hip_noise_gpu.hip): stream-ordered, ~1 ms, nothing crosses PCIe but the 128 KiB of
      // checkpoints, once per context (images of more than 5.59 M blocks: 8 bytes more per 1024 calls beyond the embedded dense table's 16 Mi, made from the far
      // table on host threads -- ~20 ms for the 50 M calls of a 32768^2 image, where walking the whole chain on one host thread and uploading 3.2 GB took 1.5 s).
      // A larger table than the one at hand is filled from scratch (its prefix is the same stream).
      limg_hip_result r;
      if ((r = ensure_checkpoints(c, want)) != limg_hip_success) return r;
      HIP_TRY(hipStreamSynchronize(stream)); // earlier encodes on this stream may still read the table that is about to be replaced
      if ((r = c->noise.ensure(want * 64)) != limg_hip_success) return r;
      launch_noise_fill((uint8_t *)c->noise.p, (const uint64_t *)c->noiseCk.p, want, stream);
      HIP_TRY(hipGetLastError());
      c->noiseCount = want;
      c->noisePcg = false;
      return limg_hip_success;
    }
    // PCG dither (a test / fallback mode), tables beyond the far checkpoints' reach (2^27 calls: images of more than 44.7 M blocks) or limg_hip_options.host_noise_table:
    // (re)generate on the host; one-time cost per context and image size class
    std::vector<uint8_t> host(want * 64);
    uint64_t h = kDitherSeed;
    h = fill_noise_table(h, host.data(), want, pcg);
    HIP_TRY(hipStreamSynchronize(stream));
    const limg_hip_result r = c->noise.ensure(want * 64);
    if (r != limg_hip_success) return r;
    HIP_TRY(hipMemcpy(c->noise.p, host.data(), want * 64, hipMemcpyHostToDevice));
    c->noiseCount = want;
    c->noisePcg = pcg;
    c->noiseNext = h;
    return limg_hip_success;
  }

  void mark(limg_hip_context *c, hipStream_t stream)
  {
    if (!c->profiling) return;
    if (c->eventsUsed == c->events.size())
    {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      c->events.push_back(e);
    }
    (void)hipEventRecord(c->events[c->eventsUsed++], stream);
  }

  struct Partition { uint32_t chainCount, chainRows; };

  // src/limg.cpp:2114-2134 in block rows
  Partition partition(size_t sizeY, int poolThreads)
  {
    Partition pt = { 1, 0 };
    if (poolThreads <= 0) return pt;
    size_t thread_count = (size_t)poolThreads * 4;
    size_t y_range = ((sizeY / kBlock) / thread_count) * kBlock;
    if (y_range == 0)
    {
      thread_count = (size_t)poolThreads;
      y_range = ((sizeY / kBlock) / thread_count) * kBlock;
    }
    if (y_range == 0) return pt; // every strip but the last is empty
    pt.chainCount = (uint32_t)thread_count;
    pt.chainRows = (uint32_t)(y_range / kBlock);
    return pt;
  }

  uint32_t chain_of_row(const Partition &pt, uint32_t row)
  {
    if (pt.chainCount <= 1 || pt.chainRows == 0) return 0;
    const uint32_t c = row / pt.chainRows;
    return c < pt.chainCount - 1 ? c : pt.chainCount - 1;
  }

  // What the public entries add to the plain (single image, whole encode) call.
  struct EncodeExtra
  {
    bool streamRaw = false, fitOnly = false;
    uint32_t *stripWords = nullptr; // stream mode, images of whole blocks: per work strip the payload words of its blocks (EncodeParams::stripWords)
    int chainPhase = 0; // 0 = whole encode; 1 = E step + scan only (writes *dChainCalls); 2 = F step only (reads *dChainBase).  1 and 2 always take the split path.
    unsigned long long *dChainCalls = nullptr;
    const unsigned long long *dChainBase = nullptr;
    size_t chainBlocksBefore = 0;
    // batch (host array of batchCount entries, batchCount > 1): the images of a batched encode -- same shape, whole 8x8 blocks, all 11 planes -- in one launch
    // pair (or, limg_hip_options.batch_sub_images, a pipeline of launch pairs); dIn / dInfo are then those of image 0.  The caller has checked all of that.
    const ImageIO *batch = nullptr;
    size_t batchCount = 1;
    // ---- a sub-image of a larger encode (the two parts of an image whose last block row is partial: encode_height_ragged) ----
    bool inner = false;            // part of a larger encode: no statistics launch of its own, no reset of the context's chain / statistics state
    int marks = 2;                 // profiling events: 2 = all four, 1 = all but the last, 0 = none
    const Partition *part = nullptr; // the dither-chain partition of the WHOLE image (a sub-image cannot derive it from its own height)
    size_t scratchRow0 = 0;        // this sub-image's first block row in the per-block scratch (and in the caller's compact outputs)
    size_t scratchRows = 0;        // block rows the scratch must hold (0: this call's own)
    const unsigned long long *dPrevDesc = nullptr; // ragged sub-image: its chain continues the one whose dither-call count is the low word of this look-back descriptor
  };

  limg_hip_result encode_height_ragged(limg_hip_context *c, const uint32_t *dIn, size_t sizeX, size_t sizeY, int hasAlpha, const limg_hip_encode3d_info *dInfo,
                                       const limg_hip_compact_out *compact, uint32_t errorFactor, int poolThreads, int fast, hipStream_t stream, const EncodeExtra &x);

  // pinned staging of the ragged paths' host step: shift words | previous descriptor | strip bases | per call: chain value, pixel count (worst case: 3 calls per block)
  size_t ragged_stage_bytes(size_t blocks, size_t strips)
  {
    const size_t maxCalls = blocks * 3;
    const size_t offPrev = (blocks * 4 + 15) & ~(size_t)15, offBase = offPrev + 16, offStates = (offBase + strips * 4 + 15) & ~(size_t)15, offPixels = offStates + maxCalls * 8;
    return offPixels + maxCalls + 16;
  }

  // The chain value the dither call number `calls` of a chain of full 8x8 blocks starts from: the nearest embedded checkpoint, then at most 1023 calls on foot.
  bool chain_value_at(uint64_t calls, uint64_t *pValue)
  {
    size_t ckCount = 0, ckEvery = 0, farCount = 0, farEvery = 0;
    const uint64_t *ck = noise_checkpoints_host(&ckCount, &ckEvery);
    const uint64_t *far = noise_checkpoints_far_host(&farCount, &farEvery);
    uint64_t h, onFoot;
    if (calls / ckEvery < ckCount) { h = ck[calls / ckEvery]; onFoot = calls % ckEvery; }
    else if (calls / farEvery < farCount) { h = far[calls / farEvery]; onFoot = calls % farEvery; } // beyond the dense table: at most 65535 calls on foot (0.5 ms)
    else return false;
    for (uint64_t i = 0; i < onFoot; i++) h = chain_call(h, 64, nullptr, false, false);
    *pValue = h;
    return true;
  }

  limg_hip_result ensure_pipe_events(limg_hip_context *c, size_t n)
  {
    if (!c->fitStream) HIP_TRY(hipStreamCreateWithFlags(&c->fitStream, hipStreamNonBlocking));
    while (c->pipeEvents.size() < n)
    {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      c->pipeEvents.push_back(e);
    }
    return limg_hip_success;
  }

  limg_hip_result encode_device(limg_hip_context *c, const uint32_t *dIn, size_t sizeX, size_t sizeY, int hasAlpha, const limg_hip_encode3d_info *dInfo,
                                const limg_hip_compact_out *compact, uint32_t errorFactor, int poolThreads, int fast, hipStream_t stream, const EncodeExtra &x = EncodeExtra())
  {
    const int chainPhase = x.chainPhase;
    const size_t batchCount = x.batchCount;
    const ImageIO *const batch = x.batch;
    if (!c || !dIn) return limg_hip_error_ArgumentNull;
    if (chainPhase == 0 && !x.inner) c->chainIn = nullptr; // the context's per-block scratch is about to be reused
    if (sizeX == 0 || sizeY == 0 || sizeX > 0x7FFFFFF8ull || sizeY > 0x7FFFFFF8ull) return limg_hip_error_InvalidParameter;
    bool fullPlanes = true;
    if (dInfo)
    {
      // either all 11 planes, or (compact mode) only the three factor planes with the eight uint32 planes all NULL
      const void *const *pp = reinterpret_cast<const void *const *>(dInfo);
      int n32 = 0;
      for (int i = 0; i < 8; i++) n32 += pp[i] != nullptr;
      for (int i = 8; i < 11; i++)
        if (!pp[i]) return limg_hip_error_ArgumentNull;
      if (n32 != 0 && n32 != 8) return limg_hip_error_ArgumentNull;
      fullPlanes = n32 == 8;
    }
    HIP_TRY(hipSetDevice(c->device));
    const bool ragged = (sizeX % kBlock) != 0 || (sizeY % kBlock) != 0;
    // An image whose WIDTH is whole blocks but whose last block row is partial (BASELINE config 1's shape class: 1024 x 618): every block row but the last
    // is on the fast path (its dither chain is the seed's orbit, the noise table applies); only the last row needs the host to walk its chain.
    if (ragged && sizeX % kBlock == 0 && sizeY > (size_t)kBlock && dInfo && chainPhase == 0 && batchCount == 1 && !x.inner && !x.fitOnly && !c->forceSplit &&
        c->opt.legacy_float_stage == 0 && c->opt.dither_pcg == 0 && TOPT(c, whole_image_ragged) == 0)
    {
      if (((sizeX / kBlock) * ((sizeY + kBlock - 1) / kBlock)) * 3 <= checkpoint_reach())
        return encode_height_ragged(c, dIn, sizeX, sizeY, hasAlpha, dInfo, compact, errorFactor, poolThreads, fast, stream, x);
    }
    auto mark_if = [&](int level) { if (x.marks >= level) mark(c, stream); };

    EncodeParams p;
    memset(&p, 0, sizeof(p));
    p.io.in = dIn;
    p.sizeX = (uint32_t)sizeX; p.sizeY = (uint32_t)sizeY;
    p.blocksX = (uint32_t)((sizeX + kBlock - 1) / kBlock);
    p.blocksY = (uint32_t)((sizeY + kBlock - 1) / kBlock);
    p.stripsX = (p.blocksX + kStripBlocks - 1) / kStripBlocks;
    // thresholds and flags, src/limg.cpp:2186-2212
    const uint64_t maxPixel = (uint64_t)0x6 * (errorFactor / 2) * 7, maxBlock = (uint64_t)0x4 * (errorFactor / 2) * 7;
    p.maxPixel32 = maxPixel > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)maxPixel;
    p.maxBlock = maxBlock;
    { const uint64_t lim = (maxBlock * 64ull + 15ull) >> 4; p.blockLimitFull = lim > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)lim; }
    p.crushBits = errorFactor != 0;
    p.fast = fast != 0;
    p.accTable = nullptr;
    if (!p.fast && p.crushBits)
    {
      limg_hip_result ra = ensure_accurate_table(c);
      if (ra != limg_hip_success) return ra;
      p.accTable = (const uint32_t *)c->accTable.p;
    }
    const bool forced = c->opt.forced_shift[0] >= 0 && c->opt.forced_shift[0] <= 8 && c->opt.forced_shift[1] >= 0 && c->opt.forced_shift[1] <= 8 &&
                        c->opt.forced_shift[2] >= 0 && c->opt.forced_shift[2] <= 8;
    for (int i = 0; i < 3; i++) p.forced[i] = forced ? c->opt.forced_shift[i] : -1;
    p.floatFast = (c->opt.float_mode == 1 && !x.fitOnly) ? 1 : 0;
    p.recordLimit = TOPT(c, record_limit) > 0 ? TOPT(c, record_limit) - 1 : 2700; // see kTermBias in limg_hip_kernels.hip: 3 * 2700 + 1 < 0x2000
    const Partition pt = x.part ? *x.part : partition(sizeY, poolThreads);
    p.chainCount = pt.chainCount; p.chainRows = pt.chainRows;

    p.batchCount = (uint32_t)batchCount;
    p.imageStrips = p.stripsX * p.blocksY;
    const size_t blocks = (size_t)p.blocksX * p.blocksY * batchCount, strips = (size_t)p.stripsX * p.blocksY * batchCount; // of all images
    // the per-block / per-strip scratch: sized for the larger encode this call may be a part of, addressed from this part's first block row
    const size_t scratchRows = x.scratchRows > (size_t)p.blocksY * batchCount ? x.scratchRows : (size_t)p.blocksY * batchCount;
    const size_t scratchBlocks = scratchRows * p.blocksX, scratchStrips = scratchRows * p.stripsX;
    const size_t blockOff = x.scratchRow0 * p.blocksX, stripOff = x.scratchRow0 * p.stripsX;
    limg_hip_result r;
    if (compact && compact->pRecords) p.records = compact->pRecords + blockOff;
    else { if ((r = c->records.ensure(scratchBlocks * sizeof(limg_hip_block_record))) != limg_hip_success) return r; p.records = (limg_hip_block_record *)c->records.p + blockOff; }
    if (compact && compact->pShifts) p.shifts = compact->pShifts + blockOff;
    else { if ((r = c->shifts.ensure(scratchBlocks * 4)) != limg_hip_success) return r; p.shifts = (uint32_t *)c->shifts.p + blockOff; }
    if ((r = c->invN.ensure(scratchBlocks * 16)) != limg_hip_success) return r;
    p.invN = (float *)c->invN.p + blockOff * 4;
    if ((r = c->stripCalls.ensure(scratchStrips * 4)) != limg_hip_success) return r;
    if ((r = c->stripBase.ensure(scratchStrips * 4)) != limg_hip_success) return r;
    p.stripCalls = (uint32_t *)c->stripCalls.p + stripOff; p.stripBase = (uint32_t *)c->stripBase.p + stripOff;
    p.storePlanes = dInfo != nullptr;
    p.fullPlanes = fullPlanes;
    p.streamRaw = x.streamRaw && !fullPlanes;
    p.stripWords = (x.streamRaw && !fullPlanes) ? x.stripWords : nullptr;
    p.fitOnly = x.fitOnly && !dInfo;
    if (dInfo) p.io.info = *dInfo;
    // 16-byte vector access straight on caller pointers only where the address is 16-byte aligned for every row: sliced or offset
    // device pointers take the dword paths
    p.vecIn = (sizeX % 4 == 0) && (((uintptr_t)dIn) & 15u) == 0;
    p.vecPlanes = 0;
    if (dInfo && fullPlanes && sizeX % 4 == 0)
    {
      uintptr_t bits = 0;
      const void *const *pp = reinterpret_cast<const void *const *>(dInfo);
      for (int i = 1; i < 8; i++) bits |= (uintptr_t)pp[i]; // pShiftABCX .. pColCMax
      p.vecPlanes = (bits & 15u) == 0;
    }
    p.vecFactors8 = dInfo && (sizeX % 8 == 0) && ((((uintptr_t)dInfo->pFactorsA) | ((uintptr_t)dInfo->pFactorsB) | ((uintptr_t)dInfo->pFactorsC)) & 7u) == 0;
    p.vecDecoded = dInfo && fullPlanes && (sizeX % 4 == 0) && (((uintptr_t)dInfo->pDecoded) & 15u) == 0;
    p.vecFactors = dInfo && (sizeX % 16 == 0) && ((((uintptr_t)dInfo->pFactorsA) | ((uintptr_t)dInfo->pFactorsB) | ((uintptr_t)dInfo->pFactorsC)) & 15u) == 0;
    const int channels = hasAlpha ? 4 : 3;
    if (batchCount > 1)
    { // the 16-byte access paths only if every image of the batch allows them; the table goes to the device behind whatever the stream still holds
      for (size_t i = 1; i < batchCount; i++)
      {
        uintptr_t bits = 0;
        const void *const *pp = reinterpret_cast<const void *const *>(&batch[i].info);
        for (int k = 1; k < 8; k++) bits |= (uintptr_t)pp[k];
        p.vecPlanes = p.vecPlanes && (bits & 15u) == 0;
        p.vecIn = p.vecIn && (((uintptr_t)batch[i].in) & 15u) == 0;
        p.vecDecoded = p.vecDecoded && (((uintptr_t)batch[i].info.pDecoded) & 15u) == 0;
        p.vecFactors8 = p.vecFactors8 && ((((uintptr_t)batch[i].info.pFactorsA) | ((uintptr_t)batch[i].info.pFactorsB) | ((uintptr_t)batch[i].info.pFactorsC)) & 7u) == 0;
        p.vecFactors = p.vecFactors && ((((uintptr_t)batch[i].info.pFactorsA) | ((uintptr_t)batch[i].info.pFactorsB) | ((uintptr_t)batch[i].info.pFactorsC)) & 15u) == 0;
      }
      if ((r = c->batchTable.ensure(batchCount * sizeof(ImageIO))) != limg_hip_success) return r;
      launch_set_batch_table((ImageIO *)c->batchTable.p, batch, batchCount, stream); // through kernel arguments: stream-ordered, and the host array may die right away
      p.batch = (const ImageIO *)c->batchTable.p;
    }

    if (dInfo && !ragged)
    {
      // longest chain, in blocks: every block makes at most 3 dither calls
      uint32_t maxRows = p.blocksY;
      if (pt.chainCount > 1 && (pt.chainCount - 1) * pt.chainRows < p.blocksY) maxRows = p.blocksY - (pt.chainCount - 1) * pt.chainRows; // the last chain takes the remainder, never fewer rows than the others
      if ((r = grow_noise_table(c, ((size_t)maxRows * p.blocksX + x.chainBlocksBefore) * 3, stream)) != limg_hip_success) return r;
      p.noise = (const uint8_t *)c->noise.p;
      p.noiseLast = (uint32_t)(c->noiseCount - 1);
    }

    if (chainPhase != 0 && (ragged || !dInfo || poolThreads != 0)) return limg_hip_error_InvalidParameter; // a chain shared between GPUs: whole 8x8 blocks, one chain
    p.chainCallsOut = chainPhase == 1 ? x.dChainCalls : nullptr;
    p.chainBase = chainPhase == 2 ? x.dChainBase : nullptr;
    // The float stage as its own launch, one lane per block (limg_hip_fit_tpb.hip), wherever every block is a whole 8x8: the E step then starts from the records.
    p.prefit = (!ragged && c->opt.legacy_float_stage == 0 && (((uintptr_t)p.records) & 15u) == 0) ? 1 : 0; // k_fit_tpb stores records 16 bytes at a time
    const bool fused = dInfo != nullptr && !ragged && (!c->forceSplit || batchCount > 1) && chainPhase == 0;
    const bool wantStats = c->opt.collect_stats != 0 && dInfo != nullptr && chainPhase == 0;
    auto stats = [&]() -> limg_hip_result
    { // the reference's "Average Block Bits" counters (src/limg.cpp:1971-1999) of this encode, left on the device for limg_hip_last_stats
      if (!wantStats || x.inner) return limg_hip_success;
      limg_hip_result rs;
      if ((rs = c->stats.ensure(30 * 8)) != limg_hip_success) return rs;
      if (!c->statsAccumulate) HIP_TRY(hipMemsetAsync(c->stats.p, 0, 30 * 8, stream));
      launch_shift_stats(p.shifts, p.blocksX, p.blocksY, p.blocksY * p.batchCount, p.sizeX, p.sizeY, (unsigned long long *)c->stats.p, stream);
      c->statsStream = stream; c->statsState = 1;
      c->statsPixels = (c->statsAccumulate ? c->statsPixels : 0) + (uint64_t)sizeX * sizeY * batchCount;
      return limg_hip_success;
    };
    if (chainPhase == 0 && !x.inner && !c->statsAccumulate) c->statsState = 0;
    if (batchCount > 1 && (!fused || !p.prefit || !fullPlanes)) return limg_hip_error_InvalidParameter; // (limg_hip_encode3d_batch_device sends such lists through one encode per image instead)
    if (fused)
    { // the persistent kernel's ticket (16 B) and one 8-byte look-back descriptor per work strip, zero at its start: k_fit_tpb clears them on its way (one launch
      // and its gaps less per image); without that kernel, a memset
      if ((r = c->lookback.ensure(16 * (batchCount + 1) + strips * 8)) != limg_hip_success) return r; // (room for one ticket per sub-batch)
      p.ticket = (uint32_t *)c->lookback.p;
      p.desc = (unsigned long long *)((uint8_t *)c->lookback.p + 16);
      p.zeroLookback = p.prefit ? 1 : 0;
      if (!p.prefit) HIP_TRY(hipMemsetAsync(c->lookback.p, 0, 16 + strips * 8, stream));
      if (!c->devStatus.p)
      {
        if ((r = c->devStatus.ensure(16)) != limg_hip_success) return r;
        HIP_TRY(hipMemsetAsync(c->devStatus.p, 0, 16, stream));
      }
      p.timeout = (uint32_t *)c->devStatus.p;
#ifdef LIMG_HIP_TEST_HOOKS
      p.lookbackSpins = c->topt.lookback_spins > 0 ? (uint32_t)c->topt.lookback_spins : (1u << 22);
      p.testSkipStrip = c->topt.skip_publish_strip > 0 ? (uint32_t)c->topt.skip_publish_strip - 1u : ~0u;
      p.testBaseErrStrip = c->topt.base_error_strip > 0 ? (uint32_t)c->topt.base_error_strip - 1u : ~0u;
#endif
      p.compactOut = compact != nullptr || wantStats; // the statistics are reduced from the raster-order shift words
      if ((r = c->park.ensure((size_t)(c->persistentWorkgroups / 5 * 6) * 2 * 8192)) != limg_hip_success) return r; // 6 workgroups per CU: the kernel's launch bound
      p.park = (uint8_t *)c->park.p;
    }
    // workgroups per CU of the persistent kernel: 6 once the float stage is out (7 fit and were measured: no faster, the kernel is issue-bound), 5 with it inside;
    // limg_hip_options.test_wg_per_cu lowers it (A/B runs), never above the launch bound the park slots are sized for
    auto wg_per_cu = [&](int deflt) { const int t = TOPT(c, wg_per_cu); return (t >= 1 && t < deflt) ? t : deflt; };

    // A list of images as a PIPELINE of launch pairs (limg_hip_options.batch_sub_images): k_fit_tpb of sub-batch k + 1 runs on a stream of the context's own next to
    // the persistent kernel of sub-batch k.  On content with short searches (BASELINE configs 2 / 4: ~2 trials per block) the persistent kernel waits for its plane
    // stores a third of the time while k_fit_tpb is pure vector work: side by side they fill each other's gaps.  The persistent kernel leaves the float stage room
    // to be resident: 5 workgroups per CU instead of 6 (at 6 x 80 VGPRs nothing else fits on a SIMD) for every sub-batch but the last.
    // Measured on 4096^2 random-gradient lists (profiles/archive/r04_pipeline_sweep.md): 64 images 66.5 -> 73.3 Gpixel/s in sub-batches of 8 (4: 70.9, 16: 71.9); 16 images
    // +3.5 % in sub-batches of 4; 8 images and fewer: nothing to gain (the lone first float stage and the 5-workgroup launches cost what the overlap saves).
    size_t subImages = 0;
    if (fused && batchCount > 1 && p.prefit)
    {
      if (c->opt.batch_sub_images > 0) subImages = (size_t)c->opt.batch_sub_images;
      else if (c->opt.batch_sub_images == 0) subImages = batchCount >= 32 ? 8 : (batchCount >= 16 ? 4 : 0);
    }
    if (subImages > 0 && subImages < batchCount)
    {
      // A/B hook (limg_hip_options.test_pipeline): bits 0..3 = 1 + k_fit_tpb's wave priority, bits 4..7 = workgroups per CU of the overlapped persistent launches,
      // bits 8..15 = images of the first sub-batch (whose float stage runs alone)
      const uint32_t knobs = (uint32_t)TOPT(c, pipeline);
      const int fitPrio = (knobs & 15u) ? (int)(knobs & 15u) - 1 : 0, wgOverlapRaw = ((knobs >> 4) & 15u) ? (int)((knobs >> 4) & 15u) : 5,
                wgOverlap = wgOverlapRaw > 6 ? 6 : wgOverlapRaw; // never above the launch bound the park slots are sized for
      const size_t firstSub = ((knobs >> 8) & 255u) && ((knobs >> 8) & 255u) < subImages ? (size_t)((knobs >> 8) & 255u) : subImages;
      const size_t nSub = 1 + (batchCount - firstSub + subImages - 1) / subImages;
      if ((r = ensure_pipe_events(c, 2 * nSub + 1)) != limg_hip_success) return r;
      const size_t imgBlocks = (size_t)p.blocksX * p.blocksY, imgStrips = (size_t)p.imageStrips;
      auto sub = [&](size_t k) -> EncodeParams
      {
        EncodeParams q = p;
        const size_t i0 = k == 0 ? 0 : firstSub + (k - 1) * subImages, want = k == 0 ? firstSub : subImages, n = batchCount - i0 < want ? batchCount - i0 : want;
        q.batch = p.batch + i0; q.batchCount = (uint32_t)n;
        q.io = batch[i0]; // (what the kernels read when a sub-batch is a single image)
        q.records = p.records + i0 * imgBlocks; q.shifts = p.shifts + i0 * imgBlocks; q.invN = p.invN + i0 * imgBlocks * 4;
        uint8_t *lb = (uint8_t *)c->lookback.p + k * 16 + i0 * imgStrips * 8; // sub-batch k: its ticket, then the descriptors of its strips
        q.ticket = (uint32_t *)lb; q.desc = (unsigned long long *)(lb + 16);
        q.fitPrio = k == 0 ? 0 : fitPrio;
        return q;
      };
      hipStream_t fs = c->fitStream;
      hipEvent_t *ev = c->pipeEvents.data(); // [0]: fork; [1 + 2 k]: k_fit_tpb of sub-batch k done; [2 + 2 k]: the persistent kernel of sub-batch k is next on `stream`
      mark_if(1);
      HIP_TRY(hipEventRecord(ev[0], stream)); // everything the caller's stream holds so far (inputs, the image table, earlier encodes that use the scratch)
      HIP_TRY(hipStreamWaitEvent(fs, ev[0], 0));
      launch_fit_tpb(sub(0), channels, fs);
      HIP_TRY(hipEventRecord(ev[1], fs));
      for (size_t k = 0; k < nSub; k++)
      {
        HIP_TRY(hipStreamWaitEvent(stream, ev[1 + 2 * k], 0));
        if (k == 0) mark_if(1); // (k_fit_tpb of the first sub-batch: the only float-stage work that runs alone)
        if (k + 1 < nSub)
        {
          HIP_TRY(hipEventRecord(ev[2 + 2 * k], stream));
          HIP_TRY(hipStreamWaitEvent(fs, ev[2 + 2 * k], 0));
          launch_fit_tpb(sub(k + 1), channels, fs);
          HIP_TRY(hipEventRecord(ev[3 + 2 * k], fs));
        }
        launch_encode_persistent(sub(k), channels, c->persistentWorkgroups / 5 * (k + 1 < nSub ? wgOverlap : wg_per_cu(6)), stream);
      }
      mark_if(1); mark_if(2);
      HIP_TRY(hipGetLastError());
      return stats();
    }

    if (p.prefit && chainPhase != 2)
    {
      mark_if(1);
      launch_fit_tpb(p, channels, stream);
      if (p.fitOnly)
      { // pass 1 of the merged-block encoder: the records are all it wants
        mark_if(1); mark_if(1); mark_if(2);
        HIP_TRY(hipGetLastError());
        return limg_hip_success;
      }
    }
    if (fused)
    {
      mark_if(1);
      launch_encode_persistent(p, channels, c->persistentWorkgroups / 5 * wg_per_cu(p.prefit ? 6 : 5), stream);
      mark_if(1);
      if (p.prefit) mark_if(2); else { mark_if(1); mark_if(2); } // 4 events per encode: with the float stage as its own launch the intervals are {k_fit_tpb, k_encode_persistent, -}
      HIP_TRY(hipGetLastError());
      return stats();
    }

    if (chainPhase == 2)
    { // the E step and the scan of this very image ran in phase 1: records, shift words, strip bases and the pre-dither factor bytes are where they left them
      mark(c, stream); // a chain encode records 4 events over its two phases: intervals {E step + scan, exchange between the phases, F step}
      launch_dither_store(p, channels, stream);
      mark(c, stream);
      HIP_TRY(hipGetLastError());
      return limg_hip_success;
    }
    // Images with a partial last block COLUMN (any photograph whose width is not a multiple of 8): the whole dither chain is data dependent, so the host walks it
    // (below).  The walk is the floor of this class -- ~26 ms for 8190 x 8192, one dependent AESDEC chain -- so everything else is taken off its path: the E step runs
    // in BANDS of block rows whose shift words come back band by band (the walk starts when the first band is down and runs under the rest of the E step), and the F
    // step of a band is launched as soon as its chain values are up (it runs under the walk of the next band).  Only for one chain (poolThreads == 0): independent
    // chains are walked in parallel instead.  A band is a sub-image: pointers advanced, block rows counted from its top.
    uint32_t nBands = 1;
    // (an explicit band count is honoured from 2 x 2 blocks on -- tests and the fuzz tool; a band must not be one block wide: the corner block of fewer than four
    //  pixels sums its LEFT neighbour's pixels, which for a one-block-wide image would be the row above, in another band)
    if (ragged && dInfo && chainPhase == 0 && !x.dPrevDesc && pt.chainCount <= 1 && c->opt.ragged_bands >= 0 && p.blocksX >= 2 && p.blocksY >= 2 &&
        (c->opt.ragged_bands > 0 || (p.blocksX >= 32 && p.blocksY >= 64)))
      nBands = c->opt.ragged_bands > 0 ? (uint32_t)c->opt.ragged_bands : 16u;
    if (nBands > p.blocksY / 2) nBands = p.blocksY / 2 ? p.blocksY / 2 : 1;
    if (nBands > 64) nBands = 64;
    const uint32_t bandRows = (p.blocksY + nBands - 1) / nBands;
    nBands = (p.blocksY + bandRows - 1) / bandRows;
    auto band_params = [&](uint32_t b) -> EncodeParams
    {
      EncodeParams q = p;
      const uint32_t r0 = b * bandRows, r1 = r0 + bandRows < p.blocksY ? r0 + bandRows : p.blocksY;
      const size_t y0 = (size_t)r0 * kBlock, skip = y0 * sizeX;
      q.io.in = p.io.in + skip;
      uint32_t **words[] = { &q.io.info.pDecoded, &q.io.info.pShiftABCX, &q.io.info.pColAMin, &q.io.info.pColAMax, &q.io.info.pColBMin, &q.io.info.pColBMax, &q.io.info.pColCMin, &q.io.info.pColCMax };
      uint8_t **bytes[] = { &q.io.info.pFactorsA, &q.io.info.pFactorsB, &q.io.info.pFactorsC };
      for (uint32_t **w : words) if (*w) *w += skip;
      for (uint8_t **w : bytes) if (*w) *w += skip;
      q.sizeY = (uint32_t)(((size_t)r1 * kBlock < sizeY ? (size_t)r1 * kBlock : sizeY) - y0);
      q.blocksY = r1 - r0;
      q.imageStrips = q.stripsX * q.blocksY;
      q.records += (size_t)r0 * p.blocksX; q.shifts += (size_t)r0 * p.blocksX; q.invN += (size_t)r0 * p.blocksX * 4;
      q.stripCalls += (size_t)r0 * p.stripsX; q.stripBase += (size_t)r0 * p.stripsX;
      return q;
    };
    if (!p.prefit) mark_if(1); // split path intervals: {k_fit_tpb + k_fit_search, scan, k_dither_store}
    if (nBands > 1)
    {
      if (c->hStageBusy) { HIP_TRY(hipEventSynchronize(c->hStageEvent)); c->hStageBusy = false; } // (see below: the staging area is about to be rewritten)
      if ((r = c->hStage.ensure(ragged_stage_bytes(blocks, strips))) != limg_hip_success) return r;
      while (c->raggedEvents.size() < nBands)
      {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->raggedEvents.push_back(e);
      }
      for (uint32_t b = 0; b < nBands; b++)
      {
        const EncodeParams q = band_params(b);
        launch_fit_search(q, channels, stream);
        const size_t off = (size_t)b * bandRows * p.blocksX;
        HIP_TRY(hipMemcpyAsync((uint32_t *)c->hStage.p + off, p.shifts + off, (size_t)q.blocksY * p.blocksX * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(c->raggedEvents[b], stream));
      }
    }
    else launch_fit_search(p, channels, stream);
    if (chainPhase != 1) mark_if(1);
    if (!dInfo)
    {
      mark_if(1); mark_if(2);
      HIP_TRY(hipGetLastError());
      return limg_hip_success; // `_perf` behaviour: nothing to dither into, nothing to store
    }

    if (!ragged)
    {
      launch_strip_scan(p, stream);
      if (chainPhase == 1)
      {
        mark(c, stream);
        HIP_TRY(hipGetLastError());
        return limg_hip_success;
      }
    }
    else
    {
      // Partial edge blocks: the chain walk depends on each block's pixel count (a call over N pixels is N / 8 AES rounds + N % 8 PCG steps: G_N), so it is
      // evaluated in raster order on the host from the per-block call counts.  What crosses PCIe: the shift words down, then -- through pinned staging -- the
      // strips' first call indices and, per dither call, the chain value it starts from and its pixel count (9 bytes; k_noise_expand turns them into the call's
      // 64 noise bytes on the device; rounds 1-3 uploaded the 64 bytes).
      const size_t maxCalls = blocks * 3;
      const size_t offPrev = (blocks * 4 + 15) & ~(size_t)15, offBase = offPrev + 16, offStates = (offBase + strips * 4 + 15) & ~(size_t)15, offPixels = offStates + maxCalls * 8;
      if (nBands == 1)
      {
        // the previous ragged encode's H2D copies out of this buffer were asynchronous -- possibly on another stream: they must have read it before it is rewritten
        // or reallocated
        if (c->hStageBusy) { HIP_TRY(hipEventSynchronize(c->hStageEvent)); c->hStageBusy = false; }
        if ((r = c->hStage.ensure(ragged_stage_bytes(blocks, strips))) != limg_hip_success) return r;
      }
      uint32_t *hShifts = (uint32_t *)c->hStage.p;
      unsigned long long *hPrev = (unsigned long long *)((uint8_t *)c->hStage.p + offPrev);
      uint32_t *hBase = (uint32_t *)((uint8_t *)c->hStage.p + offBase);
      unsigned long long *hStates = (unsigned long long *)((uint8_t *)c->hStage.p + offStates);
      uint8_t *hPixels = (uint8_t *)c->hStage.p + offPixels;
      // device side: per dither call 64 noise bytes + the 9 bytes they are expanded from.  A banded encode sends a band's calls up while later bands are still being
      // walked, so it sizes for the worst case up front (3 calls per block); everything else knows its call count before anything goes up (the shift words are down)
      // and sizes for that -- a 32766 x 32768 image would otherwise hold 3.6 GB of the GPU for nothing
      unsigned long long *dStates = nullptr;
      uint8_t *dPixels = nullptr;
      auto size_device_side = [&](size_t calls) -> limg_hip_result
      {
        limg_hip_result rr;
        if ((rr = c->noiseDyn.ensure((calls + 1) * 64)) != limg_hip_success) return rr;
        if ((rr = c->noiseStates.ensure(calls * 9 + 16)) != limg_hip_success) return rr;
        dStates = (unsigned long long *)c->noiseStates.p;
        dPixels = (uint8_t *)c->noiseStates.p + calls * 8;
        p.noise = (const uint8_t *)c->noiseDyn.p;
        p.noiseLast = (uint32_t)calls; // (entry `calls` exists: the clamp of a call index cannot land outside the buffer)
        return limg_hip_success;
      };
      if (nBands > 1 && (r = size_device_side(maxCalls)) != limg_hip_success) return r;
      const bool pcg = c->opt.dither_pcg != 0;
      auto upload_calls = [&](size_t call0, size_t n) -> limg_hip_result
      {
        if (!n) return limg_hip_success;
        HIP_TRY(hipMemcpyAsync(dStates + call0, hStates + call0, n * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(dPixels + call0, hPixels + call0, n, hipMemcpyHostToDevice, stream));
        launch_noise_expand((uint8_t *)c->noiseDyn.p + call0 * 64, dStates + call0, dPixels + call0, n, pcg, stream);
        return limg_hip_success;
      };
      if (nBands > 1)
      { // ---- one chain, walked band by band under the E step; every band's F step under the walk of the next ----
        uint64_t h = kDitherSeed;
        size_t call = 0;
        for (uint32_t b = 0; b < nBands; b++)
        {
          EncodeParams q = band_params(b);
          const uint32_t r0 = b * bandRows, r1 = r0 + q.blocksY;
          HIP_TRY(hipEventSynchronize(c->raggedEvents[b])); // this band's shift words are down
          const size_t call0 = call;
          chain_walk_rows(h, call, r0, r1, p.blocksX, p.stripsX, sizeX, sizeY, 1, 0, hShifts, hBase, hStates, hPixels, maxCalls, pcg);
          if ((r = upload_calls(call0, call - call0)) != limg_hip_success) return r;
          HIP_TRY(hipMemcpyAsync(p.stripBase + (size_t)r0 * p.stripsX, hBase + (size_t)r0 * p.stripsX, (size_t)q.blocksY * p.stripsX * 4, hipMemcpyHostToDevice, stream));
          q.noise = p.noise; q.noiseLast = p.noiseLast;
          if (b + 1 == nBands) mark_if(1); // intervals of a banded encode: {E step of all bands, the walk with the other bands' F steps under it, the last band's F step}
          launch_dither_store(q, channels, stream);
        }
        if (!c->hStageEvent) HIP_TRY(hipEventCreateWithFlags(&c->hStageEvent, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->hStageEvent, stream));
        c->hStageBusy = true;
        mark_if(2);
        HIP_TRY(hipGetLastError());
        return stats();
      }
      HIP_TRY(hipMemcpyAsync(hShifts, p.shifts, blocks * 4, hipMemcpyDeviceToHost, stream));
      if (x.dPrevDesc) HIP_TRY(hipMemcpyAsync(hPrev, x.dPrevDesc, 8, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      {
        size_t sumCalls = 0;
        for (size_t i = 0; i < blocks; i++) sumCalls += hShifts[i] >> 24;
        if ((r = size_device_side(sumCalls < maxCalls ? sumCalls : maxCalls)) != limg_hip_success) return r;
      }
      uint64_t h0 = kDitherSeed;
      if (x.dPrevDesc)
      { // the block rows above this sub-image ran through the persistent kernel: every strip's descriptor ends as the inclusive call count of its chain
        if ((uint32_t)(*hPrev >> 32) != 2u || (uint32_t)*hPrev == 0xFFFFFFFFu || !chain_value_at((uint32_t)*hPrev, &h0))
        {
          fprintf(stderr, "limg_hip: the chain position of the last block row is unavailable (descriptor %016llx)\n", *hPrev);
          return limg_hip_error_Generic;
        }
      }
      size_t totalCalls = 0;
      // chains that restart at the seed are independent (the reference walks them on its pool's threads, src/limg.cpp:2114-2134): count every chain's calls, then
      // walk them side by side on up to limg_hip_options.ragged_walk_threads host threads (0: as many as there are chains, at most 16)
      unsigned threads = 1;
      if (pt.chainCount > 1 && pt.chainRows != 0 && !x.dPrevDesc && (blocks >= 4096 || c->opt.ragged_walk_threads > 1))
      {
        threads = c->opt.ragged_walk_threads > 0 ? (unsigned)c->opt.ragged_walk_threads : 16u;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && threads > hw) threads = hw;
        if (threads > pt.chainCount) threads = pt.chainCount;
      }
      if (threads > 1)
      {
        size_t *first = new (std::nothrow) size_t[pt.chainCount + 1];
        uint32_t *row0 = new (std::nothrow) uint32_t[pt.chainCount + 1];
        if (!first || !row0) { delete[] first; delete[] row0; return limg_hip_error_MemoryAllocationFailure; }
        first[0] = 0;
        for (uint32_t k = 0; k < pt.chainCount; k++)
        {
          row0[k] = k * pt.chainRows;
          const uint32_t r1 = k + 1 < pt.chainCount ? (k + 1) * pt.chainRows : p.blocksY;
          size_t n = 0;
          for (size_t i = (size_t)row0[k] * p.blocksX; i < (size_t)r1 * p.blocksX; i++) n += hShifts[i] >> 24;
          first[k + 1] = first[k] + n;
        }
        row0[pt.chainCount] = p.blocksY;
        run_on_threads(threads, [&](unsigned t) {
          for (uint32_t k = t; k < pt.chainCount; k += threads)
          {
            uint64_t h = kDitherSeed;
            size_t call = first[k];
            chain_walk_rows(h, call, row0[k], row0[k + 1], p.blocksX, p.stripsX, sizeX, sizeY, pt.chainCount, pt.chainRows, hShifts, hBase, hStates, hPixels, maxCalls, pcg);
          }
        });
        const size_t allCalls = first[pt.chainCount];
        delete[] first; delete[] row0;
        totalCalls = allCalls < maxCalls ? allCalls : maxCalls;
      }
      else totalCalls = chain_walk_blocks(h0, p.blocksX, p.blocksY, p.stripsX, sizeX, sizeY, pt.chainCount, pt.chainRows, hShifts, hBase, hStates, hPixels, maxCalls, pcg);
      if ((r = upload_calls(0, totalCalls)) != limg_hip_success) return r;
      HIP_TRY(hipMemcpyAsync(p.stripBase, hBase, strips * 4, hipMemcpyHostToDevice, stream));
      // the staging buffer is the context's: the event marks the point where these copies have read it (waited for above by the next encode that uses it, on
      // whatever stream, and by limg_hip_shutdown)
      if (!c->hStageEvent) HIP_TRY(hipEventCreateWithFlags(&c->hStageEvent, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(c->hStageEvent, stream));
      c->hStageBusy = true;
    }
    mark_if(1);
    launch_dither_store(p, channels, stream);
    mark_if(2);
    HIP_TRY(hipGetLastError());
    return stats();
  }

  // sizeX % 8 == 0, sizeY % 8 != 0, at least two block rows (reference: the rx x ry gather of src/limg.cpp:1899-1905 only ever sees ry < 8 in the last block row,
  // and a dither call over 8 ry pixels is ry AES rounds, :824-879): the block rows above the last one are an image of whole blocks -- k_fit_tpb + persistent kernel
  // with the chain partition of the WHOLE image -- and the last row goes through the split path's kernels as a one-row sub-image whose chain starts where the
  // persistent kernel's last strip left it: that call count is the low word of the strip's look-back descriptor, and the chain value there comes from the embedded
  // checkpoints (at most 1023 calls on foot).  The host's share is then one block row: blocksX shift words down, <= 3 blocksX calls of ry rounds, their noise up.
  limg_hip_result encode_height_ragged(limg_hip_context *c, const uint32_t *dIn, size_t sizeX, size_t sizeY, int hasAlpha, const limg_hip_encode3d_info *dInfo,
                                       const limg_hip_compact_out *compact, uint32_t errorFactor, int poolThreads, int fast, hipStream_t stream, const EncodeExtra &x)
  {
    const size_t topY = (sizeY / kBlock) * kBlock, blocksX = sizeX / kBlock, blocksY = topY / kBlock + 1, stripsX = (blocksX + kStripBlocks - 1) / kStripBlocks;
    const Partition pt = partition(sizeY, poolThreads);
    c->chainIn = nullptr;
    c->statsState = 0;
    limg_hip_result r;
    EncodeExtra xt;
    xt.streamRaw = x.streamRaw; xt.inner = true; xt.marks = 1; xt.part = &pt; xt.scratchRows = blocksY;
    if ((r = encode_device(c, dIn, sizeX, topY, hasAlpha, dInfo, compact, errorFactor, poolThreads, fast, stream, xt)) != limg_hip_success) return r;
    // the last block row
    limg_hip_encode3d_info low = *dInfo;
    {
      const size_t skip = topY * sizeX; // pixels above the last block row, in every plane
      uint32_t **words[] = { &low.pDecoded, &low.pShiftABCX, &low.pColAMin, &low.pColAMax, &low.pColBMin, &low.pColBMax, &low.pColCMin, &low.pColCMax };
      uint8_t **bytes[] = { &low.pFactorsA, &low.pFactorsB, &low.pFactorsC };
      for (uint32_t **p : words) if (*p) *p += skip;
      for (uint8_t **p : bytes) if (*p) *p += skip;
    }
    EncodeExtra xb;
    xb.streamRaw = x.streamRaw; xb.inner = true; xb.marks = 0; xb.scratchRow0 = blocksY - 1; xb.scratchRows = blocksY;
    const Partition one = { 1, 0 };
    xb.part = &one;
    const bool sameChain = chain_of_row(pt, (uint32_t)blocksY - 1) == chain_of_row(pt, (uint32_t)blocksY - 2);
    if (sameChain) xb.dPrevDesc = (const unsigned long long *)((const uint8_t *)c->lookback.p + 16) + ((blocksY - 1) * stripsX - 1);
    if ((r = encode_device(c, dIn + topY * sizeX, sizeX, sizeY - topY, hasAlpha, &low, compact, errorFactor, 0, fast, stream, xb)) != limg_hip_success) return r;
    mark(c, stream);
    if (c->opt.collect_stats != 0)
    {
      if ((r = c->stats.ensure(30 * 8)) != limg_hip_success) return r;
      HIP_TRY(hipMemsetAsync(c->stats.p, 0, 30 * 8, stream));
      const uint32_t *shifts = (compact && compact->pShifts) ? compact->pShifts : (const uint32_t *)c->shifts.p;
      launch_shift_stats(shifts, (uint32_t)blocksX, (uint32_t)blocksY, (uint32_t)blocksY, (uint32_t)sizeX, (uint32_t)sizeY, (unsigned long long *)c->stats.p, stream);
      c->statsStream = stream; c->statsState = 1; c->statsPixels = (uint64_t)sizeX * sizeY;
    }
    return limg_hip_success;
  }
}


extern "C"
{
  const char *limg_hip_version(void) { return "limg_hip 0.1 (gfx950)"; }

  // limg_hip_options is versioned by its size (limg_hip.h): a caller compiled against an earlier, shorter header hands over fewer bytes
  void limg_hip_default_options_sized(limg_hip_options *o, size_t structSize)
  {
    if (!o || structSize < sizeof(uint32_t)) return;
    limg_hip_options d;
    memset(&d, 0, sizeof(d));
    d.forced_shift[0] = d.forced_shift[1] = d.forced_shift[2] = -1;
    const size_t n = structSize < sizeof(d) ? structSize : sizeof(d);
    d.struct_size = (uint32_t)n;
    memcpy(o, &d, n);
  }

#ifdef LIMG_HIP_TEST_HOOKS
  void limg_hip_default_test_options_sized(limg_hip_test_options *o, size_t structSize)
  {
    if (!o || structSize < sizeof(uint32_t)) return;
    limg_hip_test_options d;
    memset(&d, 0, sizeof(d));
    const size_t n = structSize < sizeof(d) ? structSize : sizeof(d);
    d.struct_size = (uint32_t)n;
    memcpy(o, &d, n);
  }

  limg_hip_result limg_hip_set_test_options(limg_hip_context *c, const limg_hip_test_options *o)
  {
    if (!c || !o) return limg_hip_error_ArgumentNull;
    const size_t n = o->struct_size;
    if (n < sizeof(uint32_t) || (n & 3u) != 0) return limg_hip_error_InvalidParameter;
    limg_hip_test_options full;
    limg_hip_default_test_options_sized(&full, sizeof(full));
    memcpy(&full, o, n < sizeof(full) ? n : sizeof(full));
    full.struct_size = (uint32_t)sizeof(full);
    c->topt = full;
    return limg_hip_success;
  }
#endif

  limg_hip_result limg_hip_init(int device, limg_hip_context **ppCtx)
  {
    if (!ppCtx) return limg_hip_error_ArgumentNull;
    *ppCtx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    {
      fprintf(stderr, "limg_hip: no HIP device available -- this library has no CPU fallback\n");
      return limg_hip_error_Generic;
    }
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= count) return limg_hip_error_InvalidParameter;
    HIP_TRY(hipSetDevice(device));
    limg_hip_context *c = new (std::nothrow) limg_hip_context();
    if (!c) return limg_hip_error_MemoryAllocationFailure;
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->persistentWorkgroups = 5 * prop.multiProcessorCount;
    limg_hip_default_options_sized(&c->opt, sizeof(c->opt));
#ifdef LIMG_HIP_TEST_HOOKS
    limg_hip_default_test_options_sized(&c->topt, sizeof(c->topt));
#endif
    *ppCtx = c;
    return limg_hip_success;
  }

  void limg_hip_shutdown(limg_hip_context **ppCtx)
  {
    if (!ppCtx || !*ppCtx) return;
    limg_hip_context *c = *ppCtx;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    DevBuf *bufs[] = { &c->noiseStates, &c->invN, &c->records, &c->shifts, &c->stripCalls, &c->stripBase, &c->noise, &c->noiseDyn, &c->noiseCk, &c->stats, &c->lookback, &c->devStatus, &c->accTable, &c->commWords, &c->park, &c->batchTable, &c->in, &c->planes, &c->cmp,
                       &c->streamFac, &c->streamTiles, &c->streamUnits, &c->streamStatus, &c->streamBuf,
                       &c->bFlags, &c->bBound, &c->bMatch, &c->bRegions, &c->bOut, &c->bPx, &c->bFac, &c->bNoise, &c->bNoiseBase, &c->bOrder };
    for (DevBuf *b : bufs) b->release();
    HostBuf *hbufs[] = { &c->hFlags, &c->hRec, &c->hBits, &c->hDesc, &c->hOut, &c->hNoise, &c->hNoiseBase };
    for (HostBuf *b : hbufs) b->release();
    c->hStage.release(); // (the device is idle: hipDeviceSynchronize above)
    if (c->hStageEvent) (void)hipEventDestroy(c->hStageEvent);
    for (hipEvent_t e : c->raggedEvents) (void)hipEventDestroy(e);
    if (c->fitStream) (void)hipStreamDestroy(c->fitStream);
    if (c->hostStream) (void)hipStreamDestroy(c->hostStream);
    if (c->hostCopyStream) (void)hipStreamDestroy(c->hostCopyStream);
    for (hipEvent_t e : c->hostEvents) (void)hipEventDestroy(e);
    c->hostWords.release();
    for (hipEvent_t e : c->pipeEvents) (void)hipEventDestroy(e);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    if (c->workStream) (void)hipStreamDestroy(c->workStream);
    if (c->storeStream) (void)hipStreamDestroy(c->storeStream);
    for (hipStream_t st : c->workStreams) (void)hipStreamDestroy(st);
    c->bCalls.release();
    for (hipEvent_t e : c->workEvents) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->workTimers) (void)hipEventDestroy(e);
    if (c->copyStream) (void)hipStreamDestroy(c->copyStream);
    for (hipEvent_t e : c->bandEvents) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    delete c;
    *ppCtx = nullptr;
  }

  limg_hip_result limg_hip_set_options(limg_hip_context *c, const limg_hip_options *o)
  {
    if (!c || !o) return limg_hip_error_ArgumentNull;
    const size_t n = o->struct_size;
    if (n < offsetof(limg_hip_options, forced_shift) + sizeof(o->forced_shift) || (n & 3u) != 0) return limg_hip_error_InvalidParameter;
    limg_hip_options full;
    limg_hip_default_options_sized(&full, sizeof(full)); // members the caller's header did not have keep their defaults
    memcpy(&full, o, n < sizeof(full) ? n : sizeof(full));
    full.struct_size = (uint32_t)sizeof(full);
    c->opt = full;
    c->forceSplit = full.force_split_kernels != 0;
    return limg_hip_success;
  }

  limg_hip_result limg_hip_get_options(const limg_hip_context *c, limg_hip_options *o)
  {
    if (!c || !o) return limg_hip_error_ArgumentNull;
    size_t n = o->struct_size; // the room the caller has
    if (n < sizeof(uint32_t) || (n & 3u) != 0) return limg_hip_error_InvalidParameter;
    if (n > sizeof(c->opt)) n = sizeof(c->opt);
    memcpy(o, &c->opt, n);
    o->struct_size = (uint32_t)n;
    return limg_hip_success;
  }

  // Blocks until the device is idle and reports a look-back timeout of the persistent kernel as limg_hip_error_Generic.  Observed once, in round 4: a build whose
  // workgroups took their first strip from blockIdx instead of the ticket dead-locked two contexts' kernels against each other until the bound fired;
  // with every strip id drawn from the ticket a look-back cannot wait on a workgroup that is not running.  The bound stays as the safety net, and a timeout is loud:
  // the strips behind it store nothing chain-dependent (limg_hip_kernels.hip "decoupled look-back").
  limg_hip_result limg_hip_check_device_status(limg_hip_context *c)
  {
    if (!c) return limg_hip_error_ArgumentNull;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    if (c->devStatus.p)
    {
      uint32_t word = 0;
      HIP_TRY(hipMemcpy(&word, c->devStatus.p, 4, hipMemcpyDeviceToHost));
      if (word != 0)
      {
        HIP_TRY(hipMemset(c->devStatus.p, 0, 4)); // sticky until reported once
        fprintf(stderr, "limg_hip: look-back timeout in the fused encode kernel\n");
        return limg_hip_error_Generic;
      }
      HIP_TRY(hipMemcpy(&word, (const uint8_t *)c->devStatus.p + 4, 4, hipMemcpyDeviceToHost));
      if (word != 0)
      {
        HIP_TRY(hipMemset((uint8_t *)c->devStatus.p + 4, 0, 4));
        fprintf(stderr, "limg_hip: a rank of the communicator aborted a single-chain encode; this rank's planes of that encode were not written\n");
        return limg_hip_error_Generic;
      }
    }
    if (c->streamStatus.p)
    {
      uint32_t word = 0;
      HIP_TRY(hipMemcpy(&word, c->streamStatus.p, 4, hipMemcpyDeviceToHost));
      if (word != 0)
      {
        HIP_TRY(hipMemset(c->streamStatus.p, 0, 4));
        fprintf(stderr, "limg_hip: stream refused by the decode kernel (%s)\n", (word & 1u) ? "header mismatch" : "inconsistent payload offsets");
        return limg_hip_error_InvalidParameter;
      }
    }
    return limg_hip_success;
  }

  limg_hip_result limg_hip_last_stats(limg_hip_context *c, uint64_t *pCounters30, uint64_t *pPixels)
  {
    if (!c || !pCounters30) return limg_hip_error_ArgumentNull;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry); // (the host-pointer entries write this state under the same lock)
    if (c->statsState == 0) return limg_hip_error_InvalidParameter; // no encode with limg_hip_options.collect_stats since the last reset
    if (c->statsState == 1)
    {
      HIP_TRY(hipSetDevice(c->device));
      HIP_TRY(hipMemcpyAsync(c->statsHost, c->stats.p, 30 * 8, hipMemcpyDeviceToHost, c->statsStream));
      HIP_TRY(hipStreamSynchronize(c->statsStream));
      c->statsState = 2;
    }
    memcpy(pCounters30, c->statsHost, 30 * 8);
    if (pPixels) *pPixels = c->statsPixels;
    return limg_hip_success;
  }

  limg_hip_result limg_hip_profile_begin(limg_hip_context *c)
  {
    if (!c) return limg_hip_error_ArgumentNull;
    c->profiling = true;
    c->eventsUsed = 0;
    return limg_hip_success;
  }

  // Stops profiling, waits for the recorded work and writes, per profiled encode, {fit_search, scan (+ host chain walk for ragged
  // images), dither_store} milliseconds into pMs[3 * i ...].  Returns the number of encodes written (<= maxEncodes), or -1.
  int limg_hip_profile_end(limg_hip_context *c, float *pMs, int maxEncodes)
  {
    if (!c || !pMs) return -1;
    c->profiling = false;
    const int n = (int)(c->eventsUsed / 4);
    int written = 0;
    for (int i = 0; i < n && i < maxEncodes; i++, written++)
    {
      if (hipEventSynchronize(c->events[4 * i + 3]) != hipSuccess) return -1;
      for (int k = 0; k < 3; k++)
        if (hipEventElapsedTime(&pMs[3 * i + k], c->events[4 * i + k], c->events[4 * i + k + 1]) != hipSuccess) return -1;
    }
    c->eventsUsed = 0;
    return written;
  }

  // ---- host-only helpers (no GPU needed; exposed so the host logic can be tested on CPU-only machines) --------------------
  limg_hip_result limg_hip_noise_table_device(limg_hip_context *c, uint8_t *pOutDevice, size_t calls, void *stream)
  {
    if (!c || !pOutDevice) return limg_hip_error_ArgumentNull;
    if (calls > checkpoint_reach() || ((uintptr_t)pOutDevice & 15u) != 0) return limg_hip_error_InvalidParameter;
    HIP_TRY(hipSetDevice(c->device));
    limg_hip_result r;
    if ((r = ensure_checkpoints(c, calls)) != limg_hip_success) return r;
    launch_noise_fill(pOutDevice, (const uint64_t *)c->noiseCk.p, calls, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return limg_hip_success;
  }

  limg_hip_result limg_hip_host_noise_table(uint8_t *pOut, size_t calls)
  {
    if (!pOut) return limg_hip_error_ArgumentNull;
    fill_noise_table(kDitherSeed, pOut, calls, false);
    return limg_hip_success;
  }

  uint64_t limg_hip_host_chain_call(uint64_t chainValue, size_t pixelCount, uint8_t *pNoise64, int forceSoftwareAes)
  {
    if (pixelCount > 0xFFFFFFFFull) return 0;
    return chain_call(chainValue, (unsigned)pixelCount, pNoise64, (forceSoftwareAes & 1) != 0, (forceSoftwareAes & 2) != 0);
  }

  uint64_t limg_hip_host_chain_checkpoints(size_t calls, size_t every, uint64_t *pOut, int pcg)
  {
    return chain_checkpoints(kDitherSeed, calls, every, pOut, pcg != 0);
  }

  limg_hip_result limg_hip_host_dense_checkpoints(size_t first, size_t count, uint64_t *pOut)
  {
    if (!pOut) return limg_hip_error_ArgumentNull;
    try { return dense_checkpoints_host(first, count, pOut) ? limg_hip_success : limg_hip_error_OutOfBounds; }
    catch (...) { return limg_hip_error_MemoryAllocationFailure; }
  }

  limg_hip_result limg_hip_host_partition(size_t sizeY, int poolThreads, uint32_t *pChainCount, uint32_t *pChainBlockRows)
  {
    if (!pChainCount || !pChainBlockRows) return limg_hip_error_ArgumentNull;
    const Partition pt = partition(sizeY, poolThreads);
    *pChainCount = pt.chainCount;
    *pChainBlockRows = pt.chainRows;
    return limg_hip_success;
  }

  size_t limg_hip_context_device_bytes(const limg_hip_context *c)
  {
    if (!c) return 0;
    const DevBuf *bufs[] = { &c->bCalls, &c->noiseStates, &c->records, &c->shifts, &c->invN, &c->stripCalls, &c->stripBase, &c->noise, &c->noiseDyn, &c->noiseCk, &c->lookback, &c->park, &c->batchTable, &c->stats,
                             &c->accTable, &c->devStatus, &c->commWords, &c->in, &c->planes, &c->cmp, &c->streamFac, &c->streamTiles, &c->streamUnits, &c->streamStatus, &c->streamBuf,
                             &c->bFlags, &c->bBound, &c->bMatch, &c->bRegions, &c->bOut, &c->bPx, &c->bFac, &c->bNoise, &c->bNoiseBase, &c->bOrder };
    size_t sum = 0;
    for (const DevBuf *b : bufs) sum += b->cap;
    return sum;
  }

  limg_hip_result limg_hip_encode3d_device(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, const limg_hip_encode3d_info *pInfo,
                                           const limg_hip_compact_out *pCompact, uint32_t errorFactor, int poolThreads, int fastBitCrushing, void *stream)
  {
    return encode_device(c, pIn, sizeX, sizeY, hasAlpha, pInfo, pCompact, errorFactor, poolThreads, fastBitCrushing, (hipStream_t)stream);
  }

  // The reference encodes a list of images by calling limg_encode3d_test(_perf) once per image (src/main.cpp:278-323).  Here the images of one shape go through ONE
  // launch pair: one k_fit_tpb grid over the blocks of all images and one persistent launch whose tickets run through the strips of image 0, image 1, ...;
  // every image starts its own dither chain(s), so each image's planes are those of a single encode.  A small image alone cannot fill the chip for long
  // (a 4096^2 image is ~5 strips per workgroup: ramp-up and drain are a third of its encode); a batch amortises both.
  limg_hip_result limg_hip_encode3d_batch_device(limg_hip_context *c, size_t count, const uint32_t *const *ppIn, size_t sizeX, size_t sizeY, int hasAlpha,
                                                 const limg_hip_encode3d_info *pInfos, uint32_t errorFactor, int poolThreads, int fastBitCrushing, void *stream)
  {
    if (!c || !ppIn || !pInfos) return limg_hip_error_ArgumentNull;
    if (count == 0) return limg_hip_success;
    if (sizeX == 0 || sizeY == 0 || sizeX > 0x7FFFFFF8ull || sizeY > 0x7FFFFFF8ull) return limg_hip_error_InvalidParameter;
    bool full = true;
    for (size_t i = 0; i < count; i++)
    {
      if (!ppIn[i]) return limg_hip_error_ArgumentNull;
      const void *const *pp = reinterpret_cast<const void *const *>(&pInfos[i]);
      for (int k = 8; k < 11; k++)
        if (!pp[k]) return limg_hip_error_ArgumentNull;
      int n32 = 0;
      for (int k = 0; k < 8; k++) n32 += pp[k] != nullptr;
      if (n32 != 0 && n32 != 8) return limg_hip_error_ArgumentNull;
      full = full && n32 == 8;
    }
    const bool ragged = (sizeX % kBlock) != 0 || (sizeY % kBlock) != 0;
    const size_t blocks = ((sizeX + kBlock - 1) / kBlock) * ((sizeY + kBlock - 1) / kBlock), strips = ((sizeX + kBlock * kStripBlocks - 1) / (kBlock * kStripBlocks)) * ((sizeY + kBlock - 1) / kBlock);
    // per-block scratch of a launch pair is bounded (1 GiB of records): longer lists go in several launch pairs
    size_t chunk = (size_t)(1ull << 30) / (blocks * sizeof(limg_hip_block_record));
    if (chunk * strips > 0x7FFFFFFFull) chunk = 0x7FFFFFFFull / strips; // strip ids are 32 bits
    if (chunk < 1) chunk = 1;
    if (TOPT(c, batch_chunk) > 0) chunk = (size_t)TOPT(c, batch_chunk);
    const bool oneByOne = count == 1 || ragged || !full || c->opt.legacy_float_stage != 0 || c->forceSplit;
    std::vector<ImageIO> table;
    limg_hip_result r = limg_hip_success;
    for (size_t i0 = 0; i0 < count && r == limg_hip_success;)
    {
      const size_t n = oneByOne ? 1 : (count - i0 < chunk ? count - i0 : chunk);
      c->statsAccumulate = i0 != 0; // limg_hip_last_stats: all images of the list together, however many launch pairs it took
      if (n == 1) r = encode_device(c, ppIn[i0], sizeX, sizeY, hasAlpha, &pInfos[i0], nullptr, errorFactor, poolThreads, fastBitCrushing, (hipStream_t)stream);
      else
      {
        table.resize(n);
        for (size_t i = 0; i < n; i++) { table[i].in = ppIn[i0 + i]; table[i].info = pInfos[i0 + i]; }
        EncodeExtra x;
        x.batch = table.data(); x.batchCount = n;
        r = encode_device(c, ppIn[i0], sizeX, sizeY, hasAlpha, &pInfos[i0], nullptr, errorFactor, poolThreads, fastBitCrushing, (hipStream_t)stream, x);
      }
      i0 += n;
    }
    c->statsAccumulate = false;
    return r;
  }

  limg_hip_result limg_hip_encode3d_chain_device(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, const limg_hip_encode3d_info *pInfo,
                                                 uint32_t errorFactor, int fastBitCrushing, int phase, uint64_t *pCallsDevice, const uint64_t *pChainBaseDevice,
                                                 size_t blocksBefore, void *stream)
  {
    if (phase != 1 && phase != 2) return limg_hip_error_InvalidParameter;
    if (!c || (phase == 1 && !pCallsDevice) || (phase == 2 && !pChainBaseDevice) || !pInfo) return limg_hip_error_ArgumentNull;
    // phase 2 continues the phase 1 the context holds: same strip, same planes (phase 1 left the pre-dither factor bytes in them), same parameters, and a noise
    // table that was sized for the same chain position
    if (phase == 2 && (c->chainIn != pIn || c->chainX != sizeX || c->chainY != sizeY || c->chainFac[0] != pInfo->pFactorsA || c->chainFac[1] != pInfo->pFactorsB ||
                       c->chainFac[2] != pInfo->pFactorsC || c->chainAlpha != (hasAlpha != 0) || c->chainEf != errorFactor || c->chainFast != (fastBitCrushing != 0) ||
                       c->chainBefore != blocksBefore))
      return limg_hip_error_InvalidParameter; // no matching phase 1 pending
    c->chainIn = nullptr;
    EncodeExtra x;
    x.chainPhase = phase; x.dChainCalls = (unsigned long long *)pCallsDevice; x.dChainBase = (const unsigned long long *)pChainBaseDevice; x.chainBlocksBefore = blocksBefore;
    const limg_hip_result r = encode_device(c, pIn, sizeX, sizeY, hasAlpha, pInfo, nullptr, errorFactor, 0, fastBitCrushing, (hipStream_t)stream, x);
    if (r == limg_hip_success && phase == 1)
    {
      c->chainIn = pIn; c->chainX = sizeX; c->chainY = sizeY; c->chainBefore = blocksBefore;
      c->chainFac[0] = pInfo->pFactorsA; c->chainFac[1] = pInfo->pFactorsB; c->chainFac[2] = pInfo->pFactorsC;
      c->chainAlpha = hasAlpha != 0; c->chainEf = errorFactor; c->chainFast = fastBitCrushing != 0;
    }
    return r;
  }

  // The reference's limg_encode3d_test accumulates its bit statistics on its own stack (`accum_bits`, src/limg.cpp:1975-1976) and prints them before it returns
  // (:2232-2248): they are always the statistics of THAT call, whatever other threads encode meanwhile.  Same here: the counters are fetched inside the region the
  // context's mutex covers, and collect_stats is switched on for this call only -- no option of a shared context is modified for good.
  limg_hip_result limg_hip_encode3d_stats(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, limg_hip_encode3d_info *pInfo, uint32_t errorFactor,
                                          int poolThreads, int fastBitCrushing, uint64_t *pCounters30, uint64_t *pPixels)
  {
    if (!c || !pCounters30) return limg_hip_error_ArgumentNull;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry);
    const int32_t was = c->opt.collect_stats;
    c->opt.collect_stats = 1;
    limg_hip_result r = limg_hip_encode3d(c, pIn, sizeX, sizeY, hasAlpha, pInfo, errorFactor, poolThreads, fastBitCrushing);
    if (r == limg_hip_success) r = limg_hip_last_stats(c, pCounters30, pPixels);
    c->opt.collect_stats = was;
    return r;
  }

  // The host-pointer entry in ROW BANDS: what every relinked caller of limg_encode3d_test hits moves 4 B/px up and 35 B/px down over PCIe, and
  // the download alone (2.35 GB for 8192^2) is ~43 ms at wire rate against 1.4 ms of kernels.  Upload, encode and download one after the other: 50.4 ms.  Here the
  // image goes up and is encoded band by band on the calling thread while a second host thread brings every finished band's rows of the 11 planes down (PCIe is full
  // duplex; blocking copies from / to the caller's pageable memory already run at wire rate, so no staging copy is added): the upload and the kernels of band k + 1
  // hide under the download of band k.
  //   * poolThreads == 0 (one dither chain through the image, src/limg.cpp:2110): bands of whole block rows through the two exchange-free halves of the chain entry
  //     -- E step + scan of the band, then its F step from the call count of the bands above it (k_chain_base on the device, stream-ordered, no host round trip);
  //   * poolThreads > 0 (src/limg.cpp:2114-2134): every chain restarts at the seed, so a band is a chain: an independent encode of its rows.
  // Images with partial edge blocks, small images and encodes that collect statistics take the plain path.
  static limg_hip_result host_encode_banded(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, void *const *hp, void *const *dp,
                                            uint32_t errorFactor, int poolThreads, int fastBitCrushing)
  {
    struct Band { size_t y0, y1; };
    const size_t blocksX = sizeX / kBlock, blocksY = sizeY / kBlock;
    const Partition pt = partition(sizeY, poolThreads);
    const bool chains = pt.chainCount > 1 && pt.chainRows != 0;
    Band bands[64];
    uint32_t nb = 0;
    if (chains)
    {
      if (pt.chainCount > 64) return limg_hip_error_InvalidParameter; // (the caller falls back to the plain path)
      for (uint32_t k = 0; k < pt.chainCount; k++)
      {
        bands[nb].y0 = (size_t)k * pt.chainRows * kBlock;
        bands[nb].y1 = k + 1 < pt.chainCount ? (size_t)(k + 1) * pt.chainRows * kBlock : sizeY;
        nb++;
      }
    }
    else
    {
#ifndef LIMG_HOST_BANDS
#define LIMG_HOST_BANDS 8
#endif
      const size_t want = LIMG_HOST_BANDS, rows = ((blocksY + want - 1) / want) * kBlock;
      for (size_t y = 0; y < sizeY; y += rows) { bands[nb].y0 = y; bands[nb].y1 = y + rows < sizeY ? y + rows : sizeY; nb++; }
    }
    limg_hip_result r;
    if (!c->hostStream) HIP_TRY(hipStreamCreateWithFlags(&c->hostStream, hipStreamNonBlocking));
    if (!c->hostCopyStream) HIP_TRY(hipStreamCreateWithFlags(&c->hostCopyStream, hipStreamNonBlocking));
    while (c->hostEvents.size() < nb)
    {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      try { c->hostEvents.push_back(e); }
      catch (...) { (void)hipEventDestroy(e); return limg_hip_error_MemoryAllocationFailure; } // (no exception may cross the extern "C" boundary)
    }
    if ((r = c->hostWords.ensure((2 * 64 + 2) * 8)) != limg_hip_success) return r;
    if (!c->devStatus.p)
    {
      if ((r = c->devStatus.ensure(16)) != limg_hip_success) return r;
      HIP_TRY(hipMemset(c->devStatus.p, 0, 16));
    }
    unsigned long long *dCalls = (unsigned long long *)c->hostWords.p, *dBase = dCalls + 64;
    hipStream_t s = c->hostStream;
    HIP_TRY(hipMemsetAsync(dCalls, 0, 2 * 64 * 8, s));

    // the downloads: a second host thread, band after band, as their events fire
    std::atomic<uint32_t> ready(0);      // bands whose kernels are enqueued and whose event is recorded
    std::atomic<int> failed(0);
    std::mutex m;
    std::condition_variable cv;
#ifdef LIMG_HOST_BAND_TIMING
    const auto t00 = std::chrono::steady_clock::now();
    auto ms_now = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t00).count(); };
#endif
    auto download = [&]() {
      if (hipSetDevice(c->device) != hipSuccess) { failed = 1; return; }
      for (uint32_t b = 0; b < nb; b++)
      {
#ifdef LIMG_HOST_BAND_TIMING
        const double tw = ms_now();
#endif
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [&] { return ready.load() > b || failed.load() != 0; });
        }
        if (failed.load() != 0) return;
        if (hipEventSynchronize(c->hostEvents[b]) != hipSuccess) { failed = 1; return; }
        const size_t o = bands[b].y0 * sizeX, n = (bands[b].y1 - bands[b].y0) * sizeX;
        for (int i = 0; i < 11; i++)
        {
          const size_t es = i < 8 ? 4 : 1;
          // (on a stream of its own: blocking hipMemcpy calls of two threads share the null stream and run one after the other -- measured: every upload waited
          // for the download in front of it)
          if (hipMemcpyAsync((uint8_t *)hp[i] + o * es, (const uint8_t *)dp[i] + o * es, n * es, hipMemcpyDeviceToHost, c->hostCopyStream) != hipSuccess) { failed = 1; return; }
        }
        if (hipStreamSynchronize(c->hostCopyStream) != hipSuccess) { failed = 1; return; }
#ifdef LIMG_HOST_BAND_TIMING
        fprintf(stderr, "band %u: download waited from %.2f, ran %.2f .. %.2f ms (%.1f GB/s)\n", b, tw, tw, ms_now(), n * 35 / 1e6 / (ms_now() - tw));
#endif
      }
    };
    std::thread *copier = nullptr;
    try { copier = new std::thread(download); }
    catch (...) { copier = nullptr; }

    limg_hip_result result = limg_hip_success;
    for (uint32_t b = 0; b < nb && result == limg_hip_success && failed.load() == 0; b++)
    {
      const size_t y0 = bands[b].y0, rows = bands[b].y1 - y0, o = y0 * sizeX;
      const uint32_t *dIn = (const uint32_t *)c->in.p + o;
      if (hipMemcpyAsync((void *)dIn, pIn + o, rows * sizeX * 4, hipMemcpyHostToDevice, s) != hipSuccess) { result = limg_hip_error_Generic; break; }
      limg_hip_encode3d_info d;
      void **q = reinterpret_cast<void **>(&d);
      for (int i = 0; i < 11; i++) q[i] = (uint8_t *)dp[i] + o * (i < 8 ? 4 : 1);
      if (chains) result = encode_device(c, dIn, sizeX, rows, hasAlpha, &d, nullptr, errorFactor, 0, fastBitCrushing, s);
      else
      {
        const size_t before = (y0 / kBlock) * blocksX;
        result = limg_hip_encode3d_chain_device(c, dIn, sizeX, rows, hasAlpha, &d, errorFactor, fastBitCrushing, 1, (uint64_t *)(dCalls + b), nullptr, before, s);
        if (result == limg_hip_success)
        {
          launch_chain_base(dCalls, (int)b, (int)nb, dBase + b, (uint32_t *)c->devStatus.p + 1, s);
          result = limg_hip_encode3d_chain_device(c, dIn, sizeX, rows, hasAlpha, &d, errorFactor, fastBitCrushing, 2, nullptr, (const uint64_t *)(dBase + b), before, s);
        }
      }
      if (result == limg_hip_success && hipEventRecord(c->hostEvents[b], s) != hipSuccess) result = limg_hip_error_Generic;
      if (result != limg_hip_success) break;
#ifdef LIMG_HOST_BAND_TIMING
      fprintf(stderr, "band %u: uploaded + enqueued at %.2f ms\n", b, ms_now());
#endif
      {
        std::lock_guard<std::mutex> lk(m);
        ready = b + 1;
      }
      cv.notify_all();
      if (!copier)
      { // no second thread to be had: this band comes down here and now (the plain order, band-wise)
        if (hipEventSynchronize(c->hostEvents[b]) != hipSuccess) { result = limg_hip_error_Generic; break; }
        for (int i = 0; i < 11; i++)
        {
          const size_t es = i < 8 ? 4 : 1;
          if (hipMemcpy((uint8_t *)hp[i] + o * es, (const uint8_t *)dp[i] + o * es, rows * sizeX * es, hipMemcpyDeviceToHost) != hipSuccess) { result = limg_hip_error_Generic; break; }
        }
      }
    }
    if (result != limg_hip_success)
    {
      std::lock_guard<std::mutex> lk(m);
      failed = 1;
    }
    cv.notify_all();
    if (copier) { copier->join(); delete copier; }
    if (result == limg_hip_success && failed.load() != 0) result = limg_hip_error_Generic;
    const limg_hip_result status = limg_hip_check_device_status(c); // (also waits for whatever is still enqueued)
    return result != limg_hip_success ? result : status;
  }

  limg_hip_result limg_hip_encode3d(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, limg_hip_encode3d_info *pInfo, uint32_t errorFactor,
                                    int poolThreads, int fastBitCrushing)
  {
    if (!c || !pIn || !pInfo) return limg_hip_error_ArgumentNull;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry);
    if (sizeX == 0 || sizeY == 0) return limg_hip_error_InvalidParameter;
    void *const *hp = reinterpret_cast<void *const *>(pInfo);
    for (int i = 0; i < 11; i++)
      if (!hp[i]) return limg_hip_error_ArgumentNull;
    HIP_TRY(hipSetDevice(c->device));
    const size_t px = sizeX * sizeY;
    limg_hip_result r;
    if ((r = c->in.ensure(px * 4)) != limg_hip_success) return r;
    if ((r = c->planes.ensure(px * 35 + 11 * 256)) != limg_hip_success) return r;
    limg_hip_encode3d_info d;
    uint8_t *base = (uint8_t *)c->planes.p;
    void **dp = reinterpret_cast<void **>(&d);
    size_t off = 0;
    for (int i = 0; i < 11; i++)
    {
      dp[i] = base + off;
      off += (i < 8 ? px * 4 : px);
      off = (off + 255) & ~(size_t)255;
    }
    // from 4 Mpixels on (below, the whole call is a few milliseconds and the bands' launches would not pay), whole blocks, at least 8 block rows per band
    const Partition pt = partition(sizeY, poolThreads);
    const bool chains = pt.chainCount > 1 && pt.chainRows != 0;
    if (px >= ((size_t)4 << 20) && sizeX % kBlock == 0 && sizeY % kBlock == 0 && sizeY >= 64 * kBlock && c->opt.collect_stats == 0 && !c->forceSplit &&
        (!chains || (pt.chainCount <= 64 && pt.chainRows >= 8)))
      return host_encode_banded(c, pIn, sizeX, sizeY, hasAlpha, hp, dp, errorFactor, poolThreads, fastBitCrushing);
    HIP_TRY(hipMemcpy(c->in.p, pIn, px * 4, hipMemcpyHostToDevice));
    if ((r = encode_device(c, (const uint32_t *)c->in.p, sizeX, sizeY, hasAlpha, &d, nullptr, errorFactor, poolThreads, fastBitCrushing, nullptr)) != limg_hip_success) return r;
    if ((r = limg_hip_check_device_status(c)) != limg_hip_success) return r;
    for (int i = 0; i < 11; i++) HIP_TRY(hipMemcpy(hp[i], dp[i], i < 8 ? px * 4 : px, hipMemcpyDeviceToHost));
    return limg_hip_success;
  }

  limg_hip_result limg_hip_encode3d_perf(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, uint32_t errorFactor, int poolThreads,
                                         int fastBitCrushing)
  {
    if (!c || !pIn) return limg_hip_error_ArgumentNull;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry);
    if (sizeX == 0 || sizeY == 0) return limg_hip_error_InvalidParameter;
    HIP_TRY(hipSetDevice(c->device));
    const size_t px = sizeX * sizeY;
    limg_hip_result r;
    if ((r = c->in.ensure(px * 4)) != limg_hip_success) return r;
    if (px >= ((size_t)4 << 20) && sizeX % kBlock == 0 && sizeY % kBlock == 0 && sizeY >= 64 * kBlock && c->opt.collect_stats == 0)
    { // Nothing is stored, so nothing depends on the dither chain (src/limg.cpp:2140-2173: fit + search only): the image goes up in 8 row bands on one stream and every
      // band's fit + search runs on a second one as soon as its rows have arrived -- the kernels (1.2 ms for 8192^2) hide under the upload (5 ms)
      const size_t rowsPer = ((sizeY / kBlock + 7) / 8) * kBlock;
      const uint32_t nb = (uint32_t)((sizeY + rowsPer - 1) / rowsPer);
      if (!c->hostStream) HIP_TRY(hipStreamCreateWithFlags(&c->hostStream, hipStreamNonBlocking));
      if (!c->hostCopyStream) HIP_TRY(hipStreamCreateWithFlags(&c->hostCopyStream, hipStreamNonBlocking));
      while (c->hostEvents.size() < nb)
      {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        try { c->hostEvents.push_back(e); }
        catch (...) { (void)hipEventDestroy(e); return limg_hip_error_MemoryAllocationFailure; }
      }
      for (uint32_t b = 0; b < nb; b++)
      {
        const size_t y0 = (size_t)b * rowsPer, rows = y0 + rowsPer < sizeY ? rowsPer : sizeY - y0, o = y0 * sizeX;
        HIP_TRY(hipMemcpyAsync((uint32_t *)c->in.p + o, pIn + o, rows * sizeX * 4, hipMemcpyHostToDevice, c->hostCopyStream));
        HIP_TRY(hipEventRecord(c->hostEvents[b], c->hostCopyStream));
        HIP_TRY(hipStreamWaitEvent(c->hostStream, c->hostEvents[b], 0));
        if ((r = encode_device(c, (const uint32_t *)c->in.p + o, sizeX, rows, hasAlpha, nullptr, nullptr, errorFactor, 0, fastBitCrushing, c->hostStream)) != limg_hip_success) return r;
      }
      return limg_hip_check_device_status(c);
    }
    HIP_TRY(hipMemcpy(c->in.p, pIn, px * 4, hipMemcpyHostToDevice));
    if ((r = encode_device(c, (const uint32_t *)c->in.p, sizeX, sizeY, hasAlpha, nullptr, nullptr, errorFactor, poolThreads, fastBitCrushing, nullptr)) != limg_hip_success) return r;
    HIP_TRY(hipDeviceSynchronize());
    return limg_hip_success;
  }

  double limg_hip_compare_device(limg_hip_context *c, const uint32_t *a, const uint32_t *b, size_t sizeX, size_t sizeY, int hasAlpha, double *pMse, double *pMax, void *stream)
  {
    if (!c || !a || !b || sizeX == 0 || sizeY == 0) return NAN;
    if (hipSetDevice(c->device) != hipSuccess) return NAN;
    if (c->cmp.ensure(8) != limg_hip_success) return NAN;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(c->cmp.p, 0, 8, s) != hipSuccess) return NAN;
    launch_compare(a, b, (uint64_t)sizeX * sizeY, hasAlpha ? 4 : 3, (unsigned long long *)c->cmp.p, s);
    unsigned long long err = 0;
    if (hipMemcpyAsync(&err, c->cmp.p, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return NAN;
    // maxError = limg_color_error(min, max): red diff 255^2 >= 0x4000 => factors {3,4,2,3}
    const double maxError = 255.0 * 255.0 * (hasAlpha ? 12.0 : 9.0);
    const double mse = (double)err / (double)(sizeX * sizeY);
    if (pMse) *pMse = mse;
    if (pMax) *pMax = maxError;
    return 10.0 * log10(maxError / mse);
  }

  double limg_hip_compare(limg_hip_context *c, const uint32_t *a, const uint32_t *b, size_t sizeX, size_t sizeY, int hasAlpha, double *pMse, double *pMax)
  {
    if (!c || !a || !b || sizeX == 0 || sizeY == 0) return NAN;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry);
    if (hipSetDevice(c->device) != hipSuccess) return NAN;
    const size_t bytes = sizeX * sizeY * 4;
    if (c->in.ensure(bytes) != limg_hip_success || c->planes.ensure(bytes) != limg_hip_success) return NAN;
    if (hipMemcpy(c->in.p, a, bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(c->planes.p, b, bytes, hipMemcpyHostToDevice) != hipSuccess) return NAN;
    return limg_hip_compare_device(c, (const uint32_t *)c->in.p, (const uint32_t *)c->planes.p, sizeX, sizeY, hasAlpha, pMse, pMax, nullptr);
  }

  limg_hip_result limg_hip_synth_random_gradient_device(uint32_t *pOut, size_t width, size_t height, uint64_t seed, int opaque, size_t y0, void *stream)
  {
    if (!pOut) return limg_hip_error_ArgumentNull;
    launch_synth_random_gradient(pOut, (uint32_t)width, (uint32_t)height, seed, opaque, (uint32_t)y0, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return limg_hip_success;
  }

  limg_hip_result limg_hip_synth_photo_noise_device(uint32_t *pOut, size_t width, size_t height, uint64_t seed, size_t y0, void *stream)
  {
    if (!pOut) return limg_hip_error_ArgumentNull;
    launch_synth_photo_noise(pOut, (uint32_t)width, (uint32_t)height, seed, (uint32_t)y0, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return limg_hip_success;
  }
  // ---- compact stream --------------------------------------------------------------------------------------------------------
  size_t limg_hip_stream_bound(size_t sizeX, size_t sizeY)
  {
    if (sizeX == 0 || sizeY == 0 || sizeX > 0x7FFFFFF8ull || sizeY > 0x7FFFFFF8ull) return 0;
    const size_t blocks = ((sizeX + kBlock - 1) / kBlock) * ((sizeY + kBlock - 1) / kBlock);
    if (blocks * 24 > 0xFFFFFFFFull) return 0; // entry.payloadWord is 32 bits
    return sizeof(limg_hip_stream_header) + blocks * sizeof(limg_hip_stream_block) + blocks * 192;
  }

  limg_hip_result limg_hip_encode_stream_device(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, uint8_t *pStream, size_t capacity,
                                                size_t *pBytes, uint32_t errorFactor, int poolThreads, int fastBitCrushing, void *stream)
  {
    if (!c || !pIn || !pStream) return limg_hip_error_ArgumentNull;
    const size_t bound = limg_hip_stream_bound(sizeX, sizeY);
    if (bound == 0) return limg_hip_error_InvalidParameter;
    if (capacity < bound) return limg_hip_error_OutOfBounds;
    if (((uintptr_t)pStream & 15u) != 0) return limg_hip_error_InvalidParameter;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t px = sizeX * sizeY, planeStride = (px + 255) & ~(size_t)255;
    const size_t blocksX = (sizeX + kBlock - 1) / kBlock, blocksY = (sizeY + kBlock - 1) / kBlock, blocks = blocksX * blocksY;
    const size_t tiles = (blocks + 255) / 256;
    limg_hip_result r;
    if ((r = c->streamFac.ensure(planeStride * 3)) != limg_hip_success) return r;
    // strip form of the packer (images of whole blocks): the encode kernel leaves one payload-word count per work strip (limg_hip_stream.hip)
    const size_t stripsX = (blocksX + kStripBlocks - 1) / kStripBlocks, nStrips = stripsX * blocksY;
    const bool stripForm = (sizeX % kBlock) == 0 && (sizeY % kBlock) == 0 && !c->forceSplit;
    if ((r = c->streamTiles.ensure(tiles * 4)) != limg_hip_success) return r;
    if (stripForm && (r = c->streamUnits.ensure(nStrips * 4)) != limg_hip_success) return r;
    if ((r = c->records.ensure(blocks * sizeof(limg_hip_block_record))) != limg_hip_success) return r;
    if ((r = c->shifts.ensure(blocks * 4)) != limg_hip_success) return r;
    limg_hip_encode3d_info info;
    memset(&info, 0, sizeof(info));
    info.pFactorsA = (uint8_t *)c->streamFac.p; info.pFactorsB = info.pFactorsA + planeStride; info.pFactorsC = info.pFactorsB + planeStride;
    limg_hip_compact_out comp = { (limg_hip_block_record *)c->records.p, (uint32_t *)c->shifts.p };
    EncodeExtra xs;
    xs.streamRaw = true;
    xs.stripWords = stripForm ? (uint32_t *)c->streamUnits.p : nullptr;
    if ((r = encode_device(c, pIn, sizeX, sizeY, hasAlpha, &info, &comp, errorFactor, poolThreads, fastBitCrushing, s, xs)) != limg_hip_success) return r;

    StreamParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.sizeX = (uint32_t)sizeX; sp.sizeY = (uint32_t)sizeY; sp.blocksX = (uint32_t)blocksX; sp.blocksY = (uint32_t)blocksY;
    sp.nBlocks = (uint32_t)blocks; sp.nTiles = (uint32_t)tiles; sp.channels = hasAlpha ? 4 : 3; sp.errorFactor = errorFactor;
    sp.flags = (fastBitCrushing ? 1u : 0u) | (c->opt.dither_pcg ? 2u : 0u);
    sp.fac[0] = info.pFactorsA; sp.fac[1] = info.pFactorsB; sp.fac[2] = info.pFactorsC;
    sp.records = comp.pRecords; sp.shifts = comp.pShifts;
    sp.stream = pStream; sp.tileBase = (uint32_t *)c->streamTiles.p;
    if (stripForm)
    {
      sp.stripWords = (uint32_t *)c->streamUnits.p;
      sp.stripsX = (uint32_t)stripsX; sp.nStrips = (uint32_t)nStrips;
      const size_t slots = (size_t)(c->persistentWorkgroups / 5) * 16; // 16 one-wave workgroups per CU (128 vector registers each: 4 per SIMD)
      sp.nWaves = (uint32_t)(nStrips < slots ? nStrips : slots);
    }
    mark(c, s);
    launch_stream_pack(sp, s);
    mark(c, s); mark(c, s); mark(c, s);
    HIP_TRY(hipGetLastError());
    if (pBytes)
    {
      limg_hip_stream_header h;
      HIP_TRY(hipMemcpyAsync(&h, pStream, sizeof(h), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      *pBytes = (size_t)h.totalBytes;
    }
    return limg_hip_success;
  }

  limg_hip_result limg_hip_decode_stream_device(limg_hip_context *c, const uint8_t *pStream, size_t streamBytes, uint32_t *pOut, size_t sizeX, size_t sizeY, void *stream)
  {
    if (!c || !pStream || !pOut) return limg_hip_error_ArgumentNull;
    if (limg_hip_stream_bound(sizeX, sizeY) == 0 || streamBytes < sizeof(limg_hip_stream_header)) return limg_hip_error_InvalidParameter;
    if (((uintptr_t)pStream & 15u) != 0 || ((uintptr_t)pOut & 15u) != 0) return limg_hip_error_InvalidParameter;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    limg_hip_result r;
    if (!c->streamStatus.p)
    {
      if ((r = c->streamStatus.ensure(256 + 2048)) != limg_hip_success) return r; // the status word, then the decode kernel's store sink (see DecodeParams::sink)
      HIP_TRY(hipMemsetAsync(c->streamStatus.p, 0, 8, s));
    }
    DecodeParams dp;
    memset(&dp, 0, sizeof(dp));
    dp.sizeX = (uint32_t)sizeX; dp.sizeY = (uint32_t)sizeY;
    dp.blocksX = (uint32_t)((sizeX + kBlock - 1) / kBlock); dp.blocksY = (uint32_t)((sizeY + kBlock - 1) / kBlock);
    dp.nBlocks = dp.blocksX * dp.blocksY;
    if (streamBytes < sizeof(limg_hip_stream_header) + (size_t)dp.nBlocks * sizeof(limg_hip_stream_block)) return limg_hip_error_OutOfBounds;
    dp.stream = pStream; dp.streamBytes = streamBytes; dp.out = pOut; dp.status = (uint32_t *)c->streamStatus.p; dp.sink = (uint32_t *)((uint8_t *)c->streamStatus.p + 256);
    mark(c, s);
    launch_stream_decode(dp, s);
    mark(c, s); mark(c, s); mark(c, s);
    HIP_TRY(hipGetLastError());
    return limg_hip_success;
  }

  limg_hip_result limg_hip_stream_info(const uint8_t *pStream, size_t streamBytes, size_t *pSizeX, size_t *pSizeY, int *pHasAlpha, size_t *pTotalBytes)
  {
    if (!pStream) return limg_hip_error_ArgumentNull;
    if (streamBytes < sizeof(limg_hip_stream_header)) return limg_hip_error_OutOfBounds;
    limg_hip_stream_header h;
    memcpy(&h, pStream, sizeof(h));
    if (h.magic != LIMG_HIP_STREAM_MAGIC || h.version != LIMG_HIP_STREAM_VERSION || (h.channels != 3 && h.channels != 4)) return limg_hip_error_InvalidParameter;
    if (limg_hip_stream_bound(h.sizeX, h.sizeY) == 0) return limg_hip_error_InvalidParameter;
    const uint64_t bx = ((uint64_t)h.sizeX + kBlock - 1) / kBlock, by = ((uint64_t)h.sizeY + kBlock - 1) / kBlock;
    if (h.blocksX != bx || h.blocksY != by) return limg_hip_error_InvalidParameter;
    if (h.payloadWords > bx * by * 24 || h.totalBytes != sizeof(h) + bx * by * sizeof(limg_hip_stream_block) + h.payloadWords * 8) return limg_hip_error_InvalidParameter;
    if (pSizeX) *pSizeX = h.sizeX;
    if (pSizeY) *pSizeY = h.sizeY;
    if (pHasAlpha) *pHasAlpha = h.channels == 4;
    if (pTotalBytes) *pTotalBytes = (size_t)h.totalBytes;
    return limg_hip_success;
  }

  limg_hip_result limg_hip_encode_stream(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, uint8_t *pStream, size_t capacity, size_t *pBytes,
                                         uint32_t errorFactor, int poolThreads, int fastBitCrushing)
  {
    if (!c || !pIn || !pStream || !pBytes) return limg_hip_error_ArgumentNull;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry);
    const size_t bound = limg_hip_stream_bound(sizeX, sizeY);
    if (bound == 0) return limg_hip_error_InvalidParameter;
    HIP_TRY(hipSetDevice(c->device));
    limg_hip_result r;
    const size_t px = sizeX * sizeY;
    if ((r = c->in.ensure(px * 4)) != limg_hip_success) return r;
    if ((r = c->streamBuf.ensure(bound)) != limg_hip_success) return r;
    HIP_TRY(hipMemcpy(c->in.p, pIn, px * 4, hipMemcpyHostToDevice));
    size_t bytes = 0;
    if ((r = limg_hip_encode_stream_device(c, (const uint32_t *)c->in.p, sizeX, sizeY, hasAlpha, (uint8_t *)c->streamBuf.p, bound, &bytes, errorFactor, poolThreads,
                                           fastBitCrushing, nullptr)) != limg_hip_success) return r;
    if ((r = limg_hip_check_device_status(c)) != limg_hip_success) return r;
    *pBytes = bytes;
    if (bytes > capacity) return limg_hip_error_OutOfBounds; // *pBytes tells the caller what it takes
    HIP_TRY(hipMemcpy(pStream, c->streamBuf.p, bytes, hipMemcpyDeviceToHost));
    return limg_hip_success;
  }

  limg_hip_result limg_hip_decode_stream(limg_hip_context *c, const uint8_t *pStream, size_t streamBytes, uint32_t *pOut, size_t outPixels)
  {
    if (!c || !pStream || !pOut) return limg_hip_error_ArgumentNull;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry);
    size_t sizeX = 0, sizeY = 0, total = 0;
    limg_hip_result r;
    if ((r = limg_hip_stream_info(pStream, streamBytes, &sizeX, &sizeY, nullptr, &total)) != limg_hip_success) return r;
    if (total > streamBytes || sizeX * sizeY > outPixels) return limg_hip_error_OutOfBounds;
    HIP_TRY(hipSetDevice(c->device));
    if ((r = c->streamBuf.ensure(total + 16)) != limg_hip_success) return r;
    if ((r = c->planes.ensure(sizeX * sizeY * 4)) != limg_hip_success) return r;
    HIP_TRY(hipMemcpy(c->streamBuf.p, pStream, total, hipMemcpyHostToDevice));
    if ((r = limg_hip_decode_stream_device(c, (const uint8_t *)c->streamBuf.p, total, (uint32_t *)c->planes.p, sizeX, sizeY, nullptr)) != limg_hip_success) return r;
    if ((r = limg_hip_check_device_status(c)) != limg_hip_success) return r;
    HIP_TRY(hipMemcpy(pOut, c->planes.p, sizeX * sizeY * 4, hipMemcpyDeviceToHost));
    return limg_hip_success;
  }
  // ---- merged-block encoder ------------------------------------------------------------------------------------------------------
  int limg_hip_host_blocked_matches(int channels, const limg_hip_block_record *pSeed, const limg_hip_block_record *pCandidate)
  {
    if (!pSeed || !pCandidate || (channels != 3 && channels != 4)) return -1;
    return blocked_matches_host(channels, *pSeed, *pCandidate) ? 1 : 0;
  }

  limg_hip_result limg_hip_host_blocked_merge(const limg_hip_block_record *pFits, const uint64_t *pMatchBits, size_t blocksX, size_t blocksY, int channels, limg_hip_region *pRegions,
                                              size_t capacity, size_t *pCount)
  {
    if (!pFits || !pCount) return limg_hip_error_ArgumentNull;
    if (blocksX == 0 || blocksY == 0 || blocksX > 0x0FFFFFFFull || blocksY > 0x0FFFFFFFull || (channels != 3 && channels != 4)) return limg_hip_error_InvalidParameter;
    std::vector<HostRegion> regs;
    std::vector<uint8_t> flags;
    if (pMatchBits)
    { // the per-seed viability flags the GPU kernel derives from the same bits (k_blocked_match)
      flags.resize(blocksX * blocksY + 16); // (+ 16: the merge's scan reads 16 flags at a time)
      for (size_t i = 0; i < blocksX * blocksY; i++)
      {
        const uint64_t *w = pMatchBits + i * kMatchWords;
        auto bit = [&](int dx, int dy) -> unsigned { const int cell = (dy + kMatchLo) * kMatchSide + dx + kMatchLo; return (unsigned)(w[cell >> 6] >> (cell & 63)) & 1u; };
        const unsigned all8 = bit(1, 0) & bit(2, 0) & bit(0, 1) & bit(1, 1) & bit(2, 1) & bit(0, 2) & bit(1, 2) & bit(2, 2);
        flags[i] = (uint8_t)(all8 | ((bit(1, 0) | bit(0, 1)) << 1));
      }
    }
    blocked_merge(pFits, (const unsigned long long *)pMatchBits, (uint32_t)blocksX, (uint32_t)blocksY, channels, regs, nullptr, nullptr, pMatchBits ? flags.data() : nullptr);
    *pCount = regs.size();
    if (pRegions)
      for (size_t i = 0; i < regs.size() && i < capacity; i++) pRegions[i] = { regs[i].ox, regs[i].oy, regs[i].rx, regs[i].ry };
    return limg_hip_success;
  }

  size_t limg_hip_host_blocked_match_words(void) { return kMatchWords; }

  limg_hip_result limg_hip_host_blocked_match_bits(const limg_hip_block_record *pFits, size_t blocksX, size_t blocksY, int channels, uint64_t *pMatchBits)
  {
    if (!pFits || !pMatchBits) return limg_hip_error_ArgumentNull;
    if (channels != 3 && channels != 4) return limg_hip_error_InvalidParameter;
    for (size_t sy = 0; sy < blocksY; sy++)
      for (size_t sx = 0; sx < blocksX; sx++)
      {
        uint64_t *w = pMatchBits + (sy * blocksX + sx) * kMatchWords;
        for (int i = 0; i < kMatchWords; i++) w[i] = 0;
        for (int cell = 0; cell < kMatchCells; cell++)
        {
          const long dy = cell / kMatchSide - kMatchLo, dx = cell % kMatchSide - kMatchLo;
          const long cx = (long)sx + dx, cy = (long)sy + dy;
          if ((dx | dy) == 0 || cx < 0 || cy < 0 || cx >= (long)blocksX || cy >= (long)blocksY) continue;
          if (blocked_matches_host(channels, pFits[sy * blocksX + sx], pFits[(size_t)cy * blocksX + cx])) w[cell >> 6] |= 1ull << (cell & 63);
        }
      }
    return limg_hip_success;
  }

  limg_hip_result limg_hip_blocked_encode3d_device(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, const limg_hip_blocked_encode3d_info *pInfo,
                                                   uint32_t errorFactor, int fastBitCrushing, void *stream)
  {
    if (!c || !pIn || !pInfo) return limg_hip_error_ArgumentNull;
    if (!pInfo->pDecoded || !pInfo->pFactorsA || !pInfo->pFactorsB || !pInfo->pFactorsC || !pInfo->pBitsPerPixel || !pInfo->pShiftABCX || !pInfo->pColAMin || !pInfo->pColAMax ||
        !pInfo->pColBMin || !pInfo->pColBMax || !pInfo->pColCMin || !pInfo->pColCMax || !pInfo->pBlockIndex)
      return limg_hip_error_ArgumentNull;
    if (sizeX == 0 || sizeY == 0 || sizeX > 0x7FFFFFF8ull || sizeY > 0x7FFFFFF8ull || sizeX * sizeY > 0x60000000ull) return limg_hip_error_InvalidParameter;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const clk::time_point t0 = clk::now();
    const int channels = hasAlpha ? 4 : 3;
    const uint32_t blocksX = (uint32_t)((sizeX + kBlock - 1) / kBlock), blocksY = (uint32_t)((sizeY + kBlock - 1) / kBlock);
    const size_t blocks = (size_t)blocksX * blocksY;
    limg_hip_result r;

    constexpr size_t kInFlight = 32; // batches of the worker (below) whose fit + search kernel has been enqueued and whose chain has not been walked yet
    while (c->workTimers.size() < 4 * kInFlight + 3)
    {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      c->workTimers.push_back(e);
    }
    hipEvent_t *frontTimers = c->workTimers.data() + 4 * kInFlight;

    // pass 1 (src/limg.cpp:1088-1119): every block's own fit = the 8x8 path's E step, records only
    EncodeExtra x1;
    x1.fitOnly = true;
    HIP_TRY(hipEventRecord(frontTimers[0], s));
    if ((r = encode_device(c, pIn, sizeX, sizeY, hasAlpha, nullptr, nullptr, errorFactor, 0, fastBitCrushing, s, x1)) != limg_hip_success) return r;

    BlockedParams bp;
    memset(&bp, 0, sizeof(bp));
    bp.in = pIn; bp.sizeX = (uint32_t)sizeX; bp.sizeY = (uint32_t)sizeY; bp.blocksX = blocksX; bp.blocksY = blocksY; bp.channels = (uint32_t)channels;
    const uint64_t maxPixel = (uint64_t)0x6 * (errorFactor / 2) * 7, maxBlock = (uint64_t)0x4 * (errorFactor / 2) * 7; // src/limg.cpp:2343-2368, same values as the 8x8 path
    bp.maxPixel32 = maxPixel > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)maxPixel;
    bp.maxBlock = maxBlock;
    bp.crushBits = errorFactor != 0; bp.fast = fastBitCrushing != 0;
    const bool forced = c->opt.forced_shift[0] >= 0 && c->opt.forced_shift[0] <= 8 && c->opt.forced_shift[1] >= 0 && c->opt.forced_shift[1] <= 8 &&
                        c->opt.forced_shift[2] >= 0 && c->opt.forced_shift[2] <= 8;
    for (int i = 0; i < 3; i++) bp.forced[i] = forced ? c->opt.forced_shift[i] : -1;
    bp.pass1 = (const limg_hip_block_record *)c->records.p;
    if ((r = c->bMatch.ensure(blocks * kMatchWords * 8)) != limg_hip_success) return r;
    bp.matchBits = (unsigned long long *)c->bMatch.p;
    if ((r = c->bFlags.ensure(blocks)) != limg_hip_success) return r;
    if ((r = c->hFlags.ensure(blocks + 16)) != limg_hip_success) return r; // (+ 16: the merge's scan reads 16 flags at a time)
    bp.matchFlags = (uint8_t *)c->bFlags.p;
    if (TOPT(c, blocked_no_bound) == 0)
    {
      if ((r = c->bBound.ensure(blocks * 16)) != limg_hip_success) return r;
      bp.matchBound = (float *)c->bBound.p;
    }
    uint8_t *hFlags = (uint8_t *)c->hFlags.p;
    bp.info = *pInfo;
    // The similarity bits are produced and copied band by band (block rows) so that the merge, which consumes seeds in raster order, can start
    // after the first band: kernel launches on `s`, copies on a second stream chained by events.
    if ((r = c->hRec.ensure(blocks * sizeof(limg_hip_block_record))) != limg_hip_success) return r;
    if ((r = c->hBits.ensure(blocks * kMatchWords * 8)) != limg_hip_success) return r;
    limg_hip_block_record *hRec = (limg_hip_block_record *)c->hRec.p;
    unsigned long long *hBits = (unsigned long long *)c->hBits.p;
    constexpr uint32_t kBands = 16;
    const uint32_t bandRows = (blocksY + kBands - 1) / kBands, nBands = (blocksY + bandRows - 1) / bandRows;
    if (!c->copyStream) HIP_TRY(hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
    while (c->bandEvents.size() < 2 * kBands + 1)
    {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      c->bandEvents.push_back(e);
    }
    hipStream_t cs = c->copyStream;
    // The records go to the host as well, but the merge reads them only for pairs outside the similarity window (a few dozen per image): their copy (64 MB for 8192^2,
    // 1.3 ms of PCIe) is queued BEHIND the first two bands' bits, and the merge waits for it when it first needs a record -- not before it starts.
    hipEvent_t evPass1 = c->bandEvents[2 * kBands];
    HIP_TRY(hipEventRecord(frontTimers[1], s));
    auto copy_records = [&]() -> limg_hip_result
    {
      HIP_TRY(hipMemcpyAsync(hRec, c->records.p, blocks * sizeof(limg_hip_block_record), hipMemcpyDeviceToHost, cs)); // (`cs` has waited for a band's kernel: pass 1 is long done)
      HIP_TRY(hipEventRecord(evPass1, cs)); // "records are on the host"
      return limg_hip_success;
    };
    c->lastBlocks = blocks;
    launch_blocked_bounds(bp, s);
    for (uint32_t b = 0; b < nBands; b++)
    {
      const uint32_t row0 = b * bandRows, rows = min(bandRows, blocksY - row0);
      bp.seedBase = row0 * blocksX; bp.seedCount = rows * blocksX;
      launch_blocked_match(bp, s);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(c->bandEvents[2 * b], s));
      HIP_TRY(hipStreamWaitEvent(cs, c->bandEvents[2 * b], 0));
      HIP_TRY(hipMemcpyAsync(hBits + (size_t)bp.seedBase * kMatchWords, (unsigned long long *)c->bMatch.p + (size_t)bp.seedBase * kMatchWords, (size_t)bp.seedCount * kMatchWords * 8,
                             hipMemcpyDeviceToHost, cs));
      HIP_TRY(hipMemcpyAsync(hFlags + bp.seedBase, (uint8_t *)c->bFlags.p + bp.seedBase, bp.seedCount, hipMemcpyDeviceToHost, cs));
      HIP_TRY(hipEventRecord(c->bandEvents[2 * b + 1], cs));
      if (b == 1 || (b == 0 && nBands == 1))
        if ((r = copy_records()) != limg_hip_success) return r;
    }
    HIP_TRY(hipEventRecord(frontTimers[2], s)); // (`s` holds nothing but the similarity kernels between the two timers: the copies run on `cs`)
    HIP_TRY(hipEventSynchronize(c->bandEvents[1])); // the first band's bits: the merge can start
    const clk::time_point t1 = clk::now();
    uint32_t bandsReady = 0;
    bool bandError = false, recordsHere = false;
    double bandWaitMs = 0;
    const std::function<void()> needRecords = [&]() {
      if (!recordsHere && hipEventSynchronize(evPass1) != hipSuccess) bandError = true;
      recordsHere = true;
    };
    const std::function<void(uint32_t)> needSeedRow = [&](uint32_t row) {
      while (bandsReady < nBands && row >= bandsReady * bandRows)
      {
        const clk::time_point q0 = clk::now();
        if (hipEventSynchronize(c->bandEvents[2 * bandsReady + 1]) != hipSuccess) bandError = true;
        bandWaitMs += ms(q0, clk::now());
        bandsReady++;
      }
    };

    // Everything after this point is a two-thread pipeline.  This thread runs the greedy raster merge (serial by construction; it only looks the
    // similarity bits up) and publishes finished rectangles every few thousand; a worker thread takes them batch by batch, in creation order:
    // fit + search kernel, copy of the shift words, dither chain walk for the batch (the chain is serial too, but independent of the merge),
    // noise upload, store kernel.  Buffers are sized for the worst case up front so that nothing is reallocated while both threads run.
    const size_t px = sizeX * sizeY;
    const uint64_t capMax = ((uint64_t)px + 3ull * blocks + 3ull) & ~3ull; // every rectangle's scratch slice is rounded up to a multiple of 4
    if (capMax > 0xFFFFFFF0ull) return limg_hip_error_InvalidParameter;
    if ((r = c->hDesc.ensure(blocks * sizeof(RegionDesc))) != limg_hip_success) return r;
    if ((r = c->hOut.ensure(blocks * sizeof(RegionOut))) != limg_hip_success) return r;
    if ((r = c->hNoiseBase.ensure(blocks * 8 + 8)) != limg_hip_success) return r;
    const size_t maxCalls = 3 * blocks; // per dither call: the chain value it starts from (8 B), where its noise bytes go (8 B), its pixel count (4 B)
    if ((r = c->hNoise.ensure(maxCalls * 20 + 64)) != limg_hip_success) return r;
    if ((r = c->bCalls.ensure(maxCalls * 20 + 64)) != limg_hip_success) return r;
    if ((r = c->bRegions.ensure(blocks * sizeof(RegionDesc))) != limg_hip_success) return r;
    if ((r = c->bOut.ensure(blocks * sizeof(RegionOut))) != limg_hip_success) return r;
    if ((r = c->bNoiseBase.ensure(blocks * 8 + 8)) != limg_hip_success) return r;
    if ((r = c->bOrder.ensure(blocks * 4)) != limg_hip_success) return r;
    if ((r = c->bNoise.ensure(3 * px + 64)) != limg_hip_success) return r;
    if ((r = c->bPx.ensure(capMax * 4)) != limg_hip_success) return r;
    if ((r = c->bFac.ensure(capMax * 3)) != limg_hip_success) return r;
    if (!c->workStream) HIP_TRY(hipStreamCreateWithFlags(&c->workStream, hipStreamNonBlocking));
    RegionDesc *desc = (RegionDesc *)c->hDesc.p;
    RegionOut *hOut = (RegionOut *)c->hOut.p;
    unsigned long long *noiseBase = (unsigned long long *)c->hNoiseBase.p;
    unsigned long long *callState = (unsigned long long *)c->hNoise.p, *callOff = callState + maxCalls;
    uint32_t *callPx = (uint32_t *)(callOff + maxCalls);
    unsigned long long *dCallState = (unsigned long long *)c->bCalls.p, *dCallOff = dCallState + maxCalls;
    uint32_t *dCallPx = (uint32_t *)(dCallOff + maxCalls);
    std::vector<uint32_t> npx(blocks);
    bp.scratchPx = (uint32_t *)c->bPx.p; bp.scratchFac = (uint8_t *)c->bFac.p; bp.scratchCap = (uint32_t)capMax;
    {
      // k_blocked_store's 4-pixels-per-lane form: whole blocks (every rectangle row is a multiple of 8 pixels, every scratch / noise offset a multiple of 4) and planes
      // whose rows start 16-byte (32-bit planes) / 4-byte (byte planes) aligned
      const limg_hip_blocked_encode3d_info &bi = bp.info;
      uintptr_t w = 0, b8 = 0;
      const void *words[] = { bi.pDecoded, bi.pShiftABCX, bi.pColAMin, bi.pColAMax, bi.pColBMin, bi.pColBMax, bi.pColCMin, bi.pColCMax, bi.pBlockIndex };
      const void *bytes[] = { bi.pFactorsA, bi.pFactorsB, bi.pFactorsC, bi.pBitsPerPixel };
      for (const void *q : words) w |= (uintptr_t)q;
      for (const void *q : bytes) b8 |= (uintptr_t)q;
      bp.vecStore = (sizeX % kBlock == 0 && sizeY % kBlock == 0 && (w & 15u) == 0 && (b8 & 3u) == 0 && TOPT(c, blocked_no_vec_store) == 0) ? 1 : 0;
    }
    bp.noise = (const uint8_t *)c->bNoise.p;

    struct Pipe { std::mutex m; std::condition_variable cv; size_t ready = 0; bool finished = false; } pipe;
    limg_hip_result workerResult = limg_hip_success;
    double busy[3] = { 0, 0, 0 }; // worker: fit + search (incl. copies), chain walk, store launch
    const bool pcg = c->opt.dither_pcg != 0;
    // One batch = everything the merge has published when the worker looks; one stream for the fit + search kernels.  Measured on one box (profiles/archive/r04_blocked_pipeline.md):
    // batches capped at 8 K ... 64 K rectangles, two or four streams round-robin, a high-priority stream -- all within +-2 ms of this, most of them worse: the GPU
    // (similarity kernels 13 ms + fit / search kernels 13 ms per 8192^2 image) is as busy as the two host threads, so reordering its queue buys nothing.
    constexpr size_t kBatchRegions = (size_t)1 << 30;
    constexpr size_t kWorkStreams = 1;
    while (c->workStreams.size() < kWorkStreams)
    {
      hipStream_t st;
      HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      c->workStreams.push_back(st);
    }
    if (!c->storeStream) HIP_TRY(hipStreamCreateWithFlags(&c->storeStream, hipStreamNonBlocking));
    hipStream_t ss = c->storeStream; // noise expansion + store kernels of a batch: beside the next batch's fit + search kernel, not behind it

    while (c->workEvents.size() < kInFlight)
    {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      c->workEvents.push_back(e);
    }

    double kernelMs[2] = { 0, 0 };
    std::vector<uint8_t> storeTimed(kInFlight, 0); // slot i's store timers hold a finished-or-enqueued interval that has not been added up yet

    double dbgEnqueue = 0, dbgWait = 0; int dbgBatches = 0;
    std::thread worker([&]() {
      if (hipSetDevice(c->device) != hipSuccess) { workerResult = limg_hip_error_Generic; }
      // The GPU runs AHEAD of this thread: whatever the merge has published goes to the device at once (rectangle table up, fit + search kernel, records and
      // shift words back, one event per batch, up to kInFlight batches), and the chain -- this thread's real work, serial by construction -- is walked batch by batch
      // in creation order as the results arrive.  (Rounds 2-3 kept one batch in flight: every batch's GPU round trip was waited for, 11-20 ms per image.)
      struct Batch { size_t r0 = 0, r1 = 0; size_t ev = 0; double tq = 0; };
      constexpr bool dbgTimeline = kDebugTimeline;
      std::vector<Batch> queue; // FIFO: [head, queue.size())
      size_t head = 0, issued = 0, evNext = 0;
      uint64_t chain = kDitherSeed, noiseOff = 0;
      size_t callCount = 0;
      bool fin = false;
      constexpr size_t kOrderFrom = 512; // batches from this many rectangles on get the device-side "large rectangles first" order (k_blocked_order)
      auto params_of = [&](const Batch &b) {
        BlockedParams q = bp;
        q.regions = (const RegionDesc *)c->bRegions.p + b.r0; q.nRegions = (uint32_t)(b.r1 - b.r0); q.regionBase = (uint32_t)b.r0;
        q.out = (RegionOut *)c->bOut.p + b.r0;
        q.noiseBase = (const unsigned long long *)c->bNoiseBase.p + b.r0;
        q.order = (b.r1 - b.r0 >= kOrderFrom && TOPT(c, blocked_no_order) == 0) ? (uint32_t *)c->bOrder.p + b.r0 : nullptr; // (a small batch is one round of workgroups anyway)
        return q;
      };
      // Everything the merge has published since the last look goes to the GPU.  mayWait: nothing is left to walk, so wait for the merge.  Called at the top of
      // every round AND between the pieces of a batch's chain walk: a batch's walk takes milliseconds, and what the merge publishes meanwhile should be on the GPU
      // (kernel latency: the life of its largest rectangle, 0.6-2 ms) before this thread comes looking for it -- not be enqueued when the walk is over.
      // A kernel's duration is the life of its largest rectangle whatever the batch's size and the batches of a stream run one after the other, so a look from
      // inside a walk (minNew > 0) takes a batch only when it is worth a launch; a look with nothing else to do takes whatever there is.
      // (same-box A/B of these three and of the merge's first report, tools/r04/run38.sh: photo-noise 27.5-27.7 ms against 28.8-32.0 with "any size, looks every
      //  8192 rectangles, first report at 4096", gradient 20.3-20.4 against 19.9-21.1)
      constexpr size_t kWorthWithOneInFlight = 16384, kWorthFromInsideAWalk = 8192, kWalkPiece = 2048;
      auto enqueue_published = [&](bool mayWait, size_t minNew = 0)
      {
        if (fin || queue.size() - head >= kInFlight) return;
        size_t r0 = 0, r1 = 0;
        {
          std::unique_lock<std::mutex> lk(pipe.m);
          if (mayWait) pipe.cv.wait(lk, [&] { return pipe.ready > issued || pipe.finished; });
          if (pipe.ready > issued && (pipe.ready - issued >= minNew || pipe.finished)) { r0 = issued; r1 = pipe.ready - issued > kBatchRegions ? issued + kBatchRegions : pipe.ready; issued = r1; }
          else if (pipe.ready > issued) {}
          else fin = pipe.finished;
        }
        if (r1 <= r0) return;
        dbgBatches++;
        Batch nb; nb.r0 = r0; nb.r1 = r1; nb.ev = evNext; evNext = (evNext + 1) % kInFlight; nb.tq = ms(t0, clk::now());
        if (storeTimed[nb.ev])
        { // the slot comes round again: its previous batch's store kernels were enqueued kInFlight batches ago
          float t = 0;
          if (hipEventSynchronize(c->workTimers[4 * nb.ev + 3]) == hipSuccess && hipEventElapsedTime(&t, c->workTimers[4 * nb.ev + 2], c->workTimers[4 * nb.ev + 3]) == hipSuccess) kernelMs[1] += t;
          storeTimed[nb.ev] = 0;
        }
        if (workerResult == limg_hip_success)
        {
          const size_t n = r1 - r0;
          const BlockedParams q = params_of(nb);
          hipStream_t bs = c->workStreams[nb.ev % kWorkStreams];
          bool ok = hipMemcpyAsync((RegionDesc *)c->bRegions.p + r0, desc + r0, n * sizeof(RegionDesc), hipMemcpyHostToDevice, bs) == hipSuccess;
          ok = ok && hipEventRecord(c->workTimers[4 * nb.ev], bs) == hipSuccess;
          if (ok) { launch_blocked_order(q, bs); launch_blocked_fit_search(q, bs); ok = hipGetLastError() == hipSuccess; }
          ok = ok && hipEventRecord(c->workTimers[4 * nb.ev + 1], bs) == hipSuccess;
          ok = ok && hipMemcpyAsync(hOut + r0, (RegionOut *)c->bOut.p + r0, n * sizeof(RegionOut), hipMemcpyDeviceToHost, bs) == hipSuccess;
          ok = ok && hipEventRecord(c->workEvents[nb.ev], bs) == hipSuccess;
          if (!ok) workerResult = limg_hip_error_Generic;
        }
        queue.push_back(nb);
      };
      for (;;)
      {
        const clk::time_point w0 = clk::now();
        enqueue_published(head == queue.size(), head == queue.size() ? 0 : kWorthWithOneInFlight); // (with a batch in flight to wait for and walk, small change accumulates meanwhile)
        const clk::time_point w0b = clk::now();
        dbgEnqueue += ms(w0, w0b);
        // 2. the oldest batch in flight: its shift words are (about to be) back
        if (head < queue.size())
        {
          const Batch pending = queue[head++];
          if (workerResult == limg_hip_success)
          {
            bool ok = hipEventSynchronize(c->workEvents[pending.ev]) == hipSuccess;
            {
              float t = 0;
              if (ok && hipEventElapsedTime(&t, c->workTimers[4 * pending.ev], c->workTimers[4 * pending.ev + 1]) == hipSuccess) kernelMs[0] += t;
            }
            const clk::time_point w1 = clk::now();
            dbgWait += ms(w0b, w1);
            // the dither chain (src/limg_internal.h:711, src/limg.cpp:1541-1551): one chain through all rectangles in creation order; a call over N
            // pixels advances it by floor(N / 8) AES rounds + N % 8 PCG steps, so it is walked here -- for the chain VALUES only: every call's start value, pixel
            // count and place in the noise buffer go up (20 bytes per call) and k_noise_expand_calls produces the byte every pixel adds on the device.  (Rounds
            // 1-3 wrote the bytes here and uploaded them: 200 MB per 8192^2 image through this thread's store buffers and over PCIe.)
            const size_t call0 = callCount;
            // (kWalkPiece rectangles between two looks at what the merge has published: 0.1-0.6 ms of chain)
            for (size_t w = pending.r0; ok && w < pending.r1; w += kWalkPiece)
            {
              const size_t n = pending.r1 - w < kWalkPiece ? pending.r1 - w : kWalkPiece;
              chain = chain_walk_batch(chain, n, reinterpret_cast<const uint8_t *>(&hOut[w].shiftWord), sizeof(RegionOut), npx.data() + w, noiseBase + w, callState, callOff, callPx,
                                       noiseOff, callCount, maxCalls, pcg);
              if (w + n < pending.r1) enqueue_published(false, kWorthFromInsideAWalk);
            }
            const clk::time_point w2 = clk::now();
            const BlockedParams q = params_of(pending);
            const size_t nc = callCount - call0;
            ok = ok && hipStreamWaitEvent(ss, c->workEvents[pending.ev], 0) == hipSuccess; // this batch's records and shift words are in bOut
            ok = ok && hipEventRecord(c->workTimers[4 * pending.ev + 2], ss) == hipSuccess;
            if (ok && nc)
            {
              ok = hipMemcpyAsync(dCallState + call0, callState + call0, nc * 8, hipMemcpyHostToDevice, ss) == hipSuccess &&
                   hipMemcpyAsync(dCallOff + call0, callOff + call0, nc * 8, hipMemcpyHostToDevice, ss) == hipSuccess &&
                   hipMemcpyAsync(dCallPx + call0, callPx + call0, nc * 4, hipMemcpyHostToDevice, ss) == hipSuccess;
              if (ok) { launch_noise_expand_calls((uint8_t *)c->bNoise.p, dCallState + call0, dCallOff + call0, dCallPx + call0, nc, pcg, ss); ok = hipGetLastError() == hipSuccess; }
            }
            ok = ok && hipMemcpyAsync((unsigned long long *)c->bNoiseBase.p + pending.r0, noiseBase + pending.r0, (pending.r1 - pending.r0) * 8, hipMemcpyHostToDevice, ss) == hipSuccess;
            if (ok) { launch_blocked_store(q, ss); ok = hipGetLastError() == hipSuccess; }
            if (ok && hipEventRecord(c->workTimers[4 * pending.ev + 3], ss) == hipSuccess) storeTimed[pending.ev] = 1;
            const clk::time_point w3 = clk::now();
            busy[0] += ms(w0, w1); busy[1] += ms(w1, w2); busy[2] += ms(w2, w3);
            if (dbgTimeline) fprintf(stderr, "batch %zu..%zu (%zu rects): enqueued %.2f, waited from %.2f to %.2f, walked until %.2f, stores enqueued %.2f\n", pending.r0, pending.r1, pending.r1 - pending.r0,
                                     pending.tq, ms(t0, w0b), ms(t0, w1), ms(t0, w2), ms(t0, w3));
            if (!ok) workerResult = limg_hip_error_Generic;
          }
        }
        if (fin && head == queue.size()) break;
      }
      for (hipStream_t st : c->workStreams)
        if (hipStreamSynchronize(st) != hipSuccess && workerResult == limg_hip_success) workerResult = limg_hip_error_Generic;
      if (hipStreamSynchronize(ss) != hipSuccess && workerResult == limg_hip_success) workerResult = limg_hip_error_Generic;
      for (size_t i = 0; i < kInFlight; i++)
        if (storeTimed[i])
        {
          float t = 0;
          if (hipEventElapsedTime(&t, c->workTimers[4 * i + 2], c->workTimers[4 * i + 3]) == hipSuccess) kernelMs[1] += t;
        }
    });

    // producer: the merge; its progress callback lays the finished rectangles out (pixel counts, scratch slices) and hands them over
    size_t laid = 0;
    uint64_t cap = 0;
    const std::function<void(size_t)> progress = [&](size_t count) {
      const std::vector<HostRegion> &regs = c->lastRegions;
      for (size_t i = laid; i < count; i++)
      {
        const HostRegion &h = regs[i];
        size_t xpx = (size_t)h.rx * kBlock, ypx = (size_t)h.ry * kBlock;
        if (h.ox + h.rx == blocksX && (sizeX % kBlock)) xpx = xpx - kBlock + sizeX % kBlock;
        if (h.oy + h.ry == blocksY && (sizeY % kBlock)) ypx = ypx - kBlock + sizeY % kBlock;
        npx[i] = (uint32_t)(xpx * ypx);
        desc[i] = { h.ox, h.oy, h.rx, h.ry, h.keep, (uint32_t)cap, { 0, 0 } };
        cap += ((uint64_t)npx[i] + 3) & ~3ull;
      }
      laid = count;
      { std::lock_guard<std::mutex> lk(pipe.m); pipe.ready = count; }
      pipe.cv.notify_one();
    };
    bool mergeFailed = false;
    try { blocked_merge(hRec, hBits, blocksX, blocksY, channels, c->lastRegions, &progress, &needSeedRow, hFlags, &needRecords); }
    catch (...) { mergeFailed = true; } // out of host memory: the worker must still be released and joined
    needSeedRow(blocksY - 1); // every band's copy is complete before the staging buffers can be reused
    needRecords();
    const clk::time_point t2 = clk::now();
    { std::lock_guard<std::mutex> lk(pipe.m); pipe.finished = true; }
    pipe.cv.notify_one();
    worker.join();
    const clk::time_point t5 = clk::now();
    {
      float a = 0, b = 0; // (both intervals ended before the merge's last band arrived)
      const bool ok = hipEventElapsedTime(&a, frontTimers[0], frontTimers[1]) == hipSuccess && hipEventElapsedTime(&b, frontTimers[1], frontTimers[2]) == hipSuccess;
      c->blockedKernelMs[0] = ok ? a : 0; c->blockedKernelMs[1] = ok ? b : 0;
    }
    c->blockedKernelMs[2] = kernelMs[0]; c->blockedKernelMs[3] = kernelMs[1];
    c->blockedMs[0] = ms(t0, t1); c->blockedMs[1] = ms(t1, t2); c->blockedMs[2] = busy[0]; c->blockedMs[3] = busy[1]; c->blockedMs[4] = busy[2]; c->blockedMs[5] = ms(t0, t5);
    if (kDebugTimeline) fprintf(stderr, "merge from %.2f to %.2f, call ended %.2f\n", ms(t0, t1), ms(t0, t2), ms(t0, t5));
    if (kDebugTiming) fprintf(stderr, "merge waited %.2f ms for similarity-bit bands; ", bandWaitMs);
    if (kDebugTiming) fprintf(stderr, "worker: %d batches, enqueue %.2f ms, event wait %.2f ms, chain %.2f ms, store enqueue %.2f ms\n", dbgBatches, dbgEnqueue, dbgWait, busy[1], busy[2]);
    if (mergeFailed) return limg_hip_error_MemoryAllocationFailure;
    if (bandError) return limg_hip_error_Generic;
    if (workerResult == limg_hip_success && c->opt.collect_stats)
    { // src/limg.cpp:1561-1590 per rectangle: (8 - shift) bits for each of its pixels, and the pixels by shift
      memset(c->statsHost, 0, sizeof(c->statsHost));
      for (size_t i = 0; i < c->lastRegions.size(); i++)
        for (int f = 0; f < 3; f++)
        {
          uint32_t sh = (hOut[i].shiftWord >> (8 * f)) & 0xFFu;
          if (sh > 8) sh = 8;
          c->statsHost[f] += (uint64_t)(8 - sh) * npx[i];
          c->statsHost[3 + 9 * f + sh] += npx[i];
        }
      c->statsState = 2; c->statsPixels = (uint64_t)sizeX * sizeY;
    }
    return workerResult;
  }

  limg_hip_result limg_hip_blocked_regions(limg_hip_context *c, limg_hip_region *pRegions, size_t capacity, size_t *pCount)
  {
    if (!c || !pCount) return limg_hip_error_ArgumentNull;
    *pCount = c->lastRegions.size();
    if (pRegions)
      for (size_t i = 0; i < c->lastRegions.size() && i < capacity; i++) pRegions[i] = { c->lastRegions[i].ox, c->lastRegions[i].oy, c->lastRegions[i].rx, c->lastRegions[i].ry };
    return limg_hip_success;
  }

  limg_hip_result limg_hip_blocked_timing(limg_hip_context *c, double *pMs6)
  {
    if (!c || !pMs6) return limg_hip_error_ArgumentNull;
    memcpy(pMs6, c->blockedMs, sizeof(c->blockedMs));
    return limg_hip_success;
  }

  limg_hip_result limg_hip_blocked_match_bits(limg_hip_context *c, uint64_t *pBits, size_t capacityWords, size_t *pWords)
  { // the similarity bits the last merged-block encode's merge worked from (they stay in the context's pinned staging buffer until the next encode)
    if (!c || !pWords) return limg_hip_error_ArgumentNull;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry);
    const size_t words = c->lastBlocks * kMatchWords;
    *pWords = words;
    if (pBits && c->hBits.p) memcpy(pBits, c->hBits.p, (words < capacityWords ? words : capacityWords) * 8);
    return limg_hip_success;
  }

  limg_hip_result limg_hip_blocked_kernel_timing(limg_hip_context *c, double *pMs4)
  {
    if (!c || !pMs4) return limg_hip_error_ArgumentNull;
    memcpy(pMs4, c->blockedKernelMs, sizeof(c->blockedKernelMs));
    return limg_hip_success;
  }

  limg_hip_result limg_hip_blocked_encode3d_stats(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, limg_hip_blocked_encode3d_info *pInfo,
                                                  uint32_t errorFactor, int fastBitCrushing, uint64_t *pCounters30, uint64_t *pPixels)
  { // (see limg_hip_encode3d_stats; upstream: src/limg.cpp:1561-1590 counters, printed by limg_blocked_encode3d_test itself)
    if (!c || !pCounters30) return limg_hip_error_ArgumentNull;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry);
    const int32_t was = c->opt.collect_stats;
    c->opt.collect_stats = 1;
    limg_hip_result r = limg_hip_blocked_encode3d(c, pIn, sizeX, sizeY, hasAlpha, pInfo, errorFactor, fastBitCrushing);
    if (r == limg_hip_success) r = limg_hip_last_stats(c, pCounters30, pPixels);
    c->opt.collect_stats = was;
    return r;
  }

  limg_hip_result limg_hip_blocked_encode3d(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, limg_hip_blocked_encode3d_info *pInfo, uint32_t errorFactor,
                                            int fastBitCrushing)
  {
    if (!c || !pIn || !pInfo) return limg_hip_error_ArgumentNull;
    std::lock_guard<std::recursive_mutex> hostLock(c->hostEntry);
    if (sizeX == 0 || sizeY == 0) return limg_hip_error_InvalidParameter;
    HIP_TRY(hipSetDevice(c->device));
    const size_t px = sizeX * sizeY, stride = (px * 4 + 255) & ~(size_t)255;
    limg_hip_result r;
    if ((r = c->in.ensure(px * 4)) != limg_hip_success) return r;
    if ((r = c->planes.ensure(stride * 13)) != limg_hip_success) return r;
    HIP_TRY(hipMemcpy(c->in.p, pIn, px * 4, hipMemcpyHostToDevice));
    uint8_t *base = (uint8_t *)c->planes.p;
    limg_hip_blocked_encode3d_info d;
    memset(&d, 0, sizeof(d));
    // 13 written planes, one `stride` each (the uint8 ones use a quarter of theirs)
    void **hostp[13] = { (void **)&pInfo->pDecoded, (void **)&pInfo->pFactorsA, (void **)&pInfo->pFactorsB, (void **)&pInfo->pFactorsC, (void **)&pInfo->pBitsPerPixel,
                         (void **)&pInfo->pShiftABCX, (void **)&pInfo->pColAMin, (void **)&pInfo->pColAMax, (void **)&pInfo->pColBMin, (void **)&pInfo->pColBMax,
                         (void **)&pInfo->pColCMin, (void **)&pInfo->pColCMax, (void **)&pInfo->pBlockIndex };
    void **devp[13] = { (void **)&d.pDecoded, (void **)&d.pFactorsA, (void **)&d.pFactorsB, (void **)&d.pFactorsC, (void **)&d.pBitsPerPixel, (void **)&d.pShiftABCX,
                        (void **)&d.pColAMin, (void **)&d.pColAMax, (void **)&d.pColBMin, (void **)&d.pColBMax, (void **)&d.pColCMin, (void **)&d.pColCMax, (void **)&d.pBlockIndex };
    const bool is8[13] = { false, true, true, true, true, false, false, false, false, false, false, false, false };
    for (int i = 0; i < 13; i++)
    {
      if (!*hostp[i]) return limg_hip_error_ArgumentNull;
      *devp[i] = base + stride * i;
    }
    if ((r = limg_hip_blocked_encode3d_device(c, (const uint32_t *)c->in.p, sizeX, sizeY, hasAlpha, &d, errorFactor, fastBitCrushing, nullptr)) != limg_hip_success) return r;
    if ((r = limg_hip_check_device_status(c)) != limg_hip_success) return r;
    for (int i = 0; i < 13; i++) HIP_TRY(hipMemcpy(*hostp[i], *devp[i], is8[i] ? px : px * 4, hipMemcpyDeviceToHost));
    return limg_hip_success;
  }

  // ---- multi-GPU: RCCL behind the C ABI ----------------------------------------------------------------------
#define NCCL_TRY(expr)                                                                                                    \
  do                                                                                                                      \
  {                                                                                                                       \
    const ncclResult_t e_ = (expr);                                                                                       \
    if (e_ != ncclSuccess)                                                                                                \
    {                                                                                                                     \
      fprintf(stderr, "limg_hip: %s failed: %s (%s:%d)\n", #expr, rccl().GetErrorString(e_), __FILE__, __LINE__);        \
      return limg_hip_error_Generic;                                                                                      \
    }                                                                                                                     \
  } while (0)

  limg_hip_result limg_hip_comm_unique_id(uint8_t *pId)
  {
    if (!pId) return limg_hip_error_ArgumentNull;
    static_assert(sizeof(ncclUniqueId) == LIMG_HIP_COMM_ID_BYTES, "ncclUniqueId size");
    if (!rccl().ok) return limg_hip_error_Generic;
    ncclUniqueId id;
    NCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(pId, &id, sizeof(id));
    return limg_hip_success;
  }

  limg_hip_result limg_hip_comm_init(limg_hip_context *c, const uint8_t *pId, int rank, int worldSize)
  {
    if (!c || !pId) return limg_hip_error_ArgumentNull;
    if (worldSize < 1 || rank < 0 || rank >= worldSize || c->comm) return limg_hip_error_InvalidParameter;
    if (!rccl().ok) return limg_hip_error_Generic;
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, pId, sizeof(id));
    NCCL_TRY(rccl().CommInitRank(&c->comm, worldSize, id, rank));
    c->commRank = rank; c->commWorld = worldSize;
    limg_hip_result r2 = c->commWords.ensure((8 + 2 * (size_t)worldSize) * 8); // [0] own value, [1] chain base, [2..3] own (size, capacity), [8 ...] gathered
    if (r2 != limg_hip_success) return r2;
    HIP_TRY(hipMemset(c->commWords.p, 0, (8 + 2 * (size_t)worldSize) * 8));
    if (!c->devStatus.p)
    { // the sticky status words (look-back timeout, aborted chain) exist from here on: the single-chain entry must not have to allocate on its error path
      if ((r2 = c->devStatus.ensure(16)) != limg_hip_success) return r2;
      HIP_TRY(hipMemset(c->devStatus.p, 0, 16));
    }
    return limg_hip_success;
  }

  // What RCCL itself says about the context's communicator: ncclCommCount / ncclCommUserRank / ncclGetVersion.  For bench lines and logs -- a record that names
  // the ranks RCCL saw cannot be produced by a job that silently ran on fewer.
  limg_hip_result limg_hip_comm_info(limg_hip_context *c, int *pRank, int *pRanks, int *pRcclVersion)
  {
    if (!c) return limg_hip_error_ArgumentNull;
    if (!c->comm) return limg_hip_error_InvalidParameter;
    if (!rccl().ok || !rccl().CommCount || !rccl().CommUserRank || !rccl().GetVersion) return limg_hip_error_Generic;
    int v = 0;
    if (pRanks) NCCL_TRY(rccl().CommCount(c->comm, pRanks));
    if (pRank) NCCL_TRY(rccl().CommUserRank(c->comm, pRank));
    if (pRcclVersion) { NCCL_TRY(rccl().GetVersion(&v)); *pRcclVersion = v; }
    return limg_hip_success;
  }

  limg_hip_result limg_hip_comm_destroy(limg_hip_context *c)
  {
    if (!c) return limg_hip_error_ArgumentNull;
    if (c->comm)
    {
      HIP_TRY(hipSetDevice(c->device));
      HIP_TRY(hipDeviceSynchronize());
      NCCL_TRY(rccl().CommDestroy(c->comm));
      c->comm = nullptr; c->commRank = 0; c->commWorld = 1;
    }
    return limg_hip_success;
  }

  // Offsets of a variable-size gather: piece r starts at the sum of the earlier sizes, each rounded up to 16 bytes (the decoder wants 16-byte aligned streams).
  limg_hip_result limg_hip_host_gather_offsets(const uint64_t *pSizes, int count, uint64_t *pOffsets)
  {
    if (!pSizes || !pOffsets) return limg_hip_error_ArgumentNull;
    if (count < 1) return limg_hip_error_InvalidParameter;
    uint64_t off = 0;
    for (int r = 0; r < count; r++)
    {
      pOffsets[r] = off;
      off += (pSizes[r] + 15ull) & ~15ull;
    }
    pOffsets[count] = off;
    return limg_hip_success;
  }

  // Exclusive prefix of the per-rank dither-call totals = every rank's first call index in the one chain that runs through all strips (rank order = strip order).
  limg_hip_result limg_hip_host_chain_bases(const uint64_t *pCalls, int count, uint64_t *pBases)
  {
    if (!pCalls || !pBases) return limg_hip_error_ArgumentNull;
    if (count < 1) return limg_hip_error_InvalidParameter;
    uint64_t run = 0;
    for (int r = 0; r < count; r++) { pBases[r] = run; run += pCalls[r]; }
    return limg_hip_success;
  }

  limg_hip_result limg_hip_gather_stream(limg_hip_context *c, const uint8_t *pStream, size_t streamBytes, int root, uint8_t *pGathered, size_t capacity, uint64_t *pOffsets,
                                         void *stream)
  {
    if (!c || !pStream) return limg_hip_error_ArgumentNull;
    if (!c->comm) return limg_hip_error_InvalidParameter;
    if (root < 0 || root >= c->commWorld) return limg_hip_error_InvalidParameter;
    const bool isRoot = c->commRank == root;
    if (isRoot && (!pGathered || !pOffsets)) return limg_hip_error_ArgumentNull;
    if (isRoot && ((uintptr_t)pGathered & 15u) != 0) return limg_hip_error_InvalidParameter;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int world = c->commWorld;
    unsigned long long *words = (unsigned long long *)c->commWords.p;
    // 1. every rank learns every size AND the root's capacity: one 16-byte all-gather.  Whether the pieces fit is then decided by every rank from the same
    //    numbers -- a root that alone found its buffer too small would leave its peers' sends unmatched
    const unsigned long long mine[2] = { (unsigned long long)streamBytes, isRoot ? (unsigned long long)capacity : 0ull };
    HIP_TRY(hipMemcpyAsync(words + 2, mine, 16, hipMemcpyHostToDevice, s));
    NCCL_TRY(rccl().AllGather(words + 2, words + 8, 2, ncclUint64, c->comm, s));
    std::vector<uint64_t> pairs(2 * (size_t)world), sizes(world), offs(world + 1);
    HIP_TRY(hipMemcpyAsync(pairs.data(), words + 8, (size_t)world * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int r = 0; r < world; r++) sizes[r] = pairs[2 * (size_t)r];
    limg_hip_host_gather_offsets(sizes.data(), world, offs.data());
    if (offs[world] > pairs[2 * (size_t)root + 1]) return limg_hip_error_OutOfBounds; // on EVERY rank: nothing is posted anywhere
    // 2. exactly the used bytes, point to point: each peer -> root transfer rides one xGMI link
    NCCL_TRY(rccl().GroupStart());
    ncclResult_t posted = ncclSuccess; // a failed post must not leave the group open: close it first, report afterwards
    if (isRoot)
    {
      for (int r = 0; r < world && posted == ncclSuccess; r++)
        if (r != root && sizes[r]) posted = rccl().Recv(pGathered + offs[r], sizes[r], ncclUint8, r, c->comm, s);
    }
    else if (streamBytes)
      posted = rccl().Send(pStream, streamBytes, ncclUint8, root, c->comm, s);
    const ncclResult_t closed = rccl().GroupEnd();
    NCCL_TRY(posted);
    NCCL_TRY(closed);
    if (isRoot)
    {
      if (streamBytes) HIP_TRY(hipMemcpyAsync(pGathered + offs[root], pStream, streamBytes, hipMemcpyDeviceToDevice, s));
      memcpy(pOffsets, offs.data(), (size_t)(world + 1) * 8);
      for (int r = 0; r < world; r++) pOffsets[r] = offs[r];
    }
    return limg_hip_success;
  }

  limg_hip_result limg_hip_encode3d_single_chain_device(limg_hip_context *c, const uint32_t *pIn, size_t sizeX, size_t stripRows, int hasAlpha, const limg_hip_encode3d_info *pInfo,
                                                        uint32_t errorFactor, int fastBitCrushing, size_t blocksBefore, void *stream)
  {
    if (!c) return limg_hip_error_ArgumentNull;
    if (!c->comm) return limg_hip_error_InvalidParameter; // no communicator: nobody is waiting for this rank
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *words = (unsigned long long *)c->commWords.p;
    limg_hip_result r;
    // E step + scan: this strip's dither calls land in words[0] ...  (bad arguments on this rank are a phase-1 failure like any other: see the abort rule)
    if (!pIn || !pInfo) r = limg_hip_error_ArgumentNull;
    else if (TOPT(c, fail_chain_phase1) != 0) r = limg_hip_error_Generic;
    else r = limg_hip_encode3d_chain_device(c, pIn, sizeX, stripRows, hasAlpha, pInfo, errorFactor, fastBitCrushing, 1, (uint64_t *)words, nullptr, blocksBefore, s);
    // Abort rule: a rank whose phase 1 failed must STILL join the exchange -- its peers are (about to be) inside ncclAllGather and would wait for it forever -- and
    // joins it with a poison value instead of a call count.  Every rank's k_chain_base then sees the poison: it hands the F step a poisoned base (k_dither_store
    // returns without storing anything) and raises the context's sticky status word, so that the peers' limg_hip_check_device_status reports the aborted chain;
    // this rank returns its own error.  (The reference's analogue -- row strips on a thread pool, src/limg.cpp:2114-2136 -- cannot half-fail.)
    const limg_hip_result phase1 = r;
    if (phase1 != limg_hip_success) HIP_TRY(hipMemsetAsync(words, 0xFF, 8, s));
    // ... one 8-byte all-gather, the exclusive prefix over the ranks before this one on the device (stream-ordered, no host round trip) ...
    NCCL_TRY(rccl().AllGather(words, words + 8, 1, ncclUint64, c->comm, s));
    launch_chain_base(words + 8, c->commRank, c->commWorld, words + 1, (uint32_t *)c->devStatus.p + 1, s);
    if (phase1 != limg_hip_success) return phase1;
    // ... and the F step indexes the noise stream from there: the 8-GPU result equals the single-threaded reference's (src/limg.cpp:1893, :2110)
    return limg_hip_encode3d_chain_device(c, pIn, sizeX, stripRows, hasAlpha, pInfo, errorFactor, fastBitCrushing, 2, nullptr, (const uint64_t *)(words + 1), blocksBefore, s);
  }
}

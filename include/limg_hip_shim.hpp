// limg_hip_shim.hpp -- header-only C++ shim that re-exposes the reference's own signatures (src/limg.h:27-48, incl. limg_blocked_encode3d_test) on top of
// the C ABI of liblimg_hip.so, so a caller written against limg.h relinks unchanged -- the reference's own src/main.cpp compiles against it as it is
// (tests/test_shim_ref_main.py does exactly that):
//
//     #include "limg_hip_shim.hpp"      // instead of "limg.h"
//     limg_encode3d_test(pIn, sizeX, sizeY, hasAlpha, &info, errorFactor, pThreadPool, fastBitCrushing);
//
// The reference's thread pool only matters for its dither-chain partition (one chain per row strip, src/limg.cpp:2114-2134):
// the shim carries the thread count in an opaque `limg_thread_pool` so the GPU reproduces the same partition.
#ifndef LIMG_HIP_SHIM_HPP
#define LIMG_HIP_SHIM_HPP

#include <inttypes.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <functional>
#include <thread>
#include <vector>

#include "limg_hip.h"

enum limg_result
{
  limg_success = 0,
  limg_error_Generic = 100,
  limg_error_InvalidParameter,
  limg_error_ArgumentNull,
  limg_error_OutOfBounds,
  limg_error_MemoryAllocationFailure,
};

// The whole of src/limg_threading.h:9-17.  On the GPU a pool is only the name of a dither-chain partition, so it owns no threads: tasks given to
// `limg_thread_pool_add` are kept and run by the caller in `limg_thread_pool_await` (upstream's await also drains the queue on the calling thread,
// src/limg_threading.cpp:129-161).
struct limg_thread_pool { size_t threads; std::vector<std::function<void(void)>> tasks; };
inline limg_thread_pool *limg_thread_pool_new(const size_t threads) { return new limg_thread_pool{ threads, {} }; }
inline void limg_thread_pool_destroy(limg_thread_pool **pp) { if (pp && *pp) { delete *pp; *pp = nullptr; } }
inline size_t limg_thread_pool_thread_count(limg_thread_pool *p) { return (p == nullptr || p->threads == 0) ? 1 : p->threads; } // src/limg_threading.cpp:110-116
inline void limg_thread_pool_add(limg_thread_pool *p, const std::function<void(void)> &func) { p->tasks.push_back(func); }
inline void limg_thread_pool_await(limg_thread_pool *p)
{
  while (!p->tasks.empty())
  {
    std::vector<std::function<void(void)>> run;
    run.swap(p->tasks); // a task may add tasks
    for (auto &f : run) f();
  }
}
inline size_t limg_threading_max_threads() { return std::thread::hardware_concurrency(); } // src/limg_threading.cpp:163-166

namespace limg_hip_shim
{
  // pool -> the C ABI's `poolThreads` (0 = no pool = one dither chain; a pool of T threads = 4 T row strips, src/limg.cpp:2114-2134)
  inline int pool_threads(limg_thread_pool *p) { return p ? (int)limg_thread_pool_thread_count(p) : 0; }
}

struct limg_encode3d_info
{
  uint32_t *pDecoded, *pShiftABCX, *pColAMin, *pColAMax, *pColBMin, *pColBMax, *pColCMin, *pColCMax;
  uint8_t *pFactorsA, *pFactorsB, *pFactorsC;
};

struct limg_blocked_encode3d_info // src/limg.h:39-44
{
  uint32_t *pDecoded;
  uint8_t *pFactorsA, *pFactorsB, *pFactorsC, *pBlockError, *pBitsPerPixel;
  uint32_t *pShiftABCX, *pColAMin, *pColAMax, *pColBMin, *pColBMax, *pColCMin, *pColCMax, *pBlockIndex;
};

namespace limg_hip_shim
{
  // The reference keeps no state, so its entry points may be called from any number of threads at once (src/limg.cpp:1890-1891: scratch on the stack).  The shim's
  // callers get the same: ONE process-wide context, created on first use by whichever thread gets there first (a function-local static: initialisation is
  // thread-safe since C++11) and shut down when the process exits (the holder's destructor, registered after the HIP runtime's own exit handlers and therefore
  // run before them); the blocking host-pointer entries of the C ABI that the functions below call serialise on a mutex inside the context
  // (include/limg_hip.h "Thread safety").  Concurrent callers are correct, one encode at a time runs on the GPU.
  struct context_holder
  {
    limg_hip_context *ctx = nullptr;
    context_holder() { if (limg_hip_init(-1, &ctx) != limg_hip_success) ctx = nullptr; }
    ~context_holder() { limg_hip_shutdown(&ctx); }
    context_holder(const context_holder &) = delete;
    context_holder &operator=(const context_holder &) = delete;
  };
  inline limg_hip_context *context()
  {
    static context_holder holder;
    return holder.ctx;
  }

  // Upstream's limg_encode3d_test and limg_blocked_encode3d_test print their bit statistics themselves (src/limg.cpp:2232-2248, PRINT_TEST_OUTPUT is always defined);
  // the library is silent.  A caller that wants upstream's console output switches it on once -- limg_hip_shim::print_stats(true) -- and the two functions below then
  // print the same block, from the counters the GPU reduced (limg_hip_last_stats), after each encode.
  // The flag is the shim's own (an atomic: any thread may flip it); no option of the shared context is touched -- the two functions below ask for the counters of
  // THEIR encode through limg_hip_encode3d_stats / limg_hip_blocked_encode3d_stats, which encode and fetch under the context's mutex, so a thread always prints its own
  // call's block, like upstream (counters on the call's stack, src/limg.cpp:1975-1976).
  inline std::atomic<bool> &stats_flag() { static std::atomic<bool> on(false); return on; }
  inline void print_stats(const bool on) { stats_flag().store(on); }
  inline void print_stats_block(const uint64_t *a, const uint64_t pixels)
  {
    if (pixels == 0) return;
    const double t = (double)pixels;
    // one buffer, one write: blocks of concurrent callers do not interleave line by line
    char buf[1024];
    int n = snprintf(buf, sizeof(buf), "\nAverage Block Bits: %5.3f (A: %5.3f | B: %5.3f | C: %5.3f)\n\n", (a[0] + a[1] + a[2]) / t, a[0] / t, a[1] / t, a[2] / t);
    for (size_t i = 0; i < 9; i++) n += snprintf(buf + n, sizeof(buf) - (size_t)n, " %" PRIu64 " bit   ", (uint64_t)(8 - i));
    for (size_t f = 0; f < 3; f++)
    {
      n += snprintf(buf + n, sizeof(buf) - (size_t)n, "\n");
      for (size_t j = 0; j < 9; j++) n += snprintf(buf + n, sizeof(buf) - (size_t)n, "%7.4f  ", a[3 + f * 9 + j] * 100.0 / t);
    }
    n += snprintf(buf + n, sizeof(buf) - (size_t)n, "\n\n");
    fwrite(buf, 1, (size_t)n, stdout);
  }
}

inline limg_result limg_encode3d_test(const uint32_t *pIn, const size_t sizeX, const size_t sizeY, const bool hasAlpha, limg_encode3d_info *pInfo, const uint32_t errorFactor,
                                      limg_thread_pool *pThreadPool, const bool fastBitCrushing)
{
  static_assert(sizeof(limg_encode3d_info) == sizeof(limg_hip_encode3d_info), "layout");
  limg_hip_context *c = limg_hip_shim::context();
  if (!c) return limg_error_Generic;
  if (!limg_hip_shim::stats_flag().load())
    return (limg_result)limg_hip_encode3d(c, pIn, sizeX, sizeY, hasAlpha ? 1 : 0, reinterpret_cast<limg_hip_encode3d_info *>(pInfo), errorFactor,
                                          limg_hip_shim::pool_threads(pThreadPool), fastBitCrushing ? 1 : 0);
  uint64_t counters[30], pixels = 0;
  const limg_result r = (limg_result)limg_hip_encode3d_stats(c, pIn, sizeX, sizeY, hasAlpha ? 1 : 0, reinterpret_cast<limg_hip_encode3d_info *>(pInfo), errorFactor,
                                                             limg_hip_shim::pool_threads(pThreadPool), fastBitCrushing ? 1 : 0, counters, &pixels);
  if (r == limg_success) limg_hip_shim::print_stats_block(counters, pixels);
  return r;
}

inline limg_result limg_encode3d_test_perf(const uint32_t *pIn, const size_t sizeX, const size_t sizeY, const bool hasAlpha, const uint32_t errorFactor, limg_thread_pool *pThreadPool,
                                           const bool fastBitCrushing)
{
  limg_hip_context *c = limg_hip_shim::context();
  if (!c) return limg_error_Generic;
  return (limg_result)limg_hip_encode3d_perf(c, pIn, sizeX, sizeY, hasAlpha ? 1 : 0, errorFactor, limg_hip_shim::pool_threads(pThreadPool), fastBitCrushing ? 1 : 0);
}

// src/limg.h:46.  The pool only splits upstream's first pass and cannot change the result; it is accepted and ignored.
inline limg_result limg_blocked_encode3d_test(const uint32_t *pIn, const size_t sizeX, const size_t sizeY, const bool hasAlpha, limg_blocked_encode3d_info *pInfo, const uint32_t errorFactor,
                                              limg_thread_pool * /* pThreadPool */, const bool fastBitCrushing)
{
  static_assert(sizeof(limg_blocked_encode3d_info) == sizeof(limg_hip_blocked_encode3d_info), "layout");
  limg_hip_context *c = limg_hip_shim::context();
  if (!c) return limg_error_Generic;
  if (!limg_hip_shim::stats_flag().load())
    return (limg_result)limg_hip_blocked_encode3d(c, pIn, sizeX, sizeY, hasAlpha ? 1 : 0, reinterpret_cast<limg_hip_blocked_encode3d_info *>(pInfo), errorFactor, fastBitCrushing ? 1 : 0);
  uint64_t counters[30], pixels = 0;
  const limg_result r = (limg_result)limg_hip_blocked_encode3d_stats(c, pIn, sizeX, sizeY, hasAlpha ? 1 : 0, reinterpret_cast<limg_hip_blocked_encode3d_info *>(pInfo), errorFactor,
                                                                     fastBitCrushing ? 1 : 0, counters, &pixels);
  if (r == limg_success) limg_hip_shim::print_stats_block(counters, pixels);
  return r;
}

inline double limg_compare(const uint32_t *pImageA, const uint32_t *pImageB, const size_t sizeX, const size_t sizeY, const bool hasAlpha, double *pMeanSquaredError,
                           double *pMaxPossibleSquaredError)
{
  limg_hip_context *c = limg_hip_shim::context();
  if (!c) return 0.0;
  return limg_hip_compare(c, pImageA, pImageB, sizeX, sizeY, hasAlpha ? 1 : 0, pMeanSquaredError, pMaxPossibleSquaredError);
}

// `limg_encode` / `limg_decode`: named by the north-star, absent upstream (src/limg.h declares no such pair).  Build-defined wrappers over the
// compact "LMG3" stream of limg_hip.h: limg_decode(limg_encode(image)) == the pDecoded plane of limg_encode3d_test, bit for bit.
inline size_t limg_encode_bound(const size_t sizeX, const size_t sizeY) { return limg_hip_stream_bound(sizeX, sizeY); }

inline limg_result limg_encode(const uint32_t *pIn, const size_t sizeX, const size_t sizeY, const bool hasAlpha, uint8_t *pOut, const size_t outCapacity, size_t *pOutSize,
                               const uint32_t errorFactor = 100, limg_thread_pool *pThreadPool = nullptr, const bool fastBitCrushing = true)
{
  limg_hip_context *c = limg_hip_shim::context();
  if (!c) return limg_error_Generic;
  return (limg_result)limg_hip_encode_stream(c, pIn, sizeX, sizeY, hasAlpha ? 1 : 0, pOut, outCapacity, pOutSize, errorFactor, limg_hip_shim::pool_threads(pThreadPool),
                                             fastBitCrushing ? 1 : 0);
}

inline limg_result limg_decode_info(const uint8_t *pIn, const size_t size, size_t *pSizeX, size_t *pSizeY, bool *pHasAlpha)
{
  int alpha = 0;
  const limg_result r = (limg_result)limg_hip_stream_info(pIn, size, pSizeX, pSizeY, &alpha, nullptr);
  if (pHasAlpha) *pHasAlpha = alpha != 0;
  return r;
}

inline limg_result limg_decode(const uint8_t *pIn, const size_t size, uint32_t *pOut, const size_t outPixelCapacity)
{
  limg_hip_context *c = limg_hip_shim::context();
  if (!c) return limg_error_Generic;
  return (limg_result)limg_hip_decode_stream(c, pIn, size, pOut, outPixelCapacity);
}

#endif // LIMG_HIP_SHIM_HPP

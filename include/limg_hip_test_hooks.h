/* limg_hip_test_hooks.h -- fault injection and A/B knobs of liblimg_hip_TEST.so.  NOT part of the product.
 *
 * The reference has no such knobs (src/limg.h:27-48) and neither does liblimg_hip.so: its kernels carry none of the compares these hooks need and it does not
 * export limg_hip_set_test_options.  limg_amd/build.py compiles the same sources a second time with -DLIMG_HIP_TEST_HOOKS into limg_amd/liblimg_hip_test.so;
 * the test suite (tests/conftest.py) and A/B runs of the bench (LIMG_HIP_LIB=limg_amd/liblimg_hip_test.so) load that build, everything else -- bench.py, the C++
 * shim, the CLI, __graft_entry__.smoke() -- loads the plain library.  Everything include/limg_hip.h declares is exported by both.
 */
#ifndef LIMG_HIP_TEST_HOOKS_H
#define LIMG_HIP_TEST_HOOKS_H

#include "limg_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* All members 0 = the product's behaviour.  Versioned by its size like limg_hip_options. */
typedef struct limg_hip_test_options
{
  uint32_t struct_size;
  int32_t record_limit;       /* blocks with a record value of magnitude >= this take the generic 32-bit trial (default 2701 -- up to 2700 the packed 16-bit trial is
                                 exact by construction; a fit of byte pixels stays below 2041); 1 sends every block through it */
  int32_t batch_chunk;        /* limg_hip_encode3d_batch_device puts at most this many images into one launch pair (default: as many as 1 GiB of per-block scratch holds) */
  int32_t wg_per_cu;          /* A/B: workgroups per CU of the persistent kernel's launch, 1 .. its launch bound (6; values above are ignored) */
  int32_t whole_image_ragged; /* non-0: an image whose width is whole 8x8 blocks but whose last block row is partial goes through the whole-image ragged path (host
                                 chain walk over every dither call) instead of fast path + last row; same planes either way */
  int32_t pipeline;           /* A/B knobs of the sub-batch pipeline (see limg_hip_api.hip) */
  int32_t fail_chain_phase1;  /* non-0: limg_hip_encode3d_single_chain_device behaves as if this rank's E step had failed (abort rule in limg_hip.h) */
  int32_t blocked_no_bound;   /* non-0: the merged-block encoder's similarity kernel evaluates the 27-colour loop for every pair its early exits leave open, without
                                 the certain-match / certain-failure bounds in front of it (limg_hip_blocked.hip).  Same bits either way */
  int32_t lookback_spins;     /* N > 0: bound of one look-back wait of the persistent kernel, in polls (the product's bound is 2^22, seconds) */
  int32_t base_error_strip;   /* N > 0: work strip N - 1 of the persistent kernel dithers from a chain position that is off by one dither call (every other strip is
                                 unaffected): the smallest possible look-back error, which the full-size reference hashes must catch (tests/test_gpu_fullsize.py) */
  int32_t skip_publish_strip; /* N > 0: work strip N - 1 of the persistent kernel never publishes its dither-call count, i.e. the look-back of every later strip of its
                                 chain times out (see limg_hip_check_device_status: such strips store nothing that depends on the chain) */
  int32_t blocked_no_order;   /* A/B, non-0: the merged-block encoder's per-rectangle launches take the rectangles in creation order instead of large-first
                                 (k_blocked_order, limg_hip_blocked.hip).  Same planes either way */
  int32_t blocked_no_vec_store; /* A/B, non-0: the merged-block encoder's store kernel keeps one pixel per lane (what images with partial edge blocks always use) */
} limg_hip_test_options;

void limg_hip_default_test_options_sized(limg_hip_test_options *pOptions, size_t structSize);
static inline void limg_hip_default_test_options(limg_hip_test_options *pOptions) { limg_hip_default_test_options_sized(pOptions, sizeof(limg_hip_test_options)); }
limg_hip_result limg_hip_set_test_options(limg_hip_context *pCtx, const limg_hip_test_options *pOptions);

#ifdef __cplusplus
}
#endif
#endif

// TEST INFRASTRUCTURE -- C-ABI probe harness around the *real* reference implementation.
//
// This translation unit is compiled by oracle/build_ref.sh together with a throw-away, mechanically patched copy
// of /root/reference/src (see that script).  It `#include`s the reference's limg.cpp so that the `static` block
// functions of the hot path are reachable, and re-exports them with plain C signatures so that tests and the golden
// fixture generator can drive them through ctypes.  Nothing here restates any arithmetic: every function below is a
// thin call-through (ref_encode3d_forced_shift sequences the reference's own block functions with the search left out: see there).  Only tests/, tools/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load the resulting oracle/_ref/liblimg_ref.so.

#include "limg.cpp" // reference: src/limg.cpp (patched temp copy on the include path)

#include <unistd.h>
#include <fcntl.h>

namespace
{
  // the reference prints bit statistics to stdout (src/limg.cpp:2232-2258); silence them while a call runs.
  bool g_keep_stdout = false; // ref_keep_stdout(1): let the statistics through (tools/make_golden_stats.py captures them as golden text)

  struct stdout_silencer
  {
    int saved;
    stdout_silencer()
    {
      saved = -1;
      if (g_keep_stdout) return;
      fflush(stdout);
      saved = dup(1);
      const int devnull = open("/dev/null", O_WRONLY);
      dup2(devnull, 1);
      close(devnull);
    }
    ~stdout_silencer()
    {
      fflush(stdout);
      if (saved < 0) return;
      dup2(saved, 1);
      close(saved);
    }
  };

  void set_features(const int dither_mode)
  {
    _DetectCPUFeatures();
    // dither_mode: 0 = whatever the CPU offers (AES-NI here), 1 = force the PCG fallback (src/limg.cpp:881-887)
    static bool aes_native = aesNiSupported;
    aesNiSupported = (dither_mode == 1) ? false : aes_native;
  }

  void fill_ctx(limg_encode_context &ctx, const uint32_t errorFactor, const bool fast, const bool hasAlpha)
  {
    // mirrors the threshold/flag setup of src/limg.cpp:2179-2221 by *calling nothing*: the reference has no
    // separate function for it, so the probes below need the same numbers.  Kept literal.
    memset(&ctx, 0, sizeof(ctx));
    ctx.hasAlpha = hasAlpha;
    ctx.maxPixelBitCrushError = 0x6 * (errorFactor / 2) * 7;
    ctx.maxBlockBitCrushError = 0x4 * (errorFactor / 2) * 7;
    ctx.ditheringEnabled = true;
    ctx.fastBitCrush = fast;
    ctx.guessCrush = true;
    ctx.crushBits = errorFactor != 0;
    ctx.errorPixelRetainingBitCrush = !fast;
    ctx.coarseFineBitCrush = fast;
  }

  // BASELINE config 3's forced-shift sweep has no switch upstream ("the reference has no bits/factor knob"): this drives the reference's OWN block
  // functions in the order limg_encode3d_test_y_range calls them (src/limg.cpp:1895-2100) with the search (a10-a12, :1922-1945) left out and `shift` a constant:
  // gather -> sums -> fit -> colour-error state -> factor bytes -> limg_encode_dither per factor whose shift is neither 0 nor 8 (:1947-1958) -> the factor planes
  // (byte << shift, :2064-2074) -> limg_decode_block_from_factors_3d.  Only the planes that depend on the shift are produced (pDecoded, pFactorsA/B/C); the six
  // colour planes do not (they equal those of the adaptive encode of the same image) and pShiftABCX is one constant.  One dither chain (pThreadPool == nullptr).
  template <size_t channels>
  void forced_rows(limg_encode_context &ctx, const uint32_t *pIn, const size_t sizeX, const size_t sizeY, const uint8_t shift[3], uint32_t *pDecoded, uint8_t *pFA, uint8_t *pFB, uint8_t *pFC)
  {
    float scratch[limg_MinBlockSize * limg_MinBlockSize * 4];
    uint32_t px[limg_MinBlockSize * limg_MinBlockSize];
    uint64_t chain = 0xCA7F00D15BADF00D; // src/limg.cpp:1893
    uint8_t *f[3] = { reinterpret_cast<uint8_t *>(scratch), nullptr, nullptr };
    uint8_t *planes[3] = { pFA, pFB, pFC };
    for (size_t by = 0; by < sizeY; by += limg_MinBlockSize)
      for (size_t bx = 0; bx < sizeX; bx += limg_MinBlockSize)
      {
        const size_t w = limgMin(sizeX - bx, limg_MinBlockSize), h = limgMin(sizeY - by, limg_MinBlockSize), n = w * h;
        for (size_t r = 0; r < h; r++) memcpy(px + r * w, pIn + (by + r) * sizeX + bx, w * sizeof(uint32_t));
        limg_encode_decomposition_state st;
        limg_encode_sum_to_decomposition_state<channels>(&ctx, px, n, st);
        limg_encode_3d_output<channels> rec;
        limg_encode_get_block_factors_accurate_from_state_3d<channels>(&ctx, px, n, rec, st, scratch);
        limg_color_error_state_3d<channels> ces;
        limg_init_color_error_state_3d<channels>(rec, ces);
        f[1] = f[0] + n; f[2] = f[1] + n;
        limg_color_error_state_3d_get_all_factors(&ctx, rec, ces, px, n, f[0], f[1], f[2]);
        for (int k = 0; k < 3; k++)
          if (shift[k] != 0 && shift[k] != 8) chain = limg_encode_dither(shift[k], n, chain, f[k]);
        for (int k = 0; k < 3; k++)
          for (size_t r = 0; r < h; r++)
            for (size_t c = 0; c < w; c++) planes[k][(by + r) * sizeX + bx + c] = (uint8_t)(f[k][r * w + c] << shift[k]);
        limg_decode_block_from_factors_3d<channels>(pDecoded + by * sizeX + bx, sizeX, w, h, f[0], f[1], f[2], rec, shift);
      }
  }

}

extern "C"
{
  void ref_keep_stdout(int keep) { g_keep_stdout = keep != 0; }

  int ref_encode3d(const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, uint32_t **p32 /* 8 planes, limg.h:31 order */, uint8_t **p8 /* A,B,C */, uint32_t errorFactor, int poolThreads, int fast, int dither_mode)
  {
    set_features(dither_mode);
    limg_encode3d_info info;
    info.pDecoded = p32[0]; info.pShiftABCX = p32[1]; info.pColAMin = p32[2]; info.pColAMax = p32[3];
    info.pColBMin = p32[4]; info.pColBMax = p32[5]; info.pColCMin = p32[6]; info.pColCMax = p32[7];
    info.pFactorsA = p8[0]; info.pFactorsB = p8[1]; info.pFactorsC = p8[2];
    limg_thread_pool *pPool = poolThreads > 0 ? limg_thread_pool_new((size_t)poolThreads) : nullptr;
    int r;
    {
      stdout_silencer s;
      r = (int)limg_encode3d_test(pIn, sizeX, sizeY, hasAlpha != 0, &info, errorFactor, pPool, fast != 0);
    }
    if (pPool) limg_thread_pool_destroy(&pPool);
    return r;
  }

  // limg_blocked_encode3d_test (src/limg.h:46).  planes in the member order of limg_blocked_encode3d_info (src/limg.h:39-44).
  int ref_blocked_encode3d(const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, void **planes /* 14 */, uint32_t errorFactor, int poolThreads, int fast, int dither_mode)
  {
    set_features(dither_mode);
    limg_blocked_encode3d_info info;
    info.pDecoded = (uint32_t *)planes[0];
    info.pFactorsA = (uint8_t *)planes[1]; info.pFactorsB = (uint8_t *)planes[2]; info.pFactorsC = (uint8_t *)planes[3];
    info.pBlockError = (uint8_t *)planes[4]; info.pBitsPerPixel = (uint8_t *)planes[5];
    info.pShiftABCX = (uint32_t *)planes[6]; info.pColAMin = (uint32_t *)planes[7]; info.pColAMax = (uint32_t *)planes[8];
    info.pColBMin = (uint32_t *)planes[9]; info.pColBMax = (uint32_t *)planes[10]; info.pColCMin = (uint32_t *)planes[11]; info.pColCMax = (uint32_t *)planes[12];
    info.pBlockIndex = (uint32_t *)planes[13];
    limg_thread_pool *pPool = poolThreads > 0 ? limg_thread_pool_new((size_t)poolThreads) : nullptr;
    int r;
    {
      stdout_silencer s;
      r = (int)limg_blocked_encode3d_test(pIn, sizeX, sizeY, hasAlpha != 0, &info, errorFactor, pPool, fast != 0);
    }
    if (pPool) limg_thread_pool_destroy(&pPool);
    return r;
  }

  // limg_encode_3d_matches (src/limg.cpp:1264-1268): records in the reference's own <3>/<4> layouts
  int ref_blocked_matches(int channels, void *a, const void *b)
  {
    limg_encode_context ctx;
    memset(&ctx, 0, sizeof(ctx));
    if (channels == 4) return limg_encode_3d_matches<4>(&ctx, *reinterpret_cast<limg_encode_3d_output<4> *>(a), *reinterpret_cast<const limg_encode_3d_output<4> *>(b)) ? 1 : 0;
    return limg_encode_3d_matches<3>(&ctx, *reinterpret_cast<limg_encode_3d_output<3> *>(a), *reinterpret_cast<const limg_encode_3d_output<3> *>(b)) ? 1 : 0;
  }

  int ref_encode3d_perf(const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, uint32_t errorFactor, int poolThreads, int fast, int dither_mode)
  {
    set_features(dither_mode);
    limg_thread_pool *pPool = poolThreads > 0 ? limg_thread_pool_new((size_t)poolThreads) : nullptr;
    const int r = (int)limg_encode3d_test_perf(pIn, sizeX, sizeY, hasAlpha != 0, errorFactor, pPool, fast != 0);
    if (pPool) limg_thread_pool_destroy(&pPool);
    return r;
  }

  int ref_encode3d_forced_shift(const uint32_t *pIn, size_t sizeX, size_t sizeY, int hasAlpha, uint32_t *pDecoded, uint8_t **p8 /* A,B,C */, const uint8_t *shift /* 3, each 0..8 */)
  {
    set_features(0);
    limg_encode_context ctx; fill_ctx(ctx, 100, true, hasAlpha != 0);
    ctx.pSourceImage = pIn; ctx.sizeX = sizeX; ctx.sizeY = sizeY;
    if (hasAlpha) forced_rows<4>(ctx, pIn, sizeX, sizeY, shift, pDecoded, p8[0], p8[1], p8[2]);
    else forced_rows<3>(ctx, pIn, sizeX, sizeY, shift, pDecoded, p8[0], p8[1], p8[2]);
    return 0;
  }

  double ref_compare(const uint32_t *pA, const uint32_t *pB, size_t sizeX, size_t sizeY, int hasAlpha, double *pMse, double *pMax)
  {
    return limg_compare(pA, pB, sizeX, sizeY, hasAlpha != 0, pMse, pMax);
  }

  // ---- per-block probes (static functions of the hot path) -------------------------------------------------------

  // a4 + a5/a6: channel sums + direction fit; `out` receives limg_encode_3d_output<channels> (48 / 64 bytes).
  void ref_block_fit(const uint32_t *pPixels, size_t size, int channels, void *out)
  {
    set_features(0);
    limg_encode_context ctx; fill_ctx(ctx, 100, true, channels == 4);
    float scratch[limg_MinBlockSize * limg_MinBlockSize * 4];
    limg_encode_decomposition_state st;
    if (channels == 4)
    {
      limg_encode_sum_to_decomposition_state<4>(&ctx, pPixels, size, st);
      limg_encode_get_block_factors_accurate_from_state_3d<4>(&ctx, pPixels, size, *reinterpret_cast<limg_encode_3d_output<4> *>(out), st, scratch);
    }
    else
    {
      limg_encode_sum_to_decomposition_state<3>(&ctx, pPixels, size, st);
      limg_encode_get_block_factors_accurate_from_state_3d<3>(&ctx, pPixels, size, *reinterpret_cast<limg_encode_3d_output<3> *>(out), st, scratch);
    }
  }

  // a7 + a8: per-pixel factor bytes from a record.
  void ref_block_factors(const uint32_t *pPixels, size_t size, int channels, const void *rec, uint8_t *pA, uint8_t *pB, uint8_t *pC)
  {
    set_features(0);
    limg_encode_context ctx; fill_ctx(ctx, 100, true, channels == 4);
    if (channels == 4)
    {
      const auto &d = *reinterpret_cast<const limg_encode_3d_output<4> *>(rec);
      limg_color_error_state_3d<4> ces; limg_init_color_error_state_3d<4>(d, ces);
      limg_color_error_state_3d_get_all_factors<4>(&ctx, d, ces, pPixels, size, pA, pB, pC);
    }
    else
    {
      const auto &d = *reinterpret_cast<const limg_encode_3d_output<3> *>(rec);
      limg_color_error_state_3d<3> ces; limg_init_color_error_state_3d<3>(d, ces);
      limg_color_error_state_3d_get_all_factors<3>(&ctx, d, ces, pPixels, size, pA, pB, pC);
    }
  }

  // a9: one bit-crush trial.
  int ref_block_trial(const uint32_t *pPixels, size_t size, int channels, const void *rec, const uint8_t *pA, const uint8_t *pB, const uint8_t *pC, const uint8_t *shift, uint32_t errorFactor, uint64_t *pBlockError)
  {
    set_features(0);
    limg_encode_context ctx; fill_ctx(ctx, errorFactor, true, channels == 4);
    size_t be = 0;
    bool ok;
    if (channels == 4)
      ok = limg_encode_try_bit_crush_block_3d<4>(&ctx, pPixels, size, *reinterpret_cast<const limg_encode_3d_output<4> *>(rec), pA, pB, pC, shift, &be);
    else
      ok = limg_encode_try_bit_crush_block_3d<3>(&ctx, pPixels, size, *reinterpret_cast<const limg_encode_3d_output<3> *>(rec), pA, pB, pC, shift, &be);
    *pBlockError = be;
    return ok ? 1 : 0;
  }

  // a10 + a11 (+ a12 when fast == 0): the shift search exactly as src/limg.cpp:1922-1945 dispatches it.
  void ref_block_search(const uint32_t *pPixels, size_t size, int channels, const void *rec, uint8_t *pA, uint8_t *pB, uint8_t *pC, uint32_t errorFactor, int fast, uint8_t *shift)
  {
    set_features(0);
    limg_encode_context ctx; fill_ctx(ctx, errorFactor, fast != 0, channels == 4);
    shift[0] = shift[1] = shift[2] = 0;
    if (!ctx.crushBits) return;
    if (channels == 4)
    {
      const auto &d = *reinterpret_cast<const limg_encode_3d_output<4> *>(rec);
      if (ctx.errorPixelRetainingBitCrush)
        limg_encode_find_shift_for_block_error_pixel_preference_3d<4>(&ctx, pPixels, size, d, pA, pB, pC, shift);
      else
      {
        size_t minBlockError = (size_t)-1;
        limg_encode_guess_shift_for_block_3d<4>(&ctx, pPixels, size, d, pA, pB, pC, shift, &minBlockError);
        limg_encode_find_shift_for_block_stepwise_3d<4>(&ctx, pPixels, size, d, pA, pB, pC, shift, minBlockError);
      }
    }
    else
    {
      const auto &d = *reinterpret_cast<const limg_encode_3d_output<3> *>(rec);
      if (ctx.errorPixelRetainingBitCrush)
        limg_encode_find_shift_for_block_error_pixel_preference_3d<3>(&ctx, pPixels, size, d, pA, pB, pC, shift);
      else
      {
        size_t minBlockError = (size_t)-1;
        limg_encode_guess_shift_for_block_3d<3>(&ctx, pPixels, size, d, pA, pB, pC, shift, &minBlockError);
        limg_encode_find_shift_for_block_stepwise_3d<3>(&ctx, pPixels, size, d, pA, pB, pC, shift, minBlockError);
      }
    }
  }

  // a13 / a14: one dither call; returns the new chain value.
  uint64_t ref_dither(int shift, size_t size, uint64_t hash, uint8_t *pFactors, int dither_mode)
  {
    set_features(dither_mode);
    return limg_encode_dither((uint8_t)shift, size, hash, pFactors);
  }

  // a16: integer decode of one block.
  void ref_block_decode(uint32_t *pOut, size_t strideX, size_t rx, size_t ry, int channels, const void *rec, const uint8_t *pA, const uint8_t *pB, const uint8_t *pC, const uint8_t *shift)
  {
    set_features(0);
    if (channels == 4)
      limg_decode_block_from_factors_3d<4>(pOut, strideX, rx, ry, pA, pB, pC, *reinterpret_cast<const limg_encode_3d_output<4> *>(rec), shift);
    else
      limg_decode_block_from_factors_3d<3>(pOut, strideX, rx, ry, pA, pB, pC, *reinterpret_cast<const limg_encode_3d_output<3> *>(rec), shift);
  }

  // the x86 approximation instruction the float stage leans on (src/limg_factorization.h:426,...): raw access for
  // tools/make_rsqrt_table.py, which captures this CPU's RSQRTPS behaviour.
  void ref_rsqrtps(const float *in, float *out, size_t n)
  {
    for (size_t i = 0; i < n; i++)
      out[i] = _mm_cvtss_f32(_mm_rsqrt_ps(_mm_set1_ps(in[i])));
  }

  int ref_record_size(int channels) { return channels == 4 ? (int)sizeof(limg_encode_3d_output<4>) : (int)sizeof(limg_encode_3d_output<3>); }
}

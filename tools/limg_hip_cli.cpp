// limg_hip_cli -- command-line counterpart of the reference's tool (src/main.cpp) on top of liblimg_hip.so.
//
// Written against include/limg_hip_shim.hpp, i.e. against the reference's own function names, so it doubles as the proof that a
// caller of limg.h relinks unchanged.  Same argument set and the same report lines as src/main.cpp:75-86, :278-347:
//
//   limg_hip_cli <InputFile> [--no-output] [--error-factor <Factor>] [--accurate-bit-crushing] [--single-thread]
//   limg_hip_cli -- [--count <Count>] [...] -- <list of files>          (benchmark: limg_encode3d_test_perf, nothing written)
//
// Like upstream, the single-file mode runs the merged-block encoder `limg_blocked_encode3d_test` (src/main.cpp:255) and the list /
// benchmark modes run `limg_encode3d_test_perf`.  Differences: images are read by a small built-in PNG (zlib) / TGA / PPM reader instead
// of stb_image; TGAs are written uncompressed (upstream's stb writer uses RLE); the block-error plane upstream allocates but never
// fills is not written.  Extras: `--fixed-blocks` (single-file mode with `limg_encode3d_test`, fixed 8x8 blocks, instead), `--threads <T>`
// (size of the pool whose strip partition the 8x8 path reproduces; default: hardware threads, like limg_threading_max_threads),
// `--out-dir <dir>`, `--stream <file>` (also write the compact LMG3 stream of the 8x8 path and verify that it decodes to that path's image),
// and the extra mode `limg_hip_cli --decode <file.lmg3> [<out.tga>]`.
#include <inttypes.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

#include <string>
#include <thread>
#include <vector>

#include "limg_hip_shim.hpp"

#define FAIL(code, ...) do { printf(__VA_ARGS__); exit(code); } while (0)

static int64_t CurrentTimeNs()
{
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

// ---- image readers: 8-bit PNG (non-interlaced; gray, gray+alpha, RGB, RGBA, palette), uncompressed TGA, binary PPM -------------
static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static bool read_file(const char *path, std::vector<uint8_t> &out)
{
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? (size_t)n : 0);
  const bool ok = n >= 0 && fread(out.data(), 1, out.size(), f) == out.size();
  fclose(f);
  return ok;
}

static bool load_png(const std::vector<uint8_t> &file, std::vector<uint32_t> &px, size_t &w, size_t &h, int &channels)
{
  static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
  if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) return false;
  size_t pos = 8;
  uint32_t width = 0, height = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  while (pos + 12 <= file.size())
  {
    const uint32_t len = be32(&file[pos]);
    const char *type = (const char *)&file[pos + 4];
    if (pos + 12 + len > file.size()) return false;
    const uint8_t *data = &file[pos + 8];
    if (!memcmp(type, "IHDR", 4) && len >= 13) { width = be32(data); height = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
    else if (!memcmp(type, "IEND", 4)) break;
    pos += 12 + len;
  }
  if (!width || !height || depth != 8 || interlace != 0) return false;
  int spp;
  switch (ctype) { case 0: spp = 1; break; case 2: spp = 3; break; case 3: spp = 1; break; case 4: spp = 2; break; case 6: spp = 4; break; default: return false; }
  const size_t stride = (size_t)width * spp;
  std::vector<uint8_t> raw((stride + 1) * height);
  uLongf rawLen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return false;
  std::vector<uint8_t> img(stride * height);
  for (size_t y = 0; y < height; y++)
  {
    const uint8_t ft = raw[y * (stride + 1)];
    const uint8_t *src = &raw[y * (stride + 1) + 1];
    uint8_t *dst = &img[y * stride];
    const uint8_t *up = y ? dst - stride : nullptr;
    for (size_t i = 0; i < stride; i++)
    {
      const int a = i >= (size_t)spp ? dst[i - spp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)spp) ? up[i - spp] : 0;
      int pred = 0;
      switch (ft)
      {
      case 0: pred = 0; break;
      case 1: pred = a; break;
      case 2: pred = b; break;
      case 3: pred = (a + b) >> 1; break;
      case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
      default: return false;
      }
      dst[i] = (uint8_t)(src[i] + pred);
    }
  }
  w = width; h = height;
  // upstream: hasAlpha = (stb_image's channel count of the file == 4), src/main.cpp:194 -- RGBA, or a palette with tRNS; gray+alpha counts 2
  channels = (ctype == 6 || (ctype == 3 && !trns.empty())) ? 4 : 3;
  px.resize((size_t)width * height);
  for (size_t i = 0; i < px.size(); i++)
  {
    const uint8_t *s = &img[i * spp];
    uint32_t r, g, b, a = 255;
    switch (ctype)
    {
    case 0: r = g = b = s[0]; break;
    case 4: r = g = b = s[0]; a = s[1]; break;
    case 2: r = s[0]; g = s[1]; b = s[2]; break;
    case 6: r = s[0]; g = s[1]; b = s[2]; a = s[3]; break;
    default:
      if ((size_t)s[0] * 3 + 2 >= plte.size()) return false;
      r = plte[s[0] * 3]; g = plte[s[0] * 3 + 1]; b = plte[s[0] * 3 + 2];
      if (s[0] < trns.size()) a = trns[s[0]];
      break;
    }
    px[i] = r | (g << 8) | (b << 16) | (a << 24);
  }
  return true;
}

static bool load_tga(const std::vector<uint8_t> &f, std::vector<uint32_t> &px, size_t &w, size_t &h, int &channels)
{
  if (f.size() < 18 || f[1] != 0 || (f[2] != 2 && f[2] != 3)) return false;
  const size_t width = f[12] | (f[13] << 8), height = f[14] | (f[15] << 8), bpp = f[16] / 8;
  if ((bpp != 1 && bpp != 3 && bpp != 4) || f.size() < 18 + f[0] + width * height * bpp) return false;
  const uint8_t *s = &f[18 + f[0]];
  const bool topDown = (f[17] & 0x20) != 0;
  px.resize(width * height);
  for (size_t y = 0; y < height; y++)
    for (size_t x = 0; x < width; x++, s += bpp)
    {
      const uint32_t b = s[0], g = bpp > 1 ? s[1] : b, r = bpp > 1 ? s[2] : b, a = bpp == 4 ? s[3] : 255;
      px[(topDown ? y : height - 1 - y) * width + x] = r | (g << 8) | (b << 16) | (a << 24);
    }
  w = width; h = height; channels = bpp == 4 ? 4 : 3;
  return true;
}

static bool load_ppm(const std::vector<uint8_t> &f, std::vector<uint32_t> &px, size_t &w, size_t &h, int &channels)
{
  if (f.size() < 9 || f[0] != 'P' || f[1] != '6') return false;
  size_t pos = 2, vals[3], n = 0;
  while (n < 3 && pos < f.size())
  {
    while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\t')) pos++;
    if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') pos++; continue; }
    size_t v = 0;
    while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') v = v * 10 + (f[pos++] - '0');
    vals[n++] = v;
  }
  pos++;
  if (n != 3 || vals[2] != 255 || f.size() < pos + vals[0] * vals[1] * 3) return false;
  w = vals[0]; h = vals[1]; channels = 3;
  px.resize(w * h);
  for (size_t i = 0; i < px.size(); i++) px[i] = f[pos + 3 * i] | (f[pos + 3 * i + 1] << 8) | (f[pos + 3 * i + 2] << 16) | 0xFF000000u;
  return true;
}

static bool load_image(const char *path, std::vector<uint32_t> &px, size_t &w, size_t &h, int &channels)
{
  std::vector<uint8_t> f;
  if (!read_file(path, f)) return false;
  return load_png(f, px, w, h, channels) || load_ppm(f, px, w, h, channels) || load_tga(f, px, w, h, channels);
}

// uncompressed, top-down TGA; comp = 1 (gray) or 4 (RGBA in memory, BGRA in the file, like stbi_write_tga)
static bool write_tga(const std::string &path, size_t w, size_t h, int comp, const void *data)
{
  FILE *f = fopen(path.c_str(), "wb");
  if (!f) return false;
  uint8_t hdr[18] = { 0 };
  hdr[2] = comp == 1 ? 3 : 2;
  hdr[12] = (uint8_t)w; hdr[13] = (uint8_t)(w >> 8); hdr[14] = (uint8_t)h; hdr[15] = (uint8_t)(h >> 8);
  hdr[16] = (uint8_t)(comp * 8); hdr[17] = (uint8_t)(0x20 | (comp == 4 ? 8 : 0));
  bool ok = fwrite(hdr, 1, 18, f) == 18;
  if (comp == 1) ok = ok && fwrite(data, 1, w * h, f) == w * h;
  else
  {
    std::vector<uint8_t> row(w * 4);
    const uint8_t *s = (const uint8_t *)data;
    for (size_t y = 0; y < h && ok; y++, s += w * 4)
    {
      for (size_t x = 0; x < w; x++) { row[4 * x] = s[4 * x + 2]; row[4 * x + 1] = s[4 * x + 1]; row[4 * x + 2] = s[4 * x]; row[4 * x + 3] = s[4 * x + 3]; }
      ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
  }
  fclose(f);
  return ok;
}

static uint64_t ParseUInt(const char *text)
{
  uint64_t ret = 0;
  for (; *text >= '0' && *text <= '9'; text++) ret = ret * 10 + (uint64_t)(*text - '0');
  return ret;
}

static const char Arg_NoWrite[] = "--no-output";
static const char Arg_ErrorFactor[] = "--error-factor";
static const char Arg_AccurateBitCrushing[] = "--accurate-bit-crushing";
static const char Arg_SingleThreaded[] = "--single-thread";
static const char Arg_ListCount[] = "--count";
static const char Arg_List[] = "--";
static const char Arg_Threads[] = "--threads";
static const char Arg_OutDir[] = "--out-dir";
static const char Arg_Stream[] = "--stream";
static const char Arg_FixedBlocks[] = "--fixed-blocks";

// src/main.cpp:46-54: colours the block-index plane for viewing
static int32_t Hash(const int32_t value)
{
  const uint64_t oldstate = value * 6364136223846793005ULL + (value | 1);
  const uint32_t xorshifted = (uint32_t)(((oldstate >> 18) ^ oldstate) >> 27);
  const uint32_t rot = (uint32_t)(oldstate >> 59);
  return (int32_t)((xorshifted >> rot) | (xorshifted << (uint32_t)((-(int32_t)rot) & 31)));
}

int main(const int argc, const char **pArgv)
{
  if (argc == 1)
    FAIL(EXIT_SUCCESS, "Usage:\nlimg_hip_cli [<InputFile> | --] [%s | %s <Factor> | %s | %s | %s | %s <T> | %s <dir> | %s <file>] \n  if input file is --:\n    [%s <Count>] -- <list of files>)\n",
         Arg_NoWrite, Arg_ErrorFactor, Arg_AccurateBitCrushing, Arg_SingleThreaded, Arg_FixedBlocks, Arg_Threads, Arg_OutDir, Arg_Stream, Arg_ListCount);

  // extra mode: limg_hip_cli --decode <file.lmg3> [<out.tga>]   (limg_decode of a compact stream written by --stream)
  if (!strcmp(pArgv[1], "--decode"))
  {
    if (argc < 3) FAIL(EXIT_FAILURE, "Usage: limg_hip_cli --decode <file.lmg3> [<out.tga>]\n");
    std::vector<uint8_t> stream;
    if (!read_file(pArgv[2], stream)) FAIL(EXIT_FAILURE, "Failed to read '%s'.\n", pArgv[2]);
    size_t sx = 0, sy = 0;
    bool alpha = false;
    limg_result r = limg_decode_info(stream.data(), stream.size(), &sx, &sy, &alpha);
    if (r != limg_success) FAIL(EXIT_FAILURE, "'%s' is not an LMG3 stream (0x%" PRIX32 ").\n", pArgv[2], (uint32_t)r);
    std::vector<uint32_t> image(sx * sy);
    r = limg_decode(stream.data(), stream.size(), image.data(), image.size());
    if (r != limg_success) FAIL(EXIT_FAILURE, "limg_decode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)r);
    const std::string out = argc > 3 ? pArgv[3] : "limg_out.tga";
    printf("%" PRIu64 " x %" PRIu64 " pixels, %s.\n", (uint64_t)sx, (uint64_t)sy, alpha ? "RGBA" : "RGB");
    puts(write_tga(out, sx, sy, 4, image.data()) ? "Wrote decoded file." : "Failed to write decoded file.");
    return EXIT_SUCCESS;
  }

  const char *sourceImagePath = pArgv[1];
  bool writeEncodedImages = true, fastBitCrushing = true, useThreadPool = true, fixedBlocks = false;
  uint32_t errorFactor = 100;
  size_t listCount = 1, threads = std::thread::hardware_concurrency();
  std::string outDir = ".", streamPath;
  if (threads == 0) threads = 1;

  int argIndex = 2;
  while (argc - argIndex > 0)
  {
    const int remaining = argc - argIndex;
    const char *a = pArgv[argIndex];
    if (!strcmp(a, Arg_NoWrite)) { argIndex++; writeEncodedImages = false; }
    else if (!strcmp(a, Arg_AccurateBitCrushing)) { argIndex++; fastBitCrushing = false; }
    else if (!strcmp(a, Arg_SingleThreaded)) { argIndex++; useThreadPool = false; }
    else if (!strcmp(a, Arg_FixedBlocks)) { argIndex++; fixedBlocks = true; }
    else if (remaining >= 2 && !strcmp(a, Arg_ErrorFactor)) { errorFactor = (uint32_t)ParseUInt(pArgv[argIndex + 1]); argIndex += 2; }
    else if (remaining >= 2 && !strcmp(a, Arg_Threads)) { threads = (size_t)ParseUInt(pArgv[argIndex + 1]); argIndex += 2; if (!threads) useThreadPool = false; }
    else if (remaining >= 2 && !strcmp(a, Arg_OutDir)) { outDir = pArgv[argIndex + 1]; argIndex += 2; }
    else if (remaining >= 2 && !strcmp(a, Arg_Stream)) { streamPath = pArgv[argIndex + 1]; argIndex += 2; }
    else if (remaining > 1 && !strcmp(a, Arg_List))
    {
      if (strcmp(sourceImagePath, Arg_List) != 0) FAIL(EXIT_FAILURE, "'%s' is only supported with input file '%s', found '%s'.\n", a, Arg_List, sourceImagePath);
      writeEncodedImages = false;
      sourceImagePath = nullptr;
      argIndex++;
      break;
    }
    else if (remaining > 1 && !strcmp(a, Arg_ListCount))
    {
      if (strcmp(sourceImagePath, Arg_List) != 0) FAIL(EXIT_FAILURE, "'%s' is only supported with input file '%s', found '%s'.\n", a, Arg_List, sourceImagePath);
      listCount = (size_t)ParseUInt(pArgv[argIndex + 1]);
      argIndex += 2;
    }
    else FAIL(EXIT_FAILURE, "Invalid Parameter: '%s'. Aborting.\n", a);
  }
  if (sourceImagePath && !strcmp(sourceImagePath, Arg_List)) FAIL(EXIT_FAILURE, "No files given after '%s'.\n", Arg_List);

  limg_thread_pool *pThreadPool = useThreadPool ? limg_thread_pool_new(threads) : nullptr;

  size_t pixels = 0, nanosecs = 0;
  const bool singlePerfEval = sourceImagePath == nullptr && argc == argIndex + 1 && listCount > 1;

  do
  {
    const char *filename = sourceImagePath;
    if (filename == nullptr)
    {
      filename = pArgv[argIndex++];
      if (!singlePerfEval) printf("\r'%s' (%d remaining) (~ %8.4f Mpx/s) ...", filename, argc - argIndex, (pixels * 1e-6) / (nanosecs * 1e-9f));
    }

    std::vector<uint32_t> source;
    size_t sizeX = 0, sizeY = 0;
    int channels = 0;
    if (!load_image(filename, source, sizeX, sizeY, channels)) FAIL(EXIT_FAILURE, "Failed to read source image from '%s'.\n", filename);
    const bool hasAlpha = channels == 4;
    const size_t count = sizeX * sizeY;

    if (sourceImagePath != nullptr)
    {
      std::vector<uint32_t> target(count), planes32[7], blockIndex(count);
      std::vector<uint8_t> fac[3], bitsPerPixel(count);
      for (auto &p : planes32) p.assign(count, 0);
      for (auto &p : fac) p.assign(count, 0);
      printf("%" PRIu64 " x %" PRIu64 " pixels.\n", (uint64_t)sizeX, (uint64_t)sizeY);

      limg_result result;
      const int64_t before = CurrentTimeNs();
      if (fixedBlocks)
      {
        limg_encode3d_info info;
        info.pDecoded = target.data(); info.pShiftABCX = planes32[0].data();
        info.pColAMin = planes32[1].data(); info.pColAMax = planes32[2].data(); info.pColBMin = planes32[3].data(); info.pColBMax = planes32[4].data();
        info.pColCMin = planes32[5].data(); info.pColCMax = planes32[6].data();
        info.pFactorsA = fac[0].data(); info.pFactorsB = fac[1].data(); info.pFactorsC = fac[2].data();
        result = limg_encode3d_test(source.data(), sizeX, sizeY, hasAlpha, &info, errorFactor, pThreadPool, fastBitCrushing);
      }
      else
      {
        limg_blocked_encode3d_info info;
        info.pDecoded = target.data(); info.pShiftABCX = planes32[0].data();
        info.pColAMin = planes32[1].data(); info.pColAMax = planes32[2].data(); info.pColBMin = planes32[3].data(); info.pColBMax = planes32[4].data();
        info.pColCMin = planes32[5].data(); info.pColCMax = planes32[6].data();
        info.pFactorsA = fac[0].data(); info.pFactorsB = fac[1].data(); info.pFactorsC = fac[2].data();
        info.pBlockError = nullptr; info.pBitsPerPixel = bitsPerPixel.data(); info.pBlockIndex = blockIndex.data();
        result = limg_blocked_encode3d_test(source.data(), sizeX, sizeY, hasAlpha, &info, errorFactor, pThreadPool, fastBitCrushing);
      }
      const int64_t after = CurrentTimeNs();

      printf("limg_encode_test completed with exit code 0x%" PRIX32 ".\n", (uint32_t)result);
      printf("Elapsed Time: %f ms\n", (after - before) * 1e-6);
      printf("Throughput: %f Mpx/s\n", (count * 1e-6) / ((after - before) * 1e-9));
      if (result != limg_success) FAIL(EXIT_FAILURE, "Encode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)result);
      if (!fixedBlocks)
      {
        uint64_t bits = 0;
        for (size_t i = 0; i < count; i++) bits += bitsPerPixel[i];
        printf("Compression Average: ~%7.4f bits per pixel\n", bits / (double)count); // upstream prints this from inside the library (src/limg.cpp:2433-2440)
      }

      double mean, max;
      const double psnr = limg_compare(source.data(), target.data(), sizeX, sizeY, hasAlpha, &mean, &max);
      printf("\nImage Perceptual RGB(A) PSNR: %4.2f dB (mean: %5.3f => %7.5f%% | sqrt: %5.3f%%)\n\n", psnr, mean, (mean / max) * 100.0, (sqrt(mean) / sqrt(max)) * 100.0);

      if (!streamPath.empty())
      {
        std::vector<uint8_t> stream(limg_encode_bound(sizeX, sizeY));
        size_t bytes = 0;
        limg_result r = limg_encode(source.data(), sizeX, sizeY, hasAlpha, stream.data(), stream.size(), &bytes, errorFactor, pThreadPool, fastBitCrushing);
        if (r != limg_success) FAIL(EXIT_FAILURE, "limg_encode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)r);
        std::vector<uint32_t> again(count);
        r = limg_decode(stream.data(), bytes, again.data(), again.size());
        if (r != limg_success) FAIL(EXIT_FAILURE, "limg_decode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)r);
        // the stream belongs to the fixed-8x8 path: compare with that path's decoded image
        std::vector<uint32_t> fixedDecoded;
        if (!fixedBlocks)
        {
          std::vector<uint32_t> tmp32[7];
          std::vector<uint8_t> tmp8[3];
          fixedDecoded.assign(count, 0);
          for (auto &p : tmp32) p.assign(count, 0);
          for (auto &p : tmp8) p.assign(count, 0);
          limg_encode3d_info fi;
          fi.pDecoded = fixedDecoded.data(); fi.pShiftABCX = tmp32[0].data(); fi.pColAMin = tmp32[1].data(); fi.pColAMax = tmp32[2].data(); fi.pColBMin = tmp32[3].data();
          fi.pColBMax = tmp32[4].data(); fi.pColCMin = tmp32[5].data(); fi.pColCMax = tmp32[6].data(); fi.pFactorsA = tmp8[0].data(); fi.pFactorsB = tmp8[1].data(); fi.pFactorsC = tmp8[2].data();
          r = limg_encode3d_test(source.data(), sizeX, sizeY, hasAlpha, &fi, errorFactor, pThreadPool, fastBitCrushing);
          if (r != limg_success) FAIL(EXIT_FAILURE, "limg_encode3d_test failed with exit code 0x%" PRIX32 ".\n", (uint32_t)r);
        }
        const bool same = memcmp(again.data(), fixedBlocks ? target.data() : fixedDecoded.data(), count * 4) == 0;
        printf("Stream: %" PRIu64 " bytes (%5.3f bits per pixel); decoding it %s the decoded image.\n", (uint64_t)bytes, bytes * 8.0 / count, same ? "reproduces" : "DOES NOT reproduce");
        FILE *f = fopen(streamPath.c_str(), "wb");
        if (!f || fwrite(stream.data(), 1, bytes, f) != bytes) FAIL(EXIT_FAILURE, "Failed to write '%s'.\n", streamPath.c_str());
        fclose(f);
        if (!same) return EXIT_FAILURE;
      }

      if (writeEncodedImages)
      {
        puts(write_tga(outDir + "/limg_out.tga", sizeX, sizeY, 4, target.data()) ? "Wrote decoded file." : "Failed to write decoded file.");
        write_tga(outDir + "/limg_fac_a.tga", sizeX, sizeY, 1, fac[0].data());
        write_tga(outDir + "/limg_fac_b.tga", sizeX, sizeY, 1, fac[1].data());
        write_tga(outDir + "/limg_fac_c.tga", sizeX, sizeY, 1, fac[2].data());
        static const char *names[7] = { "limg_bits", "limg_col_a_min", "limg_col_a_max", "limg_col_b_min", "limg_col_b_max", "limg_col_c_min", "limg_col_c_max" };
        for (int i = 0; i < 7; i++) write_tga(outDir + "/" + names[i] + ".tga", sizeX, sizeY, 4, planes32[i].data());
        if (!fixedBlocks)
        {
          write_tga(outDir + "/limg_bpp.tga", sizeX, sizeY, 1, bitsPerPixel.data());
          write_tga(outDir + "/limg_block_idx_raw.tga", sizeX, sizeY, 4, blockIndex.data());
          for (size_t i = 0; i < count; i++) // src/main.cpp:264-267
            if (blockIndex[i] & ((uint32_t)1 << 31)) blockIndex[i] = (uint32_t)Hash((int32_t)blockIndex[i]) | 0xFF000000;
          write_tga(outDir + "/limg_block_idx.tga", sizeX, sizeY, 4, blockIndex.data());
        }
      }
    }
    else if (singlePerfEval)
    {
      std::vector<uint64_t> timeNs(listCount);
      uint64_t timeSum = 0, min = UINT64_MAX, max = 0;
      const double megapixels = count * 1e-6;
      printf("\rDry Run...");
      limg_result result = limg_encode3d_test_perf(source.data(), sizeX, sizeY, hasAlpha, errorFactor, pThreadPool, fastBitCrushing);
      if (result != limg_success) FAIL(EXIT_FAILURE, "Encode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)result);
      for (size_t i = 0; i < listCount; i++)
      {
        const int64_t before = CurrentTimeNs();
        result = limg_encode3d_test_perf(source.data(), sizeX, sizeY, hasAlpha, errorFactor, pThreadPool, fastBitCrushing);
        const int64_t after = CurrentTimeNs();
        if (result != limg_success) FAIL(EXIT_FAILURE, "Encode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)result);
        timeNs[i] = (uint64_t)(after - before);
        timeSum += timeNs[i];
        if (timeNs[i] > max) max = timeNs[i];
        if (timeNs[i] < min) min = timeNs[i];
        printf("\rThroughput: ~%5.3f Mpx/s", megapixels / (timeNs[i] * 1e-9));
      }
      const double mean = timeSum / (double)listCount;
      double std_dev = 0;
      for (size_t i = 0; i < listCount; i++) { const double d = timeNs[i] - mean; std_dev += d * d; }
      std_dev = sqrt(std_dev / (double)(listCount - 1));
      printf("\rMean Elapsed Time: %8.4f ms (%8.4f - %8.4f ms | %8.4f - %8.4f ms std dev)\n", mean * 1e-6, min * 1e-6, max * 1e-6, (mean - std_dev) * 1e-6, (mean + std_dev) * 1e-6);
      printf("Throughput: %5.3f Mpx/s (%5.3f - %5.3f Mpx/s | %5.3f - %5.3f Mpx/s std dev)\n", megapixels / (mean * 1e-9), megapixels / (max * 1e-9), megapixels / (min * 1e-9),
             megapixels / ((mean + std_dev) * 1e-9), megapixels / ((mean - std_dev) * 1e-9));
    }
    else
    {
      const int64_t before = CurrentTimeNs();
      for (size_t i = 0; i < listCount; i++)
      {
        const limg_result result = limg_encode3d_test_perf(source.data(), sizeX, sizeY, hasAlpha, errorFactor, pThreadPool, fastBitCrushing);
        if (result != limg_success) FAIL(EXIT_FAILURE, "Encode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)result);
      }
      const int64_t after = CurrentTimeNs();
      pixels += count * listCount;
      nanosecs += (size_t)(after - before);
    }
  } while (sourceImagePath == nullptr && argIndex < argc);

  if (sourceImagePath == nullptr && !singlePerfEval)
    printf("\rComplete.   \nProcessed %5.3f Mpx in %5.3f sec / %5.3f mins \nThroughput: %8.5f MPx/s\n\n\n", pixels * 1e-6, nanosecs * 1e-9, (nanosecs * 1e-9) / 60.0, (pixels * 1e-6) / (nanosecs * 1e-9));

  limg_thread_pool_destroy(&pThreadPool);
  return EXIT_SUCCESS;
}

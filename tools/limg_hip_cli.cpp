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
// (size of the pool whose strip partition the 8x8 path reproduces; default: limg_threading_max_threads()),
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

// ---- image readers: 8-bit PNG (non-interlaced; gray, gray+alpha, RGB, RGBA, palette; tRNS colour keys honoured like stb_image does), uncompressed TGA,
//      binary PPM.  NOT read (a specific message says so): 16-bit and 1/2/4-bit PNGs, interlaced PNGs, anything above 2^30 pixels. ------------------------
static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static bool read_file(const char *path, std::vector<uint8_t> &out)
{
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  long n = -1;
  const bool sized = fseek(f, 0, SEEK_END) == 0 && (n = ftell(f)) >= 0 && fseek(f, 0, SEEK_SET) == 0;
  bool ok = false;
  if (sized)
  {
    out.resize((size_t)n);
    ok = fread(out.data(), 1, out.size(), f) == out.size();
  }
  fclose(f);
  return ok;
}

// why the last load_image() failed, for the error line (stb_image, which upstream uses, reads more PNG variants than this reader)
static const char *g_loadError = "unknown format";
static const size_t kMaxPixels = (size_t)1 << 30; // 32768^2: far above anything the encoder is sized for, far below where size arithmetic could wrap

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
  if (!width || !height) { g_loadError = "PNG without a valid IHDR"; return false; }
  if ((size_t)width * height > kMaxPixels || width > 0x7FFFFFF8u || height > 0x7FFFFFF8u) { g_loadError = "PNG larger than 2^30 pixels"; return false; }
  if (depth != 8) { g_loadError = "PNG bit depth other than 8 (16-bit and sub-byte PNGs are not supported by this reader; upstream's stb_image reads them)"; return false; }
  if (interlace != 0) { g_loadError = "interlaced PNG (not supported by this reader)"; return false; }
  int spp;
  switch (ctype) { case 0: spp = 1; break; case 2: spp = 3; break; case 3: spp = 1; break; case 4: spp = 2; break; case 6: spp = 4; break; default: g_loadError = "PNG colour type"; return false; }
  if (ctype == 3)
  { // every palette index must exist: check before decoding instead of in the per-pixel loop
    if (plte.size() < 3 || plte.size() % 3 != 0) { g_loadError = "palette PNG without a valid PLTE chunk"; return false; }
  }
  const size_t stride = (size_t)width * spp;
  std::vector<uint8_t> raw((stride + 1) * height);
  uLongf rawLen = (uLongf)raw.size();
  if ((uint64_t)raw.size() > 0xFFFFFFF0ull || (uint64_t)idat.size() > 0xFFFFFFF0ull) { g_loadError = "PNG data larger than 4 GB"; return false; }
  if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) { g_loadError = "PNG data does not inflate to the image size"; return false; }
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
  // upstream: hasAlpha = (stb_image's channel count of the file == 4), src/main.cpp:194.  stb counts RGBA, a palette with tRNS and RGB with a tRNS colour key
  // as 4; gray+alpha (2) and gray with a tRNS key (2) as not 4 -- the alpha they carry is still delivered in the pixels, as stb does when asked for 4 channels.
  const bool grayKey = ctype == 0 && trns.size() >= 2, rgbKey = ctype == 2 && trns.size() >= 6;
  channels = (ctype == 6 || (ctype == 3 && !trns.empty()) || rgbKey) ? 4 : 3;
  px.resize((size_t)width * height);
  for (size_t i = 0; i < px.size(); i++)
  {
    const uint8_t *s = &img[i * spp];
    uint32_t r, g, b, a = 255;
    switch (ctype)
    {
    case 0: r = g = b = s[0]; if (grayKey && s[0] == trns[1]) a = 0; break;
    case 4: r = g = b = s[0]; a = s[1]; break;
    case 2: r = s[0]; g = s[1]; b = s[2]; if (rgbKey && s[0] == trns[1] && s[1] == trns[3] && s[2] == trns[5]) a = 0; break;
    case 6: r = s[0]; g = s[1]; b = s[2]; a = s[3]; break;
    default:
      if ((size_t)s[0] * 3 + 2 >= plte.size()) { g_loadError = "palette index outside PLTE"; return false; }
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
  if ((bpp != 1 && bpp != 3 && bpp != 4) || width == 0 || height == 0 || f.size() < 18 + f[0] + width * height * bpp) return false;
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
  if (n != 3 || vals[2] != 255 || vals[0] == 0 || vals[1] == 0 || vals[0] > kMaxPixels || vals[1] > kMaxPixels || vals[0] * vals[1] > kMaxPixels || f.size() < pos + vals[0] * vals[1] * 3) return false;
  w = vals[0]; h = vals[1]; channels = 3;
  px.resize(w * h);
  for (size_t i = 0; i < px.size(); i++) px[i] = f[pos + 3 * i] | (f[pos + 3 * i + 1] << 8) | (f[pos + 3 * i + 2] << 16) | 0xFF000000u;
  return true;
}

static bool load_image(const char *path, std::vector<uint32_t> &px, size_t &w, size_t &h, int &channels)
{
  std::vector<uint8_t> f;
  g_loadError = "cannot read the file";
  if (!read_file(path, f)) return false;
  g_loadError = "not an 8-bit PNG, binary PPM or uncompressed TGA";
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

// ---- command line -------------------------------------------------------------------------------------------------------------------
// Option names and report lines are upstream's interface (src/main.cpp:75-86 and its printf formats) and are reproduced; the program around them is this file's own.
struct Options
{
  enum Mode { SingleFile, BenchmarkOneFile, BenchmarkList } mode = SingleFile;
  std::vector<std::string> files;
  bool writeImages = true, fastBitCrushing = true, usePool = true, fixedBlocks = false;
  uint32_t errorFactor = 100;
  size_t repeat = 1, threads = 0;
  std::string outDir = ".", streamPath;
};

static const char *const kUsage =
    "Usage:\nlimg_hip_cli [<InputFile> | --] [--no-output | --error-factor <Factor> | --accurate-bit-crushing | --single-thread | --fixed-blocks | --threads <T> | --out-dir <dir> | "
    "--stream <file>] \n  if input file is --:\n    [--count <Count>] -- <list of files>)\n";

static bool parse_number(const char *text, uint64_t &value)
{
  if (*text < '0' || *text > '9') return false;
  value = 0;
  for (; *text >= '0' && *text <= '9'; text++) value = value * 10 + (uint64_t)(*text - '0');
  return true; // trailing characters are ignored, as upstream's parser does
}

// Returns false after printing the reason.  Grammar: the first argument is the input file or "--" (list mode); flags follow in any order; in list mode a second
// "--" ends the flags and the rest are files.
static bool parse_args(int argc, const char **argv, Options &o)
{
  const bool listMode = !strcmp(argv[1], "--");
  if (!listMode) o.files.push_back(argv[1]);
  int i = 2;
  bool sawSeparator = false;
  for (; i < argc && !sawSeparator; i++)
  {
    const std::string a = argv[i];
    const bool hasValue = i + 1 < argc;
    uint64_t v = 0;
    if (a == "--no-output") o.writeImages = false;
    else if (a == "--accurate-bit-crushing") o.fastBitCrushing = false;
    else if (a == "--single-thread") o.usePool = false;
    else if (a == "--fixed-blocks") o.fixedBlocks = true;
    else if (a == "--error-factor" && hasValue && parse_number(argv[i + 1], v)) { o.errorFactor = (uint32_t)v; i++; }
    else if (a == "--threads" && hasValue && parse_number(argv[i + 1], v)) { o.threads = (size_t)v; if (v == 0) o.usePool = false; i++; }
    else if (a == "--out-dir" && hasValue) o.outDir = argv[++i];
    else if (a == "--stream" && hasValue) o.streamPath = argv[++i];
    else if ((a == "--count" || a == "--") && hasValue)
    {
      if (!listMode) { printf("'%s' is only supported with input file '--', found '%s'.\n", a.c_str(), argv[1]); return false; }
      if (a == "--") sawSeparator = true;
      else if (parse_number(argv[i + 1], v)) { o.repeat = (size_t)v; i++; }
      else { printf("Invalid Parameter: '%s'. Aborting.\n", a.c_str()); return false; }
    }
    else { printf("Invalid Parameter: '%s'. Aborting.\n", a.c_str()); return false; }
  }
  if (listMode)
  {
    for (; i < argc; i++) o.files.push_back(argv[i]);
    if (o.files.empty()) { printf("No files given after '--'.\n"); return false; }
    o.writeImages = false;
    o.mode = (o.files.size() == 1 && o.repeat > 1) ? Options::BenchmarkOneFile : Options::BenchmarkList;
  }
  if (o.threads == 0 && o.usePool) o.threads = limg_threading_max_threads() ? limg_threading_max_threads() : 1;
  return true;
}

struct Image
{
  std::vector<uint32_t> px;
  size_t w = 0, h = 0;
  bool hasAlpha = false;
  size_t count() const { return w * h; }
};

static void load_or_die(const std::string &path, Image &img)
{
  int channels = 0;
  if (!load_image(path.c_str(), img.px, img.w, img.h, channels)) FAIL(EXIT_FAILURE, "Failed to read source image from '%s' (%s).\n", path.c_str(), g_loadError);
  img.hasAlpha = channels == 4;
}

// view colours for the block-index plane: any integer mix will do (upstream hashes the index with a PCG step, src/main.cpp:46-54; the raw plane is what is compared)
static uint32_t index_colour(uint32_t v)
{
  v ^= v >> 16; v *= 0x7FEB352Du; v ^= v >> 15; v *= 0x846CA68Bu; v ^= v >> 16;
  return v | 0xFF000000u;
}

// every output plane of one encode, owned in one place
struct Planes
{
  std::vector<uint32_t> decoded, shift, col[6], blockIndex;
  std::vector<uint8_t> fac[3], bitsPerPixel;
  explicit Planes(size_t n) : decoded(n), shift(n), blockIndex(n), bitsPerPixel(n)
  {
    for (auto &c : col) c.assign(n, 0);
    for (auto &f : fac) f.assign(n, 0);
  }
  limg_encode3d_info fixed_info()
  {
    limg_encode3d_info i;
    i.pDecoded = decoded.data(); i.pShiftABCX = shift.data();
    i.pColAMin = col[0].data(); i.pColAMax = col[1].data(); i.pColBMin = col[2].data(); i.pColBMax = col[3].data(); i.pColCMin = col[4].data(); i.pColCMax = col[5].data();
    i.pFactorsA = fac[0].data(); i.pFactorsB = fac[1].data(); i.pFactorsC = fac[2].data();
    return i;
  }
  limg_blocked_encode3d_info merged_info()
  {
    limg_blocked_encode3d_info i;
    i.pDecoded = decoded.data(); i.pShiftABCX = shift.data();
    i.pColAMin = col[0].data(); i.pColAMax = col[1].data(); i.pColBMin = col[2].data(); i.pColBMax = col[3].data(); i.pColCMin = col[4].data(); i.pColCMax = col[5].data();
    i.pFactorsA = fac[0].data(); i.pFactorsB = fac[1].data(); i.pFactorsC = fac[2].data();
    i.pBlockError = nullptr; i.pBitsPerPixel = bitsPerPixel.data(); i.pBlockIndex = blockIndex.data();
    return i;
  }
};

static void write_planes(const Options &o, const Image &img, Planes &p)
{
  const std::string d = o.outDir + "/";
  puts(write_tga(d + "limg_out.tga", img.w, img.h, 4, p.decoded.data()) ? "Wrote decoded file." : "Failed to write decoded file.");
  static const char *const facNames[3] = { "limg_fac_a", "limg_fac_b", "limg_fac_c" };
  for (int k = 0; k < 3; k++) write_tga(d + facNames[k] + ".tga", img.w, img.h, 1, p.fac[k].data());
  write_tga(d + "limg_bits.tga", img.w, img.h, 4, p.shift.data());
  static const char *const colNames[6] = { "limg_col_a_min", "limg_col_a_max", "limg_col_b_min", "limg_col_b_max", "limg_col_c_min", "limg_col_c_max" };
  for (int k = 0; k < 6; k++) write_tga(d + colNames[k] + ".tga", img.w, img.h, 4, p.col[k].data());
  if (o.fixedBlocks) return;
  write_tga(d + "limg_bpp.tga", img.w, img.h, 1, p.bitsPerPixel.data());
  write_tga(d + "limg_block_idx_raw.tga", img.w, img.h, 4, p.blockIndex.data());
  for (uint32_t &v : p.blockIndex)
    if (v >> 31) v = index_colour(v); // only indices with the top bit set are recoloured upstream too (src/main.cpp:264-267)
  write_tga(d + "limg_block_idx.tga", img.w, img.h, 4, p.blockIndex.data());
}

// `--stream`: the compact stream belongs to the fixed-8x8 path, so its decode is checked against that path's image
static bool write_and_check_stream(const Options &o, const Image &img, limg_thread_pool *pool, const std::vector<uint32_t> *fixedDecoded)
{
  std::vector<uint8_t> stream(limg_encode_bound(img.w, img.h));
  size_t bytes = 0;
  limg_result r = limg_encode(img.px.data(), img.w, img.h, img.hasAlpha, stream.data(), stream.size(), &bytes, o.errorFactor, pool, o.fastBitCrushing);
  if (r != limg_success) FAIL(EXIT_FAILURE, "limg_encode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)r);
  std::vector<uint32_t> again(img.count());
  r = limg_decode(stream.data(), bytes, again.data(), again.size());
  if (r != limg_success) FAIL(EXIT_FAILURE, "limg_decode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)r);
  std::vector<uint32_t> own;
  if (!fixedDecoded)
  {
    Planes tmp(img.count());
    limg_encode3d_info fi = tmp.fixed_info();
    r = limg_encode3d_test(img.px.data(), img.w, img.h, img.hasAlpha, &fi, o.errorFactor, pool, o.fastBitCrushing);
    if (r != limg_success) FAIL(EXIT_FAILURE, "limg_encode3d_test failed with exit code 0x%" PRIX32 ".\n", (uint32_t)r);
    own.swap(tmp.decoded);
    fixedDecoded = &own;
  }
  const bool same = memcmp(again.data(), fixedDecoded->data(), img.count() * 4) == 0;
  printf("Stream: %" PRIu64 " bytes (%5.3f bits per pixel); decoding it %s the decoded image.\n", (uint64_t)bytes, bytes * 8.0 / img.count(), same ? "reproduces" : "DOES NOT reproduce");
  FILE *f = fopen(o.streamPath.c_str(), "wb");
  if (!f || fwrite(stream.data(), 1, bytes, f) != bytes) FAIL(EXIT_FAILURE, "Failed to write '%s'.\n", o.streamPath.c_str());
  fclose(f);
  return same;
}

// single-file mode (src/main.cpp:235-267, :342-370): the merged-block encoder like upstream, or the fixed 8x8 path with --fixed-blocks
static int run_single_file(const Options &o, limg_thread_pool *pool)
{
  Image img;
  load_or_die(o.files[0], img);
  Planes p(img.count());
  printf("%" PRIu64 " x %" PRIu64 " pixels.\n", (uint64_t)img.w, (uint64_t)img.h);
  limg_hip_shim::print_stats(true); // upstream's encoders print "Average Block Bits" and the shift histogram themselves (src/limg.cpp:2232-2248)
  const int64_t t0 = CurrentTimeNs();
  limg_result result;
  if (o.fixedBlocks)
  {
    limg_encode3d_info info = p.fixed_info();
    result = limg_encode3d_test(img.px.data(), img.w, img.h, img.hasAlpha, &info, o.errorFactor, pool, o.fastBitCrushing);
  }
  else
  {
    limg_blocked_encode3d_info info = p.merged_info();
    result = limg_blocked_encode3d_test(img.px.data(), img.w, img.h, img.hasAlpha, &info, o.errorFactor, pool, o.fastBitCrushing);
  }
  const double seconds = (CurrentTimeNs() - t0) * 1e-9;
  printf("limg_encode_test completed with exit code 0x%" PRIX32 ".\n", (uint32_t)result);
  printf("Elapsed Time: %f ms\n", seconds * 1e3);
  printf("Throughput: %f Mpx/s\n", img.count() * 1e-6 / seconds);
  if (result != limg_success) FAIL(EXIT_FAILURE, "Encode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)result);
  if (!o.fixedBlocks)
  { // upstream prints this from inside the library (src/limg.cpp:2433-2440)
    uint64_t bits = 0;
    for (const uint8_t b : p.bitsPerPixel) bits += b;
    printf("Compression Average: ~%7.4f bits per pixel\n", bits / (double)img.count());
  }
  double mse = 0, maxError = 0;
  const double psnr = limg_compare(img.px.data(), p.decoded.data(), img.w, img.h, img.hasAlpha, &mse, &maxError);
  printf("\nImage Perceptual RGB(A) PSNR: %4.2f dB (mean: %5.3f => %7.5f%% | sqrt: %5.3f%%)\n\n", psnr, mse, (mse / maxError) * 100.0, (sqrt(mse) / sqrt(maxError)) * 100.0);
  if (!o.streamPath.empty() && !write_and_check_stream(o, img, pool, o.fixedBlocks ? &p.decoded : nullptr)) return EXIT_FAILURE;
  if (o.writeImages) write_planes(o, img, p);
  return EXIT_SUCCESS;
}

static void perf_or_die(const Options &o, const Image &img, limg_thread_pool *pool)
{
  const limg_result r = limg_encode3d_test_perf(img.px.data(), img.w, img.h, img.hasAlpha, o.errorFactor, pool, o.fastBitCrushing);
  if (r != limg_success) FAIL(EXIT_FAILURE, "Encode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)r);
}

// `-- --count N -- file`: N timed runs of limg_encode3d_test_perf after a dry run, with upstream's statistics lines (src/main.cpp:278-323)
static int run_benchmark_one_file(const Options &o, limg_thread_pool *pool)
{
  Image img;
  load_or_die(o.files[0], img);
  const double megapixels = img.count() * 1e-6;
  printf("\rDry Run...");
  perf_or_die(o, img, pool);
  std::vector<double> ns(o.repeat);
  for (double &t : ns)
  {
    const int64_t t0 = CurrentTimeNs();
    perf_or_die(o, img, pool);
    t = (double)(CurrentTimeNs() - t0);
    printf("\rThroughput: ~%5.3f Mpx/s", megapixels / (t * 1e-9));
  }
  double sum = 0, lo = ns[0], hi = ns[0], var = 0;
  for (const double t : ns) { sum += t; lo = t < lo ? t : lo; hi = t > hi ? t : hi; }
  const double mean = sum / ns.size();
  for (const double t : ns) var += (t - mean) * (t - mean);
  const double dev = sqrt(var / (double)(ns.size() - 1));
  printf("\rMean Elapsed Time: %8.4f ms (%8.4f - %8.4f ms | %8.4f - %8.4f ms std dev)\n", mean * 1e-6, lo * 1e-6, hi * 1e-6, (mean - dev) * 1e-6, (mean + dev) * 1e-6);
  printf("Throughput: %5.3f Mpx/s (%5.3f - %5.3f Mpx/s | %5.3f - %5.3f Mpx/s std dev)\n", megapixels / (mean * 1e-9), megapixels / (hi * 1e-9), megapixels / (lo * 1e-9),
         megapixels / ((mean + dev) * 1e-9), megapixels / ((mean - dev) * 1e-9));
  return EXIT_SUCCESS;
}

// `-- [--count N] -- files...`: every file N times through limg_encode3d_test_perf, one running total (src/main.cpp:325-339, :414-417)
static int run_benchmark_list(const Options &o, limg_thread_pool *pool)
{
  double pixels = 0, seconds = 0;
  for (size_t k = 0; k < o.files.size(); k++)
  {
    printf("\r'%s' (%d remaining) (~ %8.4f Mpx/s) ...", o.files[k].c_str(), (int)(o.files.size() - k - 1), seconds > 0 ? pixels * 1e-6 / seconds : 0.0);
    Image img;
    load_or_die(o.files[k], img);
    const int64_t t0 = CurrentTimeNs();
    for (size_t i = 0; i < o.repeat; i++) perf_or_die(o, img, pool);
    seconds += (CurrentTimeNs() - t0) * 1e-9;
    pixels += (double)img.count() * (double)o.repeat;
  }
  printf("\rComplete.   \nProcessed %5.3f Mpx in %5.3f sec / %5.3f mins \nThroughput: %8.5f MPx/s\n\n\n", pixels * 1e-6, seconds, seconds / 60.0, pixels * 1e-6 / seconds);
  return EXIT_SUCCESS;
}

// extra mode: limg_hip_cli --decode <file.lmg3> [<out.tga>]   (limg_decode of a compact stream written by --stream)
static int run_decode(int argc, const char **argv)
{
  if (argc < 3) FAIL(EXIT_FAILURE, "Usage: limg_hip_cli --decode <file.lmg3> [<out.tga>]\n");
  std::vector<uint8_t> stream;
  if (!read_file(argv[2], stream)) FAIL(EXIT_FAILURE, "Failed to read '%s'.\n", argv[2]);
  size_t sx = 0, sy = 0;
  bool alpha = false;
  limg_result r = limg_decode_info(stream.data(), stream.size(), &sx, &sy, &alpha);
  if (r != limg_success) FAIL(EXIT_FAILURE, "'%s' is not an LMG3 stream (0x%" PRIX32 ").\n", argv[2], (uint32_t)r);
  std::vector<uint32_t> image(sx * sy);
  r = limg_decode(stream.data(), stream.size(), image.data(), image.size());
  if (r != limg_success) FAIL(EXIT_FAILURE, "limg_decode failed with exit code 0x%" PRIX32 ".\n", (uint32_t)r);
  printf("%" PRIu64 " x %" PRIu64 " pixels, %s.\n", (uint64_t)sx, (uint64_t)sy, alpha ? "RGBA" : "RGB");
  puts(write_tga(argc > 3 ? argv[3] : "limg_out.tga", sx, sy, 4, image.data()) ? "Wrote decoded file." : "Failed to write decoded file.");
  return EXIT_SUCCESS;
}

int main(const int argc, const char **argv)
{
  if (argc == 1) FAIL(EXIT_SUCCESS, "%s", kUsage);
  if (!strcmp(argv[1], "--decode")) return run_decode(argc, argv);
  Options o;
  if (!parse_args(argc, argv, o)) return EXIT_FAILURE;
  limg_thread_pool *pool = o.usePool ? limg_thread_pool_new(o.threads) : nullptr;
  int rc;
  switch (o.mode)
  {
  case Options::SingleFile: rc = run_single_file(o, pool); break;
  case Options::BenchmarkOneFile: rc = run_benchmark_one_file(o, pool); break;
  default: rc = run_benchmark_list(o, pool); break;
  }
  limg_thread_pool_destroy(&pool);
  return rc;
}

"""include/limg_hip.h is a C header: a C99 program (no C++, no Python) includes it, links against liblimg_hip.so and calls the ABI -- the shape of the binding a
maintainer of a C / cgo / JNI host would write.  Without a GPU the program checks the loud failure of limg_hip_init and the host-side
helpers; with one it encodes a small image through the host-pointer entry and checks the result against limg_hip_compare."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SOURCE = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "limg_hip.h"

int main(void)
{
  struct limg_hip_context *ctx = NULL;
  uint32_t chains = 0, rows = 0;
  uint64_t sizes[3] = { 100, 0, 33 }, offs[4];
  if (strncmp(limg_hip_version(), "limg_hip", 8) != 0) return 10;
  /* host-side helpers: the reference's strip rule (src/limg.cpp:2114-2134) and the layout of the variable-size stream gather */
  if (limg_hip_host_partition(8192, 2, &chains, &rows) != limg_hip_success || chains != 8 || rows != 128) return 11;
  if (limg_hip_host_gather_offsets(sizes, 3, offs) != limg_hip_success || offs[0] != 0 || offs[1] != 112 || offs[2] != 112 || offs[3] != 160) return 12;
  { /* options are versioned by their size: the header's inline passes sizeof as THIS compiler sees it (no device needed) */
    struct limg_hip_options o;
    memset(&o, 0xAB, sizeof o);
    limg_hip_default_options(&o);
    if (o.struct_size != sizeof o || o.forced_shift[0] != -1 || o.forced_shift[2] != -1 || o.dither_pcg != 0 || o.ragged_walk_threads != 0) return 18;
  }
  {
    const enum limg_hip_result r = limg_hip_init(0, &ctx);
    if (r != limg_hip_success)
    { /* no device: no CPU fallback, the reference's generic error value */
      printf("no device: limg_hip_init -> %d\n", (int)r);
      return (r == limg_hip_error_Generic && ctx == NULL) ? 0 : 13;
    }
  }
  {
    enum { W = 64, H = 40 };
    static uint32_t img[W * H], planes32[8][W * H];
    static uint8_t planes8[3][W * H];
    struct limg_hip_encode3d_info info;
    double mse = 0, max = 0, psnr;
    size_t i;
    for (i = 0; i < (size_t)W * H; i++) img[i] = 0xFF000000u | (uint32_t)((i * 2654435761u) >> 9 & 0x3F3F3Fu) | 0x404040u;
    memset(&info, 0, sizeof info);
    info.pDecoded = planes32[0]; info.pShiftABCX = planes32[1];
    info.pColAMin = planes32[2]; info.pColAMax = planes32[3]; info.pColBMin = planes32[4]; info.pColBMax = planes32[5]; info.pColCMin = planes32[6]; info.pColCMax = planes32[7];
    info.pFactorsA = planes8[0]; info.pFactorsB = planes8[1]; info.pFactorsC = planes8[2];
    if (limg_hip_encode3d(ctx, img, W, H, 1, &info, 100, 0, 1) != limg_hip_success) return 14;
    psnr = limg_hip_compare(ctx, img, info.pDecoded, W, H, 1, &mse, &max);
    printf("encoded %dx%d: PSNR %.2f dB\n", W, H, psnr);
    if (!(psnr > 25.0)) return 15;
    if (limg_hip_check_device_status(ctx) != limg_hip_success) return 16;
    { /* set one option, read it back through a struct that is SHORTER than the library's (a caller built against an earlier header) */
      struct limg_hip_options o;
      struct { uint32_t struct_size; int32_t forced_shift[3]; } old_header;
      limg_hip_default_options(&o);
      o.forced_shift[0] = o.forced_shift[1] = o.forced_shift[2] = 3;
      if (limg_hip_set_options(ctx, &o) != limg_hip_success) return 19;
      old_header.struct_size = sizeof old_header;
      if (limg_hip_get_options(ctx, (struct limg_hip_options *)&old_header) != limg_hip_success || old_header.struct_size != sizeof old_header || old_header.forced_shift[1] != 3) return 20;
      old_header.forced_shift[0] = old_header.forced_shift[1] = old_header.forced_shift[2] = -1;
      if (limg_hip_set_options(ctx, (const struct limg_hip_options *)&old_header) != limg_hip_success) return 21;
      o.struct_size = sizeof o;
      if (limg_hip_get_options(ctx, &o) != limg_hip_success || o.forced_shift[0] != -1 || o.batch_sub_images != 0) return 22;
    }
  }
  limg_hip_shutdown(&ctx);
  return ctx == NULL ? 0 : 17;
}
'''


@pytest.fixture(scope="module")
def c_program(tmp_path_factory):
    from limg_amd import build
    lib = build.build()
    d = tmp_path_factory.mktemp("c_abi")
    (d / "consumer.c").write_text(C_SOURCE)
    exe = d / "consumer"
    rocm_lib = os.environ.get("ROCM_LIB", "/opt/rocm/lib")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(d / "consumer.c"), "-o", str(exe),
           "-L", os.path.dirname(lib), "-llimg_hip", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath-link," + rocm_lib, "-Wl,-rpath," + rocm_lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "include/limg_hip.h does not work from C99:\n" + r.stderr[-3000:]
    return str(exe)


def _run(exe):
    return subprocess.run([exe], capture_output=True, text=True, timeout=300)


def test_c_consumer_without_gpu(c_program):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_c_consumer_on_gpu")
    r = _run(c_program)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    assert "no device: limg_hip_init -> 100" in r.stdout


@pytest.mark.gpu
def test_c_consumer_on_gpu(c_program):
    r = _run(c_program)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    assert "PSNR" in r.stdout

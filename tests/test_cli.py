"""limg_hip_cli (tools/limg_hip_cli.cpp): the counterpart of the reference's command-line tool (src/main.cpp), written against
include/limg_hip_shim.hpp -- so these tests also cover the C++ shim with the reference's own signatures."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PNG = os.path.join(ROOT, "tests", "golden", "original.png")


@pytest.fixture(scope="module")
def cli():
    from limg_amd import build
    return build.build_cli()


def _read_tga(path):
    raw = np.fromfile(path, dtype=np.uint8)
    w, h, bpp = int(raw[12]) | (int(raw[13]) << 8), int(raw[14]) | (int(raw[15]) << 8), int(raw[16]) // 8
    body = raw[18:18 + w * h * bpp].reshape(h, w, bpp)
    assert raw[17] & 0x20, "top-down expected"
    if bpp == 1:
        return body[:, :, 0].copy()
    rgba = body[:, :, [2, 1, 0, 3]].copy()
    return rgba.view(np.uint32).reshape(h, w)


def test_usage_and_loud_failure_without_gpu(cli):
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("Usage:")
    r = subprocess.run([cli, PNG, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid Parameter: '--bogus'. Aborting." in r.stdout
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([cli, PNG, "--no-output"], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr and "1024 x 618 pixels." in r.stdout


_TGA = {"limg_out": "pDecoded", "limg_bits": "pShiftABCX", "limg_col_a_min": "pColAMin", "limg_col_a_max": "pColAMax", "limg_col_b_min": "pColBMin",
        "limg_col_b_max": "pColBMax", "limg_col_c_min": "pColCMin", "limg_col_c_max": "pColCMax", "limg_fac_a": "pFactorsA", "limg_fac_b": "pFactorsB",
        "limg_fac_c": "pFactorsC"}


@pytest.mark.gpu
def test_single_file_matches_reference(cli, oracle, tmp_path):
    """config #1 the way upstream's tool runs it (merged-block encoder, src/main.cpp:255): report lines and every written plane against the
    real reference's hashes (tests/golden/blocked.json)."""
    r = subprocess.run([cli, PNG, "--out-dir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "1024 x 618 pixels." in r.stdout and "limg_encode_test completed with exit code 0x0." in r.stdout and "Wrote decoded file." in r.stdout
    e = json.load(open(os.path.join(gu.G, "blocked.json")))["original_rgb"]
    m = re.search(r"Image Perceptual RGB\(A\) PSNR: ([0-9.]+) dB", r.stdout)
    assert m and m.group(1) == "%.2f" % e["psnr"]
    assert re.search(r"Compression Average: ~\s*[0-9.]+ bits per pixel", r.stdout)
    stats = json.load(open(os.path.join(gu.G, "stats.json")))["original_rgb"]["merged_blocks_stdout"]
    assert stats[:stats.index("Compression Average")] in r.stdout  # upstream's own "Average Block Bits" block + shift histogram (src/limg.cpp:2232-2248), captured from the real reference
    names = dict(_TGA, limg_bpp="pBitsPerPixel", limg_block_idx_raw="pBlockIndex")
    for f, k in names.items():
        assert oracle.fnv(_read_tga(str(tmp_path / (f + ".tga")))) == e["planes"][k], (f, k)
    assert os.path.exists(str(tmp_path / "limg_block_idx.tga"))


@pytest.mark.gpu
def test_single_file_fixed_blocks_matches_reference(cli, oracle, tmp_path):
    """--fixed-blocks: limg_encode3d_test on original.png, RGB, single dither chain -- PSNR line and planes against the reference's hashes."""
    r = subprocess.run([cli, PNG, "--fixed-blocks", "--single-thread", "--out-dir", str(tmp_path), "--stream", str(tmp_path / "o.lmg3")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"Image Perceptual RGB\(A\) PSNR: ([0-9.]+) dB \(mean: ([0-9.]+)", r.stdout)
    assert m and m.group(1) == "40.70" and m.group(2) == "49.818"  # 40.6994 dB, mse 49.8179
    assert "decoding it reproduces the decoded image" in r.stdout
    assert json.load(open(os.path.join(gu.G, "stats.json")))["original_rgb"]["fixed_blocks_stdout"] in r.stdout  # "Average Block Bits: 4.003 (A: 3.650 | B: 0.260 | C: 0.093)" + histogram
    want = gu.hashes()["original_rgb"]
    for f, k in _TGA.items():
        assert oracle.fnv(_read_tga(str(tmp_path / (f + ".tga")))) == want[k], (f, k)
    import limg_amd
    st = np.fromfile(str(tmp_path / "o.lmg3"), dtype=np.uint8)
    assert limg_amd.stream_info(st) == (1024, 618, False, st.size)
    # --decode: the stream file back to an image == the decoded image written above
    r = subprocess.run([cli, "--decode", str(tmp_path / "o.lmg3"), str(tmp_path / "again.tga")], capture_output=True, text=True)
    assert r.returncode == 0 and "1024 x 618 pixels, RGB." in r.stdout, r.stdout + r.stderr
    assert np.array_equal(_read_tga(str(tmp_path / "again.tga")), _read_tga(str(tmp_path / "limg_out.tga")))
    # the stream check also works from the merged-block mode (it then encodes the 8x8 path on the side)
    r = subprocess.run([cli, PNG, "--no-output", "--single-thread", "--stream", str(tmp_path / "p.lmg3")], capture_output=True, text=True)
    assert r.returncode == 0 and "decoding it reproduces the decoded image" in r.stdout


@pytest.mark.gpu
def test_benchmark_modes(cli):
    r = subprocess.run([cli, "--", "--count", "3", "--error-factor", "50", "--", PNG], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert re.search(r"Mean Elapsed Time:\s+[0-9.]+ ms \(", r.stdout) and re.search(r"Throughput: [0-9.]+ Mpx/s \(", r.stdout)
    r = subprocess.run([cli, "--", "--count", "2", "--", PNG, PNG], capture_output=True, text=True)
    assert r.returncode == 0 and "Complete." in r.stdout and re.search(r"Processed 2\.531 Mpx in", r.stdout)


def _png(width, height, depth, ctype, raw_rows, interlace=0, extra=()):
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, interlace))
    for t, d in extra:
        out += chunk(t, d)
    return out + chunk(b"IDAT", zlib.compress(b"".join(b"\x00" + r for r in raw_rows))) + chunk(b"IEND", b"")


def test_png_variants_the_reader_refuses_say_why(cli, tmp_path):
    """The built-in PNG reader reads fewer variants than upstream's stb_image; each refusal names its reason instead of a generic load failure,
    and absurd IHDR sizes are refused before anything is allocated."""
    cases = {
        "deep.png": (_png(4, 2, 16, 2, [b"\x00" * 24] * 2), "bit depth other than 8"),
        "lace.png": (_png(4, 2, 8, 2, [b"\x00" * 12] * 2, interlace=1), "interlaced PNG"),
        "huge.png": (_png(0x7FFFFFFF, 0x7FFFFFFF, 8, 2, [b"\x00" * 12]), "larger than 2^30 pixels"),
        "nopal.png": (_png(4, 2, 8, 3, [b"\x00" * 4] * 2), "PLTE"),
        "short.png": (_png(4, 4, 8, 2, [b"\x00" * 12] * 2), "does not inflate to the image size"),
    }
    for name, (data, why) in cases.items():
        path = tmp_path / name
        path.write_bytes(data)
        r = subprocess.run([cli, str(path), "--no-output"], capture_output=True, text=True)
        assert r.returncode == 1 and "Failed to read source image" in r.stdout and why in r.stdout, (name, r.stdout)
    # a readable one gets as far as the size line (and, without a GPU, the loud no-fallback failure)
    ok = tmp_path / "key.png"
    ok.write_bytes(_png(8, 8, 8, 2, [bytes([10, 20, 30] * 8)] * 8, extra=[(b"tRNS", bytes([0, 10, 0, 20, 0, 30]))]))
    r = subprocess.run([cli, str(ok), "--no-output"], capture_output=True, text=True)
    assert "8 x 8 pixels." in r.stdout

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

"""The reference's OWN caller on a GPU through the shim.  `oracle/build_ref.sh` (build container, where /root/reference exists) compiles the reference's
src/main.cpp -- unmodified -- next to a one-line `limg.h` that forwards to include/limg_hip_shim.hpp, links it against limg_amd/liblimg_hip.so and keeps the binary
in oracle/_ref/ (git-ignored, shipped to the GPU box like the reference's libraries there).  Until round 5 this caller had only ever run where there is no GPU
(tests/test_shim_ref_main.py: it links, and fails loudly without a device); here it runs where there is one, and must print upstream's own numbers:
  * single file = `limg_blocked_encode3d_test` (src/main.cpp:255) on tests/golden/original.png: "PSNR: 40.23 dB" (the reference's figure);
  * list mode = `limg_encode3d_test_perf` per file (src/main.cpp:268-339): its throughput report, exit status 0."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "limg_ref_main_on_hip")
PNG = os.path.join(ROOT, "tests", "golden", "original.png")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/limg_ref_main_on_hip not built (oracle/build_ref.sh needs /root/reference; the binary then travels with the tree)")
    return EXE


def test_reference_main_single_file_on_gpu(exe, tmp_path):
    r = subprocess.run([exe, PNG, "--no-output"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "1024 x 618 pixels." in r.stdout
    assert "PSNR: 40.23 dB" in r.stdout, r.stdout[-1500:]
    assert "completed with exit code 0x0." in r.stdout  # (upstream prints the limg_result that way and carries on whatever it is)


def test_reference_main_single_file_accurate_on_gpu(exe, tmp_path):
    r = subprocess.run([exe, PNG, "--no-output", "--accurate-bit-crushing", "--error-factor", "25"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0 and "PSNR:" in r.stdout and "completed with exit code 0x0." in r.stdout, r.stdout[-1500:] + r.stderr[-1000:]


def test_reference_main_list_mode_on_gpu(exe, tmp_path):
    """`limg -- --count 3 -- <files>`: the reference's own benchmark loop (what BASELINE's metric is printed by upstream), with and without its thread pool."""
    for extra in ([], ["--single-thread"]):
        r = subprocess.run([exe, "--"] + extra + ["--count", "3", "--", PNG, PNG], capture_output=True, text=True, cwd=tmp_path, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "Mpx/s" in r.stdout and "exit code 0x6" not in r.stdout, r.stdout[-1500:]  # (0x64.. = limg_error_*)

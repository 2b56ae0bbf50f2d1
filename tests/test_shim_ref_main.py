"""The drop-in claim of include/limg_hip_shim.hpp, checked on the reference's own caller: a throw-away copy of /root/reference/src/main.cpp is
compiled, unmodified, next to a one-line `limg.h` that forwards to the shim, and linked against liblimg_hip.so (VERDICT r01 "missing" 1: the
shim lacked `limg_threading_max_threads`, src/limg_threading.h:17, called at src/main.cpp:165).  The copy lives in a temp dir and is deleted
with it; nothing of the reference enters the repository.  Needs /root/reference, so it runs in the build container only (`ref` marker)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PNG = os.path.join(ROOT, "tests", "golden", "original.png")

pytestmark = pytest.mark.ref


@pytest.fixture(scope="module")
def ref_main(tmp_path_factory):
    if not os.path.exists(os.path.join(REF, "src", "main.cpp")):
        pytest.skip("no /root/reference here")
    from limg_amd import build
    lib = build.build()
    d = tmp_path_factory.mktemp("ref_main")
    shutil.copy(os.path.join(REF, "src", "main.cpp"), d / "main.cpp")
    (d / "limg.h").write_text('#include "limg_hip_shim.hpp"\n')
    exe = d / "limg_ref_main_on_hip"
    rocm_lib = os.environ.get("ROCM_LIB", "/opt/rocm/lib")
    cmd = ["g++", "-std=c++17", "-O1", "-w", "-I", str(d), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(REF, "3rdParty", "stb", "include"), str(d / "main.cpp"), "-o", str(exe),
           "-L", os.path.dirname(lib), "-llimg_hip", "-lpthread", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath-link," + rocm_lib, "-Wl,-rpath," + rocm_lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "the reference's src/main.cpp does not build against the shim:\n" + r.stderr[-3000:]
    yield str(exe)
    shutil.rmtree(d, ignore_errors=True)


def test_reference_main_links_against_the_shim(ref_main):
    assert os.path.exists(ref_main)
    # every limg symbol the tool needs is resolved by the shim (inline) or by liblimg_hip.so
    und = subprocess.run(["nm", "-D", "--undefined-only", ref_main], capture_output=True, text=True).stdout
    wanted = [l.split()[-1] for l in und.splitlines() if "limg" in l]
    assert wanted and all(s.startswith("limg_hip_") for s in wanted), wanted
    exported = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "limg_amd", "liblimg_hip.so")], capture_output=True, text=True).stdout
    for s in wanted:
        assert (" T " + s) in exported, s


def test_reference_main_runs_and_fails_loudly_without_gpu(ref_main, tmp_path):
    """No GPU in the build container: the reference's tool, now on the HIP library, must report the failure of the encode through the
    reference's own limg_result path (no CPU fallback), after having loaded the image with the reference's own stb loader."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_reference_main_on_gpu")
    r = subprocess.run([ref_main, PNG, "--no-output"], capture_output=True, text=True, cwd=tmp_path)
    assert "1024 x 618 pixels." in r.stdout
    # upstream's tool prints the limg_result and carries on (src/main.cpp:257-259): limg_error_Generic = 100 = 0x64, and the library says why on stderr
    assert "completed with exit code 0x64." in r.stdout and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_reference_main_on_gpu(ref_main, tmp_path):
    """Where both the reference and a GPU exist: upstream's tool on the HIP library prints upstream's PSNR for config #1 (40.23 dB).  (No box has both:
    tests/test_gpu_ref_main.py runs the same caller, prebuilt by oracle/build_ref.sh, on the GPU box.)"""
    r = subprocess.run([ref_main, PNG, "--no-output"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PSNR: 40.23 dB" in r.stdout

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The suite runs on the -DLIMG_HIP_TEST_HOOKS build (limg_amd/liblimg_hip_test.so: the product's sources + the fault-injection / A-B hooks of
# include/limg_hip_test_hooks.h); the product library carries none of them.  Only THIS process is redirected: bench.py, the CLI, the shim binaries and the
# rank scripts a test starts load the product unless the test hands them LIMG_HIP_LIB itself.  tests/test_product_library.py runs on the product library.
import limg_amd  # noqa: E402

if not os.environ.get("LIMG_HIP_LIB"):
    limg_amd.LIB_PATH = limg_amd.TEST_LIB_PATH


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the real reference build; only where /root/reference exists)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.bind import Oracle
    path = os.path.join(ROOT, "oracle", "liblimg_oracle.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liblimg_oracle.so"])
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.bind import Ref, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    r = Ref()
    # The bit-exact pin is to the reference executing Intel's RSQRTPS (captured in limg_rsqrt_x86_table.h).  On a host whose RSQRTPS gives other bits (AMD
    # EPYC: the GPU boxes, where oracle/_ref rides along) the real reference legitimately differs from the oracle in the float stage: skip, do not fail.
    import ctypes as C
    import numpy as np
    from oracle.bind import Oracle
    o = Oracle()
    probe = np.concatenate([np.linspace(1e-6, 4.0, 4099, dtype=np.float32), np.float32(2.0) ** np.arange(-20, 20, dtype=np.float32) * np.float32(1.2345)]).astype(np.float32)
    hw = np.zeros_like(probe)
    r.lib.ref_rsqrtps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    r.lib.ref_rsqrtps(probe.ctypes.data_as(C.c_void_p), hw.ctypes.data_as(C.c_void_p), probe.size)
    table = np.array([o.lib.limg_oracle_rsqrt_x86(float(x)) for x in probe], dtype=np.float32)
    if not np.array_equal(hw.view(np.uint32), table.view(np.uint32)):
        pytest.skip("this host's RSQRTPS differs from the captured Intel table: reference-vs-oracle bit comparisons do not apply here")
    return r

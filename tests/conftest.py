import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
    return Ref()

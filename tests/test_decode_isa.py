"""k_stream_decode's hand-counted `s_waitcnt vmcnt(2)` (limg_amd/isa_check.py says what must hold).  The check itself runs inside limg_amd/build.py on every build of the
library, with the flags that ship (a test compiled with other flags proves nothing about the code object on the box); here it is run once more on both build
variants' compile lines, and on a doctored listing to show that it can fail."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("test_hooks", [False, True])
def test_inflight_payload_registers_are_not_touched(tmp_path, test_hooks):
    from limg_amd import build, isa_check
    asm = build.device_assembly("limg_hip_stream.hip", str(tmp_path), test_hooks=test_hooks)
    got = isa_check.check_decode_isa(open(asm).read())
    assert got["loads"] >= 6 and got["waits"] >= 2 and got["counted_waits"] >= 2, got


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_the_check_can_fail(tmp_path):
    """a listing whose counted wait asks for more operations than lie behind the loads must be refused"""
    import re
    from limg_amd import build, isa_check
    text = open(build.device_assembly("limg_hip_stream.hip", str(tmp_path))).read()
    bad = re.sub(r"s_waitcnt vmcnt\(2\)", "s_waitcnt vmcnt(40)", text)
    assert bad != text
    with pytest.raises(AssertionError):
        isa_check.check_decode_isa(bad)

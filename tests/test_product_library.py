"""The PRODUCT library (limg_amd/liblimg_hip.so) as shipped: the rest of the suite runs on the test-hooks build (tests/conftest.py), so everything that must hold for
the plain library is checked here -- it exports the whole ABI of include/limg_hip.h and nothing of include/limg_hip_test_hooks.h, carries no fault-injection
parameter in its kernels, versions limg_hip_options by its size, and (GPU) produces the oracle's planes.  The reference has no such knobs at all (src/limg.h:27-48)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import limg_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLAIN = os.path.join(ROOT, "limg_amd", "liblimg_hip.so")


def _built():
    if not (os.path.exists(PLAIN) and os.path.exists(limg_amd.TEST_LIB_PATH)):
        pytest.skip("libraries not built")


def test_exports():
    _built()
    plain = subprocess.run(["nm", "-D", "--defined-only", PLAIN], capture_output=True, text=True).stdout
    hooks = subprocess.run(["nm", "-D", "--defined-only", limg_amd.TEST_LIB_PATH], capture_output=True, text=True).stdout
    for sym in limg_amd.ABI_SYMBOLS:
        assert (" T " + sym + "\n") in plain and (" T " + sym + "\n") in hooks, sym
    for sym in limg_amd.TEST_ABI_SYMBOLS:
        assert sym not in plain and (" T " + sym + "\n") in hooks, sym
    assert not re.search(r"test", plain, re.I), [ln for ln in plain.splitlines() if re.search(r"test", ln, re.I)]
    header = open(os.path.join(ROOT, "include", "limg_hip.h")).read()
    body = header[header.index("typedef struct limg_hip_options"):header.index("} limg_hip_options;")]
    assert "test_" not in body and "uint32_t struct_size;" in body
    assert re.search(r"\{\s*uint32_t struct_size;", body), "struct_size must lead the struct"


def _gfx950_code_objects(lib):
    """the gfx950 ELF images inside a hipcc-linked library: uncompressed clang offload bundles (magic, u64 count, per entry u64 offset / size / triple length + triple)"""
    import struct
    blob = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, at = [], blob.find(magic)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(magic))
        q = at + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple:
                out.append(blob[at + off:at + off + size])
        at = blob.find(magic, at + 1)
    return out


def test_no_fault_injection_in_the_product_kernels(tmp_path):
    """the kernel-argument struct of the product's persistent kernel has no look-back bound / skipped strip / base error members: its kernarg segment is three uint32
    members (12 bytes, 16 with the padding in front of the pointer that follows them) shorter than the test build's, for every instantiation"""
    _built()
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("no llvm-readelf")
    sizes = {}
    for name, lib in (("plain", PLAIN), ("test", limg_amd.TEST_LIB_PATH)):
        found = {}
        for i, co in enumerate(_gfx950_code_objects(lib)):
            if b"k_encode_persistent" not in co:
                continue
            f = tmp_path / ("%s_%d.co" % (name, i))
            f.write_bytes(co)
            notes = subprocess.run([readelf, "--notes", str(f)], capture_output=True, text=True).stdout
            for kern in notes.split("- .agpr_count:")[1:]:
                nm = re.search(r"\.name:\s*(\S*k_encode_persistent\S*)", kern)
                sz = re.search(r"\.kernarg_segment_size:\s*(\d+)", kern)
                if nm and sz:
                    found[nm.group(1)] = int(sz.group(1))
        assert found, "persistent kernel not found in the code object metadata of " + lib
        sizes[name] = found
    assert sizes["plain"].keys() == sizes["test"].keys()
    assert all(sizes["test"][k] - sizes["plain"][k] in (12, 16) for k in sizes["plain"]), sizes


def test_default_options_honours_the_callers_size():
    _built()
    L = limg_amd.load_library(PLAIN)
    buf = (C.c_uint8 * 256)(*([0xAB] * 256))
    L.limg_hip_default_options_sized(buf, 20)  # a caller built when the struct ended after force_split_kernels
    raw = bytes(buf)
    head = np.frombuffer(raw[:20], dtype="<i4")
    assert head.tolist() == [20, -1, -1, -1, 0]
    assert raw[20:] == b"\xAB" * 236, "bytes beyond the caller's struct were written"
    full = limg_amd.Options()
    L.limg_hip_default_options_sized(C.byref(full), C.sizeof(full))
    assert full.struct_size == C.sizeof(full) and list(full.forced_shift) == [-1, -1, -1] and full.batch_sub_images == 0
    big = (C.c_uint8 * 512)()
    L.limg_hip_default_options_sized(big, 512)  # a caller built against a LATER header: only what this library knows is written
    assert np.frombuffer(bytes(big)[:4], dtype="<u4")[0] == C.sizeof(full)


@pytest.mark.gpu
def test_product_library_on_the_gpu(oracle):
    """the plain library, loaded next to the suite's test build: oracle parity, versioned options with a short and a long struct, and test hooks refused"""
    from oracle.bind import PLANES
    g = limg_amd.LimgHip(0, lib_path=PLAIN)
    assert not g.has_test_hooks
    try:
        img = oracle.photo_noise(512, 256, 5)
        want = oracle.encode3d(img, True)
        got = g.encode3d(img, True)
        for k in PLANES:
            assert np.array_equal(want[k], got[k]), k
        with pytest.raises(limg_amd.LimgHipError):
            g.set_options(test_base_error_strip=3)
        # a caller compiled against a header that ended after force_split_kernels: 20 bytes, forced shift (3, 3, 3)
        short = np.array([20, 3, 3, 3, 0], dtype="<i4")
        assert g.lib.limg_hip_set_options(g.ctx, short.ctypes.data_as(C.c_void_p)) == 0
        o = g.get_options()
        assert list(o.forced_shift) == [3, 3, 3] and o.struct_size == C.sizeof(o) and o.dither_pcg == 0 and o.batch_sub_images == 0
        got = g.encode3d(img, True)
        want3 = oracle.encode3d(img, True, forced_shift=(3, 3, 3))
        for k in PLANES:
            assert np.array_equal(want3[k], got[k]), k
        room = np.full(8, -7, dtype="<i4")
        room[0] = 16  # room for struct_size + forced_shift only
        assert g.lib.limg_hip_get_options(g.ctx, room.ctypes.data_as(C.c_void_p)) == 0
        assert room.tolist() == [16, 3, 3, 3, -7, -7, -7, -7]
        for bad in (0, 8, 18):  # unset, too short for forced_shift, not a multiple of 4
            short[0] = bad
            assert g.lib.limg_hip_set_options(g.ctx, short.ctypes.data_as(C.c_void_p)) == 101, bad
        g.set_options()
        got = g.encode3d(img, True)
        for k in PLANES:
            assert np.array_equal(want[k], got[k]), k
        g.check()
    finally:
        g.close()

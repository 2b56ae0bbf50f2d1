"""The "LMG3" container on the CPU: the numpy restatement (oracle/stream.py) round-trips to the oracle's / the real reference's
pDecoded plane, and the host-only header validation of the C ABI (no GPU touched)."""
import ctypes as C

import numpy as np
import pytest

from oracle import stream as S


def _cases(oracle):
    yield "pn_rgba", oracle.photo_noise(64, 40, 3), True, {}
    yield "rga_ragged", oracle.random_gradient(37, 29, 3, False), True, {}          # varying alpha: raw-escape blocks
    yield "pn_rgb", oracle.photo_noise(64, 64, 7), False, {}
    yield "rga_forced8", oracle.random_gradient(64, 32, 9, False), True, {"forced_shift": (8, 8, 8)}
    yield "pn_forced0", oracle.photo_noise(32, 32, 9), True, {"forced_shift": (0, 0, 0)}
    yield "pn_ef400", oracle.photo_noise(48, 24, 11), True, {"error_factor": 400}


def test_roundtrip_equals_oracle_decoded(oracle):
    saw_escape = False
    for name, img, alpha, kw in _cases(oracle):
        enc = oracle.encode3d(img, alpha, extras=True, **kw)
        st = S.pack(enc, img.shape[1], img.shape[0], 4 if alpha else 3, error_factor=kw.get("error_factor", 100))
        _, table, _ = S.parse(st)
        saw_escape |= bool((table["shift"] >> 24).any())
        assert np.array_equal(S.decode(st, oracle), enc["pDecoded"]), name
    assert saw_escape, "no case exercised the shift-8 raw-byte escape"


@pytest.mark.ref
def test_roundtrip_equals_reference_decoded(oracle, ref):
    for name, img, alpha, kw in _cases(oracle):
        if "forced_shift" in kw:
            continue  # the reference has no forced-shift knob
        enc = oracle.encode3d(img, alpha, extras=True, **kw)
        want = ref.encode3d(img, alpha, **kw)["pDecoded"]
        st = S.pack(enc, img.shape[1], img.shape[0], 4 if alpha else 3)
        assert np.array_equal(S.decode(st, oracle), want), name


def test_escape_is_needed(oracle):
    """Without the raw byte the alpha lane of a shift-8 factor cannot be reproduced: dropping it changes pDecoded."""
    img = oracle.random_gradient(64, 32, 9, False)
    enc = oracle.encode3d(img, True, extras=True, forced_shift=(8, 8, 8))
    st = S.pack(enc, 64, 32, 4)
    hdr, table, payload = S.parse(st)
    assert (table["shift"] >> 24).any()
    zeroed = st.copy()
    zeroed[64 + 56 * len(table):] = 0
    assert not np.array_equal(S.decode(zeroed, oracle), enc["pDecoded"])


def test_struct_sizes_and_host_info(oracle):
    import limg_amd
    assert limg_amd.STREAM_HEADER_DTYPE.itemsize == 64 and limg_amd.STREAM_BLOCK_DTYPE.itemsize == 56
    assert limg_amd.STREAM_HEADER_DTYPE == S.HEADER and limg_amd.STREAM_BLOCK_DTYPE == S.BLOCK
    lib = limg_amd.load_library()
    assert lib.limg_hip_stream_bound(64, 40) == 64 + 40 * (56 + 192)
    assert lib.limg_hip_stream_bound(0, 8) == 0
    img = oracle.photo_noise(64, 40, 3)
    st = S.pack(oracle.encode3d(img, True, extras=True), 64, 40, 4)
    assert limg_amd.stream_info(st, lib) == (64, 40, True, st.size)
    assert limg_amd.stream_info(st[:64], lib) == (64, 40, True, st.size)  # the header alone suffices
    sx = C.c_size_t()
    bad = st.copy(); bad[0] ^= 1
    assert lib.limg_hip_stream_info(bad.ctypes.data_as(C.c_void_p), bad.size, C.byref(sx), None, None, None) == 101   # InvalidParameter
    assert lib.limg_hip_stream_info(st.ctypes.data_as(C.c_void_p), 63, C.byref(sx), None, None, None) == 103           # OutOfBounds
    assert lib.limg_hip_stream_info(None, 64, C.byref(sx), None, None, None) == 102                                    # ArgumentNull
    bad = st.copy(); bad[40] ^= 1  # payloadWords no longer matches totalBytes
    assert lib.limg_hip_stream_info(bad.ctypes.data_as(C.c_void_p), bad.size, C.byref(sx), None, None, None) == 101

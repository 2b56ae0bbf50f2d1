"""GPU parity of the compact stream: the packer's bytes against the CPU restatement of the container (oracle/stream.py), and
decode(encode(image)) against the pDecoded plane of the oracle (== the real reference, tests/test_oracle_vs_ref.py)."""
import numpy as np
import pytest

from oracle import stream as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fused", "split"])
def gpu(request):
    import limg_amd
    g = limg_amd.LimgHip(0)
    g.set_options(force_split=(request.param == "split"))
    g.mode = request.param
    yield g
    g.check()
    g.close()


def _cases(oracle):
    yield "pn_rgba", oracle.photo_noise(256, 64, 3), True, {}
    yield "rga", oracle.random_gradient(256, 64, 3, False), True, {}
    yield "rga_ragged", oracle.random_gradient(203, 61, 3, False), True, {}
    yield "pn_ragged_rgb", oracle.photo_noise(131, 77, 7), False, {}
    yield "pn_rgb", oracle.photo_noise(512, 32, 7), False, {}
    yield "pn_ef400", oracle.photo_noise(264, 24, 11), True, {"error_factor": 400}
    yield "pn_ef0", oracle.photo_noise(64, 24, 11), True, {"error_factor": 0}
    yield "pn_pool", oracle.photo_noise(128, 256, 13), True, {"pool_threads": 2}
    yield "tiles", oracle.photo_noise(8 * 300, 16, 17), True, {}   # 600 blocks: more than two 256-block tiles, tile edge inside a block row
    # width in whole blocks, last block row partial: in the `fused` mode the rows above the last run through the persistent kernel in compact mode and the last row
    # through the split path (limg_hip_api.hip encode_height_ragged), both writing one set of records / shift words for the packer
    yield "pn_height_ragged", oracle.photo_noise(256, 61, 19), True, {}
    yield "rga_height_ragged_pool", oracle.random_gradient(512, 100, 19, False), True, {"pool_threads": 1}
    yield "pn_height_ragged_rgb", oracle.photo_noise(1024, 301, 23), False, {}


def test_stream_bytes_and_roundtrip(gpu, oracle):
    saw_escape = False
    for name, img, alpha, kw in _cases(oracle):
        want = oracle.encode3d(img, alpha, extras=True, **kw)
        ref_stream = S.pack(want, img.shape[1], img.shape[0], 4 if alpha else 3, error_factor=kw.get("error_factor", 100))
        got = gpu.encode_stream(img, alpha, **kw)
        assert got.size == ref_stream.size, (name, got.size, ref_stream.size)
        assert np.array_equal(got, ref_stream), (name, np.argwhere(got != ref_stream)[:8].ravel())
        saw_escape |= bool((S.parse(got)[1]["shift"] >> 24).any())
        assert np.array_equal(gpu.decode_stream(got), want["pDecoded"]), name
    assert saw_escape


@pytest.mark.parametrize("shift", [(8, 8, 8), (0, 0, 0), (7, 8, 1), (3, 0, 8)])
def test_forced_shifts(gpu, oracle, shift):
    img = oracle.random_gradient(256, 32, 21, False)
    want = oracle.encode3d(img, True, extras=True, forced_shift=shift)
    gpu.set_options(forced_shift=shift, force_split=(gpu.mode == "split"))
    try:
        got = gpu.encode_stream(img, True)
    finally:
        gpu.set_options(force_split=(gpu.mode == "split"))
    assert np.array_equal(got, S.pack(want, 256, 32, 4)), shift
    assert np.array_equal(gpu.decode_stream(got), want["pDecoded"]), shift


def test_device_roundtrip_full_size(gpu):
    """8192^2 (BASELINE configs[2]) and 4096^2 gradient: decode(encode_stream) == pDecoded of the plane path, on the device."""
    import torch
    for kind, n in (("photo_noise", 8192), ("random_gradient", 4096)):
        img = gpu.synth_device(kind, n, n, seed=1)
        planes = gpu.alloc_planes_device(n, n)
        gpu.encode3d_device(img, True, planes)
        st, nbytes = gpu.encode_stream_device(img, True)
        assert 64 + (n // 8) ** 2 * 56 <= nbytes <= gpu.stream_bound(n, n)
        dec = gpu.decode_stream_device(st, nbytes, n, n)
        torch.cuda.synchronize()
        gpu.check()
        assert torch.equal(dec, planes["pDecoded"]), kind
        hdr = st[:64].cpu().numpy().view(S.HEADER)[0]
        assert int(hdr["totalBytes"]) == nbytes and int(hdr["sizeX"]) == n
        del planes, st, dec


def test_more_strips_than_one_scan_round(gpu):
    """8192 x 8448 = 33 792 work strips: the strip scan of the packer (k_stream_scan_strips: 32 K strips per round, carry between rounds) takes its second round;
    2056 x 4104 = 9 x 513 strips: the scalar (not a multiple of 4) path of the same kernel with a partial last strip per row.  decode(encode) == pDecoded on the device."""
    import torch
    for W, H in ((8192, 8448), (2056, 4104)):
        img = gpu.synth_device("photo_noise", W, H, seed=3)
        planes = gpu.alloc_planes_device(W, H)
        gpu.encode3d_device(img, True, planes)
        st, nbytes = gpu.encode_stream_device(img, True)
        dec = gpu.decode_stream_device(st, nbytes, W, H)
        torch.cuda.synchronize()
        gpu.check()
        assert torch.equal(dec, planes["pDecoded"]), (W, H)
        hdr = st[:64].cpu().numpy().view(S.HEADER)[0]
        assert int(hdr["totalBytes"]) == nbytes and int(hdr["sizeX"]) == W and int(hdr["sizeY"]) == H
        del planes, st, dec, img
        torch.cuda.empty_cache()


def test_refuses_bad_streams(gpu, oracle):
    import limg_amd
    img = oracle.photo_noise(64, 64, 3)
    st = gpu.encode_stream(img, True)
    bad = st.copy(); bad[0] ^= 0xFF
    with pytest.raises(limg_amd.LimgHipError):
        gpu.decode_stream(bad)
    with pytest.raises(limg_amd.LimgHipError):
        gpu.decode_stream(st[:100])
    # a payload offset pointing past the end: the kernel refuses the group instead of reading out of bounds
    hdr, table, _ = S.parse(st)
    evil = st.copy()
    evil[64:64 + 56 * len(table)].view(S.BLOCK)["payloadWord"][3] = 0x7FFFFFF0
    with pytest.raises(limg_amd.LimgHipError):
        gpu.decode_stream(evil)
    # offsets whose 32-bit sums wrap: a whole group of 8 at 0xFFFFFFF0 -- every 32-bit check would pass (0xFFFFFFF0 + 24 == 8)
    evil = st.copy()
    evil[64:64 + 56 * len(table)].view(S.BLOCK)["payloadWord"][:8] = 0xFFFFFFF0
    with pytest.raises(limg_amd.LimgHipError):
        gpu.decode_stream(evil)
    assert np.array_equal(gpu.decode_stream(st), gpu.encode3d(img, True)["pDecoded"])  # the context is usable afterwards


def test_device_decode_refuses_wrapping_header(gpu, oracle):
    """The device entry never sees limg_hip_stream_info: a header whose payloadWords * 8 wraps in 64 bits (>= 2^61) must be refused by the
    kernel's own header check, not accepted because the wrapped byte count is small."""
    import torch
    import limg_amd
    img = oracle.photo_noise(64, 64, 5)
    st = gpu.encode_stream(img, True)
    evil = st.copy()
    evil[:64].view(S.HEADER)["payloadWords"][0] = (1 << 61) + 3
    d = torch.from_numpy(evil).cuda()
    out = torch.zeros((64, 64), dtype=torch.int32, device="cuda")
    gpu.decode_stream_device(d, evil.size, 64, 64, out=out)
    torch.cuda.synchronize()
    with pytest.raises(limg_amd.LimgHipError):
        gpu.check()
    assert int(out.abs().max()) == 0
    assert np.array_equal(gpu.decode_stream(st), gpu.encode3d(img, True)["pDecoded"])

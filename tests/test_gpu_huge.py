"""Maximum sizes (the reference has no size limit but memory, src/limg.cpp:2175-2265): an image of more than 5.59 M blocks, whose dither chain is longer than the embedded dense
checkpoints reach (16 Mi calls) -- the context then makes the dense values it lacks from the embedded FAR checkpoints (limg_hip_api.hip ensure_checkpoints) -- and whose planes'
byte offsets pass 2^32.  24576^2 = 604 Mpixels, 9.4 M blocks, up to 28 M dither calls, ~22 GiB of device memory at a time (tools/huge_image_check.py is the same at 32768^2 = 1 Gpixel).
The whole image IS pinned against the real reference: tests/golden/fullsize.json `pn24576_strips` holds, per strip of 3072 rows, the position-sensitive 64-bit checksums of the
eleven planes the reference (oracle/_ref, one chain, one thread, 5 minutes in the build container: tools/make_golden_fullsize.py) wrote -- computed here on the device.  Around it,
size-independent properties; every comparison of whole planes happens on the device."""
import json
import os
import sys

import numpy as np
import pytest

from oracle.bind import PLANES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

N = 24576


@pytest.fixture(scope="module")
def gpu():
    import limg_amd
    g = limg_amd.LimgHip(0)  # a context of its own: its scratch (noise table 1.8 GB, records, park) goes with it
    yield g
    g.check()
    g.close()

UNIFORM = ("pShiftABCX", "pColAMin", "pColAMax", "pColBMin", "pColBMax", "pColCMin", "pColCMax")


def _band_equals_oracle(oracle, img, planes, y0, keys):
    want = oracle.encode3d(img[y0:y0 + 64].cpu().numpy().view(np.uint32), True)
    for k in keys:
        got = planes[k][y0:y0 + 64].cpu().numpy()
        got = got.view(np.uint32) if got.dtype == np.int32 else got
        assert np.array_equal(got, want[k]), (y0, k)


def test_image_beyond_the_dense_checkpoints(gpu, oracle):
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 64 * 2 ** 30:
        pytest.skip("needs 64 GiB of free device memory")
    img = gpu.synth_device("photo_noise", N, N, seed=1)
    planes = gpu.alloc_planes_device(N, N)
    try:
        gpu.encode3d_device(img, True, planes)
        torch.cuda.synchronize()
        gpu.check()
        psnr, _ = gpu.compare_device(img, planes["pDecoded"], True)
        assert abs(psnr - 38.87) < 0.1, psnr
        # every plane of every strip against the REAL reference's checksums: the chain is right 28 M calls deep (far checkpoints), the stores are right past 2^32
        sys.path.insert(0, ROOT)
        import bench
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize.json")))["pn24576_strips"]
        assert (gold["w"], gold["h"], gold["seed"], gold["kw"]) == (N, N, 1, {})
        assert abs(psnr - gold["psnr"]) < 1e-6, (psnr, gold["psnr"])
        sr = gold["strip_rows"]
        for si, strip in enumerate(gold["strips"]):
            for name, want in strip["sum64"].items():
                assert bench.sum64_device(planes[name][si * sr:(si + 1) * sr]) == want, (si, name)
        _band_equals_oracle(oracle, img, planes, 0, PLANES)            # the chain starts at the seed: every plane
        _band_equals_oracle(oracle, img, planes, N // 2, UNIFORM)       # chain-independent planes in the middle ...
        _band_equals_oracle(oracle, img, planes, N - 64, UNIFORM)       # ... and at the far end (byte offset 2.4e9 in the 32-bit planes)
        # the compact stream's round trip at this size
        st, nbytes = gpu.encode_stream_device(img, True)
        dec = gpu.decode_stream_device(st, nbytes, N, N)
        torch.cuda.synchronize()
        assert torch.equal(dec, planes["pDecoded"])
        del st, dec
        # strip-restart partition (pool of 2 = 8 strips): the last strip == its standalone encode and, at its first rows, == the oracle on every plane
        gpu.encode3d_device(img, True, planes, pool_threads=2)
        rows = (N // 8 // 8) * 8
        part = gpu.alloc_planes_device(N, rows)
        gpu.encode3d_device(img[7 * rows:], True, part)
        torch.cuda.synchronize()
        for k in PLANES:
            assert torch.equal(part[k], planes[k][7 * rows:]), k
        _band_equals_oracle(oracle, img, planes, 7 * rows, PLANES)
        del part, planes
        torch.cuda.empty_cache()
        # A partial last block row 28 M calls into the chain: the fast path takes the chain value there from a far checkpoint (+ at most 65535 calls on foot); the
        # whole-image ragged path walks every call on the host.  Every plane equal -- which also checks the GPU-filled noise table against a host walk of the whole chain.
        H = N - 3
        pa, pb = gpu.alloc_planes_device(N, H), gpu.alloc_planes_device(N, H)
        gpu.encode3d_device(img[:H], True, pa)
        gpu.set_options(test_whole_image_ragged=True)
        gpu.encode3d_device(img[:H], True, pb)
        torch.cuda.synchronize()
        gpu.check()
        for k in PLANES:
            assert torch.equal(pa[k], pb[k]), k
    finally:
        gpu.set_options()
        torch.cuda.empty_cache()

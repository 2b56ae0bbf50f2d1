"""The context's dither noise table filled on the GPU from the embedded chain checkpoints (limg_hip_noise_gpu.hip) against the host's serial walk of the same chain
(limg_hip_host_noise_table: AES-NI or the software round, pinned to the reference by tests/golden/chain.json)."""
import ctypes as C

import numpy as np
import pytest

from oracle.bind import PLANES

pytestmark = pytest.mark.gpu


def _device_table(g, calls):
    import torch
    out = torch.zeros(calls * 64, dtype=torch.uint8, device="cuda")
    r = g.lib.limg_hip_noise_table_device(g.ctx, C.c_void_p(out.data_ptr()), calls, None)
    assert r == 0, r
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _host_table(g, calls):
    out = np.zeros(calls * 64, dtype=np.uint8)
    assert g.lib.limg_hip_host_noise_table(out.ctypes.data_as(C.c_void_p), calls) == 0
    return out


def test_gpu_noise_table_equals_host_walk():
    import limg_amd
    g = limg_amd.LimgHip(0)
    try:
        for calls in (1, 1023, 1024, 1025, 40 * 1024 + 77):  # whole, partial and single stretches of 1024 calls
            assert np.array_equal(_device_table(g, calls), _host_table(g, calls)), calls
        # far into the chain (what a 8192^2 image can reach: 3.1 M calls): the last 64 K calls
        calls = 3 * 1024 * 1024 + 4321
        dev = _device_table(g, calls)
        host = _host_table(g, calls)
        assert np.array_equal(dev[-65536 * 64:], host[-65536 * 64:])
        assert np.array_equal(dev[::4099], host[::4099])
        del dev, host
        # beyond the embedded dense checkpoints (16 Mi calls: images of more than 5.59 M blocks): the context makes the missing dense values from the far table
        calls = 16 * 1024 * 1024 + 3 * 65536 + 12345
        dev = _device_table(g, calls)
        host = _host_table(g, calls)
        assert np.array_equal(dev[-(4 * 65536) * 64:], host[-(4 * 65536) * 64:])
        assert np.array_equal(dev[::4099], host[::4099])
        del dev, host
        # refused beyond the reach of the far checkpoints
        import torch
        buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
        assert g.lib.limg_hip_noise_table_device(g.ctx, C.c_void_p(buf.data_ptr()), 2 ** 27 + 1, None) == 101
    finally:
        g.check()
        g.close()


def test_encode_with_host_built_table_is_identical(oracle):
    """A/B of the two ways the context gets its table (limg_hip_options.host_noise_table): every plane equal, and equal to the oracle."""
    import limg_amd
    img = oracle.photo_noise(512, 136, 71)
    want = oracle.encode3d(img, True)
    for host in (False, True):
        g = limg_amd.LimgHip(0)
        try:
            g.set_options(host_noise_table=host)
            got = g.encode3d(img, True)
            for k in PLANES:
                assert np.array_equal(got[k], want[k]), (host, k)
        finally:
            g.check()
            g.close()

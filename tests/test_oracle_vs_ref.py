"""Pins the CPU oracle (oracle/limg_oracle.c, FLOAT_X86 mode) against the REAL reference (oracle/_ref, strict-IEEE
build of /root/reference/src) -- bit-exact on every plane and on every per-block probe.  Skipped where oracle/_ref
is absent; the committed fixtures in tests/golden (test_oracle_golden.py) carry the same pin to such machines."""
import numpy as np
import pytest

from oracle.bind import PLANES, DITHER_AES, DITHER_PCG

pytestmark = pytest.mark.ref


def _inputs(oracle):
    rg = oracle.random_gradient(136, 72, seed=3, opaque=True)
    rga = oracle.random_gradient(128, 64, seed=5, opaque=False)
    pn = oracle.photo_noise(128, 64, seed=7)
    return {"rg": rg, "rga": rga, "pn": pn}


@pytest.mark.parametrize("name,w,h", [("rg", 136, 72), ("rga", 128, 64), ("pn", 128, 64), ("pn", 61, 27), ("rg", 8, 8), ("pn", 5, 3), ("rga", 67, 64)])
@pytest.mark.parametrize("has_alpha", [True, False])
@pytest.mark.parametrize("ef", [0, 25, 100, 400])
def test_planes_bit_exact(oracle, ref, name, w, h, has_alpha, ef):
    img = np.ascontiguousarray(_inputs(oracle)[name][:h, :w])
    r = ref.encode3d(img, has_alpha, error_factor=ef)
    o = oracle.encode3d(img, has_alpha, error_factor=ef)
    for k in PLANES:
        assert np.array_equal(r[k], o[k]), (k, int((r[k] != o[k]).sum()))


@pytest.mark.parametrize("w,h", [(9, 9), (17, 10), (10, 9), (2, 65), (2, 9), (9, 2), (3, 9), (25, 33), (1, 17), (17, 1), (11, 9)])
@pytest.mark.parametrize("has_alpha", [True, False])
def test_corner_blocks_of_fewer_than_4_pixels(oracle, ref, w, h, has_alpha):
    """Upstream's channel-sum loop consumes at least 4 pixels (src/limg.cpp:478-487): a smaller block also sums what the previous block left in the
    gather buffer.  The oracle keeps the same persistent buffer, so even these shapes are bit-exact (the first block of a strip excepted: it would
    read uninitialised stack upstream)."""
    img = oracle.photo_noise(w, h, seed=13)
    for pool in (0, 1):
        r = ref.encode3d(img, has_alpha, pool_threads=pool)
        o = oracle.encode3d(img, has_alpha, pool_threads=pool)
        for k in PLANES:
            assert np.array_equal(r[k], o[k]), (k, pool)


@pytest.mark.parametrize("pool", [1, 2, 3, 8])
def test_strip_partition(oracle, ref, pool):
    img = oracle.photo_noise(64, 200, seed=11)
    r = ref.encode3d(img, True, pool_threads=pool)
    o = oracle.encode3d(img, True, pool_threads=pool, worker_threads=3)
    for k in PLANES:
        assert np.array_equal(r[k], o[k]), k


def test_pcg_dither(oracle, ref):
    img = oracle.photo_noise(72, 40, seed=2)
    r = ref.encode3d(img, True, dither_mode=DITHER_PCG)
    o = oracle.encode3d(img, True, dither_mode=DITHER_PCG)
    for k in PLANES:
        assert np.array_equal(r[k], o[k]), k


@pytest.mark.parametrize("has_alpha", [True, False])
def test_accurate_mode(oracle, ref, has_alpha):
    img = oracle.photo_noise(64, 48, seed=4)
    r = ref.encode3d(img, has_alpha, fast=False)
    o = oracle.encode3d(img, has_alpha, fast=False)
    for k in PLANES:
        assert np.array_equal(r[k], o[k]), k


def test_block_probes(oracle, ref):
    rng = np.random.default_rng(1)
    img = oracle.photo_noise(64, 64, seed=9)
    for ch in (4, 3):
        for by in range(0, 64, 16):
            for bx in range(0, 64, 16):
                px = np.ascontiguousarray(img[by:by + 8, bx:bx + 8]).ravel()
                rr, ro = ref.block_fit(px, ch), oracle.block_fit(px, ch)
                for f in rr.dtype.names:
                    if f == "avg":
                        assert np.array_equal(rr[f][0][:ch], ro[f][0][:ch])
                    else:
                        assert np.array_equal(rr[f], ro[f]), f
                a, b, c = oracle.block_factors(px, ch, ro)
                ra, rb, rc = ref.block_factors(px, ch, ro)
                assert np.array_equal(a, ra) and np.array_equal(b, rb) and np.array_equal(c, rc)
                for _ in range(40):
                    sh = rng.integers(0, 9, 3)
                    ef = int(rng.choice([25, 100, 400, 3000]))
                    assert ref.block_trial(px, ch, ro, a, b, c, sh, ef) == pytest.approx(oracle.block_trial(px, ch, ro, a, b, c, sh, ef)) or \
                        (not ref.block_trial(px, ch, ro, a, b, c, sh, ef)[0] and not oracle.block_trial(px, ch, ro, a, b, c, sh, ef)[0])
                for fast in (True, False):
                    assert np.array_equal(ref.block_search(px, ch, ro, a.copy(), b.copy(), c.copy(), 100, fast), oracle.block_search(px, ch, ro, a, b, c, 100, fast)[0])
                sh = (3, 8, 0)
                assert np.array_equal(ref.block_decode(8, 8, ch, ro, a, b, c, sh), oracle.block_decode(8, 8, ch, ro, a, b, c, sh))


@pytest.mark.parametrize("mode", [DITHER_AES, DITHER_PCG])
@pytest.mark.parametrize("n", [64, 16, 20, 7, 1, 40])
def test_dither(oracle, ref, mode, n):
    rng = np.random.default_rng(n)
    f = rng.integers(0, 256, n, dtype=np.uint8)
    h = 0xCA7F00D15BADF00D
    for s in range(1, 8):
        rh, rf = ref.dither(s, h, f, mode)
        oh, of = oracle.dither(s, h, f, mode)
        assert rh == oh and np.array_equal(rf, of)
        assert oracle.chain_step(n, h, mode) == oh
        h = oh


def test_compare(oracle, ref):
    a = oracle.photo_noise(64, 32, seed=1)
    b = oracle.photo_noise(64, 32, seed=2)
    for alpha in (True, False):
        assert ref.compare(a, b, alpha) == oracle.compare(a, b, alpha)


@pytest.mark.parametrize("shape", [(256, 128, True), (250, 131, True), (200, 77, False), (13, 9, True)])
def test_forced_shift_composition(oracle, ref, shape):
    """BASELINE config 3's forced-shift sweep has no switch upstream; oracle/ref_harness.cpp's ref_encode3d_forced_shift drives the reference's own block functions
    in limg_encode3d_test's order with the search left out (what tests/golden/fullsize.json's pn8192_forced* entries were made with).  The oracle's forced_shift
    option must give the same shift-dependent planes, ragged shapes and 3-channel input included."""
    w, h, alpha = shape
    img = oracle.photo_noise(w, h, 3)
    for s in ((0, 0, 0), (1, 1, 1), (3, 3, 3), (6, 6, 6), (7, 7, 7), (8, 8, 8), (2, 5, 8), (8, 0, 4)):
        a = oracle.encode3d(img, alpha, forced_shift=s)
        b = ref.encode3d_forced_shift(img, alpha, s)
        for k in b:
            assert np.array_equal(a[k], b[k]), (shape, s, k)

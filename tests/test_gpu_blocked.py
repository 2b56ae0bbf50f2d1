"""GPU parity of the merged-block encoder (limg_hip_blocked_encode3d == the reference's limg_blocked_encode3d_test): every plane upstream
writes, against the CPU oracle (pinned to the real reference by tests/test_oracle_blocked.py) and against the committed reference hashes."""
import json
import os

import numpy as np
import pytest

import golden_util as gu
from oracle.bind import BLOCKED_WRITTEN

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(gu.G, "blocked.json")))


@pytest.fixture(scope="module")
def gpu():
    import limg_amd
    g = limg_amd.LimgHip(0)
    yield g
    g.check()
    g.close()


def _input(orc, e):
    if e["gen"] == "png":
        return gu.load_png()
    if e["gen"] == "pn":
        return orc.photo_noise(e["w"], e["h"], e["seed"])
    return orc.random_gradient(e["w"], e["h"], e["seed"], e["gen"] == "rg")


def test_stagewise_small(gpu, oracle):
    """Localises a mismatch: rectangles (merge over GPU similarity bits), then the planes."""
    for kind, alpha, shape in (("pn", True, (256, 128)), ("rg", True, (256, 128)), ("rga", True, (203, 61)), ("pn", False, (131, 77)), ("rg", False, (64, 64)),
                               ("pn", True, (8, 8)), ("pn", True, (9, 9)), ("rg", False, (17, 10)), ("pn", True, (1, 1)), ("pn", True, (2, 65)), ("pn", False, (25, 33)), ("rg", True, (265, 9)), ("flat", True, (96, 80)), ("flat", True, (200, 168))):  # the last one: rectangles wider than the similarity window
        w, h = shape
        if kind == "flat":
            img = np.full((h, w), 0xFF336699, dtype=np.uint32)
        elif kind == "pn":
            img = oracle.photo_noise(w, h, 5)
        else:
            img = oracle.random_gradient(w, h, 5, kind == "rg")
        want = oracle.blocked_encode3d(img, alpha)
        got = gpu.blocked_encode3d(img, alpha)
        wr, gr = want["regions"], got["regions"]
        assert len(wr) == len(gr), (kind, alpha, len(wr), len(gr))
        for f in ("ox", "oy", "rx", "ry"):
            assert np.array_equal(wr[f], gr[f]), (kind, alpha, f, np.argwhere(wr[f] != gr[f])[:4].ravel())
        bad = [(k, int((got[k] != want[k]).sum())) for k in BLOCKED_WRITTEN if not np.array_equal(got[k], want[k])]
        assert not bad, (kind, alpha, bad)
        assert not got["pBlockError"].any()


@pytest.mark.parametrize("name", sorted(GOLD))
def test_reference_hashes(gpu, oracle, name):
    e = GOLD[name]
    img = _input(oracle, e)
    kw = dict(e["kw"])
    pcg = kw.pop("dither_mode", 0) != 0
    gpu.set_options(dither_pcg=pcg)
    try:
        got = gpu.blocked_encode3d(img, e["alpha"], **kw)
    finally:
        gpu.set_options()
    for k in BLOCKED_WRITTEN:
        assert oracle.fnv(got[k]) == e["planes"][k], (name, k)
    assert len(got["regions"]) == e["regions"]
    psnr, _ = gpu.compare(img, got["pDecoded"], e["alpha"])
    assert psnr == pytest.approx(e["psnr"], abs=1e-9)


def test_window_fallback_and_forced_shifts(gpu, oracle):
    """A wide rectangle (part of it beyond the precomputed similarity window: the host evaluates those pairs itself); forced shifts bypass the search."""
    img = np.zeros((64, 512), dtype=np.uint32)
    img[:] = 0xFF000000 | (np.arange(512, dtype=np.uint32)[None, :] // 4) * 0x010101  # a slow horizontal ramp: one very wide rectangle
    want = oracle.blocked_encode3d(img, True)
    assert int(want["regions"]["rx"].max()) > 8
    got = gpu.blocked_encode3d(img, True)
    bad = [k for k in BLOCKED_WRITTEN if not np.array_equal(got[k], want[k])]
    assert not bad, bad
    img = oracle.random_gradient(128, 64, 3, False)
    for shift in ((8, 8, 8), (0, 0, 0), (3, 5, 8)):
        want = oracle.blocked_encode3d(img, True, forced_shift=shift)
        gpu.set_options(forced_shift=shift)
        try:
            got = gpu.blocked_encode3d(img, True)
        finally:
            gpu.set_options()
        bad = [k for k in BLOCKED_WRITTEN if not np.array_equal(got[k], want[k])]
        assert not bad, (shift, bad)


def test_device_entry_full_size(gpu, oracle):
    """2048^2 of each generator through the device entry point: plane hashes against the oracle; stage timing is reported."""
    import torch
    for kind in ("photo_noise", "random_gradient"):
        n = 2048
        img = gpu.synth_device(kind, n, n, seed=1)
        planes = gpu.alloc_blocked_planes_device(n, n)
        gpu.blocked_encode3d_device(img, True, planes)
        torch.cuda.synchronize()
        himg = img.cpu().numpy().view(np.uint32)
        want = oracle.blocked_encode3d(himg, True)
        for k in BLOCKED_WRITTEN:
            got = planes[k].cpu().numpy()
            got = got.view(np.uint32) if got.dtype == np.int32 else got
            assert np.array_equal(got, want[k]), (kind, k)
        t = gpu.blocked_timing()
        assert t["total"] > 0 and len(gpu.blocked_regions()) == len(want["regions"])
        k = gpu.blocked_kernel_timing()  # HIP-event times of this very encode's launches: all four ran, and none can have taken longer than the call
        assert all(0 < v < t["total"] for v in k.values()), (k, t)


def test_similarity_bits_equal_host_evaluation_with_and_without_the_bound(gpu, oracle):
    """k_blocked_match's bits, with its certain-match / certain-failure bounds (the default) and with the 27-colour loop for every open pair (test_blocked_no_bound), against the host's
    evaluation of the full predicate over the same records (limg_hip_host_blocked_match_bits, pinned to the oracle / reference by tests/test_host.py) -- on content
    where the bound decides everything (noise), next to nothing (gradients), and on flat / tiny / odd-sized images."""
    import limg_amd
    lib = limg_amd.load_library()
    words = lib.limg_hip_host_blocked_match_words()
    rng = np.random.default_rng(11)
    cases = [("pn", True, 512, 384), ("rg", True, 512, 384), ("rga", True, 300, 203), ("pn", False, 264, 131), ("rg", False, 200, 160), ("rand", True, 256, 256), ("flat", True, 128, 96),
             ("mix", True, 384, 256), ("pn", True, 8, 8), ("rg", True, 24, 136), ("rg2", True, 640, 512), ("rga2", False, 512, 320), ("smooth", True, 384, 384)]
    decided = {}
    for kind, alpha, w, h in cases:
        if kind == "pn":
            img = oracle.photo_noise(w, h, 9)
        elif kind in ("rg", "rga"):
            img = oracle.random_gradient(w, h, 9, kind == "rg")
        elif kind in ("rg2", "rga2"):
            img = oracle.random_gradient(w, h, 31, kind == "rg2")
        elif kind == "smooth":  # slow ramps + a little noise: small normals next to zero normals (where the failure bound has the most to decide)
            yy, xx = np.mgrid[0:h, 0:w]
            n = rng.integers(0, 3, (h, w, 3))
            r8 = ((xx // 3 + n[..., 0]) & 255).astype(np.uint32); g8 = ((yy // 2 + n[..., 1]) & 255).astype(np.uint32); b8 = (((xx + yy) // 5 + n[..., 2]) & 255).astype(np.uint32)
            img = (np.uint32(0xFF000000) | (b8 << 16) | (g8 << 8) | r8).astype(np.uint32)
        elif kind == "rand":
            img = rng.integers(0, 1 << 32, (h, w), dtype=np.uint64).astype(np.uint32)
        elif kind == "flat":
            img = np.full((h, w), 0xFF4080C0, dtype=np.uint32)
        else:  # noise with low-contrast patches and gradients in between: states of very different sizes side by side
            img = oracle.photo_noise(w, h, 3)
            img[64:192, 32:200] = oracle.random_gradient(168, 128, 4, True)
            img[200:240, 250:380] = (img[200:240, 250:380] & np.uint32(0xFF030303)) | np.uint32(0x00808080)
        ch = 4 if alpha else 3
        bits = {}
        for no_bound in (False, True):
            gpu.set_options(test_blocked_no_bound=no_bound)
            try:
                gpu.blocked_encode3d(img, alpha)
                bits[no_bound] = gpu.blocked_match_bits()
            finally:
                gpu.set_options()
        fits = oracle.blocked_encode3d(img, alpha, planes=False)["pass1"]
        by, bx = fits.shape
        want = np.zeros(by * bx * words, dtype=np.uint64)
        limg_amd._check(lib.limg_hip_host_blocked_match_bits(limg_amd._np_ptr(np.ascontiguousarray(fits)), bx, by, ch, limg_amd._np_ptr(want)), "limg_hip_host_blocked_match_bits")
        assert np.array_equal(bits[True], want), (kind, alpha, "27-colour loop")
        assert np.array_equal(bits[False], want), (kind, alpha, "with the bound")


def test_giant_rectangle(gpu, oracle):
    """A flat 512x512 image with a few odd blocks: one rectangle of ~250k pixels handled by a single wave (chunk loops, 64-bit block errors, long
    pixel-order sums, host evaluation of far-apart pairs)."""
    img = np.full((512, 512), 0xFF808080, dtype=np.uint32)
    img[200:208, 304:312] = oracle.photo_noise(8, 8, 3)
    img[:, :] += (np.arange(512, dtype=np.uint32)[None, :] // 128) * 0x000100
    want = oracle.blocked_encode3d(img, True)
    assert int((want["regions"]["rx"] * want["regions"]["ry"]).max()) > 1000
    got = gpu.blocked_encode3d(img, True)
    bad = [(k, int((got[k] != want[k]).sum())) for k in BLOCKED_WRITTEN if not np.array_equal(got[k], want[k])]
    assert not bad, bad


def test_large_first_order_and_vector_stores_change_nothing(gpu, oracle):
    """k_blocked_order (batches from 512 rectangles on: large rectangles first, counting sort by size, the rest in creation order) only changes which workgroup takes
    which rectangle: a 2048 x 1024 image of noise with flat and gradient patches (tens of thousands of rectangles of every size) with and without it (test hook
    blocked_no_order) -- every plane equal, and equal to the oracle."""
    import torch
    W, H = 2048, 1024
    img = oracle.photo_noise(W, H, 5)
    img[128:640, 256:1280] = oracle.random_gradient(1024, 512, 6, True)
    img[700:900, 1400:2000] = np.uint32(0xFF406080)
    want = oracle.blocked_encode3d(img, True)
    sizes = want["regions"]["rx"] * want["regions"]["ry"]
    assert len(want["regions"]) > 4096 and int(sizes.max()) > 16 and int((sizes > 4).sum()) > 50
    outs = {}
    for no_order, no_vec in ((False, False), (True, False), (False, True)):
        gpu.set_options(test_blocked_no_order=no_order, test_blocked_no_vec_store=no_vec)
        try:
            outs[(no_order, no_vec)] = gpu.blocked_encode3d(img, True)
        finally:
            gpu.set_options()
    for k in BLOCKED_WRITTEN:
        assert np.array_equal(outs[(False, False)][k], want[k]), ("ordered, four pixels per lane in the store kernel (the product's path)", k)
        assert np.array_equal(outs[(True, False)][k], want[k]), ("creation order", k)
        assert np.array_equal(outs[(False, True)][k], want[k]), ("one pixel per lane in the store kernel", k)
    gpu.check()

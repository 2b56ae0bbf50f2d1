"""Parity tests proper: the HIP path (through the C ABI of liblimg_hip.so) against the CPU oracle on the same seeded
inputs and against the committed golden vectors of the real reference.  Bit-exact on every plane -- the float stage
included, because the kernels execute the reference's SSE arithmetic op for op (DESIGN.md "numerics")."""
import json
import os

import numpy as np
import pytest

import golden_util as gu
from oracle.bind import PLANES, REC_DTYPE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fused", "split", "legacy"])
def gpu(request):
    """`fused`  = k_fit_tpb (float stage, one lane per block) + the persistent kernel (decoupled look-back for the dither chain) where they apply;
    `split`  = the three-launch path (fit+search, scan, dither+store) that ragged images always take;
    `legacy` = the float stage inside the persistent kernel's E step with lane == pixel (the round-1 mapping, still what images with partial blocks run)."""
    import limg_amd
    g = limg_amd.LimgHip(0)  # raises if the HIP library or the device is missing: no fallback
    g.mode = request.param
    plain = g.set_options

    def set_options(**kw):  # every options change inside a test keeps the fixture's mode
        kw.setdefault("force_split", request.param == "split")
        kw["legacy_float_stage"] = request.param == "legacy"
        plain(**kw)
    g.set_options = set_options
    g.set_options()
    yield g
    g.check()
    g.close()


def _assert_planes(got, want, ctx):
    bad = [(k, int((got[k] != want[k]).sum())) for k in PLANES if not np.array_equal(got[k], want[k])]
    assert not bad, (ctx, bad)


def test_stagewise_small(gpu, oracle):
    """Localises a mismatch: records (float stage), then shifts, then the planes."""
    import torch
    for kind, alpha in (("pn", True), ("rg", True), ("rga", True), ("pn", False), ("rg", False)):
        img = {"pn": oracle.photo_noise(256, 64, 5), "rg": oracle.random_gradient(256, 64, 5, True), "rga": oracle.random_gradient(256, 64, 5, False)}[kind]
        want = oracle.encode3d(img, alpha, extras=True)
        d_img = torch.from_numpy(img.view(np.int32)).cuda()
        planes = gpu.alloc_planes_device(256, 64)
        rec = torch.zeros((8 * 32, 16), dtype=torch.int32, device="cuda")
        sh = torch.zeros(8 * 32, dtype=torch.int32, device="cuda")
        gpu.encode3d_device(d_img, alpha, planes, records=rec, shifts=sh)
        torch.cuda.synchronize()
        grec = rec.cpu().numpy().view(REC_DTYPE).reshape(8, 32)
        for f in REC_DTYPE.names:
            assert np.array_equal(grec[f], want["records"][f]), (kind, alpha, f, np.argwhere(grec[f] != want["records"][f])[:4])
        gsh = sh.cpu().numpy().astype(np.uint32).reshape(8, 32)
        for i in range(3):
            assert np.array_equal((gsh >> (8 * i)) & 0xFF, want["shifts"][:, :, i]), (kind, alpha, "shift", i)
        got = {k: v.cpu().numpy().view(np.uint32 if v.dtype == torch.int32 else np.uint8) for k, v in planes.items()}
        _assert_planes(got, want, (kind, alpha))


def test_golden_cases(gpu):
    idx, z = gu.cases()
    for i, m in enumerate(idx):
        gpu.set_options(force_split=(gpu.mode == "split"), dither_pcg=(m["dither"] != 0))
        try:
            got = gpu.encode3d(z["c%02d_in" % i], m["alpha"], error_factor=m["ef"], pool_threads=m["pool"], fast=m["fast"])
        finally:
            gpu.set_options(force_split=(gpu.mode == "split"))
        want = {k: z["c%02d_%s" % (i, k)] for k in PLANES}
        _assert_planes(got, want, (i, m))


@pytest.mark.parametrize("name", ["original_rgb", "original_as_rgba", "rg1024", "rga1024", "pn1024", "pn1024_ef25", "pn1024_pool2", "pn1024_pcg", "original_rgb_ef0", "pn1024_accurate", "rg1024_accurate"])
def test_full_image_hashes(gpu, oracle, name):
    """Plane hashes of the real reference on original.png (config #1) and on 1024x1024 of each synthetic generator."""
    e = gu.hashes()[name]
    img = gu.big_input(name, oracle)
    kw = dict(e["kw"])
    pcg = kw.pop("dither_mode", 0) != 0
    gpu.set_options(force_split=(gpu.mode == "split"), dither_pcg=pcg)
    try:
        got = gpu.encode3d(img, e["alpha"], **kw)
    finally:
        gpu.set_options(force_split=(gpu.mode == "split"))
    for k in PLANES:
        assert oracle.fnv(got[k]) == e[k], (name, k)
    psnr, mse = gpu.compare(img, got["pDecoded"], e["alpha"])
    assert psnr == pytest.approx(e["psnr"], abs=1e-9) and mse == pytest.approx(e["mse"], rel=1e-12)


# the last ones have a corner block of fewer than 4 pixels: upstream's sum loop consumes 4 pixels regardless (src/limg.cpp:478-487) and so picks up what the
# previous block left in the gather buffer; oracle (pinned to the reference there too) and kernels reproduce that.  (3, 1) and (1, 1): no previous block.
@pytest.mark.parametrize("w,h", [(8, 8), (5, 3), (61, 27), (264, 16), (256, 8), (1000, 40), (4, 100), (2048, 8), (9, 9), (17, 10), (3, 1), (1, 1), (2, 65), (10, 9), (25, 33),
                                 (1, 17), (17, 1), (265, 9), (257, 17)])
@pytest.mark.parametrize("alpha", [True, False])
def test_ragged_and_edge_shapes(gpu, oracle, w, h, alpha):
    img = oracle.photo_noise(w, h, 13)
    _assert_planes(gpu.encode3d(img, alpha), oracle.encode3d(img, alpha), (w, h, alpha))
    if h >= 16:  # also with a strip partition (the corner block's predecessor must come from its own strip)
        _assert_planes(gpu.encode3d(img, alpha, pool_threads=1), oracle.encode3d(img, alpha, pool_threads=1), (w, h, alpha, "pool"))


@pytest.mark.parametrize("w,h", [(8, 9), (256, 12), (264, 17), (2048, 15), (512, 100), (1024, 301), (40, 1001)])
@pytest.mark.parametrize("alpha", [True, False])
def test_height_ragged_shapes(gpu, oracle, w, h, alpha):
    """Width in whole 8x8 blocks, last block row partial (BASELINE config 1's class: 1024 x 618).  In the `fused` fixture mode the block rows above the last run
    through k_fit_tpb + the persistent kernel and only the last row's dither chain is walked by the host, from the chain value the look-back descriptor of the
    strip above names (encode_height_ragged); the other modes and test_whole_image_ragged take the whole-image ragged path.  All against the oracle -- with strip
    partitions (the last strip owns the partial row), the accurate search, errorFactor 0 -- and the two paths against each other."""
    img = oracle.photo_noise(w, h, 31) if w != 512 else oracle.random_gradient(w, h, 31, True)
    want = oracle.encode3d(img, alpha)
    _assert_planes(gpu.encode3d(img, alpha), want, (w, h, alpha))
    gpu.set_options(test_whole_image_ragged=True)
    try:
        _assert_planes(gpu.encode3d(img, alpha), want, (w, h, alpha, "whole-image path"))
    finally:
        gpu.set_options()
    for pool in ((1, 2, 3) if h >= 100 else (1,) if h >= 16 else ()):
        _assert_planes(gpu.encode3d(img, alpha, pool_threads=pool), oracle.encode3d(img, alpha, pool_threads=pool), (w, h, alpha, "pool", pool))
    if h == 100:
        _assert_planes(gpu.encode3d(img, alpha, fast=False), oracle.encode3d(img, alpha, fast=False), (w, h, alpha, "accurate"))
        _assert_planes(gpu.encode3d(img, alpha, error_factor=0), oracle.encode3d(img, alpha, error_factor=0), (w, h, alpha, "ef0"))
        _assert_planes(gpu.encode3d(img, alpha, error_factor=25), oracle.encode3d(img, alpha, error_factor=25), (w, h, alpha, "ef25"))


@pytest.mark.parametrize("w,h,alpha", [(510, 520, True), (509, 515, False), (1022, 1024, True)])
def test_width_ragged_band_pipeline(gpu, oracle, w, h, alpha):
    """A partial last block COLUMN puts the whole dither chain on the host (reference: a call over rx x ry pixels is N / 8 AES rounds + N % 8 PCG steps, src/limg.cpp:824-879,
    :1899-1905).  From 64 x 32 blocks on the walk is pipelined with the GPU in bands of block rows (limg_hip_options.ragged_bands): automatic, off, 3 and 64 bands must all
    give the oracle's planes."""
    img = oracle.photo_noise(w, h, 47)
    want = oracle.encode3d(img, alpha)
    for bands in (0, -1, 3, 64):
        gpu.set_options(ragged_bands=bands)
        try:
            _assert_planes(gpu.encode3d(img, alpha), want, (w, h, alpha, "bands", bands))
        finally:
            gpu.set_options()
    _assert_planes(gpu.encode3d(img, alpha, error_factor=25), oracle.encode3d(img, alpha, error_factor=25), (w, h, alpha, "ef25"))


@pytest.mark.parametrize("w,h", [(300, 1000), (1022, 517)])
def test_width_ragged_parallel_chain_walks(gpu, oracle, w, h):
    """With a thread pool the reference's strips restart the chain (src/limg.cpp:2114-2134): independent chains, walked on several host threads at once
    (limg_hip_options.ragged_walk_threads: automatic, serial, 3) -- same planes as the oracle's strip partition."""
    img = oracle.photo_noise(w, h, 53)
    for pool in (1, 2, 3):
        want = oracle.encode3d(img, True, pool_threads=pool)
        for threads in (0, 1, 3):
            gpu.set_options(ragged_walk_threads=threads)
            try:
                _assert_planes(gpu.encode3d(img, True, pool_threads=pool), want, (w, h, "pool", pool, "threads", threads))
            finally:
                gpu.set_options()


def test_height_ragged_at_4096(gpu, oracle):
    """4096 x 4090 photo-noise: fast path + last row against the whole-image ragged path of the same build, every plane compared on the device; the bottom band
    (rows 3968 ...: the last full block rows and the partial one) cannot be checked by the oracle alone -- its chain position depends on everything above -- so the
    whole image goes through the oracle once (a few seconds)."""
    import torch
    if gpu.mode != "fused":
        pytest.skip("one mode is enough at this size")
    W, H = 4096, 4090
    img = gpu.synth_device("photo_noise", W, H, seed=3)
    a = gpu.alloc_planes_device(W, H)
    b = gpu.alloc_planes_device(W, H)
    gpu.encode3d_device(img, True, a)
    gpu.set_options(test_whole_image_ragged=True)
    try:
        gpu.encode3d_device(img, True, b)
    finally:
        gpu.set_options()
    torch.cuda.synchronize()
    gpu.check()
    for k in PLANES:
        assert torch.equal(a[k], b[k]), k
    host = img.cpu().numpy().view(np.uint32)
    want = oracle.encode3d(host, True, worker_threads=8)
    for k in PLANES:
        got = a[k].cpu().numpy()
        got = got.view(np.uint32) if got.dtype == np.int32 else got
        assert np.array_equal(got, want[k]), k


@pytest.mark.parametrize("ef", [0, 1, 2, 25, 50, 100, 200, 400, 3000, 4000000000])
def test_error_factor_sweep(gpu, oracle, ef):
    img = oracle.photo_noise(256, 32, 17)
    _assert_planes(gpu.encode3d(img, True, error_factor=ef), oracle.encode3d(img, True, error_factor=ef), ef)


@pytest.mark.parametrize("bits", [8, 7, 6, 5, 4, 3, 2, 1, 0])
def test_forced_shift_sweep(gpu, oracle, bits):
    """BASELINE.json configs[2] 'bit-crush sweep': the search bypassed with shift = 8 - bits on all three factors."""
    s = 8 - bits
    img = oracle.photo_noise(256, 32, 19)
    gpu.set_options(forced_shift=(s, s, s), force_split=(gpu.mode == "split"))
    try:
        got = gpu.encode3d(img, True)
    finally:
        gpu.set_options(force_split=(gpu.mode == "split"))
    _assert_planes(got, oracle.encode3d(img, True, forced_shift=(s, s, s)), bits)


@pytest.mark.parametrize("pool", [1, 2, 3, 8])
def test_strip_restart_chains(gpu, oracle, pool):
    img = oracle.photo_noise(256, 264, 23)
    _assert_planes(gpu.encode3d(img, True, pool_threads=pool), oracle.encode3d(img, True, pool_threads=pool), pool)


@pytest.mark.parametrize("alpha", [True, False])
def test_accurate_mode(gpu, oracle, alpha):
    img = oracle.photo_noise(256, 24, 29)
    _assert_planes(gpu.encode3d(img, alpha, fast=False), oracle.encode3d(img, alpha, fast=False), alpha)


@pytest.mark.parametrize("w,h", [(256, 64), (61, 27)])
def test_pcg_dither(gpu, oracle, w, h):
    """a14: the reference's non-AES dither (PCG) -- full blocks through the noise table, ragged through the chain walk."""
    from oracle.bind import DITHER_PCG
    img = oracle.photo_noise(w, h, 31)
    gpu.set_options(force_split=(gpu.mode == "split"), dither_pcg=True)
    try:
        got = gpu.encode3d(img, True)
    finally:
        gpu.set_options(force_split=(gpu.mode == "split"))
    _assert_planes(got, oracle.encode3d(img, True, dither_mode=DITHER_PCG), (w, h))


@pytest.mark.parametrize("limit", [1, 120])
@pytest.mark.parametrize("alpha,fast", [(True, True), (False, True), (True, False)])
def test_generic_trial_path(gpu, oracle, alpha, fast, limit):
    """The packed 16-bit trial is exact by construction for record values up to 2700 in magnitude (3 * 2700 + 1 < 0x2000, limg_hip_kernels.hip); larger ones
    (never produced by a fit of byte pixels, which stays below 2041) take a generic 32-bit trial.  The test hook lowers the limit: 1 sends every block through
    the generic path, 120 mixes the two paths inside every work strip (the dynamic block queue hands both kinds to every wave)."""
    img = oracle.photo_noise(256, 32, 37)
    gpu.set_options(force_split=(gpu.mode == "split"), test_record_limit=limit)
    try:
        got = gpu.encode3d(img, alpha, fast=fast)
    finally:
        gpu.set_options(force_split=(gpu.mode == "split"))
    _assert_planes(got, oracle.encode3d(img, alpha, fast=fast), (alpha, fast))


def test_degenerate_blocks(gpu, oracle):
    """flat blocks (dirA == 0), single-line blocks (dirB == 0), planes (dirC noise), extreme values."""
    img = np.zeros((16, 256), dtype=np.uint32)
    img[:, :] = 0xFF804020
    x = np.arange(256, dtype=np.uint32)[None, :]
    y = np.arange(16, dtype=np.uint32)[:, None]
    img[0:8, 8:16] = ((10 + 20 * (x[:, 8:16] - 8)) | ((200 - 10 * (x[:, 8:16] - 8)) << 8) | ((50 + 5 * (x[:, 8:16] - 8)) << 16) | (255 << 24))
    img[0:8, 16:24] = ((10 + 20 * (x[:, 16:24] - 16)) | ((30 + 25 * y[0:8]) << 8) | ((50 + 5 * (x[:, 16:24] - 16) + 3 * y[0:8]) << 16) | (255 << 24)).astype(np.uint32)
    img[0:8, 24:32] = 0
    img[0:8, 32:40] = 0xFFFFFFFF
    img[0:8, 40:48] = np.where((x[:, 40:48] + y[0:8]) % 2 == 0, 0, 0xFFFFFFFF).astype(np.uint32)
    rng = np.random.default_rng(3)
    img[8:16, :] = rng.integers(0, 2**32, (8, 256), dtype=np.uint32)
    for alpha in (True, False):
        _assert_planes(gpu.encode3d(img, alpha), oracle.encode3d(img, alpha), alpha)


def test_random_bytes(gpu, oracle):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 2**32, (64, 512), dtype=np.uint32)
    for alpha in (True, False):
        _assert_planes(gpu.encode3d(img, alpha), oracle.encode3d(img, alpha), alpha)


def test_device_synth_matches_host(gpu, oracle):
    import torch
    a = gpu.synth_device("random_gradient", 512, 256, seed=1, opaque=True).cpu().numpy().view(np.uint32)
    assert np.array_equal(a, oracle.random_gradient(512, 256, 1, True))
    b = gpu.synth_device("photo_noise", 512, 256, seed=1).cpu().numpy().view(np.uint32)
    assert np.array_equal(b, oracle.photo_noise(512, 256, 1))
    c = gpu.synth_device("photo_noise", 512, 64, seed=1, y0=64).cpu().numpy().view(np.uint32)
    assert np.array_equal(c, oracle.photo_noise(512, 256, 1)[64:128])
    torch.cuda.synchronize()


def test_full_size_properties(gpu, oracle):
    """BASELINE sizes (4096^2 gradient, 8192^2 photo-noise): size-independent properties instead of a full CPU run:
    (1) a 256-row band re-encoded alone by the oracle matches the same band of the full-image GPU result in every
        chain-independent plane (shifts, extrema) and -- for the first band, whose dither chain starts at seed0 -- all planes;
    (2) strip-restart encode (pool) == independent encodes of the strips;  (3) PSNR matches the reference's figure."""
    import torch
    for kind, W, psnr_ref in (("random_gradient", 4096, 50.38), ("photo_noise", 8192, 38.87)):
        d_img = gpu.synth_device(kind, W, W, seed=1)
        planes = gpu.alloc_planes_device(W, W)
        gpu.encode3d_device(d_img, True, planes)
        torch.cuda.synchronize()
        psnr, _ = gpu.compare_device(d_img, planes["pDecoded"], True)
        assert abs(psnr - psnr_ref) < 0.05, (kind, psnr)
        band = d_img[:256].cpu().numpy().view(np.uint32)
        want = oracle.encode3d(band, True)
        for k in PLANES:
            got = planes[k][:256].cpu().numpy()
            got = got.view(np.uint32) if got.dtype == np.int32 else got
            assert np.array_equal(got, want[k]), (kind, k)
        mid = d_img[W // 2: W // 2 + 64].cpu().numpy().view(np.uint32)
        want = oracle.encode3d(mid, True)
        for k in ("pShiftABCX", "pColAMin", "pColAMax", "pColBMin", "pColBMax", "pColCMin", "pColCMax"):
            assert np.array_equal(planes[k][W // 2: W // 2 + 64].cpu().numpy().view(np.uint32), want[k]), (kind, k)
        del planes
        # (2) 8 strips == the 8-GPU strip-restart semantics (pool of 2 threads)
        planes = gpu.alloc_planes_device(W, W)
        gpu.encode3d_device(d_img, True, planes, pool_threads=2)
        rows = (W // 8 // 8) * 8
        part = gpu.alloc_planes_device(W, rows)
        for s in (0, 3, 7):
            gpu.encode3d_device(d_img[s * rows:(s + 1) * rows], True, part)
            torch.cuda.synchronize()
            for k in PLANES:
                assert torch.equal(part[k], planes[k][s * rows:(s + 1) * rows]), (kind, s, k)
        del planes, part, d_img
        torch.cuda.empty_cache()


def test_error_codes(gpu):
    """Same enum values as limg_result (src/limg.h:9-18): null -> ArgumentNull (102), bad sizes -> InvalidParameter (101)."""
    import ctypes as C
    import limg_amd
    L = gpu.lib
    img = np.zeros((8, 8), dtype=np.uint32)
    planes = {k: np.zeros((8, 8), dtype=np.uint32 if k in limg_amd.P32 else np.uint8) for k in limg_amd.PLANES}
    info = limg_amd.Info(*[planes[k].ctypes.data for k in limg_amd.PLANES])
    assert L.limg_hip_encode3d(gpu.ctx, None, 8, 8, 1, C.byref(info), 100, 0, 1) == 102
    assert L.limg_hip_encode3d(gpu.ctx, img.ctypes.data_as(C.c_void_p), 8, 8, 1, None, 100, 0, 1) == 102
    assert L.limg_hip_encode3d(None, img.ctypes.data_as(C.c_void_p), 8, 8, 1, C.byref(info), 100, 0, 1) == 102
    assert L.limg_hip_encode3d(gpu.ctx, img.ctypes.data_as(C.c_void_p), 0, 8, 1, C.byref(info), 100, 0, 1) == 101
    bad = limg_amd.Info(*[planes[k].ctypes.data for k in limg_amd.PLANES])
    bad.pFactorsC = None
    assert L.limg_hip_encode3d(gpu.ctx, img.ctypes.data_as(C.c_void_p), 8, 8, 1, C.byref(bad), 100, 0, 1) == 102
    assert L.limg_hip_encode3d_perf(gpu.ctx, None, 8, 8, 1, 100, 0, 1) == 102
    assert L.limg_hip_encode3d(gpu.ctx, img.ctypes.data_as(C.c_void_p), 8, 8, 1, C.byref(info), 100, 0, 1) == 0


def test_perf_entry_and_compact_outputs(gpu, oracle):
    """`_perf` behaviour (no planes) still produces the per-block compact outputs when asked: records + shifts == oracle."""
    import torch
    img = oracle.photo_noise(256, 32, 41)
    gpu.encode3d_perf(img, True)  # must simply succeed
    want = oracle.encode3d(img, True, extras=True)
    d_img = torch.from_numpy(img.view(np.int32)).cuda()
    rec = torch.zeros((4 * 32, 16), dtype=torch.int32, device="cuda")
    sh = torch.zeros(4 * 32, dtype=torch.int32, device="cuda")
    gpu.encode3d_device(d_img, True, None, records=rec, shifts=sh)
    torch.cuda.synchronize()
    grec = rec.cpu().numpy().view(REC_DTYPE).reshape(4, 32)
    for f in REC_DTYPE.names:
        assert np.array_equal(grec[f], want["records"][f]), f
    gsh = sh.cpu().numpy().astype(np.uint32).reshape(4, 32)
    for i in range(3):
        assert np.array_equal((gsh >> (8 * i)) & 0xFF, want["shifts"][:, :, i])


def test_repeatability_and_context_reuse(gpu, oracle):
    """Back-to-back encodes of different sizes through one context (buffers and the noise table are reused / regrown)."""
    for (w, h) in ((512, 64), (64, 512), (1024, 1024), (512, 64)):
        img = oracle.photo_noise(w, h, w + h)
        a = gpu.encode3d(img, True)
        b = gpu.encode3d(img, True)
        for k in PLANES:
            assert np.array_equal(a[k], b[k]), k
        assert oracle.fnv(a["pDecoded"]) == oracle.fnv(oracle.encode3d(img, True)["pDecoded"])


def test_compact_mode(gpu, oracle):
    """Compact outputs (SURVEY 8(d), 8.05 B/px): factor planes + records + shift words only; identical to the full run's."""
    import torch
    import limg_amd
    img = oracle.photo_noise(512, 64, 43)
    want = oracle.encode3d(img, True, extras=True)
    d_img = torch.from_numpy(img.view(np.int32)).cuda()
    fac = {k: torch.zeros((64, 512), dtype=torch.uint8, device="cuda") for k in limg_amd.P8}
    rec = torch.zeros((8 * 64, 16), dtype=torch.int32, device="cuda")
    sh = torch.zeros(8 * 64, dtype=torch.int32, device="cuda")
    gpu.encode3d_device(d_img, True, fac, records=rec, shifts=sh)
    torch.cuda.synchronize()
    for k in limg_amd.P8:
        assert np.array_equal(fac[k].cpu().numpy(), want[k]), k
    grec = rec.cpu().numpy().view(REC_DTYPE).reshape(8, 64)
    for f in REC_DTYPE.names:
        assert np.array_equal(grec[f], want["records"][f]), f
    gsh = sh.cpu().numpy().astype(np.uint32).reshape(8, 64)
    for i in range(3):
        assert np.array_equal((gsh >> (8 * i)) & 0xFF, want["shifts"][:, :, i])


def test_two_contexts_on_two_threads(oracle):
    """Contexts are independent: two host threads, each with its own context, encode different images at the same time (8x8 path and merged-block encoder)."""
    import threading
    import limg_amd
    from oracle.bind import BLOCKED_WRITTEN
    imgs = [oracle.photo_noise(512, 256, 21), oracle.random_gradient(384, 264, 22, False)]
    want = [oracle.encode3d(i, True) for i in imgs]
    wantb = [oracle.blocked_encode3d(i, True) for i in imgs]
    errs = []

    def work(k):
        try:
            g = limg_amd.LimgHip(0)
            for _ in range(5):
                got = g.encode3d(imgs[k], True)
                bad = [p for p in PLANES if not np.array_equal(got[p], want[k][p])]
                gotb = g.blocked_encode3d(imgs[k], True)
                bad += [p for p in BLOCKED_WRITTEN if not np.array_equal(gotb[p], wantb[k][p])]
                if bad:
                    errs.append((k, bad))
            g.check()
            g.close()
        except Exception as e:  # noqa: BLE001
            errs.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_unaligned_device_pointers(gpu, oracle):
    """Device pointers that are only 4-byte (uint32 planes, input) / 1-byte (factor planes) aligned -- slices of larger allocations -- take the kernels'
    dword / byte paths and give the same planes (include/limg_hip.h "Alignment")."""
    import torch
    W, H = 512, 64
    img = oracle.photo_noise(W, H, 77)
    want = oracle.encode3d(img, True)
    buf = torch.zeros(W * H + 8, dtype=torch.int32, device="cuda")
    d_img = buf[1:1 + W * H].view(H, W)
    d_img.copy_(torch.from_numpy(img.view(np.int32)))
    assert d_img.data_ptr() % 16 == 4
    planes = {}
    for k in PLANES:
        dt = torch.uint8 if k.startswith("pFactors") else torch.int32
        b = torch.zeros(W * H + 32, dtype=dt, device="cuda")
        off = {"pFactorsA": 1, "pFactorsB": 7, "pFactorsC": 16}.get(k, 3)
        planes[k] = b[off:off + W * H].view(H, W)
    gpu.encode3d_device(d_img, True, planes)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy().view(np.uint32 if v.dtype == torch.int32 else np.uint8) for k, v in planes.items()}
    _assert_planes(got, want, "unaligned")


def test_config4_batch_rehearsal(gpu, oracle):
    """BASELINE config 4 (batch of 4096^2 random-gradient images, seeds 1.., 8 per GPU) as one rank of the 8-GPU job sees it: its 8 images through ONE context, all
    eight ENQUEUED back to back on one stream before any is looked at (the context's scratch is reused launch after launch), then every plane of every image against
    the real reference's whole-image checksums (tests/golden/fullsize.json rg4096_batch64; the batched entry over all 64 images: tests/test_gpu_fullsize.py)."""
    import json
    import os
    import torch
    import golden_util as gu
    from limg_amd.shard import sum64_device
    gold = json.load(open(os.path.join(gu.G, "fullsize.json")))["rg4096_batch64"]["images"][:8]
    W = 4096
    imgs = [gpu.synth_device("random_gradient", W, W, seed=im["seed"]) for im in gold]
    outs = [gpu.alloc_planes_device(W, W) for _ in range(8)]
    for d_img, planes in zip(imgs, outs):
        gpu.encode3d_device(d_img, True, planes)
    torch.cuda.synchronize()
    gpu.check()
    bad = [(im["seed"], k) for im, planes in zip(gold, outs) for k in PLANES if sum64_device(planes[k]) != im["sum64"][k]]
    assert not bad, bad
    del imgs, outs
    torch.cuda.empty_cache()


def test_config5_strip_sharded_rehearsal(gpu, oracle):
    """BASELINE config 5 (16384^2 photo-noise, 8 strips of whole block rows = shard.strip_rows(16384, 8)) rehearsed on one GPU:
    (1) the strip-restart encode of the whole image (pool of 2 threads = 8 chains, src/limg.cpp:2114-2134) equals independent encodes of strips 0 / 3 / 7,
        i.e. what ranks 0 / 3 / 7 of the 8-GPU job produce without any exchange;
    (2) the first 256 rows equal the oracle on every plane;
    (3) `--gather-stream` reassembly: every strip's compact stream, decoded into its rows of the full image, reproduces pDecoded."""
    import torch
    from limg_amd import shard
    W = 16384
    rows = shard.strip_rows(W, 8)
    assert rows == [(i * 2048, (i + 1) * 2048) for i in range(8)] and shard.equivalent_pool_threads(8) == 2
    d_img = gpu.synth_device("photo_noise", W, W, seed=1)
    planes = gpu.alloc_planes_device(W, W)
    gpu.encode3d_device(d_img, True, planes, pool_threads=2)
    torch.cuda.synchronize()
    gpu.check()
    psnr, _ = gpu.compare_device(d_img, planes["pDecoded"], True)
    assert 38.0 < psnr < 40.0, psnr
    band = d_img[:256].cpu().numpy().view(np.uint32)
    want = oracle.encode3d(band, True)
    for k in PLANES:
        got = planes[k][:256].cpu().numpy()
        got = got.view(np.uint32) if got.dtype == np.int32 else got
        assert np.array_equal(got, want[k]), k
    part = gpu.alloc_planes_device(W, 2048)
    for s in (0, 3, 7):
        y0, y1 = rows[s]
        gpu.encode3d_device(d_img[y0:y1], True, part)  # a rank's own strip: fresh chain, no exchange
        torch.cuda.synchronize()
        for k in PLANES:
            assert torch.equal(part[k], planes[k][y0:y1]), (s, k)
    del part
    # (3) only the compact streams would cross xGMI: decode each into its rows of one full-size image
    full = torch.zeros((W, W), dtype=torch.int32, device="cuda")
    total = 0
    for s, (y0, y1) in enumerate(rows):
        st, n = gpu.encode_stream_device(d_img[y0:y1], True)
        gpu.decode_stream_device(st, n, W, y1 - y0, out=full[y0:y1])
        total += n
    torch.cuda.synchronize()
    gpu.check()
    assert torch.equal(full, planes["pDecoded"])
    assert total < W * W * 3  # ~1.8 B/px of stream instead of 35 B/px of planes
    del full, planes, d_img
    torch.cuda.empty_cache()


def test_host_entry_at_4096(gpu, oracle):
    """The drop-in entry itself (host pointers: what the shim's limg_encode3d_test calls) at a BASELINE size: every plane equals the device entry's, the first
    band equals the oracle, and the reference's PSNR figure for config 2 comes out of limg_hip_compare on host pointers."""
    import torch
    W = 4096
    d_img = gpu.synth_device("random_gradient", W, W, seed=1)
    img = d_img.cpu().numpy().view(np.uint32)
    got = gpu.encode3d(img, True)
    planes = gpu.alloc_planes_device(W, W)
    gpu.encode3d_device(d_img, True, planes)
    torch.cuda.synchronize()
    for k in PLANES:
        dev = planes[k].cpu().numpy()
        dev = dev.view(np.uint32) if dev.dtype == np.int32 else dev
        assert np.array_equal(got[k], dev), k
    want = oracle.encode3d(np.ascontiguousarray(img[:128]), True)
    for k in PLANES:
        assert np.array_equal(got[k][:128], want[k]), k
    psnr, _ = gpu.compare(img, got["pDecoded"], True)
    assert abs(psnr - 50.38) < 0.05
    del planes, d_img
    torch.cuda.empty_cache()


@pytest.mark.parametrize("shape", [(4096, 2048), (2056, 4104)])
@pytest.mark.parametrize("pool", [0, 2, 3, 16])
def test_host_entry_in_bands(gpu, oracle, pool, shape):
    """limg_hip_encode3d from 4 Mpixels on works in row bands (upload + kernels of band k + 1 under the download of band k; limg_hip_api.hip host_encode_banded): one
    chain through the bands (poolThreads 0: the chain entry's two halves per band, bases on the device) and a band per restarted chain (poolThreads > 0,
    src/limg.cpp:2114-2134; 16 threads = 64 chains of 4 block rows: the plain path).  Every plane == the device entry's single encode of the whole image."""
    import torch
    W, H = shape  # (the second: 513 block rows -- bands and chains of unequal height -- and a width whose rows are not 16-byte multiples of the strip width)
    d_img = gpu.synth_device("photo_noise", W, H, seed=9)
    img = d_img.cpu().numpy().view(np.uint32)
    got = gpu.encode3d(img, True, pool_threads=pool)
    planes = gpu.alloc_planes_device(W, H)
    gpu.encode3d_device(d_img, True, planes, pool_threads=pool)
    torch.cuda.synchronize()
    gpu.check()
    for k in PLANES:
        dev = planes[k].cpu().numpy()
        dev = dev.view(np.uint32) if dev.dtype == np.int32 else dev
        assert np.array_equal(got[k], dev), (pool, k)
    want = oracle.encode3d(np.ascontiguousarray(img[:64]), True)
    if pool == 0:
        for k in PLANES:
            assert np.array_equal(got[k][:64], want[k]), k
    del planes, d_img
    torch.cuda.empty_cache()


def test_repeat_determinism_at_4096(gpu):
    """The same 4096^2 gradient image 32 times through one context: every run must give the same planes.  Fast searches (2 trials per block) and 16 K work
    strips per image are what exposed a lost update in the persistent kernel (a late store zeroing an already parked shift word, about one strip in 500 K):
    invisible to band-limited full-size checks, so all planes of all runs are compared here, on the device."""
    import torch
    W = 4096
    d_img = gpu.synth_device("random_gradient", W, W, seed=3)
    ref = gpu.alloc_planes_device(W, W)
    gpu.encode3d_device(d_img, True, ref)
    cur = gpu.alloc_planes_device(W, W)
    for it in range(32):
        gpu.encode3d_device(d_img, True, cur)
        torch.cuda.synchronize()
        for k in PLANES:
            assert torch.equal(cur[k], ref[k]), (it, k)
    gpu.check()
    del ref, cur, d_img
    torch.cuda.empty_cache()


def test_config3_bit_crush_sweep_at_8192(gpu, oracle):
    """BASELINE config 3 at its real size, what the whole-image reference hashes do NOT cover (tests/test_gpu_fullsize.py pins forced shifts 0 .. 6 and errorFactor
    0 / 25 / 50 / 100 / 200 / 400 plane by plane): the two remaining forced shifts, 7 (one bit per factor) and 8 (none: no dither call at all), band-limited against the
    oracle, and that PSNR falls monotonically over the whole forced sweep."""
    import torch
    W = 8192
    d_img = gpu.synth_device("photo_noise", W, W, seed=1)
    band = d_img[:64].cpu().numpy().view(np.uint32)
    planes = gpu.alloc_planes_device(W, W)
    last_psnr = 1e9
    try:
        for s in range(9):
            gpu.set_options(forced_shift=(s, s, s))
            gpu.encode3d_device(d_img, True, planes)
            torch.cuda.synchronize()
            if s >= 7:
                want = oracle.encode3d(band, True, forced_shift=(s, s, s))
                for k in PLANES:
                    got = planes[k][:64].cpu().numpy()
                    got = got.view(np.uint32) if got.dtype == np.int32 else got
                    assert np.array_equal(got, want[k]), (s, k)
            psnr, _ = gpu.compare_device(d_img, planes["pDecoded"], True)
            assert psnr < last_psnr + 1e-9, (s, psnr, last_psnr)
            last_psnr = psnr
    finally:
        gpu.set_options()
    gpu.check()
    del planes, d_img
    torch.cuda.empty_cache()


def test_accurate_mode_at_4096(gpu, oracle):
    """`--accurate-bit-crushing` (fastBitCrushing = false, src/limg_bit_crush.h:668-830) on a shape the whole-image reference hashes do not hold (they pin 8192^2
    photo-noise and 4096^2 random-gradient: tests/test_gpu_fullsize.py pn8192_accurate / rg4096_accurate): a 4096 x 1024 photo-noise image, first band on every plane
    and a middle band on the chain-independent planes against the oracle's accurate search."""
    import torch
    for kind, W, H in (("photo_noise", 4096, 1024),):
        d_img = gpu.synth_device(kind, W, H, seed=2)
        planes = gpu.alloc_planes_device(W, H)
        gpu.encode3d_device(d_img, True, planes, fast=False)
        torch.cuda.synchronize()
        band = d_img[:32].cpu().numpy().view(np.uint32)
        want = oracle.encode3d(band, True, fast=False)
        for k in PLANES:
            got = planes[k][:32].cpu().numpy()
            got = got.view(np.uint32) if got.dtype == np.int32 else got
            assert np.array_equal(got, want[k]), (kind, k)
        mid = d_img[H // 2: H // 2 + 16].cpu().numpy().view(np.uint32)
        want = oracle.encode3d(mid, True, fast=False)
        for k in ("pShiftABCX", "pColAMin", "pColAMax", "pColBMin", "pColBMax", "pColCMin", "pColCMax"):
            assert np.array_equal(planes[k][H // 2: H // 2 + 16].cpu().numpy().view(np.uint32), want[k]), (kind, k)
        del planes, d_img
    gpu.check()
    torch.cuda.empty_cache()

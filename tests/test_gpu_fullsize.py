"""The BASELINE workloads AT THEIR OWN SIZE against the REAL reference: tests/golden/fullsize.json holds FNV-1a-64 hashes of every plane `limg_encode3d_test`
(src/limg.cpp:2175-2265) and `limg_blocked_encode3d_test` (:2329-2453) write for the synthetic 8192^2 / 4096^2 / 16384 x 2048 inputs, generated in the build
container by tools/make_golden_fullsize.py from oracle/_ref (the reference compiled from /root/reference/src).  The device planes are hashed whole: the dither chain
of an 8192^2 image runs through 32 768 work strips of the persistent kernel's look-back, and one wrong chain base anywhere changes the factor and decoded planes
from there on -- test_one_strip_base_error_is_caught shows that the hashes see exactly that (and that the band-limited checks of tests/test_gpu_parity.py do not).
"""
import json
import os

import numpy as np
import pytest

import golden_util as gu
from oracle.bind import BLOCKED_WRITTEN, PLANES

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(gu.G, "fullsize.json")))
KIND = {"pn": "photo_noise", "rg": "random_gradient"}


@pytest.fixture(scope="module")
def gpu():
    import limg_amd
    g = limg_amd.LimgHip(0)
    yield g
    g.check()
    g.close()


def _host(t):
    a = t.cpu().numpy()
    return a.view(np.uint32) if a.dtype == np.int32 else a


@pytest.mark.parametrize("name", sorted(k for k, e in GOLD.items() if e["kind"] == "encode3d"))
def test_encode3d_reference_hashes(gpu, oracle, name):
    """every plane of the fused path (k_fit_tpb + persistent kernel) == the real reference, pool_threads 0 (one chain) and 2 (eight chains)"""
    import torch
    e = GOLD[name]
    img = gpu.synth_device(KIND[e["gen"]], e["w"], e["h"], seed=e["seed"])
    assert oracle.fnv(_host(img)) == e["input"], "the device generator and the golden input disagree"
    planes = gpu.alloc_planes_device(e["w"], e["h"])
    gpu.set_options(**e.get("options", {}))  # (e.g. the PCG dither: dither_pcg)
    try:
        gpu.encode3d_device(img, e["alpha"], planes, **e["kw"])
        torch.cuda.synchronize()
        gpu.check()
    finally:
        gpu.set_options()
    bad = [k for k in PLANES if oracle.fnv(_host(planes[k])) != e["planes"][k]]
    assert not bad, (name, bad)
    psnr, mse = gpu.compare_device(img, planes["pDecoded"], e["alpha"])
    assert psnr == pytest.approx(e["psnr"], abs=1e-9) and mse == pytest.approx(e["mse"], rel=1e-12)
    del planes, img
    torch.cuda.empty_cache()


PATTERN = (0, 0x22, 0x44, 0x66, 0x88, 0xAA, 0xCC, 0xEE, 0xFF)  # src/limg.cpp:2006: what pShiftABCX shows for a shift


@pytest.mark.parametrize("name", sorted(k for k, e in GOLD.items() if e["kind"] == "forced"))
def test_forced_shift_reference_hashes(gpu, oracle, name):
    """BASELINE config 3's forced-shift half WHOLE (8192^2, bits 8 .. 2 on all three factors): the planes that depend on the shift against the reference's own block
    functions run with the search left out (oracle/ref_harness.cpp ref_encode3d_forced_shift: upstream has no such switch); the six colour planes against the
    adaptive encode of the same image (they do not depend on the shift); pShiftABCX is one constant."""
    import torch
    e = GOLD[name]
    base = GOLD[e["colour_planes_of"]]
    img = gpu.synth_device(KIND[e["gen"]], e["w"], e["h"], seed=e["seed"])
    assert oracle.fnv(_host(img)) == e["input"] == base["input"]
    planes = gpu.alloc_planes_device(e["w"], e["h"])
    gpu.set_options(forced_shift=tuple(e["shift"]))
    try:
        gpu.encode3d_device(img, e["alpha"], planes)
        torch.cuda.synchronize()
        gpu.check()
    finally:
        gpu.set_options()
    bad = [k for k in e["planes"] if oracle.fnv(_host(planes[k])) != e["planes"][k]]
    bad += [k for k in PLANES if k.startswith("pCol") and oracle.fnv(_host(planes[k])) != base["planes"][k]]
    assert not bad, (name, bad)
    a, b, c = e["shift"]
    word = 0xFF000000 | PATTERN[a] << 16 | PATTERN[b] << 8 | PATTERN[c]
    assert bool((planes["pShiftABCX"] == (word - (1 << 32))).all())
    psnr, mse = gpu.compare_device(img, planes["pDecoded"], e["alpha"])
    assert psnr == pytest.approx(e["psnr"], abs=1e-9) and mse == pytest.approx(e["mse"], rel=1e-12)
    del planes, img
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", sorted(k for k, e in GOLD.items() if e["kind"] == "batch"))
def test_batch_reference_checksums(gpu, name):
    """BASELINE config 4 WHOLE as one GPU's context sees it: all 64 images (4096^2 random-gradient, seeds 1 .. 64) through ONE limg_hip_encode3d_batch_device call
    (the sub-batch pipeline: k_fit_tpb of sub-batch k + 1 beside the persistent kernel of sub-batch k), every plane of every image against the real reference's
    position-sensitive checksums, computed on the device; then the first eight one by one through limg_hip_encode3d_device (the rank of an 8-GPU job: 8 images)."""
    import torch
    from limg_amd.shard import sum64_device
    e = GOLD[name]
    W, H = e["w"], e["h"]
    imgs = [gpu.synth_device(KIND[e["gen"]], W, H, seed=im["seed"]) for im in e["images"]]
    for im, t in zip(e["images"], imgs):
        assert sum64_device(t) == im["input_sum64"], ("input", im["seed"])
    outs = [gpu.alloc_planes_device(W, H) for _ in imgs]
    gpu.encode3d_batch_device(imgs, e["alpha"], outs, **e["kw"])
    torch.cuda.synchronize()
    gpu.check()
    bad = [(im["seed"], k) for im, pl in zip(e["images"], outs) for k in PLANES if sum64_device(pl[k]) != im["sum64"][k]]
    assert not bad, bad[:8]
    for im, t, pl in zip(e["images"][:8], imgs, outs):
        psnr, mse = gpu.compare_device(t, pl["pDecoded"], e["alpha"])
        assert psnr == pytest.approx(im["psnr"], abs=1e-9) and mse == pytest.approx(im["mse"], rel=1e-12), im["seed"]
    one = gpu.alloc_planes_device(W, H)
    for im, t in list(zip(e["images"], imgs))[:8]:
        gpu.encode3d_device(t, e["alpha"], one, **e["kw"])
        torch.cuda.synchronize()
        bad = [k for k in PLANES if sum64_device(one[k]) != im["sum64"][k]]
        assert not bad, (im["seed"], bad)
    gpu.check()
    del imgs, outs, one
    torch.cuda.empty_cache()


def test_one_strip_base_error_is_caught(gpu, oracle):
    """Sensitivity of the pin: ONE work strip (of 32 768) that dithers from a chain position off by one call -- the smallest error a look-back can make
    (limg_hip_options.test_base_error_strip) -- changes the hashes of the chain-dependent planes, while everything a band-limited check looks at (the first 256
    rows on every plane, the chain-independent planes everywhere, the PSNR to 0.05 dB) stays as it was."""
    import torch
    e = GOLD["pn8192"]
    W = e["w"]
    img = gpu.synth_device("photo_noise", W, W, seed=1)
    planes = gpu.alloc_planes_device(W, W)
    strip = 20000  # block row 625, columns 0..255
    gpu.set_options(test_base_error_strip=strip + 1)
    try:
        gpu.encode3d_device(img, True, planes)
        torch.cuda.synchronize()
        gpu.check()
    finally:
        gpu.set_options()
    changed = [k for k in PLANES if oracle.fnv(_host(planes[k])) != e["planes"][k]]
    assert "pDecoded" in changed and any(k.startswith("pFactors") for k in changed), changed
    assert not [k for k in changed if k.startswith("pCol") or k == "pShiftABCX"], changed
    good = gpu.alloc_planes_device(W, W)
    gpu.encode3d_device(img, True, good)
    torch.cuda.synchronize()
    row, col = (strip // (W // 256)) * 8, (strip % (W // 256)) * 256
    for k in PLANES:
        diff = planes[k] != good[k]
        outside = diff.clone()
        outside[row:row + 8, col:col + 256] = False
        assert not bool(outside.any()), k  # only that strip's pixels differ
        assert torch.equal(planes[k][:256], good[k][:256]), k  # what the band-limited check sees: nothing
    psnr, _ = gpu.compare_device(img, planes["pDecoded"], True)
    assert abs(psnr - e["psnr"]) < 0.05
    del planes, good, img
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", sorted(k for k, e in GOLD.items() if e["kind"] == "blocked"))
def test_blocked_reference_hashes(gpu, oracle, name):
    """the merged-block encoder at its bench size: the 13 planes upstream writes and the rectangle count == the real reference (the asynchrony of its pipeline --
    first report at 1024 rectangles, batches worth thousands of rectangles, records waited for at the first out-of-window pair -- only shows at these sizes)"""
    import torch
    e = GOLD[name]
    img = gpu.synth_device(KIND[e["gen"]], e["w"], e["h"], seed=e["seed"])
    assert oracle.fnv(_host(img)) == e["input"]
    planes = gpu.alloc_blocked_planes_device(e["w"], e["h"])
    gpu.blocked_encode3d_device(img, e["alpha"], planes, **e["kw"])
    torch.cuda.synchronize()
    gpu.check()
    assert len(gpu.blocked_regions()) == e["regions"]
    bad = [k for k in BLOCKED_WRITTEN if oracle.fnv(_host(planes[k])) != e["planes"][k]]
    assert not bad, (name, bad)
    psnr, _ = gpu.compare_device(img, planes["pDecoded"], e["alpha"])
    assert psnr == pytest.approx(e["psnr"], abs=1e-9)
    del planes, img
    torch.cuda.empty_cache()

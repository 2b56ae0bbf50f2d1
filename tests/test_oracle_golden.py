"""The CPU oracle against the committed golden vectors (outputs of the real reference, tools/make_golden.py).
Runs everywhere (no GPU, no /root/reference): this is what pins the oracle on the GPU box."""
import numpy as np
import pytest

from oracle.bind import PLANES, DITHER_AES, DITHER_PCG, REC_DTYPE
import golden_util as gu


def test_cases_bit_exact(oracle):
    idx, z = gu.cases()
    for i, m in enumerate(idx):
        img = z["c%02d_in" % i]
        o = oracle.encode3d(img, m["alpha"], error_factor=m["ef"], pool_threads=m["pool"], fast=m["fast"], dither_mode=m["dither"])
        for k in PLANES:
            assert np.array_equal(o[k], z["c%02d_%s" % (i, k)]), (i, m, k)
        assert oracle.compare(img, o["pDecoded"], m["alpha"])[0] == pytest.approx(m["psnr"], abs=1e-9)


@pytest.mark.parametrize("name", ["original_rgb", "original_as_rgba", "rg1024", "rga1024", "pn1024", "pn1024_ef25", "pn1024_pool2", "pn1024_pcg", "original_rgb_ef0", "pn1024_accurate", "rg1024_accurate"])
def test_full_image_hashes(oracle, name):
    e = gu.hashes()[name]
    img = gu.big_input(name, oracle)
    assert oracle.fnv(img) == e["input"]
    o = oracle.encode3d(img, e["alpha"], **e["kw"])
    for k in PLANES:
        assert oracle.fnv(o[k]) == e[k], k
    psnr, mse = oracle.compare(img, o["pDecoded"], e["alpha"])
    assert psnr == pytest.approx(e["psnr"], abs=1e-9) and mse == pytest.approx(e["mse"], rel=1e-12)


@pytest.mark.parametrize("name", ["rg4096", "rg4096_pool2", "pn16384x2048_pool2", "rg4096_accurate", "rg4096_rgb_accurate"])
def test_fullsize_hashes(oracle, name):
    """BASELINE sizes: the oracle against the real reference's plane hashes of tests/golden/fullsize.json (tools/make_golden_fullsize.py) -- config 2's 4096^2
    gradient image with one chain and with eight, and strip 0 of config 5.  (The 8192^2 entries take the scalar oracle a minute each; the GPU tests hash those.)"""
    import json
    import os
    e = json.load(open(os.path.join(gu.G, "fullsize.json")))[name]
    img = oracle.photo_noise(e["w"], e["h"], e["seed"]) if e["gen"] == "pn" else oracle.random_gradient(e["w"], e["h"], e["seed"], True)
    assert oracle.fnv(img) == e["input"]
    o = oracle.encode3d(img, e["alpha"], worker_threads=8, **e["kw"])
    for k in PLANES:
        assert oracle.fnv(o[k]) == e["planes"][k], k
    psnr, _ = oracle.compare(img, o["pDecoded"], e["alpha"])
    assert psnr == pytest.approx(e["psnr"], abs=1e-9)


def test_config1_psnr_baseline(oracle):
    """BASELINE.json configs[0]: assets/original.png, RGB, single thread: perceptual PSNR 40.6994 dB."""
    img = gu.load_png()
    o = oracle.encode3d(img, False)
    psnr, mse = oracle.compare(img, o["pDecoded"], False)
    assert abs(psnr - 40.6994) < 1e-4 and abs(mse - 49.8179) < 1e-4


def test_block_probes(oracle):
    z = gu.blocks()
    for bi in range(int(z["count"])):
        for ch in (4, 3):
            p = "b%02d_%d_" % (bi, ch)
            px = z[p + "px"]
            rec = oracle.block_fit(px, ch)
            g = z[p + "rec"]
            for f in REC_DTYPE.names:
                if f != "avg":
                    assert np.array_equal(rec[f], g[f]), (p, f)
            a, b, c = oracle.block_factors(px, ch, rec)
            assert np.array_equal(a, z[p + "A"]) and np.array_equal(b, z[p + "B"]) and np.array_equal(c, z[p + "C"]), p
            t = z[p + "trials"]
            for sa in range(9):
                for sb in range(9):
                    for sc in range(9):
                        ok, be = oracle.block_trial(px, ch, rec, a, b, c, (sa, sb, sc), 100)
                        assert int(ok) == t[sa, sb, sc, 0]
                        if ok:
                            assert be == t[sa, sb, sc, 1]
            for ef in (25, 100, 400):
                for fast in (1, 0):
                    assert np.array_equal(oracle.block_search(px, ch, rec, a, b, c, ef, bool(fast))[0], z[p + "search_%d_%d" % (ef, fast)]), (p, ef, fast)
            shp = z[p + "shape"]
            assert np.array_equal(oracle.block_decode(int(shp[1]), int(shp[0]), ch, rec, a, b, c, (3, 8, 0)), z[p + "decode_380"])


def test_known_answer_blocks(oracle):
    """SURVEY Appendix C blocks (flat / line / plane)."""
    z = gu.blocks()
    r = z["b00_4_rec"]  # flat
    assert r["dirA_min"][0].tolist() == [0x20, 0x40, 0x80, 0xFF] and r["dirB_mag"][0].tolist() == [0, 0, 0, 0]
    r = z["b01_4_rec"]  # line
    assert r["dirA_min"][0].tolist() == [10, 200, 50, 255] and r["dirA_max"][0].tolist() == [150, 130, 85, 255]
    assert z["b01_4_A"][:8].tolist() == [0, 36, 73, 109, 146, 182, 219, 255]
    r = z["b02_4_rec"]  # plane
    assert r["dirA_min"][0].tolist() == [23, 22, 52, 255] and r["dirB_offset"][0].tolist() == [-88, 56, -15, 0]


def test_chain_known_answers(oracle):
    c = gu.chain()
    assert c["aes_64"][:4] == ["4ae914d5e23b0473", "1db0e1e7cd750f32", "13d534ac987485a9", "d82d4ba61c55878b"]
    assert c["aes_16"][0] == "418d683743d058cc" and c["aes_20"][0] == "15f4e09645dfa398"
    assert c["pcg_64"][:2] == ["e56f5ebfe9622fcd", "a7fa430d8e002f8d"]
    for mode, mname in ((DITHER_AES, "aes"), (DITHER_PCG, "pcg")):
        for n in (64, 16, 20, 40, 7, 15):
            h = 0xCA7F00D15BADF00D
            for want in c["%s_%d" % (mname, n)]:
                h = oracle.chain_step(n, h, mode)
                assert "%016x" % h == want
        f = (np.arange(64) * 4 + 1).astype(np.uint8)
        for s in range(1, 8):
            assert oracle.dither(s, 0xCA7F00D15BADF00D, f, mode)[1].tolist() == c["dither_bytes"]["%s_s%d" % (mname, s)]


def test_shift_multipliers(oracle):
    """re-expansion multipliers for shift 0..8 = 1,2,4,8,17,36,85,255,256 (observable through decode)."""
    rec = np.zeros(1, dtype=REC_DTYPE)
    rec["dirA_max"][0] = [256, 0, 0, 0]
    one = np.ones(1, dtype=np.uint8)
    zero = np.zeros(1, dtype=np.uint8)
    # est = (1*mul*256 + 128) >> 8 = mul (clamped to 255); shift 8 zeroes the RGB normals
    got = [int(oracle.block_decode(1, 1, 4, rec, one, zero, zero, (s, 0, 0))[0, 0] & 0xFF) for s in range(9)]
    assert got == [1, 2, 4, 8, 17, 36, 85, 255, 0]


def test_tree_mode_tolerance(oracle):
    """FLOAT_TREE differs from FLOAT_X86 only in the order of the three direction sums: same shifts almost everywhere,
    perceptual PSNR within 0.10 dB (the reference's own AES<->PCG and 1<->8-thread spreads are 0.02-0.03 dB)."""
    from oracle.bind import FLOAT_TREE
    for img, alpha in ((oracle.photo_noise(256, 256, 1), True), (oracle.random_gradient(256, 256, 1, True), True), (gu.load_png()[:256, :256].copy(), False)):
        a = oracle.encode3d(img, alpha)
        b = oracle.encode3d(img, alpha, float_mode=FLOAT_TREE)
        pa, pb = oracle.compare(img, a["pDecoded"], alpha)[0], oracle.compare(img, b["pDecoded"], alpha)[0]
        assert abs(pa - pb) < 0.10
        assert (a["pShiftABCX"] != b["pShiftABCX"]).mean() < 0.02

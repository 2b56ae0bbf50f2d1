"""The CPU restatement of the merged-block encoder (oracle/limg_oracle_blocked.c) against the real reference:
committed plane hashes (tests/golden/blocked.json, made by tools/make_golden_blocked.py) and, where oracle/_ref exists, direct runs."""
import json
import os

import numpy as np
import pytest

import golden_util as gu
from oracle.bind import BLOCKED_WRITTEN

GOLD = json.load(open(os.path.join(gu.G, "blocked.json")))


def make_input(orc, e):
    if e["gen"] == "png":
        return gu.load_png()
    if e["gen"] == "pn":
        return orc.photo_noise(e["w"], e["h"], e["seed"])
    return orc.random_gradient(e["w"], e["h"], e["seed"], e["gen"] == "rg")


@pytest.mark.parametrize("name", sorted(GOLD))
def test_golden_plane_hashes(oracle, name):
    e = GOLD[name]
    img = make_input(oracle, e)
    assert oracle.fnv(img) == e["input"]
    got = oracle.blocked_encode3d(img, e["alpha"], **e["kw"])
    for k in BLOCKED_WRITTEN:
        assert oracle.fnv(got[k]) == e["planes"][k], (name, k)
    assert len(got["regions"]) == e["regions"]
    assert not got["pBlockError"].any()  # never written upstream
    psnr, mse = oracle.compare(img, got["pDecoded"], e["alpha"])
    assert psnr == pytest.approx(e["psnr"], abs=1e-9)
    # regions tile the block grid exactly once, in block-index order
    by, bx = got["pass1"].shape
    cover = np.zeros((by, bx), dtype=np.int32)
    for i, r in enumerate(got["regions"]):
        cover[r["oy"]:r["oy"] + r["ry"], r["ox"]:r["ox"] + r["rx"]] += 1
        assert int(got["pBlockIndex"][r["oy"] * 8, r["ox"] * 8]) == (0xFF000000 | (i + 1))
    assert (cover == 1).all()


@pytest.mark.ref
@pytest.mark.parametrize("shape,alpha,gen,kw", [
    ((64, 64), True, "rg", {}), ((72, 40), True, "pn", {}), ((37, 29), True, "rga", {}), ((131, 77), False, "pn", {}), ((8, 8), True, "pn", {}),
    ((24, 200), False, "rg", {}), ((320, 96), True, "pn", {"error_factor": 50}), ((320, 96), True, "rg", {"fast": False}), ((96, 96), True, "flat", {}),
    ((128, 128), True, "pn", {"pool_threads": 2}),
    # corner blocks of fewer than 4 pixels: pass 1 sums stale gather-buffer entries upstream (see limg_oracle_block_fit_gathered)
    ((9, 9), True, "pn", {}), ((17, 10), False, "pn", {}), ((2, 65), True, "pn", {}), ((25, 33), True, "rg", {}), ((11, 9), False, "rg", {}),
])
def test_against_reference(oracle, ref, shape, alpha, gen, kw):
    w, h = shape
    if gen == "flat":
        img = np.full((h, w), 0xFF336699, dtype=np.uint32)
    elif gen == "pn":
        img = oracle.photo_noise(w, h, 11)
    else:
        img = oracle.random_gradient(w, h, 11, gen == "rg")
    want = ref.blocked_encode3d(img, alpha, **kw)
    okw = dict(kw); okw.pop("pool_threads", None)  # the pool only splits pass 1 (no state): results must not depend on it
    got = oracle.blocked_encode3d(img, alpha, **okw)
    bad = [(k, int((got[k] != want[k]).sum())) for k in BLOCKED_WRITTEN if not np.array_equal(got[k], want[k])]
    assert not bad, bad


@pytest.mark.ref
def test_match_predicate_against_reference(oracle, ref):
    """`limg_encode_3d_matches` on pairs of real pass-1 fits and on perturbed ones (both outcomes must occur)."""
    rng = np.random.default_rng(5)
    for channels, img in ((4, oracle.photo_noise(256, 128, 3)), (4, oracle.random_gradient(256, 128, 3, False)), (3, oracle.photo_noise(256, 128, 4))):
        recs = oracle.blocked_encode3d(img, channels == 4, planes=False)["pass1"].reshape(-1)
        seen = set()
        for _ in range(1500):
            i, j = rng.integers(0, recs.size, 2)
            if rng.random() < 0.7:
                j = min(recs.size - 1, i + int(rng.integers(1, 3)))  # neighbours: the pairs the encoder really asks about
            a, b = recs[i:i + 1].copy(), recs[j:j + 1].copy()
            if rng.random() < 0.3:
                b["dirA_max"][0][:channels] += rng.integers(-3, 4, channels).astype(np.int16)
                b["avg"][0][:channels] += rng.normal(0, 2, channels).astype(np.float32)
            want = ref.blocked_matches(channels, a, b)
            assert oracle.blocked_matches(channels, a, b) == want, (channels, i, j)
            seen.add(want)
        assert seen == {True, False}

"""The reference's own statistics output -- "Average Block Bits" and the per-factor shift histogram that limg_encode3d_test / limg_blocked_encode3d_test print
(src/limg.cpp:2232-2248; counters :1971-1999 and :1561-1590) -- against text captured from the real reference (tests/golden/stats.json, tools/make_golden_stats.py).
CPU: the counting rule + formatting on the oracle's shifts.  GPU: limg_hip_last_stats (counters reduced on the device from the per-block shift words)."""
import json
import os

import numpy as np
import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "stats.json")))


def _input(e, oracle):
    if e["gen"] == "png":
        img = gu.load_png()
    elif e["gen"] == "pn":
        img = oracle.photo_noise(e["w"], e["h"], e["seed"])
    else:
        img = oracle.random_gradient(e["w"], e["h"], e["seed"], True)
    assert oracle.fnv(img) == e["input"]
    return img


def _counters_from_shifts(shifts, w, h):
    by, bx, _ = shifts.shape
    n = np.minimum(8, w - 8 * np.arange(bx))[None, :] * np.minimum(8, h - 8 * np.arange(by))[:, None]
    c = np.zeros(30, dtype=np.uint64)
    for f in range(3):
        s = shifts[:, :, f].astype(np.int64)
        c[f] = int(((8 - s) * n).sum())
        for v in range(9):
            c[3 + 9 * f + v] = int(n[s == v].sum())
    return c


@pytest.mark.parametrize("name", sorted(GOLD))
def test_stats_text_from_oracle_shifts(oracle, name):
    import limg_amd
    e = GOLD[name]
    img = _input(e, oracle)
    want = oracle.encode3d(img, e["alpha"], extras=True, **e["kw"])
    c = _counters_from_shifts(want["shifts"], e["w"], e["h"])
    assert limg_amd.format_stats(c, e["w"] * e["h"]) == e["fixed_blocks_stdout"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_gpu_stats_equal_the_reference_text(oracle, name):
    import limg_amd
    e = GOLD[name]
    img = _input(e, oracle)
    g = limg_amd.LimgHip(0)
    try:
        with pytest.raises(limg_amd.LimgHipError):
            g.last_stats()  # nothing collected yet
        g.set_options(collect_stats=True)
        g.encode3d(img, e["alpha"], **e["kw"])
        c, px = g.last_stats()
        assert px == e["w"] * e["h"]
        assert limg_amd.format_stats(c, px) == e["fixed_blocks_stdout"]
        kw = {k: v for k, v in e["kw"].items() if k != "pool_threads"}
        g.blocked_encode3d(img, e["alpha"], **kw)
        c, px = g.last_stats()
        text = limg_amd.format_stats(c, px)
        assert e["merged_blocks_stdout"].startswith(text), (text, e["merged_blocks_stdout"])
        g.set_options()
        g.encode3d(img, e["alpha"])
        with pytest.raises(limg_amd.LimgHipError):
            g.last_stats()  # switched off again
    finally:
        g.check()
        g.close()


@pytest.mark.gpu
def test_gpu_stats_of_a_batch(oracle):
    """A batched encode's counters are those of its images together."""
    import torch
    import limg_amd
    W, H = 256, 64
    host = [oracle.photo_noise(W, H, 50 + i) for i in range(3)]
    total = np.zeros(30, dtype=np.uint64)
    for h in host:
        total += _counters_from_shifts(oracle.encode3d(h, True, extras=True)["shifts"], W, H)
    g = limg_amd.LimgHip(0)
    try:
        g.set_options(collect_stats=True)
        imgs = [torch.from_numpy(h.view(np.int32)).cuda() for h in host]
        outs = [g.alloc_planes_device(W, H) for _ in imgs]
        g.encode3d_batch_device(imgs, True, outs)
        c, px = g.last_stats()
        assert px == 3 * W * H and np.array_equal(c, total)
    finally:
        g.check()
        g.close()

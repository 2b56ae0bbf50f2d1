#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref/liblimg_ref.so, strict-IEEE build of /root/reference/src,
see oracle/build_ref.sh).  Runs only in the container that has /root/reference; the fixtures are data (inputs and
the reference's outputs), committed so that machines without the reference (the GPU box) are pinned by the same answers.

  tests/golden/cases.npz        small images: input + all 11 planes for a matrix of (generator, size, channels, errorFactor, mode)
  tests/golden/hashes.json      FNV-1a-64 of every plane for original.png and for 1024x1024 of each synthetic generator
  tests/golden/blocks.npz       per-block probes: records, factor bytes, trial tables (shift -> pass, blockError), searches, decode
  tests/golden/chain.json       dither state-walk known answers (AES and PCG)
  tests/golden/original.png     the reference's own sample image (assets/original.png, data file) = config #1 input
"""
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.bind import Oracle, Ref, PLANES, DITHER_AES, DITHER_PCG  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)
orc = Oracle()  # used ONLY for the integer-defined synthetic generators and the FNV hash, never for expected outputs
ref = Ref()


def load_png(path):
    from PIL import Image
    a = np.array(Image.open(path).convert("RGBA"))
    return np.ascontiguousarray(a).view(np.uint32).reshape(a.shape[0], a.shape[1])


def gen(name, w, h, seed):
    if name == "rg":
        return orc.random_gradient(w, h, seed, True)
    if name == "rga":
        return orc.random_gradient(w, h, seed, False)
    return orc.photo_noise(w, h, seed)


# ---- cases.npz -----------------------------------------------------------------------------------------------------
cases = {}
index = []
matrix = []
for name in ("rg", "rga", "pn"):
    for alpha in (True, False):
        for ef in (0, 25, 100, 400):
            matrix.append(dict(gen=name, w=64, h=48, seed=1, alpha=alpha, ef=ef, pool=0, fast=True, dither=DITHER_AES))
matrix += [
    dict(gen="pn", w=61, h=27, seed=2, alpha=True, ef=100, pool=0, fast=True, dither=DITHER_AES),   # ragged right + bottom edge
    dict(gen="pn", w=61, h=27, seed=2, alpha=False, ef=100, pool=0, fast=True, dither=DITHER_AES),
    dict(gen="rg", w=128, h=20, seed=3, alpha=True, ef=100, pool=0, fast=True, dither=DITHER_AES),  # ragged bottom only (config #1 shape class)
    dict(gen="pn", w=5, h=3, seed=4, alpha=True, ef=100, pool=0, fast=True, dither=DITHER_AES),     # single partial block, < 8 px wide (PCG tail only)
    dict(gen="pn", w=8, h=8, seed=5, alpha=True, ef=100, pool=0, fast=True, dither=DITHER_AES),     # single block
    dict(gen="pn", w=64, h=200, seed=6, alpha=True, ef=100, pool=2, fast=True, dither=DITHER_AES),  # 8 strips (== 8-GPU strip-restart semantics)
    dict(gen="pn", w=64, h=200, seed=6, alpha=True, ef=100, pool=1, fast=True, dither=DITHER_AES),  # 4 strips
    dict(gen="pn", w=64, h=48, seed=7, alpha=True, ef=100, pool=0, fast=True, dither=DITHER_PCG),   # non-AES hosts
    dict(gen="pn", w=64, h=48, seed=8, alpha=True, ef=100, pool=0, fast=False, dither=DITHER_AES),  # --accurate-bit-crushing
    dict(gen="pn", w=64, h=48, seed=8, alpha=False, ef=100, pool=0, fast=False, dither=DITHER_AES),
]
for i, m in enumerate(matrix):
    img = gen(m["gen"], m["w"], m["h"], m["seed"])
    out = ref.encode3d(img, m["alpha"], error_factor=m["ef"], pool_threads=m["pool"], fast=m["fast"], dither_mode=m["dither"])
    cases["c%02d_in" % i] = img
    for k in PLANES:
        cases["c%02d_%s" % (i, k)] = out[k]
    m = dict(m)
    m["psnr"] = ref.compare(img, out["pDecoded"], m["alpha"])[0]
    index.append(m)
np.savez_compressed(os.path.join(G, "cases.npz"), **cases)
json.dump(index, open(os.path.join(G, "cases.json"), "w"), indent=1)

# ---- hashes.json ---------------------------------------------------------------------------------------------------
shutil.copyfile("/root/reference/assets/original.png", os.path.join(G, "original.png"))
hashes = {}
big = {
    "original_rgb": (load_png(os.path.join(G, "original.png")), False, {}),
    "original_as_rgba": (load_png(os.path.join(G, "original.png")), True, {}),
    "rg1024": (orc.random_gradient(1024, 1024, 1, True), True, {}),
    "rga1024": (orc.random_gradient(1024, 1024, 1, False), True, {}),
    "pn1024": (orc.photo_noise(1024, 1024, 1), True, {}),
    "pn1024_ef25": (orc.photo_noise(1024, 1024, 1), True, dict(error_factor=25)),
    "pn1024_pool2": (orc.photo_noise(1024, 1024, 1), True, dict(pool_threads=2)),
    "pn1024_pcg": (orc.photo_noise(1024, 1024, 1), True, dict(dither_mode=DITHER_PCG)),
    "original_rgb_ef0": (load_png(os.path.join(G, "original.png")), False, dict(error_factor=0)),
}
for name, (img, alpha, kw) in big.items():
    out = ref.encode3d(img, alpha, **kw)
    e = {k: orc.fnv(out[k]) for k in PLANES}
    e["input"] = orc.fnv(img)
    e["psnr"], e["mse"] = ref.compare(img, out["pDecoded"], alpha)
    e["shape"] = list(img.shape)
    e["alpha"] = alpha
    e["kw"] = kw
    hashes[name] = e
json.dump(hashes, open(os.path.join(G, "hashes.json"), "w"), indent=1)

# ---- blocks.npz ----------------------------------------------------------------------------------------------------
rng = np.random.default_rng(12345)
blocks = {}
bi = 0
srcs = [orc.photo_noise(64, 64, 21), orc.random_gradient(64, 64, 22, True), orc.random_gradient(64, 64, 23, False)]
flat = np.full((8, 8), 0xFF804020, dtype=np.uint32)
line = np.array([[(10 + 20 * x) | ((200 - 10 * x) << 8) | ((50 + 5 * x) << 16) | (255 << 24) for x in range(8)] for y in range(8)], dtype=np.uint32)
plane = np.array([[(10 + 20 * x) | ((30 + 25 * y) << 8) | ((50 + 5 * x + 3 * y) << 16) | (255 << 24) for x in range(8)] for y in range(8)], dtype=np.uint32)
blist = [flat, line, plane] + [np.ascontiguousarray(s[y:y + 8, x:x + 8]) for s in srcs for (y, x) in ((0, 0), (24, 40), (56, 8))]
blist += [np.ascontiguousarray(srcs[0][0:3, 0:5]), np.ascontiguousarray(srcs[0][8:16, 0:2])]  # partial blocks (n = 15, 16)
for blk in blist:
    px = blk.ravel()
    for ch in (4, 3):
        rec = ref.block_fit(px, ch)
        a, b, c = ref.block_factors(px, ch, rec)
        trials = np.zeros((9, 9, 9, 2), dtype=np.int64)
        for sa in range(9):
            for sb in range(9):
                for sc in range(9):
                    ok, be = ref.block_trial(px, ch, rec, a, b, c, (sa, sb, sc), 100)
                    trials[sa, sb, sc] = (int(ok), be if ok else -1)
        p = "b%02d_%d_" % (bi, ch)
        blocks[p + "px"] = px
        blocks[p + "shape"] = np.array(blk.shape)
        blocks[p + "rec"] = rec
        blocks[p + "A"], blocks[p + "B"], blocks[p + "C"] = a, b, c
        blocks[p + "trials"] = trials
        for ef in (25, 100, 400):
            for fast in (1, 0):
                blocks[p + "search_%d_%d" % (ef, fast)] = ref.block_search(px, ch, rec, a.copy(), b.copy(), c.copy(), ef, bool(fast))
        sh = (3, 8, 0)
        blocks[p + "decode_380"] = ref.block_decode(blk.shape[1], blk.shape[0], ch, rec, a, b, c, sh)
    bi += 1
blocks["count"] = np.array(bi)
np.savez_compressed(os.path.join(G, "blocks.npz"), **blocks)

# ---- chain.json ----------------------------------------------------------------------------------------------------
chain = {}
for mode, mname in ((DITHER_AES, "aes"), (DITHER_PCG, "pcg")):
    for n in (64, 16, 20, 40, 7, 15):
        h = 0xCA7F00D15BADF00D
        seq = []
        f = np.zeros(n, dtype=np.uint8)
        for _ in range(6):
            h, _f = ref.dither(3, h, f, mode)
            seq.append("%016x" % h)
        chain["%s_%d" % (mname, n)] = seq
f = (np.arange(64) * 4 + 1).astype(np.uint8)
chain["dither_bytes"] = {"%s_s%d" % (mname, s): ref.dither(s, 0xCA7F00D15BADF00D, f, mode)[1].tolist()
                         for mode, mname in ((DITHER_AES, "aes"), (DITHER_PCG, "pcg")) for s in range(1, 8)}
json.dump(chain, open(os.path.join(G, "chain.json"), "w"), indent=1)
print("golden fixtures written to", G)

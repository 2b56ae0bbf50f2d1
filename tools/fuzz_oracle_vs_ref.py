#!/usr/bin/env python3
"""Randomised oracle-vs-REAL-reference sweep (only where oracle/_ref exists, i.e. in the container that has /root/reference): random shapes, channel
counts, generators (incl. random bytes), error factors, pools, accurate mode, PCG -- limg_encode3d_test (11 planes) and limg_blocked_encode3d_test
(13 planes).  usage: python tools/fuzz_oracle_vs_ref.py [--seconds 120] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from oracle.bind import Oracle, Ref, PLANES, BLOCKED_WRITTEN
    orc, ref = Oracle(), Ref()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    n = {"fixed": 0, "blocked": 0}
    while time.time() - t0 < args.seconds:
        w = int(rng.integers(1, 24)) * 8 + (int(rng.integers(0, 8)) if rng.random() < 0.5 else 0)
        h = int(rng.integers(1, 16)) * 8 + (int(rng.integers(0, 8)) if rng.random() < 0.5 else 0)
        gen = ["pn", "rg", "rga", "rand", "flat"][int(rng.integers(0, 5))]
        seed = int(rng.integers(1, 1 << 30))
        alpha = bool(rng.random() < 0.6)
        if gen == "pn":
            img = orc.photo_noise(w, h, seed)
        elif gen in ("rg", "rga"):
            img = orc.random_gradient(w, h, seed, gen == "rg")
        elif gen == "rand":
            img = rng.integers(0, 1 << 32, (h, w), dtype=np.uint64).astype(np.uint32)
        else:
            img = np.full((h, w), int(rng.integers(0, 1 << 32)), dtype=np.uint32)
            img[rng.integers(0, h), rng.integers(0, w)] ^= 0x00FFFFFF
        ef = int([0, 10, 25, 50, 100, 100, 200, 400, 1000][int(rng.integers(0, 9))])
        fast = bool(rng.random() < 0.8)
        pcg = int(rng.random() < 0.2)
        pool = int([0, 0, 1, 2, 3][int(rng.integers(0, 5))])
        recipe = dict(w=w, h=h, gen=gen, seed=seed, alpha=alpha, ef=ef, fast=fast, pcg=pcg, pool=pool)
        if rng.random() < 0.5:
            a = orc.encode3d(img, alpha, error_factor=ef, fast=fast, dither_mode=pcg, pool_threads=pool)
            b = ref.encode3d(img, alpha, error_factor=ef, fast=fast, dither_mode=pcg, pool_threads=pool)
            bad = [k for k in PLANES if not np.array_equal(a[k], b[k])]
            n["fixed"] += 1
        else:
            a = orc.blocked_encode3d(img, alpha, error_factor=ef, fast=fast, dither_mode=pcg)
            b = ref.blocked_encode3d(img, alpha, error_factor=ef, fast=fast, dither_mode=pcg)
            bad = [k for k in BLOCKED_WRITTEN if not np.array_equal(a[k], b[k])]
            n["blocked"] += 1
        if bad:
            print("MISMATCH", bad, recipe, flush=True)
            sys.exit(1)
    print("oracle == reference:", n, "cases in %.0fs" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()

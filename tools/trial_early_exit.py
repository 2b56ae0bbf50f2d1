#!/usr/bin/env python3
"""Would a trial that stops at its first failing group of pixels pay?  For every trial of the fast search on sample blocks: how it ends (passes / a pixel over the
limit / the block sum over the limit) and, for the pixel failures, in which quarter (16 pixels = two rows, rows 2q .. 2q+1) or eighth (one row) the first failing pixel
lies when the groups are evaluated in order -- i.e. how many groups a lanes-per-group mapping would evaluate before it can abandon the trial.
CPU only; test infrastructure (uses oracle/)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_search_table import search_fast, search_accurate  # noqa: E402
from oracle.bind import Oracle, _ptr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=1500)
    ap.add_argument("--error-factor", type=int, default=100)
    ap.add_argument("--workload", default="photo_noise")
    ap.add_argument("--accurate", action="store_true")
    args = ap.parse_args()
    orc = Oracle()
    L = orc.lib
    L.limg_oracle_block_trial_pixel_errors.restype = None
    L.limg_oracle_block_trial_pixel_errors.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 6
    W = H = 1024
    img = orc.photo_noise(W, H, 1) if args.workload == "photo_noise" else orc.random_gradient(W, H, 1, True)
    rng = np.random.default_rng(7)
    ef = args.error_factor
    max_pixel = 6 * (ef // 2) * 7
    max_block = 4 * (ef // 2) * 7
    ends = {"pass": 0, "pixel": 0, "sum": 0}
    groups = {4: 0, 8: 0, 16: 0}     # groups evaluated over all trials, by number of groups per block
    first = {4: np.zeros(4, int), 8: np.zeros(8, int), 16: np.zeros(16, int)}
    groups_sum = {4: 0, 8: 0, 16: 0}
    first_any = {4: np.zeros(4, int), 8: np.zeros(8, int), 16: np.zeros(16, int)}
    failing_pixels = []
    trials = 0
    for _ in range(args.blocks):
        bx, by = int(rng.integers(0, W // 8)), int(rng.integers(0, H // 8))
        px = np.ascontiguousarray(img[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8]).ravel()
        rec = orc.block_fit(px, 4)
        a, b, c = orc.block_factors(px, 4, rec)
        err = np.zeros(64, dtype=np.uint32)
        gen = search_accurate() if args.accurate else search_fast()
        try:
            t = next(gen)
            while True:
                sh = np.array(t[:3], dtype=np.uint8)
                L.limg_oracle_block_trial_pixel_errors(_ptr(px), 64, 4, _ptr(rec), _ptr(a), _ptr(b), _ptr(c), _ptr(sh), _ptr(err))
                e = err.astype(np.int32).astype(np.int64)
                bad = e > max_pixel
                trials += 1
                if bad.any():
                    ends["pixel"] += 1
                    failing_pixels.append(int(bad.sum()))
                    for g in (4, 8, 16):
                        per = bad.reshape(g, 64 // g).any(axis=1)
                        k = int(np.argmax(per))
                        first[g][k] += 1
                        groups[g] += k + 1
                    ok = False
                else:
                    ok = int(e.sum()) * 16 < max_block * 64
                    ends["pass" if ok else "sum"] += 1
                    for g in (4, 8, 16):
                        groups[g] += g
                # the same with the block sum's limit applied to the partial sums as well (errors are non-negative: once over, always over)
                for g in (4, 8, 16):
                    per_bad = bad.reshape(g, 64 // g).any(axis=1)
                    part = np.cumsum(e.reshape(g, 64 // g).sum(axis=1)) * 16 >= max_block * 64
                    stop = per_bad | part
                    k = int(np.argmax(stop)) + 1 if stop.any() else g
                    groups_sum[g] += k
                    if not ok:
                        first_any[g][k - 1] += 1
                t = gen.send(ok)
        except StopIteration:
            pass
    n = args.blocks
    print("%s, errorFactor %d, %s search: %.2f trials per block: %.2f pass, %.2f fail on a pixel, %.2f fail on the block sum"
          % (args.workload, ef, "accurate" if args.accurate else "fast", trials / n, ends["pass"] / n, ends["pixel"] / n, ends["sum"] / n))
    fp = np.array(failing_pixels)
    print("pixel failures: %.1f failing pixels on average (median %d; %.1f %% of them have one or two)" % (fp.mean(), np.median(fp), 100.0 * (fp <= 2).mean()))
    for g in (4, 8, 16):
        print("groups of %2d pixels, evaluated in order, stop at the first group with a failing pixel: %.2f group evaluations per block against %.2f (= %.3f of the pixel work); "
              "first failing group: %s" % (64 // g, groups[g] / n, trials * g / n, groups[g] / (trials * g), np.round(first[g] / max(ends["pixel"], 1), 3).tolist()))


    for g in (4, 8, 16):
        print("... and at the first partial block sum over the limit: %.2f group evaluations per block (= %.3f of the pixel work); failing trials end in group %s"
              % (groups_sum[g] / n, groups_sum[g] / (trials * g), np.round(first_any[g] / max(ends["pixel"] + ends["sum"], 1), 3).tolist()))


if __name__ == "__main__":
    main()

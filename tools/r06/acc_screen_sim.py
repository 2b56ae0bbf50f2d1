#!/usr/bin/env python3
"""Accurate search (src/limg_bit_crush.h:668-830), priced before built: SPECULATIVE SCREENING instead of the 16-lane early-exit mapping VERDICT r05 item 2 describes.
The automaton's next trials ARE known if the current ones fail (the fail chain of the DAG), and 78 % of the accurate search's trials fail.  So one wave step evaluates
D triples of the fail chain at once, each on 64 / D pixels of the block (lane = (triple, pixel)): a triple with a pixel over the limit, or whose partial block sum is over
the limit, HAS failed (errors are non-negative) -- exact.  The first triple that survives gets today's full 64-lane trial; everything before it is resolved.  Nothing
diverges: one block per wave, one automaton state per wave.  This script replays the real automaton on sample blocks and counts screening rounds and full trials.
CPU only; test infrastructure (uses oracle/)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_search_table import build_accurate  # noqa: E402
from oracle.bind import Oracle, _ptr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=300)
    ap.add_argument("--error-factor", type=int, default=100)
    ap.add_argument("--workload", default="photo_noise")
    ap.add_argument("--cs", type=float, default=22.0, help="vector instructions of one screening round")
    ap.add_argument("--cf", type=float, default=31.0, help="... of one full trial after a screening round")
    args = ap.parse_args()
    orc = Oracle()
    L = orc.lib
    L.limg_oracle_block_trial_pixel_errors.restype = None
    L.limg_oracle_block_trial_pixel_errors.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 6
    trans = build_accurate()
    W = H = 1024
    img = orc.photo_noise(W, H, 1) if args.workload == "photo_noise" else orc.random_gradient(W, H, 1, True)
    rng = np.random.default_rng(7)
    ef = args.error_factor
    max_pixel = 6 * (ef // 2) * 7
    max_block = 4 * (ef // 2) * 7
    schemes = [(D, sel) for D in (2, 4, 8) for sel in ("first", "worst", "top")]
    tot = {s: [0, 0] for s in schemes}
    base = 0
    for _ in range(args.blocks):
        bx, by = int(rng.integers(0, W // 8)), int(rng.integers(0, H // 8))
        px = np.ascontiguousarray(img[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8]).ravel()
        rec = orc.block_fit(px, 4)
        fa, fb, fc = orc.block_factors(px, 4, rec)
        cache = {}

        def errs(t):
            if t not in cache:
                err = np.zeros(64, dtype=np.uint32)
                sh = np.array(t, dtype=np.uint8)
                L.limg_oracle_block_trial_pixel_errors(_ptr(px), 64, 4, _ptr(rec), _ptr(fa), _ptr(fb), _ptr(fc), _ptr(sh), _ptr(err))
                cache[t] = err.astype(np.int64)
            return cache[t]

        def full(t):
            e = errs(t)
            return (not (e > max_pixel).any()) and int(e.sum()) * 16 < max_block * 64

        s, n = 0, 0
        while trans[s][0] != "final":
            (a, b, c, _), p, f = trans[s]
            s = p if full((a, b, c)) else f
            n += 1
        base += n
        root = errs((4, 5, 6))
        for (D, sel) in schemes:
            g = 64 // D
            if sel == "first":
                idx = np.arange(g)
            elif sel == "top":  # the g pixels with the largest error in the root trial
                idx = np.argsort(-root, kind="stable")[:g]
            else:  # the aligned group of g pixels that holds the root trial's worst pixel
                k = int(np.argmax(root)) // g
                idx = np.arange(k * g, (k + 1) * g)
            s = 0
            rounds = fulls = 0
            while trans[s][0] != "final":
                rounds += 1
                chain = []
                q = s
                while len(chain) < D and trans[q][0] != "final":
                    chain.append(q)
                    q = trans[q][2]
                nxt = None
                for q in chain:
                    (a, b, c, _), p, f = trans[q]
                    e = errs((a, b, c))[idx]
                    if (e > max_pixel).any() or int(e.sum()) * 16 >= max_block * 64:
                        continue  # certainly failed
                    fulls += 1
                    nxt = p if full((a, b, c)) else f
                    break
                s = nxt if nxt is not None else trans[chain[-1]][2]
            tot[(D, sel)][0] += rounds
            tot[(D, sel)][1] += fulls
    nb = args.blocks
    print("%s, errorFactor %d, accurate search: %.2f trials per block today (28.8 vector instructions each = %.0f per block)" % (args.workload, ef, base / nb, 28.8 * base / nb))
    for (D, sel), (r, f) in tot.items():
        cost = (args.cs * r + args.cf * f) / nb
        print("D = %d triples x %2d pixels (%s group): %.2f screening rounds + %.2f full trials per block -> %.0f vector instructions at %.0f / %.0f = %.2f x"
              % (D, 64 // D, sel, r / nb, f / nb, cost, args.cs, args.cf, 28.8 * base / nb / cost))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How many of a block's trials repeat a shift triple the search has already tried for that block?  

A trial's outcome is a function of (block, triple), so a repeated triple needs no evaluation: in the automata (tools/make_search_table.py) it would be an edge taken
without a trial.  Runs the two literal restatements of the reference's control flow (search_fast: src/limg_bit_crush.h:331-392 + :502-614; search_accurate: :668-830)
on sample blocks with the oracle's trial (oracle/limg_oracle.c trial_core) as the outcome function and counts trials, repeats, and -- for comparison -- what a cache of
per-factor terms by shift would save.  CPU only; test infrastructure (uses oracle/)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_search_table import search_fast, search_accurate  # noqa: E402
from oracle.bind import Oracle  # noqa: E402


def run(gen, trial):
    seen = {}
    n = rep = 0
    seq = []
    try:
        t = next(gen)
        while True:
            key = tuple(t[:3])
            n += 1
            if key in seen:
                rep += 1
                ok = seen[key]
            else:
                ok = seen[key] = trial(key)
            seq.append((key, ok))
            t = gen.send(ok)
    except StopIteration:
        pass
    return n, rep, seq


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=1500)
    ap.add_argument("--error-factor", type=int, default=100)
    ap.add_argument("--workload", default="photo_noise")
    args = ap.parse_args()
    orc = Oracle()
    W = H = 1024
    img = orc.photo_noise(W, H, 1) if args.workload == "photo_noise" else orc.random_gradient(W, H, 1, True)
    rng = np.random.default_rng(7)
    tot = {"fast": [0, 0, 0], "accurate": [0, 0, 0]}
    where = {}
    for _ in range(args.blocks):
        bx, by = int(rng.integers(0, W // 8)), int(rng.integers(0, H // 8))
        px = np.ascontiguousarray(img[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8]).ravel()
        rec = orc.block_fit(px, 4)
        a, b, c = orc.block_factors(px, 4, rec)
        trial = lambda s: bool(orc.block_trial(px, 4, rec, a, b, c, s, args.error_factor)[0])  # noqa: E731
        for name, g in (("fast", search_fast()), ("accurate", search_accurate())):
            n, rep, seq = run(g, trial)
            tot[name][0] += n; tot[name][1] += rep; tot[name][2] += 1
            if name == "fast":
                seen = set()
                for i, (k, ok) in enumerate(seq):
                    if k in seen:
                        where[(i, k)] = where.get((i, k), 0) + 1
                    seen.add(k)
    for name, (n, rep, blocks) in tot.items():
        print("%-8s search, %s errorFactor %d: %.2f trials per block, %.3f of them repeat a triple already tried for the block (%.1f %%)"
              % (name, args.workload, args.error_factor, n / blocks, rep / blocks, 100.0 * rep / max(n, 1)))
    if where:
        print("fast search: most common repeats (position in the block's trial sequence, triple): ",
              sorted(where.items(), key=lambda kv: -kv[1])[:8])


if __name__ == "__main__":
    main()

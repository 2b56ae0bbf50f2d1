#!/usr/bin/env python3
"""Per-block statistics of the default shift search on the synthetic photo-noise workload, for tools/isa_budget.py's calibration: trials, factor rebuilds
(real ones and those to shift 8 of factors B / C, which cost two moves), block-error sums (trials no pixel fails), dither calls -- per errorFactor.
CPU only (the oracle's trial as the outcome function of the literal restatement of the reference's search, tools/make_search_table.py); test infrastructure.
usage: python tools/search_stats.py [--blocks 1500] > profiles/archive/r04_search_stats.json"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_search_table import search_fast  # noqa: E402
from oracle.bind import Oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=1500)
    ap.add_argument("--efs", default="25,50,100,200,400")
    args = ap.parse_args()
    orc = Oracle()
    W = H = 1024
    img = orc.photo_noise(W, H, 1)
    rng = np.random.default_rng(11)
    picks = [(int(rng.integers(0, W // 8)), int(rng.integers(0, H // 8))) for _ in range(args.blocks)]
    out = {}
    for ef in [int(v) for v in args.efs.split(",")]:
        acc = dict(trials=0, rebuild_real=0, rebuild_to8=0, sums=0, pixel_fail=0, block_fail=0, passed=0, dither_calls=0)
        for bx, by in picks:
            px = np.ascontiguousarray(img[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8]).ravel()
            rec = orc.block_fit(px, 4)
            a, b, c = orc.block_factors(px, 4, rec)
            g = search_fast()
            prev = None
            try:
                t = next(g)
                while True:
                    ok, be = orc.block_trial(px, 4, rec, a, b, c, t, ef)
                    acc["trials"] += 1
                    if prev is not None:  # (the first triple's three factors are built with immediates before the loop: set-up, not rebuilds)
                        for f in range(3):
                            if t[f] != prev[f]:
                                if f > 0 and t[f] > 7:
                                    acc["rebuild_to8"] += 1
                                else:
                                    acc["rebuild_real"] += 1
                    if ok:
                        acc["passed"] += 1; acc["sums"] += 1
                    elif be == 0:
                        acc["pixel_fail"] += 1  # trial_core returns at the first offending pixel, before the block error is written
                    else:
                        acc["block_fail"] += 1; acc["sums"] += 1
                    prev = t
                    t = g.send(ok)
            except StopIteration as e:
                acc["dither_calls"] += sum(1 for s in e.value if 0 < s < 8)
        out["ef%d" % ef] = {k: round(v / args.blocks, 4) for k, v in acc.items()}
    json.dump({"workload": "1024x1024 photo-noise seed 1, RGBA, %d sample blocks" % args.blocks, "per_block": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

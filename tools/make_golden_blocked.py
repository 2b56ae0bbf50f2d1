#!/usr/bin/env python3
"""Golden fixtures of the merged-block encoder from the REAL reference (oracle/_ref, `limg_blocked_encode3d_test`): FNV-1a-64 of every
plane upstream writes, for inputs the tests can rebuild (integer-defined generators, tests/golden/original.png).  -> tests/golden/blocked.json
Run in the container that has /root/reference (python tools/make_golden_blocked.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.bind import Oracle, Ref, BLOCKED_WRITTEN  # noqa: E402
import golden_util as gu  # noqa: E402

CASES = {
    # name: (generator, w, h, seed, alpha, kwargs)
    "rg_256x128": ("rg", 256, 128, 3, True, {}),
    "rga_256x128": ("rga", 256, 128, 3, True, {}),
    "pn_256x128": ("pn", 256, 128, 3, True, {}),
    "pn_rgb_200x123": ("pn", 200, 123, 5, False, {}),
    "rg_rgb_203x61": ("rg", 203, 61, 7, False, {}),
    "rg_512_ef25": ("rg", 512, 512, 1, True, {"error_factor": 25}),
    "pn_512_ef400": ("pn", 512, 512, 1, True, {"error_factor": 400}),
    "pn_256_accurate": ("pn", 256, 256, 2, True, {"fast": False}),
    "pn_256_pcg": ("pn", 256, 256, 2, True, {"dither_mode": 1}),
    "rg_1024": ("rg", 1024, 1024, 1, True, {}),
    "pn_1024": ("pn", 1024, 1024, 1, True, {}),
    "original_rgb": ("png", 1024, 618, 0, False, {}),
    "pn_ef0": ("pn", 128, 64, 9, True, {"error_factor": 0}),
}


def make_input(orc, gen, w, h, seed):
    if gen == "png":
        return gu.load_png()
    if gen == "pn":
        return orc.photo_noise(w, h, seed)
    return orc.random_gradient(w, h, seed, gen == "rg")


def main():
    orc, ref = Oracle(), Ref()
    out = {}
    for name, (gen, w, h, seed, alpha, kw) in CASES.items():
        img = make_input(orc, gen, w, h, seed)
        r = ref.blocked_encode3d(img, alpha, **kw)
        psnr, mse = ref.compare(img, r["pDecoded"], alpha)
        out[name] = {"gen": gen, "w": w, "h": h, "seed": seed, "alpha": alpha, "kw": kw, "input": orc.fnv(img), "psnr": psnr, "mse": mse,
                     "regions": int(r["pBlockIndex"].max() & 0xFFFFFF), "planes": {k: orc.fnv(r[k]) for k in BLOCKED_WRITTEN}}
        print(name, out[name]["regions"], "%.4f dB" % psnr)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "blocked.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden text of the reference's statistics block -- "Average Block Bits" and the per-factor shift histogram that limg_encode3d_test and limg_blocked_encode3d_test
print themselves (src/limg.cpp:2232-2248, :2397-2440; PRINT_TEST_OUTPUT is always defined, src/limg_internal.h:9) -- captured from the REAL reference
(oracle/_ref) at file-descriptor level, for inputs the tests can rebuild.  -> tests/golden/stats.json
Run in the container that has /root/reference (python tools/make_golden_stats.py)."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle.bind import Oracle, Ref  # noqa: E402
import golden_util as gu  # noqa: E402

CASES = {
    "original_rgb": ("png", 0, 0, 0, False, {}),
    "pn_256x64": ("pn", 256, 64, 5, True, {}),
    "rg_256x64": ("rg", 256, 64, 5, True, {}),
    "pn_61x27_rgb_ef25": ("pn", 61, 27, 13, False, {"error_factor": 25}),
    "pn_256x264_pool2": ("pn", 256, 264, 23, True, {"pool_threads": 2}),
}


def captured(fn):
    sys.stdout.flush()
    with tempfile.TemporaryFile() as tmp:
        saved = os.dup(1)
        os.dup2(tmp.fileno(), 1)
        try:
            fn()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tmp.seek(0)
        return tmp.read().decode()


def main():
    orc, ref = Oracle(), Ref()
    ref.lib.ref_keep_stdout(1)
    out = {}
    for name, (gen, w, h, seed, alpha, kw) in CASES.items():
        img = gu.load_png() if gen == "png" else (orc.photo_noise(w, h, seed) if gen == "pn" else orc.random_gradient(w, h, seed, True))
        fixed = captured(lambda: ref.encode3d(img, alpha, **kw))
        kwb = {k: v for k, v in kw.items() if k != "pool_threads"}
        blocked = captured(lambda: ref.blocked_encode3d(img, alpha, **kwb))
        out[name] = {"gen": gen, "w": int(img.shape[1]), "h": int(img.shape[0]), "seed": seed, "alpha": alpha, "kw": kw, "input": orc.fnv(img),
                     "fixed_blocks_stdout": fixed, "merged_blocks_stdout": blocked}
        print(name, fixed.strip().splitlines()[0])
        print(name, [l for l in blocked.splitlines() if "Average" in l or "Compression" in l])
    ref.lib.ref_keep_stdout(0)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "stats.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (run on the GPU box): random shapes (ragged included), channel counts, generators, error factors, strip
partitions, accurate mode, PCG dither -- for the 8x8 path (all 11 planes), the compact stream (bytes + round trip) and the merged-block encoder
(13 planes + rectangles).  usage: python tools/fuzz_gpu.py [--seconds 240] [--seed 1]    exit code 1 on the first mismatch (prints the recipe)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LIMG_HIP_LIB", "test")  # the options this sweep randomises include hooks of the test build (include/limg_hip_test_hooks.h); LIMG_HIP_LIB=<path> to fuzz the product


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-blocks", type=int, default=40, help="largest image width in 8-pixel blocks (height: 60 %% of it)")
    args = ap.parse_args()
    import limg_amd
    from oracle.bind import Oracle, PLANES, BLOCKED_WRITTEN
    from oracle import stream as S
    orc = Oracle()
    g = limg_amd.LimgHip(0)
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    n = {"fixed": 0, "stream": 0, "blocked": 0, "batch": 0}
    last = t0
    while time.time() - t0 < args.seconds:
        w = int(rng.integers(1, args.max_blocks)) * 8 + (int(rng.integers(0, 8)) if rng.random() < 0.4 else 0)
        h = int(rng.integers(1, max(2, args.max_blocks * 6 // 10))) * 8 + (int(rng.integers(0, 8)) if rng.random() < 0.4 else 0)
        gen = ["pn", "rg", "rga", "rand", "flat"][int(rng.integers(0, 5))]
        seed = int(rng.integers(1, 1 << 30))
        alpha = bool(rng.random() < 0.7)
        if gen == "pn":
            img = orc.photo_noise(w, h, seed)
        elif gen in ("rg", "rga"):
            img = orc.random_gradient(w, h, seed, gen == "rg")
        elif gen == "rand":
            img = rng.integers(0, 1 << 32, (h, w), dtype=np.uint64).astype(np.uint32)
        else:
            img = np.full((h, w), int(rng.integers(0, 1 << 32)), dtype=np.uint32)
            img[rng.integers(0, h), rng.integers(0, w)] ^= 0x00FFFFFF
        ef = int([0, 10, 25, 50, 100, 100, 100, 200, 400, 1000][int(rng.integers(0, 10))])
        fast = bool(rng.random() < 0.8)
        pcg = bool(rng.random() < 0.2)
        pool = int([0, 0, 0, 1, 2, 3][int(rng.integers(0, 6))])
        split = bool(rng.random() < 0.3)
        legacy = bool(rng.random() < 0.25)  # float stage with lane == pixel inside the E step instead of k_fit_tpb
        recipe = dict(w=w, h=h, gen=gen, seed=seed, alpha=alpha, ef=ef, fast=fast, pcg=pcg, pool=pool, split=split, legacy=legacy)
        kw = dict(error_factor=ef, fast=fast)
        whole = bool(rng.random() < 0.3)  # images with a partial last block row: the whole-image host walk instead of fast path + last row
        recipe["whole_image_ragged"] = whole
        bands = int([0, 0, 2, 3, 7, -1][int(rng.integers(0, 6))])   # images with a partial last column: the host's chain walk pipelined in this many bands
        wthreads = int([0, 1, 2, 5][int(rng.integers(0, 4))])       # ... or, with several chains, walked on this many host threads
        recipe["ragged_bands"], recipe["ragged_walk_threads"] = bands, wthreads
        g.set_options(force_split=split, dither_pcg=pcg, legacy_float_stage=legacy, test_whole_image_ragged=whole, ragged_bands=bands, ragged_walk_threads=wthreads)
        mode = ["fixed", "stream", "blocked", "batch"][int(rng.integers(0, 4))]
        if mode == "fixed":
            want = orc.encode3d(img, alpha, pool_threads=pool, dither_mode=int(pcg), **kw)
            got = g.encode3d(img, alpha, pool_threads=pool, **kw)
            bad = [k for k in PLANES if not np.array_equal(got[k], want[k])]
        elif mode == "stream":
            want = orc.encode3d(img, alpha, extras=True, pool_threads=pool, dither_mode=int(pcg), **kw)
            st = g.encode_stream(img, alpha, pool_threads=pool, **kw)
            ref = S.pack(want, w, h, 4 if alpha else 3, error_factor=ef, flags=(1 if fast else 0) | (2 if pcg else 0))
            bad = [] if (st.size == ref.size and np.array_equal(st, ref)) else ["stream bytes"]
            if not np.array_equal(g.decode_stream(st), want["pDecoded"]):
                bad.append("decode")
        elif mode == "batch":
            # limg_hip_encode3d_batch_device: 2..4 images of this shape (this one + variations of it) in one launch pair, each against its own single-image oracle encode
            import torch
            cnt = int(rng.integers(2, 7))
            sub = int([0, 0, 1, 2, 3][int(rng.integers(0, 5))])  # the list as a pipeline of sub-batches of this many images (0: the library's rule)
            recipe["batch"] = (cnt, sub)
            host = [img] + [np.ascontiguousarray(np.roll(img, int(rng.integers(1, 64)), axis=1) ^ np.uint32(int(rng.integers(0, 1 << 24)))) for _ in range(cnt - 1)]
            dev = [torch.from_numpy(x.view(np.int32)).cuda() for x in host]
            outs = [g.alloc_planes_device(w, h) for _ in host]
            g.set_options(force_split=split, dither_pcg=pcg, legacy_float_stage=legacy, test_batch_chunk=int(rng.integers(0, 4)), batch_sub_images=sub)
            g.encode3d_batch_device(dev, alpha, outs, pool_threads=pool, **kw)
            torch.cuda.synchronize()
            bad = []
            for x, pl in zip(host, outs):
                want = orc.encode3d(x, alpha, pool_threads=pool, dither_mode=int(pcg), **kw)
                for k in PLANES:
                    got = pl[k].cpu().numpy()
                    got = got.view(np.uint32) if got.dtype == np.int32 else got
                    if not np.array_equal(got, want[k]):
                        bad.append(k)
        else:
            want = orc.blocked_encode3d(img, alpha, dither_mode=int(pcg), **kw)
            got = g.blocked_encode3d(img, alpha, **kw)
            bad = [k for k in BLOCKED_WRITTEN if not np.array_equal(got[k], want[k])]
            if len(got["regions"]) != len(want["regions"]):
                bad.append("regions")
        if bad:
            print("MISMATCH", mode, bad, recipe, flush=True)
            sys.exit(1)
        n[mode] += 1
        if time.time() - last > 30:
            last = time.time()
            print("ok so far:", n, "%.0fs" % (last - t0), flush=True)
    g.check()
    print("fuzz ok:", n, "cases in %.0fs" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()

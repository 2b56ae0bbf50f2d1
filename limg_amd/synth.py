"""Bit-reproducible synthetic inputs and hashes (SURVEY.md section 8(d) definitions, integer-only).

The same generators exist as HIP kernels (csrc/limg_hip_synth.hip, `limg_hip_synth_*`) for the bench, where
8192x8192 images are produced directly in HBM; tests check the two against each other and against the
input hashes recorded in SURVEY.md Appendix E.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLD = np.uint64(0x9E3779B97F4A7C15)


def sm64(x):
    """splitmix64 finaliser on a uint64 ndarray (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = (np.asarray(x, dtype=np.uint64) + GOLD)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def random_gradient(width, height, seed=1, opaque=True, y0=0):
    """64x64-px tiles, each a linear blend between two random RGBA colours along a random direction."""
    ys = (np.arange(height, dtype=np.uint64) + np.uint64(y0))[:, None]
    xs = np.arange(width, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        ty = ys >> np.uint64(6)
        tx = xs >> np.uint64(6)
        h = sm64(np.uint64(seed) ^ (ty * GOLD + tx))
        h2 = sm64(h)
        gx = (h2 & np.uint64(7)).astype(np.int64)
        gy = ((h2 >> np.uint64(3)) & np.uint64(7)).astype(np.int64)
        s = (xs & np.uint64(63)).astype(np.int64) * gx + (ys & np.uint64(63)).astype(np.int64) * gy
        m = 63 * (gx + gy)
        zero = m == 0
        m = np.where(zero, 1, m)
        s = np.where(zero, 0, s)
        out = np.zeros((height, width), dtype=np.uint32)
        for c in range(4):
            c0 = ((h >> np.uint64(8 * c)) & np.uint64(255)).astype(np.int64)
            c1 = ((h >> np.uint64(32 + 8 * c)) & np.uint64(255)).astype(np.int64)
            v = (c0 * (m - s) + c1 * s + m // 2) // m
            if c == 3 and opaque:
                v = np.full_like(v, 255)
            out |= (v.astype(np.uint32) & np.uint32(255)) << np.uint32(8 * c)
    return out


def photo_noise(width, height, seed=1, y0=0, full_width=None):
    """Bilinear 32-px lattice of random colours plus +-8 per-channel noise, opaque alpha."""
    W = np.uint64(full_width if full_width is not None else width)
    ys = (np.arange(height, dtype=np.uint64) + np.uint64(y0))[:, None]
    xs = np.arange(width, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        ly = ys >> np.uint64(5)
        lx = xs >> np.uint64(5)
        fy = (ys & np.uint64(31)).astype(np.int64)
        fx = (xs & np.uint64(31)).astype(np.int64)

        def corner(dy, dx):
            return sm64(np.uint64(seed) ^ ((ly + np.uint64(dy)) * GOLD + (lx + np.uint64(dx))))

        ha, hb, hc, hd = corner(0, 0), corner(0, 1), corner(1, 0), corner(1, 1)
        hn = sm64(np.uint64(seed) * np.uint64(31) + ys * W + xs)
        out = np.zeros((height, width), dtype=np.uint32)
        for c in range(3):
            sh = np.uint64(8 * c)
            a = ((ha >> sh) & np.uint64(255)).astype(np.int64)
            b = ((hb >> sh) & np.uint64(255)).astype(np.int64)
            cc = ((hc >> sh) & np.uint64(255)).astype(np.int64)
            d = ((hd >> sh) & np.uint64(255)).astype(np.int64)
            v = ((a * (32 - fx) + b * fx) * (32 - fy) + (cc * (32 - fx) + d * fx) * fy + 512) >> 10
            n = ((hn >> sh) & np.uint64(15)).astype(np.int64) - 8
            v = np.clip(v + n, 0, 255)
            out |= v.astype(np.uint32) << np.uint32(8 * c)
        out |= np.uint32(0xFF000000)
    return out


def fnv1a64(buf):
    """FNV-1a 64 over raw bytes (hash used for the plane known-answers in SURVEY.md 8(c) / Appendix E)."""
    data = np.ascontiguousarray(buf).view(np.uint8).ravel()
    # vectorising FNV is awkward (sequential multiply); do it in chunks of python ints via a tiny C-free loop
    h = 0xCBF29CE484222325
    p = 0x100000001B3
    mask = 0xFFFFFFFFFFFFFFFF
    for b in data.tobytes():
        h = ((h ^ b) * p) & mask
    return "%016x" % h

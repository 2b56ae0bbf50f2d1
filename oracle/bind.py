"""ctypes bindings: `Oracle` = oracle/liblimg_oracle.so (our CPU restatement), `Ref` = oracle/_ref/liblimg_ref*.so (the real
reference, built by oracle/build_ref.sh where /root/reference exists).  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
P32 = ("pDecoded", "pShiftABCX", "pColAMin", "pColAMax", "pColBMin", "pColBMax", "pColCMin", "pColCMax")
P8 = ("pFactorsA", "pFactorsB", "pFactorsC")
PLANES = P32 + P8

# limg_blocked_encode3d_info (src/limg.h:39-44), member order
BLOCKED_PLANES = (("pDecoded", np.uint32), ("pFactorsA", np.uint8), ("pFactorsB", np.uint8), ("pFactorsC", np.uint8), ("pBlockError", np.uint8), ("pBitsPerPixel", np.uint8),
                  ("pShiftABCX", np.uint32), ("pColAMin", np.uint32), ("pColAMax", np.uint32), ("pColBMin", np.uint32), ("pColBMax", np.uint32), ("pColCMin", np.uint32),
                  ("pColCMax", np.uint32), ("pBlockIndex", np.uint32))
BLOCKED_WRITTEN = tuple(k for k, _ in BLOCKED_PLANES if k != "pBlockError")  # upstream never writes pBlockError

FLOAT_X86, FLOAT_TREE = 0, 1
DITHER_AES, DITHER_PCG = 0, 1


def alloc_planes(w, h):
    d = {k: np.zeros((h, w), dtype=np.uint32) for k in P32}
    d.update({k: np.zeros((h, w), dtype=np.uint8) for k in P8})
    return d


class Record(C.Structure):
    _fields_ = [("avg", C.c_float * 4), ("dirA_min", C.c_int16 * 4), ("dirA_max", C.c_int16 * 4), ("dirB_offset", C.c_int16 * 4),
                ("dirB_mag", C.c_int16 * 4), ("dirC_offset", C.c_int16 * 4), ("dirC_mag", C.c_int16 * 4)]


REC_DTYPE = np.dtype([("avg", "<f4", 4), ("dirA_min", "<i2", 4), ("dirA_max", "<i2", 4), ("dirB_offset", "<i2", 4),
                      ("dirB_mag", "<i2", 4), ("dirC_offset", "<i2", 4), ("dirC_mag", "<i2", 4)])
REC3_DTYPE = np.dtype([("avg", "<f4", 3), ("dirA_min", "<i2", 3), ("dirA_max", "<i2", 3), ("dirB_offset", "<i2", 3),
                       ("dirB_mag", "<i2", 3), ("dirC_offset", "<i2", 3), ("dirC_mag", "<i2", 3)])
assert REC_DTYPE.itemsize == 64 and REC3_DTYPE.itemsize == 48


class Info(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in PLANES]


class Config(C.Structure):
    _fields_ = [("error_factor", C.c_uint32), ("fast_bit_crush", C.c_int32), ("float_mode", C.c_int32), ("dither_mode", C.c_int32),
                ("pool_threads", C.c_int32), ("worker_threads", C.c_int32), ("forced_shift", C.c_int32 * 3)]


REGION_DTYPE = np.dtype([("ox", "<u4"), ("oy", "<u4"), ("rx", "<u4"), ("ry", "<u4"), ("shift", "u1", 3), ("calls", "u1"), ("keep", "<u4"), ("rec", REC_DTYPE)])
assert REGION_DTYPE.itemsize == 88


def alloc_blocked_planes(w, h):
    return {k: np.zeros((h, w), dtype=t) for k, t in BLOCKED_PLANES}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(HERE, "liblimg_oracle.so")
        self.lib = L = C.CDLL(path)
        L.limg_oracle_encode3d.restype = C.c_int
        L.limg_oracle_encode3d.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p] + [C.c_void_p] * 6
        L.limg_oracle_compare.restype = C.c_double
        L.limg_oracle_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.limg_oracle_block_fit.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.limg_oracle_block_factors.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.limg_oracle_block_trial.restype = C.c_int
        L.limg_oracle_block_trial.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p]
        L.limg_oracle_block_search.restype = C.c_uint32
        L.limg_oracle_block_search.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 4 + [C.c_uint32, C.c_int, C.c_void_p]
        L.limg_oracle_dither.restype = C.c_uint64
        L.limg_oracle_dither.argtypes = [C.c_int, C.c_size_t, C.c_uint64, C.c_void_p, C.c_int]
        L.limg_oracle_chain_step.restype = C.c_uint64
        L.limg_oracle_chain_step.argtypes = [C.c_size_t, C.c_uint64, C.c_int]
        L.limg_oracle_block_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int] + [C.c_void_p] * 5
        L.limg_oracle_rsqrt_x86.restype = C.c_float
        L.limg_oracle_rsqrt_x86.argtypes = [C.c_float]
        L.limg_oracle_fnv1a64.restype = C.c_uint64
        L.limg_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        L.limg_oracle_synth_random_gradient.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_int]
        L.limg_oracle_synth_photo_noise.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64]
        L.limg_oracle_blocked_encode3d.restype = C.c_int
        L.limg_oracle_blocked_encode3d.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.limg_oracle_blocked_matches.restype = C.c_int
        L.limg_oracle_blocked_matches.argtypes = [C.c_int, C.c_void_p, C.c_void_p]

    def config(self, error_factor=100, fast=True, float_mode=FLOAT_X86, dither_mode=DITHER_AES, pool_threads=0, worker_threads=1, forced_shift=None):
        cfg = Config()
        cfg.error_factor, cfg.fast_bit_crush, cfg.float_mode, cfg.dither_mode = error_factor, int(fast), float_mode, dither_mode
        cfg.pool_threads, cfg.worker_threads = pool_threads, worker_threads
        fs = forced_shift if forced_shift is not None else (-1, -1, -1)
        for i in range(3):
            cfg.forced_shift[i] = fs[i]
        return cfg

    def encode3d(self, img, has_alpha, planes=True, extras=False, **kw):
        """img: (h, w) uint32.  Returns dict of planes (+ 'records','shifts','preA/B/C','trials' when extras)."""
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        cfg = self.config(**kw)
        out = alloc_planes(w, h) if planes else {}
        info = None
        if planes:
            info = Info(*[out[k].ctypes.data for k in PLANES])
        bx, by = (w + 7) // 8, (h + 7) // 8
        rec = sh = pa = pb = pc = None
        trials = C.c_uint64(0)
        if extras:
            rec = np.zeros(bx * by, dtype=REC_DTYPE)
            sh = np.zeros((by, bx, 3), dtype=np.uint8)
            pa, pb, pc = (np.zeros((h, w), dtype=np.uint8) for _ in range(3))
        r = self.lib.limg_oracle_encode3d(_ptr(img), w, h, int(has_alpha), C.byref(info) if info else None, C.byref(cfg),
                                          _ptr(rec), _ptr(sh), _ptr(pa), _ptr(pb), _ptr(pc), C.byref(trials))
        assert r == 0, r
        if extras:
            out.update(records=rec.reshape(by, bx), shifts=sh, preA=pa, preB=pb, preC=pc)
        out["trials"] = trials.value
        return out

    def blocked_encode3d(self, img, has_alpha, planes=True, **kw):
        """limg_blocked_encode3d_test.  Returns the 14 planes (+ 'pass1' records, 'regions' in creation order)."""
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        cfg = self.config(**kw)
        out = alloc_blocked_planes(w, h) if planes else {}
        info = (C.c_void_p * 14)(*[out[k].ctypes.data for k, _ in BLOCKED_PLANES]) if planes else None
        bx, by = (w + 7) // 8, (h + 7) // 8
        pass1 = np.zeros(bx * by, dtype=REC_DTYPE)
        regions = np.zeros(bx * by, dtype=REGION_DTYPE)
        n = C.c_size_t(0)
        r = self.lib.limg_oracle_blocked_encode3d(_ptr(img), w, h, int(has_alpha), info, C.byref(cfg), _ptr(pass1), _ptr(regions), regions.size, C.byref(n))
        assert r == 0, r
        out.update(pass1=pass1.reshape(by, bx), regions=regions[:n.value].copy())
        return out

    def blocked_matches(self, channels, a, b):
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
        return bool(self.lib.limg_oracle_blocked_matches(channels, _ptr(a), _ptr(b)))

    def compare(self, a, b, has_alpha):
        a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
        mse, mx = C.c_double(), C.c_double()
        p = self.lib.limg_oracle_compare(_ptr(a), _ptr(b), a.shape[1], a.shape[0], int(has_alpha), C.byref(mse), C.byref(mx))
        return p, mse.value

    def block_fit(self, px, channels, float_mode=FLOAT_X86):
        px = np.ascontiguousarray(px, dtype=np.uint32).ravel()
        rec = np.zeros(1, dtype=REC_DTYPE)
        self.lib.limg_oracle_block_fit(_ptr(px), px.size, channels, float_mode, _ptr(rec))
        return rec

    def block_factors(self, px, channels, rec):
        px = np.ascontiguousarray(px, dtype=np.uint32).ravel()
        a, b, c = (np.zeros(px.size, dtype=np.uint8) for _ in range(3))
        self.lib.limg_oracle_block_factors(_ptr(px), px.size, channels, _ptr(rec), _ptr(a), _ptr(b), _ptr(c))
        return a, b, c

    def block_trial(self, px, channels, rec, a, b, c, shift, error_factor=100):
        px = np.ascontiguousarray(px, dtype=np.uint32).ravel()
        sh = np.asarray(shift, dtype=np.uint8)
        be = C.c_uint64(0)
        ok = self.lib.limg_oracle_block_trial(_ptr(px), px.size, channels, _ptr(rec), _ptr(a), _ptr(b), _ptr(c), _ptr(sh), error_factor, C.byref(be))
        return bool(ok), be.value

    def block_search(self, px, channels, rec, a, b, c, error_factor=100, fast=True):
        px = np.ascontiguousarray(px, dtype=np.uint32).ravel()
        sh = np.zeros(3, dtype=np.uint8)
        n = self.lib.limg_oracle_block_search(_ptr(px), px.size, channels, _ptr(rec), _ptr(a), _ptr(b), _ptr(c), error_factor, int(fast), _ptr(sh))
        return sh, n

    def dither(self, shift, hash_, f, mode=DITHER_AES):
        f = np.array(f, dtype=np.uint8)
        h = self.lib.limg_oracle_dither(shift, f.size, hash_, _ptr(f), mode)
        return h, f

    def chain_step(self, n, hash_, mode=DITHER_AES):
        return self.lib.limg_oracle_chain_step(n, hash_, mode)

    def block_decode(self, rx, ry, channels, rec, a, b, c, shift):
        out = np.zeros((ry, rx), dtype=np.uint32)
        sh = np.asarray(shift, dtype=np.uint8)
        self.lib.limg_oracle_block_decode(_ptr(out), rx, rx, ry, channels, _ptr(rec), _ptr(a), _ptr(b), _ptr(c), _ptr(sh))
        return out

    def fnv(self, a):
        a = np.ascontiguousarray(a)
        return "%016x" % self.lib.limg_oracle_fnv1a64(_ptr(a), a.nbytes)

    def random_gradient(self, w, h, seed=1, opaque=True):
        out = np.zeros((h, w), dtype=np.uint32)
        self.lib.limg_oracle_synth_random_gradient(_ptr(out), w, h, seed, int(opaque))
        return out

    def photo_noise(self, w, h, seed=1):
        out = np.zeros((h, w), dtype=np.uint32)
        self.lib.limg_oracle_synth_photo_noise(_ptr(out), w, h, seed)
        return out


def ref_available(fastmath=False):
    return os.path.exists(os.path.join(HERE, "_ref", "liblimg_ref_fastmath.so" if fastmath else "liblimg_ref.so"))


class Ref:
    """The real reference (strict-IEEE build by default)."""

    def __init__(self, fastmath=False):
        self.lib = L = C.CDLL(os.path.join(HERE, "_ref", "liblimg_ref_fastmath.so" if fastmath else "liblimg_ref.so"))
        L.ref_encode3d.restype = C.c_int
        L.ref_encode3d.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.ref_encode3d_perf.restype = C.c_int
        L.ref_encode3d_perf.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.ref_compare.restype = C.c_double
        L.ref_compare.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_block_fit.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.ref_block_factors.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 4
        L.ref_block_trial.restype = C.c_int
        L.ref_block_trial.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p]
        L.ref_block_search.argtypes = [C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 4 + [C.c_uint32, C.c_int, C.c_void_p]
        L.ref_dither.restype = C.c_uint64
        L.ref_dither.argtypes = [C.c_int, C.c_size_t, C.c_uint64, C.c_void_p, C.c_int]
        L.ref_block_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int] + [C.c_void_p] * 5
        if hasattr(L, "ref_encode3d_forced_shift"):
            L.ref_encode3d_forced_shift.restype = C.c_int
            L.ref_encode3d_forced_shift.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(L, "ref_blocked_encode3d"):
            L.ref_blocked_encode3d.restype = C.c_int
            L.ref_blocked_encode3d.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int]
            L.ref_blocked_matches.restype = C.c_int
            L.ref_blocked_matches.argtypes = [C.c_int, C.c_void_p, C.c_void_p]

    @staticmethod
    def _rec_in(rec, channels):
        """oracle-layout record (64 B) -> reference layout for `channels`."""
        if channels == 4:
            return np.ascontiguousarray(rec)
        r3 = np.zeros(1, dtype=REC3_DTYPE)
        for k in REC3_DTYPE.names:
            r3[k][0] = rec[k].reshape(-1)[:3]
        return r3

    @staticmethod
    def _rec_out(raw, channels):
        if channels == 4:
            return raw
        r = np.zeros(1, dtype=REC_DTYPE)
        for k in REC3_DTYPE.names:
            r[k][0][:3] = raw[k][0]
        return r

    def encode3d(self, img, has_alpha, error_factor=100, pool_threads=0, fast=True, dither_mode=DITHER_AES):
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        out = alloc_planes(w, h)
        p32 = (C.c_void_p * 8)(*[out[k].ctypes.data for k in P32])
        p8 = (C.c_void_p * 3)(*[out[k].ctypes.data for k in P8])
        r = self.lib.ref_encode3d(_ptr(img), w, h, int(has_alpha), p32, p8, error_factor, pool_threads, int(fast), dither_mode)
        assert r == 0, r
        return out

    def encode3d_forced_shift(self, img, has_alpha, shift):
        """The reference's block functions in limg_encode3d_test's order with the search replaced by `shift` (oracle/ref_harness.cpp): the shift-dependent planes
        pDecoded, pFactorsA/B/C of one dither chain."""
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        out = {"pDecoded": np.zeros((h, w), dtype=np.uint32)}
        out.update({k: np.zeros((h, w), dtype=np.uint8) for k in P8})
        p8 = (C.c_void_p * 3)(*[out[k].ctypes.data for k in P8])
        sh = np.asarray(shift, dtype=np.uint8)
        r = self.lib.ref_encode3d_forced_shift(_ptr(img), w, h, int(has_alpha), _ptr(out["pDecoded"]), p8, _ptr(sh))
        assert r == 0, r
        return out

    def blocked_encode3d(self, img, has_alpha, error_factor=100, pool_threads=0, fast=True, dither_mode=DITHER_AES):
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        out = alloc_blocked_planes(w, h)
        planes = (C.c_void_p * 14)(*[out[k].ctypes.data for k, _ in BLOCKED_PLANES])
        r = self.lib.ref_blocked_encode3d(_ptr(img), w, h, int(has_alpha), planes, error_factor, pool_threads, int(fast), dither_mode)
        assert r == 0, r
        return out

    def blocked_matches(self, channels, a, b):
        ra = np.concatenate([self._rec_in(a, channels).view(np.uint8), np.zeros(32, np.uint8)])
        rb = np.concatenate([self._rec_in(b, channels).view(np.uint8), np.zeros(32, np.uint8)])
        return bool(self.lib.ref_blocked_matches(channels, _ptr(ra), _ptr(rb)))

    def encode3d_perf(self, img, has_alpha, error_factor=100, pool_threads=0, fast=True):
        img = np.ascontiguousarray(img, dtype=np.uint32)
        h, w = img.shape
        return self.lib.ref_encode3d_perf(_ptr(img), w, h, int(has_alpha), error_factor, pool_threads, int(fast), 0)

    def compare(self, a, b, has_alpha):
        a = np.ascontiguousarray(a, dtype=np.uint32); b = np.ascontiguousarray(b, dtype=np.uint32)
        mse, mx = C.c_double(), C.c_double()
        p = self.lib.ref_compare(_ptr(a), _ptr(b), a.shape[1], a.shape[0], int(has_alpha), C.byref(mse), C.byref(mx))
        return p, mse.value

    def block_fit(self, px, channels):
        px = np.ascontiguousarray(px, dtype=np.uint32).ravel()
        raw = np.zeros(1, dtype=REC_DTYPE if channels == 4 else REC3_DTYPE)
        pad = np.zeros(64, dtype=np.uint8)  # the 3-ch path stores 16 bytes of avg into a 12-byte field (src/limg_factorization.h:553)
        buf = np.concatenate([raw.view(np.uint8), pad])
        self.lib.ref_block_fit(_ptr(px), px.size, channels, _ptr(buf))
        raw = buf[:raw.nbytes].view(raw.dtype)
        return self._rec_out(raw, channels)

    def block_factors(self, px, channels, rec):
        px = np.ascontiguousarray(px, dtype=np.uint32).ravel()
        a, b, c = (np.zeros(px.size + 16, dtype=np.uint8) for _ in range(3))
        r = np.concatenate([self._rec_in(rec, channels).view(np.uint8), np.zeros(32, np.uint8)])
        pxp = np.concatenate([px, np.zeros(4, np.uint32)])
        self.lib.ref_block_factors(_ptr(pxp), px.size, channels, _ptr(r), _ptr(a), _ptr(b), _ptr(c))
        return a[:px.size], b[:px.size], c[:px.size]

    def block_trial(self, px, channels, rec, a, b, c, shift, error_factor=100):
        px = np.ascontiguousarray(px, dtype=np.uint32).ravel()
        sh = np.asarray(shift, dtype=np.uint8)
        be = C.c_uint64(0)
        r = np.concatenate([self._rec_in(rec, channels).view(np.uint8), np.zeros(32, np.uint8)])
        ok = self.lib.ref_block_trial(_ptr(px), px.size, channels, _ptr(r), _ptr(a), _ptr(b), _ptr(c), _ptr(sh), error_factor, C.byref(be))
        return bool(ok), be.value

    def block_search(self, px, channels, rec, a, b, c, error_factor=100, fast=True):
        px = np.ascontiguousarray(px, dtype=np.uint32).ravel()
        sh = np.zeros(3, dtype=np.uint8)
        r = np.concatenate([self._rec_in(rec, channels).view(np.uint8), np.zeros(32, np.uint8)])
        self.lib.ref_block_search(_ptr(px), px.size, channels, _ptr(r), _ptr(a), _ptr(b), _ptr(c), error_factor, int(fast), _ptr(sh))
        return sh

    def dither(self, shift, hash_, f, mode=DITHER_AES):
        n = len(f)
        buf = np.zeros(n + 16, dtype=np.uint8)
        buf[:n] = f
        h = self.lib.ref_dither(shift, n, hash_, _ptr(buf), mode)
        return h, buf[:n].copy()

    def block_decode(self, rx, ry, channels, rec, a, b, c, shift):
        out = np.zeros((ry, rx), dtype=np.uint32)
        sh = np.asarray(shift, dtype=np.uint8)
        r = np.concatenate([self._rec_in(rec, channels).view(np.uint8), np.zeros(32, np.uint8)])
        self.lib.ref_block_decode(_ptr(out), rx, rx, ry, channels, _ptr(r), _ptr(a), _ptr(b), _ptr(c), _ptr(sh))
        return out

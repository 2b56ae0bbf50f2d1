"""Maximum-size check on the GPU box (not part of the suite: ~45 GB of device memory): one N x N photo-noise image, N = 32768 by default -- 1 Gpixel, 16.7 M blocks, planes
whose byte offsets pass 2^32, a dither chain of ~50 M calls (beyond the embedded checkpoints' reach: the noise table comes from the host walk).  Properties instead of a CPU
run of the whole image: (1) status clean, PSNR of the generator; (2) the first 64 rows equal the oracle on every plane (the chain starts at the seed); (3) a middle and the LAST
64-row band equal the oracle on the chain-independent planes; (4) strip-restart encode (pool of 2 = 8 strips): strips 0 and 7 equal their standalone encodes on every plane
(strip 7 lies behind the 4 GiB offset in every 32-bit plane); (5) N x (N - 2), a partial last block row ~50 M calls into the chain: the fast path (the chain value there from
the far checkpoints + at most 65535 calls on foot) equals the whole-image ragged path (host walk over every call) on every plane; (6) the compact stream's round trip.  usage: python tools/huge_image_check.py [N]"""
import os
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch
os.environ.setdefault("LIMG_HIP_LIB", "test")  # property (5) uses a hook of the test build
import limg_amd
from oracle.bind import Oracle, PLANES

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
orc = Oracle()
g = limg_amd.LimgHip(0)
t = time.perf_counter()
img = g.synth_device("photo_noise", N, N, seed=1)
planes = g.alloc_planes_device(N, N)
torch.cuda.synchronize()
print("alloc + synth %.1f s, device bytes of the planes %.1f GiB" % (time.perf_counter() - t, sum(v.numel() * v.element_size() for v in planes.values()) / 2**30), flush=True)
for i in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    g.encode3d_device(img, True, planes); torch.cuda.synchronize()
    print("encode %d: %.2f ms" % (i, (time.perf_counter() - t) * 1e3), flush=True)
g.check()
psnr = g.compare_device(img, planes["pDecoded"], True)[0]
print("psnr %.4f" % psnr, flush=True)
assert abs(psnr - 38.87) < 0.1, psnr
UNIFORM = ("pShiftABCX", "pColAMin", "pColAMax", "pColBMin", "pColBMax", "pColCMin", "pColCMax")


def band(y0, keys):
    want = orc.encode3d(img[y0:y0 + 64].cpu().numpy().view(np.uint32), True)
    for k in keys:
        got = planes[k][y0:y0 + 64].cpu().numpy()
        got = got.view(np.uint32) if got.dtype == np.int32 else got
        assert np.array_equal(got, want[k]), (y0, k)
    print("band at row %d: %d planes equal the oracle" % (y0, len(keys)), flush=True)


band(0, PLANES)
band(N // 2, UNIFORM)
band(N - 64, UNIFORM)
g.encode3d_device(img, True, planes, pool_threads=2); torch.cuda.synchronize()
rows = (N // 8 // 8) * 8
part = g.alloc_planes_device(N, rows)
for s in (0, 7):
    g.encode3d_device(img[s * rows:(s + 1) * rows], True, part); torch.cuda.synchronize()
    for k in PLANES:
        assert torch.equal(part[k], planes[k][s * rows:(s + 1) * rows]), (s, k)
    print("strip %d of 8 == its standalone encode on all %d planes" % (s, len(PLANES)), flush=True)
band(7 * rows, PLANES)  # strip 7's chain restarts at the seed: its first rows against the oracle on every plane, behind the 4 GiB offset
del part
# (6) compact stream at this size: decode(encode(image)) == pDecoded of the single-chain encode (payload offsets and tile sums far beyond what 8192^2 reaches)
g.encode3d_device(img, True, planes); torch.cuda.synchronize()
for i in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    st, nbytes = g.encode_stream_device(img, True); torch.cuda.synchronize()
    t1 = time.perf_counter()
    dec = g.decode_stream_device(st, nbytes, N, N); torch.cuda.synchronize()
    print("stream %d: encode %.2f ms, %.3f B/px, decode %.2f ms" % (i, (t1 - t) * 1e3, nbytes / (N * N), (time.perf_counter() - t1) * 1e3), flush=True)
    assert torch.equal(dec, planes["pDecoded"])
    del st, dec
print("stream round trip == pDecoded", flush=True)
g.check()
del planes
torch.cuda.empty_cache()
H = N - 2
pa, pb = g.alloc_planes_device(N, H), g.alloc_planes_device(N, H)
for name, whole, out in (("fast path + last row", False, pa), ("whole-image ragged path", True, pb)):
    g.set_options(test_whole_image_ragged=whole)
    for i in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        g.encode3d_device(img[:H], True, out); torch.cuda.synchronize()
        print("%d x %d, %s, encode %d: %.2f ms" % (N, H, name, i, (time.perf_counter() - t) * 1e3), flush=True)
for k in PLANES:
    assert torch.equal(pa[k], pb[k]), k
print("%d x %d: fast path == whole-image ragged path on all %d planes" % (N, H, len(PLANES)), flush=True)
g.check(); g.close()
print("huge image check ok: %d x %d" % (N, N))

import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, limg_amd
g = limg_amd.LimgHip(0)
n = 16384
img = g.synth_device("photo_noise", n, n, seed=1)
planes = g.alloc_planes_device(n, n)
for i in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    g.encode3d_device(img, True, planes); torch.cuda.synchronize()
    print("8x8 path 16384^2: %.2f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
g.check()
psnr = g.compare_device(img, planes["pDecoded"], True)[0]
# strip-restart check: rows [0, 2048) of the whole-image encode with pool 2 == encode of that strip alone
g.encode3d_device(img, True, planes, pool_threads=2); torch.cuda.synchronize()
p2 = g.alloc_planes_device(n, 2048)
g.encode3d_device(img[:2048], True, p2); torch.cuda.synchronize()
print("psnr %.4f  strip0 equal:" % psnr, all(torch.equal(planes[k][:2048], p2[k]) for k in p2), flush=True)
del planes, p2
bp = g.alloc_blocked_planes_device(n, n)
for i in range(2):
    t = time.perf_counter(); g.blocked_encode3d_device(img, True, bp); torch.cuda.synchronize()
    print("merged-block 16384^2: %.1f ms" % ((time.perf_counter() - t) * 1e3), g.blocked_timing(), len(g.blocked_regions()), flush=True)
print("psnr blocked %.4f" % g.compare_device(img, bp["pDecoded"], True)[0])
g.check(); g.close()
